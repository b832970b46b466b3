"""Parity of the HIP solve (K1) through the C-ABI against the reference's golden
vectors and the oracle.  Tolerance: 1e-6 norm-relative on the coefficients
(BASELINE.json north_star); observed errors are ~1e-12 and the tighter bound
1e-9 is asserted too so a regression is visible long before the gate."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, norm_rel

pytestmark = pytest.mark.gpu

TOL = 1e-6        # north-star gate
TIGHT9 = 1e-9     # order 9: pinned by the 60-digit solves of tests/golden/order9_golden.npz (SURVEY.md 8c iii); the
                  # fp64 oracles are within 2e-11 of them on every fixture case, the HIP kernels within 1e-10
TIGHT = 1e-9      # what the algorithm actually delivers (regression tripwire)

SINGLE = ["cfg1", "testdata", "m1", "m2", "t0quirk", "path49"]
BATCH = ["cfg2", "cfg2s", "m20", "m3"]


@pytest.mark.parametrize("name", SINGLE)
def test_single_trajectory_golden(ctx7, golden, name):
    wp, t, ref = golden[name + "_wp"], golden[name + "_t"], golden[name + "_coef"]
    coef, dur, status = ctx7.solve_batch(wp[None], t[None])
    assert status[0] == 0
    err = norm_rel(coef[0], ref)
    assert err <= TOL, err
    assert err <= (1e-7 if name == "path49" else TIGHT), err
    np.testing.assert_array_equal(dur[0], golden[name + "_dur"])


@pytest.mark.parametrize("name", BATCH)
def test_batch_golden(ctx7, golden, name):
    wp, t, ref = golden[name + "_wp"], golden[name + "_t"], golden[name + "_coef"]
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    err = norm_rel(coef, ref)
    assert err <= TIGHT, err
    rdur = golden[name + "_dur"]
    np.testing.assert_array_equal(dur, rdur if rdur.ndim == 2 else np.broadcast_to(rdur, dur.shape))


# ---------------------------------------------------------------------------
# shapes: every tile remainder, every kernel variant
# ---------------------------------------------------------------------------
def _c_ref(wp, t, ncoef=8):
    import c_oracle
    coef, dur, info, _ = c_oracle.solve_batch(wp, t, ncoef=ncoef, faithful=False, n_threads=4)
    assert (info == 0).all()
    return coef, dur


@pytest.mark.parametrize("n", [1, 3, 15, 16, 17, 33, 100])
def test_partial_tiles(ctx7, n):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(11, n, 10)
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    ref, rdur = _c_ref(wp, t)
    assert norm_rel(coef, ref) <= TIGHT
    np.testing.assert_array_equal(dur, rdur)


@pytest.mark.parametrize("m", [1, 2, 3, 5, 11, 12, 13, 14, 20, 37, 49])
def test_segment_counts_register_and_lds_variants(ctx7, m):
    """A 21-drone batch takes the small-batch two-sided kernel up to 24 segments and the rolled
    LDS-stash kernel above; the register-resident throughput kernel is reached by
    test_twisted_and_one_sided_kernels_agree (no_twist) and, in its multi-tile regime, by
    test_persistent_solve_walks_several_tiles."""
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(12 + m, 21, m)
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    ref, rdur = _c_ref(wp, t)
    assert norm_rel(coef, ref) <= 1e-8
    np.testing.assert_array_equal(dur, rdur)


def test_long_path_global_scratch_variant(ctx7):
    """150 segments exceed 160 KiB of LDS: the global-slab variant must agree too."""
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(99, 19, 150)
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    ref, rdur = _c_ref(wp, t)
    assert norm_rel(coef, ref) <= 1e-8
    np.testing.assert_array_equal(dur, rdur)


def test_shared_time_grid_equals_per_drone_grid(ctx7):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(5, 40, 10, shared_times=True)
    c1, d1, s1 = ctx7.solve_batch(wp, t)
    c2, d2, s2 = ctx7.solve_batch(wp, np.broadcast_to(t, (40, 11)).copy())
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(d1, d2)


def test_sharded_solve_is_bitwise_the_unsharded_solve(ctx7):
    from drone_path_planning_python_amd import swarm as sw
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(6, 203, 10)
    full, _, _ = ctx7.solve_batch(wp, t)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = sw.shard_bounds(203, world, r)
            parts.append(ctx7.solve_batch(wp[lo:hi], t[lo:hi])[0])
        np.testing.assert_array_equal(np.concatenate(parts), full)


# ---------------------------------------------------------------------------
# per-drone status instead of exceptions
# ---------------------------------------------------------------------------
def test_status_codes_and_nan_fill(ctx7):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(7, 20, 6)
    t[3, 4] = t[3, 3]              # zero-length segment
    t[5, 2] = t[5, 1] - 0.1        # decreasing
    wp[8, 2, 1] = np.nan
    t[9, 5] = np.inf
    wp[11, 0, 3] = np.inf
    coef, dur, status = ctx7.solve_batch(wp, t)
    expect = np.zeros(20, dtype=np.int32)
    expect[[3, 5]] = 2
    expect[[8, 9, 11]] = 3
    np.testing.assert_array_equal(status, expect)
    assert np.isnan(coef[[3, 5, 8, 9, 11]]).all()
    good = status == 0
    ref, _ = _c_ref(wp[good], t[good])
    assert norm_rel(coef[good], ref) <= TIGHT


def test_first_time_nonzero_quirk_batch(ctx7):
    """Appendix A: the reference evaluates the start rows at t[0] -- kept."""
    import msnap_oracle as O
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(8, 9, 5)
    t = t + np.linspace(0.0, 0.2, 9)[:, None]
    t[:, 1:] += 0.5
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    ref, rdur = O.solve_batch_fast(wp, t)
    assert norm_rel(coef, ref) <= TIGHT
    np.testing.assert_array_equal(dur, rdur)


def test_error_codes(ctx7):
    from drone_path_planning_python_amd import MsnapError
    with pytest.raises(MsnapError) as e:
        ctx7.solve_batch(np.zeros((2, 5000, 4)), np.zeros((2, 5000)))
    assert e.value.code == -4
    with pytest.raises(ValueError):
        ctx7.solve_batch(np.zeros((2, 5, 3)), np.zeros((2, 5)))
    c, d, s = ctx7.solve_batch(np.zeros((0, 5, 4)), np.zeros((0, 5)))
    assert c.shape == (0, 4, 4, 8) and s.shape == (0,)


# ---------------------------------------------------------------------------
# order 9 (no reference implementation: calculatingTrajectories.py:48-49,63 hard-code 8 coefficients).
# Pinned by extended precision instead: tests/golden/order9_golden.npz holds 60-digit
# mpmath solutions of the generalised collocation system, cross-checked there against the order-9 KKT / QP
# formulation (tests/golden/make_order9_golden.py); the oracle's generalisation reduces to the pinned
# order-7 system.
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gold9():
    return np.load(os.path.join(GOLDEN_DIR, "order9_golden.npz"))


ORDER9_CASES = ["m2", "m3", "m4", "m5", "m6", "m7", "m8", "m9", "m10", "m10s", "m10q", "m12", "m16", "m20", "m20s"]


@pytest.mark.parametrize("name", ORDER9_CASES)
def test_order9_against_extended_precision(gold9, name):
    """Every order-9 kernel family against the 60-digit fixture at <= 1e-9: the small-batch two-sided
    kernel, the throughput kernels (two-sided column-split for even segment counts <= 10, the one-sided
    register kernel otherwise), the shared-grid MFMA GEMM for the shared-grid cases."""
    from drone_path_planning_python_amd import Context
    wp, t, ref = gold9[name + "_wp"], gold9[name + "_t"], gold9[name + "_coef"]
    M = wp.shape[1] - 1
    errs = {}
    with Context(order=9, max_segments=64) as ctx:
        coef, dur, status = ctx.solve_batch(wp, t)
        assert (status == 0).all()
        errs[ctx.last_kernel()] = norm_rel(coef, ref)
        np.testing.assert_array_equal(dur, np.broadcast_to(np.diff(t), dur.shape))
        ctx.set_option("no_twist", 1)
        coef, _, status = ctx.solve_batch(wp, t)
        assert (status == 0).all()
        errs[ctx.last_kernel()] = norm_rel(coef, ref)
        ctx.set_option("no_twin", 1)
        coef, _, status = ctx.solve_batch(wp, t)
        assert (status == 0).all()
        errs[ctx.last_kernel()] = norm_rel(coef, ref)
        if t.ndim == 1:
            ctx.prepare_grid(t)
            coef, _, status = ctx.solve_grid(wp)
            assert (status == 0).all()
            errs[ctx.last_kernel()] = norm_rel(coef, ref)
    print(name, {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs and max(errs.values()) <= TIGHT9, errs
    if 4 <= M <= 20:      # (13..20 segments: the one-wave-per-SIMD instances of round 4)
        assert any("solve_kernel_twin<5" in k for k in errs), errs


@pytest.mark.parametrize("m", [1, 2, 7, 10, 12, 16])
def test_order9_against_generalised_oracle(ctx9, m):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(40 + m, 18, m)
    coef, dur, status = ctx9.solve_batch(wp, t)
    assert (status == 0).all()
    assert coef.shape == (18, m, 4, 10)
    ref, rdur = _c_ref(wp, t, ncoef=10)
    assert norm_rel(coef, ref) <= TIGHT9
    np.testing.assert_array_equal(dur, rdur)


@pytest.mark.parametrize("m", [13, 14, 15, 16, 17, 18, 19, 20])
def test_order9_long_paths_inside_a_guarded_arena(m):
    """solve_kernel_twin<5, 13..20> (one wave per SIMD, up to 146 accumulation registers as spill space) with every
    device buffer inside a pattern-filled arena: no byte outside the outputs changes, and the outputs equal the
    one-sided register kernel's and the C oracle's -- for whole tiles, every partial-tile size, one tile per wave and
    several tiles per wave (the cross-tile prefetch).  The first build of these instances read its prefetch indices
    from accumulation registers that a copy under exec == 0 had never written: `Memory access fault by GPU` at
    base + 16 * garbage, even for 8 drones (tools/check_exec_isa.py is the build-time tripwire)."""
    import torch
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    dev = torch.device("cuda", 0)
    PAT, GAP = 0xA5, 1 << 16
    with Context(order=9, max_segments=64) as ctx, Context(order=9, max_segments=64) as ref_ctx:
        ref_ctx.set_option("no_twist", 1)
        ref_ctx.set_option("no_twin", 1)
        for n, waves in ((1, 0), (4, 0), (7, 0), (8, 0), (9, 0), (17, 0), (100, 3), (1029, 16)):
            wp, t = swarm(900 + 31 * m + n, n, m)
            sizes = {"wp": wp.nbytes, "t": t.nbytes, "coef": n * m * 40 * 8, "dur": n * m * 8, "status": n * 4}
            off, cur = {}, 4 << 20
            for k in sizes:
                off[k] = cur
                cur += (sizes[k] + GAP + 255) & ~255
            arena = torch.full((cur + (4 << 20),), PAT, dtype=torch.uint8, device=dev)
            view = lambda k, dt: arena[off[k]:off[k] + sizes[k]].view(dt)      # noqa: E731
            view("wp", torch.float64).copy_(torch.from_numpy(wp.reshape(-1)))
            view("t", torch.float64).copy_(torch.from_numpy(t.reshape(-1)))
            torch.cuda.synchronize()
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.set_option("solve_grid_waves", waves)
            base = arena.data_ptr()
            ctx.solve_batch_device(n, m, base + off["wp"], base + off["t"], False, base + off["coef"],
                                   base + off["dur"], base + off["status"])
            torch.cuda.synchronize()
            assert ctx.last_kernel() == f"msnap::solve_kernel_twin<5, {m}>"
            ctx.use_own_stream()
            mask = arena != PAT
            for k in off:
                mask[off[k]:off[k] + sizes[k]] = False
            assert not bool(mask.any()), f"stray writes at {torch.nonzero(mask).flatten()[:8].tolist()} (n = {n})"
            coef = view("coef", torch.float64).cpu().numpy().reshape(n, m, 4, 10)
            assert not view("status", torch.int32).any()
            np.testing.assert_array_equal(view("dur", torch.float64).cpu().numpy().reshape(n, m), np.diff(t, axis=1))
            rcoef, _, rstatus = ref_ctx.solve_batch(wp, t)
            assert ref_ctx.last_kernel() == "msnap::solve_kernel_reg<5, 20>" and (rstatus == 0).all()
            assert norm_rel(coef, rcoef) <= TIGHT9, (n, waves)
            oref, _ = _c_ref(wp, t, ncoef=10)
            assert norm_rel(coef, oref) <= TIGHT9, (n, waves)


def test_long_lived_context_after_many_short_lived_ones(ctx9, gold9):
    """The sequence of round 3's one abort (gpurun_out/r3w_pytest.log): a long-lived context on its
    OWN stream has run small instances; fifteen short-lived contexts are created, run every kernel family of their
    segment count and are destroyed (streams, staging, bounce buffers and events freed each time); then the long-lived
    context launches, for the first time, the instances of 13..20 segments -- from host pointers, then from device
    pointers on a borrowed torch stream.  Every result is checked.  (Round 4 found what the abort was: a GPU memory
    fault whose message pytest's fd capture had swallowed, caused by compiler-placed spill copies under exec == 0 behind
    a loop with a divergent trip count -- DESIGN.md 9.3; tools/check_exec_isa.py and
    test_order9_long_paths_inside_a_guarded_arena are the guards, this sequence stays as the historical regression.)"""
    import torch
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(71, 18, 4)
    assert (ctx9.solve_batch(wp, t)[2] == 0).all()
    for name in ORDER9_CASES:
        wpn, tn = gold9[name + "_wp"], gold9[name + "_t"]
        with Context(order=9, max_segments=64) as ctx:
            for opt in (None, "no_twist", "no_twin"):
                if opt:
                    ctx.set_option(opt, 1)
                assert (ctx.solve_batch(wpn, tn)[2] == 0).all()
            big_wp, big_t = swarm(72, 3000, wpn.shape[1] - 1)
            assert (ctx.solve_batch(big_wp, big_t)[2] == 0).all()      # the staged (non-bounce) host path too
    for m in (16, 13, 20):
        wp, t = swarm(40 + m, 18, m)
        coef, dur, status = ctx9.solve_batch(wp, t)
        assert (status == 0).all()
        ref, rdur = _c_ref(wp, t, ncoef=10)
        assert norm_rel(coef, ref) <= TIGHT9
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            ctx9.set_stream(side.cuda_stream)
            for m, n in ((16, 18), (20, 5000)):
                wp, t = swarm(50 + m, n, m)
                dwp, dt_ = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
                coef = torch.empty((n, m, 4, 10), dtype=torch.float64, device=dev)
                dur = torch.empty((n, m), dtype=torch.float64, device=dev)
                status = torch.empty((n,), dtype=torch.int32, device=dev)
                ctx9.solve_batch_device(n, m, dwp, dt_, False, coef, dur, status)
                side.synchronize()
                assert int(status.abs().sum()) == 0
                ref, _ = _c_ref(wp[:64], t[:64], ncoef=10)
                assert norm_rel(coef[:64].cpu().numpy(), ref) <= TIGHT9
    finally:
        ctx9.use_own_stream()


# ---------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full sizes
# ---------------------------------------------------------------------------
def _poly_derivs_at(c, x, nder):
    """values of d^j/dt^j, j = 0..nder-1, of ascending-power polynomials c[..., nc] at x[...]"""
    nc = c.shape[-1]
    out = []
    cur = c.copy()
    for j in range(nder):
        k = cur.shape[-1]
        pw = x[..., None] ** np.arange(k)
        out.append((cur * pw).sum(-1))
        cur = cur[..., 1:] * np.arange(1, k)
    return np.stack(out, -1)


@pytest.mark.parametrize("cfg,n,m,order", [(2, 256, 10, 7), (3, 4096, 10, 7), (4, 4096, 20, 7), (5, 65536, 10, 9)])
def test_full_size_properties(ctx7, ctx9, cfg, n, m, order):
    """interpolation, zero end derivatives, C^(2k-2) continuity at every knot."""
    from drone_path_planning_python_amd.synthetic import swarm
    ctx = ctx7 if order == 7 else ctx9
    wp, t = swarm(cfg, n, m)
    coef, dur, status = ctx.solve_batch(wp, t)
    assert (status == 0).all()
    k = (order + 1) // 2
    wpt = wp.transpose(0, 2, 1)                          # [N,4,m+1]
    c = coef.transpose(0, 2, 1, 3)                       # [N,4,M,nc]
    scale = np.abs(wp).max()
    # starts: c0 = w_i exactly
    np.testing.assert_array_equal(c[..., 0], wpt[..., :-1])
    T = np.broadcast_to(dur[:, None, :], c.shape[:-1])
    end = _poly_derivs_at(c, T, 2 * k - 1)               # [N,4,M,2k-1]
    start = _poly_derivs_at(c, np.zeros_like(T), 2 * k - 1)
    assert np.abs(end[..., 0] - wpt[..., 1:]).max() <= 1e-9 * scale
    # C^1..C^(2k-2) at interior knots, relative to the derivative's own magnitude
    for j in range(1, 2 * k - 1):
        num = np.abs(end[:, :, :-1, j] - start[:, :, 1:, j]).max(axis=(1, 2))
        den = np.maximum(np.abs(start[..., j]).max(axis=(1, 2)), 1e-30)
        assert (num / den).max() <= 1e-7, (j, (num / den).max())
    # zero velocity .. d^(k-1) at both ends
    for j in range(1, k):
        den = np.maximum(np.abs(start[..., j]).max(axis=(1, 2)), 1e-30)
        assert (np.abs(start[:, :, 0, j]).max(axis=1) / den).max() <= 1e-9
        assert (np.abs(end[:, :, -1, j]).max(axis=1) / den).max() <= 1e-7


# ---------------------------------------------------------------------------
# the reference-shaped Python API on top
# ---------------------------------------------------------------------------
def test_calculate_trajectory4D_api(golden):
    from drone_path_planning_python_amd.optimizations import (PiecewisePolynomial, Point_time, Polynomial, Waypoint,
                                                            calculate_trajectory4D)
    from drone_path_planning_python_amd.optimizations.calculatingTrajectories import calculate_trajectory1D
    wp, t = golden["cfg1_wp"], golden["cfg1_t"]
    pts = [Point_time(Waypoint(*w), float(tt)) for w, tt in zip(wp, t)]
    pols, pcs = calculate_trajectory4D(pts)
    assert len(pols) == 4 and len(pcs) == 4 and len(pols[0]) == 3
    assert isinstance(pols[0][0], Polynomial) and isinstance(pcs[0], PiecewisePolynomial)
    assert pols[2][1].p.shape == (8, 1) and pols[2][1].p.dtype == np.float64
    assert pcs[0].time_durations == [1.0, 2.0, 1.0] and all(isinstance(x, float) for x in pcs[0].time_durations)
    got = np.array([[pols[a][j].p.ravel() for a in range(4)] for j in range(3)])
    assert norm_rel(got, golden["cfg1_coef"]) <= TIGHT
    for tt, ref in zip(golden["pweval_t"], golden["pweval_val"]):
        for a in range(4):
            v = float(np.ravel(pcs[a].eval(float(tt)))[0])
            assert abs(v - ref[a]) <= 1e-9 * max(1.0, abs(ref[a]))
    px, pcx = calculate_trajectory1D(pts, Waypoint.WP_TYPE_Z)
    np.testing.assert_array_equal(px[1].p, pols[2][1].p)
    # singular input -> the reference's exception type
    bad = [Point_time(Waypoint(0, 0, 0, 0), 0.0), Point_time(Waypoint(1, 1, 1, 0), 1.0),
           Point_time(Waypoint(2, 1, 0, 0), 1.0)]
    with pytest.raises(np.linalg.LinAlgError):
        calculate_trajectory4D(bad)


def test_testdata_demo_known_answer(golden):
    """The reference's own __main__ demo (calculatingTrajectories.py:240-273)."""
    from drone_path_planning_python_amd.optimizations import Point_time, Waypoint
    from drone_path_planning_python_amd.optimizations.calculatingTrajectories import calculate_trajectory1D, timestep
    pts = [Point_time(Waypoint(*p), t=i * timestep) for i, p in enumerate(golden["testdata_wp"])]
    pols, total = calculate_trajectory1D(pts, Waypoint.WP_TYPE_X)
    assert abs(float(np.ravel(total.eval(17.0))[0]) - (-0.28721862765096506)) < 1e-10
    np.testing.assert_allclose(np.ravel(pols[0].p)[4:], [3.0270487768666903e-02, -2.0840517637741526e-02,
                                                        5.1579487927348776e-03, -4.5360490807955125e-04], rtol=1e-8)


# ---------------------------------------------------------------------------
# K2: shared time grid -> one fp64 MFMA GEMM
# ---------------------------------------------------------------------------
def test_grid_gemm_golden_shared_grid(ctx7, golden):
    """cfg2s: 64 drones on the reference's uniform grid, reference outputs."""
    wp, t, ref = golden["cfg2s_wp"], golden["cfg2s_t"], golden["cfg2s_coef"]
    ctx7.prepare_grid(t)
    coef, dur, status = ctx7.solve_grid(wp)
    assert (status == 0).all()
    assert norm_rel(coef, ref) <= TIGHT
    np.testing.assert_array_equal(dur, np.broadcast_to(golden["cfg2s_dur"], dur.shape))
    np.testing.assert_array_equal(coef[:, :, :, 0], wp[:, :-1, :].transpose(0, 1, 2))   # c0 == w_i exactly
    k1, _, _ = ctx7.solve_batch(wp, t)
    assert norm_rel(coef, k1) <= 1e-11


@pytest.mark.parametrize("n,m", [(1, 10), (3, 10), (5, 1), (7, 2), (4, 3), (130, 12), (33, 13), (9, 20)])
def test_grid_gemm_shapes(ctx7, n, m):
    """row-tile remainders, odd column-tile remainders, and (m >= 13) the K1 fallback."""
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(60 + m, n, m, shared_times=True)
    t = t * np.linspace(0.7, 1.3, m + 1).cumsum() / np.linspace(0.7, 1.3, m + 1).cumsum()[-1] * 1.3 + t * 0.2
    t[0] = 0.0
    ctx7.prepare_grid(t)
    coef, dur, status = ctx7.solve_grid(wp)
    assert (status == 0).all()
    ref, rdur = _c_ref(wp, t)
    assert norm_rel(coef, ref) <= 1e-8
    np.testing.assert_array_equal(dur, rdur)


def test_grid_gemm_order9(ctx9):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(77, 21, 10, shared_times=True)
    ctx9.prepare_grid(t)
    coef, dur, status = ctx9.solve_grid(wp)
    assert (status == 0).all()
    ref, rdur = _c_ref(wp, t, ncoef=10)
    assert norm_rel(coef, ref) <= 1e-6
    k1, _, _ = ctx9.solve_batch(wp, t)
    assert norm_rel(coef, k1) <= 1e-9


def test_grid_status_and_errors():
    from drone_path_planning_python_amd import Context, MsnapError
    from drone_path_planning_python_amd.synthetic import swarm
    with Context(order=7, max_segments=64) as ctx:
        wp, t = swarm(78, 10, 6, shared_times=True)
        with pytest.raises(MsnapError) as e:
            ctx.solve_grid(wp)
        assert e.value.code == -7
        ctx.prepare_grid(t)
        wp[2, 3, 1] = np.nan
        wp[7, 0, 3] = np.inf
        coef, dur, status = ctx.solve_grid(wp)
        expect = np.zeros(10, dtype=np.int32)
        expect[[2, 7]] = 3
        np.testing.assert_array_equal(status, expect)
        assert np.isnan(coef[[2, 7]]).all() and np.isfinite(coef[status == 0]).all()
        bad_t = t.copy()
        bad_t[3] = bad_t[2]
        ctx.prepare_grid(bad_t)
        wp, _ = swarm(78, 10, 6, shared_times=True)
        coef, dur, status = ctx.solve_grid(wp)
        assert (status == 2).all() and np.isnan(coef).all()


# ---------------------------------------------------------------------------
# small-batch two-sided ("twisted") kernel vs the one-sided kernels
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("order,m", [(7, m) for m in (2, 3, 4, 5, 9, 10, 11, 12, 13, 17, 20, 24)] +
                         [(9, m) for m in (2, 3, 6, 9, 10, 12)])
def test_twisted_and_one_sided_kernels_agree(order, m, monkeypatch):
    """Small batches take solve_kernel_twist (one instance per segment count); MSNAP_NO_TWIST=1
    keeps them on the one-sided kernels (register-resident up to 20 segments, LDS stash above).
    Both must match the oracle, and each other to ~1e-11."""
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(500 + m, 37, m)
    t[5] += 0.3                     # one drone with t[0] != 0 (quirk on the start side)
    with Context(order=order, max_segments=64) as ctx:
        c_tw, d_tw, s_tw = ctx.solve_batch(wp, t)
    with Context(order=order, max_segments=64) as ctx:
        ctx.set_option("no_twist", 1)
        c_os, d_os, s_os = ctx.solve_batch(wp, t)
    assert (s_tw == 0).all() and (s_os == 0).all()
    import msnap_oracle as O
    ref, rdur = O.solve_batch_fast(wp, t, ncoef=order + 1)
    tol = 1e-9                               # both orders (order 9: TIGHT9, see the order-9 fixture)
    assert norm_rel(c_tw, ref) <= tol
    assert norm_rel(c_os, ref) <= tol
    agree = norm_rel(c_tw, c_os)
    print(f"order {order} m {m}: two-sided vs one-sided {agree:.2e}, vs oracle {norm_rel(c_tw, ref):.2e} / {norm_rel(c_os, ref):.2e}")
    assert agree <= (1e-11 if order == 7 else 1e-9), agree
    np.testing.assert_array_equal(d_tw, rdur)
    np.testing.assert_array_equal(d_os, rdur)
    keep = np.arange(37) != 5                                        # drone 5 has the t[0] quirk: c0 != w_0 there
    np.testing.assert_array_equal(c_tw[keep][..., 0], wp[keep, :-1, :])   # c0 == w_i exactly on both sides


def test_failed_drones_in_small_batches_are_nan_filled():
    """Status codes and NaN fill of the two-sided kernel, for failures on either side of the path
    and at the meeting knot."""
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    M = 10
    wp, t = swarm(77, 24, M)
    t[1, 2] = t[1, 1]                 # side 0: zero-length segment
    t[2, 9] = t[2, 8] - 0.1           # side 1: decreasing time
    wp[3, 5, 2] = np.nan              # waypoint at the meeting knot
    wp[4, 10, 0] = np.inf             # last waypoint
    t[6, 0] = -1.0                    # negative start time
    with Context(order=7, max_segments=16) as ctx:
        coef, dur, status = ctx.solve_batch(wp, t)
    expect = np.zeros(24, np.int32)
    expect[[1, 2, 6]] = 2
    expect[[3, 4]] = 3
    np.testing.assert_array_equal(status, expect)
    for d in range(24):
        if expect[d]:
            assert np.isnan(coef[d]).all(), d
        else:
            assert np.isfinite(coef[d]).all(), d
    ok = expect == 0
    import msnap_oracle as O
    ref, _ = O.solve_batch_fast(wp[ok], t[ok])
    assert norm_rel(coef[ok], ref) <= 1e-9


# ---------------------------------------------------------------------------
# robustness: wide ranges of segment lengths and coordinates (no pivoting, rcp + Newton)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("scale_t,scale_w", [(1e-3, 1.0), (1.0, 1e6), (50.0, 1e-4), (1e-2, 1e3)])
def test_extreme_scales(ctx7, scale_t, scale_w):
    """Very short / long segments and large / tiny coordinates: the solve is scale-covariant,
    c_k(scaled) = c_k * scale_w / scale_t^k, so the check is independent of the oracle's own
    conditioning (the dense LU degrades long before the block LDL^T does)."""
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(600, 24, 10)
    base, dur0, st0 = ctx7.solve_batch(wp, t)
    coef, dur, st = ctx7.solve_batch(wp * scale_w, t * scale_t)
    assert (st0 == 0).all() and (st == 0).all()
    k = np.arange(8)
    # compare in the units of the base problem (the raw coefficients span 20+ decades)
    assert norm_rel(coef * scale_t ** k / scale_w, base) <= 1e-9
    np.testing.assert_allclose(dur, dur0 * scale_t, rtol=1e-12)


@pytest.mark.parametrize("seed", range(4))
def test_uneven_segment_lengths(ctx7, seed):
    """Segment lengths spanning 1:200 inside one path (the case where an unpivoted, unscaled
    factorisation would be at risk), against the extended-precision reference solve."""
    rng = np.random.default_rng(700 + seed)
    N, M = 6, 8
    wp = rng.uniform(-5, 5, size=(N, M + 1, 4))
    T = 10.0 ** rng.uniform(-1.3, 1.0, size=(N, M))
    t = np.concatenate([np.zeros((N, 1)), np.cumsum(T, axis=1)], axis=1)
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    import msnap_oracle as O
    for d in range(N):
        for a in range(4):
            A, b = O.assemble_1d(t[d], wp[d, :, a])
            x = np.linalg.solve(A.astype(np.longdouble).astype(np.float64), b)
            # iterative refinement in long double pins the reference solution itself
            Al, bl = A.astype(np.longdouble), b.astype(np.longdouble)
            xl = x.astype(np.longdouble)
            for _ in range(4):
                r = bl - Al @ xl
                xl = xl + np.linalg.solve(A, r.astype(np.float64)).astype(np.longdouble)
            ref = xl.astype(np.float64).reshape(M, 8)
            num = np.abs(coef[d, :, a, :] - ref).max()
            # observed worst case 1.5e-8 (1:200 length ratios); the gate is the north-star's 1e-6
            assert num / np.abs(ref).max() <= 1e-6, (d, a, num / np.abs(ref).max())


# ---------------------------------------------------------------------------
# host-pointer boundary: chunked upload / kernel / download pipeline, page-locked buffers
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("n", [4096 + 1, 3 * 4096, 5 * 4096 + 777])
def test_chunked_host_solve_is_bitwise_the_single_shot_solve(n, shared, monkeypatch):
    """Batches above one chunk are cut into chunks of drones that alternate between two
    streams.  With the smallest chunk (4096 drones) the result must be bit-identical to the
    device-pointer solve of the whole batch, for pageable and for page-locked host arrays."""
    import torch
    from drone_path_planning_python_amd import Context, pinned_empty
    from drone_path_planning_python_amd.synthetic import swarm
    M = 6
    wp, t = swarm(900, n, M)
    if shared:
        t = np.ascontiguousarray(t[0])
    with Context(order=7, max_segments=16) as ctx:
        ctx.set_option("pipe_chunk_mb", 1)   # 1 MB / 1536 B per drone < 4096 -> chunk = 4096 drones
        ctx.set_option("no_twist", 1)        # chunks and whole batch on the same kernel variant
        dev = torch.device("cuda:0")
        dwp, dt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
        dcoef = torch.empty((n, M, 4, 8), dtype=torch.float64, device=dev)
        ddur = torch.empty((n, M), dtype=torch.float64, device=dev)
        dst = torch.empty((n,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.solve_batch_device(n, M, dwp, dt, shared, dcoef, ddur, dst)
        ctx.sync()
        ref = (dcoef.cpu().numpy(), ddur.cpu().numpy(), dst.cpu().numpy())
        got = ctx.solve_batch(wp, t)                                  # pageable in and out
        for g, r in zip(got, ref):
            np.testing.assert_array_equal(g, r)
        pwp, pt = pinned_empty(wp.shape), pinned_empty(t.shape)
        pwp[...] = wp
        pt[...] = t
        out = (pinned_empty((n, M, 4, 8)), pinned_empty((n, M)), pinned_empty((n,), np.int32))
        for a in out:
            a.view(np.uint8)[...] = 0xA5
        res = ctx.solve_batch(pwp, pt, out=out)                       # page-locked in and out
        assert all(r is o for r, o in zip(res, out))
        for g, r in zip(out, ref):
            np.testing.assert_array_equal(g, r)
        # the shared-grid GEMM path goes through the same pipeline
        if shared:
            ctx.prepare_grid(t)
            gc, gd, gs = ctx.solve_grid(pwp, out=out)
            assert (gs == 0).all() and norm_rel(gc, ref[0]) <= 1e-11
            np.testing.assert_array_equal(gd, ref[1])


def test_out_argument_is_validated(ctx7):
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(901, 5, 4)
    good = (np.empty((5, 4, 4, 8)), np.empty((5, 4)), np.empty((5,), np.int32))
    ctx7.solve_batch(wp, t, out=good)
    for bad in [(good[0], good[1]),
                (np.empty((5, 4, 4, 8), np.float32), good[1], good[2]),
                (np.empty((5, 4, 4, 9)), good[1], good[2]),
                (good[0], good[1], np.empty((5,), np.int64)),
                (np.empty((8, 4, 5, 4)).transpose(3, 1, 2, 0)[:5], good[1], good[2])]:
        with pytest.raises(ValueError):
            ctx7.solve_batch(wp, t, out=bad)


def test_pinned_arrays_outlive_their_views():
    import gc
    from drone_path_planning_python_amd import pinned_empty
    a = pinned_empty((128, 3))
    a[...] = 1.5
    row = a[7]
    del a
    gc.collect()
    assert (row == 1.5).all()
    assert pinned_empty((0, 4)).shape == (0, 4)


# ---------------------------------------------------------------------------
# seeded sweep over shapes: every kernel-selection branch against the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_against_oracle(ctx7, ctx9, seed):
    """Random batch size, segment count, order, shared / per-drone grid, start-time quirk: whatever
    kernel variant the launcher picks must reproduce the oracle's dense solve."""
    import msnap_oracle as O
    rng = np.random.default_rng(9000 + seed)
    order = 7 if seed % 3 else 9
    ctx = ctx7 if order == 7 else ctx9
    M = int(rng.integers(1, 27 if order == 7 else 15))
    N = int(rng.choice([1, 2, 7, 8, 9, 16, 31, 64, 65, 130]))
    shared = bool(rng.integers(0, 2))
    wp = rng.uniform(-5, 5, size=(N, M + 1, 4))
    wp[..., 3] = rng.uniform(-np.pi, np.pi, size=(N, M + 1))
    T = rng.uniform(0.4, 2.5, size=(1 if shared else N, M))
    t = np.concatenate([np.zeros((T.shape[0], 1)), np.cumsum(T, axis=1)], axis=1)
    if not shared and N > 2:
        t[1] += 0.25                       # the reference's start-row quirk on one drone
    tt = t[0] if shared else t
    coef, dur, status = ctx.solve_batch(wp, tt)
    assert (status == 0).all()
    ref, rdur = O.solve_batch_fast(wp, np.broadcast_to(t, (N, M + 1)) if shared else t, ncoef=order + 1)
    err = norm_rel(coef, ref)
    assert err <= (TIGHT if order == 7 else TIGHT9), (order, N, M, shared, err)
    np.testing.assert_array_equal(dur, rdur)


# ---------------------------------------------------------------------------
# the persistent throughput kernels in the regime bench.py's saturated legs time:
# every wave walks SEVERAL tiles (cross-tile asm prefetch retired by a hand-counted
# s_waitcnt vmcnt in solve_kernel_reg, the `rt += gridDim.x` loop of the shared-grid GEMMs)
# ---------------------------------------------------------------------------
def _solve_in_shards(ctx, wp, t, shard):
    parts = []
    for lo in range(0, wp.shape[0], shard):
        tt = t if t.ndim == 1 else t[lo:lo + shard]
        parts.append(ctx.solve_batch(wp[lo:lo + shard], tt))
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("order,m,waves,twin", [(7, 10, 5, False), (7, 7, 3, False), (7, 20, 5, False), (7, 14, 4, False),
                                                (9, 10, 5, False), (9, 16, 3, False), (9, 20, 4, False),
                                                (7, 10, 5, True), (7, 4, 3, True), (7, 12, 7, True), (7, 18, 6, True),
                                                (9, 10, 5, True), (9, 6, 4, True), (9, 12, 5, True),
                                                (7, 5, 3, True), (7, 7, 4, True), (7, 13, 5, True), (7, 17, 6, True),
                                                (7, 19, 5, True), (7, 20, 6, True),
                                                (9, 5, 3, True), (9, 9, 5, True), (9, 11, 4, True)])
def test_persistent_solve_walks_several_tiles(order, m, waves, shared, twin):
    """The persistent throughput kernels with a grid of `waves` wavefronts -- solve_kernel_reg<4|5, 10|20> (16 drones
    per tile: 600-odd drones are 38 tiles) and the two-sided solve_kernel_twin<4|5, M> (8 per tile: 76 tiles) -- so
    every wave walks 7-25 tiles and the last tile is partial.  The result must equal, bit for
    bit, the same batch solved 16 drones (one or two tiles, no prefetch wait) at a time, and match the C oracle."""
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    n = 16 * 37 + 9
    wp, t = swarm(4000 + 10 * m + order, n, m, shared_times=shared)
    if not shared:
        t[3] += 0.2                                      # one drone with the t[0] != 0 quirk
    with Context(order=order, max_segments=64) as ctx:
        ctx.set_option("no_twist", 1)                   # keep every launch on the throughput kernels
        ctx.set_option("no_twin", 0 if twin else 1)
        one_tile = _solve_in_shards(ctx, wp, t, 16)      # grid = 1 wave, one tile: the prefetch never waits
        assert ("solve_kernel_twin" in ctx.last_kernel()) == twin, ctx.last_kernel()
        ctx.set_option("solve_grid_waves", waves)
        assert ctx.get_option("solve_grid_waves") == waves
        multi = ctx.solve_batch(wp, t)
        # a failing drone in the middle of a wave's sequence must not disturb its neighbours' tiles
        wpb = wp.copy()
        wpb[16 * 11 + 5, 2, 1] = np.nan
        bad = ctx.solve_batch(wpb, t)
    assert (multi[2] == 0).all()
    for a, b in zip(multi, one_tile):
        np.testing.assert_array_equal(a, b)
    ref, rdur = _c_ref(wp, t, ncoef=order + 1)
    assert norm_rel(multi[0], ref) <= (TIGHT if order == 7 else TIGHT9)
    np.testing.assert_array_equal(multi[1], rdur if rdur.ndim == 2 else np.broadcast_to(rdur, multi[1].shape))
    keep = np.arange(n) != 16 * 11 + 5
    assert bad[2][16 * 11 + 5] == 3 and np.isnan(bad[0][16 * 11 + 5]).all()
    np.testing.assert_array_equal(bad[0][keep], multi[0][keep])


@pytest.mark.parametrize("name,twin", [("cfg2", False), ("m20", False), ("cfg2", True)])
def test_persistent_solve_goldens_multi_tile(golden, name, twin):
    """The reference's own outputs (cfg2: 64 x 10, m20: 8 x 20, tiled to several tiles per wave)
    through the throughput kernels with two persistent waves: solve_kernel_reg, and for cfg2 also the
    two-sided solve_kernel_twin<4, 10>."""
    from drone_path_planning_python_amd import Context
    wp, t, ref = golden[name + "_wp"], golden[name + "_t"], golden[name + "_coef"]
    reps = 200 // wp.shape[0] + 1
    wpx, tx, refx = np.tile(wp, (reps, 1, 1)), np.tile(t, (reps, 1)), np.tile(ref, (reps, 1, 1, 1))
    with Context(order=7, max_segments=64) as ctx:
        ctx.set_option("no_twist", 1)
        ctx.set_option("no_twin", 0 if twin else 1)
        ctx.set_option("solve_grid_waves", 2)
        coef, dur, status = ctx.solve_batch(wpx, tx)
        assert ("solve_kernel_twin" in ctx.last_kernel()) == twin
    assert (status == 0).all()
    assert norm_rel(coef, refx) <= TIGHT
    np.testing.assert_array_equal(dur, np.tile(golden[name + "_dur"], (reps, 1)))


@pytest.mark.parametrize("order,m,waves", [(7, 10, 3), (7, 15, 5), (7, 1, 2), (9, 10, 3), (9, 12, 4),
                                           (7, 20, 3), (7, 49, 2), (9, 20, 3), (7, 63, 3)])
def test_persistent_grid_gemm_walks_several_tiles(order, m, waves):
    """K2 with a grid of `waves` wavefronts (register-resident operator up to 15 / 12 segments,
    streaming operator above): 523 drones are 131 row tiles incl. a partial one.  Bit-identical
    to the same batch in 4-drone shards (one row tile per launch), and equal to the K1 solve and
    the C oracle."""
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    n = 523
    wp, t = swarm(4100 + 10 * m + order, n, m, shared_times=True)
    with Context(order=order, max_segments=64) as ctx:
        ctx.prepare_grid(t)
        parts = [ctx.solve_grid(wp[lo:lo + 4]) for lo in range(0, n, 4)]
        one_tile = tuple(np.concatenate([p[k] for p in parts]) for k in range(3))
        ctx.set_option("gemm_grid_waves", waves)
        multi = ctx.solve_grid(wp)
        k1 = ctx.solve_batch(wp, t)
    assert (multi[2] == 0).all()
    for a, b in zip(multi, one_tile):
        np.testing.assert_array_equal(a, b)
    ref, rdur = _c_ref(wp, t, ncoef=order + 1)
    tol = 1e-8 if order == 7 else 1e-6
    assert norm_rel(multi[0], ref) <= tol
    assert norm_rel(multi[0], k1[0]) <= (1e-9 if m <= 20 else 1e-7)
    np.testing.assert_array_equal(multi[1], np.broadcast_to(rdur, multi[1].shape))


def test_streaming_gemm_large_batch_row_groups(ctx7):
    """>= 16384 drones take the 16-drones-per-wave (RT = 4) form of the streaming GEMM; 20 segments,
    shared grid, partial last row group, a few persistent waves.  Checked against K1 on every
    drone and against the C oracle on a sample."""
    from drone_path_planning_python_amd.synthetic import swarm
    n, m = 16384 + 37, 20
    wp, t = swarm(4242, n, m, shared_times=True)
    ctx7.prepare_grid(t)
    try:
        ctx7.set_option("gemm_grid_waves", 7)
        coef, dur, status = ctx7.solve_grid(wp)
    finally:
        ctx7.set_option("gemm_grid_waves", 0)
    k1, kd, _ = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    assert norm_rel(coef, k1) <= 1e-9
    np.testing.assert_array_equal(dur, kd)
    pick = np.r_[0:64, n - 64:n]
    ref, _ = _c_ref(wp[pick], t)
    assert norm_rel(coef[pick], ref) <= 1e-8


def test_reference_live_shape_through_the_mfma_path(ctx7, golden):
    """The reference's own workload: 50 poses -> 49 segments on the 10/n grid
    (scripts/drones_pols_generator.py:44-46; RB_planning_sep_coll_check.py:164 interpolates 50
    states).  golden path49 is the reference's output for such a path; it must come out of the
    shared-grid GEMM (streaming variant), including when several identical drones share a launch."""
    wp, t, ref = golden["path49_wp"], golden["path49_t"], golden["path49_coef"]
    ctx7.prepare_grid(t)
    for n in (1, 2, 9):
        coef, dur, status = ctx7.solve_grid(np.tile(wp[None], (n, 1, 1)))
        assert (status == 0).all()
        assert norm_rel(coef, np.tile(ref[None], (n, 1, 1, 1))) <= 1e-7
        np.testing.assert_array_equal(dur, np.tile(golden["path49_dur"][None], (n, 1)))
    # more than 63 segments: no operator is built, the call is the K1 solve on the stored grid
    from drone_path_planning_python_amd.synthetic import swarm
    wpl, tl = swarm(4343, 5, 70, shared_times=True)
    ctx7.prepare_grid(tl)
    c, d, s = ctx7.solve_grid(wpl)
    k, kd, _ = ctx7.solve_batch(wpl, tl)
    np.testing.assert_array_equal(c, k)
    np.testing.assert_array_equal(d, kd)


def test_options_api(ctx7):
    from drone_path_planning_python_amd import MsnapError
    with pytest.raises(MsnapError):
        ctx7.set_option("no_such_option", 1)
    with pytest.raises(MsnapError):
        ctx7.get_option("no_such_option")
    for name in ("solve_grid_waves", "gemm_grid_waves", "twist_max_drones", "no_twist", "collide_waves_per_cu"):
        assert ctx7.get_option(name) == 0
    assert ctx7.get_option("pipe_chunk_mb") == 64
    # the context's own stream at another priority: re-created, still the current one, still solves
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    wp, t = swarm(3, 40, 5)
    c = Context(ctx7.device_id, 7, 16)
    try:
        ref = c.solve_batch(wp, t)[0]
        for prio in (1, 2, 0):
            c.set_option("own_stream_priority", prio)
            assert c.get_option("own_stream_priority") == prio and c.stream() != 0
            np.testing.assert_array_equal(c.solve_batch(wp, t)[0], ref)
        with pytest.raises(MsnapError):
            c.set_option("own_stream_priority", 3)
    finally:
        c.close()


def test_two_threads_two_contexts_and_one_shared_context(golden):
    """rospy runs callback1 / callback2 on separate threads (scripts/drones_pols_generator.py:102-103).
    Independent contexts must run concurrently on their own streams, and one shared context must
    serialise its callers (the wrapper's lock): every call returns the reference's coefficients."""
    import threading
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.nodes import drones_pols_generator as dpg
    from drone_path_planning_python_amd.nodes import msgs
    wp49, ref49 = golden["path49_wp"], golden["path49_coef"]
    wp2, t2, ref2 = golden["cfg2_wp"], golden["cfg2_t"], golden["cfg2_coef"]
    errors = []

    def worker(ctx, kind, reps):
        try:
            for _ in range(reps):
                if kind == 0:      # the node's own call: a 50-pose path on the uniform grid (shared-grid GEMM)
                    quat = np.zeros((50, 4))
                    quat[:, 2] = np.sin(0.5 * wp49[:, 3])
                    quat[:, 3] = np.cos(0.5 * wp49[:, 3])
                    _, coef, _ = dpg.paths_to_pols([msgs.path_from_arrays(wp49[:, :3], quat)], ctx)
                    assert norm_rel(coef[0], ref49) <= 1e-7
                elif kind == 1:    # a batch on per-drone grids
                    coef, _, status = ctx.solve_batch(wp2, t2)
                    assert (status == 0).all() and norm_rel(coef, ref2) <= TIGHT
                else:              # another shared grid on the same context (prepare + solve must be one step)
                    coef, _, status = ctx.solve_on_grid(golden["cfg2s_t"], golden["cfg2s_wp"])
                    assert (status == 0).all() and norm_rel(coef, golden["cfg2s_coef"]) <= TIGHT
        except Exception as e:     # noqa: BLE001 (collected and re-raised on the main thread)
            errors.append(repr(e))

    with Context(order=7, max_segments=64) as a, Context(order=7, max_segments=64) as b:
        threads = [threading.Thread(target=worker, args=(c, k, 40))
                   for c, k in ((a, 0), (b, 1), (a, 1), (b, 0), (a, 2), (b, 2))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    assert not errors, errors


def test_last_kernel_reports_the_launched_instance():
    """msnap_last_kernel: the library names the kernel instance a solve launched (bench.py labels its
    rooflines with it instead of mirroring the launcher's thresholds)."""
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    with Context(order=7, max_segments=64) as ctx:
        assert ctx.last_kernel() == ""
        for n, m, want in [(256, 10, "msnap::solve_kernel_twist<4, 4, 10>"),
                           (300, 20, "msnap::solve_kernel_twist<4, 9, 20>"),
                           (64, 30, "msnap::solve_kernel<4, false>"),       # stash in LDS
                           (64, 49, "msnap::solve_kernel<4, true>")]:       # 167 KB of stash: the global slab
                                                                            # (round 2's bench guessed <4, false> here)
            wp, t = swarm(77, n, m)
            ctx.solve_batch(wp, t)
            assert ctx.last_kernel() == want
        ctx.set_option("no_twist", 1)
        wp, t = swarm(78, 300, 10)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_twin<4, 10>"      # even segment counts <= 12: two-sided kernel
        ctx.set_option("no_twin", 1)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_reg<4, 10>"
        ctx.set_option("no_twin", 0)
        wp, t = swarm(78, 300, 7)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_twin<4, 7>"       # odd: side 1 owns a knot more
        wp, t = swarm(78, 300, 20)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_twin<4, 20>"
        ctx.set_option("no_twin", 1)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_reg<4, 20>"       # the one-sided register kernel
        ctx.set_option("no_twin", 0)
        wp, t = swarm(79, 40, 10, shared_times=True)
        ctx.prepare_grid(t)
        ctx.solve_grid(wp)
        assert ctx.last_kernel() == "msnap::grid_gemm_kernel<8, 10>"
    with Context(order=9, max_segments=64) as ctx:
        ctx.set_option("no_twist", 1)
        for m, want in [(10, "msnap::solve_kernel_twin<5, 10>"), (6, "msnap::solve_kernel_twin<5, 6>"),
                        (7, "msnap::solve_kernel_twin<5, 7>"), (2, "msnap::solve_kernel_reg<5, 10>"),
                        (16, "msnap::solve_kernel_twin<5, 16>"),        # round 4: 13..20 segments, one wave per SIMD
                        (21, "msnap::solve_kernel<5, false>")]:
            wp, t = swarm(80 + m, 100, m)
            ctx.solve_batch(wp, t)
            assert ctx.last_kernel() == want
        ctx.set_option("no_twin", 1)
        wp, t = swarm(90, 100, 10)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_reg<5, 10>"
        wp, t = swarm(91, 100, 16)
        ctx.solve_batch(wp, t)
        assert ctx.last_kernel() == "msnap::solve_kernel_reg<5, 20>"


# ---------------------------------------------------------------------------
# the only OUTPUT files the reference itself ships for this path
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fname", ["Pol_matrix_1.csv", "Pol_matrix_2.csv", "Pol_matrix_1_simple.csv",
                                   "Pol_matrix_2_simple.csv"])
def test_hip_path_reproduces_the_reference_output_files(ctx7, fname):
    """resources/trajectories/Pol_matrix_*.csv (copied as data), written by scripts/drones_pols_generator.py:63-81:
    49 segments x 33 float32 columns on the 10/50 grid.  The waypoints are recovered from the file (c0 of each
    piece, the last one by evaluating the last piece at its duration), solved on the GPU along BOTH routes --
    the node's batched entry `paths_to_pols` (shared grid: streaming fp64 MFMA GEMM) and `solve_batch` (K1) --
    packed with msnap_pack_pol_matrix and compared with the file at its float32 storage limit."""
    import msnap_oracle as O
    from drone_path_planning_python_amd.nodes import drones_pols_generator as gen
    from drone_path_planning_python_amd.nodes.msgs import Path, PoseStamped
    mat = np.loadtxt(os.path.join(GOLDEN_DIR, fname), delimiter=",")
    assert mat.shape == (49, 33)
    M = mat.shape[0]
    wp = np.empty((M + 1, 4))
    for a in range(4):
        wp[:M, a] = mat[:, 1 + 8 * a]
        wp[M, a] = O.poly_eval(mat[M - 1, 1 + 8 * a:9 + 8 * a], mat[M - 1, 0])
    t = np.array([i * (10.0 / (M + 1)) for i in range(M + 1)])     # drones_pols_generator.py:44-46

    def check(packed, what):
        packed = np.asarray(packed, dtype=np.float64)
        assert packed.shape == mat.shape, what
        for a in range(4):
            blk = slice(1 + 8 * a, 9 + 8 * a)
            den = np.abs(mat[:, blk]).max()
            if den > 0:
                err = np.abs(packed[:, blk] - mat[:, blk]).max() / den
                assert err <= 2e-5, (what, a, err)
        np.testing.assert_allclose(packed[:, 0], mat[:, 0], rtol=1e-6, err_msg=what)

    # route 1: K1 (per-drone recurrence) + the pack kernel
    coef, dur, status = ctx7.solve_batch(wp[None], t[None])
    assert status[0] == 0 and "solve_kernel" in ctx7.last_kernel()
    check(ctx7.pack_pol_matrix(coef, dur)[0], "solve_batch + pack")
    # route 2: the node's batched path (poses -> yaw -> shared 10/n grid -> MFMA GEMM -> pack)
    path = Path()
    for i in range(M + 1):
        ps = PoseStamped()
        ps.pose.position.x, ps.pose.position.y, ps.pose.position.z = wp[i, 0], wp[i, 1], wp[i, 2]
        ps.pose.orientation.z, ps.pose.orientation.w = np.sin(0.5 * wp[i, 3]), np.cos(0.5 * wp[i, 3])
        path.poses.append(ps)
    matrix, _, _ = gen.paths_to_pols([path], ctx7)
    assert "grid_gemm_stream_kernel" in ctx7.last_kernel()
    check(matrix[0], "paths_to_pols")


def test_buffer_growth_inside_a_stream_capture_is_refused():
    """A context's scratch buffers grow by synchronising and freeing -- illegal while the stream is being captured
    into a graph.  A first-time-larger call inside a capture must come back with MSNAP_ECAPTURE instead of
    recording a launch on a buffer that is about to go; after one call outside the capture the same capture works
    and its replay gives the eager result."""
    import torch
    from drone_path_planning_python_amd import Context, MsnapError
    from drone_path_planning_python_amd.synthetic import swarm
    n, m = 48, 49                                   # 49 segments: the stash lives on the context's global slab
    wp, t = swarm(4242, n, m)
    dev = torch.device("cuda", 0)
    twp, tt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
    coef = torch.zeros((n, m, 4, 8), dtype=torch.float64, device=dev)
    dur = torch.zeros((n, m), dtype=torch.float64, device=dev)
    st = torch.zeros((n,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    with Context(order=7, max_segments=64) as ctx:
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with pytest.raises(MsnapError) as e:
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                    ctx.solve_batch_device(n, m, twp, tt, False, coef, dur, st)
            assert e.value.code == -8
            ctx.set_stream(side.cuda_stream)
            ctx.solve_batch_device(n, m, twp, tt, False, coef, dur, st)      # sizes the slab
            assert "solve_kernel<4, true>" in ctx.last_kernel()
            side.synchronize()
            eager = coef.clone()
            coef.zero_()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                ctx.solve_batch_device(n, m, twp, tt, False, coef, dur, st)
            ctx.set_stream(side.cuda_stream)
            g2.replay()
            side.synchronize()
            assert torch.equal(coef, eager) and int(st.abs().sum().item()) == 0
        ctx.use_own_stream()
