"""BASELINE.json configs[2] / configs[3] at full size (4096 drones), pinned end to end by
tests/golden/formation_golden.npz (tests/golden/make_formation_golden.py): formation-like
inputs through a8, solve, sample, pairwise pass, sweep against the reference's STL obstacles.

not gpu:  the oracle pipeline reproduces the fixture (so the fixture pins the oracle too);
gpu:      the HIP pipeline through the C-ABI gives the same hit sets, partners and distances.
The collision passes have no reference implementation (parity unpinned): the
fixture holds this repo's definition evaluated by the CPU oracle; the coefficients of every
128th drone in it were computed by the reference itself.
"""
import os

import numpy as np
import pytest

import c_oracle
import msnap_oracle as O
from conftest import GOLDEN_DIR, norm_rel
from drone_path_planning_python_amd import stl, synthetic

TIGHT = 1e-9


@pytest.fixture(scope="module")
def fgold():
    return np.load(os.path.join(GOLDEN_DIR, "formation_golden.npz"))


def _scene():
    return np.concatenate([stl.load_stl(os.path.join(GOLDEN_DIR, "env-scene-hole.stl")),
                           stl.load_stl(os.path.join(GOLDEN_DIR, "env-scene-ltu-experiment.stl"))])


def _inputs(cfg, fgold):
    import hashlib
    rb, off, t = synthetic.formation_config(cfg)
    h = hashlib.sha256()
    for a in (rb, off, t):
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    assert h.hexdigest() == str(fgold[f"cfg{cfg}_sha256"]), "synthetic.formation_config drifted from the fixture"
    return rb, off, t


@pytest.mark.parametrize("cfg", [2, 3])
def test_oracle_reproduces_formation_fixture(cfg, fgold):
    rb, off, t = _inputs(cfg, fgold)
    G, m, _ = rb.shape
    assert G * off.shape[0] == 4096 and m - 1 == synthetic.FORMATION_SEGMENTS[cfg]
    wp = synthetic.formation_waypoints(O.formation_transform(rb.reshape(G * m, 7), off), G)
    coef, dur, info, _ = c_oracle.solve_batch(wp, t, faithful=True, n_threads=0)
    assert not info.any()
    idx = fgold[f"cfg{cfg}_ref_idx"]
    assert norm_rel(coef[idx], fgold[f"cfg{cfg}_ref_coef"]) <= TIGHT     # the reference's own numbers
    S = synthetic.formation_sample_count(t)
    pos = c_oracle.sample_positions(coef, dur, synthetic.SAMPLE_DT, S)
    np.testing.assert_array_equal(pos[:3], O.sample_positions(coef[:3], dur[:3], synthetic.SAMPLE_DT, S))
    md, partner, hit = c_oracle.formation_collide(pos, synthetic.DRONE_RADIUS)
    np.testing.assert_array_equal(md, fgold[f"cfg{cfg}_pair_min_dist"])
    np.testing.assert_array_equal(np.nonzero(hit)[0], fgold[f"cfg{cfg}_pair_hit_idx"])
    np.testing.assert_array_equal(partner[hit], fgold[f"cfg{cfg}_pair_partner"])
    assert 0 < hit.sum() < 0.1 * hit.size          # sparse, not empty
    assert not (partner[hit] // 8 == np.nonzero(hit)[0] // 8).any()      # never a team mate
    if cfg == 3:
        mmd, mhit = c_oracle.mesh_sweep(pos, _scene(), synthetic.DRONE_RADIUS)
        np.testing.assert_array_equal(mmd, fgold["cfg3_mesh_min_dist"])
        np.testing.assert_array_equal(np.nonzero(mhit)[0], fgold["cfg3_mesh_hit_idx"])
        assert 0 < mhit.sum() < 0.1 * mhit.size


def test_c_collision_oracles_follow_the_numpy_definitions():
    wp, t = synthetic.swarm(3, 60, 6)
    wp[..., :3] *= 0.5
    coef, dur, _, _ = c_oracle.solve_batch(wp, t)
    pos = c_oracle.sample_positions(coef, dur, 0.1, 40)
    np.testing.assert_array_equal(pos, O.sample_positions(coef, dur, 0.1, 40))
    pos[7] = np.nan                  # a failed drone: +inf / -1 / no hit, invisible to the others (msnap.h)
    a, b = O.formation_collide(pos, 0.2), c_oracle.formation_collide(pos, 0.2)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert np.isinf(a[0][7]) and a[1][7] == -1 and not a[2][7] and not (a[1] == 7).any()
    tris = _scene()
    a, b = O.mesh_sweep(pos[:5], tris, 0.3), c_oracle.mesh_sweep(pos[:5], tris, 0.3)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [2, 3])
def test_full_size_formation_pipeline_matches_fixture(cfg, fgold, ctx7):
    rb, off, t = _inputs(cfg, fgold)
    G, m, _ = rb.shape
    K = off.shape[0]
    N = G * K
    # a8 on the GPU, against the oracle's restatement
    poses = ctx7.formation_transform(rb.reshape(G * m, 7), off)
    np.testing.assert_allclose(poses, O.formation_transform(rb.reshape(G * m, 7), off), rtol=0, atol=1e-12)
    wp = synthetic.formation_waypoints(poses, G)
    # the solve: per-drone-grid kernel and the shared-grid MFMA GEMM
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert (status == 0).all()
    ctx7.prepare_grid(t)
    gcoef, gdur, gstatus = ctx7.solve_grid(wp)
    assert (gstatus == 0).all()
    idx = fgold[f"cfg{cfg}_ref_idx"]
    assert norm_rel(coef[idx], fgold[f"cfg{cfg}_ref_coef"]) <= TIGHT
    assert norm_rel(gcoef[idx], fgold[f"cfg{cfg}_ref_coef"]) <= TIGHT
    assert norm_rel(gcoef, coef) <= 1e-10
    # sample -> pairwise pass
    S = synthetic.formation_sample_count(t)
    pos = ctx7.sample(coef, dur, synthetic.SAMPLE_DT, S, 3)
    md, partner, hit = ctx7.formation_collide(pos, pos, synthetic.DRONE_RADIUS)
    np.testing.assert_array_equal(np.nonzero(hit)[0], fgold[f"cfg{cfg}_pair_hit_idx"])
    np.testing.assert_array_equal(partner[hit], fgold[f"cfg{cfg}_pair_partner"])
    np.testing.assert_allclose(md, fgold[f"cfg{cfg}_pair_min_dist"], rtol=0, atol=1e-9)
    # sharded as on 8 GPUs: rows of one rank against all columns
    for r in (0, 5):
        lo, hi = r * N // 8, (r + 1) * N // 8
        smd, sp, sh = ctx7.formation_collide(pos[lo:hi], pos, synthetic.DRONE_RADIUS, row_offset=lo)
        np.testing.assert_array_equal(smd, md[lo:hi])
        np.testing.assert_array_equal(sp, partner[lo:hi])
        np.testing.assert_array_equal(sh, hit[lo:hi])
    # as on G GPUs: every rank of G = 2, 4, 8 emulated here -- rank p evaluates part p of the swarm's pairs
    # (each unordered pair on exactly one rank), the [G] part blocks are what the second all-gather moves,
    # and the fold must be the single-GPU result bit for bit
    for Gp in (2, 4, 8):
        parts = np.stack([ctx7.formation_collide_part(pos, p, Gp) for p in range(Gp)])
        for r in {0, Gp - 1, Gp // 2}:
            lo, hi = r * N // Gp, (r + 1) * N // Gp
            fmd, fp, fh = ctx7.formation_collide_finish(parts, N, synthetic.DRONE_RADIUS, row_offset=lo, n_rows=hi - lo)
            np.testing.assert_array_equal(fmd, md[lo:hi])
            np.testing.assert_array_equal(fp, partner[lo:hi])
            np.testing.assert_array_equal(fh, hit[lo:hi])
        # the parts are disjoint in pairs: a drone's minimum is attained in at least one of them and no part
        # reports a partner the drone never met
        d2s = np.stack([ctx7.split_part(parts[p], N)[0] for p in range(Gp)])
        assert np.array_equal(np.sqrt(d2s.min(axis=0)), md)
    if cfg == 3:
        mmd, mhit = ctx7.mesh_sweep(pos, _scene(), synthetic.DRONE_RADIUS)
        np.testing.assert_array_equal(np.nonzero(mhit)[0], fgold["cfg3_mesh_hit_idx"])
        np.testing.assert_allclose(mmd, fgold["cfg3_mesh_min_dist"], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_collision_passes_with_failed_and_missing_drones(ctx7):
    """msnap.h contract: n_cols == 0 -> (+inf, -1, 0); NaN samples never win a minimum; the Python
    pipeline refuses a shard whose solve reported failures."""
    from drone_path_planning_python_amd import swarm as sw
    wp, t = synthetic.swarm(3, 70, 6)
    wp[..., :3] *= 0.5
    wp[9, 2, 0] = np.nan
    coef, dur, status = ctx7.solve_batch(wp, t)
    assert status[9] == 3
    pos = ctx7.sample(coef, dur, 0.1, 40, 3)
    assert np.isnan(pos[9]).all()
    md, partner, hit = ctx7.formation_collide(pos, pos, 0.2)
    rmd, rpartner, rhit = O.formation_collide(pos, 0.2)
    np.testing.assert_array_equal(md, rmd)
    np.testing.assert_array_equal(partner, rpartner)
    np.testing.assert_array_equal(hit, rhit)
    assert np.isinf(md[9]) and partner[9] == -1 and not hit[9] and not (partner == 9).any()
    mmd, mhit = ctx7.mesh_sweep(pos[8:11], _scene(), 0.3)
    assert np.isinf(mmd[1]) and not mhit[1]
    md0, p0, h0 = ctx7.formation_collide(pos[:5], pos[:0], 0.2)
    assert np.isinf(md0).all() and (p0 == -1).all() and not h0.any()

    class Comp:   # formation_pass must refuse before touching the device
        def sample(self, *a):
            raise AssertionError("sampled a shard with failed drones")
    with pytest.raises(ValueError):
        sw.formation_pass(Comp(), coef, dur, 70, 1, 0, 0.1, 40, 0.2, status_local=status)


@pytest.mark.gpu
@pytest.mark.parametrize("n,S,parts", [(70, 40, 3), (129, 13, 2), (300, 31, 5), (515, 7, 8), (1000, 24, 7),
                                       (40, 4, 3), (1, 12, 2), (2, 12, 4), (257, 91, 16)])
def test_collide_in_parts_equals_the_oracle(ctx7, n, S, parts):
    """msnap_formation_collide_part / _finish on odd shapes: parts that start and end inside a row block, more
    parts than shares, a single drone, paths shorter than one sample chunk, a NaN drone, exact ties on a
    lattice.  Folded over the parts: bit for bit the oracle's pass over the whole swarm."""
    rng = np.random.default_rng(1000 * n + S)
    pos = rng.uniform(-2.0, 2.0, size=(n, S, 3))
    pos[: n // 2] = np.round(pos[: n // 2] * 2.0) / 2.0           # lattice positions: exact ties between partners
    if n > 20:
        pos[11] = np.nan                                           # a failed drone
    ref = c_oracle.formation_collide(pos, 0.3)
    blocks = np.stack([ctx7.formation_collide_part(pos, p, parts) for p in range(parts)])
    md, partner, hit = ctx7.formation_collide_finish(blocks, n, 0.3)
    np.testing.assert_array_equal(md, ref[0])
    np.testing.assert_array_equal(partner, ref[1])
    np.testing.assert_array_equal(hit, ref[2])
    lo, hi = n // 3, max(n // 3, 2 * n // 3)
    md2, p2, h2 = ctx7.formation_collide_finish(blocks, n, 0.3, row_offset=lo, n_rows=hi - lo)
    np.testing.assert_array_equal(md2, ref[0][lo:hi])
    np.testing.assert_array_equal(p2, ref[1][lo:hi])
    # forced sample parts (what a small part launch chooses by itself)
    ctx7.set_option("collide_sample_parts", 4)
    try:
        blocks4 = np.stack([ctx7.formation_collide_part(pos, p, parts) for p in range(parts)])
    finally:
        ctx7.set_option("collide_sample_parts", 0)
    md4, p4, _ = ctx7.formation_collide_finish(blocks4, n, 0.3)
    np.testing.assert_array_equal(md4, ref[0])
    np.testing.assert_array_equal(p4, ref[1])


def _broad_phase_swarm(kind, n, S, rng):
    """Swarms that stress the exact broad phase: what it may skip, and what it must not."""
    if kind == "dense":            # nothing can be culled
        return rng.uniform(-2.0, 2.0, size=(n, S, 3))
    if kind == "sparse":           # far-apart drones that wander a little: most shares are culled
        start = rng.uniform(-150.0, 150.0, size=(n, 1, 3)) * np.array([1.0, 1.0, 0.05])
        return start + np.cumsum(rng.normal(0.0, 0.05, size=(n, S, 3)), axis=1)
    if kind == "teams":            # rigid teams of 8 on a lattice: exact ties between partners, across groups too
        centre = np.round(rng.uniform(-60.0, 60.0, size=(n // 8 + 1, 1, 3)))
        drift = np.cumsum(np.round(rng.normal(0.0, 0.6, size=(n // 8 + 1, S, 3))) * 0.25, axis=1)
        off = np.array([[dx, dy, dz] for dx in (0.0, 0.5) for dy in (0.0, 0.5) for dz in (0.0, 0.5)])
        pos = (centre + drift)[:, None] + off[None, :, None, :]
        return pos.reshape(-1, S, 3)[rng.permutation((n // 8 + 1) * 8)[:n]]
    if kind == "far":              # beyond the +-1 km of the sort lattice, with a crossing pair and a far-away loner
        pos = 5000.0 + rng.uniform(-300.0, 300.0, size=(n, 1, 3)) + np.cumsum(rng.normal(0.0, 0.1, size=(n, S, 3)), axis=1)
        pos[5] = pos[n - 3] + 0.01
        pos[7] = -8000.0
        return pos
    raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,S", [("dense", 300, 19), ("sparse", 256, 91), ("sparse", 1000, 37), ("sparse", 2049, 13),
                                      ("teams", 777, 24), ("teams", 1536, 91), ("far", 600, 30), ("sparse", 513, 6),
                                      ("teams", 3072, 12), ("sparse", 9001, 7), ("teams", 16384, 6),
                                      ("dense", 2049, 38)])      # (2193 shares in 2 parts each: more items than wave slots)
def test_broad_phase_equals_the_full_pass(ctx7, kind, n, S):
    """The whole-swarm pass behind its exact broad phase (spatial sort, per-drone bounds, box test per 8-column share,
    surviving shares only): distances, partners and hits bit for bit those of the oracle's all-pairs pass -- with failed
    drones (NaN paths), drones with a few NaN samples, coincident drones and lattice ties whose partner is decided by the
    ORIGINAL index -- and of the pass without the broad phase."""
    rng = np.random.default_rng(77 * n + S)
    pos = _broad_phase_swarm(kind, n, S, rng)
    pos[11] = np.nan                                   # a failed drone
    pos[n // 2, S // 3:S // 3 + 2] = np.nan            # two missing samples
    pos[n - 1] = pos[3]                                # coincident paths: distance 0, each the other's partner
    ref = c_oracle.formation_collide(pos, 0.3)
    ctx7.set_option("collide_cull_min_drones", 256)
    try:
        got = ctx7.formation_collide(pos, pos, 0.3)
        assert ctx7.get_option("collide_last_cull") == 1
        shares, survivors = ctx7.get_option("collide_last_shares"), ctx7.get_option("collide_last_survivors")
        groups = ctx7.get_option("collide_last_group_pairs")
        ng = (n + 7) // 8
        assert ng <= groups <= ng * (ng + 1) // 2
        # both evaluators behind the broad phase: the surviving 128 x 8 shares (also with forced sample parts) and the
        # surviving 8 x 8 group pairs
        ctx7.set_option("collide_cull_mode", 2)
        by_groups = ctx7.formation_collide(pos, pos, 0.3)
        for a, b in zip(by_groups, ref):
            np.testing.assert_array_equal(a, b)
        ctx7.set_option("collide_cull_mode", 1)
        for sp in (0, 1, 3, 8):
            ctx7.set_option("collide_sample_parts", sp)
            forced = ctx7.formation_collide(pos, pos, 0.3)
            for a, b in zip(forced, ref):
                np.testing.assert_array_equal(a, b)
        ctx7.set_option("collide_sample_parts", 0)
        ctx7.set_option("collide_cull_mode", 0)
        ctx7.set_option("collide_no_cull", 1)
        plain = ctx7.formation_collide(pos, pos, 0.3)
        assert ctx7.get_option("collide_last_cull") == 0
    finally:
        ctx7.set_option("collide_no_cull", 0)
        ctx7.set_option("collide_sample_parts", 0)
        ctx7.set_option("collide_cull_mode", 0)
        ctx7.set_option("collide_cull_min_drones", 0)
    for a, b, c in zip(got, ref, plain):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(c, b)
    assert got[0][3] == 0.0 and got[0][n - 1] == 0.0
    assert 0 < survivors <= shares
    if kind == "sparse" and n >= 1000:
        assert survivors < shares * 3 // 4, (survivors, shares)     # the broad phase does cull
    if kind == "dense":
        assert survivors > shares * 9 // 10


@pytest.mark.gpu
def test_large_swarm_group_list_overflow_falls_back_to_the_shares(ctx7):
    """Swarms above 8192 drones get 2^18 group-pair list slots and both evaluators are launched: the share evaluator and
    its merge return at once unless the survivors overflowed the list.  A sparse swarm goes through the group pairs, a
    dense one (652 653 group pairs, all surviving) overflows -- both equal to the oracle."""
    rng = np.random.default_rng(31)
    n, S = 9136, 6
    ctx7.set_option("collide_cull_mode", 2)
    try:
        for kind, by_groups in (("sparse", 1), ("dense", 0)):
            pos = _broad_phase_swarm(kind, n, S, rng)
            ref = c_oracle.formation_collide(pos, 0.3)
            got = ctx7.formation_collide(pos, pos, 0.3)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
            assert ctx7.get_option("collide_last_cull") == 1
            assert ctx7.get_option("collide_last_by_groups") == by_groups
            assert (ctx7.get_option("collide_last_group_pairs") > (1 << 18)) == (not by_groups)
    finally:
        ctx7.set_option("collide_cull_mode", 0)


@pytest.mark.gpu
def test_evaluator_choice_follows_the_previous_pass():
    """"collide_cull_mode" 0: which evaluator runs behind the broad phase is a host decision (the launch sequences
    differ: the group pairs finish inside their evaluator, the shares are followed by a merge), taken from the survivor
    counts the context's previous pass over a swarm of this size left in page-locked memory.  A first pass takes the
    shares; a sparse swarm then moves to the group pairs, a dense one back to the shares -- results equal throughout."""
    from drone_path_planning_python_amd import Context
    rng = np.random.default_rng(21)
    n, S = 3072, 13
    sparse, dense = _broad_phase_swarm("sparse", n, S, rng), _broad_phase_swarm("dense", n, S, rng)
    other = _broad_phase_swarm("sparse", n + 64, S, rng)
    ref_s, ref_d = c_oracle.formation_collide(sparse, 0.3), c_oracle.formation_collide(dense, 0.3)
    ref_o = c_oracle.formation_collide(other, 0.3)

    def run(ctx, pos, ref):
        got = ctx.formation_collide(pos, pos, 0.3)           # (host entry: synchronises, the hint is in place)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        assert ctx.get_option("collide_last_cull") == 1
        return ctx.get_option("collide_last_by_groups")

    with Context(order=7, max_segments=16) as ctx:
        assert run(ctx, sparse, ref_s) == 0                  # no counts yet: the shares
        assert run(ctx, sparse, ref_s) == 1                  # few group pairs survive: the group evaluator
        assert run(ctx, sparse, ref_s) == 1
        assert run(ctx, other, ref_o) == 0                   # another swarm size: the counts do not apply
        assert run(ctx, dense, ref_d) == 0                   # ... and now they are that swarm's
        assert run(ctx, sparse, ref_s) == 0                  # stale for this size too by now
        assert run(ctx, sparse, ref_s) == 1
        assert run(ctx, dense, ref_d) == 1                   # a swarm that turned dense pays once ...
        assert run(ctx, dense, ref_d) == 0                   # ... and goes back to the shares
        ctx.set_option("collide_cull_mode", 2)
        assert run(ctx, dense, ref_d) == 1
        ctx.set_option("collide_cull_mode", 1)
        assert run(ctx, sparse, ref_s) == 0


@pytest.mark.gpu
def test_broad_phase_is_the_default_for_a_whole_large_swarm_only(ctx7):
    rng = np.random.default_rng(5)
    pos = _broad_phase_swarm("sparse", 3072, 7, rng)
    ctx7.formation_collide(pos, pos, 0.3)
    assert ctx7.get_option("collide_last_cull") == 1
    ctx7.formation_collide(pos[:2048], pos[:2048], 0.3)
    assert ctx7.get_option("collide_last_cull") == 0          # small swarm
    ctx7.formation_collide(pos[:3071], pos, 0.3)
    assert ctx7.get_option("collide_last_cull") == 0          # a shard, not the whole swarm


@pytest.mark.gpu
def test_row_image_query_follows_the_broad_phase(ctx7):
    """msnap_formation_collide_reads_rows_t: a caller that has the sampler write the pairwise pass's row image asks
    first -- a whole swarm behind the broad phase builds its own."""
    assert ctx7.collide_reads_rows_t(1024, 0, 1024, 91)            # small swarm: the plain pass reads the row image
    assert ctx7.collide_reads_rows_t(4096, 0, 4096, 91)            # whole large swarm: the sampler's boxes and sort keys
    assert ctx7.collide_reads_rows_t(2048, 0, 4096, 91)            # a shard
    assert ctx7.collide_reads_rows_t(2048, 2048, 4096, 91)
    assert not ctx7.collide_reads_rows_t(4096, 0, 4096, 3)         # shorter than a sample chunk: plain loops
    from drone_path_planning_python_amd import swarm as sw
    import torch
    comp = sw.DeviceCompute(ctx7, torch)
    try:
        wp, t = synthetic.swarm(41, 3072, 4)
        coef, dur, status = comp.solve(torch.from_numpy(wp).cuda(), torch.from_numpy(t).cuda())
        pos, keys = comp.sample_rows_t(coef, dur, 0.1, 12, n_cols=3072)
        pos2, rows_t2 = comp.sample_rows_t(coef[:1000], dur[:1000], 0.1, 12, n_cols=1000)
        assert keys is not None and rows_t2 is not None and torch.equal(pos[:1000], pos2)
        pos3, none = comp.sample_rows_t(coef, dur, 0.1, 4, n_cols=3072)
        assert none is None
        # the hand-over of the whole swarm (boxes and sort keys) and its pass; then the same buffer offered to passes it
        # was not written for -- the broad phase switched off, another swarm size: ignored, never misread
        ref = c_oracle.formation_collide(pos.cpu().numpy(), 0.3)
        for no_cull, rows in ((0, 3072), (1, 3072), (0, 3000)):
            ctx7.set_option("collide_no_cull", no_cull)
            try:
                got = comp.collide(pos[:rows], 0, pos[:rows], 0.3, rows_t=keys)
                assert ctx7.get_option("collide_last_cull") == (0 if no_cull or rows < 3072 else 1)
            finally:
                ctx7.set_option("collide_no_cull", 0)
            want = ref if rows == 3072 else c_oracle.formation_collide(pos[:rows].cpu().numpy(), 0.3)
            np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
            np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])
    finally:
        comp.close()


@pytest.mark.gpu
def test_broad_phase_pass_replays_from_a_graph():
    """The whole-swarm pass behind its broad phase is plain stream work -- counters zeroed by its first kernel, lists and
    per-drone atomics rebuilt per pass, no host read-back: captured into a graph (after one eager call has sized the
    context's buffers) it replays with new positions in the same buffers and gives the eager results."""
    import torch
    from drone_path_planning_python_amd import Context
    rng = np.random.default_rng(99)
    n, S = 3072, 19
    dev = torch.device("cuda", 0)
    pos_a = torch.from_numpy(_broad_phase_swarm("teams", n, S, rng)).to(dev)
    pos_b = torch.from_numpy(_broad_phase_swarm("sparse", n, S, rng)).to(dev)
    buf = pos_a.clone()
    md = torch.empty((n,), dtype=torch.float64, device=dev)
    partner = torch.empty((n,), dtype=torch.int32, device=dev)
    hit = torch.empty((n,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    with Context(order=7, max_segments=16) as ctx:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            eager = {}
            for name, p in (("a", pos_a), ("b", pos_b)):
                buf.copy_(p)
                ctx.formation_collide_device(n, 0, n, S, buf, buf, 0.3, md, partner, hit)
                assert ctx.get_option("collide_last_cull") == 1
                side.synchronize()
                eager[name] = (md.clone(), partner.clone(), hit.clone())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                ctx.formation_collide_device(n, 0, n, S, buf, buf, 0.3, md, partner, hit)
            ctx.set_stream(side.cuda_stream)
            for name, p in (("a", pos_a), ("b", pos_b), ("a", pos_a)):
                buf.copy_(p)
                md.zero_()
                partner.zero_()
                g.replay()
                side.synchronize()
                for got, want in zip((md, partner, hit), eager[name]):
                    assert torch.equal(got, want)
        ctx.use_own_stream()
    ref = c_oracle.formation_collide(pos_b.cpu().numpy(), 0.3)
    np.testing.assert_array_equal(eager["b"][0].cpu().numpy(), ref[0])
    np.testing.assert_array_equal(eager["b"][1].cpu().numpy(), ref[1])


@pytest.mark.gpu
def test_a_captured_pass_survives_a_later_larger_pass():
    """A graph holds raw pointers into the context's scratch block.  A later EAGER pass that needs the block larger
    must not free it under the graph (msnap.h "Stream capture"): the old block is retired, the graph keeps replaying
    the pass it captured with the right results, and msnap_release_graph_buffers frees it once the graph is gone."""
    import torch
    from drone_path_planning_python_amd import Context
    rng = np.random.default_rng(5)
    n, S, nbig = 3072, 13, 6000
    dev = torch.device("cuda", 0)
    pos = torch.from_numpy(_broad_phase_swarm("teams", n, S, rng)).to(dev)
    big = torch.from_numpy(_broad_phase_swarm("sparse", nbig, S, rng)).to(dev)
    md, partner, hit = (torch.empty((nbig,), dtype=dt, device=dev) for dt in (torch.float64, torch.int32, torch.int32))
    ref = c_oracle.formation_collide(pos.cpu().numpy(), 0.3)
    ref_big = c_oracle.formation_collide(big.cpu().numpy(), 0.3)
    side = torch.cuda.Stream()
    with Context(order=7, max_segments=16) as ctx:
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            ctx.formation_collide_device(n, 0, n, S, pos, pos, 0.3, md, partner, hit)      # sizes the buffers
            side.synchronize()
            assert ctx.release_graph_buffers() == 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                # a query that has to synchronise is refused inside the capture instead of invalidating it
                with pytest.raises(Exception) as ei:
                    ctx.get_option("collide_last_survivors")
                assert getattr(ei.value, "code", None) == -8
                ctx.formation_collide_device(n, 0, n, S, pos, pos, 0.3, md, partner, hit)
            ctx.set_stream(side.cuda_stream)
            # the eager pass needs a much larger block: without retention this frees the graph's memory
            ctx.formation_collide_device(nbig, 0, nbig, S, big, big, 0.3, md, partner, hit)
            side.synchronize()
            np.testing.assert_array_equal(md.cpu().numpy(), ref_big[0])
            np.testing.assert_array_equal(partner.cpu().numpy(), ref_big[1])
            for _ in range(3):
                md.fill_(-1.0)
                partner.fill_(-7)
                g.replay()
                side.synchronize()
                np.testing.assert_array_equal(md[:n].cpu().numpy(), ref[0])
                np.testing.assert_array_equal(partner[:n].cpu().numpy(), ref[1])
                np.testing.assert_array_equal(hit[:n].cpu().numpy().astype(bool), ref[2])
                # and the larger pass still works beside it
                ctx.formation_collide_device(nbig, 0, nbig, S, big, big, 0.3, md, partner, hit)
                side.synchronize()
                np.testing.assert_array_equal(md.cpu().numpy(), ref_big[0])
            del g
            side.synchronize()
            assert ctx.release_graph_buffers() > 0
            assert ctx.release_graph_buffers() == 0
        ctx.use_own_stream()


@pytest.mark.gpu
def test_read_only_options_and_last_pass_reports(ctx7):
    """"collide_last_*" report what the last pass did and cannot be set; "collide_last_by_groups" /
    "collide_last_pairs_evaluated" and msnap_formation_whole_pass_pays are the library's own statements of its
    device-side choice (what bench.py and swarm.py read instead of re-typing the cost model)."""
    from drone_path_planning_python_amd._lib import MsnapError
    rng = np.random.default_rng(8)
    n, S = 3072, 12
    for name in ("collide_last_cull", "collide_last_shares", "collide_last_sym", "collide_last_survivors",
                 "collide_last_group_pairs", "collide_last_by_groups", "collide_last_pairs_evaluated"):
        with pytest.raises(MsnapError):
            ctx7.set_option(name, 1)
    assert ctx7.collide_takes_broad_phase(n, 0, n, S) and not ctx7.collide_takes_broad_phase(n, 0, n + 1, S)
    assert not ctx7.collide_takes_broad_phase(1024, 0, 1024, S) and not ctx7.collide_takes_broad_phase(n, 0, n, 3)
    sparse = _broad_phase_swarm("sparse", n, S, rng)
    ctx7.formation_collide(sparse, sparse, 0.3)
    surv, groups = ctx7.get_option("collide_last_survivors"), ctx7.get_option("collide_last_group_pairs")
    by_groups, pairs = ctx7.get_option("collide_last_by_groups"), ctx7.get_option("collide_last_pairs_evaluated")
    assert ctx7.get_option("collide_last_cull") == 1 and 0 < surv <= ctx7.get_option("collide_last_shares")
    assert pairs == (groups * 64 if by_groups else surv * 1024) and pairs < n * (n - 1) // 2 // 2 and groups * 64 < pairs
    assert ctx7.whole_pass_pays(n, 4) and ctx7.whole_pass_pays(n, 8)
    dense = _broad_phase_swarm("dense", n, S, rng)
    ctx7.formation_collide(dense, dense, 0.3)          # (by whichever evaluator the sparse pass's counts chose)
    assert ctx7.get_option("collide_last_pairs_evaluated") >= n * (n - 1) // 2
    ctx7.formation_collide(dense, dense, 0.3)          # nothing was culled: back to the shares
    assert ctx7.get_option("collide_last_by_groups") == 0
    assert ctx7.get_option("collide_last_pairs_evaluated") >= n * (n - 1) // 2
    assert not ctx7.whole_pass_pays(n, 2)
    ctx7.formation_collide(dense[:1000], dense[:1000], 0.3)       # not a broad-phase pass: nothing to report
    assert ctx7.get_option("collide_last_cull") == 0 and ctx7.get_option("collide_last_pairs_evaluated") == -1
    assert ctx7.get_option("collide_last_group_pairs") == 0 and not ctx7.whole_pass_pays(1000, 4)


@pytest.mark.gpu
def test_multi_rank_pairwise_mode_follows_the_survivor_counts(ctx7):
    """swarm.DeviceCompute.pairwise_mode: several ranks run the whole gathered swarm behind the broad phase where it
    culls, and fall back to the parts (every pair on one rank) where it does not -- decided once per swarm shape from
    the survivor counts of the first whole pass, hence identically on every rank."""
    import torch
    from drone_path_planning_python_amd import swarm as sw
    rng = np.random.default_rng(3)
    comp = sw.DeviceCompute(ctx7, torch)
    try:
        assert comp.pairwise_mode(4096, 91, 1) == "parts"                 # one rank: not a question
        assert comp.pairwise_mode(1024, 91, 4) == "parts"                 # too small for the broad phase
        n, S = 3072, 12
        assert comp.pairwise_mode(n, S, 4) == "whole"
        sparse = torch.from_numpy(_broad_phase_swarm("sparse", n, S, rng)).cuda()
        comp.collide(sparse, 0, sparse, 0.3)
        comp.note_whole_pass(n, S, 4)
        assert comp.pairwise_mode(n, S, 4) == "whole"
        dense = torch.from_numpy(_broad_phase_swarm("dense", n, S + 1, rng)).cuda()
        assert comp.pairwise_mode(n, S + 1, 4) == "whole"
        comp.collide(dense, 0, dense, 0.3)
        comp.note_whole_pass(n, S + 1, 4)
        assert comp.pairwise_mode(n, S + 1, 4) == "parts"                 # nothing culled: divide the pairs instead
        assert comp.pairwise_mode(n, S, 4) == "whole"                     # (per swarm shape)
    finally:
        comp.close()


@pytest.mark.gpu
def test_broad_phase_on_random_shapes():
    """tools/cull_stress.py, 60 seeded cases: random swarm sizes (also one short of and one past the tile sizes), sample
    counts, kinds, NaN drones and samples, coincident drones, every evaluator and forced sample parts -- each equal to
    the oracle.  (The tool found the shape of ("dense", 2049, 38) above: shares cut into more items than the row-side
    entries were laid out for.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cull_stress", os.path.join(os.path.dirname(GOLDEN_DIR), "..", "tools", "cull_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(budget=120.0, seed=5, max_cases=60, big=False) == 60


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,S,mesh", [(4096, 10, 91, False), (300, 8, 40, True), (3500, 20, 50, True)])
def test_formation_pass_from_waypoints_equals_the_staged_pass(n, m, S, mesh):
    """swarm.formation_pass_from_waypoints (solve and sampler as one launch where the library fuses them, two inside the
    same call at 20 segments) against solve_grid -> formation_pass: the same coefficients, samples, minima, partners
    and hits, bit for bit -- behind the broad phase (4096, 3500 drones) and without it, with and without a mesh; a
    failed drone is refused after the solve unless the caller takes the status check on itself."""
    import torch
    from drone_path_planning_python_amd import Context, swarm as sw
    from drone_path_planning_python_amd.synthetic import swarm
    ctx = Context(order=7, max_segments=32)
    try:
        wp, _ = swarm(4000 + n, n, m)
        wp[..., :3] *= 0.4
        ctx.prepare_grid(np.linspace(0.0, 10.0, m + 1))
        comp = sw.DeviceCompute(ctx, torch)
        d_wp = torch.from_numpy(wp).cuda()
        tris = torch.from_numpy(_scene()).cuda() if mesh else None
        coef, dur, status = comp.solve_grid(d_wp)
        a = sw.formation_pass(comp, coef, dur, n, 1, 0, 0.1, S, 0.2, torch=torch, status_local=status, mesh_tris=tris)
        b, coef2, dur2, status2 = sw.formation_pass_from_waypoints(comp, d_wp, n, 1, 0, 0.1, S, 0.2, torch=torch, mesh_tris=tris)
        for x, y in ((coef, coef2), (dur, dur2), (status, status2), (a.positions_all, b.positions_all), (a.min_dist, b.min_dist),
                     (a.partner, b.partner), (a.hit, b.hit)):
            np.testing.assert_array_equal(x.cpu().numpy().view(np.uint8), y.cpu().numpy().view(np.uint8))
        if mesh:
            np.testing.assert_array_equal(a.mesh_min_dist.cpu().numpy(), b.mesh_min_dist.cpu().numpy())
            np.testing.assert_array_equal(a.mesh_hit.cpu().numpy(), b.mesh_hit.cpu().numpy())
        bad = d_wp.clone()
        bad[5, 1, 0] = float("nan")
        with pytest.raises(ValueError):
            sw.formation_pass_from_waypoints(comp, bad, n, 1, 0, 0.1, S, 0.2, torch=torch)
        res, _, _, st = sw.formation_pass_from_waypoints(comp, bad, n, 1, 0, 0.1, S, 0.2, torch=torch, check_status=False)
        assert int(st[5]) != 0 and np.isinf(res.min_dist[5].item()) and int(res.partner[5]) == -1
        torch.cuda.synchronize()
    finally:
        ctx.close()
