"""The oracle is pinned here: it must reproduce what the REFERENCE computed
(tests/golden/ref_golden.npz, produced by tests/golden/make_golden.py importing
the reference unmodified), the known-answer vector of SURVEY.md Appendix C and
the reference's own float32 output files (resources/trajectories/Pol_matrix_*.csv,
copied as data)."""
import os

import numpy as np
import pytest

import c_oracle
import msnap_oracle as O
from conftest import GOLDEN_DIR, norm_rel

SINGLE = ["cfg1", "testdata", "m1", "m2", "t0quirk", "path49"]
BATCH = ["cfg2", "cfg2s", "m20", "m3"]


@pytest.mark.parametrize("name", SINGLE)
def test_numpy_oracle_bit_identical_single(golden, name):
    coef, dur = O.calculate_trajectory4D(golden[name + "_wp"], golden[name + "_t"])
    np.testing.assert_array_equal(coef, golden[name + "_coef"])
    np.testing.assert_array_equal(dur, golden[name + "_dur"])


@pytest.mark.parametrize("name", BATCH)
def test_numpy_oracle_bit_identical_batch(golden, name):
    n = 8   # the faithful restatement is slow; 8 drones per case keep the CPU suite short
    t = golden[name + "_t"]
    coef, dur = O.solve_batch(golden[name + "_wp"][:n], t if t.ndim == 1 else t[:n])
    np.testing.assert_array_equal(coef, golden[name + "_coef"][:n])


@pytest.mark.parametrize("name", BATCH)
@pytest.mark.parametrize("faithful", [True, False])
def test_c_oracle_matches_reference(golden, name, faithful):
    coef, dur, info, _ = c_oracle.solve_batch(golden[name + "_wp"], golden[name + "_t"], faithful=faithful)
    assert (info == 0).all()
    assert norm_rel(coef, golden[name + "_coef"]) < 1e-10
    rdur = golden[name + "_dur"]
    np.testing.assert_array_equal(dur, rdur)


def test_fast_numpy_oracle(golden):
    coef, dur = O.solve_batch_fast(golden["cfg2_wp"][:16], golden["cfg2_t"][:16])
    assert norm_rel(coef, golden["cfg2_coef"][:16]) < 1e-10


def test_appendix_c_known_answer(golden):
    """config 1, full-precision reference output."""
    coef = golden["cfg1_coef"]
    x1 = [2.0, 5.640237077579474, 3.6053781455735336, -2.1836605189543383, -1.9836819799625636,
          0.6011424658810961, 0.2332205334418886, -0.06762862464760394]
    yaw2 = [0.0, 1.101463296964554, -0.06436634858876582, -0.9704678440823047, -0.003731077074223242,
            0.4557143271358343, 0.19468791443672137, -0.21330026879181638]
    np.testing.assert_allclose(coef[1, 0], x1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(coef[2, 3], yaw2, rtol=0, atol=1e-12)
    got, _ = O.calculate_trajectory4D(golden["cfg1_wp"], golden["cfg1_t"])
    np.testing.assert_allclose(got[1, 0], x1, rtol=0, atol=1e-12)


def test_testdata_known_answer(golden):
    """the reference's own __main__ demo (18 waypoints, dt = 2)."""
    c = golden["testdata_coef"][0, 0]
    np.testing.assert_allclose(c[4:], [3.0270487768666903e-02, -2.0840517637741526e-02,
                                       5.1579487927348776e-03, -4.5360490807955125e-04], rtol=1e-9)
    assert abs(golden["testdata_eval17"][0] - (-0.28721862765096506)) < 1e-12
    coef, dur = O.calculate_trajectory4D(golden["testdata_wp"], golden["testdata_t"])
    assert abs(O.piecewise_eval(coef[:, 0, :], dur, 17.0) - golden["testdata_eval17"][0]) < 1e-13


@pytest.mark.parametrize("fname", ["Pol_matrix_1.csv", "Pol_matrix_2.csv", "Pol_matrix_1_simple.csv",
                                   "Pol_matrix_2_simple.csv"])
def test_reference_csv_outputs(fname):
    """The reference's own output files: 49 segments x 33 float32 columns.  The
    waypoints are recoverable (c0 of each piece; the last one by evaluating the
    last piece at its duration); re-solving must reproduce the file to float32
    storage precision (0.9e-6 .. 2.3e-6 norm-relative)."""
    mat = np.loadtxt(os.path.join(GOLDEN_DIR, fname), delimiter=",")
    assert mat.shape == (49, 33)
    M = mat.shape[0]
    dur = mat[:, 0]
    wp = np.empty((M + 1, 4))
    for a in range(4):
        wp[:M, a] = mat[:, 1 + 8 * a]
        wp[M, a] = O.poly_eval(mat[M - 1, 1 + 8 * a:9 + 8 * a], dur[M - 1])
    t = np.array([0.2 * i for i in range(M + 1)])   # the 10/50 grid of drones_pols_generator.py:44-46
    coef, d = O.calculate_trajectory4D(wp, t)
    packed = O.pack_pol_matrix(coef, d).astype(np.float64)
    for a in range(4):
        blk = slice(1 + 8 * a, 9 + 8 * a)
        den = np.abs(mat[:, blk]).max()
        if den == 0:
            continue
        assert np.abs(packed[:, blk] - mat[:, blk]).max() / den < 2e-5
    np.testing.assert_allclose(packed[:, 0], mat[:, 0], rtol=1e-6)


def test_piecewise_eval_matches_reference(golden):
    coef, dur = O.calculate_trajectory4D(golden["cfg1_wp"], golden["cfg1_t"])
    for t, ref in zip(golden["pweval_t"], golden["pweval_val"]):
        got = [O.piecewise_eval(coef[:, a, :], dur, float(t)) for a in range(4)]
        np.testing.assert_array_equal(got, ref)


def test_flatness_eval_matches_reference(golden):
    """Trajectory.eval on the reference's src/traj.csv (crazyswarm format, header row)."""
    mat = np.loadtxt(os.path.join(GOLDEN_DIR, "traj.csv"), delimiter=",", skiprows=1, usecols=range(33))
    assert abs(float(np.sum(mat[:, 0])) - golden["flat_duration"][0]) < 1e-12
    for t, ref in zip(golden["flat_t"], golden["flat_out"]):
        pos, vel, acc, omega, yaw = O.trajectory_eval(mat, float(t))
        got = np.concatenate([pos, vel, acc, omega, [yaw]])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)


def test_order9_reduces_and_is_consistent():
    """Order 9 has no reference (parity unpinned): the generalisation must reduce to
    the order-7 system for ncoef = 8 and satisfy its own defining conditions."""
    rng = np.random.default_rng(3)
    M = 6
    wp = rng.uniform(-3, 3, size=(M + 1, 4))
    t = np.concatenate([[0.0], np.cumsum(rng.uniform(0.6, 1.6, size=M))])
    c9, dur = O.calculate_trajectory4D(wp, t, ncoef=10)
    for a in range(4):
        for i in range(M):
            assert abs(O.poly_eval(c9[i, a], 0.0) - wp[i, a]) < 1e-9
            assert abs(O.poly_eval(c9[i, a], dur[i]) - wp[i + 1, a]) < 1e-8
        # C^8 continuity at the interior knots
        for i in range(M - 1):
            left, right = c9[i, a].copy(), c9[i + 1, a].copy()
            for _ in range(8):
                left, right = O.poly_derivative(left), O.poly_derivative(right)
                lv, rv = O.poly_eval(left, dur[i]), O.poly_eval(right, 0.0)
                assert abs(lv - rv) <= 1e-6 * max(1.0, abs(lv))


def test_order9_fixture_pins_the_oracles():
    """tests/golden/order9_golden.npz (make_order9_golden.py): 60-digit mpmath solutions of the order-9
    collocation system, cross-checked there against the order-9 KKT / QP formulation (SURVEY.md 8c (iii):
    order 9 has no reference implementation).  Both oracles must sit within 1e-9 of it."""
    import c_oracle
    g = np.load(os.path.join(GOLDEN_DIR, "order9_golden.npz"))
    assert float(g["kkt_vs_collocation_max"]) < 1e-40 and int(g["digits"]) >= 50
    names = sorted({k[:-3] for k in g.files if k.endswith("_wp")})
    assert {"m10", "m10s", "m10q", "m16", "m20"} <= set(names)
    worst = 0.0
    for name in names:
        wp, t, ref = g[name + "_wp"], g[name + "_t"], g[name + "_coef"]
        assert ref.shape == (wp.shape[0], wp.shape[1] - 1, 4, 10)
        for coef in (O.solve_batch_fast(wp, t, ncoef=10)[0],
                     c_oracle.solve_batch(wp, t, ncoef=10, faithful=True, n_threads=0)[0]):
            num = np.abs(coef - ref).max(axis=(1, 3))
            worst = max(worst, float((num / np.abs(ref).max(axis=(1, 3))).max()))
    assert worst <= 1e-9, worst
    # ... and the generalisation is the reference's system at 8 coefficients (k = 4): same assembly code
    A8, _ = O.assemble_1d(g["m4_t"][0], g["m4_wp"][0, :, 0], 8)
    assert A8.shape == (32, 32)
