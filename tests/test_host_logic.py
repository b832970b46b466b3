"""Host-side mirror of the reference interface: value types, evaluators, node
helpers, sharding arithmetic, STL I/O.  No GPU."""
import math
import os

import numpy as np
import pytest

import msnap_oracle as O
from conftest import GOLDEN_DIR
from drone_path_planning_python_amd import stl, swarm
from drone_path_planning_python_amd.nodes import drones_pols_generator as dpg
from drone_path_planning_python_amd.nodes import msgs
from drone_path_planning_python_amd.optimizations import (PiecewisePolynomial, Point_time, Polynomial, Trajectory,
                                                        Waypoint)
from drone_path_planning_python_amd.optimizations.calculatingTrajectories import (raise_for_status,
                                                                                waypoints_to_arrays)


def test_polynomial_eval_derivative_rowbuilder():
    p = Polynomial([1.0, -2.0, 0.5, 3.0])
    assert p.eval(2.0) == ((3.0 * 2 + 0.5) * 2 - 2.0) * 2 + 1.0
    assert p.derivative().p == [-2.0, 1.0, 9.0]
    ones = Polynomial([1] * 8)
    for j in range(7):
        row = np.pad(ones.pol_coeffs_at_t(0.7), (8 - len(ones.p), 0))
        np.testing.assert_allclose(row, O.deriv_row(j, 0.7), rtol=1e-15)
        ones = ones.derivative()
    with pytest.raises(AssertionError):
        p.eval(-1.0)
    # (8,1) coefficients evaluate to shape (1,) like the reference
    q = Polynomial(np.arange(8.0).reshape(8, 1))
    assert np.shape(q.eval(0.3)) == (1,)


def test_piecewise_polynomial_matches_reference_samples(golden):
    coef, dur = golden["cfg1_coef"], golden["cfg1_dur"]
    for a in range(4):
        pc = PiecewisePolynomial([Polynomial(coef[j, a].reshape(8, 1)) for j in range(3)], [float(x) for x in dur])
        for t, ref in zip(golden["pweval_t"], golden["pweval_val"]):
            assert float(np.ravel(pc.eval(float(t)))[0]) == ref[a]


def test_trajectory_loadcsv_and_flatness(golden):
    tr = Trajectory()
    tr.loadcsv(os.path.join(GOLDEN_DIR, "traj.csv"))
    assert tr.n_pieces() == 10
    assert abs(tr.duration - golden["flat_duration"][0]) < 1e-12
    for t, ref in zip(golden["flat_t"], golden["flat_out"]):
        o = tr.eval(float(t))
        got = np.concatenate([o.pos, o.vel, o.acc, o.omega, [o.yaw]])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)
    # the reference's skiprows=1 drops the first piece of a header-less Pol_matrix file
    tr2 = Trajectory()
    tr2.loadcsv(os.path.join(GOLDEN_DIR, "Pol_matrix_1.csv"))
    assert tr2.n_pieces() == 48
    tr3 = Trajectory()
    tr3.loadcsv(os.path.join(GOLDEN_DIR, "Pol_matrix_1.csv"), skiprows=0)
    assert tr3.n_pieces() == 49


def test_waypoint_types_and_array_conversion():
    w = Waypoint(1.0, 2.0, 3.0, 0.4)
    assert [w.getType(k) for k in range(4)] == [1.0, 2.0, 3.0, 0.4]
    pts = [Point_time(Waypoint(i, 2 * i, 3 * i, 0.1 * i), t=0.5 * i) for i in range(4)]
    wp, t = waypoints_to_arrays(pts)
    assert wp.shape == (4, 4) and t.tolist() == [0.0, 0.5, 1.0, 1.5]
    assert wp[2].tolist() == [2.0, 4.0, 6.0, 0.2]


def test_status_to_reference_exceptions():
    t = np.array([0.0, 1.0, 2.0])
    raise_for_status(0, t)
    with pytest.raises(np.linalg.LinAlgError):
        raise_for_status(1, t)
    with pytest.raises(np.linalg.LinAlgError):
        raise_for_status(2, np.array([0.0, 1.0, 1.0]))      # zero-length segment: singular A
    with pytest.raises(AssertionError):
        raise_for_status(2, np.array([0.0, 1.0, 0.5]))      # negative step: reference asserts t >= 0
    with pytest.raises(np.linalg.LinAlgError):
        raise_for_status(3, t)


def test_yaw_from_quaternion_and_time_grid():
    for yaw in (-3.0, -1.2, 0.0, 0.4, 2.9):
        q = (0.0, 0.0, math.sin(yaw / 2), math.cos(yaw / 2))
        assert abs(dpg.yaw_from_quaternion(q) - yaw) < 1e-14
        assert dpg.yaw_from_quaternion(q) == O.quat_to_yaw(q)
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4)
        assert dpg.yaw_from_quaternion(q) == O.quat_to_yaw(q)
    t = dpg.path_time_grid(50)
    assert len(t) == 50 and t[0] == 0.0 and t[1] == 0.2 and abs(t[-1] - 9.8) < 1e-12
    np.testing.assert_array_equal(t, O.path_times(50))


def test_message_helpers_roundtrip():
    pos = np.arange(12.0).reshape(4, 3)
    quat = np.tile([0, 0, 0, 1.0], (4, 1))
    p = msgs.path_from_arrays(pos, quat)
    pos2, quat2 = msgs.path_to_arrays(p)
    np.testing.assert_array_equal(pos, pos2)
    np.testing.assert_array_equal(quat, quat2)
    mat = np.arange(66, dtype=np.float32).reshape(2, 33)
    m = dpg.build_message(mat, 7)
    assert m.cf_id == 7 and len(m.poly_x) == 16 and m.durations == [0.0, 33.0]
    assert m.poly_yaw[:2] == [25.0, 26.0]


def test_save_pol_matrix_format(tmp_path):
    mat = (np.arange(66, dtype=np.float32).reshape(2, 33) / 7).astype(np.float32)
    fn = dpg.save_pol_matrix(mat, 3, str(tmp_path))
    assert os.path.basename(fn) == "Pol_matrix_3.csv"
    first = open(fn).readline().split(",")
    assert len(first) == 33 and "e" in first[1] and len(first[1].split("e")[0]) == 20   # %.18e
    back = np.loadtxt(fn, delimiter=",")
    np.testing.assert_array_equal(back.astype(np.float32), mat)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 256, 4096, 65537):
        for world in (1, 2, 3, 8):
            spans = [swarm.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == swarm.shard_sizes(n, world)
    with pytest.raises(ValueError):
        swarm.shard_bounds(4, 2, 2)


def test_stl_roundtrip_and_box(tmp_path):
    box = stl.box_mesh((-2, 3.9, 0), (2, 4.1, 1.6))
    assert box.shape == (12, 3, 3)
    fn = str(tmp_path / "wall.stl")
    stl.save_stl(fn, box)
    assert os.path.getsize(fn) == 84 + 50 * 12      # the size of the reference's 12-triangle files
    back = stl.load_stl(fn)
    np.testing.assert_allclose(back, box.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(back, O.load_stl_binary(fn))
    assert back.min(axis=(0, 1)).tolist() == pytest.approx([-2, 3.9, 0], abs=1e-6)


def test_oracle_formation_and_collision_semantics():
    rb = np.array([[1.0, 2.0, 3.0, 0.0, 0.0, math.sin(0.25), math.cos(0.25)]])
    out = O.formation_transform(rb, np.array([[0.5, 0, 0], [-0.5, 0, 0]]))
    np.testing.assert_allclose(out[0, 0, :3], [1 + 0.5 * math.cos(0.5), 2 + 0.5 * math.sin(0.5), 3.0], atol=1e-15)
    np.testing.assert_allclose(out[1, 0, :3], [1 - 0.5 * math.cos(0.5), 2 - 0.5 * math.sin(0.5), 3.0], atol=1e-15)
    np.testing.assert_allclose(out[0, 0, 3:], rb[0, 3:], atol=1e-15)
    pos = np.zeros((3, 2, 3))
    pos[1, :, 0] = [1.0, 0.2]
    pos[2, :, 0] = 5.0
    md, partner, hit = O.formation_collide(pos, 0.15)
    assert md.tolist() == pytest.approx([0.2, 0.2, 4.0]) and partner.tolist() == [1, 0, 1]
    assert hit.tolist() == [True, True, False]
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0.0]]])
    assert O.point_triangle_dist2(np.array([0.2, 0.2, 0.5]), *tri[0]) == pytest.approx(0.25)
    assert O.point_triangle_dist2(np.array([-1.0, -1.0, 0.0]), *tri[0]) == pytest.approx(2.0)
    assert O.point_triangle_dist2(np.array([1.0, 1.0, 0.0]), *tri[0]) == pytest.approx(0.5)


def test_allocate_times_modes():
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 2, 0], [1, 2, 0], [4, 6, 0.0]])
    tu = dpg.allocate_times(pos, 10.0, "uniform")
    np.testing.assert_array_equal(tu, dpg.path_time_grid(5))
    td = dpg.allocate_times(pos, 10.0, "distance")
    assert td[0] == 0.0 and abs(td[-1] - 8.0) < 1e-12 and (np.diff(td) > 0).all()
    legs = np.diff(td)
    assert abs(legs[1] / legs[0] - 2.0) < 1e-12 and abs(legs[3] / legs[0] - 5.0) < 1e-12
    with pytest.raises(ValueError):
        dpg.allocate_times(pos, 10.0, "nope")


def test_bench_accounting_helpers():
    """bench.py's roofline arithmetic: SURVEY.md 8d bytes per trajectory and the kernel-name mirror."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_bytes(1, 10, 7) == 3080
    assert bench.algorithmic_bytes(1, 20, 7) == 6120
    assert bench.algorithmic_bytes(1, 10, 9) == 3720
    assert bench.algorithmic_bytes(256, 10, 7) == 788480
    assert bench.METRIC.startswith("minimum-snap trajectories/sec")
    tr, kn = bench.pmc_traffic(256, 10, 7)
    assert tr is None or tr > 788480 * 0.9


def test_nav_path_quaternion_and_arrays():
    """Host side of get_nav_path_msg: quaternion_from_euler(0, 0, yaw) for the 'sxyz' axes and the
    Trajectory -> array conversion (no GPU: the evaluator itself is tested with -m gpu)."""
    from drone_path_planning_python_amd.trajectory_visualising import Trajectory, quaternion_from_yaw, trajectory_arrays
    for yaw in (0.0, 0.3, -2.5, math.pi):
        q = quaternion_from_yaw(yaw)
        assert q[0] == 0.0 and q[1] == 0.0
        assert abs(q[2] - math.sin(yaw / 2)) < 1e-15 and abs(q[3] - math.cos(yaw / 2)) < 1e-15
        # the rotation it encodes turns x into (cos yaw, sin yaw, 0)
        v = O.quat_rotate(np.array(q), np.array([1.0, 0.0, 0.0]))
        np.testing.assert_allclose(v, [math.cos(yaw), math.sin(yaw), 0.0], atol=1e-15)
    tr = Trajectory()
    tr.loadcsv(os.path.join(GOLDEN_DIR, "traj.csv"))
    coef, dur = trajectory_arrays(tr)
    mat = np.loadtxt(os.path.join(GOLDEN_DIR, "traj.csv"), delimiter=",", skiprows=1, usecols=range(33))
    np.testing.assert_array_equal(coef[0].reshape(-1, 32), mat[:, 1:])
    np.testing.assert_array_equal(dur[0], mat[:, 0])
    ref = O.nav_path_poses(mat, 0.5)
    assert ref.shape == (len(np.arange(0, tr.duration, 0.5)), 7)


def test_fma_square_is_exactly_rounded():
    """The NumPy oracle's fused multiply-add (the pairwise pass's squared distance, include/msnap.h)
    against rational arithmetic: random metres, values whose sum lands half way between two doubles with
    a lower-order term breaking the tie, cancellation, and the C oracle's fma() on a whole swarm."""
    import c_oracle
    from fractions import Fraction
    rng = np.random.default_rng(12)
    y = rng.uniform(-50, 50, 3000)
    c = rng.uniform(0, 5000, 3000)
    c[:1000] *= 1e-6
    c[1000:1500] *= 1e6
    got = O.fma_square(y, c)
    assert all(float(Fraction(a) * Fraction(a) + Fraction(b)) == g for a, b, g in zip(y, c, got))
    # integers: y*y needs up to 60 bits, c a few units in the last place around the half-way points
    y = rng.integers(1 << 26, 1 << 30, size=3000).astype(np.float64)
    ce = rng.integers(50, 58, size=3000)
    c = (np.ldexp(rng.integers(1 << 20, 1 << 21, size=3000).astype(np.float64), ce - 20)
         + rng.integers(-3, 4, size=3000) * np.ldexp(1.0, ce - 53))
    got = O.fma_square(y, c)
    assert all(float(Fraction(int(a)) ** 2 + Fraction(b)) == g for a, b, g in zip(y, c, got))
    y = rng.uniform(1, 2, 3000)
    c = -(y * y) * (1 + rng.integers(-4, 5, size=3000) * 2.0 ** -52)
    got = O.fma_square(y, c)
    assert all(float(Fraction(a) * Fraction(a) + Fraction(b)) == g for a, b, g in zip(y, c, got))
    pos = rng.uniform(-3, 3, size=(90, 11, 3))
    pos[:30] = np.round(pos[:30] * 2) / 2
    for a, b in zip(O.formation_collide(pos, 0.3), c_oracle.formation_collide(pos, 0.3)):
        np.testing.assert_array_equal(a, b)


def test_device_solve_grid_refuses_batches_that_do_not_match_the_prepared_grid():
    """The library itself refuses a segment count that is not the prepared grid's (MSNAP_ESEGMENTS,
    tests/c_abi/abi_smoke.c); the torch wrapper additionally checks the tensor shapes before anything is launched (no
    GPU needed: the context is a stand-in that counts launches)."""
    import torch
    from drone_path_planning_python_amd import swarm

    class FakeCtx:
        ncoef = 8
        launches = 0

        def grid_waypoints(self):
            return 11

        def solve_grid_device(self, n, n_seg, wp, coef, dur, status):
            assert n_seg == 10 and tuple(coef.shape) == (n, 10, 4, 8)
            self.launches += 1

    comp = object.__new__(swarm.DeviceCompute)
    comp.ctx, comp.torch, comp.device = FakeCtx(), torch, torch.device("cpu")
    with pytest.raises(ValueError):
        comp.solve_grid(torch.zeros((4, 21, 4), dtype=torch.float64))      # 20 segments on a 10-segment grid
    with pytest.raises(ValueError):
        comp.solve_grid(torch.zeros((4, 11, 3), dtype=torch.float64))
    assert comp.ctx.launches == 0
    comp.solve_grid(torch.zeros((4, 11, 4), dtype=torch.float64))
    assert comp.ctx.launches == 1
