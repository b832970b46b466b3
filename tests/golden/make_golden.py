#!/usr/bin/env python3
"""Generate the fp64 golden vectors by running the REFERENCE itself.

Runs only in the build container (it imports /root/reference/src/optimizations,
unmodified, read-only; nothing of the reference is copied -- only inputs and
the outputs it computes are stored).  The GPU box never sees the reference; it
sees the .npz files this script writes next to itself.

    MPLBACKEND=Agg python3 -W ignore tests/golden/make_golden.py

Cases (names are the npz keys' prefixes):
  cfg1        BASELINE.json configs[0]: 4 waypoints, t = 0,1,3,4
  testdata    calculatingTrajectories.py:240-273 demo (18 waypoints, dt = 2.0)
  m1, m2      1- and 2-segment edge cases
  t0quirk     first time != 0 (Appendix A quirk: start rows evaluated at t[0])
  cfg2        config-2 shape: 64 seeded drones x 10 segments, per-drone times
  cfg2s       same waypoints on the reference's shared uniform grid
  m20         8 seeded drones x 20 segments
  m3          8 seeded drones x 3 segments
  path49      the 49-segment / T = 0.2 s stress (50 poses on the 10/n grid)
  pweval      PiecewisePolynomial.eval samples (knots, interior, past the end)
  flat        Trajectory.eval (differential flatness) on src/traj.csv rows
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
import optimizations as O  # noqa: E402  (the reference package)

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_solve(wp, t):
    """wp [m,4], t [m] -> coef [M,4,8], dur [M] via the reference."""
    pts = [O.Point_time(O.Waypoint(float(w[0]), float(w[1]), float(w[2]), float(w[3])), float(tt))
           for w, tt in zip(wp, t)]
    pols, pcs = O.calculate_trajectory4D(pts)
    M = len(pols[0])
    coef = np.empty((M, 4, 8))
    for a in range(4):
        for j in range(M):
            coef[j, a, :] = np.asarray(pols[a][j].p, dtype=np.float64).reshape(8)
    dur = np.array([float(x) for x in pcs[0].time_durations])
    return coef, dur, pcs


def ref_batch(wp, t):
    N = wp.shape[0]
    M = wp.shape[1] - 1
    coef = np.empty((N, M, 4, 8))
    dur = np.empty((N, M))
    for d in range(N):
        tt = t if t.ndim == 1 else t[d]
        coef[d], dur[d], _ = ref_solve(wp[d], tt)
    return coef, dur


def synth(seed, N, M, shared):
    """SURVEY.md 8d synthetic inputs."""
    rng = np.random.default_rng(seed)
    wp = np.empty((N, M + 1, 4))
    wp[..., :3] = rng.uniform(-5.0, 5.0, size=(N, M + 1, 3))
    wp[..., 3] = rng.uniform(-np.pi, np.pi, size=(N, M + 1))
    if shared:
        t = np.arange(M + 1) * (10.0 / (M + 1))
    else:
        T = rng.uniform(0.5, 2.0, size=(N, M))
        t = np.concatenate([np.zeros((N, 1)), np.cumsum(T, axis=1)], axis=1)
    return wp, t


def main():
    out = {}

    # cfg1
    wp = np.array([[0, 0, 0, 0], [2, 2.2, 0.3, 0], [4, 8, 0.8, 0], [1, 2, 0.4, 0.5]], dtype=np.float64)
    t = np.array([0.0, 1.0, 3.0, 4.0])
    c, d, _ = ref_solve(wp, t)
    out.update(cfg1_wp=wp, cfg1_t=t, cfg1_coef=c, cfg1_dur=d)

    # testdata (the reference's own __main__ demo inputs, read from the module)
    from optimizations import calculatingTrajectories as CT
    wp = np.array(CT.test_data, dtype=np.float64)
    t = np.arange(wp.shape[0]) * float(CT.timestep)
    c, d, pcs = ref_solve(wp, t)
    out.update(testdata_wp=wp, testdata_t=t, testdata_coef=c, testdata_dur=d,
               testdata_eval17=np.array([float(np.ravel(pcs[0].eval(17.0))[0])]))

    # edge cases
    wp = np.array([[1.0, -2.0, 0.5, 0.1], [3.0, 1.0, 1.5, -0.4]])
    t = np.array([0.0, 1.7])
    c, d, _ = ref_solve(wp, t)
    out.update(m1_wp=wp, m1_t=t, m1_coef=c, m1_dur=d)
    wp = np.array([[1.0, -2.0, 0.5, 0.1], [3.0, 1.0, 1.5, -0.4], [-1.0, 0.0, 2.0, 0.9]])
    t = np.array([0.0, 0.8, 2.9])
    c, d, _ = ref_solve(wp, t)
    out.update(m2_wp=wp, m2_t=t, m2_coef=c, m2_dur=d)

    # t[0] != 0 quirk
    wp, _ = synth(777, 1, 4, False)
    wp = wp[0]
    t = np.array([0.25, 1.5, 2.4, 3.9, 5.0])
    c, d, _ = ref_solve(wp, t)
    out.update(t0quirk_wp=wp, t0quirk_t=t, t0quirk_coef=c, t0quirk_dur=d)

    # config-2 shape, per-drone times and shared grid (seed per SURVEY 8d)
    wp, t = synth(20260104 + 2, 64, 10, False)
    c, d = ref_batch(wp, t)
    out.update(cfg2_wp=wp, cfg2_t=t, cfg2_coef=c, cfg2_dur=d)
    wps, ts = synth(20260104 + 2, 64, 10, True)
    c, d = ref_batch(wps, ts)
    out.update(cfg2s_wp=wps, cfg2s_t=ts, cfg2s_coef=c, cfg2s_dur=d)

    wp, t = synth(20260104 + 4, 8, 20, False)
    c, d = ref_batch(wp, t)
    out.update(m20_wp=wp, m20_t=t, m20_coef=c, m20_dur=d)
    wp, t = synth(20260104 + 1, 8, 3, False)
    c, d = ref_batch(wp, t)
    out.update(m3_wp=wp, m3_t=t, m3_coef=c, m3_dur=d)

    # 49 segments, T = 0.2: the ill-conditioned stress, on the 10/n grid of
    # drones_pols_generator.py:44-46 with a smooth synthetic path
    n = 50
    s = np.linspace(0.0, 1.0, n)
    wp = np.stack([0.5 + 0.1 * np.sin(6 * s), 3.0 + 2.0 * s, 1.0 + 0.3 * np.sin(3 * s + 0.2),
                   0.6 * s], axis=1)
    t = np.array([(10.0 / n) * i for i in range(n)])
    c, d, _ = ref_solve(wp, t)
    out.update(path49_wp=wp, path49_t=t, path49_coef=c, path49_dur=d)

    # PiecewisePolynomial.eval samples on the cfg1 x-axis
    wp = out["cfg1_wp"]
    _, _, pcs = ref_solve(wp, out["cfg1_t"])
    ts = np.array([0.0, 0.3, 1.0, 1.0 - 1e-12, 2.2, 3.0, 3.999, 4.0, 4.5, 6.0])
    ev = np.array([[float(np.ravel(pcs[a].eval(float(x)))[0]) for a in range(4)] for x in ts])
    out.update(pweval_t=ts, pweval_val=ev)

    # Trajectory.eval (flatness) on the reference's crazyswarm sample src/traj.csv
    tr = O.Trajectory()
    tr.loadcsv(os.path.join(REF, "src", "traj.csv"))
    ts = np.array([0.0, 0.5, 1.05, 2.0, 3.3, 5.0, float(tr.duration)])
    rows = []
    for x in ts:
        o = tr.eval(float(x))
        rows.append(np.concatenate([o.pos, o.vel, o.acc, o.omega, [o.yaw]]))
    out.update(flat_t=ts, flat_out=np.array(rows), flat_duration=np.array([float(tr.duration)]))

    np.savez_compressed(os.path.join(HERE, "ref_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_golden.npz"), "keys:", len(out))


if __name__ == "__main__":
    main()
