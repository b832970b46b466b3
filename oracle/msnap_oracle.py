"""CPU oracle for the minimum-snap hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference algorithm.  It is the
checker the parity tests, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg compare the HIP path against.  Nothing under
``drone_path_planning_python_amd/`` imports it; the product path fails loudly
when the HIP library is missing instead of falling back to this file.

Pinning (see tests/test_oracle_golden.py):
  * fp64 golden vectors produced by importing the reference itself in the build
    container (tests/golden/make_golden.py -> tests/golden/ref_*.npz),
  * the config-1 known-answer vector of SURVEY.md Appendix C,
  * the reference's own float32 outputs resources/trajectories/Pol_matrix_*.csv
    (copied as data fixtures, ~1e-6 relative: float32 storage limit).
Unpinned parts (no reference implementation exists) say so in their docstring:
order 9, the two collision passes.

Every function cites the reference file:line it restates (paths relative to
the reference repository root).
"""
from __future__ import annotations

import math

import numpy as np

AXES = 4  # x, y, z, yaw  (uav_trajectory.py:179-182)


# --------------------------------------------------------------------------
# a3  Polynomial.pol_coeffs_at_t / .derivative   (uav_trajectory.py:25-36)
# --------------------------------------------------------------------------
def deriv_row(j: int, t: float, ncoef: int = 8) -> np.ndarray:
    """Row of the collocation matrix: j-th derivative of sum_k c_k t^k at t.

    The reference builds it as ``Polynomial([1]*8)`` differentiated j times
    (uav_trajectory.py:25-26: p'[i] = (i+1) p[i+1]), evaluated term-wise
    (uav_trajectory.py:28-36: p[i] * t**i) and left-padded with j zeros
    (calculatingTrajectories.py:68-71, 93-98).  Entry k is therefore
    k!/(k-j)! * t**(k-j) for k >= j, 0 otherwise; Python's 0**0 == 1 makes the
    t == 0 row equal j! at column j.
    """
    row = np.zeros(ncoef)
    for k in range(j, ncoef):
        f = 1.0
        for q in range(k - j + 1, k + 1):
            f *= q
        row[k] = f * (t ** (k - j))
    return row


# --------------------------------------------------------------------------
# a1  calculate_trajectory1D   (calculatingTrajectories.py:37-197)
# --------------------------------------------------------------------------
def assemble_1d(times: np.ndarray, values: np.ndarray, ncoef: int = 8):
    """Dense (ncoef*M x ncoef*M) collocation system of one axis.

    Row layout follows calculatingTrajectories.py:55-131 for ncoef == 8
    (k = ncoef/2 = 4):
      rows 0..k-1                 d^0..d^(k-1) of segment 0 at t = times[0]
                                  (:65-73; note: evaluated at the ABSOLUTE first
                                  time, not at 0 -- Appendix A quirk, kept)
      interior waypoint i, base = k + (i-1)*ncoef  (:112-113):
        base .. base+ncoef-3      d^1..d^(ncoef-2): prev(T_{i-1}) - next(0) = 0
                                  (:115-119)
        base+ncoef-2              prev(T_{i-1}) = wp_i     (:124,127)
        base+ncoef-1              next(0)       = wp_i     (:125,128)
      last k rows                 d^0..d^(k-1) of segment M-1 at T_{M-1}
                                  (:75-79, 87)
    For ncoef != 8 this is the natural order-(2k-1) generalisation (no
    reference exists: parity unpinned).
    """
    times = np.asarray(times, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    m = times.shape[0]
    M = m - 1
    k = ncoef // 2
    n = ncoef * M
    A = np.zeros((n, n))
    b = np.zeros(n)
    prev_t = 0.0
    for i in range(m):
        t = times[i] - prev_t              # :59
        if i == 0 or i == M:               # :65
            for j in range(k):
                row = deriv_row(j, t, ncoef)
                if i == 0:
                    A[j, 0:ncoef] = row                          # :73
                else:
                    A[n - k + j, ncoef * (M - 1):ncoef * M] = row  # :75-79
            if i == 0:
                b[0] = values[0]                                 # :82-85
            else:
                b[n - k] = values[i]                             # :87
        else:
            base = k + (i - 1) * ncoef                           # :112
            for j in range(1, ncoef - 1):
                A[base + j - 1, ncoef * (i - 1):ncoef * i] = deriv_row(j, t, ncoef)
                A[base + j - 1, ncoef * i:ncoef * (i + 1)] = -deriv_row(j, 0.0, ncoef)
            A[base + ncoef - 2, ncoef * (i - 1):ncoef * i] = deriv_row(0, t, ncoef)
            A[base + ncoef - 1, ncoef * i:ncoef * (i + 1)] = deriv_row(0, 0.0, ncoef)
            b[base + ncoef - 2] = values[i]
            b[base + ncoef - 1] = values[i]
        prev_t = times[i]                                        # :131
    return A, b


def calculate_trajectory1D(times, values, ncoef: int = 8) -> np.ndarray:
    """One axis: coefficients [M, ncoef], ascending powers (:137, :141-144)."""
    A, b = assemble_1d(times, values, ncoef)
    c = np.linalg.solve(A, b)                                    # :137
    return c.reshape(-1, ncoef)


# --------------------------------------------------------------------------
# a2  calculate_trajectory4D   (calculatingTrajectories.py:200-213)
# --------------------------------------------------------------------------
def calculate_trajectory4D(wp: np.ndarray, times: np.ndarray, ncoef: int = 8):
    """wp [m,4] (x,y,z,yaw), times [m] absolute -> (coef [M,4,ncoef], dur [M]).

    Four independent 1-D solves on the same time grid (:203-206); durations
    are the successive time differences the reference collects in
    ``time_points`` (:59-61, :191).
    """
    wp = np.asarray(wp, dtype=np.float64)
    times = np.asarray(times, dtype=np.float64)
    M = times.shape[0] - 1
    coef = np.empty((M, AXES, ncoef))
    for a in range(AXES):
        coef[:, a, :] = calculate_trajectory1D(times, wp[:, a], ncoef)
    return coef, np.diff(times)


def solve_batch(wp: np.ndarray, times: np.ndarray, ncoef: int = 8):
    """Batched a2: wp [N,m,4]; times [N,m] or shared [m]."""
    wp = np.asarray(wp, dtype=np.float64)
    times = np.asarray(times, dtype=np.float64)
    N, m, _ = wp.shape
    coef = np.empty((N, m - 1, AXES, ncoef))
    dur = np.empty((N, m - 1))
    for d in range(N):
        t = times if times.ndim == 1 else times[d]
        coef[d], dur[d] = calculate_trajectory4D(wp[d], t, ncoef)
    return coef, dur


def solve_batch_fast(wp: np.ndarray, times: np.ndarray, ncoef: int = 8):
    """Same result as solve_batch, one LU per drone for the 4 axes.

    "B1 optimised CPU" baseline of BASELINE.md 3: the matrix of a1 does not
    depend on the axis, so it is assembled once and solved for 4 right-hand
    sides.  Used where the faithful version would take minutes.
    """
    wp = np.asarray(wp, dtype=np.float64)
    times = np.asarray(times, dtype=np.float64)
    N, m, _ = wp.shape
    M = m - 1
    coef = np.empty((N, M, AXES, ncoef))
    dur = np.empty((N, M))
    cache = None
    for d in range(N):
        t = times if times.ndim == 1 else times[d]
        if times.ndim == 1 and cache is not None:
            A = cache
        else:
            A, _ = assemble_1d(t, wp[d, :, 0], ncoef)
            cache = A
        B = np.zeros((ncoef * M, AXES))
        for a in range(AXES):
            _, B[:, a] = _rhs_only(t, wp[d, :, a], ncoef)
        X = np.linalg.solve(A, B)
        coef[d] = X.reshape(M, ncoef, AXES).transpose(0, 2, 1)
        dur[d] = np.diff(t)
    return coef, dur


def _rhs_only(times, values, ncoef):
    m = len(times)
    M = m - 1
    k = ncoef // 2
    n = ncoef * M
    b = np.zeros(n)
    b[0] = values[0]
    b[n - k] = values[M]
    for i in range(1, M):
        base = k + (i - 1) * ncoef
        b[base + ncoef - 2] = values[i]
        b[base + ncoef - 1] = values[i]
    return None, b


# --------------------------------------------------------------------------
# a4 / a5  Polynomial.eval, PiecewisePolynomial.eval  (uav_trajectory.py:17-22,154-169)
# --------------------------------------------------------------------------
def poly_eval(c: np.ndarray, t: float) -> float:
    """Horner on ascending coefficients (uav_trajectory.py:17-22)."""
    assert t >= 0
    x = 0.0
    for i in range(len(c)):
        x = x * t + c[len(c) - 1 - i]
    return x


def piecewise_eval(coef: np.ndarray, dur: np.ndarray, t: float) -> float:
    """coef [M, ncoef] of one axis.  Segment lookup with strict '<'
    (uav_trajectory.py:165) and extrapolation of the last piece beyond the end
    (uav_trajectory.py:161-163)."""
    assert t >= 0
    acc = 0.0
    M = len(dur)
    for i in range(M):
        if t < acc + dur[i]:
            return poly_eval(coef[i], t - acc)
        acc = acc + dur[i]
    return poly_eval(coef[M - 1], t - sum(float(x) for x in dur[:-1]))


def sample_positions(coef: np.ndarray, dur: np.ndarray, dt: float, n_samples: int,
                     naxes: int = 3) -> np.ndarray:
    """coef [N,M,4,ncoef], dur [N,M] -> pos [N, n_samples, naxes] at t = s*dt
    with a5 semantics (the sampling convention of path_vis.py:28 /
    visualization.py:53: np.arange(0, duration, timestep))."""
    N = coef.shape[0]
    out = np.empty((N, n_samples, naxes))
    for d in range(N):
        for s in range(n_samples):
            for a in range(naxes):
                out[d, s, a] = piecewise_eval(coef[d, :, a, :], dur[d], s * dt)
    return out


# --------------------------------------------------------------------------
# a7  path_to_pol   (scripts/drones_pols_generator.py:40-90)
# --------------------------------------------------------------------------
def quat_to_yaw(q) -> float:
    """tf.transformations.euler_from_quaternion(q)[2], default axes 'sxyz'
    (drones_pols_generator.py:51-53).  tf is not vendored in the reference;
    this is the published algorithm of transformations.py (quaternion_matrix
    followed by euler_from_matrix for 'sxyz'): with the homogeneous matrix Rm,
    cy = sqrt(Rm[0,0]^2 + Rm[1,0]^2); az = atan2(Rm[1,0], Rm[0,0]) if
    cy > 4*eps else 0.  q = (x, y, z, w)."""
    x, y, z, w = (float(v) for v in q)
    n = x * x + y * y + z * z + w * w
    eps = np.finfo(float).eps * 4.0
    if n < eps:
        r00, r10 = 1.0, 0.0
    else:
        s = 2.0 / n   # quaternion_matrix: q *= sqrt(2/n); outer(q, q)
        r00 = 1.0 - s * (y * y + z * z)
        r10 = s * (x * y + z * w)
    cy = math.sqrt(r00 * r00 + r10 * r10)
    if cy > eps:
        return math.atan2(r10, r00)
    return 0.0


def path_times(n_poses: int, total_duration: float = 10.0) -> np.ndarray:
    """Uniform grid t_i = i * total_duration / n_poses
    (drones_pols_generator.py:44-46, 56) -- the last waypoint is at
    total*(n-1)/n, not at total."""
    time_step = total_duration / n_poses
    return np.array([time_step * i for i in range(n_poses)])


def pack_pol_matrix(coef: np.ndarray, dur: np.ndarray) -> np.ndarray:
    """coef [M,4,ncoef], dur [M] -> float32 [M, 1+4*ncoef] =
    [T | x c0.. | y | z | yaw]   (drones_pols_generator.py:63-77)."""
    M, _, ncoef = coef.shape
    mat = np.zeros((M, 1 + AXES * ncoef), dtype=np.float32)
    for a in range(AXES):
        mat[:, 1 + ncoef * a:1 + ncoef * (a + 1)] = coef[:, a, :]
    mat[:, 0] = dur
    return mat


def path_to_pol(positions: np.ndarray, quats: np.ndarray, total_duration: float = 10.0):
    """positions [m,3], quats [m,4] (x,y,z,w) -> (matrix f32 [M,33], coef, dur)
    (drones_pols_generator.py:40-77)."""
    m = positions.shape[0]
    t = path_times(m, total_duration)
    wp = np.empty((m, 4))
    wp[:, :3] = positions
    for i in range(m):
        wp[i, 3] = quat_to_yaw(quats[i])
    coef, dur = calculate_trajectory4D(wp, t)
    return pack_pol_matrix(coef, dur), coef, dur


# --------------------------------------------------------------------------
# a8  transform   (scripts/drones_traj_generator.py:56-89)
# --------------------------------------------------------------------------
def quat_mul(a, b):
    """Hamilton product a (x) b, (x,y,z,w) layout."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz])


def quat_rotate(q, v):
    """Rotate v by unit quaternion q via the rotation matrix PyKDL builds in
    Rotation.Quaternion (do_transform_pose: tf2_geometry_msgs builds
    KDL.Frame(Rotation.Quaternion(q), Vector(t)) and multiplies the pose frame;
    tf2 / PyKDL are not vendored -- published algorithm, parity unpinned)."""
    x, y, z, w = q
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    R = np.array([
        [w2 + x2 - y2 - z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
        [2 * x * y + 2 * w * z, w2 - x2 + y2 - z2, 2 * y * z - 2 * w * x],
        [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, w2 - x2 - y2 + z2]])
    return R @ np.asarray(v, dtype=np.float64)


def formation_transform(rb_pose: np.ndarray, offsets: np.ndarray,
                        offset_quat=(0.0, 0.0, 0.0, 1.0)) -> np.ndarray:
    """rb_pose [P,7] (xyz + qx qy qz qw), offsets [K,3] body frame ->
    out [K,P,7]: p' = R(q_rb) p_k + t_rb ; q' = q_rb (x) q_k
    (drones_traj_generator.py:67-82; the reference hard-codes K = 2 offsets
    (+-0.5,0,0) with identity orientation, :22-38)."""
    rb_pose = np.asarray(rb_pose, dtype=np.float64)
    offsets = np.asarray(offsets, dtype=np.float64)
    P = rb_pose.shape[0]
    K = offsets.shape[0]
    out = np.empty((K, P, 7))
    for k in range(K):
        for p in range(P):
            q = rb_pose[p, 3:7]
            R = kdl_rotation_from_quaternion(q)
            out[k, p, 0:3] = R @ offsets[k] + rb_pose[p, 0:3]
            # drone orientation is identity (drones_traj_generator.py:31,38): the
            # result frame's rotation is R itself, returned as KDL's GetQuaternion
            out[k, p, 3:7] = kdl_get_quaternion(R)
    return out


def kdl_rotation_from_quaternion(q) -> np.ndarray:
    """orocos_kdl Rotation::Quaternion(x,y,z,w) (frames.cpp; not vendored in the
    reference, published algorithm): no normalisation of q."""
    x, y, z, w = (float(v) for v in q)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    return np.array([
        [w2 + x2 - y2 - z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
        [2 * x * y + 2 * w * z, w2 - x2 + y2 - z2, 2 * y * z - 2 * w * x],
        [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, w2 - x2 - y2 + z2]])


def kdl_get_quaternion(R) -> np.ndarray:
    """orocos_kdl Rotation::GetQuaternion (frames.cpp): (x, y, z, w)."""
    trace = R[0, 0] + R[1, 1] + R[2, 2]
    if trace > 1e-12:
        s = 0.5 / math.sqrt(trace + 1.0)
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, 0.25 / s])
    if R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = 2.0 * math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2])
        return np.array([0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s])
    if R[1, 1] > R[2, 2]:
        s = 2.0 * math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2])
        return np.array([(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s])
    s = 2.0 * math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1])
    return np.array([(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s])


# --------------------------------------------------------------------------
# f1  Polynomial4D.eval  (uav_trajectory.py:55-85)  differential flatness
# --------------------------------------------------------------------------
def poly_derivative(c: np.ndarray) -> np.ndarray:
    """uav_trajectory.py:25-26."""
    return np.array([(i + 1) * c[i + 1] for i in range(len(c) - 1)])


def polynomial4d_eval(px, py, pz, pyaw, t: float):
    """Returns pos[3], vel[3], acc[3], omega[3], yaw  (uav_trajectory.py:55-85)."""
    P = [np.asarray(px, float), np.asarray(py, float), np.asarray(pz, float), np.asarray(pyaw, float)]
    pos = np.array([poly_eval(P[0], t), poly_eval(P[1], t), poly_eval(P[2], t)])
    yaw = poly_eval(P[3], t)
    D1 = [poly_derivative(p) for p in P]
    vel = np.array([poly_eval(D1[0], t), poly_eval(D1[1], t), poly_eval(D1[2], t)])
    dyaw = poly_eval(D1[3], t)
    D2 = [poly_derivative(p) for p in D1]
    acc = np.array([poly_eval(D2[0], t), poly_eval(D2[1], t), poly_eval(D2[2], t)])
    D3 = [poly_derivative(p) for p in D2]
    jerk = np.array([poly_eval(D3[0], t), poly_eval(D3[1], t), poly_eval(D3[2], t)])
    thrust = acc + np.array([0.0, 0.0, 9.81])
    nt = np.linalg.norm(thrust)
    z_body = thrust / nt
    x_world = np.array([math.cos(yaw), math.sin(yaw), 0.0])
    yb = np.cross(z_body, x_world)
    y_body = yb / np.linalg.norm(yb)
    x_body = np.cross(y_body, z_body)
    jerk_orth = jerk - np.dot(jerk, z_body) * z_body
    h_w = jerk_orth / nt
    omega = np.array([-np.dot(h_w, y_body), np.dot(h_w, x_body), z_body[2] * dyaw])
    return pos, vel, acc, omega, yaw


def trajectory_eval(matrix: np.ndarray, t: float):
    """Trajectory.eval (uav_trajectory.py:119-127): rows [T | x8 | y8 | z8 | yaw8],
    segment lookup with '<=' (differs from a5's strict '<')."""
    assert t >= 0
    duration = float(np.sum(matrix[:, 0]))
    assert t <= duration
    cur = 0.0
    for row in matrix:
        if t <= cur + row[0]:
            return polynomial4d_eval(row[1:9], row[9:17], row[17:25], row[25:33], t - cur)
        cur = cur + row[0]
    return None


def nav_path_poses(matrix: np.ndarray, timestep: float, offset=(0.0, 0.0, 0.0)) -> np.ndarray:
    """get_nav_path_msg (src/trajectory_visualising/visualization.py:39-71) as an array [n, 7] of
    (x, y, z, qx, qy, qz, qw): one row per t in np.arange(0, duration, timestep) (:53), position =
    Trajectory.eval(t).pos + offset (:58-60), orientation = quaternion_from_euler(0, 0, -yaw) (:62-63).
    tf is not vendored in the reference; for roll = pitch = 0 its published 'sxyz' algorithm gives
    (0, 0, sin(yaw/2), cos(yaw/2)) -- parity unpinned."""
    duration = float(np.sum(matrix[:, 0]))
    rows = []
    for t in np.arange(0, duration, timestep):
        pos, _, _, _, yaw = trajectory_eval(matrix, float(t))
        h = 0.5 * (-yaw)
        rows.append([pos[0] + offset[0], pos[1] + offset[1], pos[2] + offset[2], 0.0, 0.0, math.sin(h), math.cos(h)])
    return np.array(rows).reshape(-1, 7)


def snap_cost(coef: np.ndarray, dur: np.ndarray) -> np.ndarray:
    """J = sum_seg int_0^T (p^(k))^2 dt per axis, k = ncoef/2, by exact integration of the
    squared k-th derivative polynomial.  coef [M,4,ncoef], dur [M] -> [4].  (The reference
    never evaluates its objective; the conditions it solves are the stationarity conditions
    of this functional -- calculatingTrajectories.py:13-33.)"""
    M, _, nc = coef.shape
    k = nc // 2
    out = np.zeros(4)
    for a in range(4):
        for i in range(M):
            p = np.polynomial.Polynomial(coef[i, a])
            q = p.deriv(k)
            sq = (q * q).integ()
            out[a] += sq(dur[i]) - sq(0.0)
    return out


# --------------------------------------------------------------------------
# Collision passes -- NEW capability, no reference implementation: PARITY
# UNPINNED.  Semantics defined by this repo:
# drones are spheres of radius r sampled every dt with a5 semantics.
# --------------------------------------------------------------------------
def _two_sum(a, b):
    """a + b = s + e exactly (Knuth)."""
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def fma_square(y: np.ndarray, c: np.ndarray) -> np.ndarray:
    """fma(y, y, c) = y*y + c with ONE rounding, vectorised in plain float64 arithmetic.

    y*y = p + e exactly (Dekker's product with Veltkamp splitting); c + p + e is then accumulated with
    error-free sums, so that the last addition rounds the exact value once.  The one configuration a final
    floating-point addition gets wrong -- the leading part lands exactly half way between two doubles and
    a lower-order term breaks the tie -- is detected and fixed.  Valid while nothing overflows or
    underflows (positions are metres)."""
    y = np.asarray(y, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    t = 134217729.0 * y                  # 2^27 + 1
    h = t - (t - y)
    lo = y - h
    p = y * y
    e = ((h * h - p) + 2.0 * (h * lo)) + lo * lo
    s1, t1 = _two_sum(c, p)
    s2, t2 = _two_sum(t1, e)             # exact value = s1 + s2 + t2
    r, u = _two_sum(s1, s2)              # r = fl(s1 + s2), exact value = r + u + t2
    out = r + (u + t2)
    # u exactly half an ulp of r (r + 2u is the neighbouring double) and t2 on the same side: the exact value
    # is beyond the half-way point, but fl(u + t2) == u ties back to r
    step = (r + 2.0 * u) - r
    tie = (u != 0.0) & (step == 2.0 * u) & (t2 != 0.0) & (np.sign(t2) == np.sign(u)) & ((u + t2) == u)
    out = np.where(tie, r + 2.0 * u, out)
    bad = ~(np.isfinite(y) & np.isfinite(c))
    return np.where(bad, y * y + c, out)


def formation_collide(pos: np.ndarray, radius: float):
    """pos [N,S,3] -> (min_dist [N] over other drones and samples, partner [N],
    hit [N] bool: min_dist < 2r).  partner = lowest index attaining the min.
    Squared distances are fma(dz, dz, fma(dy, dy, dx*dx)) (include/msnap.h: the differences and dx*dx
    rounded once each, then two fused multiply-adds -- `fma_square` restates them exactly);
    NaN samples never win a minimum."""
    N = pos.shape[0]
    mind = np.full(N, np.inf)
    partner = np.full(N, -1, dtype=np.int32)
    for i in range(N):
        d = pos - pos[i][None, :, :]
        dx2 = d[..., 0] * d[..., 0]
        d2 = np.fmin.reduce(fma_square(d[..., 2], fma_square(d[..., 1], dx2)), axis=1)
        d2 = np.where(np.isnan(d2), np.inf, d2)
        d2[i] = np.inf
        j = int(np.argmin(d2))
        if N > 1 and d2[j] < np.inf:
            mind[i] = math.sqrt(d2[j])
            partner[i] = j
    return mind, partner, mind < 2.0 * radius


def point_triangle_dist2(p, a, b, c) -> float:
    """Squared distance from point p to triangle abc (closest-point regions,
    Ericson, Real-Time Collision Detection 5.1.5)."""
    ab = b - a
    ac = c - a
    ap = p - a
    d1 = ab @ ap
    d2 = ac @ ap
    if d1 <= 0 and d2 <= 0:
        q = a
    else:
        bp = p - b
        d3 = ab @ bp
        d4 = ac @ bp
        if d3 >= 0 and d4 <= d3:
            q = b
        else:
            vc = d1 * d4 - d3 * d2
            if vc <= 0 and d1 >= 0 and d3 <= 0:
                q = a + (d1 / (d1 - d3)) * ab
            else:
                cp = p - c
                d5 = ab @ cp
                d6 = ac @ cp
                if d6 >= 0 and d5 <= d6:
                    q = c
                else:
                    vb = d5 * d2 - d1 * d6
                    if vb <= 0 and d2 >= 0 and d6 <= 0:
                        q = a + (d2 / (d2 - d6)) * ac
                    else:
                        va = d3 * d6 - d5 * d4
                        if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
                            q = b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b)
                        else:
                            denom = 1.0 / (va + vb + vc)
                            q = a + ab * (vb * denom) + ac * (vc * denom)
    e = p - q
    return float(e @ e)


def mesh_sweep(pos: np.ndarray, tris: np.ndarray, radius: float):
    """pos [N,S,3], tris [Tn,3,3] -> (min_dist [N], hit [N] bool: min_dist < r)."""
    N, S, _ = pos.shape
    mind = np.full(N, np.inf)
    tris = np.asarray(tris, dtype=np.float64)
    for i in range(N):
        best = np.inf
        for s in range(S):
            for t in tris:
                d2 = point_triangle_dist2(pos[i, s], t[0], t[1], t[2])
                if d2 < best:
                    best = d2
        mind[i] = math.sqrt(best)
    return mind, mind < radius


def load_stl_binary(path: str) -> np.ndarray:
    """Binary STL (80-byte header, u32 count, 50 B / triangle) -> float64 [Tn,3,3]."""
    with open(path, 'rb') as f:
        raw = f.read()
    n = int(np.frombuffer(raw, dtype='<u4', count=1, offset=80)[0])
    rec = np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('attr', '<u2')])
    body = np.frombuffer(raw, dtype=rec, count=n, offset=84)
    return body['v'].astype(np.float64)


# --------------------------------------------------------------------------
# f4: batched isStateValid (RB_planning_sep_coll_check.py:208-226, fcl_checker.py:93-100).
# python-fcl is not vendored: PARITY UNPINNED.  Predicate: closed triangles intersect,
# by the 17-axis separating-axis test; same operation order as the kernel.
# --------------------------------------------------------------------------
def _separates(P, Q, L):
    p = [P[v][0] * L[0] + P[v][1] * L[1] + P[v][2] * L[2] for v in range(3)]
    q = [Q[v][0] * L[0] + Q[v][1] * L[1] + Q[v][2] * L[2] for v in range(3)]
    return min(p) > max(q) or max(p) < min(q)


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def tri_tri_intersect(P, Q) -> bool:
    e = [[P[1][c] - P[0][c] for c in range(3)], [P[2][c] - P[1][c] for c in range(3)],
         [P[0][c] - P[2][c] for c in range(3)]]
    f = [[Q[1][c] - Q[0][c] for c in range(3)], [Q[2][c] - Q[1][c] for c in range(3)],
         [Q[0][c] - Q[2][c] for c in range(3)]]
    n1 = _cross(e[0], e[1])
    if _separates(P, Q, n1):
        return False
    n2 = _cross(f[0], f[1])
    if _separates(P, Q, n2):
        return False
    for i in range(3):
        for j in range(3):
            if _separates(P, Q, _cross(e[i], f[j])):
                return False
    for i in range(3):
        if _separates(P, Q, _cross(n1, e[i])):
            return False
        if _separates(P, Q, _cross(n2, f[i])):
            return False
    return True


def mesh_validity(states: np.ndarray, robot_tris: np.ndarray, env_tris: np.ndarray) -> np.ndarray:
    """states [N,4] (x,y,z,yaw) -> valid [N] bool (True = no collision)."""
    out = np.empty(states.shape[0], dtype=bool)
    R = np.asarray(robot_tris, dtype=np.float64).tolist()
    E = np.asarray(env_tris, dtype=np.float64).tolist()
    for n, (tx, ty, tz, yaw) in enumerate(np.asarray(states, dtype=np.float64).tolist()):
        qz, qw = math.sin(0.5 * yaw), math.cos(0.5 * yaw)
        c, s2 = 1.0 - 2.0 * (qz * qz), 2.0 * (qz * qw)
        hit = False
        for tri in R:
            P = [[(c * v[0] - s2 * v[1]) + tx, (s2 * v[0] + c * v[1]) + ty, v[2] + tz] for v in tri]
            for Q in E:
                if tri_tri_intersect(P, Q):
                    hit = True
                    break
            if hit:
                break
        out[n] = not hit
    return out
