"""calculate_trajectory1D / calculate_trajectory4D on the MI355X.

Drop-in for the reference's src/optimizations/calculatingTrajectories.py:37-213:
same arguments (a list of Point_time), same return shape (lists of Polynomial
whose ``.p`` is an (8,1) float64 array + PiecewisePolynomial with Python-float
durations), same exceptions (numpy.linalg.LinAlgError for a singular system,
AssertionError for negative times -- the reference raises those from
np.linalg.solve :137 and uav_trajectory.py:30).  The arithmetic runs in
libmsnap.so (HIP, gfx950); there is no CPU path here.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from ..context import ST_NONFINITE, ST_OK, ST_SINGULAR, ST_TIMES, Context, default_context
from .uav_trajectory import PiecewisePolynomial, Point_time, Polynomial, Waypoint

ORDER = 7            # "A 7th rank polynomial is used" (reference :15)


def waypoints_to_arrays(waypoints: Sequence[Point_time]):
    """list of Point_time -> (wp [m,4], t [m]) float64."""
    m = len(waypoints)
    wp = np.empty((m, 4), dtype=np.float64)
    t = np.empty((m,), dtype=np.float64)
    for i, pt in enumerate(waypoints):
        w = pt.wp
        wp[i, 0] = w.getType(Waypoint.WP_TYPE_X)
        wp[i, 1] = w.getType(Waypoint.WP_TYPE_Y)
        wp[i, 2] = w.getType(Waypoint.WP_TYPE_Z)
        wp[i, 3] = w.getType(Waypoint.WP_TYPE_YAW)
        t[i] = pt.t
    return wp, t


def raise_for_status(status: int, t: np.ndarray) -> None:
    """Map a per-drone msnap_status onto the exception the reference raises."""
    if status == ST_OK:
        return
    if status == ST_TIMES:
        steps = np.diff(np.concatenate([[0.0], t]))
        # negative local time trips `assert t >= 0` (uav_trajectory.py:30);
        # a zero-length segment makes A singular (LinAlgError from :137)
        assert not np.any(steps < 0), "negative time step"
        raise np.linalg.LinAlgError("Singular matrix")
    if status == ST_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    if status == ST_NONFINITE:
        raise np.linalg.LinAlgError("non-finite waypoint or time")
    raise RuntimeError(f"unknown msnap status {status}")


def solve_arrays(wp: np.ndarray, t: np.ndarray, ctx: Context | None = None):
    """wp [m,4], t [m] -> coef [M,4,8], dur [M] for ONE trajectory (N = 1 batch)."""
    ctx = ctx or default_context(ORDER)
    coef, dur, status = ctx.solve_batch(wp[None], t[None])
    raise_for_status(int(status[0]), t)
    return coef[0], dur[0]


def _wrap_axis(coef_axis: np.ndarray, dur: np.ndarray):
    pols = [Polynomial(np.array(coef_axis[j], dtype=np.float64).reshape(-1, 1)) for j in range(coef_axis.shape[0])]
    return pols, PiecewisePolynomial(pols, [float(x) for x in dur])


def calculate_trajectory1D(waypoints: Sequence[Point_time], wp_type=Waypoint.WP_TYPE_X, ctx: Context | None = None):
    """One axis of the trajectory through `waypoints` (reference :37-197)."""
    wp, t = waypoints_to_arrays(waypoints)
    coef, dur = solve_arrays(wp, t, ctx)
    return _wrap_axis(coef[:, int(wp_type), :], dur)


def calculate_trajectory4D(waypoints: Sequence[Point_time], ctx: Context | None = None):
    """x, y, z, yaw trajectories through `waypoints` (reference :200-213).

    Returns (pols_coeffs, pc_pols): 4 lists of Polynomial and 4
    PiecewisePolynomial, ordered x, y, z, yaw."""
    wp, t = waypoints_to_arrays(waypoints)
    coef, dur = solve_arrays(wp, t, ctx)
    pols_coeffs, pc_pols = [], []
    for axis in range(4):
        pols, pc = _wrap_axis(coef[:, axis, :], dur)
        pols_coeffs.append(pols)
        pc_pols.append(pc)
    return pols_coeffs, pc_pols


# the reference's demo inputs are data, kept for the known-answer test
timestep = 100 / 50
