"""Host-side value types of the trajectory path.

The public names and their behaviour follow the reference's
src/optimizations/uav_trajectory.py so that code written against it keeps working:
Waypoint (:172-195), Point_time (:198-202), Point_time1D (:205-209), Polynomial (:12-36),
PiecewisePolynomial (:130-169), Polynomial4D (:49-85), Trajectory (:88-127),
TrajectoryOutput (:39-45), normalize (:6-9).  The code is this repository's own; the
batched equivalents run on the GPU (Context.solve_batch / sample / eval_flat).
"""
from __future__ import annotations

import numpy as np

GRAVITY = 9.81   # m/s^2, added to the z acceleration to obtain the thrust direction


# ----------------------------------------------------------------------------- records
class Waypoint:
    """x, y, z [m] and yaw [rad]; `getType(k)` picks one of them by the WP_TYPE_* index."""
    WP_TYPE_X, WP_TYPE_Y, WP_TYPE_Z, WP_TYPE_YAW = range(4)

    def __init__(self, x, y, z, yaw):
        self.x, self.y, self.z, self.yaw = x, y, z, yaw

    def as_tuple(self):
        return (self.x, self.y, self.z, self.yaw)

    def getType(self, type):
        fields = self.as_tuple()
        if type in (0, 1, 2, 3):
            return fields[type]
        print("Sorry, invalid type")   # the reference prints and falls through to None
        return None


class Point_time:
    """A waypoint and the absolute time at which it must be reached."""

    def __init__(self, wp: Waypoint, t: float):
        self.wp, self.t = wp, t


class Point_time1D:
    """Single-axis variant: a value and its time."""

    def __init__(self, wp: float, t: float):
        self.wp, self.t = wp, t


class TrajectoryOutput:
    """Flat outputs at one instant: pos [m], vel [m/s], acc [m/s^2], omega [rad/s], yaw [rad]."""
    __slots__ = ("pos", "vel", "acc", "omega", "yaw")

    def __init__(self):
        for name in self.__slots__:
            setattr(self, name, None)


def normalize(v):
    """Unit vector of v (asserts |v| > 0)."""
    length = np.linalg.norm(v)
    assert length > 0
    return v / length


# ----------------------------------------------------------------------------- polynomials
class Polynomial:
    """p[k] is the coefficient of t**k (ascending powers)."""

    def __init__(self, p):
        self.p = p

    def __len__(self):
        return len(self.p)

    def eval(self, t):
        """Horner's rule from the highest power down; negative t is rejected.
        An (n,1) coefficient array gives a shape-(1,) result, as in the reference."""
        assert t >= 0
        value = 0.0
        for k in range(len(self.p) - 1, -1, -1):
            value = value * t + self.p[k]
        return value

    def derivative(self) -> "Polynomial":
        """d/dt as a new Polynomial: coefficient k of the result is (k+1) * p[k+1]."""
        return Polynomial([(k + 1) * c for k, c in enumerate(self.p[1:])])

    def pol_coeffs_at_t(self, t):
        """The individual terms p[k] * t**k (not summed): with p = [1]*8 and repeated
        derivative() this is the row builder of the collocation matrix."""
        assert t >= 0
        terms = np.zeros(len(self.p))
        for k, c in enumerate(self.p):
            terms[k] = c * (t ** k)
        return terms


class PiecewisePolynomial:
    """Polynomials glued end to end.  eval(t) walks the cumulative durations with a strict
    '<' test; beyond the last knot the last piece is extrapolated."""

    def __init__(self, pols: list, time_durations: list):
        self.pols = pols
        self.time_durations = time_durations
        self.nOfPols = len(pols)

    def eval(self, t):
        assert t >= 0
        elapsed = 0
        for piece, length in zip(self.pols, self.time_durations):
            if t < elapsed + length:
                return piece.eval(t - elapsed)
            elapsed = elapsed + length
        return self.pols[-1].eval(t - sum(self.time_durations[:-1]))


class Polynomial4D:
    """One x / y / z / yaw piece with its duration; eval() gives the differential-flatness
    outputs of a quadrotor following it."""

    def __init__(self, duration, px, py, pz, pyaw):
        self.duration = duration
        self.px, self.py, self.pz, self.pyaw = (Polynomial(c) for c in (px, py, pz, pyaw))

    def derivative(self) -> "Polynomial4D":
        d = [q.derivative().p for q in (self.px, self.py, self.pz, self.pyaw)]
        return Polynomial4D(self.duration, *d)

    def _xyz(self, t):
        return np.array([self.px.eval(t), self.py.eval(t), self.pz.eval(t)])

    def eval(self, t) -> TrajectoryOutput:
        first = self.derivative()
        second = first.derivative()
        third = second.derivative()

        out = TrajectoryOutput()
        out.pos, out.vel, out.acc = self._xyz(t), first._xyz(t), second._xyz(t)
        out.yaw = self.pyaw.eval(t)
        yaw_rate = first.pyaw.eval(t)
        jerk = third._xyz(t)

        thrust = out.acc + np.array([0, 0, GRAVITY])
        thrust_len = np.linalg.norm(thrust)
        body_z = normalize(thrust)
        heading = np.array([np.cos(out.yaw), np.sin(out.yaw), 0])
        body_y = normalize(np.cross(body_z, heading))
        body_x = np.cross(body_y, body_z)
        h_w = (jerk - np.dot(jerk, body_z) * body_z) / thrust_len
        out.omega = np.array([-np.dot(h_w, body_y), np.dot(h_w, body_x), body_z[2] * yaw_rate])
        return out


class Trajectory:
    """A list of Polynomial4D pieces (rows [T | x8 | y8 | z8 | yaw8]); eval(t) finds the piece
    with a '<=' test on the cumulative durations (note: PiecewisePolynomial uses '<')."""

    def __init__(self):
        self.polynomials = None
        self.duration = None

    def n_pieces(self):
        return len(self.polynomials)

    def load_matrix(self, rows):
        rows = np.atleast_2d(np.asarray(rows))
        self.polynomials = [Polynomial4D(r[0], r[1:9], r[9:17], r[17:25], r[25:33]) for r in rows]
        self.duration = np.sum(rows[:, 0])

    def loadcsv(self, filename, skiprows: int = 1):
        """The reference always skips one row -- right for the crazyswarm traj.csv (header row),
        but it silently drops the first piece of the header-less Pol_matrix_*.csv files the
        generator node writes; pass skiprows=0 for those."""
        self.load_matrix(np.loadtxt(filename, delimiter=",", skiprows=skiprows, usecols=range(33)))

    def eval(self, t) -> TrajectoryOutput:
        assert t >= 0
        assert t <= self.duration
        begin = 0.0
        for piece in self.polynomials:
            if t <= begin + piece.duration:
                return piece.eval(t - begin)
            begin = begin + piece.duration
        return None
