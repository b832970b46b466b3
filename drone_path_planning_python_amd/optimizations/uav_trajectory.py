"""Host-side value types of the trajectory path.

Same names and semantics as the reference's src/optimizations/uav_trajectory.py
(Polynomial :12-36, Polynomial4D :49-85, Trajectory :88-127,
PiecewisePolynomial :130-169, Waypoint :172-195, Point_time :198-202,
Point_time1D :205-209) so code written against the reference keeps working;
the implementation is this repo's own.  These are small per-object helpers; the
batched equivalents run on the GPU (Context.sample, Context.solve_batch).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np


def normalize(v):
    """v / |v|; asserts a non-zero norm (reference :6-9)."""
    n = float(np.linalg.norm(v))
    assert n > 0
    return np.asarray(v) / n


class Polynomial:
    """Coefficients in ASCENDING powers: p[k] multiplies t**k."""

    def __init__(self, p):
        self.p = p

    def __len__(self):
        return len(self.p)

    def eval(self, t):
        """Horner evaluation, highest power first; t must be >= 0 (reference :17-22).
        With an (n,1) coefficient array the result has shape (1,), as in the reference."""
        assert t >= 0
        acc = 0.0
        for c in reversed(range(len(self.p))):
            acc = acc * t + self.p[c]
        return acc

    def derivative(self) -> "Polynomial":
        """d/dt: q[k] = (k+1) p[k+1]  (reference :25-26)."""
        return Polynomial([(k + 1) * self.p[k + 1] for k in range(len(self.p) - 1)])

    def pol_coeffs_at_t(self, t):
        """Term-wise values p[k] * t**k -- the collocation row builder (reference :28-36)."""
        assert t >= 0
        n = len(self.p)
        out = np.zeros(n)
        for k in range(n):
            out[k] = self.p[k] * (t ** k)
        return out


class TrajectoryOutput:
    """pos [m], vel [m/s], acc [m/s^2], omega [rad/s], yaw [rad] (reference :39-45)."""

    def __init__(self):
        self.pos = None
        self.vel = None
        self.acc = None
        self.omega = None
        self.yaw = None


class Polynomial4D:
    """One x/y/z/yaw piece with its duration (reference :49-85)."""

    def __init__(self, duration, px, py, pz, pyaw):
        self.duration = duration
        self.px = Polynomial(px)
        self.py = Polynomial(py)
        self.pz = Polynomial(pz)
        self.pyaw = Polynomial(pyaw)

    def derivative(self) -> "Polynomial4D":
        return Polynomial4D(self.duration, self.px.derivative().p, self.py.derivative().p,
                            self.pz.derivative().p, self.pyaw.derivative().p)

    def _xyz(self, t):
        return np.array([self.px.eval(t), self.py.eval(t), self.pz.eval(t)])

    def eval(self, t) -> TrajectoryOutput:
        """Differential-flatness outputs at local time t (reference :64-85)."""
        out = TrajectoryOutput()
        out.pos = self._xyz(t)
        out.yaw = self.pyaw.eval(t)
        d1 = self.derivative()
        out.vel = d1._xyz(t)
        yaw_rate = d1.pyaw.eval(t)
        d2 = d1.derivative()
        out.acc = d2._xyz(t)
        jerk = d2.derivative()._xyz(t)

        thrust = out.acc + np.array([0, 0, 9.81])
        thrust_norm = np.linalg.norm(thrust)
        zb = normalize(thrust)
        heading = np.array([np.cos(out.yaw), np.sin(out.yaw), 0])
        yb = normalize(np.cross(zb, heading))
        xb = np.cross(yb, zb)
        h_w = (jerk - np.dot(jerk, zb) * zb) / thrust_norm
        out.omega = np.array([-np.dot(h_w, yb), np.dot(h_w, xb), zb[2] * yaw_rate])
        return out


class Trajectory:
    """Rows [T | x8 | y8 | z8 | yaw8]; piece lookup with '<=' (reference :88-127)."""

    def __init__(self):
        self.polynomials = None
        self.duration = None

    def n_pieces(self):
        return len(self.polynomials)

    def loadcsv(self, filename, skiprows: int = 1):
        """The reference always skips one row (:113-114) -- correct for the
        crazyswarm traj.csv (header), but it drops the first piece of the
        header-less Pol_matrix_*.csv; pass skiprows=0 for those."""
        data = np.loadtxt(filename, delimiter=",", skiprows=skiprows, usecols=range(33))
        data = np.atleast_2d(data)
        self.load_matrix(data)

    def load_matrix(self, data):
        self.polynomials = [Polynomial4D(r[0], r[1:9], r[9:17], r[17:25], r[25:33]) for r in data]
        self.duration = np.sum(np.asarray(data)[:, 0])

    def eval(self, t) -> TrajectoryOutput:
        assert t >= 0
        assert t <= self.duration
        start = 0.0
        for piece in self.polynomials:
            if t <= start + piece.duration:
                return piece.eval(t - start)
            start = start + piece.duration
        return None


class PiecewisePolynomial:
    """Polynomials glued at cumulative durations; strict '<' lookup, the last
    piece extrapolates past the end (reference :147-169)."""

    def __init__(self, pols: list, time_durations: list):
        self.pols = pols
        self.nOfPols = len(pols)
        self.time_durations = time_durations

    def eval(self, t):
        assert t >= 0
        start = 0
        for k in range(self.nOfPols):
            if t < start + self.time_durations[k]:
                return self.pols[k].eval(t - start)
            start = start + self.time_durations[k]
        return self.pols[-1].eval(t - sum(self.time_durations[:-1]))


class Waypoint:
    WP_TYPE_X = 0
    WP_TYPE_Y = 1
    WP_TYPE_Z = 2
    WP_TYPE_YAW = 3

    def __init__(self, x, y, z, yaw):
        self.x = x
        self.y = y
        self.z = z
        self.yaw = yaw

    def getType(self, type):
        if type == Waypoint.WP_TYPE_X:
            return self.x
        if type == Waypoint.WP_TYPE_Y:
            return self.y
        if type == Waypoint.WP_TYPE_Z:
            return self.z
        if type == Waypoint.WP_TYPE_YAW:
            return self.yaw
        print("Sorry, invalid type")
        return None

    def as_tuple(self):
        return (self.x, self.y, self.z, self.yaw)


class Point_time:
    """Waypoint + the absolute time it must be reached (reference :198-202)."""

    def __init__(self, wp: Waypoint, t: float):
        self.wp = wp
        self.t = t


class Point_time1D:
    def __init__(self, wp: float, t: float):
        self.wp = wp
        self.t = t
