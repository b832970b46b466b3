"""`optimizations` package of the reference (src/optimizations/__init__.py:1-2)."""
from .uav_trajectory import *  # noqa: F401,F403
from .uav_trajectory import (Point_time, Point_time1D, Polynomial, Polynomial4D, PiecewisePolynomial,  # noqa: F401
                             Trajectory, TrajectoryOutput, Waypoint, normalize)
from .calculatingTrajectories import calculate_trajectory1D, calculate_trajectory4D  # noqa: F401
