"""`trajectory_visualising` package of the reference (src/trajectory_visualising/__init__.py:1-2):
the trajectory evaluator and the Path sampler the visual nodes use."""
from ..optimizations.uav_trajectory import Trajectory, TrajectoryOutput  # noqa: F401
from .visualization import get_nav_path_msg, get_nav_path_msgs, quaternion_from_yaw, trajectory_arrays  # noqa: F401
