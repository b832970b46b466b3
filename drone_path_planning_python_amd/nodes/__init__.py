"""ROS-node level API of the reference's scripts/ kept as importable modules."""
from . import msgs  # noqa: F401
