"""MI355X-native minimum-snap trajectory generation (drop-in for the
`optimizations` path of mjmyt/drone_path_planning_python).

    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.optimizations import calculate_trajectory4D

Importing the package does not touch the GPU; creating a Context loads
csrc/libmsnap.so (raises if it is missing -- there is no CPU fallback).
"""
from .context import Context, default_context, pinned_empty, STATUS_TEXT  # noqa: F401
from ._lib import MsnapError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
