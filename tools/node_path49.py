#!/usr/bin/env python3
"""The reference's live shape through the node API: a 50-pose rigid-body path (OMPL interpolate(50),
src/RigidBodyPlanners/RB_planning_sep_coll_check.py:164) -> transform() -> two drone paths
(scripts/drones_traj_generator.py:56-89) -> paths_to_pols() on the uniform 10/50 grid
(scripts/drones_pols_generator.py:44-46) = 49 segments -> Pol_matrix files.  Profiling target: the solve
must be the streaming fp64 MFMA GEMM (grid_gemm_stream_kernel<8, 13, 1>)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.nodes import drones_pols_generator as dpg  # noqa: E402
from drone_path_planning_python_amd.nodes import drones_traj_generator as dtg  # noqa: E402
from drone_path_planning_python_amd.nodes import msgs  # noqa: E402

n = 50
s = np.linspace(0.0, 1.0, n)
pos = np.stack([0.5 + 0.1 * np.sin(6 * s), 3.0 + 2.0 * s, 1.0 + 0.3 * np.sin(3 * s + 0.2)], axis=1)   # start (0,3,1) -> goal (0,5,1)
yaw = 0.6 * s
quat = np.stack([np.zeros(n), np.zeros(n), np.sin(0.5 * yaw), np.cos(0.5 * yaw)], axis=1)
path = msgs.path_from_arrays(pos, quat)
with Context(0, 7, 64) as ctx:
    p1, p2 = dtg.transform_formation(path, dtg.drone_positions, ctx=ctx)
    for rep in range(5):
        matrix, coef, dur = dpg.paths_to_pols([p1, p2], ctx)
    assert matrix.shape == (2, 49, 33) and coef.shape == (2, 49, 4, 8)
    k1, _, st = ctx.solve_batch(*dpg.paths_to_waypoints([p1, p2]))
    assert (st == 0).all()
    err = float((np.abs(coef - k1).max(axis=(1, 3)) / np.abs(k1).max(axis=(1, 3))).max())
    with tempfile.TemporaryDirectory() as d:
        f = dpg.save_pol_matrix(matrix[0], 1, d)
        back = np.loadtxt(f, delimiter=",")
    assert back.shape == (49, 33)
    print(f"paths_to_pols: 2 drones x 49 segments through the shared-grid GEMM, vs the per-drone solve {err:.2e}")
    assert err < 1e-7
