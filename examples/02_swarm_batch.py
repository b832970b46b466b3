#!/usr/bin/env python3
"""A whole swarm in one call: 4096 drones x 10 segments, per-drone time grids, then the
shared-grid fast path (one fp64 MFMA GEMM) for the node's uniform grid."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

with Context(device_id=0, order=7, max_segments=64) as ctx:
    wp, t = swarm(3, 4096, 10)                         # per-drone random segment lengths
    t0 = time.perf_counter()
    coef, dur, status = ctx.solve_batch(wp, t)
    print(f"per-drone grids: {wp.shape[0]} trajectories in {(time.perf_counter() - t0) * 1e3:.2f} ms "
          f"(host buffers, PCIe included), failures: {int((status != 0).sum())}")
    cost = ctx.snap_cost(coef, dur)
    print("mean snap cost per axis:", cost.mean(axis=0))

    wps, ts = swarm(3, 4096, 10, shared_times=True)    # the reference's uniform grid
    ctx.prepare_grid(ts)
    t0 = time.perf_counter()
    coef2, dur2, status2 = ctx.solve_grid(wps)
    print(f"shared grid (MFMA GEMM): {(time.perf_counter() - t0) * 1e3:.2f} ms")
    k1, _, _ = ctx.solve_batch(wps, ts)
    print("GEMM vs sweep, max |diff| / max |coef|:", float(np.abs(coef2 - k1).max() / np.abs(k1).max()))
    mat = ctx.pack_pol_matrix(coef2[:1], dur2[:1])[0]   # the float32 rows the node writes / publishes
    print("Pol_matrix row 0:", mat[0, :5], "...")
