#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point msnap_solve_batch (H2D + kernel + D2H,
pageable NumPy buffers) -- quoted in DESIGN.md 7, never bench.py's `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context, pinned_empty  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

out = []
with Context(0, 7, 64) as ctx:
    for n in (1, 256, 4096, 65536, 1 << 20):
        wp, t = swarm(2, n, 10)
        ctx.solve_batch(wp, t)
        reps = 200 if n <= 4096 else (20 if n <= 65536 else 3)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.solve_batch(wp, t)
        dt = (time.perf_counter() - t0) / reps
        row = {"drones": n, "ms_per_call": dt * 1e3, "traj_per_s": n / dt, "GBps_moved": n * 3080 / dt / 1e9}
        # page-locked inputs and caller-owned page-locked outputs (pinned_empty + out=)
        pwp, pt = pinned_empty(wp.shape), pinned_empty(t.shape)
        pwp[...] = wp
        pt[...] = t
        res = (pinned_empty((n, 10, 4, 8)), pinned_empty((n, 10)), pinned_empty((n,), np.int32))
        ctx.solve_batch(pwp, pt, out=res)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.solve_batch(pwp, pt, out=res)
        dt = (time.perf_counter() - t0) / reps
        row.update({"pinned_ms_per_call": dt * 1e3, "pinned_traj_per_s": n / dt,
                    "pinned_GBps_moved": n * 3080 / dt / 1e9})
        out.append(row)
print(json.dumps(out))
