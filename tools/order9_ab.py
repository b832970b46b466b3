#!/usr/bin/env python3
"""Order-9 large-batch solve, A/B over the kernel choice ("no_twin": 0 two-sided column-split kernel,
1 solve_kernel_reg<5,10>) and the persistent grid.
   python tools/order9_ab.py [drones=65536] [segments=10]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
order = 9
dev = torch.device("cuda", 0)
wp, t = swarm(5, min(N, 65536), M)
reps = (N + wp.shape[0] - 1) // wp.shape[0]
wp, t = np.tile(wp, (reps, 1, 1))[:N], np.tile(t, (reps, 1))[:N]
twp, tt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
coef = torch.empty((N, M, 4, order + 1), dtype=torch.float64, device=dev)
dur = torch.empty((N, M), dtype=torch.float64, device=dev)
st = torch.empty((N,), dtype=torch.int32, device=dev)
ctx = Context(0, order, 64)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_option("no_twist", 1)
bytes_ = N * (8 * 5 * (M + 1) + 8 * M * (1 + 4 * (order + 1)))
ref = None
for mode in (0, 1, 0):
    ctx.set_option("no_twin", mode)
    for waves in (0, 2048, 4096):
        ctx.set_option("solve_grid_waves", waves)
        for _ in range(40):
            ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        c = coef.cpu().numpy()
        if ref is None:
            ref = c.copy()
        err = float((np.abs(c - ref).max(axis=(1, 3)) / np.abs(ref).max(axis=(1, 3))).max())
        print(f"no_twin={mode} {N} x {M}: grid {waves or 'default'}: {us:.1f} us = {bytes_ / us / 1e3 / 8000:.3f} of HBM peak; "
              f"vs solve_kernel_reg {err:.2e}; status ok {bool((st == 0).all())}", flush=True)
ctx.close()
