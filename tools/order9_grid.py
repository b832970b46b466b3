#!/usr/bin/env python3
"""The throughput solve (order 9, or PROBE_ORDER=7), launch time over the persistent grid ("solve_grid_waves").
   python tools/order9_grid.py <drones> <segments> <waves> [<waves> ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

N, M = int(sys.argv[1]), int(sys.argv[2])
order = int(os.environ.get("PROBE_ORDER", "9"))
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
for waves in [int(x) for x in sys.argv[3:]]:
    ctx.set_option("solve_grid_waves", waves)
    for _ in range(30):
        ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"order {order} {N} x {M}: grid {waves or 'default'}: {us:.1f} us = {bytes_ / us / 1e3 / 8000:.3f} of HBM peak", flush=True)
ctx.close()
