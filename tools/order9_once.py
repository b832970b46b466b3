#!/usr/bin/env python3
"""A few launches of the order-9 large-batch solve for profilers.
   python tools/order9_once.py [drones=65536] [segments=10] [no_twin=0] [launches=20]"""
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
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 20
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
ctx.set_option("no_twin", mode)
for _ in range(launches):
    ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
torch.cuda.synchronize()
ctx.close()
