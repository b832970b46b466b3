#!/usr/bin/env python3
"""A handful of launches of the pairwise pass on the configs[2] shape (profiling target)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 91
wpc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
pos = torch.from_numpy(np.random.default_rng(5).uniform(-50, 50, size=(N, S, 3))).to(dev)
md = torch.empty((N,), dtype=torch.float64, device=dev)
partner = torch.empty((N,), dtype=torch.int32, device=dev)
hit = torch.empty((N,), dtype=torch.int32, device=dev)
ctx = Context(0, 7, 16)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_option("collide_waves_per_cu", wpc)
for _ in range(6):
    ctx.formation_collide_device(N, 0, N, S, pos, pos, 0.15, md, partner, hit)
torch.cuda.synchronize()
ctx.close()
