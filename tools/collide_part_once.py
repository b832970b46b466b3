#!/usr/bin/env python3
"""A few launches of ONE part of the pairwise pass for profilers.
   python tools/collide_part_once.py [N=4096] [S=91] [parts=8] [part=3] [launches=20]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 91
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
p = int(sys.argv[4]) if len(sys.argv) > 4 else 3
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda", 0)
pos = torch.from_numpy(np.random.default_rng(5).uniform(-50, 50, size=(N, S, 3))).to(dev)
ctx = Context(0, 7, 16)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sp = int(os.environ.get("COLLIDE_SPARTS", "0"))
if sp:
    ctx.set_option("collide_sample_parts", sp)
out = torch.empty((ctx.formation_part_bytes(N),), dtype=torch.uint8, device=dev)
for _ in range(launches):
    ctx.formation_collide_part_device(N, S, pos, p, G, out)
torch.cuda.synchronize()
ctx.close()
