#!/usr/bin/env python3
"""Phase timeline of the shared-grid GEMM kernel (K2) on the headline shape; needs a library built
with -DMSNAP_TOOLS_TIMELINE (MSNAP_LIB_PATH)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context, _lib  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

N, M = 256, 10
wp, _ = swarm(2, N, M)
tgrid = np.arange(M + 1) * 10.0 / (M + 1)
dev = torch.device("cuda:0")
dwp = torch.from_numpy(wp).to(dev)
coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device=dev)
dur = torch.empty((N, M), dtype=torch.float64, device=dev)
st = torch.empty((N,), dtype=torch.int32, device=dev)
lib = _lib.load()
lib.msnap_debug_read_grid_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
with Context(0, 7, 64) as ctx:
    ctx.prepare_grid(tgrid)
    for _ in range(20):
        ctx.solve_grid_device(N, M, dwp, coef, dur, st)
    ctx.sync()
    nw = N // 4
    buf = np.zeros((nw, 8), dtype=np.uint64)
    assert lib.msnap_debug_read_grid_timeline(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf[:, :5].astype(np.int64)
print("waves", nw, "start skew %.2f us, span %.2f us" % ((t[:, 0].max() - t[:, 0].min()) / 100.0, (t[:, 4].max() - t[:, 0].min()) / 100.0))
for k, nm in enumerate(["prologue + operand loads (B fragments, first A tile)", "MFMAs + exchange + coefficient stores", "durations + status", "loop exit (store drain)"]):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print("%-55s %.2f us (min %.2f max %.2f)" % (nm, np.median(d), d.min(), d.max()))
