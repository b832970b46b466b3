#!/usr/bin/env python3
"""Phase timeline of the small-batch (twisted) solve kernel on the headline shape.
Needs a library built with -DMSNAP_TOOLS_TIMELINE (MSNAP_LIB_PATH points at it):
each wave records s_memrealtime (100 MHz) and s_memtime at 5 points."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context, _lib  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

N, M = 256, 10
wp, t = swarm(2, N, M)
dev = torch.device("cuda:0")
dwp, dt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device=dev)
dur = torch.empty((N, M), dtype=torch.float64, device=dev)
st = torch.empty((N,), dtype=torch.int32, device=dev)
lib = _lib.load()
lib.msnap_debug_read_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
with Context(0, 7, 64) as ctx:
    for _ in range(20):
        ctx.solve_batch_device(N, M, dwp, dt, False, coef, dur, st)
    ctx.sync()
    half = bool(os.environ.get("MSNAP_TL_HALF_GRID"))
    nw = (N + 7) // 8
    if half:
        nw = (nw + 1) // 2
    buf = np.zeros((nw, 32), dtype=np.uint64)
    assert lib.msnap_debug_read_timeline(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
if half:
    r2 = buf[:, 16:][:, 0::2][:, :5].astype(np.int64)
    print("second tile of each wave, per phase (us):", np.round(np.median(np.diff(r2, axis=1), axis=0) / 100.0, 2),
          " total %.2f" % (np.median(r2[:, 4] - r2[:, 0]) / 100.0))
buf = buf[:, :16]
real = buf[:, 0::2][:, :5].astype(np.int64)
shad = buf[:, 1::2][:, :5].astype(np.int64)
names = ["stage inputs (load + LDS + barrier)", "forward sweep", "status + merge", "backward + recovery + stores"]
print("waves", nw, " start skew between waves: %.2f us" % ((real[:, 0].max() - real[:, 0].min()) / 100.0))
print("kernel span first start -> last end: %.2f us" % ((real[:, 4].max() - real[:, 0].min()) / 100.0))
for k, nm in enumerate(names):
    d_real = (real[:, k + 1] - real[:, k]) / 100.0
    d_sh = shad[:, k + 1] - shad[:, k]
    print("%-40s %.2f us (median; min %.2f max %.2f)   %d shader-clock ticks" %
          (nm, np.median(d_real), d_real.min(), d_real.max(), int(np.median(d_sh))))
tot = (real[:, 4] - real[:, 0]) / 100.0
print("per-wave total %.2f us median" % np.median(tot))
