#!/usr/bin/env python3
"""Phase timeline of the fused shared-grid solve + sampler kernel (grid_sample_kernel) at 4096 drones x 10 segments;
needs a library built with -DMSNAP_TOOLS_TIMELINE (MSNAP_LIB_PATH, tools/build_timeline_lib.sh)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context, _lib, swarm as sw  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

N, M, S = 4096, 10, 91
wp, _ = swarm(2, N, M)
lib = _lib.load()
lib.msnap_debug_read_gs_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
with Context(0, 7, 64) as ctx:
    comp = sw.DeviceCompute(ctx, torch)
    ctx.prepare_grid(np.linspace(0.0, 10.0, M + 1))
    d_wp = torch.from_numpy(wp).cuda()
    for _ in range(20):
        comp.solve_grid_sample(d_wp, 0.1, S, n_cols=N)
    ctx.sync()
    assert ctx.last_kernel().startswith("msnap::grid_sample_kernel")
    nb = 512
    buf = np.zeros((1024, 8), dtype=np.uint64)
    assert lib.msnap_debug_read_gs_timeline(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf[:nb, :7].astype(np.int64)
t7 = buf[:nb, 7].astype(np.int64)
print("  (wave 0: operands + MFMAs + results written %.2f us after the barrier, then %.2f us at the next barrier)" % (
    np.median(t7 - t[:, 2]) / 100.0, np.median(t[:, 3] - t7) / 100.0))
print("workgroups", nb, "start skew %.2f us, span %.2f us" % ((t[:, 0].max() - t[:, 0].min()) / 100.0, (t[:, 6].max() - t[:, 0].min()) / 100.0))
names = ["running sums, waypoint + operator loads (first flight)", "waypoints -> LDS (transposed)", "MFMAs -> coefficients in LDS",
         "coefficient / duration / status stores issued", "Horner loops -> LDS image", "image -> positions, boxes, keys"]
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print("%-58s %.2f us (min %.2f max %.2f)" % (nm, np.median(d), d.min(), d.max()))
