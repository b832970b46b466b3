#!/usr/bin/env python3
"""The solve over batch sizes (configs[4] sharded over 1..8 GPUs is 65536..8192 drones per rank), order 9 or -- with
PROBE_ORDER=7 -- order 7: the launcher's choice against the alternatives ("no_twist" keeps small batches off the
two-sided latency kernel, "no_twin" keeps larger ones on solve_kernel_reg), each under a hipGraph of 50 launches.
   [PROBE_ORDER=7] [PROBE_GRIDS="1024 2048 ..."] python tools/order_sizes.py [segments=10] [N ...]
PROBE_GRIDS: the launcher's choice again with the persistent grid forced to these wave counts (solve_grid_waves)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sizes = [int(x) for x in sys.argv[2:]] or [2048, 4096, 8192, 16384, 32768, 65536]
order = int(os.environ.get("PROBE_ORDER", "9"))
dev = torch.device("cuda", 0)
for N in sizes:
    wp, t = swarm(5, N, M)
    twp, tt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
    coef = torch.empty((N, M, 4, order + 1), dtype=torch.float64, device=dev)
    dur = torch.empty((N, M), dtype=torch.float64, device=dev)
    st = torch.empty((N,), dtype=torch.int32, device=dev)
    bytes_ = N * (8 * 5 * (M + 1) + 8 * M * (1 + 4 * (order + 1)))
    grids = [{"solve_grid_waves": int(g)} for g in os.environ.get("PROBE_GRIDS", "").split()]
    if os.environ.get("PROBE_TWIN_MAX"):      # the column-split kernel beyond the launcher's batch-size limit (order 7)
        grids.append({"twin_max_drones": int(os.environ["PROBE_TWIN_MAX"])})
    for opts in [{}, {"no_twist": 1}, {"no_twist": 1, "no_twin": 1}] + grids:
        ctx = Context(0, order, 64)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for k, v in opts.items():
            ctx.set_option(k, v)
        g = torch.cuda.CUDAGraph()
        for _ in range(5):
            ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(50):
                ctx.solve_batch_device(N, M, twp, tt, False, coef, dur, st)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tag = f" grid {opts['solve_grid_waves']}" if "solve_grid_waves" in opts else ""
        tag = " (twin_max_drones)" if "twin_max_drones" in opts else tag
        print(f"order {order} {N:6d} x {M}: {ctx.last_kernel() + tag:40s} {us:7.2f} us = {bytes_ / us / 1e3 / 8000:.3f} of HBM peak "
              f"({N / us:.1f} traj/us)", flush=True)
        ctx.use_own_stream()
        ctx.close()
