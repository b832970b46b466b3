#!/usr/bin/env python3
"""Shared-grid solve + sampler: the fused launch (msnap_solve_grid_sample_device) against the two kernels.

    python tools/grid_sample_probe.py [N] [reps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from drone_path_planning_python_amd import Context, swarm as sw
    from drone_path_planning_python_amd.synthetic import swarm
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    for order, M, S in ((7, 10, 91), (7, 20, 96), (7, 5, 50), (7, 15, 91), (9, 10, 91)):
        ctx = Context(order=order)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        comp = sw.DeviceCompute(ctx, torch)
        wp, _ = swarm(1, N, M)
        ctx.prepare_grid(np.linspace(0.0, 10.0, M + 1))
        d_wp = torch.from_numpy(wp).cuda()
        # buffers allocated once and the C-ABI entries called directly: with five torch.empty per call the loop is
        # bound by the host (15 us per call), not by the kernels
        coef = torch.empty((N, M, 4, ctx.ncoef), dtype=torch.float64, device="cuda")
        dur = torch.empty((N, M), dtype=torch.float64, device="cuda")
        status = torch.empty((N,), dtype=torch.int32, device="cuda")
        pos = torch.empty((N, S, 3), dtype=torch.float64, device="cuda")
        pos_t = torch.empty((ctx.collide_rows_t_doubles(N, S),), dtype=torch.float64, device="cuda")

        def two():
            ctx.solve_grid_device(N, M, d_wp, coef, dur, status)
            ctx.sample_collide_device(N, M, coef, dur, 0.1, S, pos, pos_t)

        def fused():
            ctx.solve_grid_sample_device(N, M, d_wp, 0.1, S, coef, dur, status, pos, pos_t)

        out = {"two launches": [], "fused": []}
        for trial in range(3):      # interleaved, best of three: the first loop of a process runs on ramping clocks
            for name, fn in (("two launches", two), ("fused", fused)):
                for _ in range(20):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out[name].append(e0.elapsed_time(e1) / reps * 1e3)
        out = {k: min(v) for k, v in out.items()}
        print(f"order {order} N {N} M {M} S {S}: " + ", ".join(f"{k} {v:.2f} us" for k, v in out.items()), flush=True)
        ctx.use_own_stream()
        ctx.close()


if __name__ == "__main__":
    main()
