#!/usr/bin/env python3
"""Times the pairwise formation pass (K5) on the configs[2] shape for the tile shapes and wave counts
the context options expose; prints one line per setting.  GPU box only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402

VALU_F64_OPS = 256 * 4 * 16 * 2.4e9


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 91
    R = int(sys.argv[3]) if len(sys.argv) > 3 else N          # rows of a shard (default: all)
    RO = int(sys.argv[4]) if len(sys.argv) > 4 else 0         # its first global row
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    if os.environ.get("SPREAD"):
        # a sparse swarm: start points over a square of SPREAD metres, each drone wanders a couple of metres
        ext = float(os.environ["SPREAD"])
        start = rng.uniform(-ext / 2, ext / 2, size=(N, 1, 3)) * np.array([1.0, 1.0, 0.05])
        pos = torch.from_numpy(start + np.cumsum(rng.normal(0, 0.05, size=(N, S, 3)), axis=1)).to(dev)
    else:
        # the dense worst case: every sample anywhere in a 100 m cube (nothing can be culled)
        pos = torch.from_numpy(rng.uniform(-50, 50, size=(N, S, 3))).to(dev)
    md = torch.empty((R,), dtype=torch.float64, device=dev)
    partner = torch.empty((R,), dtype=torch.int32, device=dev)
    hit = torch.empty((R,), dtype=torch.int32, device=dev)
    rows = pos[RO:RO + R].contiguous()
    ctx = Context(0, 7, 16)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ref = None
    for variant in (0,):
        for wpc in [int(x) for x in os.environ.get("WPC", "64,0,64,0").split(",")]:
            ctx.set_option("collide_waves_per_cu", wpc)
            for _ in range(25):
                ctx.formation_collide_device(R, RO, N, S, rows, pos, 0.15, md, partner, hit)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                ctx.formation_collide_device(R, RO, N, S, rows, pos, 0.15, md, partner, hit)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            out = (md.cpu().numpy().copy(), partner.cpu().numpy().copy())
            if ref is None:
                ref = out
            same = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
            # a shard evaluates its rows against the other columns, its own block once per pair
            alg = (R * (N - R) + R * (R - 1) / 2) * S * 7      # 3 differences, 1 product, 2 FMAs, 1 minimum
            print(json.dumps({"N": N, "S": S, "rows": R, "waves_per_cu": wpc, "us": round(us, 1),
                              "frac_of_f64_issue_peak_on_unordered_pairs": round(alg / (us * 1e-6) / VALU_F64_OPS, 3),
                              "broad_phase": bool(ctx.get_option("collide_last_cull")),
                              "shares": ctx.get_option("collide_last_shares"),
                              "survivors": ctx.get_option("collide_last_survivors"),
                              "group_pairs": ctx.get_option("collide_last_group_pairs"),
                              "same_result": bool(same)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
