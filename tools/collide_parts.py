#!/usr/bin/env python3
"""The pairwise pass in parts (every unordered pair on exactly one rank), every rank of G emulated on one GPU:
per part the time of msnap_formation_collide_part (transpose + span + merge of that rank), then the fold, and
whether the folded result equals the single-launch pass bit for bit.   python tools/collide_parts.py [N] [S] [G ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context  # noqa: E402

VALU_F64_OPS = 256 * 4 * 16 * 2.4e9


def timed(fn, warm=25, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 91
    Gs = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    pos = torch.from_numpy(rng.uniform(-50, 50, size=(N, S, 3))).to(dev)
    ctx = Context(0, 7, 16)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    md = torch.empty((N,), dtype=torch.float64, device=dev)
    partner = torch.empty((N,), dtype=torch.int32, device=dev)
    hit = torch.empty((N,), dtype=torch.int32, device=dev)
    whole = timed(lambda: ctx.formation_collide_device(N, 0, N, S, pos, pos, 0.15, md, partner, hit))
    ref = (md.cpu().numpy().copy(), partner.cpu().numpy().copy())
    alg = N * (N - 1) / 2 * S * 7
    print(json.dumps({"N": N, "S": S, "parts": 0, "what": "msnap_formation_collide, one launch over the whole swarm",
                      "us": round(whole, 1), "frac_of_f64_issue_peak_on_unordered_pairs":
                      round(alg / (whole * 1e-6) / VALU_F64_OPS, 3)}), flush=True)
    stride = ctx.formation_part_bytes(N)
    sp = int(os.environ.get("COLLIDE_SPARTS", "0"))       # tuning: force the waves per share of the part launches
    if sp:
        ctx.set_option("collide_sample_parts", sp)
    for G in Gs:
        blocks = torch.empty((G, stride), dtype=torch.uint8, device=dev)
        us = [timed(lambda p=p: ctx.formation_collide_part_device(N, S, pos, p, G, blocks[p])) for p in range(G)]
        n_own = N // G
        fmd = torch.empty((n_own,), dtype=torch.float64, device=dev)
        fp = torch.empty((n_own,), dtype=torch.int32, device=dev)
        fh = torch.empty((n_own,), dtype=torch.int32, device=dev)
        fus = timed(lambda: ctx.formation_collide_finish_device(N, G, blocks, 0, n_own, 0.15, fmd, fp, fh))
        same = True
        for r in range(G):
            lo = r * n_own
            ctx.formation_collide_finish_device(N, G, blocks, lo, n_own, 0.15, fmd, fp, fh)
            same &= np.array_equal(fmd.cpu().numpy(), ref[0][lo:lo + n_own]) and np.array_equal(fp.cpu().numpy(), ref[1][lo:lo + n_own])
        print(json.dumps({"N": N, "S": S, "parts": G, "sample_parts": sp or "auto", "part_us": [round(u, 1) for u in us], "max_part_us": round(max(us), 1),
                          "finish_us": round(fus, 1), "exchange_bytes_per_rank": stride,
                          "frac_of_f64_issue_peak_per_rank": round(alg / G / (max(us) * 1e-6) / VALU_F64_OPS, 3),
                          "folded_equals_one_launch": bool(same)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
