#!/usr/bin/env python3
"""Streaming kernels next to their HBM roofline: sampler (K4), float32 pack, formation transform,
flatness evaluator, snap cost.  Device-resident buffers, 40 warm-up + 30 timed launches each."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_path_planning_python_amd import Context  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda:0")
N, M = 1 << 18, 10
wp, t = swarm(2, 4096, M)
wp = np.tile(wp, (N // 4096, 1, 1)); t = np.tile(t, (N // 4096, 1))
out = []


def timed(fn):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e-3


with Context(0, 7, 64) as ctx:
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dwp, dt = torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev)
    coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device=dev)
    dur = torch.empty((N, M), dtype=torch.float64, device=dev)
    st = torch.empty((N,), dtype=torch.int32, device=dev)
    ctx.solve_batch_device(N, M, dwp, dt, False, coef, dur, st)
    torch.cuda.synchronize()
    cbytes, dbytes = coef.numel() * 8, dur.numel() * 8

    S = 91
    pos = torch.empty((N, S, 3), dtype=torch.float64, device=dev)
    s = timed(lambda: ctx.sample_device(N, M, coef, dur, 0.1, S, 3, pos))
    b = cbytes + dbytes + pos.numel() * 8
    out.append({"kernel": "sample (91 samples x 3 axes)", "ms": s * 1e3, "GBps": b / s / 1e9, "frac": b / s / 1e9 / PEAK})

    mat = torch.empty((N, M, 33), dtype=torch.float32, device=dev)
    s = timed(lambda: ctx.pack_pol_matrix_device(N, M, coef, dur, mat))
    b = cbytes + dbytes + mat.numel() * 4
    out.append({"kernel": "pack float32 [T|x|y|z|yaw]", "ms": s * 1e3, "GBps": b / s / 1e9, "frac": b / s / 1e9 / PEAK})

    cost = torch.empty((N, 4), dtype=torch.float64, device=dev)
    s = timed(lambda: ctx.snap_cost_device(N, M, coef, dur, cost))
    b = cbytes + dbytes + cost.numel() * 8
    out.append({"kernel": "snap cost", "ms": s * 1e3, "GBps": b / s / 1e9, "frac": b / s / 1e9 / PEAK})

    ns = 16
    ts = torch.from_numpy(np.sort(np.random.default_rng(0).uniform(0, 5.0, size=ns))).to(dev)
    flat = torch.empty((N, ns, 13), dtype=torch.float64, device=dev)
    s = timed(lambda: ctx.eval_flat_device(N, M, coef, dur, ns, ts, flat))
    b = cbytes + dbytes + ts.numel() * 8 + flat.numel() * 8
    out.append({"kernel": "flatness evaluator (16 instants, 13 outputs)", "ms": s * 1e3, "GBps": b / s / 1e9, "frac": b / s / 1e9 / PEAK})

    P, K = 1 << 16, 64
    rng = np.random.default_rng(1)
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = torch.from_numpy(np.concatenate([rng.uniform(-5, 5, (P, 3)), q], axis=1)).to(dev)
    offs = torch.from_numpy(rng.uniform(-2, 2, (K, 3))).to(dev)
    fo = torch.empty((K, P, 7), dtype=torch.float64, device=dev)
    s = timed(lambda: ctx.formation_transform_device(P, K, pose, offs, fo))
    b = pose.numel() * 8 + fo.numel() * 8
    out.append({"kernel": "formation transform (64 offsets x 65536 poses)", "ms": s * 1e3, "GBps": b / s / 1e9, "frac": b / s / 1e9 / PEAK})

for r in out:
    print("%-50s %.3f ms  %7.0f GB/s  %.0f %% of HBM peak" % (r["kernel"], r["ms"], r["GBps"], 100 * r["frac"]))
print(json.dumps(out))
