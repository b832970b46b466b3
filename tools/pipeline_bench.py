#!/usr/bin/env python3
"""Times the BASELINE.json configs[2]/[3] pipelines on one GPU, device-resident:
config 3: 4096 drones x 10 segments  solve -> sample (dt = 0.1 s) -> pairwise formation pass
config 4: 4096 drones x 20 segments  solve -> sample -> STL-mesh sweep (56 triangles)
Prints one JSON object (kernel-level HIP-event times, ms)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drone_path_planning_python_amd import Context, stl  # noqa: E402
from drone_path_planning_python_amd.synthetic import swarm  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(cfg, N, M, mesh_tris=None):
    dev = torch.device("cuda", 0)
    ctx = Context(0, 7, 64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    wp, t = swarm(cfg, N, M, shared_times=True)
    wp[..., :3] *= 3.0
    t = t.copy()
    tw = torch.from_numpy(wp).to(dev)
    tt = torch.from_numpy(t).to(dev)
    coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device=dev)
    dur = torch.empty((N, M), dtype=torch.float64, device=dev)
    st = torch.empty((N,), dtype=torch.int32, device=dev)
    S = len(np.arange(0.0, float(t[-1]), 0.1))
    pos = torch.empty((N, S, 3), dtype=torch.float64, device=dev)
    md = torch.empty((N,), dtype=torch.float64, device=dev)
    partner = torch.empty((N,), dtype=torch.int32, device=dev)
    hit = torch.empty((N,), dtype=torch.int32, device=dev)
    out = {"config": cfg, "drones": N, "segments": M, "samples": S}
    out["solve_ms"] = timeit(lambda: ctx.solve_batch_device(N, M, tw, tt, True, coef, dur, st))
    out["sample_ms"] = timeit(lambda: ctx.sample_device(N, M, coef, dur, 0.1, S, 3, pos))
    if mesh_tris is None:
        out["collide_ms"] = timeit(lambda: ctx.formation_collide_device(N, 0, N, S, pos, pos, 0.15, md, partner, hit), 5)
        out["pair_samples_per_s"] = N * (N - 1) * S / (out["collide_ms"] * 1e-3)
    else:
        tr = torch.from_numpy(mesh_tris).to(dev)
        out["mesh_tris"] = int(mesh_tris.shape[0])
        out["mesh_ms"] = timeit(lambda: ctx.mesh_sweep_device(N, S, pos, mesh_tris.shape[0], tr, 0.15, md, hit), 5)
        out["point_tri_tests_per_s"] = N * S * mesh_tris.shape[0] / (out["mesh_ms"] * 1e-3)
    out["hits"] = int(hit.sum().item())
    ctx.close()
    return out


if __name__ == "__main__":
    wall = stl.box_mesh((-2, 3.9, 0), (2, 4.1, 1.6))
    mesh = np.concatenate([wall + np.array([dx, 0, 0]) for dx in (-6, -3, 0, 3, 6)])[:56]
    print(json.dumps([run(3, 4096, 10), run(4, 4096, 20, mesh)]))
