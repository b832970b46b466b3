#!/usr/bin/env python3
"""BASELINE.json configs[2]/[3] across the GPUs of one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/formation_pipeline.py [--drones 4096] [--segments 10] [--mesh]

Every rank solves its contiguous shard of the swarm (no communication), samples it,
ALL-GATHERS the sampled positions (RCCL over xGMI with --backend nccl; the one real
exchange step of the path), checks its own rows against all columns and,
with --mesh, sweeps its shard against a replicated STL-like mesh.  Rank 0 prints one JSON
object with per-stage times (max over ranks) and the global hit counts.

--backend gloo rehearses the same code on a box with fewer GPUs than ranks (ranks share
device local_rank % device_count; the collective then runs on host copies).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drones", type=int, default=4096)
    ap.add_argument("--segments", type=int, default=10)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--radius", type=float, default=0.15)
    ap.add_argument("--mesh", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from drone_path_planning_python_amd import Context, stl, swarm
    from drone_path_planning_python_amd.synthetic import swarm as synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    N, M = args.drones, args.segments
    wp, t = synth(3, N, M, shared_times=True)           # every rank builds the same swarm, keeps its shard
    wp[..., :3] *= 3.0
    lo, hi = swarm.shard_bounds(N, world, rank)
    S = swarm.default_sample_count(float(t[-1]), args.dt)
    ctx = Context(device_id=dev_index, order=7, max_segments=max(M, 16))
    comp = swarm.DeviceCompute(ctx, torch)
    twp = torch.from_numpy(np.ascontiguousarray(wp[lo:hi])).to(device)
    tt = torch.from_numpy(t).to(device)
    tris = None
    if args.mesh:
        wall = stl.box_mesh((-2, 3.9, 0), (2, 4.1, 1.6))
        tris = torch.from_numpy(np.concatenate([wall + np.array([dx, 0, 0]) for dx in (-6, -3, 0, 3, 6)])[:56]).to(device)

    class HostGather:
        """gloo rehearsal only: the collective on host copies of the position blocks."""
        @staticmethod
        def all_gather_into_tensor(out, inp):
            o, i = out.cpu(), inp.cpu()
            dist.all_gather_into_tensor(o, i)
            out.copy_(o)

    coll = dist if args.backend == "nccl" else HostGather

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    stages = {"solve": 0.0, "sample+allgather+collide": 0.0, "mesh": 0.0}
    res = None
    for rep in range(args.reps + 1):
        sync()
        t0 = time.perf_counter()
        coef, dur, status = comp.solve(twp, tt)
        sync()
        t1 = time.perf_counter()
        res = swarm.formation_pass(comp, coef, dur, N, world, rank, args.dt, S, args.radius, dist=coll, torch=torch)
        sync()
        t2 = time.perf_counter()
        mesh_hits = 0
        if tris is not None:
            lo_pos = res.positions_all[lo:hi] if world > 1 else res.positions_all
            md, mh = comp.mesh(lo_pos.contiguous(), tris, args.radius)
            mesh_hits = int(mh.sum().item())
        sync()
        t3 = time.perf_counter()
        if rep > 0:                      # rep 0 is the warm-up
            stages["solve"] += (t1 - t0) / args.reps
            stages["sample+allgather+collide"] += (t2 - t1) / args.reps
            stages["mesh"] += (t3 - t2) / args.reps
    assert int(status.abs().sum().item()) == 0
    counts = torch.tensor([int(res.hit.sum().item()), mesh_hits], dtype=torch.int64,
                          device=device if args.backend == "nccl" else "cpu")
    times = torch.tensor(list(stages.values()), dtype=torch.float64, device=counts.device)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            "workload": f"{N} drones x {M} segments, {S} samples, world {world} ({args.backend})",
            "shard": [lo, hi], "formation_hits": int(counts[0]), "mesh_hits": int(counts[1]),
            "ms": {k: float(v) * 1e3 for k, v in zip(stages, times.tolist())},
        }), flush=True)
    ctx.use_own_stream()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
