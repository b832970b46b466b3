#!/usr/bin/env python3
"""BASELINE.json configs[2] / configs[3] across the GPUs of one node, one process per GPU:

    python tools/formation_pipeline.py [--config 2|3] [--reps 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/formation_pipeline.py --config 3 [--backend gloo]

The pipeline itself is bench.py's (run_formation_config): the pinned formation-like swarm of
tests/golden/formation_golden.npz (synthetic.formation_config, 512 rigid bodies x 8 offsets through the
a8 kernel), solve -> sample -> all-gather (RCCL over xGMI with --backend nccl; the one real exchange
step of the path) -> pairwise pass -> (config 3) sweep against the reference's
env-scene-hole.stl + env-scene-ltu-experiment.stl.  Rank 0 prints one JSON object with per-stage times
(max over ranks), roofline fractions and the hit counts next to the fixture's.

--backend gloo rehearses the same code on a box with fewer GPUs than ranks (ranks share
device local_rank % device_count; the collective then runs on host copies).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=(2, 3))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from drone_path_planning_python_amd import Context

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_pg = world > 1
    if use_pg:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
    ctx = Context(device_id=dev_index, order=7, max_segments=64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    env = dict(torch=torch, dist=dist, ctx7=ctx, ctx9=None, device=device, rank=rank, world=world, use_pg=use_pg,
               red_dev=device if args.backend == "nccl" else "cpu",
               coll=dist if args.backend == "nccl" else bench.HostGather(dist))
    rep = bench.run_formation_config(args.config, env, args.reps, 5)
    if rank == 0:
        assert rep["pairwise_hits"] == rep["pairwise_hits_fixture"], rep
        assert args.config == 2 or rep["mesh_hits"] == rep["mesh_hits_fixture"], rep
        print(json.dumps(rep), flush=True)
    ctx.use_own_stream()
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
