#!/usr/bin/env python3
"""bench.py -- minimum-snap trajectories/sec on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (libmsnap's solve kernel, K1) over one
batch of synthetic input: BASELINE.json configs[1], 256 drones x 10 segments,
order 7, seeded random waypoints, per-drone random time grids (the general case:
nothing is shared or precomputed between drones).  Inputs and outputs live in
HBM (torch tensors are only the allocator); every rank solves its own 256-drone
shard (weak scaling, no data-path collective: the solve shards by drone).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     HBM roofline of the solve kernel at the headline workload
               (algorithmic bytes: SURVEY.md 8d, 3080 B/trajectory at M=10,p=7)
  cpu_baseline the C restatement of the reference algorithm (oracle/, "port")
               timed on this box's host cores on a bounded sample
  saturated    the same kernel on a batch large enough to fill the chip
               (2^20 drones), with its own roofline fraction -- the 256-drone
               headline is launch/latency-bound by construction
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "minimum-snap trajectories/sec (N drones × M segments, order 7) at 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s copy-achievable)
HBM_COPY_GBS = 6290.0


def algorithmic_bytes(n_drones: int, n_seg: int, order: int) -> int:
    """in = 8*5*(M+1) (x,y,z,yaw,t), out = 8*M*(1 + 4*(p+1))."""
    return n_drones * (8 * 5 * (n_seg + 1) + 8 * n_seg * (1 + 4 * (order + 1)))


def pmc_traffic(n_drones: int, n_seg: int, order: int):
    """(HBM bytes per launch, kernel name) of the solve at this workload, from the committed PMC
    passes (profiles/pmc_traffic.json, written by tools/make_profiles.sh on an MI355X: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    correction).  (None, None) when no matching profile exists."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tp) as f:
            rec = json.load(f)
        ent = rec.get(f"{n_drones}x{n_seg}o{order}") or {}
        return ent.get("hbm_bytes_per_launch"), ent.get("kernel")
    except Exception:
        return None, None


def solve_kernel_name(n_drones: int, n_seg: int, order: int, n_cu: int = 256) -> str:
    """Which K1 variant libmsnap launches (mirror of launch_solve_k in csrc/msnap_solve.hip)."""
    k = (order + 1) // 2
    if 2 <= n_seg <= (24 if k == 4 else 12) and n_drones <= n_cu * 32:
        return "msnap::solve_kernel_twist<%d, %d, %d>" % (k, (n_seg - 2) - (n_seg - 2) // 2, n_seg)
    if n_seg <= 20:
        return "msnap::solve_kernel_reg<%d, %d>" % (k, 10 if n_seg <= 10 else 20)
    return "msnap::solve_kernel<%d, false>" % k


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--drones", type=int, default=256, help="drones per GPU per step (configs[1] = 256)")
    ap.add_argument("--segments", type=int, default=10)
    ap.add_argument("--order", type=int, default=7, choices=(7, 9))
    ap.add_argument("--shared-times", action="store_true", help="reference's shared uniform time grid")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-saturated", action="store_true")
    ap.add_argument("--no-shared-grid", action="store_true")
    ap.add_argument("--saturated-drones", type=int, default=1 << 20)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-pg", action="store_true", help="initialise the process group even for one rank (rehearsal)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N > 1 on a 1-GPU box "
                         "(ranks then share device local_rank %% device_count)")
    return ap.parse_args()


class DeviceBatch:
    """Device-resident inputs/outputs of one batch + prebuilt ctypes arguments."""

    def __init__(self, torch, ctx, wp, t, n_seg, order, device):
        self.n = wp.shape[0]
        self.n_seg = n_seg
        self.shared = int(t.ndim == 1)
        self.wp = torch.from_numpy(np.ascontiguousarray(wp)).to(device)
        self.t = torch.from_numpy(np.ascontiguousarray(t)).to(device)
        self.coef = torch.empty((self.n, n_seg, 4, order + 1), dtype=torch.float64, device=device)
        self.dur = torch.empty((self.n, n_seg), dtype=torch.float64, device=device)
        self.status = torch.empty((self.n,), dtype=torch.int32, device=device)
        lib = ctx._lib
        self._fn = lib.msnap_solve_batch_device
        self._args = (ctx._h, self.n, n_seg, ctypes.c_void_p(self.wp.data_ptr()),
                      ctypes.c_void_p(self.t.data_ptr()), self.shared, ctypes.c_void_p(self.coef.data_ptr()),
                      ctypes.c_void_p(self.dur.data_ptr()), ctypes.c_void_p(self.status.data_ptr()))

    def step(self):
        rc = self._fn(*self._args)
        if rc != 0:
            raise RuntimeError(f"msnap_solve_batch_device failed: {rc}")


class GridBatch:
    """Same for the shared-grid path (msnap_solve_grid_device, K2)."""

    def __init__(self, torch, ctx, wp, n_seg, order, device):
        self.n = wp.shape[0]
        self.wp = torch.from_numpy(np.ascontiguousarray(wp)).to(device)
        self.coef = torch.empty((self.n, n_seg, 4, order + 1), dtype=torch.float64, device=device)
        self.dur = torch.empty((self.n, n_seg), dtype=torch.float64, device=device)
        self.status = torch.empty((self.n,), dtype=torch.int32, device=device)
        self._fn = ctx._lib.msnap_solve_grid_device
        self._args = (ctx._h, self.n, ctypes.c_void_p(self.wp.data_ptr()), ctypes.c_void_p(self.coef.data_ptr()),
                      ctypes.c_void_p(self.dur.data_ptr()), ctypes.c_void_p(self.status.data_ptr()))

    def step(self):
        rc = self._fn(*self._args)
        if rc != 0:
            raise RuntimeError(f"msnap_solve_grid_device failed: {rc}")


def timed_steps(torch, dist, batch, ctx, steps, warmup, use_graph, world):
    """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize."""
    stream = torch.cuda.current_stream()
    graph = None
    if use_graph:
        # the K-step inner loop is launch-bound: capture it once, replay it timed
        for _ in range(3):
            batch.step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: a process-group watchdog thread polling its events must not abort the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            # inside the context the current stream is torch's capture stream
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(steps):
                batch.step()
        ctx.set_stream(stream.cuda_stream)
        torch.cuda.synchronize()
    for _ in range(warmup):
        batch.step()
    if graph is not None:
        graph.replay()           # one untimed replay (graph upload)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        for _ in range(steps):
            batch.step()
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1)     # HIP events on the stream the kernels ran on
    return wall, dev_ms


def python_restatement_rate(wp, t, order, n=24):
    """Calibration only: the NumPy restatement that follows the reference step for step
    (oracle/msnap_oracle.py, dense assembly + np.linalg.solve per axis, bit-identical to the
    reference), single core -- the reference itself measures ~100 trajectories/s."""
    import msnap_oracle
    t0 = time.perf_counter()
    msnap_oracle.solve_batch(wp[:n], t[:n], ncoef=order + 1)
    return n / (time.perf_counter() - t0)


def cpu_baseline(n_seg, order, seconds):
    """C restatement of the reference algorithm (oracle/msnap_oracle.c, one dense
    LU with partial pivoting per axis like calculate_trajectory4D) on the host
    cores, bounded to ~`seconds` of wall time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from drone_path_planning_python_amd.synthetic import swarm
    # the box gives one GPU a 16-core share of the host; never oversubscribe it
    threads = max(1, min(c_oracle.max_threads(), os.cpu_count() or 1, 16))
    wp, t = swarm(2, 256, n_seg)
    # calibrate on ~1 s of work, then size the sample to ~`seconds`
    reps = 8
    while True:
        wpx, tx = np.tile(wp, (reps, 1, 1)), np.tile(t, (reps, 1))
        t0 = time.perf_counter()
        c_oracle.solve_batch(wpx, tx, ncoef=order + 1, faithful=True, n_threads=threads)
        probe = time.perf_counter() - t0
        if probe > 0.5 or reps >= 1 << 14:
            break
        reps *= 4
    reps = int(max(reps, min(1 << 16, reps * seconds / probe)))
    n = 256 * reps
    wpx, tx = np.tile(wp, (reps, 1, 1)), np.tile(t, (reps, 1))
    t0 = time.perf_counter()
    _, _, info, used = c_oracle.solve_batch(wpx, tx, ncoef=order + 1, faithful=True, n_threads=threads)
    dt = time.perf_counter() - t0
    n1 = 256 * max(1, reps // (4 * threads))
    t1 = time.perf_counter()
    c_oracle.solve_batch(wpx[:n1], tx[:n1], ncoef=order + 1, faithful=True, n_threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": n / dt, "unit": "trajectories/s", "cores": int(used), "kind": "port",
        "sample": f"{n} trajectories ({reps} x the 256-drone x {n_seg}-segment batch), dense "
                  f"{(order + 1) * n_seg}x{(order + 1) * n_seg} LU with partial pivoting per axis "
                  f"(C restatement of the reference algorithm), OpenMP over drones, {dt:.1f} s",
        "single_thread_value": n1 / dt1,
        "host_cpus": os.cpu_count(),
        "python_restatement_single_core": python_restatement_rate(wp, t, order),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_pg = world > 1 or (args.force_pg and "RANK" in os.environ)
    if use_pg:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)   # RCCL
        else:
            dist.init_process_group(backend="gloo")

    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm

    M, order = args.segments, args.order
    ctx = Context(device_id=dev_index, order=order, max_segments=max(M, 64))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    wp, t = swarm(2, args.drones, M, shared_times=args.shared_times, rank=rank)
    batch = DeviceBatch(torch, ctx, wp, t, M, order, device)
    use_graph = not args.no_graph
    wall, dev_ms = timed_steps(torch, dist, batch, ctx, args.steps, args.warmup, use_graph, world)
    assert int(batch.status.abs().sum().item()) == 0, "solve reported per-drone failures"

    times = torch.tensor([wall, dev_ms], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if use_pg:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    wall_max, dev_ms_max = float(times[0].item()), float(times[1].item())

    sat = None
    if rank == 0 and not args.no_saturated:
        nbig = args.saturated_drones
        wpb, tb = swarm(2, 4096, M, shared_times=args.shared_times)
        reps = (nbig + 4095) // 4096
        wpb = np.tile(wpb, (reps, 1, 1))[:nbig]
        tb = tb if tb.ndim == 1 else np.tile(tb, (reps, 1))[:nbig]
        big = DeviceBatch(torch, ctx, wpb, tb, M, order, device)
        # ~30 ms of sustained load before timing: the GPU's clocks settle over the first ~40 launches
        # (0.82 -> 0.68 ms per launch, tools/sat_ramp.py); the headline leg is warmed by its own replays
        ksat, wsat = 40, 40
        _, sat_ms = timed_steps(torch, dist, big, ctx, ksat, wsat, False, 1)
        assert int(big.status.abs().sum().item()) == 0
        per = sat_ms / ksat * 1e-3
        b = algorithmic_bytes(nbig, M, order)
        sat = {
            "workload": f"{nbig} drones x {M} segments, order {order}, one launch",
            "value": nbig / per, "unit": "trajectories/s", "ms_per_launch": per * 1e3,
            "roofline": {"bound": "hbm", "achieved": b / per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b / per / 1e9 / HBM_PEAK_GBS, "frac_of_copy_achievable": b / per / 1e9 / HBM_COPY_GBS,
                         "kernel": solve_kernel_name(nbig, M, order),
                         "traffic": pmc_traffic(nbig, M, order)[0]},
        }
        del big

    # the reference's own usage pattern: every drone on one shared uniform grid -> K2
    # (operator built once by msnap_grid_prepare, then one fp64 MFMA GEMM per batch).
    # Reported beside the headline, never as `value`: the factorisation is outside the step.
    grid = None
    if rank == 0 and not args.no_shared_grid:
        wps, ts = swarm(2, args.drones, M, shared_times=True)
        ctx.prepare_grid(ts)
        gsmall = GridBatch(torch, ctx, wps, M, order, device)
        _, g_ms = timed_steps(torch, dist, gsmall, ctx, args.steps, args.warmup, use_graph, 1)
        assert int(gsmall.status.abs().sum().item()) == 0
        nbig = args.saturated_drones
        wpb = np.tile(wps, ((nbig + args.drones - 1) // args.drones, 1, 1))[:nbig]
        gbig = GridBatch(torch, ctx, wpb, M, order, device)
        _, gb_ms = timed_steps(torch, dist, gbig, ctx, 40, 40, False, 1)
        per_s, per_b = g_ms / args.steps * 1e-3, gb_ms / 40 * 1e-3
        bs, bb = algorithmic_bytes(args.drones, M, order), algorithmic_bytes(nbig, M, order)
        grid = {
            "workload": f"shared uniform time grid t_i = i*10/(M+1) (scripts/drones_pols_generator.py:44-46), "
                        f"operator prepared once, K2 fp64 MFMA GEMM per step",
            "kernel": "msnap::grid_gemm_kernel<%d, %d>" % (order + 1, M),
            "headline_shape": {"drones": args.drones, "value": args.drones / per_s, "us_per_step": per_s * 1e6,
                               "roofline_frac": bs / per_s / 1e9 / HBM_PEAK_GBS},
            "saturated": {"drones": nbig, "value": nbig / per_b, "ms_per_launch": per_b * 1e3,
                          "achieved_GBps": bb / per_b / 1e9, "roofline_frac": bb / per_b / 1e9 / HBM_PEAK_GBS},
        }
        del gbig, gsmall

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(M, order, args.cpu_seconds)

    if use_pg:
        dist.barrier()
    if rank == 0:
        total = args.drones * world * args.steps
        per_launch_s = dev_ms_max * 1e-3 / args.steps
        bytes_launch = algorithmic_bytes(args.drones, M, order)
        kname = solve_kernel_name(args.drones, M, order)
        traffic = pmc_traffic(args.drones, M, order)[0]
        line = {
            "metric": METRIC,
            "value": total / wall_max,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: {args.drones} drones x {M} segments, order {order}, seeded random "
                            f"waypoints, {'shared uniform' if args.shared_times else 'per-drone random'} time "
                            "grid, per GPU per step",
                "drones_per_gpu": args.drones, "segments": M, "order": order,
                "parallelism": f"drone-sharded x{world}, no data-path collective",
                "launch": "hipGraph of K steps" if use_graph else "eager",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": bytes_launch / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_launch / per_launch_s / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_launch,
                "avg_launch_us": per_launch_s * 1e6,
            },
            "cpu_baseline": cpu,
            "saturated": sat,
            "shared_grid": grid,
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
