#!/usr/bin/env python3
"""bench.py -- minimum-snap trajectories/sec on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms give N ranks, one process per GPU.  Started plainly with `--gpus N`, N > 1 (no RANK / WORLD_SIZE in the
environment), this process starts the second form as a child BEFORE anything here touches HIP, relays rank 0's
JSON line and exits with the child's code.  `--gpus N` with a WORLD_SIZE that is not N, or with fewer than N GPUs
visible to the nccl backend, exits non-zero instead of printing a line for a job that is not the one asked for.

Headline (`value`, `roofline`): a "step" is one pass of the hot path (libmsnap's solve kernel, K1)
over one batch of synthetic input: BASELINE.json configs[1], 256 drones x 10 segments, order 7,
seeded random waypoints, per-drone random time grids (the general case: nothing is shared or
precomputed between drones).  Inputs and outputs live in HBM (torch tensors are only the
allocator); every rank solves its own 256-drone batch (weak scaling, no data-path collective:
the solve shards by drone).  K steps are captured in one hipGraph and replayed between
barrier + synchronize brackets.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     HBM roofline of the solve kernel at the headline workload
               (algorithmic bytes: SURVEY.md 8d, 3080 B/trajectory at M=10,p=7)
  cpu_baseline the C restatement of the reference algorithm (oracle/, "port") timed on this
               box's host cores on a bounded sample: B0 (one dense LU per axis, as the
               reference) and B1 (one LU for the four right-hand sides), one thread and all cores
  configs      every BASELINE.json config on this many GPUs, each with per-stage microseconds
               (HIP events on the launching stream, median over the repetitions, max over ranks), roofline fraction and bound:
                 [1] 256 x 10 (the headline again)
                 [2] 4096 x 10: solve -> sample -> pairwise formation pass
                 [3] 4096 x 20: the same + sweep against resources/stl/env-scene-hole.stl and
                     env-scene-ltu-experiment.stl
                 [4] 65536 x 10, order 9 (its launches under a hipGraph, like the headline's)
               For N > 1 configs [2]-[4] are SHARDED by drone over the ranks (strong scaling): the
               formation pass becomes sample -> RCCL all-gather of the positions -> either this rank's PART
               of the swarm's unordered pairs (every pair on exactly one rank) -> all-gather of the
               12-byte-per-drone partial minima -> fold, or -- where the exact broad phase leaves so few
               pairs that a rank is quicker alone -- the pass over the whole gathered swarm on every rank;
               the collectives are inside the timed region.
               [2]/[3] run on the formation-like inputs pinned by tests/golden/formation_golden.npz
               and report their hit counts beside the fixture's.
  strong_scaling  total work fixed and sharded over the ranks (what a SCALE curve over N should be read
               from; `value` is the 256-drone-per-GPU config of the metric and scales trivially):
               configs[4], 65536 x 10 at order 7, the 4096-drone and a 16384-drone formation pipeline
               (one GPU: the 4096-drone pipeline also with the pairwise pass's broad phase off)
  saturated    the solve kernel on a batch large enough to fill the chip (2^20 drones), with its
               own roofline fraction and the parity of its first and last 4096 drones against
               the C oracle -- the 256-drone headline is launch/latency-bound by construction
  shared_grid  the reference's own usage (every drone on one uniform grid): K2, fp64 MFMA GEMM
  end_to_end   configs[1] through the host-pointer entry point with page-locked arrays
               (H2D + kernel + D2H per call); PCIe-inclusive, never `value`
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
# fp64 vector issue peak: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz (= 78.6 TFLOP/s of FMA / 2)
VALU_F64_OPS = 256 * 4 * 16 * 2.4e9
XGMI_LINK_GBS = 153.0          # per link and direction; 7 links per GPU (full mesh)
PAIR_OPS = 7                   # vector instructions per pair and sample: 3 differences, 1 product, 2 FMAs, 1 minimum
PAIR_FLOPS = 9                 # the same as floating-point operations (an FMA counts two)
MESH_TEST_OPS = 82             # fp64 vector operations of one full point-triangle closest-point evaluation (pt_tri_d2)
PAIR_MIX_ATTAINABLE = 0.897    # the 7-instruction mix alone, 4 waves per SIMD: 4.46 nominal cycles per wave-instruction
                               # (tools/micro/f64_rate_micro.hip: the SIMDs run at ~2.15 GHz under sustained fp64 load)


def algorithmic_bytes(n_drones: int, n_seg: int, order: int) -> int:
    """in = 8*5*(M+1) (x,y,z,yaw,t), out = 8*M*(1 + 4*(p+1))."""
    return n_drones * (8 * 5 * (n_seg + 1) + 8 * n_seg * (1 + 4 * (order + 1)))


def sampler_bytes(n_drones: int, n_seg: int, order: int, n_samples: int) -> int:
    """K4: reads the x, y, z rows and the durations, writes [S][3] positions."""
    return n_drones * 8 * (n_seg * (3 * (order + 1) + 1) + 3 * n_samples)


def built_csrc_sha():
    from drone_path_planning_python_amd._lib import csrc_sha
    return csrc_sha()


def pmc_traffic(n_drones: int, n_seg: int, order: int, kernel: str = None, prefix: str = ""):
    """(HBM bytes per launch, source note) of the solve at this workload from the committed PMC passes
    (profiles/pmc_traffic.json, written by tools/make_profiles.sh on an MI355X: separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  The counters are only quoted when
    they were taken from the kernel sources this run was built from (csrc_sha) and, if `kernel` is given, from the
    kernel instance this run launched; otherwise the bytes are None and the note says why."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tp) as f:
            rec = json.load(f)
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    ent = rec.get(f"{prefix}{n_drones}x{n_seg}o{order}")      # prefix "grid": the shared-grid GEMM's pass
    if not ent:
        return None, "profiles/pmc_traffic.json holds no pass for this workload"
    sha = built_csrc_sha()
    if ent.get("csrc_sha") != sha:
        return None, (f"profiles/pmc_traffic.json was taken from csrc {ent.get('csrc_sha', 'unrecorded')}, this build is "
                      f"{sha}: stale counters are not quoted (tools/make_profiles.sh renews them)")
    if kernel and ent.get("kernel") != kernel:
        return None, f"profiles/pmc_traffic.json measured {ent.get('kernel')}, this run launched {kernel}"
    return ent.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json, csrc {sha}, {ent.get('kernel')}"


def pmc_counter(kernel: str, counter: str):
    """(mean per launch of a PMC counter of one kernel on the bench's own workload, source note) from the committed
    passes (profiles/pmc_counters.json, tools/make_pipeline_profile.sh); None unless taken from this build's sources."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_counters.json")) as f:
            rec = json.load(f)
    except Exception:
        return None, "no profiles/pmc_counters.json"
    sha = built_csrc_sha()
    if rec.get("csrc_sha") != sha:
        return None, (f"profiles/pmc_counters.json was taken from csrc {rec.get('csrc_sha', 'unrecorded')}, this build is "
                      f"{sha}: not quoted")
    return rec.get(kernel, {}).get(counter), f"profiles/pmc_counters.json, csrc {sha}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE if a launcher set it, else 1); without a "
                         "launcher environment N > 1 starts the N ranks itself")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring the ranks up: process group, one all-reduce, a short JSON line (no kernels; "
                         "with --backend gloo this needs no GPU)")
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
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config pipelines")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the several-batches-in-flight leg")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling legs (65536 x 10 order 7, 16384-drone formation)")
    ap.add_argument("--saturated-drones", type=int, default=1 << 20)
    ap.add_argument("--config-reps", type=int, default=0, help="timed repetitions per config pipeline (0: from --steps)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--force-pg", action="store_true", help="initialise the process group even for one rank (rehearsal)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse N > 1 on a 1-GPU box "
                         "(ranks then share device local_rank %% device_count, the all-gather runs on host copies)")
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
        self._args = (ctx._h, self.n, n_seg, ctypes.c_void_p(self.wp.data_ptr()), ctypes.c_void_p(self.coef.data_ptr()),
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
    # the events exist (and have been recorded once) before the clock starts: torch creates the HIP event at its first
    # record(), and two creations inside a 20-step region were ~10 us of its ~125 (tools/replay_overhead.py: the first
    # region of a process against the later ones)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    if graph is not None:
        graph.replay()           # one untimed replay (graph upload)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
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


def concurrent_streams_leg(torch, Context, dev_index, device, wp, t, n_seg, order, steps, n_streams):
    """The headline shape as `n_streams` independent batches in flight: one context (own stream) per batch, the
    launches captured as parallel branches of ONE hipGraph (fork from and join into the capture stream), each branch
    steps / n_streams dependent launches on its own buffers.  256 drones are 32 wavefronts on 32 of 256 CUs, so
    batches of different swarms can share the chip; reported beside the headline, never as `value` (the contract's
    step is serial)."""
    per = max(1, steps // n_streams)
    ctxs = [Context(device_id=dev_index, order=order, max_segments=max(64, n_seg)) for _ in range(n_streams)]
    batches = [DeviceBatch(torch, c, wp, t, n_seg, order, device) for c in ctxs]
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    try:
        for c, b, st in zip(ctxs, batches, streams):      # buffers sized, kernels loaded, outside the capture
            c.set_stream(st.cuda_stream)
            for _ in range(3):
                b.step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            cap = torch.cuda.current_stream()
            for c, b, st in zip(ctxs, batches, streams):
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    for _ in range(per):
                        b.step()
            for st in streams:
                cap.wait_stream(st)
        torch.cuda.synchronize()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        ms = float("inf")
        for _ in range(3):      # best of three replays: this leg has measured 2.6 and 4.4 us per launch on different boxes
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = min(ms, e0.elapsed_time(e1))
        ok = all(int(b.status.abs().sum().item()) == 0 for b in batches)
        same = all(torch.equal(b.coef.view(torch.uint8), batches[0].coef.view(torch.uint8)) for b in batches[1:])
        return {"streams": n_streams, "steps": per * n_streams, "ms": ms, "us_per_step": ms * 1e3 / (per * n_streams),
                "value": wp.shape[0] * per * n_streams / (ms * 1e-3), "unit": "trajectories/s",
                "all_status_ok": ok, "branches_bit_identical": same}
    finally:
        torch.cuda.synchronize()
        del graph
        for c in ctxs:
            c.use_own_stream()
            c.close()


# ------------------------------------------------------------------------------------------
# CPU baseline (oracle/, checker code: timed here, never on the product path)
# ------------------------------------------------------------------------------------------
def host_core_count() -> int:
    """Cores this job may use: the affinity mask, capped by a cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(n_seg, order, seconds):
    """C restatement of the reference algorithm (oracle/msnap_oracle.c) on the host cores, bounded to
    ~`seconds` of wall time in all: B0 = one dense LU with partial pivoting per axis like
    calculate_trajectory4D (the `value`), B1 = one LU for the four right-hand sides; each on all cores
    this job may use and on one thread.  Plus the NumPy restatements on one core (calibration: the
    reference itself measures ~100 trajectories/s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import msnap_oracle
    from drone_path_planning_python_amd.synthetic import swarm
    threads = max(1, min(c_oracle.max_threads(), host_core_count()))
    wp, t = swarm(2, 256, n_seg)

    def rate(faithful, n_threads, budget):
        reps, probe = 2, 0.0
        while True:     # calibrate, then size the sample to the budget
            wpx, tx = np.tile(wp, (reps, 1, 1)), np.tile(t, (reps, 1))
            t0 = time.perf_counter()
            c_oracle.solve_batch(wpx, tx, ncoef=order + 1, faithful=faithful, n_threads=n_threads)
            probe = time.perf_counter() - t0
            if probe > 0.25 or reps >= 1 << 14:
                break
            reps *= 4
        reps = int(max(1, min(1 << 16, reps * budget / probe)))
        wpx, tx = np.tile(wp, (reps, 1, 1)), np.tile(t, (reps, 1))
        t0 = time.perf_counter()
        _, _, _, used = c_oracle.solve_batch(wpx, tx, ncoef=order + 1, faithful=faithful, n_threads=n_threads)
        dt = time.perf_counter() - t0
        return 256 * reps / dt, 256 * reps, int(used), dt

    b0_all, n0, used, dt0 = rate(True, threads, 0.45 * seconds)
    b1_all, n1, _, dt1 = rate(False, threads, 0.2 * seconds)
    b0_one, _, _, _ = rate(True, 1, 0.12 * seconds)
    b1_one, _, _, _ = rate(False, 1, 0.08 * seconds)
    t0 = time.perf_counter()
    msnap_oracle.solve_batch(wp[:24], t[:24], ncoef=order + 1)
    np_b0 = 24 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    msnap_oracle.solve_batch_fast(wp, t, ncoef=order + 1)
    np_b1 = 256 / (time.perf_counter() - t0)
    n = (order + 1) * n_seg
    return {
        "value": b0_all, "unit": "trajectories/s", "cores": used, "kind": "port",
        "sample": f"B0: {n0} trajectories (copies of the 256-drone x {n_seg}-segment batch), one dense {n}x{n} LU "
                  f"with partial pivoting per axis (C restatement of the reference algorithm), OpenMP over drones "
                  f"on all {used} cores this job may use, {dt0:.1f} s; B1: {n1} trajectories, {dt1:.1f} s",
        "single_thread_value": b0_one,
        "b1_one_lu_four_rhs": {"value": b1_all, "cores": used, "single_thread_value": b1_one},
        "numpy_restatement_single_core": {"b0_reference_faithful": np_b0, "b1_vectorised_batched_solve": np_b1},
        "host_cpus": os.cpu_count(),
    }


def oracle_parity(coef_dev, wp, t, order, idx):
    """max norm-relative error of the device coefficients of drones `idx` against the C oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    tt = t if t.ndim == 1 else t[idx]
    ref, _, info, _ = c_oracle.solve_batch(wp[idx], tt, ncoef=order + 1, faithful=False, n_threads=0)
    got = coef_dev[idx].cpu().numpy()
    num = np.abs(got - ref).max(axis=(1, 3))
    den = np.abs(ref).max(axis=(1, 3))
    return float((num / np.where(den == 0, 1.0, den)).max()), bool(info.any())


# ------------------------------------------------------------------------------------------
# BASELINE.json configs[2..4]: pipelines sharded by drone over the ranks
# ------------------------------------------------------------------------------------------
class HostGather:
    """gloo rehearsal only: the collective on host copies of the position blocks."""

    def __init__(self, dist):
        self.dist = dist

    def all_gather_into_tensor(self, out, inp):
        o, i = out.cpu(), inp.cpu()
        self.dist.all_gather_into_tensor(o, i)
        out.copy_(o)


def max_over_ranks(torch, dist, values, device, use_pg):
    tv = torch.tensor(values, dtype=torch.float64, device=device)
    if use_pg:
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
    return [float(x) for x in tv.tolist()]


def sum_over_ranks(torch, dist, values, device, use_pg):
    tv = torch.tensor(values, dtype=torch.int64, device=device)
    if use_pg:
        dist.all_reduce(tv, op=dist.ReduceOp.SUM)
    return [int(x) for x in tv.tolist()]


def run_formation_config(cfg, env, reps, warm, n_groups=512):
    """configs[2] (cfg = 2) / configs[3] (cfg = 3): solve -> sample -> [all-gather ->] pairwise pass
    [-> all-gather of the partial minima -> fold] (-> mesh sweep), n_groups x 8 drones (4096 at the default)
    sharded by drone.  One rank: one symmetric launch over the whole swarm, fed by the sampler's row image.
    Several ranks: every unordered pair on exactly one rank (msnap_formation_collide_part), two collectives.
    Returns the report dict (rank 0) or None."""
    torch, dist, ctx, device = env["torch"], env["dist"], env["ctx7"], env["device"]
    rank, world, use_pg, coll, red_dev = env["rank"], env["world"], env["use_pg"], env["coll"], env["red_dev"]
    from drone_path_planning_python_amd import stl, swarm, synthetic
    rb, off, t = synthetic.formation_config(cfg, n_groups)
    G, m, _ = rb.shape
    M, N, order = m - 1, G * off.shape[0], 7
    # inputs through a8 on the GPU (untimed: the producer of the waypoints, not the path being timed)
    wp = synthetic.formation_waypoints(ctx.formation_transform(rb.reshape(G * m, 7), off), G)
    lo, hi = swarm.shard_bounds(N, world, rank)
    n = hi - lo
    S = synthetic.formation_sample_count(t)
    tris = None
    side_ctx = None
    if cfg == 3:
        gd = os.path.join(ROOT, "tests", "golden")    # the reference's resources/stl files, copied as data
        tris = torch.from_numpy(np.concatenate([stl.load_stl(os.path.join(gd, "env-scene-hole.stl")),
                                                stl.load_stl(os.path.join(gd, "env-scene-ltu-experiment.stl"))])
                                ).to(device)
        from drone_path_planning_python_amd import Context
        side_ctx = Context(ctx.device_id, 7, 16)      # the mesh sweep's own stream (swarm.DeviceCompute)
    # (output tensors allocated once and reused by every pipeline: a pass's results are consumed before the next starts)
    comp = swarm.DeviceCompute(ctx, torch, side_ctx=side_ctx, reuse_outputs=True)
    twp = torch.from_numpy(np.ascontiguousarray(wp[lo:hi])).to(device)
    tt = torch.from_numpy(t).to(device)
    # several ranks: "parts" (every pair on exactly one rank, second all-gather, fold) or "whole" (every rank the whole
    # gathered swarm behind the broad phase, no second collective) -- swarm.DeviceCompute.pairwise_mode decides from the
    # survivor counts of a first whole pass, here in an untimed probe pipeline below
    mode = {"pairwise": "parts"}
    stage_names, nst, ev = [], 0, []

    def set_stages():
        nonlocal stage_names, nst, ev
        multi = (["allgather", "pairwise"] if mode["pairwise"] == "whole" else ["allgather", "pairwise", "allgather_parts", "fold"])
        stage_names = ["solve", "sample"] + (multi if world > 1 else ["pairwise"]) + (["mesh"] if cfg == 3 else [])
        nst = len(stage_names)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(reps)]
    set_stages()

    def one(rec, overlap, fused=False):
        """one pipeline; `overlap`: the mesh sweep on the side stream next to the exchange and the pairwise
        pass (what is timed as the pipeline), else everything in stream order (what the stage times are);
        `fused`: solve and sampler through msnap_solve_grid_sample_device (one launch up to 11 segments)"""
        k = 0

        def mark():
            nonlocal k
            if rec:
                rec[k].record()
            k += 1
        mark()
        rows_t = None
        if fused:
            coef, dur, status, pos, rows_t = comp.solve_grid_sample(twp, synthetic.SAMPLE_DT, S, n_cols=N if world == 1 else None)
        else:
            coef, dur, status = comp.solve_grid(twp)      # the swarm shares the reference's uniform grid: K2
            mark()
            if world == 1:      # the sampler also writes its hand-over to the pairwise pass, if that is going to read it
                pos, rows_t = comp.sample_rows_t(coef, dur, synthetic.SAMPLE_DT, S, n_cols=N)
            else:
                pos = comp.sample(coef, dur, synthetic.SAMPLE_DT, S)
        mark()
        try:
            if tris is not None and overlap:
                comp.mesh_begin(pos, tris, synthetic.DRONE_RADIUS)
            if world > 1 and mode["pairwise"] == "whole":
                pos_all = swarm.all_gather_positions(pos, N, world, rank, coll, torch)
                mark()
                md, partner, hit = (x[lo:hi] for x in comp.collide(pos_all, 0, pos_all, synthetic.DRONE_RADIUS))
            elif world > 1:
                pos_all = swarm.all_gather_positions(pos, N, world, rank, coll, torch)
                mark()
                part = comp.collide_part(pos_all, rank, world)        # this rank's share of the swarm's pairs
                mark()
                parts = swarm.all_gather_parts(part, world, coll, torch)
                mark()
                md, partner, hit = comp.collide_finish(parts, N, lo, n, synthetic.DRONE_RADIUS)
            else:
                md, partner, hit = comp.collide(pos, lo, pos, synthetic.DRONE_RADIUS, rows_t=rows_t)
            mark()
            mh = None
            if tris is not None:
                _, mh = comp.mesh_end() if overlap else comp.mesh(pos, tris, synthetic.DRONE_RADIUS)
                mark()
        finally:
            comp.mesh_abort()       # joins the side stream if something above raised between begin and end
        return status, hit, mh, md, pos

    host = {"enqueue_s": 0.0}

    def timed(overlap, recs, fused=False):
        for _ in range(warm):
            one(None, overlap, fused)
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            out = one(recs[r] if recs else None, overlap, fused)
        host["enqueue_s"] = time.perf_counter() - t0      # the host's share: the loop without the final drain
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        if use_pg:
            dist.barrier()
        return w, out

    ctx.prepare_grid(t)      # the grid's operator, once per grid (untimed, like the context itself)
    if world > 1 and comp.pairwise_mode(N, S, world) == "whole":
        # probe (untimed): one pipeline with the whole-swarm pass, then its survivor counts decide (same on every rank)
        mode["pairwise"] = "whole"
        set_stages()
        one(None, False)
        comp.note_whole_pass(N, S, world, dist if use_pg else None)      # (the ranks agree: one MIN all-reduce)
        mode["pairwise"] = comp.pairwise_mode(N, S, world)
        set_stages()
    # pass 1, stream order: per-stage times (events between the stages)
    wall, (status, hit, mh, md, pos_keep) = timed(False, ev)
    grid_kernel = ctx.last_kernel()     # what the library launched for the shared-grid solve (msnap_last_kernel)
    # median over the repetitions: one preempted launch must not pass for a stage's time
    stage_us = [float(np.median([ev[r][k].elapsed_time(ev[r][k + 1]) for r in range(reps)])) * 1e3 for k in range(nst)]
    # pass 2: the pipeline as it is run -- no events between the stages, and (configs[3]) the mesh sweep beside the
    # exchanges and the pairwise pass
    wall_serial = wall
    stage_out = [x.clone() for x in (status, hit, md, pos_keep)]
    wall, (status, hit, mh, md, pos_keep) = timed(tris is not None, None, fused=True)
    host_us = host["enqueue_s"] / reps * 1e6
    fused_kernel = None
    # the fused launch on its own (and that it gives the separate stages' results bit for bit)
    for a, b in zip(stage_out, (status, hit, md, pos_keep)):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), "fused solve + sampler differs from the two stages"
    # (buffers allocated once, the C-ABI entry called directly: with five torch.empty per call the loop is host-bound)
    fb = comp.solve_grid_sample(twp, synthetic.SAMPLE_DT, S, n_cols=N if world == 1 else None)
    fused_kernel = ctx.last_kernel()

    def fused_call():
        ctx.solve_grid_sample_device(n, M, twp, synthetic.SAMPLE_DT, S, fb[0], fb[1], fb[2], fb[3], fb[4])
    for _ in range(10):
        fused_call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5 * reps):
        fused_call()
    e1.record()
    torch.cuda.synchronize()
    fused_us = e0.elapsed_time(e1) / (5 * reps) * 1e3

    def separate_call():      # the same outputs by the two launches, timed the same way
        ctx.solve_grid_device(n, M, twp, fb[0], fb[1], fb[2])
        if fb[4] is not None:
            ctx.sample_collide_device(n, M, fb[0], fb[1], synthetic.SAMPLE_DT, S, fb[3], fb[4])
        else:
            ctx.sample_device(n, M, fb[0], fb[1], synthetic.SAMPLE_DT, S, 3, fb[3])
    for _ in range(10):
        separate_call()
    e0.record()
    for _ in range(5 * reps):
        separate_call()
    e1.record()
    torch.cuda.synchronize()
    separate_us = e0.elapsed_time(e1) / (5 * reps) * 1e3
    # the per-drone-grid kernel (K1: any time grids) on the same shard, outside the pipeline
    for _ in range(3):
        comp.solve(twp, tt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        comp.solve(twp, tt)
    e1.record()
    torch.cuda.synchronize()
    gemm_us = e0.elapsed_time(e1) / reps * 1e3
    k1_kernel = ctx.last_kernel()
    whole = world > 1 and mode["pairwise"] == "whole"      # every rank the whole gathered swarm behind the broad phase
    pair_once = bool(ctx.get_option("collide_last_sym")) if world == 1 else not whole   # (parts: by construction)
    # a whole swarm of 3072..16384 drones: the pass runs behind its exact broad phase (msnap.h "collide_no_cull")
    broad = bool(ctx.get_option("collide_last_cull")) if (world == 1 or whole) else False
    shares, survivors = (ctx.get_option("collide_last_shares"), ctx.get_option("collide_last_survivors")) if broad else (0, 0)
    group_pairs = ctx.get_option("collide_last_group_pairs") if broad else 0
    # which list the pass's evaluator walked -- the library's own report (msnap.h "collide_last_by_groups")
    by_groups = bool(ctx.get_option("collide_last_by_groups")) if broad else False
    mesh_extra = None
    if cfg == 3:
        # how many point-triangle tests the kernel's exact bounding-box cull leaves to evaluate (one counted launch)
        ctx.set_option("mesh_count_tests", 1)
        comp.mesh(pos_keep, tris, synthetic.DRONE_RADIUS)
        mesh_extra = ctx.get_option("mesh_count_tests")
        ctx.set_option("mesh_count_tests", 0)
    torch.cuda.synchronize()
    comp.close()                 # drops the wrapper of the side stream before its context goes
    if side_ctx is not None:
        side_ctx.close()
    mx = max_over_ranks(torch, dist, [wall] + stage_us + [gemm_us, wall_serial, fused_us, separate_us], red_dev, use_pg)
    cnt = sum_over_ranks(torch, dist, [int(status.abs().sum().item()), int(hit.sum().item()),
                                       int(mh.sum().item()) if mh is not None else 0], red_dev, use_pg)
    if rank != 0:
        return None
    wall_max, st, gemm, wall_serial_max = mx[0], dict(zip(stage_names, mx[1:1 + nst])), mx[1 + nst], mx[2 + nst]
    fused_max, separate_max = mx[3 + nst], mx[4 + nst]
    is_fused = fused_kernel.startswith("msnap::grid_sample_kernel")
    per = wall_max / reps
    fix = np.load(os.path.join(ROOT, "tests", "golden", "formation_golden.npz")) if n_groups == 512 else None
    n_max = max(swarm.shard_sizes(N, world))
    pair_alg = N * (N - 1) / 2 * S * PAIR_OPS                 # every unordered pair once
    # what a rank evaluates: 1 / world of the swarm's (128-row block, column) units, each pair of them once; inside the
    # diagonal blocks both orders of a pair are computed (a 128 x 128 block instead of its triangle)
    pair_exec = (N * (N - 1) / 2 + 64.0 * N) / world * S * PAIR_OPS
    if broad:       # the surviving shares (128 rows x 8 columns each) or the surviving group pairs (8 x 8)
        pair_exec = (group_pairs * 64.0 if by_groups else survivors * 128.0 * 8.0) * S * PAIR_OPS
    rep = {
        "workload": f"configs[{cfg}]: {N} drones x {M} segments, order 7, formation-like swarm ({G} rigid bodies x "
                    f"{off.shape[0]} offsets through a8) on the reference's uniform grid, {S} samples at dt = "
                    f"{synthetic.SAMPLE_DT} s" + (f", {tris.shape[0]}-triangle scene (env-scene-hole.stl + "
                                                   f"env-scene-ltu-experiment.stl)" if tris is not None else ""),
        "sharding": f"{world} rank(s) x {n_max} drones (by drone, strong scaling" +
                    ("; the pairwise pass on the whole gathered swarm on every rank" if whole else "") + ")", "rccl_ranks": world,
        "reps": reps, "warm_reps": warm, "value": N / per, "unit": "trajectories/s", "us_per_pipeline": per * 1e6,
        "stage_us": st,
        "host_enqueue_us_per_pipeline": host_us,
        "host_note": "host_enqueue_us_per_pipeline: Python time to enqueue one pipeline (the loop without its final drain); "
                     "outputs are reused buffers (swarm.DeviceCompute(reuse_outputs=True)): with eleven torch.empty per "
                     "pipeline the host needed 93 us per configs[3] pipeline and the GPU waited for it",
        "stage_us_note": "stages timed in stream order (events between them), solve and sampler as separate launches; "
                         "us_per_pipeline from a second pass without events" + (
            ", in which solve and sampler are the one fused launch of stages.solve_sample" if is_fused else "") + (
            "; us_per_pipeline is the pipeline as run: the mesh sweep on a side stream beside the "
            f"{'all-gather and the ' if world > 1 else ''}pairwise pass ({wall_serial_max / reps * 1e6:.1f} us in stream order)"
            if tris is not None else ""),
        "solve_failures": cnt[0], "pairwise_hits": cnt[1],
        "pairwise_hits_fixture": int(fix[f"cfg{cfg}_pair_hit_idx"].size) if fix is not None else None,
        "stages": {
            "solve": {"kernel": grid_kernel, "bound": "hbm",
                      "frac": algorithmic_bytes(n_max, M, order) / (st["solve"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                      "note": "the swarm shares the reference's uniform time grid: fp64 MFMA GEMM against the grid's "
                              "operator (prepared once, untimed)"},
            "solve_any_grid_k1": {"kernel": k1_kernel, "bound": "hbm", "us": gemm,
                                  "frac": algorithmic_bytes(n_max, M, order) / (gemm * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  "note": "same shard through the per-drone recurrence (what a swarm with per-drone "
                                          "time grids takes), outside the pipeline time"},
            "sample": {"kernel": "msnap::sample_kernel", "bound": "hbm",
                       "frac": sampler_bytes(n_max, M, order, S) / (st["sample"] * 1e-6) / 1e9 / HBM_PEAK_GBS},
            "solve_sample": {"kernel": fused_kernel if is_fused else fused_kernel + " + msnap::sample_kernel",
                             "fused": is_fused, "us": fused_max, "separate_us": separate_max, "bound": "hbm",
                             # one launch: waypoints in; coefficients, durations, status and positions out (the
                             # sampler's read-back of the coefficients stays in LDS)
                             "frac": (algorithmic_bytes(n_max, M, order) + n_max * S * 24) / (fused_max * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "note": "msnap_solve_grid_sample_device, what the pipeline as run uses (us: back to back on "
                                     "preallocated buffers; separate_us: msnap_solve_grid_device + the sampler the same way): the fp64 MFMA "
                                     "product and the Horner loops in one launch up to 11 segments (coefficients through "
                                     "LDS, outputs bit-identical to the two stages -- asserted here on every run); longer "
                                     "paths gain nothing from it and run the two kernels"},
            "pairwise": {"kernel": (("msnap::" + ("" if world == 1 else "collide_key_kernel + ") + "collide_rank_kernel + "
                                     "collide_gather_kernel + collide_select_kernel + " +
                                     ("collide_eval_groups_kernel + collide_finish_groups_kernel (exact broad phase; the pass's "
                                      "arithmetic on the surviving 8 x 8 group pairs" if by_groups else
                                      "collide_eval_shares_kernel + collide_merge_kernel (exact broad phase; the pass's "
                                      "arithmetic on the surviving 128 x 8 shares") +
                                     ("; boxes and sort keys from the sampler)" if world == 1 else ")")
                                     if broad else
                                     "msnap::collide_span_kernel + collide_merge_kernel (rows from the sampler's image)")
                                    if (world == 1 or whole) else "msnap::collide_transpose_kernel + collide_span_kernel + "
                                    "collide_merge_kernel on this rank's part of the swarm's pairs"),
                         "launches_per_pass": (5 + (0 if world == 1 else 1)) if broad else (2 if world == 1 else 3),
                         "ranks": (None if world == 1 else
                                   "every rank runs the pass over the whole gathered swarm behind the broad phase and keeps "
                                   "its rows: no second collective (swarm.DeviceCompute.pairwise_mode, decided from the "
                                   "survivor counts of an untimed probe pass)" if whole else
                                   "every unordered pair on exactly one rank, second all-gather of the partial minima, fold"),
                         "bound": "valu_f64", "pairs_evaluated_once_over_all_ranks": pair_once,
                         "broad_phase": ({"shares": shares, "shares_surviving": survivors,
                                          "group_pairs": (N // 8) * (N // 8 + 1) // 2, "group_pairs_surviving": group_pairs,
                                          "evaluated": "group pairs" if by_groups else "shares",
                                          "cull_ratio": pair_exec / S / PAIR_OPS / (N * (N - 1) / 2)} if broad else None),
                         # the work the kernels evaluate (with the broad phase: the surviving shares) over time and peak
                         "frac": pair_exec / (st["pairwise"] * 1e-6) / VALU_F64_OPS,
                         "speedup_vs_all_pairs_issue_peak": pair_alg / world / (st["pairwise"] * 1e-6) / VALU_F64_OPS,
                         "tflops_on_all_pairs": pair_alg / PAIR_OPS * PAIR_FLOPS / world / (st["pairwise"] * 1e-6) / 1e12,
                         "frac_of_attainable": pair_exec / (st["pairwise"] * 1e-6) / VALU_F64_OPS / PAIR_MIX_ATTAINABLE,
                         "note": "frac = the pair-samples the kernels evaluate (N(N-1)/2 + the doubled triangle of the "
                                 "diagonal blocks; behind the broad phase: 128 x 8 per surviving share or 8 x 8 per surviving group pair) x 7 vector "
                                 "instructions (3 differences, d2 = fma(dz, dz, fma(dy, dy, dx*dx)), the minimum; SURVEY.md "
                                 "8d) / the stage's time -- sort, bounds, selection and merge included -- / the fp64 issue "
                                 "peak (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz); speedup_vs_all_pairs_issue_peak is NOT a roofline "
                                 "fraction: it is how many times faster the stage ran than an all-pairs pass at the issue "
                                 "peak would (every unordered pair credited once, culled or not; above 1 behind the broad "
                                 "phase); tflops_on_all_pairs "
                                 "counts that as 9 flops per pair-sample (peak 78.6 with nothing but FMAs), per GPU; "
                                 "frac_of_attainable divides frac by what the bare instruction mix reaches (0.897: "
                                 "tools/micro/f64_rate_micro.hip)"},
        },
    }
    if world > 1:
        coll_name = "all_gather_into_tensor (RCCL)" if env["coll"] is dist else "host rehearsal (gloo)"
        pb = ctx.formation_part_bytes(N)
        rep["stages"]["allgather"] = {
            "collective": coll_name, "bound": "xgmi", "bytes_received_per_rank": (N - n_max) * S * 24,
            "frac": (N - n_max) * S * 24 / (st["allgather"] * 1e-6) / 1e9 / (7 * XGMI_LINK_GBS)}
        if not whole:
            rep["stages"]["allgather_parts"] = {
                "collective": coll_name, "bound": "xgmi (latency)", "bytes_received_per_rank": (world - 1) * pb,
                "what": "per-drone partial minima of every rank's part: 12 bytes per drone and rank",
                "frac": (world - 1) * pb / (st["allgather_parts"] * 1e-6) / 1e9 / (7 * XGMI_LINK_GBS)}
            rep["stages"]["fold"] = {"kernel": "msnap::collide_finish_kernel", "bound": "latency"}
    if cfg == 3:
        rep["mesh_hits"] = cnt[2]
        rep["mesh_hits_fixture"] = int(fix["cfg3_mesh_hit_idx"].size) if fix is not None else None
        tests = n_max * S * int(tris.shape[0])
        evaluated = mesh_extra
        valu, valu_src = pmc_counter("mesh_sweep_kernel", "SQ_INSTS_VALU") if world == 1 else (None, "one rank only")
        rep["stages"]["mesh"] = {
            "kernel": "msnap::mesh_sweep_kernel", "bound": "valu_f64",
            "point_triangle_pairs": tests, "tests_evaluated": evaluated, "cull_ratio": evaluated / tests,
            "point_triangle_pairs_per_s": tests / (st["mesh"] * 1e-6),
            "tests_evaluated_per_s": evaluated / (st["mesh"] * 1e-6),
            # algorithmic work: a full closest-point evaluation is MESH_TEST_OPS fp64 vector operations
            "frac": evaluated * MESH_TEST_OPS / (st["mesh"] * 1e-6) / VALU_F64_OPS,
            "speedup_vs_all_pairs_issue_peak": tests * MESH_TEST_OPS / (st["mesh"] * 1e-6) / VALU_F64_OPS,
            "issue_utilisation": (valu * 64 / (st["mesh"] * 1e-6) / VALU_F64_OPS) if valu else None,
            "issue_utilisation_source": valu_src,
            "note": "exact cull: a triangle whose bounding box is farther from the wave's stretch of path than the best "
                    f"distance so far is skipped (cull_ratio = evaluated / all pairs).  frac = evaluated tests x {MESH_TEST_OPS} "
                    "fp64 vector operations (one full closest-point evaluation, Ericson 5.1.5: the dot products d1..d6, "
                    "the three edge and the face region, the squared distance; counted in csrc/msnap_aux.hip::pt_tri_d2) "
                    "/ time / fp64 issue peak; speedup_vs_all_pairs_issue_peak counts the culled pairs as done -- not a "
                    "roofline fraction, it may exceed 1; "
                    "issue_utilisation = SQ_INSTS_VALU x 64 lanes / time / peak from the committed counter pass, quoted "
                    "only when that pass was taken from this build's kernel sources"}
    return rep


def run_solve_config(env, N, M, order, reps, warm, label):
    """configs[4]: N drones x M segments sharded by drone, solve only.  The `reps` launches are captured in one
    hipGraph like the headline's steps (a launch of tens of microseconds is otherwise timed together with the gap
    an eager launch leaves between two kernels)."""
    torch, dist, device = env["torch"], env["dist"], env["device"]
    rank, world, use_pg, red_dev = env["rank"], env["world"], env["use_pg"], env["red_dev"]
    from drone_path_planning_python_amd import swarm
    from drone_path_planning_python_amd.synthetic import swarm as synth
    ctx = env["ctx9"] if order == 9 else env["ctx7"]
    lo, hi = swarm.shard_bounds(N, world, rank)
    wp, t = synth(5, N, M)                        # every rank builds the same swarm and keeps its shard
    batch = DeviceBatch(torch, ctx, wp[lo:hi], t[lo:hi], M, order, device)
    wall, dev_ms = timed_steps(torch, dist, batch, ctx, reps, warm, env.get("use_graph", True), world if use_pg else 1)
    kernel = ctx.last_kernel()
    fails = int(batch.status.abs().sum().item())
    pick = np.r_[0:min(256, hi - lo), max(0, hi - lo - 256):hi - lo]
    err, bad = oracle_parity(batch.coef, wp[lo:hi], t[lo:hi], order, pick)
    mx = max_over_ranks(torch, dist, [wall, dev_ms, err], red_dev, use_pg)
    cnt = sum_over_ranks(torch, dist, [fails], red_dev, use_pg)
    if rank != 0:
        return None
    n_max = max(swarm.shard_sizes(N, world))
    per, kus = mx[0] / reps, mx[1] / reps * 1e3
    return {
        "workload": f"{label}: {N} drones x {M} segments, order {order}, per-drone random time grids",
        "sharding": f"{world} rank(s) x {n_max} drones (by drone, strong scaling, no collective)",
        "reps": reps, "warm_reps": warm, "value": N / per, "unit": "trajectories/s", "us_per_pipeline": per * 1e6,
        "stage_us": {"solve": kus}, "solve_failures": cnt[0], "max_norm_rel_err_vs_oracle": mx[2],
        "stages": {"solve": {"kernel": kernel, "bound": "hbm",
                             "algorithmic_bytes_per_launch": algorithmic_bytes(n_max, M, order),
                             "traffic": pmc_traffic(n_max, M, order, kernel)[0],
                             "traffic_source": pmc_traffic(n_max, M, order, kernel)[1],
                             "launch": "hipGraph of the timed launches" if env.get("use_graph", True) else "eager",
                             "frac": algorithmic_bytes(n_max, M, order) / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS}},
    }


def end_to_end(ctx, wp, t, order, calls=60):
    """configs[1] through msnap_solve_batch with page-locked host arrays: H2D + kernel + D2H per call."""
    from drone_path_planning_python_amd import pinned_empty
    n, m, _ = wp.shape
    pwp, pt = pinned_empty(wp.shape), pinned_empty(t.shape)
    pwp[...] = wp
    pt[...] = t
    out = (pinned_empty((n, m - 1, 4, order + 1)), pinned_empty((n, m - 1)), pinned_empty((n,), np.int32))
    for _ in range(5):
        ctx.solve_batch(pwp, pt, out=out)
    t0 = time.perf_counter()
    for _ in range(calls):
        ctx.solve_batch(pwp, pt, out=out)
    per = (time.perf_counter() - t0) / calls
    assert not out[2].any()
    moved = wp.nbytes + t.nbytes + sum(a.nbytes for a in out)
    return {"value": n / per, "unit": "trajectories/s", "us_per_call": per * 1e6, "calls": calls,
            "host_link_GBps": moved / per / 1e9,
            "what": "msnap_solve_batch (host pointers, page-locked): upload, kernel, download, synchronise per call"}


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def resolve_world(args, argv):
    """The job's rank count from `--gpus` and the launcher environment -- they must agree.
      * WORLD_SIZE set (torchrun started this rank): that is the world; an explicit --gpus that differs is an error;
      * no WORLD_SIZE, --gpus N > 1: this process becomes the launcher of N ranks (see launch_ranks) and never
        touches the GPU itself;
      * neither: one rank.
    Returns (world, exit_code): exit_code is not None when this process is done (launcher or error)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if args.gpus is not None and args.gpus != world:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}: "
                             "refusing to report a job that is not the one asked for (start it as "
                             f"`python bench.py --gpus {args.gpus}` without a launcher environment, or let "
                             f"torch.distributed.run start {args.gpus} ranks)\n")
            return world, 2
        return world, None
    n = 1 if args.gpus is None else args.gpus
    if n < 1:
        sys.stderr.write(f"bench.py: --gpus {n}: need at least one rank\n")
        return n, 2
    if n == 1:
        return 1, None
    return n, launch_ranks(args, argv, n)


def launch_ranks(args, argv, n):
    """Start `python -m torch.distributed.run --nproc-per-node n ... bench.py <argv>` and wait for it.  Nothing in
    this process has called into HIP at this point (argparse only; torch.cuda.device_count() does not initialise the
    runtime on this image), so no process that holds the GPU is ever replaced or forked: the children are fresh
    interpreters, one per GPU.  Their stdout/stderr are this process's: rank 0's JSON line passes straight through."""
    import subprocess
    if args.backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} over RCCL needs {n} visible GPUs, this node shows {have}: not "
                             "starting (a rehearsal on fewer GPUs is `--backend gloo`, whose timings mean nothing)\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_core_count() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, world, rank, local_rank):
    """--launch-check: the process group of this job and one all-reduce over it, nothing else."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    if args.backend == "nccl":
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=device)
        one = torch.ones(1, dtype=torch.int64, device=device)
    else:
        dist.init_process_group(backend="gloo")
        one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    counted = int(one.item())
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": dist.get_world_size(),
                          "backend": dist.get_backend(), "ranks_counted_by_all_reduce": counted}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if counted == world else 3


def main():
    args = parse()
    world, done = resolve_world(args, sys.argv[1:])
    if done is not None:
        sys.exit(done)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        sys.exit(launch_check(args, world, rank, local_rank))
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_pg = world > 1 or args.force_pg
    if use_pg:
        if world == 1:      # --force-pg without a launcher: a one-rank group on this process
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)   # RCCL
        else:
            dist.init_process_group(backend="gloo")
    red_dev = device if args.backend == "nccl" else "cpu"

    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm

    M, order = args.segments, args.order
    ctx = Context(device_id=dev_index, order=order, max_segments=max(M, 64))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    wp, t = swarm(2, args.drones, M, shared_times=args.shared_times, rank=rank)
    batch = DeviceBatch(torch, ctx, wp, t, M, order, device)
    use_graph = not args.no_graph
    wall, dev_ms = timed_steps(torch, dist, batch, ctx, args.steps, args.warmup, use_graph, world)
    kname = ctx.last_kernel()       # the kernel instance libmsnap chose for this launch (msnap_last_kernel)
    assert int(batch.status.abs().sum().item()) == 0, "solve reported per-drone failures"
    head_err, _ = oracle_parity(batch.coef, wp, t, order, np.arange(args.drones))
    wall_max, dev_ms_max, head_err = max_over_ranks(torch, dist, [wall, dev_ms, head_err], red_dev, use_pg)

    # ---- BASELINE.json configs[2..4] on all ranks (sharded), with the RCCL all-gather timed ----
    configs = None
    strong = None
    if not args.no_configs:
        reps = args.config_reps or max(10, min(args.steps, 50))
        ctx7 = ctx if order == 7 else Context(device_id=dev_index, order=7, max_segments=64)
        ctx9 = ctx if order == 9 else Context(device_id=dev_index, order=9, max_segments=64)
        for c in (ctx7, ctx9):
            c.set_stream(torch.cuda.current_stream().cuda_stream)
        env = dict(torch=torch, dist=dist, ctx7=ctx7, ctx9=ctx9, device=device, rank=rank, world=world,
                   use_pg=use_pg, red_dev=red_dev, use_graph=use_graph,
                   coll=dist if args.backend == "nccl" else HostGather(dist))
        # ~15 ms of the same work first: the GPU's clocks settle over tens of milliseconds of load
        c2 = run_formation_config(2, env, reps, 40)
        c3 = run_formation_config(3, env, reps, 40)
        c4 = run_solve_config(env, 65536, 10, 9, reps, 40, "configs[4]")
        configs = {"2": c2, "3": c3, "4": c4}
        # STRONG scaling (total work fixed, sharded over the ranks): what a SCALE curve over N should be read from --
        # `value` itself is the config the metric is quoted on (256 drones per GPU, weak scaling, no collective)
        if not args.no_strong:
            s7 = run_solve_config(env, 65536, 10, 7, reps, 40, "65536 x 10, order 7")
            big = run_formation_config(2, env, max(5, reps // 5), 5, n_groups=2048)
            c2_all = None
            if world == 1:      # the one-GPU pass without its broad phase: the arithmetic the sharded pass divides
                ctx7.set_option("collide_no_cull", 1)
                try:
                    c2_all = run_formation_config(2, env, reps, 10)
                finally:
                    ctx7.set_option("collide_no_cull", 0)
            if rank == 0:
                def brief(c):
                    return {"workload": c["workload"], "sharding": c["sharding"], "value": c["value"], "unit": c["unit"],
                            "us_per_pipeline": c["us_per_pipeline"], "stage_us": c["stage_us"]}
                strong = {"n_gpus": world,
                          "note": "total work fixed and sharded by drone over the ranks; value = drones / max-over-ranks time; "
                                  "the formation pipelines include the collectives of the pairwise pass.  One GPU runs a "
                                  "whole swarm of 3072..16384 drones behind the exact broad phase of the pairwise pass "
                                  "(formation_4096x10; formation_4096x10_all_pairs is the same pipeline with it off).  "
                                  "Several ranks shard the solve and the sampler and then either divide the pairs (every "
                                  "pair on exactly one rank, a second all-gather, a fold) or -- where the broad phase "
                                  "leaves so few pairs that one GPU is quicker alone -- each run the whole gathered swarm "
                                  "and keep their rows (`sharding` says which): for these formation swarms the pairwise "
                                  "stage then no longer scales, it is 68 us on any number of ranks",
                          "solve_order9_65536x10": brief(c4), "solve_order7_65536x10": brief(s7),
                          "formation_4096x10": brief(c2), "formation_16384x10": brief(big)}
                if c2_all is not None:
                    strong["formation_4096x10_all_pairs"] = brief(c2_all)
        for c in (ctx7, ctx9):
            if c is not ctx:
                c.use_own_stream()
                c.close()

    sat = None
    if rank == 0 and not args.no_saturated:
        nbig = args.saturated_drones
        wpb, tb = swarm(2, 4096, M, shared_times=args.shared_times)
        reps = (nbig + 4095) // 4096
        wpb = np.tile(wpb, (reps, 1, 1))[:nbig]
        tb = tb if tb.ndim == 1 else np.tile(tb, (reps, 1))[:nbig]
        # the two ends of the batch get their own inputs, so a tile that read a neighbour's (or a stale
        # prefetch's) waypoints cannot pass as its periodic twin
        wpe, te = swarm(12, 8192, M, shared_times=args.shared_times)
        wpb[:4096], wpb[-4096:] = wpe[:4096], wpe[4096:]
        if tb.ndim == 2:
            tb[:4096], tb[-4096:] = te[:4096], te[4096:]
        big = DeviceBatch(torch, ctx, wpb, tb, M, order, device)
        # ~30 ms of sustained load before timing: the GPU's clocks settle over the first ~40 launches
        # (0.82 -> 0.68 ms per launch, tools/sat_ramp.py); the headline leg is warmed by its own replays
        ksat, wsat = 40, 40
        _, sat_ms = timed_steps(torch, dist, big, ctx, ksat, wsat, False, 1)
        sat_kernel = ctx.last_kernel()
        assert int(big.status.abs().sum().item()) == 0
        # parity of what was just timed (every wave walked several tiles): both ends against the C oracle
        ends = np.r_[0:4096, nbig - 4096:nbig]
        sat_err, sat_bad = oracle_parity(big.coef, wpb, tb, order, ends)
        per = sat_ms / ksat * 1e-3
        b = algorithmic_bytes(nbig, M, order)
        sat = {
            "workload": f"{nbig} drones x {M} segments, order {order}, one launch",
            "value": nbig / per, "unit": "trajectories/s", "ms_per_launch": per * 1e3,
            "max_norm_rel_err_vs_oracle": sat_err, "parity_sample": "first and last 4096 drones of the timed output",
            "roofline": {"bound": "hbm", "achieved": b / per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b / per / 1e9 / HBM_PEAK_GBS, "frac_of_copy_achievable": b / per / 1e9 / HBM_COPY_GBS,
                         "kernel": sat_kernel,
                         "traffic": pmc_traffic(nbig, M, order, sat_kernel)[0],
                         "traffic_source": pmc_traffic(nbig, M, order, sat_kernel)[1]},
        }
        assert sat_err <= 1e-6 and not sat_bad, f"saturated leg parity {sat_err:.3e}"
        del big

    # the reference's own usage pattern: every drone on one shared uniform grid -> K2
    # (operator built once by msnap_grid_prepare, then one fp64 MFMA GEMM per batch).
    # Reported beside the headline, never as `value`: the factorisation is outside the step.
    grid = None
    if rank == 0 and not args.no_shared_grid:
        wps, ts = swarm(2, args.drones, M, shared_times=True)
        ctx.prepare_grid(ts)
        gsmall = GridBatch(torch, ctx, wps, M, order, device)
        # (twice, the second pass is reported: this leg follows the saturated leg's host-side parity check and a 3 GB
        # free, and a 6 ms replay right after that idle stretch has been measured at 9 us per step instead of 2.9 on
        # clocks that had not come back up)
        timed_steps(torch, dist, gsmall, ctx, args.steps, args.warmup, use_graph, 1)
        _, g_ms = timed_steps(torch, dist, gsmall, ctx, args.steps, args.warmup, use_graph, 1)
        gk_small = ctx.last_kernel()
        assert int(gsmall.status.abs().sum().item()) == 0
        nbig = args.saturated_drones
        wpb = np.tile(wps, ((nbig + args.drones - 1) // args.drones, 1, 1))[:nbig]
        wpe, _ = swarm(13, 8192, M, shared_times=True)
        wpb[:4096], wpb[-4096:] = wpe[:4096], wpe[4096:]
        gbig = GridBatch(torch, ctx, wpb, M, order, device)
        _, gb_ms = timed_steps(torch, dist, gbig, ctx, 40, 40, False, 1)
        gk_big = ctx.last_kernel()
        g_err, g_bad = oracle_parity(gbig.coef, wpb, ts, order, np.r_[0:4096, nbig - 4096:nbig])
        assert g_err <= 1e-6 and not g_bad, f"shared-grid saturated leg parity {g_err:.3e}"
        per_s, per_b = g_ms / args.steps * 1e-3, gb_ms / 40 * 1e-3
        bs, bb = algorithmic_bytes(args.drones, M, order), algorithmic_bytes(nbig, M, order)
        grid = {
            "workload": f"shared uniform time grid t_i = i*10/(M+1) (scripts/drones_pols_generator.py:44-46), "
                        f"operator prepared once, K2 fp64 MFMA GEMM per step",
            "kernel": gk_small,
            "headline_shape": {"drones": args.drones, "value": args.drones / per_s, "us_per_step": per_s * 1e6,
                               "roofline_frac": bs / per_s / 1e9 / HBM_PEAK_GBS},
            "saturated": {"drones": nbig, "value": nbig / per_b, "ms_per_launch": per_b * 1e3,
                          "achieved_GBps": bb / per_b / 1e9, "roofline_frac": bb / per_b / 1e9 / HBM_PEAK_GBS,
                          "kernel": gk_big, "algorithmic_bytes_per_launch": bb,
                          "traffic": pmc_traffic(nbig, M, order, gk_big, prefix="grid")[0],
                          "traffic_source": pmc_traffic(nbig, M, order, gk_big, prefix="grid")[1],
                          "max_norm_rel_err_vs_oracle": g_err},
        }
        del gbig, gsmall

    conc = None
    if rank == 0 and world == 1 and use_graph and not args.no_concurrent:
        conc = {"what": "the headline shape as several independent batches in flight: one context and stream per "
                        "batch, parallel branches of one hipGraph, every branch its own buffers (results asserted "
                        "bit-identical across branches).  Not a second headline -- the contract's step is serial -- "
                        "but what separates the two parts of the serial step's 4.5 us: independent batches overlap "
                        "(256 drones are 32 wavefronts on 32 of 256 CUs), and the rate they reach is what the queue "
                        "can dispatch; the rest of the serial step is the kernel's own dependent latency",
                "by_streams": [concurrent_streams_leg(torch, Context, dev_index, device, wp, t, M, order, args.steps, k)
                               for k in (2, 4, 8)]}
        assert all(r["all_status_ok"] and r["branches_bit_identical"] for r in conc["by_streams"])

    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        ctx.use_own_stream()
        e2e = end_to_end(ctx, wp, t, order)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(M, order, args.cpu_seconds)

    if use_pg:
        dist.barrier()
    if rank == 0:
        total = args.drones * world * args.steps
        per_launch_s = dev_ms_max * 1e-3 / args.steps
        bytes_launch = algorithmic_bytes(args.drones, M, order)
        traffic, traffic_source = pmc_traffic(args.drones, M, order, kname)
        roof = {
            "bound": "hbm",
            "kernel": kname,
            "achieved": bytes_launch / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_launch / per_launch_s / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": bytes_launch,
            "avg_launch_us": per_launch_s * 1e6,
            "note": "256 drones are 788 KB: 0.1 us of HBM time under a launch floor of ~2.5 us, so this config "
                    "cannot exceed ~4 % by construction; configs[4] and `saturated` show the same kernel family "
                    "where the batch fills the chip",
        }
        if configs is not None:
            configs = {"1": {"workload": f"configs[1]: {args.drones} drones x {M} segments, order {order} "
                                         "(the headline; per GPU, weak scaling)",
                             "value": total / wall_max, "unit": "trajectories/s",
                             "stage_us": {"solve": per_launch_s * 1e6}, "max_norm_rel_err_vs_oracle": head_err,
                             "stages": {"solve": {"kernel": kname, "bound": "hbm (launch-bound at this size)",
                                                  "frac": roof["frac"]}}}, **configs}
        line = {
            "metric": METRIC,
            "value": total / wall_max,
            "unit": "trajectories/s",
            "n_gpus": world,
            # what the process group itself says (None: one rank without a process group) -- n_gpus is the same number
            # by construction (resolve_world), this is the evidence
            "rccl_ranks": dist.get_world_size() if use_pg else None,
            "backend": (("nccl (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend()) if use_pg else None),
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
            "max_norm_rel_err_vs_oracle": head_err,
            "roofline": roof,
            "cpu_baseline": cpu,
            "configs": configs,
            "strong_scaling": strong,
            "saturated": sat,
            "shared_grid": grid,
            "end_to_end": e2e,
            "concurrent_batches": conc,
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
