#!/usr/bin/env python3
"""What the driver's `bench.py --steps 20` timed region is made of besides the 20 kernels.

The headline step is 4.5 us under a 2000-step graph and 6.0 us under the driver's 20-step graph: one replay's fixed
cost (host launch, start of the first node, completion wake-up of the synchronize) is divided by 20 instead of 2000.
This probe times exactly bench.timed_steps' region (same DeviceBatch, same capture) in fresh child processes, one per
runtime setting, several repetitions each:

    python tools/replay_overhead.py [--steps 20] [--reps 15]

Child settings are environment variables of the HIP / ROCr runtime (process-local; nothing on the machine changes).
Prints per setting: median / min wall per step, the HIP-event time per step, and wall - events (the host share).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SETTINGS = [
    ("default", {}),
    ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}),
    ("ROC_ACTIVE_WAIT_TIMEOUT=1000", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
    ("both", {"HSA_ENABLE_INTERRUPT": "0", "ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
    ("spin flag (hipDeviceScheduleSpin)", {"MSNAP_PROBE_SPIN": "1"}),
    ("DEBUG_CLR_GRAPH_PACKET_CAPTURE=0", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}),
    ("DEBUG_CLR_GRAPH_PACKET_CAPTURE=1", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}),
    ("no events in the region", {"MSNAP_PROBE_NO_EVENTS": "1"}),
    ("eager (no graph)", {"MSNAP_PROBE_EAGER": "1"}),
]


def child(steps, reps):
    import statistics
    import time
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from drone_path_planning_python_amd import Context
    from drone_path_planning_python_amd.synthetic import swarm
    if os.environ.get("MSNAP_PROBE_SPIN"):
        import ctypes
        path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
        rc = ctypes.CDLL(path).hipSetDeviceFlags(ctypes.c_uint(1))      # hipDeviceScheduleSpin
        sys.stderr.write(f"hipSetDeviceFlags(hipDeviceScheduleSpin) -> {rc} ({path})\n")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    ctx = Context(device_id=0, order=7, max_segments=64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    wp, t = swarm(2, 256, 10, shared_times=False, rank=0)
    batch = bench.DeviceBatch(torch, ctx, wp, t, 10, 7, device)
    stream = torch.cuda.current_stream()
    eager = bool(os.environ.get("MSNAP_PROBE_EAGER"))
    no_events = bool(os.environ.get("MSNAP_PROBE_NO_EVENTS"))
    graph = None
    for _ in range(3):
        batch.step()
    torch.cuda.synchronize()
    if not eager:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(steps):
                batch.step()
        ctx.set_stream(stream.cuda_stream)
        torch.cuda.synchronize()
        graph.replay()
    walls, devs = [], []
    for _ in range(reps):
        for _ in range(5):
            batch.step()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not no_events:
            ev0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                batch.step()
        if not no_events:
            ev1.record(stream)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e6)
        devs.append(ev0.elapsed_time(ev1) * 1e3 if not no_events else float("nan"))
    assert int(batch.status.abs().sum().item()) == 0
    out = {"wall_us_median": statistics.median(walls), "wall_us_min": min(walls), "wall_us_first": walls[0],
           "events_us_median": statistics.median(devs), "steps": steps}
    print(json.dumps(out), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.steps, a.reps)
    print(f"{'setting':42s} {'first':>8s} {'median':>8s} {'min':>8s} {'events':>8s} {'host':>7s}   us per {a.steps}-step region "
          f"(per step: median / {a.steps})")
    for name, env in SETTINGS:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(a.steps), "--reps",
                            str(a.reps)], env=e, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{name:42s} failed rc={r.returncode}: {r.stderr.strip()[-200:]}", flush=True)
            continue
        d = json.loads(line[-1])
        print(f"{name:42s} {d['wall_us_first']:8.1f} {d['wall_us_median']:8.1f} {d['wall_us_min']:8.1f} "
              f"{d['events_us_median']:8.1f} {d['wall_us_median'] - d['events_us_median']:7.1f}   "
              f"{d['wall_us_median'] / a.steps:.2f} us/step", flush=True)


if __name__ == "__main__":
    main()
