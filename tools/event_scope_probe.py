#!/usr/bin/env python3
"""Experiment: how long a cross-stream hand-over takes with torch's events (wait_stream) against HIP events created
with hipEventDisableTiming | hipEventReleaseToDevice.  Two streams ping-pong a tiny kernel; prints us per hand-over.
GPU box only."""
import ctypes
import sys
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")
DISABLE_TIMING, RELEASE_TO_DEVICE = 0x2, 0x40000000


def main():
    dev = torch.device("cuda", 0)
    a = torch.zeros(1 << 20, device=dev)
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    n = 2000

    def run(handover):
        for _ in range(50):
            handover()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            handover()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def torch_way():
        a.add_(1.0)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            a.add_(1.0)
        main_s.wait_stream(side)

    for flags, name in ((DISABLE_TIMING, "hip events, disable timing"), (DISABLE_TIMING | RELEASE_TO_DEVICE, "hip events, release to device")):
        e1, e2 = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e1), ctypes.c_uint(flags)) == 0
        assert hip.hipEventCreateWithFlags(ctypes.byref(e2), ctypes.c_uint(flags)) == 0
        ms, ss = ctypes.c_void_p(main_s.cuda_stream), ctypes.c_void_p(side.cuda_stream)

        def hip_way():
            a.add_(1.0)
            hip.hipEventRecord(e1, ms)
            hip.hipStreamWaitEvent(ss, e1, 0)
            with torch.cuda.stream(side):
                a.add_(1.0)
            hip.hipEventRecord(e2, ss)
            hip.hipStreamWaitEvent(ms, e2, 0)
        print(f"{name}: {run(hip_way):.2f} us per ping-pong (two kernels, two hand-overs)")
    print(f"torch wait_stream: {run(torch_way):.2f} us per ping-pong")

    def same_stream():
        a.add_(1.0)
        a.add_(1.0)
    print(f"same stream, no hand-over: {run(same_stream):.2f} us per two kernels")


if __name__ == "__main__":
    main()
