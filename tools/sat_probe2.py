"""Saturated solve launch time under different launch modes (null stream, side stream, hipGraph)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from drone_path_planning_python_amd import Context
from drone_path_planning_python_amd.synthetic import swarm
N = 1 << 20; M = 10
wp, t = swarm(2, 4096, M); wp = np.tile(wp, (256, 1, 1)); t = np.tile(t, (256, 1))
tw = torch.from_numpy(wp).cuda(); tt = torch.from_numpy(t).cuda()
coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device="cuda")
dur = torch.empty((N, M), dtype=torch.float64, device="cuda"); st = torch.empty((N,), dtype=torch.int32, device="cuda")
ctx = Context(0, 7, 64)

def run(label, stream, graph=False, own=False):
    with torch.cuda.stream(stream):
        if own:
            ctx.use_own_stream()
        else:
            ctx.set_stream(stream.cuda_stream)
        for _ in range(3): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
        ctx.sync(); torch.cuda.synchronize()
        g = None
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                for _ in range(20): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
            ctx.set_stream(stream.cuda_stream)
            g.replay(); torch.cuda.synchronize()
        if own:
            ctx.timer_start()
            for _ in range(20): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
            ms = ctx.timer_stop()
        else:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if g: g.replay()
            else:
                for _ in range(20): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
            e1.record(stream); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
    print("%-34s %.4f ms/launch" % (label, ms / 20))

run("null stream", torch.cuda.default_stream())
side = torch.cuda.Stream()
run("side stream", side)
run("context's own stream", side, own=True)
run("hipGraph on side stream", side, graph=True)
run("null stream again", torch.cuda.default_stream())
