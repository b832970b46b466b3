"""How the saturated solve's launch time settles with sustained load (clock / power ramp)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from drone_path_planning_python_amd import Context
from drone_path_planning_python_amd.synthetic import swarm
N = 1 << 20; M = 10
wp, t = swarm(2, 4096, M); wp = np.tile(wp, (256, 1, 1)); t = np.tile(t, (256, 1))
tw = torch.from_numpy(wp).cuda(); tt = torch.from_numpy(t).cuda()
coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device="cuda")
dur = torch.empty((N, M), dtype=torch.float64, device="cuda"); st = torch.empty((N,), dtype=torch.int32, device="cuda")
ctx = Context(0, 7, 64); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for b in range(40):
    for _ in range(10): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
    evs[b + 1].record()
torch.cuda.synchronize()
print("ms/launch per block of 10:", " ".join("%.3f" % (evs[b].elapsed_time(evs[b + 1]) / 10) for b in range(40)))
