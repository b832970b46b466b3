import sys, numpy as np, torch
sys.path.insert(0, ".")
from drone_path_planning_python_amd import Context
from drone_path_planning_python_amd.synthetic import swarm
import os
ORDER = int(os.environ.get("PROBE_ORDER", "7")); M = int(os.environ.get("PROBE_M", "10")); N = int(os.environ.get("PROBE_N", str(1 << 20)))
ctx = Context(0, ORDER, 64); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
wp, t = swarm(2, 4096, M); wp = np.tile(wp, (N // 4096, 1, 1)); t = np.tile(t, (N // 4096, 1))
tw = torch.from_numpy(wp).cuda(); tt = torch.from_numpy(t).cuda()
coef = torch.empty((N, M, 4, ORDER + 1), dtype=torch.float64, device="cuda")
dur = torch.empty((N, M), dtype=torch.float64, device="cuda"); st = torch.empty((N,), dtype=torch.int32, device="cuda")
for _ in range(40): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ctx.solve_batch_device(N, M, tw, tt, False, coef, dur, st)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/launch", e0.elapsed_time(e1) / 20)
