import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from drone_path_planning_python_amd import Context
from drone_path_planning_python_amd.synthetic import swarm
N = 1 << 20; M = 10
ctx = Context(0, 7, 64); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
wp, _ = swarm(2, 4096, M); wp = np.tile(wp, (N // 4096, 1, 1))
ctx.prepare_grid(np.arange(M + 1) * 10.0 / (M + 1))
tw = torch.from_numpy(wp).cuda()
coef = torch.empty((N, M, 4, 8), dtype=torch.float64, device="cuda")
dur = torch.empty((N, M), dtype=torch.float64, device="cuda"); st = torch.empty((N,), dtype=torch.int32, device="cuda")
for _ in range(40): ctx.solve_grid_device(N, tw, coef, dur, st)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): ctx.solve_grid_device(N, tw, coef, dur, st)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], "K2 ms/launch %.4f" % (e0.elapsed_time(e1) / 30))
