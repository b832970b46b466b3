import torch, time
torch.cuda.init()
n = 2_772_000_000 // 8
a = torch.empty(n, dtype=torch.float64, device='cuda')
b = torch.empty(n, dtype=torch.float64, device='cuda')
def t(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/k
ms = t(lambda: a.fill_(1.5))
print("fill  %.3f ms  %.2f TB/s write" % (ms, n*8/ms/1e9))
ms = t(lambda: b.copy_(a))
print("copy  %.3f ms  %.2f TB/s (r+w)" % (ms, 2*n*8/ms/1e9))
ms = t(lambda: torch.mul(a, 2.0, out=b))
print("mul   %.3f ms  %.2f TB/s (r+w)" % (ms, 2*n*8/ms/1e9))
