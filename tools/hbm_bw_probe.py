import torch
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    w = t(lambda: x.fill_(1.0)); r = t(lambda: x.sum()); c = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB: write {mb*1.048576e-3/w:7.0f} GB/s | read {mb*1.048576e-3/r:7.0f} GB/s | copy (r+w) {2*mb*1.048576e-3/c:7.0f} GB/s")
