import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, C, F, T) in [(1, 64, 64, 4096), (1, 128, 256, 512), (1, 256, 448, 64), (2, 128, 256, 512)]:
    x = torch.randn(B, C, F, T, device="cuda"); da = torch.randn_like(x); gy = torch.randn_like(x)
    gamma = torch.rand(C, device="cuda") + 0.5; film = torch.randn(B, C, device="cuda")
    stats, scale = ops.gn_scale(x, gamma, film)
    a = torch.empty_like(x); gx = torch.empty_like(x)
    n = x.numel() * 4 / 1e6
    us = t(lambda: ops.scale_gelu(x, scale, a)); print(f"{(B,C,F,T)} scale_gelu {us:6.1f} us {2*n/us*1e3:6.0f} GB/s", end=" | ")
    us = t(lambda: ops.gn_bwd(x, da, gy, scale, stats, gx, 0.7)); print(f"gn_bwd(partial+apply) {us:6.1f} us {6*n/us*1e3:6.0f} GB/s", end=" | ")
    us = t(lambda: ops.gn_scale(x, gamma, film)); print(f"gn_scale {us:6.1f} us {n/us*1e3:6.0f} GB/s")
