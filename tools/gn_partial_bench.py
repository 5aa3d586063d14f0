import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
from babe_amd._lib import lib, ptr, stream, check
for (B, C, F, T) in [(1, 64, 64, 4096), (1, 128, 256, 512), (1, 256, 448, 64)]:
    x = torch.randn(B, C, F, T, device="cuda"); G = 8
    n = (C // G) * F * T; S = ops._splits(n, B, G)
    part = torch.empty(B * G * S * 2, device="cuda", dtype=torch.float64)
    f = lambda: check(lib().babe_gn_partial(ptr(x), ptr(part), B, G, n, S, stream()), "p")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    da = torch.randn_like(x); sc = torch.rand(B, C, device="cuda") + 0.5
    part2 = torch.empty(B * G * S, device="cuda", dtype=torch.float64)
    f2 = lambda: check(lib().babe_gn_bwd_partial(ptr(x), ptr(da), ptr(sc), ptr(part2), B, C, G, F * T, S, stream()), "bp")
    for _ in range(3): f2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): f2()
    e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) / 50 * 1e3
    print((B, C, F, T), f"S={S}: gn_partial {us:.1f} us {x.numel()*4/us/1e3:.0f} GB/s | gn_bwd_partial {us2:.1f} us {x.numel()*8/us2/1e3:.0f} GB/s")
