"""Duration / bandwidth of the time-axis resamplers (babe_resample) on the shapes the UNet calls them with (B = 1, one lane)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(os.environ.get("B", "1"))
Ns = [64, 96, 96, 128, 128, 256, 256]
Ts = [4096, 2048, 1024, 512, 256, 128, 64]
for i in range(6):
    C, F, T = Ns[i], 64 * (i + 1), Ts[i]
    x = torch.randn(B, C, F, T, device="cuda")
    for mode, Tout in ((0, T // 2), (2, T)):
        src = x if mode == 0 else torch.randn(B, C, F, T // 2, device="cuda")
        out = torch.empty(B, C, F, Tout, device="cuda")
        us = t(lambda: ops.resample(src, out, mode))
        by = (src.numel() + out.numel()) * 4
        print(f"mode {mode} C={C:3d} F={F:3d} T={T:4d}: {us:7.1f} us {by/us/1e3:7.0f} GB/s")
    # up (mode 1) and its adjoint (mode 3) on the decoder side: input [B, C, F - 64 ...] approximated by the same plane
    src = torch.randn(B, C, F, T // 2, device="cuda")
    out = torch.empty(B, C, F, T, device="cuda")
    us = t(lambda: ops.resample(src, out, 1)); by = (src.numel() + out.numel()) * 4
    print(f"mode 1 C={C:3d} F={F:3d} T={T//2:4d}->{T}: {us:7.1f} us {by/us/1e3:7.0f} GB/s")
    src2 = torch.randn(B, C, F, T, device="cuda"); out2 = torch.empty(B, C, F, T // 2, device="cuda")
    us = t(lambda: ops.resample(src2, out2, 3)); by = (src2.numel() + out2.numel()) * 4
    print(f"mode 3 C={C:3d} F={F:3d} T={T}->{T//2:4d}: {us:7.1f} us {by/us/1e3:7.0f} GB/s")
