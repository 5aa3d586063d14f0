"""Per-tile cost of the pipelined F(4,3) kernel: time vs number of tiles per CU (128 ch, T=512, F varies), and vs the number of
K-slabs per tile (Cin varies at fixed tile count) -> per-tile fixed overhead (prologue + epilogue) and per-slab time."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
def t(Cin, Cout, F, T, dil=1, B=1, n=10):
    x = torch.randn(B, Cin, F, T, device="cuda")
    w = torch.randn(Cout, Cin, 5, 3, device="cuda") / math.sqrt(Cin * 15)
    pc = ops.PackedConv(w, os.environ.get("PRECISION", "f32"))
    out = torch.empty(B, Cout, F, T, device="cuda")
    kw = dict(dil=dil)
    if os.environ.get("RES", "1") == "1":          # the forward layers' epilogue: out = (res + gate * conv) / sqrt2
        kw.update(res=torch.randn(B, Cout, F, T, device="cuda"), oscale=torch.randn(B, Cout, device="cuda"), alpha=0.7, rbeta=0.7)
    for _ in range(3): ops.conv2d(x, pc, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv2d(x, pc, out, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("tiles per CU sweep (Cin=Cout=128, T=512, dil=8: all 5 taps valid for most rows)")
for F in (128, 256, 512, 1024):
    us = t(128, 128, F, 512, dil=8)
    tiles = F * 512 // 256
    print(f"  F={F:5d}: {tiles:5d} tiles ({tiles/256:.0f}/CU) {us:8.1f} us  -> {us/(tiles/256):7.1f} us per tile-round")
print("K sweep (Cout=128, F=256, T=512 = 2 tiles per CU)")
for Cin in (64, 128, 256, 512):
    us = t(Cin, 128, 256, 512, dil=8)
    print(f"  Cin={Cin:4d}: {Cin//8*5:4d} slabs/tile {us:8.1f} us -> {us/2:7.1f} us per tile")
