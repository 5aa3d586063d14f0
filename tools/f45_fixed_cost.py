"""Fixed cost per workgroup of the F(4,5) x F(4,3) kernel: the same geometry (F = 256, T = 512, dil 4: 512 workgroups = 2 rounds on 256
CUs; and B = 2: 4 rounds) with 32 .. 512 input channels = 4 .. 64 super-slabs; time = a + b * super-slabs."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from babe_amd import ops

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for Cout in (128, 64):
    for B in (1, 2):
        pts = []
        for Cin in (32, 64, 128, 256, 512):
            g = torch.Generator().manual_seed(1)
            x = torch.randn(B, Cin, 256, 512, generator=g).cuda()
            pc = ops.PackedConv((torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)).cuda())
            out = torch.empty(B, Cout, 256, 512, device="cuda")
            t = bench(lambda: ops.conv2d(x, pc, out, dil=4, force_f45=True))
            pts.append((2 * Cin // 16, t))
            print(f"Cout={Cout} B={B} Cin={Cin:3d} ({2 * Cin // 16:2d} super-slabs): {t:7.1f} us", flush=True)
        n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
        sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        rounds = 2 * B
        print(f"  fit: {a:.1f} us + {b:.2f} us per super-slab  ->  per workgroup round ({rounds} rounds): fixed {a / rounds:.1f} us, {b / rounds:.2f} us per super-slab")
