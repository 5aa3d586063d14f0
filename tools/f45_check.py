"""Experimental F(4,5) x F(4,3) conv kernel (csrc/conv_wino85.hip) against the float64 direct convolution and against the production
nested kernel: accuracy (forward with the fused epilogue, input-VJP with a per-channel input scale) and stand-alone timing."""
import math, os, sys, time
os.environ["BABE_CONV_F45"] = "1"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from babe_amd import ops
from babe_amd._lib import dispatch_counts
from oracle import unet as UN

def rel(a, b):
    a, b = a.detach().double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

bad = 0
for (B, Cin, Cout, Fq, T, dil) in [(1, 128, 128, 48, 128, 2), (1, 256, 256, 28, 64, 4), (2, 128, 256, 40, 100, 1), (1, 96, 128, 56, 64, 8),
                                   (1, 128, 128, 24, 192, 16), (1, 256, 128, 448, 64, 64), (1, 96, 96, 40, 128, 2), (2, 128, 192, 20, 68, 1),
                                   (1, 64, 64, 24, 128, 1), (2, 32, 64, 36, 100, 3), (1, 96, 64, 20, 64, 4)]:
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + dil)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    ref = UN.conv_same(x.double(), w.double(), dil)
    pc = ops.PackedConv(w.cuda())
    assert pc.fwd_wino85 is not None
    out = torch.empty(B, Cout, Fq, T, device="cuda")
    ops.conv2d(x.cuda(), pc, out, dil=dil, force_f45=True)
    e0 = rel(out, ref)
    out45 = torch.empty_like(out)
    os.environ["X"] = "1"
    pc45 = pc
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    out2 = res.cuda().clone()
    ops.conv2d(x.cuda(), pc, out2, dil=dil, res=out2, oscale=osc.cuda(), alpha=0.7, rbeta=0.3, force_f45=True)
    e1 = rel(out2, 0.7 * ref * osc[:, :, None, None].double() + 0.3 * res.double())
    e2 = float("nan")
    if pc.bwd_wino85 is not None:
        gy = torch.randn(B, Cout, Fq, T, generator=g)
        isc = torch.randn(B, Cout, generator=g)
        xr = x.double().requires_grad_(True)
        y = UN.conv_same(xr, w.double(), dil)
        gref, = torch.autograd.grad((y * (gy * isc[:, :, None, None]).double()).sum(), xr)
        gx = torch.empty(B, Cin, Fq, T, device="cuda")
        ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda(), force_f45=True)
        e2 = rel(gx, gref)
    ok = e0 < 2e-5 and e1 < 2e-5 and not (e2 > 2e-5)
    bad += int(not ok)
    print(f"F45 Cin={Cin} Cout={Cout} F={Fq} T={T} dil={dil} B={B}: fwd {e0:.2e}, epilogue {e1:.2e}, vjp {e2:.2e} {'OK' if ok else 'BAD'}", flush=True)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print("timing, forward, us (F45 / production nested kernel), algorithmic TFLOP/s")
for name, C, Fq, T, dil in [("enc0", 64, 64, 4096, 2), ("dec1", 64, 128, 2048, 4), ("enc1", 96, 128, 2048, 2), ("enc2", 96, 192, 1024, 4), ("dec3", 96, 256, 512, 8), ("enc3", 128, 256, 512, 4), ("enc4", 128, 320, 256, 8), ("dec5", 128, 384, 128, 8), ("enc5", 256, 384, 128, 8), ("enc6", 256, 448, 64, 8),
                            ("enc6.d64", 256, 448, 64, 64), ("enc5.d64", 256, 384, 128, 64)]:
    for B in (1, 2):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, C, Fq, T, generator=g).cuda()
        w = (torch.randn(C, C, 5, 3, generator=g) / math.sqrt(C * 15)).cuda()
        pc = ops.PackedConv(w)
        out = torch.empty(B, C, Fq, T, device="cuda")
        t85 = bench(lambda: ops.conv2d(x, pc, out, dil=dil, force_f45=True))
        keep = pc.fwd_wino85
        pc.fwd_wino85 = None
        t45 = bench(lambda: ops.conv2d(x, pc, out, dil=dil))
        pc.fwd_wino85 = keep
        fl = 2.0 * B * C * C * 15 * Fq * T
        print(f"{name:9s} B={B} C={C} F={Fq} T={T} dil={dil}: {t85:7.1f} / {t45:7.1f} us   {fl / t85 * 1e-6:6.1f} / {fl / t45 * 1e-6:6.1f}   x{t45 / t85:.2f}", flush=True)
print("bad:", bad)
