"""Do the conv kernels read (and use) memory outside their input tensor?  The input lives inside a larger buffer whose surroundings
are filled once with zeros and once with huge values; the outputs must be identical."""
import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd import ops
torch.manual_seed(0)
def probe(prec, Cin, Cout, F, T, kh, dil, kind, B=2):
    kw = 3 if kh == 5 else 1
    w = torch.randn(Cout, Cin, kh, kw, device="cuda") / math.sqrt(Cin * kh * kw)
    pc = ops.PackedConv(w, prec)
    Ci, Co = (Cout, Cin) if kind == "vjp" else (Cin, Cout)
    n = B * Ci * F * T
    pad = 1 << 20
    xin = torch.randn(n, device="cuda")
    outs = []
    for fill in (0.0, 3e30, float("nan")):
        big = torch.full((n + 2 * pad,), fill, device="cuda")
        big[pad:pad + n] = xin
        x = big[pad:pad + n].view(B, Ci, F, T)
        out = torch.empty(B, Co, F, T, device="cuda")
        if kind == "units":
            scale = torch.ones(B, Ci, device="cuda")
            nu = ops.lib().babe_units_size(Ci, F, T) * 8 * B
            aubig = torch.full((nu + 2 * pad,), 0x7fc0 if fill != fill else (0x7f7f if fill else 0), dtype=torch.int16, device="cuda")
            au = aubig[pad:pad + nu]
            ops.scale_gelu_units(x, scale, au)
            ops.conv2d_units(au, pc, out, Ci, dil=dil)
        elif kind == "vjp":
            ops.conv2d(x, pc, out, dil=dil, transpose=True, in_scale=torch.ones(B, Ci, device="cuda"))
        else:
            ops.conv2d(x, pc, out, dil=dil)
        torch.cuda.synchronize()
        outs.append(out)
    d1 = float((outs[1] - outs[0]).abs().max()); nan = int(torch.isnan(outs[2]).sum())
    flag = "  <-- OUT-OF-BOUNDS DATA USED" if (d1 > 0 or nan > 0 or not math.isfinite(d1)) else ""
    print(f"{prec:6s} {kind:5s} k={kh}x{kw} Cin={Cin} Cout={Cout} F={F} T={T} dil={dil}: max diff (3e30 surroundings) {d1:.3e}, NaNs with NaN surroundings {nan}{flag}", flush=True)
shapes53 = [(64, 64, 64, 512, 1), (128, 128, 256, 64, 4), (256, 256, 448, 8, 1), (256, 256, 384, 16, 64), (96, 96, 128, 256, 2), (128, 128, 320, 32, 32), (64, 64, 64, 4096, 2)]
for prec in ("bf16", "bf16x3", "f32"):
    for (ci, co, F, T, d) in shapes53:
        for kind in (("fwd", "vjp", "units") if prec == "bf16" else ("fwd", "vjp")):
            if kind == "units" and (T % 4 or ci % 8): continue
            probe(prec, ci, co, F, T, 5, d, kind)
    for (ci, co, F, T) in [(128, 64, 64, 512), (512, 256, 448, 8), (96, 96, 192, 128), (64, 2, 64, 4096), (2, 64, 64, 4096), (256, 96, 256, 512)]:
        for kind in ("fwd", "vjp"):
            probe(prec, ci, co, F, T, 1, 1, kind)
