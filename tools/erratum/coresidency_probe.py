import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd import ops
from babe_amd import cqt as _cq
_cq._register_sigs()
from babe_amd.cqt import RealFFT
torch.manual_seed(0)
L = 368368
fft = RealFFT(L, torch.device("cuda"))
x = torch.randn(2, L, device="cuda")
w = torch.randn(256, 2, 5, 3, device="cuda") / math.sqrt(30); pcw = ops.PackedConv(w)
gy = torch.randn(2, 256, 448, 64, device="cuda")
def fewco():
    out = torch.empty(2, 2, 448, 64, device="cuda"); ops.conv2d(gy, pcw, out, transpose=True, alpha=0.7); return out
victims = {"rfft": lambda: fft.rfft(x), "fewco": fewco}
def mkconv(prec, Cin, Cout, F, T, kh, dil, kind, B=2):
    kw = 3 if kh == 5 else 1
    ww = torch.randn(Cout, Cin, kh, kw, device="cuda") / math.sqrt(Cin * kh * kw); pc = ops.PackedConv(ww, prec)
    xx = torch.randn(B, Cin, F, T, device="cuda"); out = torch.empty(B, Cout, F, T, device="cuda")
    scale = torch.ones(B, Cin, device="cuda")
    au = torch.empty(ops.lib().babe_units_size(Cin, F, T) * 8 * B, dtype=torch.int16, device="cuda") if kind in ("units", "sgu") else None
    def f():
        if kind == "units": ops.scale_gelu_units(xx, scale, au); ops.conv2d_units(au, pc, out, Cin, dil=dil)
        elif kind == "sgu": ops.scale_gelu_units(xx, scale, au if au is not None else None)
        elif kind == "vjp": ops.conv2d(out, pc, xx, dil=dil, transpose=True, in_scale=torch.ones(B, Cout, device="cuda"))
        else: ops.conv2d(xx, pc, out, dil=dil)
    return f
partners = {
    "bf16p units 256ch": mkconv("bf16", 256, 256, 448, 64, 5, 2, "units"),
    "bf16p fwd 256ch": mkconv("bf16", 256, 256, 448, 64, 5, 2, "fwd"),
    "bf16p vjp 256ch": mkconv("bf16", 256, 256, 448, 64, 5, 2, "vjp"),
    "bf16p units 64ch": mkconv("bf16", 64, 64, 64, 4096, 5, 1, "units"),
    "sgu only 64ch": mkconv("bf16", 64, 64, 64, 4096, 5, 1, "sgu"),
    "bf16p fwd 64ch": mkconv("bf16", 64, 64, 64, 4096, 5, 1, "fwd"),
    "bf16p units 128ch": mkconv("bf16", 128, 128, 256, 512, 5, 4, "units"),
    "bf16p 1x1 512->256": mkconv("bf16", 512, 256, 448, 64, 1, 1, "fwd"),
    "bf16p units 96ch": mkconv("bf16", 96, 96, 192, 1024, 5, 4, "units"),
    "f32 wino45 256ch": mkconv("f32", 256, 256, 448, 64, 5, 2, "fwd"),
    "f32 conv11p 512->256": mkconv("f32", 512, 256, 448, 64, 1, 1, "fwd"),
}
sel = os.environ.get("PARTNERS")
if sel:
    partners = {k: v for k, v in partners.items() if any(t in k for t in sel.split(","))}
vsel = os.environ.get("VICTIMS")
if vsel:
    victims = {k: v for k, v in victims.items() if k in vsel.split(",")}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for vn, vf in victims.items():
    ref = vf().clone(); torch.cuda.synchronize()
    for pn, pf in partners.items():
        outs = []
        for i in range(30):
            with torch.cuda.stream(sB): pf()
            with torch.cuda.stream(sA): outs.append(vf())
            with torch.cuda.stream(sB): pf()
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o, ref)) for o in outs)
        print(f"victim {vn:6s} beside {pn:22s}: {bad} of 30 runs differ", flush=True)
