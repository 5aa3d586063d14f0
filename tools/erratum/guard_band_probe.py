import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd import ops
torch.manual_seed(0)
PAD = 1 << 18
def guarded(shape):
    n = 1
    for s_ in shape: n *= s_
    big = torch.full((n + 2 * PAD,), 777.0, device="cuda")
    return big, big[PAD:PAD + n].view(*shape)
def check(big, n, name):
    lo = big[:PAD]; hi = big[PAD + n:]
    nlo = int((lo != 777.0).sum()); nhi = int((hi != 777.0).sum())
    if nlo or nhi:
        ih = torch.nonzero(hi != 777.0).flatten()
        il = torch.nonzero(lo != 777.0).flatten()
        print(f"  !! {name}: {nlo} elements BEFORE and {nhi} AFTER the output tensor were overwritten; first after-offset {int(ih[0]) if nhi else None}, last {int(ih[-1]) if nhi else None}; before-offsets {il[:3].tolist() if nlo else None}", flush=True)
    return nlo + nhi
def run(prec, Cin, Cout, F, T, kh, dil, kind, B=2):
    kw = 3 if kh == 5 else 1
    w = torch.randn(Cout, Cin, kh, kw, device="cuda") / math.sqrt(Cin * kh * kw); pc = ops.PackedConv(w, prec)
    Ci, Co = (Cout, Cin) if kind == "vjp" else (Cin, Cout)
    x = torch.randn(B, Ci, F, T, device="cuda")
    big, out = guarded((B, Co, F, T))
    if kind == "units":
        au = torch.empty(ops.lib().babe_units_size(Ci, F, T) * 8 * B, dtype=torch.int16, device="cuda")
        ops.scale_gelu_units(x, torch.ones(B, Ci, device="cuda"), au); ops.conv2d_units(au, pc, out, Ci, dil=dil)
    elif kind == "vjp": ops.conv2d(x, pc, out, dil=dil, transpose=True, in_scale=torch.ones(B, Ci, device="cuda"))
    else: ops.conv2d(x, pc, out, dil=dil)
    torch.cuda.synchronize()
    bad = check(big, out.numel(), f"{prec} {kind} k={kh} Cin={Cin} Cout={Cout} F={F} T={T} dil={dil}")
    print(f"{prec:6s} {kind:5s} k={kh}x{kw} Cin={Cin} Cout={Cout} F={F} T={T} dil={dil}: {'OK' if not bad else 'OUT-OF-BOUNDS WRITES'}", flush=True)
for prec in ("bf16", "bf16x3", "f32"):
    for (ci, co, F, T, d) in [(256, 256, 448, 64, 2), (64, 64, 64, 4096, 1), (128, 128, 256, 512, 4), (96, 96, 192, 1024, 4), (128, 128, 320, 256, 32), (256, 256, 384, 128, 64), (64, 64, 128, 2048, 2)]:
        for kind in (("fwd", "vjp", "units") if prec == "bf16" else ("fwd", "vjp")):
            run(prec, ci, co, F, T, 5, d, kind)
    for (ci, co, F, T) in [(512, 256, 448, 64), (128, 64, 64, 4096), (64, 2, 64, 4096), (2, 64, 64, 4096), (192, 64, 128, 2048)]:
        for kind in ("fwd", "vjp"): run(prec, ci, co, F, T, 1, 1, kind)
