import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from babe_amd import ops
from oracle import unet as UN
def run(B, Cin, Cout, Fq, T, dil, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    rb = lambda t: t.to(torch.bfloat16).double()
    ref = UN.conv_same(rb(x), rb(w), dil)
    pc = ops.PackedConv(w.cuda(), "bf16")
    out = torch.empty(B, Cout, Fq, T, device="cuda")
    ops.conv2d(x.cuda(), pc, out, dil=dil)
    err = (out.double().cpu() - ref).abs()
    print((B, Cin, Cout, Fq, T, dil), "max err", float(err.max()), "ref max", float(ref.abs().max()))
    if err.max() > 1e-4:
        print(" err by co block of 32:", [round(float(err[:, c:c+32].max()), 4) for c in range(0, Cout, 32)])
        print(" err by row:", [round(float(err[:, :, f].max()), 3) for f in range(Fq)][:64])
        print(" err by t (first 72):", [round(float(err[:, :, :, t].max()), 3) for t in range(min(T, 72))])
        # which taps/channels are missing: probe with delta weights
for c in [(1, 96, 96, 128, 64, 64), (1, 96, 96, 128, 64, 1), (1, 64, 64, 16, 64, 1), (1, 64, 64, 16, 512, 1), (1, 32, 64, 8, 512, 1), (1, 128, 128, 8, 512, 1)]:
    run(*c)
