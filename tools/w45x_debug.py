"""Where does the wide nested-Winograd kernel differ from the 64-channel-tile kernel?  (debug aid)"""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
for Cout, Cin, Fq, T, dil in ((128, 128, 48, 128, 2), (128, 16, 8, 64, 1), (128, 32, 8, 64, 1), (128, 64, 8, 64, 1), (192, 96, 32, 256, 1)):
    g = torch.Generator().manual_seed(Cout + Cin)
    x = torch.randn(2, Cin, Fq, T, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)).cuda()
    out = torch.empty(2, Cout, Fq, T, device="cuda")
    ops.conv2d(x, ops.PackedConv(w), out, dil=dil, force_nested=True)
    parts = torch.empty_like(out)
    for c0 in range(0, Cout, 64):
        ops.conv2d(x, ops.PackedConv(w[c0:c0 + 64].contiguous()), parts[:, c0:c0 + 64], dil=dil, force_nested=True)
    d = (out - parts).abs()
    print(f"Cout={Cout} Cin={Cin} F={Fq} T={T} dil={dil}: max {float(d.max()):.3e}  bad elements {int((d > 1e-5).sum())} / {d.numel()}")
    if float(d.max()) > 1e-5:
        bad = d > 1e-5
        print("  per batch:", bad.flatten(1).sum(1).tolist())
        print("  per 16-channel tile:", bad.view(2, Cout // 16, 16, Fq, T).sum((0, 2, 3, 4)).tolist())
        print("  per row:", bad.sum((0, 1, 3)).tolist())
        print("  per 16-step block:", bad.view(2, Cout, Fq, T // 16, 16).sum((0, 1, 2, 4)).tolist())
        print("  per t%4:", bad.view(2, Cout, Fq, T // 4, 4).sum((0, 1, 2, 3)).tolist())
