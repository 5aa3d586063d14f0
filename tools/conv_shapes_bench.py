"""Per-shape throughput of babe_conv2d for every conv shape of the 44.1 kHz CQTDiff+ UNet (B=2 segments)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops

B = int(os.environ.get("B", "2"))
Ns = [64, 96, 96, 128, 128, 256, 256]
nd = [2, 3, 4, 5, 6, 7, 7]
shapes = []   # (name, Cin, Cout, F, T, KH, KW, dil, count)
for i in range(7):
    F, T = 64 * (i + 1), 4096 >> i
    N = Ns[i]
    for d in range(nd[i]):
        shapes.append((f"enc{i}.H{d}", N, N, F, T, 5, 3, 2 ** d, 1))
    shapes.append((f"enc{i}.proj_in", Ns[max(i - 1, 0)], N, F, T, 1, 1, 1, 1))
    shapes.append((f"enc{i}.pyr", 2, N, F, T if i == 6 else T // 2, 5, 3, 1, 1))
for i in range(7):
    F, T = 64 * (i + 1), 4096 >> i
    N = Ns[max(i - 1, 0)]
    for d in range(nd[i]):
        shapes.append((f"dec{i}.H{d}", N, N, F, T, 5, 3, 2 ** d, 1))
    shapes.append((f"dec{i}.proj_in", 2 * Ns[i], N, F, T, 1, 1, 1, 1))
    shapes.append((f"dec{i}.out_proj", N, 2, F, T, 1, 1, 1, 1))
for d in range(7):
    shapes.append((f"mid.H{d}", 256, 256, 448, 64, 5, 3, 2 ** d, 1))
tot_f = tot_t = 0
flt = os.environ.get("SHAPES")
if flt:
    shapes = [s for s in shapes if any(s[0].startswith(f) for f in flt.split(","))]
for name, Cin, Cout, F, T, KH, KW, dil, cnt in shapes:
    x = torch.randn(B, Cin, F, T, device="cuda")
    w = torch.randn(Cout, Cin, KH, KW, device="cuda") / math.sqrt(Cin * KH * KW)
    pc = ops.PackedConv(w, os.environ.get('PRECISION', 'f32'))
    out = torch.empty(B, Cout, F, T, device="cuda")
    VJP = os.environ.get("VJP") == "1"          # time the input-VJP form instead: transposed weights, in_scale (the gate)
    if VJP:
        x, out = out, torch.empty(B, Cin, F, T, device="cuda")
        x.normal_()
        kw = dict(dil=dil, transpose=True, in_scale=torch.randn(B, Cout, device="cuda"))
    else:
        kw = dict(dil=dil)
    for _ in range(2):
        ops.conv2d(x, pc, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        ops.conv2d(x, pc, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * Cout * Cin * KH * KW * F * T
    tot_f += fl; tot_t += ms
    print(f"{name:16s} Cin={Cin:4d} Cout={Cout:4d} F={F:4d} T={T:5d} k={KH}x{KW} dil={dil:3d}  {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF/s")
print(f"TOTAL fwd convs: {tot_t:.2f} ms, {tot_f/tot_t/1e9:.1f} TF/s")
