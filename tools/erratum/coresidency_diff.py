"""What exactly differs when conv_fewco runs beside the 64-channel bf16 conv (tools/coresidency_probe.py found 30 of 30 runs)?"""
import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd import ops
torch.manual_seed(0)
w = torch.randn(256, 2, 5, 3, device="cuda") / math.sqrt(30); pcw = ops.PackedConv(w)
gy = torch.randn(2, 256, 448, 64, device="cuda")
gy0 = gy.clone()
def fewco():
    out = torch.empty(2, 2, 448, 64, device="cuda"); ops.conv2d(gy, pcw, out, transpose=True, alpha=0.7); return out
ww = torch.randn(64, 64, 5, 3, device="cuda") / math.sqrt(64 * 15); pc = ops.PackedConv(ww, "bf16")
xx = torch.randn(2, 64, 64, 4096, device="cuda"); xx0 = xx.clone(); oo = torch.empty(2, 64, 64, 4096, device="cuda")
def partner(): ops.conv2d(xx, pc, oo, dil=1)
partner(); torch.cuda.synchronize(); oo_ref = oo.clone()
ref = fewco().clone(); torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for i in range(6):
    with torch.cuda.stream(sB): partner()
    with torch.cuda.stream(sA): o = fewco()
    with torch.cuda.stream(sB): partner()
    torch.cuda.synchronize()
    d = (o != ref)
    n = int(d.sum())
    print(f"run {i}: {n} of {o.numel()} outputs differ; input gy intact {bool(torch.equal(gy, gy0))}; partner input intact {bool(torch.equal(xx, xx0))}; partner output == alone {bool(torch.equal(oo, oo_ref))}")
    if n:
        idx = d.nonzero()
        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
        print(f"   index range (b, c, f, t): {lo} .. {hi}")
        for k in idx[:6].tolist() + idx[-3:].tolist():
            b, c, f, t = k
            print(f"   [{b},{c},{f},{t}] ref {float(ref[b,c,f,t]):+.6f} got {float(o[b,c,f,t]):+.6f}")
        fl = (d[0, 0] | d[0, 1] | d[1, 0] | d[1, 1]).nonzero()
        print(f"   distinct rows f: {sorted(set(fl[:, 0].tolist()))[:20]}  t range {int(fl[:,1].min())}..{int(fl[:,1].max())}")

# ---- which term is wrong?  float64 partial sums per channel slice (CS = 8: 32 channels each) for the first corrupted outputs
import torch.nn.functional as Fn
gyc, wc = gy.double().cpu(), w.double().cpu()
def partials(b, c, f, t):
    """[8] partial sums (alpha excluded) of output (b, c, f, t) of the input-VJP: sum over ci in slice, kh, kw of
    gy[b, ci, f + (kh-2), t + (kw-1)] * w[ci, c, 4-kh, 2-kw]"""
    out = []
    for s in range(8):
        acc = 0.0
        for ci in range(32 * s, 32 * s + 32):
            for kh in range(5):
                fr = f + kh - 2
                if fr < 0 or fr >= 448: continue
                for kw in range(3):
                    tt = t + kw - 1
                    if tt < 0 or tt >= 64: continue
                    acc += float(gyc[b, ci, fr, tt]) * float(wc[ci, c, 4 - kh, 2 - kw])
        out.append(acc)
    return out
if n:
    for k in idx[:4].tolist():
        b, c, f, t = k
        e = (float(o[b, c, f, t]) - float(ref[b, c, f, t])) / 0.7
        P = partials(b, c, f, t)
        print(f"[{b},{c},{f},{t}] error/alpha {e:+.6f}; slice partials {[round(v, 4) for v in P]}; sum {sum(P):+.6f} ref/alpha {float(ref[b,c,f,t])/0.7:+.6f}")
        # candidates: a slice partial of ANOTHER element (t-1, t+1, other row) taken instead of the own one
        for (df, dt) in ((0, -1), (0, 1), (0, -2), (0, 2), (-1, 0), (1, 0)):
            Q = partials(b, c, f + df, t + dt) if 0 <= f + df < 448 and 0 <= t + dt < 64 else None
            if Q:
                hits = [s for s in range(8) if abs((Q[s] - P[s]) - e) < 2e-4]
                if hits: print(f"     == slice {hits} of element (f{df:+d}, t{dt:+d}) taken instead of the own one")

# ---- is the error of a row ONE term with a wrong weight?  e(t) = k * gy[b, ci, fr, t + s] for the affected t of one row
if n:
    b, c, f, _ = idx[0].tolist()
    ts = sorted(set(i[3] for i in idx.tolist() if i[0] == b and i[1] == c and i[2] == f))
    ev = torch.tensor([(float(o[b, c, f, t]) - float(ref[b, c, f, t])) / 0.7 for t in ts], dtype=torch.float64)
    print(f"row (b={b}, c={c}, f={f}): affected t = {ts}")
    best = []
    for s in (-2, -1, 0, 1, 2):
        tt = torch.tensor([t + s for t in ts])
        ok = (tt >= 0) & (tt < 64)
        if not bool(ok.all()): continue
        for fr in range(max(0, f - 2), min(448, f + 3)):
            X = gyc[b, :, fr, :][:, tt]                       # [256, nt]
            k = (X @ ev) / (X * X).sum(1)                     # least-squares scale per channel
            res = ((X * k[:, None] - ev[None, :]) ** 2).sum(1).sqrt() / ev.norm()
            j = int(res.argmin())
            best.append((float(res[j]), j, fr, s, float(k[j])))
    best.sort()
    for r_, ci, fr, s, k in best[:5]:
        print(f"   ci {ci} row {fr} (kh {fr - f + 2}) shift {s:+d}: k = {k:+.6f}, relative residual {r_:.3e}; weights of that (ci, kh): {[round(float(v), 6) for v in wc[ci, c, 4 - (fr - f + 2)]]}")
