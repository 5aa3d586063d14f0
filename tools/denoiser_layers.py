"""Per-launch timing of the denoiser's convolutions (debug tool): groups by (kernel, Cin, Cout, H, W, stride)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd.networks import denoiser as dn

cfg = dict(depth=6, num_tfc=3, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
net = dn.MultiStage_denoise(cfg)
net.load_state_dict(dn.init_state_dict(cfg, seed=0))
net.to("cuda")
X = torch.randn(1, 2, 431, 513, device="cuda")
net(X)
rec = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig_conv, orig_tconv = dn._conv, dn._tconv


def timed(fn, key, flops, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(*a, **k)
    e1.record()
    torch.cuda.synchronize()
    r = rec[key]
    r[0] += 1
    r[1] += e0.elapsed_time(e1)
    r[2] += flops


def conv(x, pc, out, **k):
    fl = 2.0 * out.numel() * pc.Cin * pc.KH * pc.KW
    timed(orig_conv, (f"{pc.KH}x{pc.KW}s{k.get('stride', 1)}", pc.Cin, pc.Cout, out.shape[2], out.shape[3]), fl, x, pc, out, **k)


def tconv(x, pc, out, ch, cw):
    fl = 2.0 * x.numel() * pc.Cout * 16
    timed(orig_tconv, ("tconv", pc.Cin, pc.Cout, out.shape[2], out.shape[3]), fl, x, pc, out, ch, cw)


dn._conv, dn._tconv = conv, tconv
net(X)
tot = sum(r[1] for r in rec.values())
for k, r in sorted(rec.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{str(k):40s} n={r[0]:3d} {r[1]:8.2f} ms {100*r[1]/tot:5.1f}%  {r[2]/r[1]/1e9:7.1f} TF/s")
print(f"total {tot:.1f} ms")
