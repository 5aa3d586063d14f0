"""Time every (1,1) conv call of one UNet evaluation (fwd + VJP) stand-alone, with the call's own arguments."""
import sys, os, torch, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from babe_amd import ops
from babe_amd.networks import unet_engine
orig = ops.conv2d
calls = []
def conv2d(x, pc, out, dil=1, transpose=False, x2=None, **kw):
    if pc.KH == 1 and pc.KW == 1:
        calls.append((x, pc, out, dict(dil=dil, transpose=transpose, x2=x2, **kw)))
    return orig(x, pc, out, dil=dil, transpose=transpose, x2=x2, **kw)
class OpsProxy:
    def __getattr__(self, k): return conv2d if k == "conv2d" else getattr(ops, k)
unet_engine.ops = OpsProxy()
from babe_amd.config import default_args
from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
dev = torch.device("cuda", 0)
args = default_args(sample_rate=44100, audio_len=368368, T=35)
net = Unet_CQT_oct_with_attention(args, dev)
net.load_state_dict(init_state_dict(args.network.Ns, args.network.num_dils, seed=0, gate_scale=1.0))
B = int(os.environ.get("B", "2"))
x = torch.randn(B, 368368, device=dev)
y = net.fwd_nograd(x, torch.full((B, 1), 0.3, device=dev))
g = net.vjp(torch.randn_like(y))
torch.cuda.synchronize()
agg = collections.OrderedDict()
for (x, pc, out, kw) in calls:
    Bx, C1, F, T = x.shape
    Cin = C1 + (kw["x2"].shape[1] if kw.get("x2") is not None else 0)
    Cout = out.shape[1]
    key = (Cin, Cout, F, T, kw.get("x2") is not None, kw.get("res") is not None, kw.get("in_scale") is not None, kw.get("oscale") is not None, bool(kw["transpose"]))
    if key in agg:
        agg[key][0] += 1
        continue
    for _ in range(2): orig(x, pc, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): orig(x, pc, out, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    mb = 4.0 * Bx * F * T * (Cin + Cout * (2 if kw.get("res") is not None else 1)) / 1e6
    agg[key] = [1, us, mb]
tot = 0
print("count Cin Cout F T 2src res isc osc T | us MB GB/s")
for k, (n, us, mb) in sorted(agg.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    tot += n * us
    print(n, *[int(v) for v in k], "| %.1f %.1f %.0f" % (us, mb, mb / us * 1e3), " total_us=%.0f" % (n * us))
print("TOTAL us", tot, "launches", sum(v[0] for v in agg.values()))
