import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
from babe_amd import ops
from babe_amd.networks import unet_engine
orig = ops.conv2d
log = collections.Counter()
def conv2d(x, pc, out, dil=1, transpose=False, x2=None, **kw):
    if pc.KH == 1 and pc.KW == 1:
        B, C1, F, T = x.shape
        Cin = C1 + (x2.shape[1] if x2 is not None else 0)
        Cout = out.shape[1]
        why = []
        if T % 4: why.append("T%4")
        if x.data_ptr() % 16: why.append("in align")
        if x.stride(0) % 4 or x.stride(1) % 4: why.append("in strides")
        if x2 is not None and (x2.data_ptr() % 16 or x2.stride(0) % 4 or x2.stride(1) % 4 or C1 % 16): why.append("in2 C1=%d" % C1)
        if F * T < 4096 and Cin < 256: why.append("tiny")
        log[(tuple(x.shape), Cin, Cout, ",".join(why) or "ok", transpose)] += 1
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
x = torch.randn(1, 368368, device=dev)
y = net.fwd_nograd(x, torch.full((1, 1), 0.3, device=dev))
g = net.vjp(torch.randn_like(y))
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: kv[0][3]):
    print(v, k)
