"""Host time to ENQUEUE one score evaluation (UNet forward + input-VJP, one lane) vs the GPU time it takes: is the Python
side ahead of the GPU?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from babe_amd.config import default_args
from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
prec = os.environ.get("PRECISION", "f32")
dev = torch.device("cuda", 0)
args = default_args(sample_rate=44100, audio_len=368368, T=35)
net = Unet_CQT_oct_with_attention(args, dev, precision=prec)
net.load_state_dict(init_state_dict(args.network.Ns, args.network.num_dils, seed=0, gate_scale=1.0))
x = torch.randn(1, 368368, device=dev); cn = torch.full((1, 1), 0.3, device=dev); g = torch.randn(1, 368368, device=dev)
for _ in range(2):
    y = net.fwd_nograd(x, cn); gx = net.vjp(g)
torch.cuda.synchronize()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    y = net.fwd_nograd(x, cn); gx = net.vjp(g)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{prec}: host enqueue {1e3*(t1-t0)/N:.1f} ms per evaluation; GPU (wall incl. drain) {1e3*(t2-t0)/N:.1f} ms per evaluation")
