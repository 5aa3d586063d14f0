"""One score evaluation (UNet forward + input-VJP at full width, B=1) for a rocprofv3 --kernel-trace run; with an argument:
aggregate the trace CSV by (kernel, grid) - the per-call-site view that the per-kernel summary hides."""
import sys, os
if len(sys.argv) > 1:
    import csv, collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(sys.argv[1])):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm.split("(")[0][:60]
        k = (nm, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[k][0] += 1; agg[k][1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"total {tot/1e3:.2f} ms over {sum(v[0] for v in agg.values())} launches")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
        print(f"{us/tot*100:5.2f}%  {n:4d} x {us/n:8.1f} us  grid={k[1]:>9s} wg={k[2]:>4s}  {k[0]}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from babe_amd.config import default_args
from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
dev = torch.device("cuda", 0)
args = default_args(sample_rate=44100, audio_len=368368, T=35)
net = Unet_CQT_oct_with_attention(args, dev, precision=os.environ.get("PRECISION", "f32"))
net.load_state_dict(init_state_dict(args.network.Ns, args.network.num_dils, seed=0, gate_scale=1.0))
x = torch.randn(1, 368368, device=dev)
for _ in range(2):
    y = net.fwd_nograd(x, torch.full((1, 1), 0.3, device=dev))
    g = net.vjp(torch.randn_like(y))
torch.cuda.synchronize()
