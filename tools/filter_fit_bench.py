"""Duration of the single-workgroup projected-gradient filter fit (babe_filter_fit) on synthetic statistics."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd.stft import STFTOps, make_fit_cfg
dev = torch.device("cuda", 0)
st = STFTOps(4096, 368368, 44100, dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 368368, generator=g).to(dev)
H = st.design_filter(torch.tensor([[3000.0], [-30.0]], device=dev))
y = st.apply_filter(x, H)
stats = st.mag_stats(st.stft(x), st.stft(y))
cfg = make_fit_cfg(tol=(0.0, 0.0))          # never converges early: max_iter iterations
K = int(os.environ.get("K", "5"))
p0 = torch.tensor([[[1000.0 * (i + 1) for i in range(K)], [-20.0 - 5.0 * i for i in range(K)]]], device=dev)
for _ in range(2):
    p = p0.clone(); nit = st.filter_fit(stats, p, cfg)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
p = p0.clone()
e0.record(); nit = st.filter_fit(stats, p, cfg); e1.record(); torch.cuda.synchronize()
print(f"filter_fit (K={K}): {e0.elapsed_time(e1)*1e3:.0f} us for {int(nit[0])} iterations -> fc={[round(float(v), 2) for v in p[0,0]]} Hz A={[round(float(v), 4) for v in p[0,1]]} dB")
# trajectory: parameters after n iterations (compare BABE_FIT_FAST=0 / 1 runs of this script)
for n in (1, 2, 3, 5, 10, 20, 40):
    cfgn = make_fit_cfg(tol=(0.0, 0.0), max_iter=n) if "max_iter" in make_fit_cfg.__code__.co_varnames else None
    if cfgn is None:
        break
    p = p0.clone(); st.filter_fit(stats, p, cfgn); torch.cuda.synchronize()
    print(f"  after {n:3d}: fc={[round(float(v), 3) for v in p[0,0]]} A={[round(float(v), 5) for v in p[0,1]]}")
