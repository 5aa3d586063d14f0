"""HIP filter fit against the oracle's own trajectory on dumped evaluations (tools/tmp_fit_diag.npz, written in the build
container from the oracle run of clip 0 of tests/golden/sampler_full_46046.npz: x_den, y, parameters in / out and the
per-iteration trajectory of every fit_params call)."""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd.stft import STFTOps, make_fit_cfg
d = np.load(os.path.join(R, "tools", "tmp_fit_diag.npz"))
L = d["xden0"].shape[-1]
st = STFTOps(4096, L, 44100, "cuda")
n = len([k for k in d.files if k.startswith("xden")])
for i in range(n):
    xd = torch.from_numpy(d[f"xden{i}"]).cuda(); y = torch.from_numpy(d[f"y{i}"]).cuda()
    pin = torch.from_numpy(d[f"pin{i}"]).cuda(); traj = torch.from_numpy(d[f"traj{i}"])
    stats = st.mag_stats(st.stft(xd), st.stft(y))
    line = f"evaluation {i}: "
    for it in (1, 2, 5, 20, 50, 100):
        cfg = make_fit_cfg(mu=[100.0, 1.0], tol=[5e-3, 5e-3], max_iter=it, fcmin=20, fcmax=22050, Amin=-50, Amax=30,
                           clamp_fc=True, clamp_A=True, only_negative_A=True, weighting="sqrt")
        p = pin.unsqueeze(0).clone().contiguous()
        st.filter_fit(stats, p, cfg)
        ref = traj[it - 1]
        line += f"it{it}: dfc {float((p[0,0].cpu()-ref[0]).abs().max()):.3g} Hz dA {float((p[0,1].cpu()-ref[1]).abs().max()):.3g} | "
    print(line, flush=True)
    print("    HIP  after 100:", p[0].cpu().tolist()); print("    ref  after 100:", traj[-1].tolist(), flush=True)
