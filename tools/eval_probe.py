"""One score evaluation of the HIP sampler on the oracle's own inputs (tools/tmp_eval_diag.npz: x, t, params in; score, x_den,
params out of every evaluate() call of the oracle run of clip 0 of tests/golden/sampler_full_46046.npz)."""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import tests.test_gpu_unet_full as T
s = T.load("sampler_full_46046.npz")
d = np.load(os.path.join(R, "tools", "tmp_eval_diag.npz"))
L = int(s["L"])
net = T.full_net(L)
smp = T._full_sampler(net, s)
y = torch.from_numpy(d["y"]).cuda()
st = smp.stft_ops(L, y.device)
specY = st.stft(y)
rel = lambda a, b: float((a.double().cpu() - torch.as_tensor(b).double()).norm() / torch.as_tensor(b).double().norm())
n = len([k for k in d.files if k.startswith("xin")])
for i in range(n):
    x = torch.from_numpy(d[f"xin{i}"]).cuda(); t = float(d[f"t{i}"]); pin = torch.from_numpy(d[f"pin{i}"]).cuda().unsqueeze(0).contiguous()
    dd, xden, pout = smp.evaluate(x, t, y, specY, pin, True)
    score = -dd / t
    sref = torch.from_numpy(d[f"score{i}"])
    # split the score into its two terms: (x_den - x)/t^2 and the guidance
    tw_h = (xden - x) / t ** 2; tw_r = (torch.from_numpy(d[f"xden{i}"]) - torch.from_numpy(d[f"xin{i}"])) / t ** 2
    g_h = tw_h - score; g_r = tw_r - sref
    print(f"evaluation {i} (t = {t:.4g}): x_den rel {rel(xden, d[f'xden{i}']):.2e}, score rel {rel(score, sref):.2e}, guidance term rel {rel(g_h, g_r):.2e} "
          f"(|guidance|/|score| = {float(g_r.norm() / sref.norm()):.3f}), params max dfc {float((pout[0,0].cpu() - torch.from_numpy(d[f'pout{i}'])[0]).abs().max()):.3g} Hz", flush=True)
