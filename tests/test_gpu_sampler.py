"""End-to-end: HIP UNet (+CQT) forward/VJP and the BlindSampler vs golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  Needs a MI355X."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name)).items()}


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rms_err(a, b):
    return float((a.detach().double().cpu() - b.double().cpu()).pow(2).mean().sqrt())


def small_net(T=3, start_sigma=0.05, precision="f32"):
    from babe_amd.config import default_args
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    g = load("unet_small.npz")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    args = default_args(sample_rate=22050, audio_len=92092, Ns=[8, 8, 8, 8, 16, 16, 16], T=T, start_sigma=start_sigma)
    net = Unet_CQT_oct_with_attention(args, "cuda", precision=precision)
    missing = net.load_state_dict(sd, strict=True)
    return g, args, net


def test_unet_forward_and_autograd_vjp_vs_reference_golden():
    g, args, net = small_net()
    gen = torch.Generator().manual_seed(int(g["unet_seed"]))
    x = (0.1 * torch.randn(1, 92092, generator=gen)).cuda().requires_grad_(True)
    y = net(x, g["unet_cnoise"].cuda())
    assert rel(y, g["unet_y"]) < 2e-5
    wv = torch.randn(y.shape, generator=gen)
    gx, = torch.autograd.grad((y * wv.cuda()).sum(), x)          # reference-style autograd through the HIP VJP
    assert rel(gx, g["unet_gx"]) < 2e-4


class ResidualNet:
    """Same wrapper as tests/golden/make_golden.py: a*net(x,c) + (sigma/sigma_data)*x, sigma = exp(4c)."""

    def __init__(self, inner, a, sigma_data):
        self.inner, self.a, self.sd = inner, a, sigma_data
        self.CQTransform = inner.CQTransform

    supports_lanes = True

    @property
    def concurrent_lanes_ok(self):
        return getattr(self.inner, "concurrent_lanes_ok", True)

    def lanes_ok_for(self, noise_device="cpu"):
        f = getattr(self.inner, "lanes_ok_for", None)
        return f(noise_device) if f is not None else self.concurrent_lanes_ok

    def fwd_nograd(self, x, cn, lane=None):
        self.k = float(torch.exp(4 * cn[0, 0])) / self.sd          # (one sigma per call; same for every lane of a step)
        kw = {} if lane is None else {"lane": lane}
        return self.a * self.inner.fwd_nograd(x, cn, **kw) + self.k * x

    def vjp(self, g, lane=None):
        kw = {} if lane is None else {"lane": lane}
        return self.a * self.inner.vjp(g, **kw) + self.k * g


def params_close(p, q):
    """fc within 1 %, A within 1 dB/oct: the reference's 100-iteration GD is not contractive."""
    p, q = p.cpu(), q.cpu()
    return bool(torch.allclose(p[0], q[0], rtol=1e-2, atol=0) and torch.allclose(p[1], q[1], rtol=0, atol=1.0))


def test_blind_sampler_T3_vs_reference_golden():
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_small.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    assert torch.equal(t, s["t"])
    for i in range(3):
        assert rel(dden[i], s["data_denoised"][i]) < 1e-3, i
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    # north-star parity bar: 1e-3 RMS (fp32) on the output
    assert rms_err(x, s["x"]) < 1e-3
    assert rel(x, s["x"]) < 2e-3
    assert params_close(fp, s["filter_params"])
    # known-filter variant
    it = iter(noises)
    xk = smp.predict_bwe(s["y"].cuda(), torch.tensor([[2000.0], [-40.0]]), "fc_A")
    assert rms_err(xk, s["x_known"]) < 1e-3 and rel(xk, s["x_known"]) < 2e-3


def test_per_clip_batch_equals_single_clip_runs():
    """Independent clips batched (per_clip semantics) == each clip alone (the reference's B=1 behaviour)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    g, args, net = small_net(T=2, start_sigma=0.05)
    L = 92092
    gen = torch.Generator().manual_seed(99)
    y = 0.1 * torch.randn(2, L, generator=gen)
    noises = [torch.randn(2, L, generator=gen) for _ in range(3)]
    rn = ResidualNet(net, 0.3, 0.063)
    smp = BlindSampler(rn, EDM(args), args, batch_semantics="per_clip")
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    xb, fpb = smp.predict_blind_bwe(y.cuda())
    for b in range(2):
        it = iter([n[b:b + 1] for n in noises])
        smp._randn = lambda shape, device: next(it).to(device)
        x1, fp1 = smp.predict_blind_bwe(y[b:b + 1].cuda())
        assert rel(xb[b:b + 1], x1) < 1e-4
        assert params_close(fpb[b], fp1)


def test_fir_kernel_and_adjoint():
    from babe_amd.stft import fir_same
    s = load("edm_sampler_firwin.npz")
    gen = torch.Generator().manual_seed(int(s["fir_x_seed"]))
    x = torch.randn(2, 5000, generator=gen)
    taps = s["taps_22050"]
    y = fir_same(x.cuda(), taps.cuda())
    assert rel(y, s["fir_y"]) < 1e-5
    g = torch.randn(2, 5000, generator=gen)
    gx = fir_same(g.cuda(), taps.cuda(), adjoint=True)
    lhs = float((y.double().cpu() * g.double()).sum())
    rhs = float((x.double() * gx.double().cpu()).sum())
    assert abs(lhs - rhs) < 1e-4 * (abs(lhs) + abs(rhs)) + 1e-6
    from babe_amd.utils.bandwidth_extension import get_FIR_lowpass
    assert torch.equal(get_FIR_lowpass(500, 1000, 1, 22050)[0, 0], taps)


def test_edm_sampler_firwin_T3_vs_reference_golden():
    """Config #1 (known 500-tap FIR, testing/edm_sampler.py) on the HIP path vs the reference's output."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.edm_sampler import Sampler
    s = load("edm_sampler_firwin.npz")
    g, args, net = small_net(T=3)
    args.tester.posterior_sampling.xi = float(s["xi"])
    args.tester.diff_params.ro = float(s["ro"])
    args.tester.diff_params.sigma_max = float(s["sigma_max"])
    args.tester.diff_params.Schurn = float(s["Schurn"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = Sampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x = smp.predict_bwe(s["y"].cuda(), s["taps_22050"], "firwin")
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3

@pytest.mark.parametrize("mode", ["dc", "xi0"])
def test_edm_sampler_replacement_modes_T3_vs_reference_golden(mode):
    """testing/edm_sampler.py with posterior_sampling.data_consistency = True (guided score, then x0 <- y + x0 - A(x0) on the
    Tweedie estimate, :113-122) and with xi = 0 (the replacement step on the plain denoised estimate, :124-130) on the HIP path
    vs the reference's outputs (tests/golden/make_golden.py::g22)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.edm_sampler import Sampler
    s9, s = load("edm_sampler_firwin.npz"), load("edm_sampler_modes.npz")
    g, args, net = small_net(T=3)
    args.tester.posterior_sampling.xi = float(s9["xi"]) if mode == "dc" else 0.0
    args.tester.posterior_sampling.data_consistency = mode == "dc"
    args.tester.diff_params.ro = float(s9["ro"])
    args.tester.diff_params.sigma_max = float(s9["sigma_max"])
    args.tester.diff_params.Schurn = float(s9["Schurn"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = Sampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x = smp.predict_bwe(s9["y"].cuda(), s9["taps_22050"], "firwin")
    ref = s[f"x_{mode}"]
    print(f"edm_sampler {mode}: RMS err {rms_err(x, ref):.2e}, rel {rel(x, ref):.2e}")
    assert rms_err(x, ref) < 1e-3 and rel(x, ref) < 2e-3


@pytest.mark.parametrize("precision,tol_y,tol_g", [("bf16x3", 1e-4, 1e-3), ("bf16", 2e-2, 6e-2)])
def test_unet_reduced_precision_modes_vs_reference_golden(precision, tol_y, tol_g):
    """bf16-MFMA conv modes (fp32 storage and accumulation): stated, looser tolerances."""
    g, args, net = small_net(precision=precision)
    gen = torch.Generator().manual_seed(int(g["unet_seed"]))
    x = (0.1 * torch.randn(1, 92092, generator=gen)).cuda().requires_grad_(True)
    y = net(x, g["unet_cnoise"].cuda())
    assert rel(y, g["unet_y"]) < tol_y
    wv = torch.randn(y.shape, generator=gen)
    gx, = torch.autograd.grad((y * wv.cuda()).sum(), x)
    assert rel(gx, g["unet_gx"]) < tol_g


def test_blind_sampler_bf16x3_meets_fp32_parity_bar():
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_small.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]), precision="bf16x3")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(s["y"].cuda())
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 5e-3


def test_predict_bwe_AR_T3_vs_reference_golden():
    """AR out-painting mode ("next" row 2): mask-mixed degradation + replacement data consistency."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_ar.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    clean = 0.1 * torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    ov = int(s["overlap"])
    mask = torch.ones(1, L)
    mask[..., ov:] = 0
    y_masked = torch.zeros(1, L)
    y_masked[..., :ov] = clean[..., :ov]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    assert torch.equal(smp.prepare_smooth_mask(mask, 50)[0, : ov + 8], s["smooth_mask"])
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x = smp.predict_bwe_AR(s["ylpf"].cuda(), y_masked.cuda(), torch.tensor([[2000.0], [-40.0]]), "fc_A", mask=mask.cuda())
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3
    # the observed (masked) region is reproduced by the data-consistency step
    assert float((x[:, : ov - 60].cpu() - clean[:, : ov - 60]).abs().max()) < 5e-2


@pytest.mark.parametrize("fs,L,B", [(16000, 184184, 1), (22050, 92092, 3)])
def test_unet_other_geometries_vs_oracle(fs, L, B):
    """Config-#5 geometry (16 kHz CocoChorales, audio_len 184184: T_j = 32..2048) and an odd batch size:
    HIP UNet forward + input-VJP vs the CPU oracle with autograd (reduced width)."""
    from babe_amd.config import default_args
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    from oracle import unet as UN
    from oracle.nsgt import CQT_nsgt as OracleCQT
    g = load("unet_small.npz")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    args = default_args(sample_rate=fs, audio_len=L, Ns=[8, 8, 8, 8, 16, 16, 16])
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(L + B)
    x = 0.1 * torch.randn(B, L, generator=gen)
    cn = torch.full((B, 1), -0.3)
    wv = torch.randn(B, L, generator=gen)
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=[2, 3, 4, 5, 6, 7, 7])
    ocqt = OracleCQT(7, 64, "oct", ("kaiser", 1), fs, L)
    xr = x.clone().requires_grad_(True)
    yref = UN.unet_forward(sd, cfg, ocqt, xr, cn)
    gref, = torch.autograd.grad((yref * wv).sum(), xr)
    xg = x.cuda().requires_grad_(True)
    y = net(xg, cn.cuda())
    gx, = torch.autograd.grad((y * wv.cuda()).sum(), xg)
    assert rel(y, yref) < 5e-5
    assert rel(gx, gref) < 5e-4


def test_unet_stream_lanes_equal_single_stream():
    """Batch items on separate HIP streams (Unet_CQT_oct_with_attention.MAX_LANES) give bit-identical forward and VJP."""
    from babe_amd.config import default_args
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    Ns = [16, 16, 16, 16, 32, 32, 32]
    args = default_args(sample_rate=22050, audio_len=92092, Ns=Ns, T=3)
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(init_state_dict(Ns, args.network.num_dils, seed=0, gate_scale=1.0))
    g = torch.Generator().manual_seed(0)
    x = (0.1 * torch.randn(3, 92092, generator=g)).cuda()
    cn = torch.tensor([[-0.4], [-1.0], [0.3]]).cuda()
    gy = torch.randn(3, 92092, generator=g).cuda()
    ref = None
    for lanes in (1, 2, 3, 2):
        net.MAX_LANES = lanes
        y = net.fwd_nograd(x, cn)
        gx = net.vjp(gy)
        torch.cuda.synchronize()
        if ref is None:
            ref = (y.clone(), gx.clone())
        else:
            assert torch.equal(y, ref[0]) and torch.equal(gx, ref[1]), f"lanes={lanes}"


# ---------------------------------------------------------------------------------------------------------------------
# Round-2 goldens: the benchmark's T=35 schedule, teacher-forced single steps, B=2 with the reference's batch coupling,
# predict_unconditional and BlindSampler.predict_bwe('firwin')  (tests/golden/make_golden.py::g13, g14)

def _noises(seed, L, pre, n, B=1):
    gen = torch.Generator().manual_seed(seed)
    for _ in range(pre):
        torch.randn(L, generator=gen)                  # the draws that made the observation
    return [torch.randn(B, L, generator=gen) for _ in range(n)]


def test_blind_sampler_T35_vs_reference_golden():
    """predict_blind_bwe at the benchmark's schedule: T=35 from sigma=0.2 down to 1e-4 (69 score evaluations,
    score = (x_den - x)/t^2 with t down to 1e-4), network unwrapped, vs the imported reference
    (testing/blind_bwe_sampler.py:619-769).  Bars: per-step denoised estimate 1e-3 rel, filters fc 1 % / A 1 dB/oct,
    output RMS error < 1e-3 (north star)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_T35.npz")
    g, args, net = small_net(T=35, start_sigma=float(s["start_sigma"]))
    L = 92092
    noises = _noises(int(s["seed"]), L, 1, 36)
    smp = BlindSampler(net, EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    assert torch.equal(t, s["t"])
    worst = 0.0
    for i in range(35):
        e = rel(dden[i][:, ::16], s["data_denoised_sub16"][i])
        worst = max(worst, e)
        assert e < 1e-3, (i, e)
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    print(f"T=35: worst per-step x_den rel {worst:.2e}, output RMS err {rms_err(x, s['x']):.2e}, rel {rel(x, s['x']):.2e}")
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3
    assert params_close(fp, s["filter_params"])


def test_blind_sampler_teacher_forced_steps_vs_reference():
    """Single steps from recorded reference state (x_i, filter parameters entering step i, the step's noise) ->
    (x_{i+1}, filter parameters leaving it): localises any late-step mismatch of the T=35 run (steps 0, 16, 33 and the
    final Euler step 34 where t_next = 0)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_T35.npz")
    g, args, net = small_net(T=35, start_sigma=float(s["start_sigma"]))
    L = 92092
    noises = _noises(int(s["seed"]), L, 1, 36)
    smp = BlindSampler(net, EDM(args), args)
    y = s["y"].cuda()
    st = smp.stft_ops(L, y.device)
    specY = st.stft(y)
    t = s["t"]
    gamma = smp.diff_params.get_gamma(t)
    for i in [int(v) for v in s["tf_steps"]]:
        fp_in = s[f"tf{i}_fp_in"].unsqueeze(0).contiguous().cuda()
        x1, fp1, _ = smp.step(s[f"tf{i}_x_in"].cuda(), t[i], gamma[i], t[i + 1], noises[1 + i].cuda().contiguous(), y,
                              specY, fp_in, blind=True)
        e = rel(x1, s[f"tf{i}_x_out"])
        print(f"teacher-forced step {i}: x_next rel {e:.2e}")
        assert e < 1e-4, (i, e)
        assert params_close(fp1[0], s[f"tf{i}_fp_out"]), (i, fp1[0], s[f"tf{i}_fp_out"])


def test_blind_sampler_B2_reference_batch_semantics_vs_golden():
    """batch_semantics='reference': ONE filter fitted on the flattened batch (utils/blind_bwe_utils.py:295) and a
    whole-batch guidance norm (testing/blind_bwe_sampler.py:125), B=2, vs the imported reference."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_B2.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]     # see make_golden.g13: keeps the fit contractive
    L = 92092
    noises = _noises(int(s["seed"]), L, 2, 4, B=2)
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args, batch_semantics="reference")
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    assert fp.shape == (2, 5) and dfil.shape == (3, 2, 5)
    for i in range(3):
        assert rel(dden[i][:, ::16], s["data_denoised_sub16"][i]) < 1e-3, i
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3
    assert params_close(fp, s["filter_params"])


def test_add_obs_noise_matches_the_reference_formula():
    """babe_add_obs_noise: y += sqrt(var(y, -1) / snr) * noise in place, per clip (blind_bwe_sampler.py:80-86)."""
    from babe_amd.stft import add_obs_noise
    g = torch.Generator().manual_seed(7)
    y = (torch.randn(3, 92092, generator=g) * torch.tensor([[0.05], [0.2], [1.0]]) + 0.01).cuda()
    n = torch.randn(3, 92092, generator=g).cuda()
    snr = 10.0 ** (30 / 10)
    want = y.cpu() + torch.sqrt(torch.var(y.cpu(), -1) / snr).unsqueeze(-1) * n.cpu()
    got = add_obs_noise(y, n, snr)
    assert got.data_ptr() == y.data_ptr()
    assert rel(got, want) < 1e-6


def test_blind_sampler_observation_noise_regularisation_vs_golden():
    """posterior_sampling.SNR_observations + blind_bwe.sigma_den_estimate (conf/tester/blind_bwe_2.yaml; get_rec_grads :80-86,
    fit_params :542-552): noise added to the observations in place before every fit and every guidance evaluation, the fit on a
    noisy denoised estimate; 19 draws in the reference's order, vs the imported reference (G23)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_obs_noise.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    args.tester.posterior_sampling.SNR_observations = float(s["snr_db"])
    args.tester.blind_bwe.sigma_den_estimate = float(s["sigma_den"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(L, generator=gen)                      # the draw that made the observation
    noises = [torch.randn(1, L, generator=gen) for _ in range(int(s["ndraw"]))]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    y = s["y"].cuda()
    x, fp, dden, t, dfil = smp.predict_blind_bwe(y, rid=True)
    assert next(it, None) is None                          # every draw consumed, none missing
    assert torch.equal(y.cpu(), s["y"])                    # the caller's tensor is not touched (the reference mutates it)
    for i in range(3):
        assert rel(dden[i][:, ::16], s["data_denoised_sub16"][i]) < 1e-3, i
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3
    assert params_close(fp, s["filter_params"])


def test_predict_unconditional_and_predict_bwe_firwin_vs_reference_golden():
    """BlindSampler.predict_unconditional (:366-374) and BlindSampler.predict_bwe(..., 'firwin') (:306-364) through
    predict (:406-498): rid=True returns (x, guided Tweedie estimates, scores, t) like the reference."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_uncond_firwin.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    xu, dden, dscore, t = smp.predict_unconditional((1, L), "cuda", rid=True)
    assert torch.equal(t, s["unc_t"])
    # (the score of the LAST step, t = 1e-4, is (x_den - x)/1e-8 of two nearly equal fp32 vectors: its rounding noise is
    # percent-level in any fp32 implementation, so it is pinned through the Tweedie estimate and the output instead)
    for i in range(3):
        assert rel(dden[i][:, ::16], s["unc_den_sub16"][i]) < 1e-3, i
        if i < 2:
            assert rel(dscore[i][:, ::16], s["unc_score_sub16"][i]) < 2e-3, i
    assert rms_err(xu, s["unc_x"]) < 1e-3 and rel(xu, s["unc_x"]) < 2e-3
    _ = torch.randn(1, L, generator=gen)                 # the clean signal of the FIR case
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    it = iter(noises)
    xf, dden, dscore, t = smp.predict_bwe(s["fir_y"].cuda(), s["fir_taps"], "firwin", rid=True)
    for i in range(3):
        assert rel(dden[i][:, ::16], s["fir_den_sub16"][i]) < 1e-3, i
        if i < 2:
            assert rel(dscore[i][:, ::16], s["fir_score_sub16"][i]) < 2e-3, i
    assert rms_err(xf, s["fir_x"]) < 1e-3 and rel(xf, s["fir_x"]) < 2e-3


def test_sub_batching_equals_one_batch_and_bf16_sampler_tolerance():
    """(i) configs[2]-style batches: per-clip semantics restores a batch in sub-batches of max_segments_in_flight
    segments; with the same per-clip noise the result equals the one-batch run bit for bit.  (ii) the plain-bf16 conv mode
    through the whole sampler (T=3 golden): the UNet's bf16 error (2e-2 forward) reaches the output attenuated by
    c_out = sigma: stated tolerance 5e-3 RMS on a 0.1-RMS signal, filters fc 5 % / A 2 dB/oct (fp32 bar: 1e-3, 1 %, 1 dB)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_small.npz")
    g, args, net = small_net(T=2, start_sigma=0.05)
    L = 92092
    gen = torch.Generator().manual_seed(123)
    y = 0.1 * torch.randn(3, L, generator=gen)
    noises = [torch.randn(3, L, generator=gen) for _ in range(3)]
    rn = ResidualNet(net, 0.3, 0.063)
    smp = BlindSampler(rn, EDM(args), args, batch_semantics="per_clip")
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x_all, fp_all = smp.predict_blind_bwe(y.cuda())
    smp2 = BlindSampler(rn, EDM(args), args, batch_semantics="per_clip", max_segments_in_flight=2)
    state = {"i": 0, "row": 0}

    def sub_randn(shape, device):            # sub-batch k draws rows [row, row + b) of the same per-step noise tensors
        n = noises[state["i"] % 3][state["row"]:state["row"] + shape[0]]
        state["i"] += 1
        if state["i"] % 3 == 0:
            state["row"] += shape[0]
        return n.to(device)

    smp2._randn = sub_randn
    x_sub, fp_sub = smp2.predict_blind_bwe(y.cuda())
    assert torch.equal(x_sub, x_all) and torch.equal(fp_sub, fp_all)
    # (ii) bf16 sampler vs the fp32 reference golden
    g, args, netb = small_net(T=3, start_sigma=float(s["start_sigma"]), precision="bf16")
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    nz = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smpb = BlindSampler(ResidualNet(netb, float(s["res_a"]), 0.063), EDM(args), args)
    it2 = iter(nz)
    smpb._randn = lambda shape, device: next(it2).to(device)
    xb, fpb = smpb.predict_blind_bwe(s["y"].cuda())
    e = rms_err(xb, s["x"])
    print(f"bf16 sampler: output RMS err {e:.2e} (fp32 bar 1e-3), filters {fpb.cpu().tolist()} vs {s['filter_params'].tolist()}")
    assert e < 5e-3
    assert torch.allclose(fpb.cpu()[0], s["filter_params"][0], rtol=5e-2) and torch.allclose(fpb.cpu()[1], s["filter_params"][1], atol=2.0)


def test_clip_lanes_equal_single_stream_loop():
    """per-clip batches run their clips' whole evaluation chains on separate HIP streams (BlindSampler._sample_lanes, one
    network engine state per lane); with the same noise the result is bit-identical to the single-stream loop and to
    single-clip runs, for an even and an odd batch."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    g, args, net = small_net(T=3, start_sigma=0.05)
    L = 92092
    gen = torch.Generator().manual_seed(77)
    y = 0.1 * torch.randn(3, L, generator=gen)
    noises = [torch.randn(3, L, generator=gen) for _ in range(4)]
    for B in (2, 3):
        outs = []
        for lanes in (2, 1):
            smp = BlindSampler(net, EDM(args), args, batch_semantics="per_clip")
            smp.LANES = lanes
            it = iter([n[:B] for n in noises])
            smp._randn = lambda shape, device: next(it).to(device)
            outs.append(smp.predict_blind_bwe(y[:B].cuda()))
            torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), B
    smp = BlindSampler(net, EDM(args), args)
    it = iter([n[2:3] for n in noises])
    smp._randn = lambda shape, device: next(it).to(device)
    x1, fp1 = smp.predict_blind_bwe(y[2:3].cuda())
    assert torch.equal(outs[0][0][2:3], x1) and torch.equal(outs[0][1][2], fp1)


@pytest.mark.parametrize("norm", ["cosine", "smoothl1"])
def test_blind_sampler_alternative_guidance_distances(norm):
    """posterior_sampling.norm = 'cosine' / 'smoothl1' (get_rec_grads :99-103) on the HIP path (babe_cos_partial,
    babe_residual_seed_alt): T=3 blind run against the imported reference (G16)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_altnorm.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    args.tester.posterior_sampling.norm = norm
    args.tester.posterior_sampling.smoothl1_beta = float(s["smoothl1_beta"])
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    for i in range(3):
        assert params_close(dfil[i], s[f"data_filters_{norm}"][i]), (i, dfil[i], s[f"data_filters_{norm}"][i])
    assert rms_err(x, s[f"x_{norm}"]) < 1e-3 and rel(x, s[f"x_{norm}"]) < 2e-3
    assert params_close(fp, s[f"filter_params_{norm}"])


@pytest.mark.parametrize("tag,mag,logmag,fw", [("complex", False, False, "sqrt"), ("logmag", True, True, "None")])
def test_blind_sampler_stft_domain_guidance_distances(tag, mag, logmag, fw):
    """posterior_sampling.stft_distance.use (get_rec_grads :105-115; blind_bwe_2.yaml is the log-magnitude variant) on the HIP
    path (babe_stft_dist_partial / _grad + the STFT adjoint): T=3 blind run against the imported reference (G17).  The
    log-magnitude distance divides by |S| + 1e-8 on bins at the rounding floor: the reference's own guidance term moves by
    1 % with the rounding of the backward pass (oracle test), so its bar is looser and stated here."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_stftdist.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    ps = args.tester.posterior_sampling
    ps.stft_distance.use, ps.stft_distance.mag, ps.stft_distance.logmag = True, mag, logmag
    ps.stft_distance.nfft, ps.freq_weighting = int(s["nfft"]), fw
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    if tag == "complex":
        for i in range(3):
            assert params_close(dfil[i], s[f"data_filters_{tag}"][i]), (i, dfil[i], s[f"data_filters_{tag}"][i])
        assert rms_err(x, s[f"x_{tag}"]) < 1e-3 and rel(x, s[f"x_{tag}"]) < 2e-3
    else:
        # measured: HIP 0.104 rel / 8.7e-3 RMS, the CPU oracle (bit-identical forward STFT, torch autograd) 0.099 / 8.3e-3 against
        # the same golden: that is the reproducibility of the reference's own log-magnitude run, not a kernel error (the
        # kernels' gradient is pinned at 2e-4 by test_stft_distance_gradient_vs_autograd)
        assert rms_err(x, s[f"x_{tag}"]) < 2e-2 and rel(x, s[f"x_{tag}"]) < 0.2, (rms_err(x, s[f"x_{tag}"]), rel(x, s[f"x_{tag}"]))


@pytest.mark.parametrize("mode,fw,tol", [(0, "sqrt", 2e-5), (1, "linear", 2e-5), (2, "None", None)])
def test_stft_distance_gradient_vs_autograd(mode, fw, tol):
    """d D / d rec of the STFT-domain distances from the HIP kernels (+ STFT adjoint) against torch autograd through the
    oracle's restatement of utils/blind_bwe_utils.py:148-247.  Log-magnitude (mode 2): compared on a signal whose bins all sit
    well above the 1e-8 floor and the fp32 rounding floor - the only regime where that distance is well-conditioned."""
    from babe_amd.stft import STFTOps, freq_weights
    from oracle import bwe_utils as U
    gen = torch.Generator().manual_seed(3 + mode)
    L, nfft = 46046, 2048
    y = 0.1 * torch.randn(2, L, generator=gen)
    rec = (y + 0.05 * torch.randn(2, L, generator=gen)).requires_grad_(True)
    D = U.stft_distance(y, rec, nfft, weight=fw, mag=mode > 0, logmag=mode == 2)
    gref, = torch.autograd.grad(D, rec)
    ops_ = STFTOps(nfft, L, 22050, torch.device("cuda"))
    g = ops_.distance_grad(rec.detach().cuda(), y.cuda(), freq_weights(ops_.nbins, fw).cuda(), mode, shared=True)
    assert rel(g, gref) < (tol or 2e-4), rel(g, gref)


def test_blind_sampler_data_consistency():
    """posterior_sampling.data_consistency=True (blind_bwe_DC.yaml / bwe_formal_1000_DC.yaml): replacement step on the
    Tweedie estimate after every evaluation, blind and known-filter runs against the imported reference (G18)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_dc.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    args.tester.posterior_sampling.data_consistency = True
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    args.tester.blind_bwe.optimization.max_iter = int(s["max_iter"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    for i in range(3):
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3 and params_close(fp, s["filter_params"])
    it = iter(noises)
    xk = smp.predict_bwe(s["y"].cuda(), torch.tensor([[2000.0], [-40.0]]), "fc_A")
    assert rms_err(xk, s["x_known"]) < 1e-3 and rel(xk, s["x_known"]) < 2e-3


def test_blind_sampler_nfft_1024():
    """tester.blind_bwe.NFFT = 1024 (blind_bwe_cocochorales.yaml - BASELINE configs[4] - _vctk.yaml, _multislope.yaml): the
    513-bin STFT / filter design / fit / guidance kernels inside a T=3 blind run against the imported reference (G19)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sampler_nfft1024.npz")
    g, args, net = small_net(T=3, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.NFFT = int(s["nfft"])
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    for i in range(3):
        assert params_close(dfil[i], s["data_filters"][i]), (i, dfil[i], s["data_filters"][i])
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3 and params_close(fp, s["filter_params"])


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the reference's diagnostics and helper methods (tests/golden/make_golden.py::g25)

def test_compute_sweep_vs_reference_golden():
    """BlindSampler.compute_sweep (testing/blind_bwe_sampler.py:598-616): fit objective and gradient on the 15 x 12 (fc, A) grid,
    ONE launch from the per-bin statistics (babe_filter_loss_grad) against the reference's 180 autograd passes."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sweep_helpers.npz")
    g, args, net = small_net(T=3)
    smp = BlindSampler(net, EDM(args), args)
    norms, grads = smp.compute_sweep(s["den"].cuda(), s["y"].cuda())
    assert norms.shape == (15, 12) and grads.shape == (15, 12, 2)
    en = float((norms - s["norms"]).abs().max() / s["norms"].abs().max())
    eg = float((grads - s["grads"]).abs().max() / s["grads"].abs().max())
    print(f"compute_sweep: norms {en:.2e}, grads {eg:.2e} (relative to the largest entry)")
    assert en < 3e-5 and eg < 1e-4                        # (float32 filter values against the reference's float32 autograd)


def test_reference_helper_methods_vs_golden():
    """get_denoised_estimate + get_rec_grads under the reference's signature (:75-135, called as get_score_rec_guidance :136-149
    does), apply_filter_fcA (:518-520), fit_params on signals (:533-595) and move_timestep (:509-516)."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sweep_helpers.npz")
    g, args, net = small_net(T=3)
    smp = BlindSampler(net, EDM(args), args)
    x, y, t = s["x"].cuda(), s["y"].cuda(), float(s["t"])
    x_hat = smp.get_denoised_estimate(x, t)
    assert rel(x_hat, s["x_hat"]) < 1e-4
    rg = smp.get_rec_grads(x_hat, y, x, t, None, s["fp"])
    print(f"get_rec_grads: rel {rel(rg, s['rec_grads']):.2e}")
    assert rel(rg, s["rec_grads"]) < 1e-3
    # apply_filter_fcA = design_filter + apply_filter (pinned by stft_filter.npz): here against the composite of the ops
    st = smp.stft_ops(y.shape[1], y.device)
    assert torch.equal(smp.apply_filter_fcA(y, s["fp"]), st.apply_filter(y, st.design_filter(s["fp"].cuda())))
    # fit_params on signals == the spectra form the loop uses
    p0 = smp._init_params(1, y.device)
    a = smp.fit_params_signal(s["den"].cuda(), y, p0[0])
    b, _ = smp.fit_params(st.stft(s["den"].cuda()), st.stft(y), p0)
    assert torch.equal(a, b[0])
    # move_timestep: x_hat = x + sqrt(t_hat^2 - t^2) Snoise eps with the sampler's own noise source
    eps = torch.randn(1, y.shape[1], generator=torch.Generator().manual_seed(1))
    smp._randn = lambda shape, device: eps.to(device)
    xh, th = smp.move_timestep(x, torch.tensor(0.05), torch.tensor(0.2), 1.0)
    assert abs(float(th) - 0.06) < 1e-7 and rel(xh, x.cpu() + (0.06 ** 2 - 0.05 ** 2) ** 0.5 * eps) < 1e-6


def test_known_filter_run_with_the_diagnostics():
    """predict_bwe(..., 'fc_A', rid=True, test_filter_fit=True, compute_sweep=True) (predict :419-466): the diagnostics run on the
    guided Tweedie estimate of every step and leave the trajectory untouched."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    s = load("sweep_helpers.npz")
    g, args, net = small_net(T=3)
    smp = BlindSampler(net, EDM(args), args)
    y, L = s["y"].cuda(), s["y"].shape[1]
    noises = [torch.randn(1, L, generator=torch.Generator().manual_seed(40 + i)) for i in range(4)]
    outs = []
    for flags in (dict(), dict(test_filter_fit=True, compute_sweep=True)):
        it = iter(noises)
        smp._randn = lambda shape, device: next(it).to(device)
        outs.append(smp.predict_bwe(y, s["fp"], "fc_A", rid=True, **flags))
    assert len(outs[0]) == 4 and len(outs[1]) == 7
    assert torch.equal(outs[0][0], outs[1][0])                                 # same trajectory
    fits, norms, grads = outs[1][4:]
    assert fits.shape == (3, 2, 5) and norms.shape == (3, 15, 12) and grads.shape == (3, 15, 12, 2)
    assert torch.isfinite(fits).all() and torch.isfinite(norms).all() and (norms > 0).all()


def test_edm_sampler_inpainting_T3_vs_reference_golden():
    """edm_sampler.Sampler.predict_inpainting (testing/edm_sampler.py:231-243): guidance through A(x) = mask * x."""
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.edm_sampler import Sampler
    s = load("edm_sampler_inpainting.npz")
    g, args, net = small_net(T=3)
    args.tester.posterior_sampling.xi = float(s["xi"])
    args.tester.posterior_sampling.data_consistency = bool(int(s["data_consistency"]))
    args.tester.diff_params.ro = float(s["ro"])
    args.tester.diff_params.sigma_max = float(s["sigma_max"])
    args.tester.diff_params.Schurn = float(s["Schurn"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    smp = Sampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x = smp.predict_inpainting(s["y"].cuda(), s["mask"])
    print(f"inpainting: RMS err {rms_err(x, s['x']):.2e}, rel {rel(x, s['x']):.2e}")
    assert rms_err(x, s["x"]) < 1e-3 and rel(x, s["x"]) < 2e-3
    assert smp.inpaint_mask is None
