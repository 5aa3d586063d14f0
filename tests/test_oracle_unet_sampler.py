"""Oracle UNet body + sampler vs golden vectors from the imported reference (G7, G8)."""
import os

import numpy as np
import pytest
import torch

from oracle import edm as E
from oracle import unet as UN
from oracle.nsgt import CQT_nsgt
from oracle.sampler import OracleBlindSampler

G = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(num_octs=7, bins_per_oct=64, num_dils=[2, 3, 4, 5, 6, 7, 7])


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name)).items()}


def rel(a, b):
    return float((a.detach().double() - b.double()).norm() / (b.double().norm() + 1e-30))


def small_net():
    g = load("unet_small.npz")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    cqt = CQT_nsgt(7, 64, "oct", ("kaiser", 1), 22050, 92092)
    return g, sd, cqt


def test_unet_forward_and_vjp():
    g, sd, cqt = small_net()
    gen = torch.Generator().manual_seed(int(g["unet_seed"]))
    x = (0.1 * torch.randn(1, 92092, generator=gen)).requires_grad_(True)
    y = UN.unet_forward(sd, CFG, cqt, x, g["unet_cnoise"])
    assert rel(y, g["unet_y"]) < 1e-5
    wv = torch.randn(y.shape, generator=gen)
    gx, = torch.autograd.grad((y * wv).sum(), x)
    assert rel(gx, g["unet_gx"]) < 1e-4


def scaled_out(sd, sc):
    sd = dict(sd)
    for k in sd:
        if (k.startswith("middle.0.0.") or (k.startswith("ups.") and k.split(".")[2] == "0")) and \
                (k.endswith("proj_out.weight") or k.endswith("res_conv.weight")):
            sd[k] = sd[k] * sc
    return sd


def params_close(p, q):
    """fc within 1 %, A within 0.5 dB/oct.  The reference's 100-iteration projected GD (mu=[1000,10]) is
    not contractive: 1e-7 relative input perturbations move its own result by this much."""
    return bool(torch.allclose(p[0], q[0], rtol=1e-2, atol=0) and torch.allclose(p[1], q[1], rtol=0, atol=0.5))


def test_sampler_T3():
    g, sd, cqt = small_net()
    s = load("sampler_small.npz")
    a = float(s["res_a"])
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)          # the draw used for the clean signal
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]))
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    assert torch.equal(E.schedule(p, 3, float(s["start_sigma"])), s["t"])
    for i in range(3):
        assert rel(rec[i]["x_den"], s["data_denoised"][i]) < 1e-3, i
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    assert rel(x, s["x"]) < 1e-3
    assert params_close(fp, s["filter_params"])
    xk, _ = smp.predict_blind_bwe(s["y"], noises, blind=False, params=torch.tensor([[2000.0], [-40.0]]))
    assert rel(xk, s["x_known"]) < 1e-3


def test_edm_sampler_firwin_T3():
    """Config #1: known 500-tap Kaiser FIR, edm_sampler.Sampler.predict_bwe('firwin') (G9)."""
    from oracle.sampler import OracleEDMSampler
    g, sd, cqt = small_net()
    s = load("edm_sampler_firwin.npz")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, float(s["sigma_max"]), float(s["ro"]), Schurn=float(s["Schurn"]), Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleEDMSampler(net, cqt, p, audio_len=L, T=3, xi=float(s["xi"]))
    x = smp.predict_bwe(s["y"], s["taps_22050"], noises)
    assert rel(x, s["x"]) < 1e-3

@pytest.mark.parametrize("mode", ["dc", "xi0"])
def test_edm_sampler_replacement_modes_T3(mode):
    """edm_sampler.Sampler in its two other modes (G22, /root/reference/testing/edm_sampler.py:96-132): guided score +
    replacement step (posterior_sampling.data_consistency), and xi = 0 (replacement step on the plain denoised estimate)."""
    from oracle.sampler import OracleEDMSampler
    g, sd, cqt = small_net()
    s9, s = load("edm_sampler_firwin.npz"), load("edm_sampler_modes.npz")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, float(s9["sigma_max"]), float(s9["ro"]), Schurn=float(s9["Schurn"]), Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleEDMSampler(net, cqt, p, audio_len=L, T=3, xi=float(s9["xi"]) if mode == "dc" else 0.0, data_consistency=mode == "dc")
    x = smp.predict_bwe(s9["y"], s9["taps_22050"], noises)
    assert rel(x, s[f"x_{mode}"]) < 1e-3


def test_fir_taps_and_apply():
    import scipy.signal
    s = load("edm_sampler_firwin.npz")
    taps = torch.tensor(scipy.signal.firwin(numtaps=500, cutoff=1000, width=1, window="kaiser", fs=22050), dtype=torch.float32)
    assert torch.equal(taps, s["taps_22050"])
    gen = torch.Generator().manual_seed(int(s["fir_x_seed"]))
    x = torch.randn(2, 5000, generator=gen)
    y = torch.zeros_like(x)
    xp = torch.nn.functional.pad(x, (249, 250))
    for k in range(500):                                   # out[n] = sum_k taps[k] x[n+k-249]
        y += taps[k] * xp[:, k:k + 5000]
    assert rel(y, s["fir_y"]) < 1e-5


def _ar_inputs(s, L=92092):
    gen = torch.Generator().manual_seed(int(s["seed"]))
    clean = 0.1 * torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    ov = int(s["overlap"])
    mask = torch.ones(1, L)
    mask[..., ov:] = 0
    y_masked = torch.zeros(1, L)
    y_masked[..., :ov] = clean[..., :ov]
    return noises, mask, y_masked


def test_predict_bwe_AR_T3():
    """AR out-painting (mask-mixed degradation + replacement data consistency) vs the reference (G10)."""
    from oracle.sampler import predict_bwe_AR, smooth_mask
    g, sd, cqt = small_net()
    s = load("sampler_ar.npz")
    L = 92092
    noises, mask, y_masked = _ar_inputs(s)
    assert torch.equal(smooth_mask(mask, 50)[0, : int(s["overlap"]) + 8], s["smooth_mask"])
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]))
    x = predict_bwe_AR(smp, s["ylpf"], y_masked, torch.tensor([[2000.0], [-40.0]]), mask, noises)
    assert rel(x, s["x"]) < 1e-3


def test_unet_full_width_vs_reference_golden():
    """G7 at FULL width - Ns=[64,96,96,128,128,256,256], 44.1 kHz, L=46046 (1/8 segment): oracle forward and
    autograd input-gradient vs the imported reference (networks/cqtdiff+.py:730-845); pins the checker bench.py's
    cpu_baseline and the full-width GPU parity tests rely on."""
    from tests.golden_weights import FULL_DILS, full_width_sd
    g = load("unet_full_46046.npz")
    L = 46046
    sd = full_width_sd(int(g["wseed"]))
    cqt = CQT_nsgt(7, 64, "oct", ("kaiser", 1), 44100, L)
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = (0.1 * torch.randn(1, L, generator=gen)).requires_grad_(True)
    y = UN.unet_forward(sd, dict(num_octs=7, bins_per_oct=64, num_dils=FULL_DILS), cqt, x, g["cnoise"])
    assert rel(y, g["y"]) < 1e-5
    wv = torch.randn(y.shape, generator=gen)
    gx, = torch.autograd.grad((y * wv).sum(), x)
    assert rel(gx, g["gx"]) < 1e-4


def _oracle_sampler(sd, cqt, a, T, start_sigma, L=92092, mu=(1000.0, 10.0)):
    """a=None: the network unwrapped (T=35 golden); else the noisy-identity wrapper of the T=3 goldens."""
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    if a is None:
        net = lambda x, cn: UN.unet_forward(sd, CFG, cqt, x, cn)
    else:
        net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    return OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=T, start_sigma=start_sigma, mu=mu)


def test_sampler_T35_vs_reference_golden():
    """G8 at the benchmark's schedule (T=35 from sigma=0.2 down to 1e-4, 69 score evaluations): oracle vs the imported
    reference's predict_blind_bwe (testing/blind_bwe_sampler.py:619-769), per-step denoised estimates and filters."""
    g, sd, cqt = small_net()
    s = load("sampler_T35.npz")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(L, generator=gen)                      # the draw that made the observation
    noises = [torch.randn(1, L, generator=gen) for _ in range(36)]
    smp = _oracle_sampler(sd, cqt, None, 35, float(s["start_sigma"]))
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    assert torch.equal(E.schedule(smp.p, 35, float(s["start_sigma"])), s["t"])
    for i in range(35):
        assert rel(rec[i]["x_den"][:, ::16], s["data_denoised_sub16"][i]) < 1e-3, i
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    rms = float((x - s["x"]).pow(2).mean().sqrt())
    assert rms < 1e-3 and rel(x, s["x"]) < 2e-3, (rms, rel(x, s["x"]))
    assert params_close(fp, s["filter_params"])


def test_sampler_B2_reference_batch_semantics():
    """B=2 through the reference's own coupling: ONE filter fitted on the flattened batch (blind_bwe_utils.py:295) and a
    whole-batch guidance norm (blind_bwe_sampler.py:125)."""
    g, sd, cqt = small_net()
    s = load("sampler_B2.npz")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = [torch.randn(L, generator=gen) for _ in range(2)]
    noises = [torch.randn(2, L, generator=gen) for _ in range(4)]
    smp = _oracle_sampler(sd, cqt, float(s["res_a"]), 3, float(s["start_sigma"]), mu=tuple(float(v) for v in s["mu"]))
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    for i in range(3):
        assert rel(rec[i]["x_den"][:, ::16], s["data_denoised_sub16"][i]) < 1e-3, i
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    assert rel(x, s["x"]) < 1e-3 and params_close(fp, s["filter_params"])


@pytest.mark.parametrize("norm", ["cosine", "smoothl1"])
def test_sampler_alternative_guidance_distances(norm):
    """posterior_sampling.norm = 'cosine' (conf/tester/blind_bwe_cossim.yaml) / 'smoothl1' (get_rec_grads :99-103): the
    oracle's single guidance term and its T=3 blind run against the imported reference (G16)."""
    g, sd, cqt = small_net()
    s = load("sampler_altnorm.npz")
    L = 92092
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]), norm=norm,
                             smoothl1_beta=float(s["smoothl1_beta"]), mu=tuple(float(v) for v in s["mu"]))
    x0 = s[f"rg_x0_{norm}"].clone().requires_grad_(True)
    rg = smp.rec_grads(smp.denoised(x0, torch.tensor(0.04)), s["y"], x0, torch.tensor(0.04), torch.tensor([[2000.0], [-40.0]]))
    assert rel(rg, s[f"rg_{norm}"]) < 1e-4
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    for i in range(3):
        assert params_close(rec[i]["params"], s[f"data_filters_{norm}"][i]), i
    assert rel(x, s[f"x_{norm}"]) < 1e-3 and params_close(fp, s[f"filter_params_{norm}"])


@pytest.mark.parametrize("tag,kw", [("logmag", dict(nfft=2048, weight="None", mag=True, logmag=True)),
                                    ("complex", dict(nfft=2048, weight="sqrt", mag=False, logmag=False))])
def test_sampler_stft_domain_guidance_distances(tag, kw):
    """posterior_sampling.stft_distance.use (get_rec_grads :105-115; conf/tester/blind_bwe_2.yaml is the log-magnitude
    variant): the oracle's guidance term and T=3 blind run against the imported reference (G17)."""
    g, sd, cqt = small_net()
    s = load("sampler_stftdist.npz")
    L = 92092
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]),
                             mu=tuple(float(v) for v in s["mu"]), stft_distance=kw)
    x0 = s[f"rg_x0_{tag}"].clone().requires_grad_(True)
    rg = smp.rec_grads(smp.denoised(x0, torch.tensor(0.04)), s["y"], x0, torch.tensor(0.04), torch.tensor([[2000.0], [-40.0]]))
    # the log-magnitude distance is ill-conditioned (1/(|S| + 1e-8) on bins at the rounding floor): its guidance term depends on
    # the rounding of the backward pass at the 1 % level even with a bit-identical forward STFT
    assert rel(rg, s[f"rg_{tag}"]) < (2e-2 if tag == "logmag" else 1e-4)
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    if tag == "logmag":
        # measured 0.099: the reference's own log-magnitude run is reproducible to ~10 % only (see above)
        assert rel(x, s[f"x_{tag}"]) < 0.2 and torch.allclose(fp[0], s[f"filter_params_{tag}"][0], rtol=0.05)
        return
    for i in range(3):
        assert params_close(rec[i]["params"], s[f"data_filters_{tag}"][i]), i
    assert rel(x, s[f"x_{tag}"]) < 1e-3 and params_close(fp, s[f"filter_params_{tag}"])


def test_sampler_observation_noise_regularisation():
    """posterior_sampling.SNR_observations = 50 (conf/tester/blind_bwe_2.yaml) + blind_bwe.sigma_den_estimate: y += sqrt(var(y)/snr)
    randn IN PLACE before every fit (:542-548) and every guidance evaluation (:80-86), the fit on a noisy denoised estimate
    (:551-552); 19 draws in the reference's order (G23)."""
    g, sd, cqt = small_net()
    s = load("sampler_obs_noise.npz")
    L = 92092
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]),
                             mu=tuple(float(v) for v in s["mu"]), SNR_observations=float(s["snr_db"]),
                             sigma_den_estimate=float(s["sigma_den"]))
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(L, generator=gen)                      # the draw that made the observation
    noises = [torch.randn(1, L, generator=gen) for _ in range(int(s["ndraw"]))]
    rec = []
    y = s["y"].clone()
    x, fp = smp.predict_blind_bwe(y, noises, record=rec)
    assert torch.equal(y, s["y"])                          # (the oracle works on a copy; the reference mutated its argument:)
    assert float((s["y_after"] - s["y"]).pow(2).mean().sqrt()) > 0
    for i in range(3):
        assert rel(rec[i]["x_den"][:, ::16], s["data_denoised_sub16"][i]) < 1e-3, i
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    assert rel(x, s["x"]) < 1e-3 and params_close(fp, s["filter_params"])


def test_sampler_data_consistency():
    """posterior_sampling.data_consistency=True (conf/tester/blind_bwe_DC.yaml, bwe_formal_1000_DC.yaml): the replacement step
    of data_consistency_step_classic :63-73 after every score evaluation; blind and known-filter T=3 runs (G18)."""
    g, sd, cqt = small_net()
    s = load("sampler_dc.npz")
    L = 92092
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]),
                             mu=tuple(float(v) for v in s["mu"]), max_iter=int(s["max_iter"]), data_consistency=True)
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    for i in range(3):
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    assert rel(x, s["x"]) < 1e-3 and params_close(fp, s["filter_params"])
    xk, _ = smp.predict_blind_bwe(s["y"], noises, blind=False, params=torch.tensor([[2000.0], [-40.0]]))
    assert rel(xk, s["x_known"]) < 1e-3


def test_sampler_nfft_1024():
    """tester.blind_bwe.NFFT = 1024 (blind_bwe_cocochorales.yaml, _vctk.yaml, ...): 513-bin filter fit / degradation (G19)."""
    g, sd, cqt = small_net()
    s = load("sampler_nfft1024.npz")
    L = 92092
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=22050, audio_len=L, T=3, start_sigma=float(s["start_sigma"]),
                             mu=tuple(float(v) for v in s["mu"]), nfft=int(s["nfft"]))
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    rec = []
    x, fp = smp.predict_blind_bwe(s["y"], noises, record=rec)
    for i in range(3):
        assert params_close(rec[i]["params"], s["data_filters"][i]), i
    assert rel(x, s["x"]) < 1e-3 and params_close(fp, s["filter_params"])


def test_sampler_full_width_vs_reference_golden():
    """G20: the benchmarked composition at FULL width (Ns=[64,96,96,128,128,256,256], 44.1 kHz, L=46046, T=3) - oracle vs the
    imported reference's predict_blind_bwe run at B = 1 (clip 0 of tests/golden/sampler_full_46046.npz; the GPU test runs
    both clips as one per-clip batch on two stream lanes)."""
    from tests.golden_weights import FULL_DILS, full_width_sd
    s = load("sampler_full_46046.npz")
    L, T, a = int(s["L"]), int(s["T"]), float(s["res_a"])
    sd = full_width_sd(int(s["wseed"]))
    cqt = CQT_nsgt(7, 64, "oct", ("kaiser", 1), 44100, L)
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=FULL_DILS)
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10, Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, cfg, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleBlindSampler(net, cqt, p, fs=44100, audio_len=L, T=T, start_sigma=float(s["start_sigma"]),
                             mu=tuple(float(v) for v in s["mu"]))
    noises = [s["noises0"][i:i + 1] for i in range(T + 1)]
    rec = []
    x, fp = smp.predict_blind_bwe(s["y0"], noises, record=rec)
    assert torch.equal(E.schedule(p, T, float(s["start_sigma"])), s["t"])
    for i in range(T):
        assert rel(rec[i]["x_den"], s["den0"][i]) < 1e-3, i
        assert params_close(rec[i]["params"], s["filt0"][i]), i
    rms = float((x - s["x0"]).pow(2).mean().sqrt())
    assert rms < 1e-3 and rel(x, s["x0"]) < 1e-3, (rms, rel(x, s["x0"]))
    assert params_close(fp, s["fp0"])


def test_edm_sampler_inpainting_T3():
    """edm_sampler.Sampler.predict_inpainting (masking degradation, /root/reference/testing/edm_sampler.py:231-243; G26)."""
    from oracle.sampler import OracleEDMSampler, edm_predict_inpainting
    g, sd, cqt = small_net()
    s = load("edm_sampler_inpainting.npz")
    L = 92092
    gen = torch.Generator().manual_seed(int(s["seed"]))
    _ = torch.randn(1, L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(4)]
    a = float(s["res_a"])
    p = E.EDMParams(0.063, 1e-4, float(s["sigma_max"]), float(s["ro"]), Schurn=float(s["Schurn"]), Stmin=0, Stmax=50, Snoise=1.0)
    net = lambda x, cn: a * UN.unet_forward(sd, CFG, cqt, x, cn) + (torch.exp(4 * cn) / 0.063) * x
    smp = OracleEDMSampler(net, cqt, p, audio_len=L, T=3, xi=float(s["xi"]), data_consistency=bool(int(s["data_consistency"])))
    x = edm_predict_inpainting(smp, s["y"], s["mask"], noises)
    assert rel(x, s["x"]) < 1e-3
