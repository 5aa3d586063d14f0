"""Whole-recording flows on the HIP path vs the reference's OWN driver method,
BlindTester.test_real_blind_bwe_complete (/root/reference/testing/denoise_and_bwe_tester.py:248-411; identical to
testing/blind_bwe_tester.py:710-867 without the denoiser), run in the build container with file I/O stubbed
(tests/golden/make_golden.py::g15): [denoiser pre-pass ->] std normalisation -> blind filter estimate on 2 random
segments in one batch with the reference's batch coupling -> non-blind autoregressive pass with that filter.  Needs a MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rms_err(a, b):
    return float((a.detach().double().cpu() - torch.as_tensor(b).double().cpu()).pow(2).mean().sqrt())


class ResidualNet:
    """Same wrapper as make_golden.ResidualNetRef: a*net(x,c) + (sigma/sigma_data)*x, sigma = exp(4c)."""

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


def synth_recording(L, fs, seed):
    """The input file of g15, re-derived from its seed (oracle utilities are the checker's, not the product's)."""
    from oracle import bwe_utils as U
    g = torch.Generator().manual_seed(seed)
    t_ax = torch.arange(L) / fs
    clean = sum(0.05 / (k + 1) * torch.sin(2 * np.pi * 196.0 * (k + 1) * t_ax) * torch.exp(-(t_ax % 2.0) * (1 + k)) for k in range(12))
    clean = clean + 0.1 * torch.randn(L, generator=g)
    f = torch.fft.rfftfreq(4096, d=1 / fs)
    return U.apply_filter(clean[None], U.design_filter(torch.tensor([2500.0]), torch.tensor([-35.0]), f), 4096)[0]


@pytest.mark.parametrize("use_denoiser", [False, True])
def test_complete_recording_flow_vs_reference_driver(use_denoiser):
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks import denoiser as dn
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.denoise import DenoiserPrepass
    from babe_amd.testing.long_file import restore_recording_complete
    s = np.load(os.path.join(G, "complete_recording.npz"))
    u = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "unet_small.npz")).items()}
    sd = {k[3:]: v for k, v in u.items() if k.startswith("sd.")}
    fs, segL, L = 22050, 92092, int(s["L"])
    args = default_args(sample_rate=fs, audio_len=segL, Ns=[8, 8, 8, 8, 16, 16, 16], T=3, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(sd, strict=True)
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args, batch_semantics="per_clip")
    gn = torch.Generator().manual_seed(int(s["noise_seed"]))
    smp._randn = lambda shape, device: torch.randn(*shape, generator=gn).to(device)      # the reference's draw order
    pre = None
    if use_denoiser:
        cfg = dict(depth=3, num_tfc=1, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
        dnet = dn.MultiStage_denoise(cfg)
        dnet.load_state_dict(dn.init_state_dict(cfg, seed=int(s["dn_seed"])))
        dnet.to("cuda")
        pre = DenoiserPrepass(dnet, dict(sample_rate_denoiser=fs, segment_size=5, stft_win_size=1024, stft_hop_size=256,
                                         num_stages=2), "cuda")
    rec = synth_recording(L, fs, int(s["seed"])).cuda()
    np.random.seed(int(s["np_seed"]))                         # the reference picks the blind-step segments with np.random
    out, filt, blind_pred = restore_recording_complete(smp, rec, n_segments_blindstep=2, ix_start=0, std=0.1, overlap_s=0.25,
                                                        typefilter="fc_A", denoiser=pre)
    key = "dn" if use_denoiser else "plain"
    assert smp.batch_semantics == "per_clip"                  # restored after the coupled blind step
    if not use_denoiser:
        # A SECOND recording on the same sampler: predict_bwe_AR flipped smp.data_consistency for good (the reference does,
        # blind_bwe_sampler.py:300), but the blind loop reads the CONFIG value (:704, :748), so the blind estimate of the
        # next recording must not pick up a replacement step.  Same seeds -> identical results.
        assert smp.data_consistency is True and smp._dc_cfg is False
        gn.manual_seed(int(s["noise_seed"]))
        np.random.seed(int(s["np_seed"]))
        out2, filt2, blind_pred2 = restore_recording_complete(smp, rec, n_segments_blindstep=2, ix_start=0, std=0.1,
                                                              overlap_s=0.25, typefilter="fc_A", denoiser=None)
        assert rel(blind_pred2, blind_pred) < 1e-5 and rel(filt2, filt) < 1e-5
        # (the AR pass of the second recording DOES differ, as it would in the reference: its first segment goes through the
        # known-filter predict_bwe, whose loop reads the flipped attribute, blind_bwe_sampler.py:178, :360-362)
        assert bool(torch.isfinite(out2).all()) and out2.shape == out.shape
    assert rel(blind_pred[:, ::16], s[f"{key}_blind_pred_sub16"]) < 2e-3
    fr = torch.from_numpy(s[f"{key}_blind_filter"])
    assert torch.allclose(filt.cpu()[0], fr[0], rtol=1e-2) and torch.allclose(filt.cpu()[1], fr[1], atol=1.0), (filt, fr)
    ref = torch.from_numpy(s[f"{key}_final"])[0]
    e_rms, e_rel = rms_err(out, ref), rel(out, ref)
    print(f"complete recording ({key}): RMS err {e_rms:.2e} (signal RMS {float(ref.std()):.3f}), rel {e_rel:.2e}")
    assert out.shape == ref.shape and e_rel < 3e-3

def test_formal_test_bwe_non_ar_flow_vs_reference_driver():
    """SURVEY 8f row 1, non-AR path: restore_file (segments of audio_len every segL - 200 - OLA samples, blind restoration of
    each, Hann cross-fade) against the reference's OWN BlindTester.formal_test_bwe(typefilter='fc_A', blind=True) run with its
    file I/O stubbed (tests/golden/make_golden.py::g21; /root/reference/testing/blind_bwe_tester.py:320-578).  The reference
    restores its segments one at a time, drawing noise as it goes: batch_size = 1 with the same generator reproduces its draw
    order; a second run with all three segments in ONE per-clip batch (what the benchmark and the multi-GPU sharding do) draws
    different noise, so it is checked for structure only (finite, same loudness)."""
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.long_file import restore_file
    s = np.load(os.path.join(G, "formal_test_bwe.npz"))
    u = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "unet_small.npz")).items()}
    sd = {k[3:]: v for k, v in u.items() if k.startswith("sd.")}
    fs, segL, L = 22050, 92092, int(s["L"])
    args = default_args(sample_rate=fs, audio_len=segL, Ns=[8, 8, 8, 8, 16, 16, 16], T=3, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(sd, strict=True)
    smp = BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args, batch_semantics="per_clip")
    gn = torch.Generator().manual_seed(int(s["noise_seed"]))
    smp._randn = lambda shape, device: torch.randn(*shape, generator=gn).to(device)
    y = torch.from_numpy(s["degraded"])[0].cuda()
    out, filters = restore_file(smp, y, batch_size=1, blind=True)
    ref = torch.from_numpy(s["final"])[0]
    e_rms, e_rel = rms_err(out, ref), rel(out, ref)
    print(f"formal_test_bwe (non-AR, blind, 3 segments): RMS err {e_rms:.2e} (signal RMS {float(ref.std()):.3f}), rel {e_rel:.2e}")
    assert out.shape == ref.shape and e_rms < 1e-3 and e_rel < 3e-3
    assert [span[0] for span, _ in filters] == [int(v) for v in s["seg_starts"]]
    fr = torch.from_numpy(s["seg_filters"])
    for (_, f), r in zip(filters, fr):
        assert torch.allclose(f.cpu()[0], r[0], rtol=1e-2) and torch.allclose(f.cpu()[1], r[1], atol=1.0), (f, r)
    out_b, _ = restore_file(smp, y, batch_size=8, blind=True)
    assert out_b.shape == ref.shape and bool(torch.isfinite(out_b).all()) and 0.5 < float(out_b.std()) / float(ref.std()) < 2.0


@pytest.mark.parametrize("layout", ["trainer_ema", "ema_weights_all", "ema_weights_trainable"])
def test_reference_format_checkpoint_restores_into_hip_net(layout, tmp_path):
    """a checkpoint written the way the reference's trainer writes it (training/trainer.py:273-285:
    {'it','network','optimizer','ema','args'}) or in the two older 'ema_weights' list layouts that
    BlindTester.load_checkpoint (testing/blind_bwe_tester.py:238-272) accepts is restored through io.load_checkpoint into
    the HIP network, which must then reproduce the imported reference's forward (tests/golden/unet_small.npz)."""
    from babe_amd.config import default_args
    from babe_amd.io import load_checkpoint
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    u = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "unet_small.npz")).items()}
    ema = {k[3:]: v for k, v in u.items() if k.startswith("sd.")}
    Ns = [8, 8, 8, 8, 16, 16, 16]
    args = default_args(sample_rate=22050, audio_len=92092, Ns=Ns)
    raw = init_state_dict(Ns, args.network.num_dils, seed=123)            # the non-EMA weights: must NOT be what gets loaded
    buffers = {k for k in raw if k.endswith(".kernel") or k == "embedding.RFF_freq"}
    model = {k: (raw[k].clone().requires_grad_(k not in buffers) if k not in buffers else ema[k].clone()) for k in ema}
    if layout == "trainer_ema":
        state = {"it": 4321, "network": raw, "optimizer": {"state": {}, "param_groups": []}, "ema": ema, "args": {"exp": "x"}}
    elif layout == "ema_weights_all":
        state = {"it": 4321, "model": model, "ema_weights": [ema[k] for k in model]}
    else:
        state = {"it": 4321, "model": model, "ema_weights": [ema[k] for k in model if k not in buffers]}
    path = str(tmp_path / "ckpt.pt")
    torch.save(state, path)
    net = Unet_CQT_oct_with_attention(args, "cuda")
    assert load_checkpoint(net, path) == 4321
    gen = torch.Generator().manual_seed(int(u["unet_seed"]))
    x = (0.1 * torch.randn(1, 92092, generator=gen)).cuda()
    y = net(x, u["unet_cnoise"].cuda())
    assert rel(y, u["unet_y"]) < 2e-5


def test_config5_geometry_denoise_then_bwe_bf16_runs():
    """configs[4] of BASELINE.json as ONE flow at its own geometry (conf/exp/CocoChorales_16k_8s.yaml: 16 kHz, audio_len
    184184 = 11.5 s; conf/tester/blind_bwe_denoise_brass.yaml: sigma_data 0.15, sigma_max 2, rho 9, start_sigma 0.6): a
    30 s recording -> denoiser pre-pass -> blind estimate on 2 random segments -> AR bandwidth extension, bf16 conv
    arithmetic.  No reference output exists at this size (the 30 s flow takes hours on the CPU), so the checks are the
    size-independent ones: finite, length preserved, loudness bounded, filter inside its constraint set."""
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks import denoiser as dn
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from babe_amd.stft import STFTOps
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.denoise import DenoiserPrepass
    from babe_amd.testing.long_file import restore_recording_complete
    fs, segL, L = 16000, 184184, 480000
    Ns = [8, 8, 8, 8, 16, 16, 16]
    # start_sigma 0.05 instead of the config's 0.6: with an UNTRAINED network and T = 2 the replacement data-consistency
    # step of the AR mode at t' = sigma_min divides an O(0.1) mismatch in the known region by 1e-4 (the reference does the
    # same); near the observation the trajectory stays consistent and the flow's plumbing is what is exercised
    args = default_args(sample_rate=fs, audio_len=segL, Ns=Ns, T=2, start_sigma=0.05)
    dpar = args.tester.diff_params
    dpar.sigma_data, dpar.sigma_max, dpar.ro, dpar.Schurn = 0.15, 2.0, 9, 5
    net = Unet_CQT_oct_with_attention(args, "cuda", precision="bf16")
    net.load_state_dict(init_state_dict(Ns, args.network.num_dils, seed=1, gate_scale=1.0))
    smp = BlindSampler(ResidualNet(net, 0.3, 0.15), EDM(args), args, noise_device="cuda")
    cfg = dict(depth=3, num_tfc=1, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
    dnet = dn.MultiStage_denoise(cfg)
    dnet.load_state_dict(dn.init_state_dict(cfg, seed=3))
    dnet.to("cuda")
    # the denoiser runs at ITS rate, 22.05 kHz (conf/tester/blind_bwe_denoise_brass.yaml:160), the model at 16 kHz
    # (conf/exp/CocoChorales_16k_8s.yaml:51): the recording goes fs -> 22050 -> denoiser -> 16000 -> BWE as
    # testing/denoise_and_bwe_tester.py:279-289 does (babe_amd/resample.py)
    srd = 22050
    pre = DenoiserPrepass(dnet, dict(sample_rate_denoiser=srd, segment_size=5, stft_win_size=1024, stft_hop_size=256,
                                     num_stages=2), "cuda")
    g = torch.Generator().manual_seed(5)
    t_ax = torch.arange(L) / fs
    rec = sum(0.1 / (k + 1) * torch.sin(2 * np.pi * 233.0 * (k + 1) * t_ax) for k in range(6)) + 0.02 * torch.randn(L, generator=g)
    torch.manual_seed(0)
    np.random.seed(1)
    out, filt, _ = restore_recording_complete(smp, rec.cuda(), n_segments_blindstep=2, ix_start=0, std=0.15, overlap_s=0.25,
                                              typefilter="fc_A", denoiser=pre, fs=fs, sample_rate_denoiser=srd)
    torch.cuda.synchronize()
    from babe_amd.resample import resample, resampled_length
    assert out.shape == (resampled_length(resampled_length(L, fs, srd), srd, fs),) and abs(out.shape[0] - L) <= 1
    assert bool(torch.isfinite(out).all())
    fc, A = filt[0].cpu(), filt[1].cpu()
    assert bool((fc >= 20).all()) and bool((fc <= fs / 2).all()) and bool((fc[1:] >= fc[:-1] + 1 - 1e-3).all())
    assert bool((A <= -1 + 1e-5).all()) and bool((A >= -50 - 1e-5).all()) and bool((A[1:] <= A[:-1] + 1e-5).all())
    ratio = float(out.std()) / float(pre.apply_denoiser(resample(rec.cuda().unsqueeze(0), fs, srd)).std())
    print(f"config-5 flow: output/denoised loudness ratio {ratio:.2f} (random weights), filter fc {fc.tolist()} A {A.tolist()}")
    assert 0.01 < ratio < 100.0                      # (an untrained network + sigma 0.6 start: only boundedness is meaningful)


def _config5_flow(T, start_sigma, Ns=(64, 96, 96, 128, 128, 256, 256), precision="bf16", dn_cfg=None, L=480000, fs_file=16000,
                  seed=5, timing=None):
    """BASELINE configs[4] on one GPU, at the sizes its YAMLs give: model 16 kHz / audio_len 184184 (conf/exp/
    CocoChorales_16k_8s.yaml:51-52), sigma_data 0.15, sigma_max 2, rho 9 (conf/tester/blind_bwe_denoise_brass.yaml:58-63),
    denoiser at 22.05 kHz with the shipped depth 6 / num_tfc 3 (same file, `denoiser:` node), 30 s file."""
    import time
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks import denoiser as dn
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.denoise import DenoiserPrepass
    from babe_amd.testing.long_file import restore_recording_complete
    fs, segL, srd = 16000, 184184, 22050
    args = default_args(sample_rate=fs, audio_len=segL, Ns=list(Ns), T=T, start_sigma=start_sigma)
    dpar = args.tester.diff_params
    dpar.sigma_data, dpar.sigma_max, dpar.ro, dpar.Schurn = 0.15, 2.0, 9, 5
    net = Unet_CQT_oct_with_attention(args, "cuda", precision=precision)
    net.load_state_dict(init_state_dict(list(Ns), args.network.num_dils, seed=1, gate_scale=1.0))
    smp = BlindSampler(ResidualNet(net, 0.3, 0.15), EDM(args), args, noise_device="cuda")
    cfg = dn_cfg or dict(depth=6, num_tfc=3, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
    dnet = dn.MultiStage_denoise(cfg)
    dnet.load_state_dict(dn.init_state_dict(cfg, seed=3))
    dnet.to("cuda")
    pre = DenoiserPrepass(dnet, dict(sample_rate_denoiser=srd, segment_size=5, stft_win_size=1024, stft_hop_size=256,
                                     num_stages=2), "cuda")
    g = torch.Generator().manual_seed(seed)
    t_ax = torch.arange(L) / fs_file
    rec = sum(0.1 / (k + 1) * torch.sin(2 * np.pi * 233.0 * (k + 1) * t_ax) for k in range(6)) + 0.02 * torch.randn(L, generator=g)
    rec = rec.cuda()
    runs = 2 if timing is not None else 1
    for r in range(runs):                                  # (timing: the second run, tables / packed weights / allocator warm)
        torch.manual_seed(0)
        np.random.seed(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, filt, pred = restore_recording_complete(smp, rec, n_segments_blindstep=2, ix_start=0, std=0.15, overlap_s=0.25,
                                                     typefilter="fc_A", denoiser=pre, fs=fs_file, sample_rate_denoiser=srd)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if timing is not None:
        timing["seconds"] = dt
        timing["audio_seconds"] = L / fs_file
    return out, filt, pred, pre, rec, net


def test_config5_full_width_bf16_at_its_real_size():
    """configs[4] ONCE at its real size: the FULL-width bf16 network (Ns = [64, 96, 96, 128, 128, 256, 256]) on
    184184-sample 16 kHz segments with the config's sigma_data 0.15 / sigma_max 2 / rho 9 and its start_sigma 0.6, the full
    denoiser (depth 6, num_tfc 3) at 22.05 kHz with both rate conversions, blind step on 2 segments, AR pass over the 30 s file
    (3 segments); T = 2.  No reference output can exist at this size (hours on the CPU, weights unavailable): the checks are the
    size-independent ones - finite, length rule of the two conversions, loudness bounded, filter inside its constraint set, the
    bf16 conv kernels actually dispatched."""
    from babe_amd import _lib
    from babe_amd.resample import resample, resampled_length
    _lib.dispatch_counts(reset=True)
    L, fs = 480000, 16000
    out, filt, pred, pre, rec, net = _config5_flow(T=2, start_sigma=0.6)
    counts = _lib.dispatch_counts()
    assert out.shape == (resampled_length(resampled_length(L, fs, 22050), 22050, fs),) and abs(out.shape[0] - L) <= 1
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(pred).all()) and pred.shape == (2, 184184)
    fc, A = filt[0].cpu(), filt[1].cpu()
    assert bool((fc >= 20).all()) and bool((fc <= fs / 2).all()) and bool((fc[1:] >= fc[:-1] + 1 - 1e-3).all())
    assert bool((A <= -1 + 1e-5).all()) and bool((A >= -50 - 1e-5).all()) and bool((A[1:] <= A[:-1] + 1e-5).all())
    assert counts["conv_bf16p"] + counts["conv_bf16"] > 1000 and counts["conv53_wino45"] == 0, counts     # bf16 arithmetic, full width
    assert counts["denoiser"] > 100, counts
    den = pre.apply_denoiser(resample(rec.unsqueeze(0), fs, 22050))
    ratio = float(out.std()) / float(den.std())
    print(f"config-5 full-size flow: output/denoised loudness ratio {ratio:.3g} (random weights), fc {fc.tolist()} A {A.tolist()}")
    # With an UNTRAINED network, the config's start_sigma 0.6 and T = 2, the AR mode's replacement data-consistency step at
    # t' = sigma_min divides an O(0.1) mismatch in the known region by 1e-4 (the reference does the same): the output is finite but
    # loud (measured 1.2e3 x the denoised input on MI355X); only finiteness and a sanity ceiling are asserted.
    assert 1e-3 < ratio < 1e6
