"""Full-width parity of the BENCHMARKED network: HIP UNet (+CQT) forward and input-VJP at
Ns=[64,96,96,128,128,256,256], 44.1 kHz, vs outputs of the imported reference (tests/golden/make_golden.py::g12,
/root/reference/networks/cqtdiff+.py:730-845) - L=46046 and the benchmark's own segment L=368368 - with the conv
dispatch asserted: every (5,3) layer of the 368368 geometry must run on conv_wino4_kernel (the kernels that are ~70 % of
bench.py's GPU time), otherwise the test would silently exercise the F(2,3) / direct fallbacks.  Needs a MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

# tolerances (relative L2): same bars as the reduced-width goldens (tests/test_gpu_sampler.py)
TOL_FWD, TOL_VJP = 2e-5, 2e-4


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name)).items()}


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


_NETS = {}


def full_net(L, precision="f32"):
    """One full-width HIP network per (length, precision) for the whole module (packing 1.8 GB of Winograd weights takes
    seconds)."""
    if (L, precision) not in _NETS:
        from babe_amd.config import default_args
        from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
        from tests.golden_weights import full_width_sd
        _NETS.clear()                                   # keep one geometry resident at a time
        args = default_args(sample_rate=44100, audio_len=L)
        net = Unet_CQT_oct_with_attention(args, "cuda", precision=precision)
        net.load_state_dict(full_width_sd(0), strict=True)
        _NETS[(L, precision)] = net
    return _NETS[(L, precision)]


def run_fwd_vjp(net, g, L, B=1):
    from babe_amd._lib import dispatch_counts
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = (0.1 * torch.randn(1, L, generator=gen)).cuda()
    cn = g["cnoise"].cuda()
    dispatch_counts(reset=True)
    y = net.fwd_nograd(x.expand(B, L).contiguous(), cn.expand(B, 1).contiguous())
    torch.cuda.synchronize()
    cf = dispatch_counts(reset=True)
    wv = torch.randn(1, L, generator=gen).cuda()
    gx = net.vjp(wv.expand(B, L).contiguous())
    torch.cuda.synchronize()
    cb = dispatch_counts(reset=True)
    return y, gx, cf, cb


def test_full_width_L46046_vs_reference_golden():
    g = load("unet_full_46046.npz")
    L = 46046
    y, gx, cf, cb = run_fwd_vjp(full_net(L), g, L)
    ey, eg = rel(y, g["y"]), rel(gx, g["gx"])
    print(f"full width L={L}: fwd rel {ey:.2e}, vjp rel {eg:.2e}; dispatch fwd {cf} vjp {cb}")
    assert ey < TOL_FWD and eg < TOL_VJP
    # frame counts 512 ... 8: the T=8 layers (enc6, the two pyramid convs that run at 8 frames, mid, dec6 = 23 convs) are
    # below F(4,3)'s 16-frame minimum and run the direct kernel; everything else must be on the F(4,3) kernel
    assert cf["conv53_wino4"] + cf["conv53_wino45"] + cf["conv53_wino85"] + cf["conv53_direct"] == 82 and cf["conv53_direct"] <= 23 and cf["conv53_wino2"] == 0, cf
    assert cf["conv_bf16"] == 0 and cb["conv_bf16"] == 0


def test_full_width_L368368_vs_reference_golden_all_wino4():
    """The benchmark's segment geometry (conf/exp/maestro44k_8s.yaml:51-52): frames 4096 ... 64."""
    g = load("unet_full_368368.npz")
    L = 368368
    y, gx, cf, cb = run_fwd_vjp(full_net(L), g, L)
    ey, eg = rel(y, g["y"]), rel(gx, g["gx"])
    print(f"full width L={L}: fwd rel {ey:.2e}, vjp rel {eg:.2e}; dispatch fwd {cf} vjp {cb}")
    assert ey < TOL_FWD and eg < TOL_VJP
    # forward: 75 dilated ResnetBlock convs + 7 pyramid projections, all on the Winograd kernels: the nested
    # F(2,5) x F(4,3) kernel where its tiles are full (64 / 128 / 256 channels), conv_wino4p elsewhere (96 channels, the
    # 2-channel pyramid inputs, the high dilations of the 320- and 384-bin levels)
    # (round 5: the 128- and 256-channel layers whose row quads are at least 80 % full take the F(4,5) x F(4,3) kernel)
    assert cf["conv53_wino4"] + cf["conv53_wino45"] + cf["conv53_wino85"] == 82 and cf["conv53_wino45"] + cf["conv53_wino85"] >= 40, cf
    assert cf["conv53_wino85"] >= 25 and cb["conv53_wino85"] >= 25, (cf, cb)
    assert cf["conv53_wino2"] == 0 and cf["conv53_direct"] == 0, cf
    # VJP: the 75 dilated convs on F(4,3); the 7 pyramid projections transposed have 2 output channels and run on the
    # vector-ALU kernel (csrc/conv_fewco.hip)
    assert cb["conv53_wino4"] + cb["conv53_wino45"] + cb["conv53_wino85"] == 75 and cb["conv53_wino45"] + cb["conv53_wino85"] >= 40, cb
    assert cb["conv53_fewco"] == 7 and cb["conv53_wino2"] == 0 and cb["conv53_direct"] == 0, cb
    assert cf["conv_bf16"] == 0 and cb["conv_bf16"] == 0


def test_full_width_two_lanes_equal_single_stream_bit_exact():
    """bench.py runs the two segments of a clip on two HIP streams (one engine state per lane over shared weights):
    the lanes must give bit-identical results to the same batch on one stream, and each row must equal the B=1 run
    (same kernels per batch item), hence the reference golden."""
    g = load("unet_full_368368.npz")
    L = 368368
    net = full_net(L)
    keep = net.MAX_LANES
    try:
        net.MAX_LANES = 2
        y2, g2, _, _ = run_fwd_vjp(net, g, L, B=2)
        net.MAX_LANES = 1
        y1, g1, cf, cb = run_fwd_vjp(net, g, L, B=2)
    finally:
        net.MAX_LANES = keep
    assert torch.equal(y2, y1) and torch.equal(g2, g1)
    assert torch.equal(y1[0], y1[1]) and torch.equal(g1[0], g1[1])
    assert rel(y1[:1], g["y"]) < TOL_FWD and rel(g1[:1], g["gx"]) < TOL_VJP
    assert cf["conv53_wino4"] + cf["conv53_wino45"] + cf["conv53_wino85"] == 82
    assert cb["conv53_wino4"] + cb["conv53_wino45"] + cb["conv53_wino85"] == 75


# ---- reduced-precision builds at FULL width.  The reference is fp32-only; 'bf16x3' (hi/lo split, three
# bf16 products per multiply) and 'bf16' (plain bf16 operands, fp32 accumulate) are this build's opt-in modes and carry their
# own dtype strings in bench.py.  Stated tolerances (relative L2 against the imported reference's fp32 outputs):
#   bf16x3: forward 1e-4, input-VJP 1e-3      bf16: forward 2e-2, input-VJP 6e-2
# (bf16 has an 8-bit mantissa: 2^-9 = 2e-3 per operand, growing over the ~150 conv layers of a forward + VJP chain.)
RP_TOL = {"bf16x3": (1e-4, 1e-3), "bf16": (2e-2, 6e-2)}


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("L", [46046, 368368])
def test_full_width_reduced_precision_vs_reference_golden(precision, L):
    g = load(f"unet_full_{L}.npz")
    y, gx, cf, cb = run_fwd_vjp(full_net(L, precision), g, L)
    ey, eg = rel(y, g["y"]), rel(gx, g["gx"])
    print(f"full width L={L} precision={precision}: fwd rel {ey:.2e}, vjp rel {eg:.2e}; dispatch fwd {cf} vjp {cb}")
    ty, tg = RP_TOL[precision]
    assert ey < ty and eg < tg
    # the (5,3) layers must have run on the bf16 MFMA kernels (pipelined kernel for plain bf16), not on an fp32 fallback
    n_bf16 = cf["conv_bf16"] + cf["conv_bf16p"]
    # (the 7 two-input-channel pyramid projections stay on the exact fp32 kernels in every mode: ops.PackedConv)
    assert n_bf16 >= 75 and cf["conv53_wino4"] + cf["conv53_direct"] <= 7 and cf["conv53_wino2"] == 0 and cf["conv53_wino45"] == 0, cf
    assert cb["conv_bf16"] + cb["conv_bf16p"] >= 75 and cb["conv53_wino4"] == 0, cb
    if precision == "bf16" and L == 368368:
        assert cf["conv_bf16p"] >= 75 and cb["conv_bf16p"] >= 75, (cf, cb)     # every dilated layer on conv_bf16p_kernel


# ---- the benchmarked COMPOSITION at full width (VERDICT r2 missing #3): predict_blind_bwe, two clips as one per-clip
# batch on two stream lanes, against two B = 1 runs of the imported reference (make_golden.py::g20).
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
        self.k = float(torch.exp(4 * cn[0, 0])) / self.sd
        kw = {} if lane is None else {"lane": lane}
        return self.a * self.inner.fwd_nograd(x, cn, **kw) + self.k * x

    def vjp(self, g, lane=None):
        kw = {} if lane is None else {"lane": lane}
        return self.a * self.inner.vjp(g, **kw) + self.k * g


def _full_sampler(net, s, precision_tag="f32"):
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    L, T = int(s["L"]), int(s["T"])
    args = default_args(sample_rate=44100, audio_len=L, T=T, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    return BlindSampler(ResidualNet(net, float(s["res_a"]), 0.063), EDM(args), args, batch_semantics="per_clip")


def test_full_width_blind_sampler_two_lanes_vs_reference_golden():
    from babe_amd._lib import dispatch_counts
    s = load("sampler_full_46046.npz")
    L, T = int(s["L"]), int(s["T"])
    net = full_net(L)
    smp = _full_sampler(net, s)
    assert smp.LANES == 2
    # per-step records first (B = 1, rid=True: single stream): denoised estimate and filter after the first evaluation of a step
    for b in range(2):
        itb = iter([s[f"noises{b}"][i:i + 1] for i in range(T + 1)])
        smp._randn = lambda shape, device: next(itb).to(device)
        xb, fpb, den, tt, filt = smp.predict_blind_bwe(s[f"y{b}"].cuda(), rid=True)
        assert torch.equal(tt, s["t"])
        for i in range(T):
            print(f"  clip {b} step {i}: x_den rel {rel(den[i], s[f'den{b}'][i]):.2e}, filter {filt[i].tolist()} vs {s[f'filt{b}'][i].tolist()}")
        print(f"  clip {b} B=1 single stream: RMS err {rms_err(xb, s[f'x{b}']):.2e}, rel {rel(xb, s[f'x{b}']):.2e}")
    y = torch.cat([s["y0"], s["y1"]], 0).cuda()
    noises = [torch.cat([s["noises0"][i:i + 1], s["noises1"][i:i + 1]], 0) for i in range(T + 1)]
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    dispatch_counts(reset=True)
    x, fp = smp.predict_blind_bwe(y)
    torch.cuda.synchronize()
    cnt = dispatch_counts(reset=True)
    assert smp._use_lanes(2, y, False, fp)                                  # the two-lane path is the one that ran
    for b in range(2):
        e_rms, e_rel = rms_err(x[b:b + 1], s[f"x{b}"]), rel(x[b:b + 1], s[f"x{b}"])
        print(f"full-width sampler clip {b}: RMS err {e_rms:.2e}, rel {e_rel:.2e}; filter {fp[b].tolist()} vs {s[f'fp{b}'].tolist()}")
        # clip 1 agrees to 2e-5.  Clip 0 carries a documented discontinuity of the REFERENCE algorithm: in its second score
        # evaluation the fitted fc[2] comes out at 409.1317 Hz in the reference and 409.1301 Hz here (difference 0.0016 Hz after
        # 100 GD iterations, the fit itself matches to 0.003 Hz on the reference's own inputs: tools/fit_trajectory_probe.py),
        # and the frequency of STFT bin 38, 409.1309 Hz, lies between the two - the mask `f >= fc` of design_filter
        # (utils/blind_bwe_utils.py:96-111) puts that bin in different filter segments, the guidance term of that evaluation
        # differs by 4 % (tools/eval_probe.py: every other evaluation of the run agrees to 4e-5) and the output by 2.9e-3
        # relative = 2.5e-4 RMS, inside the 1e-3 RMS bar.
        assert e_rms < 1e-3 and e_rel < (5e-3 if b == 0 else 1e-4)
        assert torch.allclose(fp[b, 0].cpu(), s[f"fp{b}"][0], rtol=1e-2) and torch.allclose(fp[b, 1].cpu(), s[f"fp{b}"][1], atol=1.0)
    # T = 3, order 2: 5 score evaluations per lane = 5 forwards + 5 VJPs per lane; frames 512..8, so the layers with >= 16
    # frames are on the F(4,3) kernel and only the 8-frame ones on the direct kernel
    print("full-width sampler dispatch:", {k: v for k, v in cnt.items() if v})
    assert cnt["conv53_wino4"] + cnt["conv53_wino45"] + cnt["conv53_wino85"] >= 2 * 5 * (59 + 52) and cnt["conv53_wino2"] == 0 and cnt["conv_bf16"] == 0, cnt
    assert cnt["conv53_direct"] <= 2 * 5 * (23 + 23), cnt


def rms_err(a, b):
    return float((a.detach().double().cpu() - b.double().cpu()).pow(2).mean().sqrt())


def test_full_width_bf16_sampler_B4_per_clip_vs_fp32_reference_runs(monkeypatch):
    """configs[2]'s arithmetic on the sampler: four clips (the two golden clips, twice) as one per-clip bf16 batch, each
    row against the imported reference's fp32 B = 1 run.  Stated bar for plain bf16: RMS error < 5e-3 (signal RMS 0.1).
    One stream is the default for a bf16 network (what else runs beside the bf16 conv is outside the library's control,
    networks/cqtdiff_plus.py); BABE_BF16_LANES=1 opts in to two clip lanes, which is what this test runs: the library's own
    kernels beside conv_bf16p must stay exact (no packed-fp32 instructions, tests/test_no_packed_fp32.py)."""
    s = load("sampler_full_46046.npz")
    L, T = int(s["L"]), int(s["T"])
    net = full_net(L, "bf16")
    monkeypatch.delenv("BABE_BF16_LANES", raising=False)
    assert not net.concurrent_lanes_ok and not net.lanes_ok_for("cpu")        # the default: one stream
    monkeypatch.setenv("BABE_BF16_LANES", "1")
    assert net.concurrent_lanes_ok and net.lanes_ok_for("cpu")                # the opt-in
    assert not net.lanes_ok_for("cuda")                                       # device-side noise = ATen kernels in the lane loop
    smp = _full_sampler(net, s)
    y = torch.cat([s["y0"], s["y1"], s["y0"], s["y1"]], 0).cuda()
    noises = [torch.cat([s["noises0"][i:i + 1], s["noises1"][i:i + 1]] * 2, 0) for i in range(T + 1)]
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(y)
    torch.cuda.synchronize()
    for b in range(4):
        ref = s[f"x{b % 2}"]
        e_rms, e_rel = rms_err(x[b:b + 1], ref), rel(x[b:b + 1], ref)
        print(f"full-width bf16 sampler clip {b}: RMS err {e_rms:.2e}, rel {e_rel:.2e}")
        assert e_rms < 5e-3
    # per-clip semantics: same clip, same result - bit for bit, on TWO clip lanes (until the library was built without
    # packed-fp32 instructions, babe_amd/build.py, two-lane bf16 runs differed about one time in four)
    assert net.concurrent_lanes_ok and smp._use_lanes(4, y, False, fp.reshape(4, 2, -1))
    assert torch.equal(x[0], x[2]) and torch.equal(x[1], x[3])


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the BENCHMARKED composition at its real size - full width, L = 368368, 44.1 kHz, network UNWRAPPED, the F(4,5) x F(4,3)
# kernels inside the sampler loop - against the imported reference (tests/golden/make_golden.py::g24)

def _noises_full(s, B):
    L, T = int(s["L"]), int(s["T"])
    gen = torch.Generator().manual_seed(int(s["seed"]))
    torch.randn(L, generator=gen)                                            # the draw that made the observation (synth_obs)
    return [torch.randn(1, L, generator=gen).repeat(B, 1) for _ in range(T + 1)]


def _full_size_sampler(net, s):
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    L, T = int(s["L"]), int(s["T"])
    args = default_args(sample_rate=44100, audio_len=L, T=T, start_sigma=float(s["start_sigma"]))
    args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
    return BlindSampler(net, EDM(args), args, batch_semantics="per_clip")


def test_full_size_blind_sampler_two_lanes_vs_reference_golden():
    """predict_blind_bwe (testing/blind_bwe_sampler.py:619-769) exactly as bench.py runs it - full width, 368368-sample segment,
    two stream lanes, default conv dispatch (the nested F(4,5) x F(4,3) kernels), fast fit kernel - against the imported
    reference's own run: T = 2 from sigma 0.2 = 3 score evaluations, network unwrapped, so UNet error reaches the output
    unattenuated.  The golden's conditioning is stored with it: a 1e-6 relative perturbation of y moves the REFERENCE's output
    by `probe_moved_x` (1.4e-4), which is what two correct fp32 implementations may differ by.  Bars: output RMS error < 1e-3
    (north star), per-step denoised estimate < 1e-3 relative, filter within params_close."""
    from babe_amd._lib import dispatch_counts
    s = load("sampler_full_368368.npz")
    L, T = int(s["L"]), int(s["T"])
    net = full_net(L)
    smp = _full_size_sampler(net, s)
    assert smp.LANES == 2
    print(f"golden: mu = {s['mu'].tolist()}, 1e-6 perturbation of y moves the reference's x by {float(s['probe_moved_x']):.2e}, the filter by {float(s['probe_moved_fp']):.2e}")
    # per-step records (B = 1, rid=True: single-stream loop)
    it = iter(_noises_full(s, 1))
    smp._randn = lambda shape, device: next(it).to(device)
    dispatch_counts(reset=True)
    x1, fp1, den, tt, filt = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    torch.cuda.synchronize()
    cnt1 = dispatch_counts(reset=True)
    assert torch.equal(tt, s["t"])
    for i in range(T):
        e = rel(den[i][:, ::16], s["data_denoised_sub16"][i])
        print(f"  step {i}: x_den rel {e:.2e}, filter {filt[i].tolist()} vs {s['data_filters'][i].tolist()}")
        assert e < 1e-3, (i, e)
        assert torch.allclose(filt[i][0], s["data_filters"][i][0], rtol=1e-2) and torch.allclose(filt[i][1], s["data_filters"][i][1], atol=1.0)
    print(f"  B = 1 single stream: RMS err {rms_err(x1, s['x']):.2e}, rel {rel(x1, s['x']):.2e}")
    assert rms_err(x1, s["x"]) < 1e-3 and rel(x1, s["x"]) < 5e-3
    assert cnt1["conv53_wino85"] >= 3 * 142, cnt1
    # both lanes: the clip twice as one per-clip batch on two streams - what bench.py times
    y2 = s["y"].repeat(2, 1).cuda()
    it = iter(_noises_full(s, 2))
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(y2)
    torch.cuda.synchronize()
    cnt = dispatch_counts(reset=True)
    assert smp._use_lanes(2, y2, False, fp)
    print("full-size sampler dispatch (two lanes):", {k: v for k, v in cnt.items() if v})
    assert cnt["conv53_wino85"] >= 2 * 3 * 142 and cnt["conv_bf16"] == 0 and cnt["conv53_direct"] == 0, cnt
    for b in range(2):
        e_rms, e_rel = rms_err(x[b:b + 1], s["x"]), rel(x[b:b + 1], s["x"])
        print(f"full-size sampler lane {b}: RMS err {e_rms:.2e}, rel {e_rel:.2e}; filter {fp[b].tolist()} vs {s['filter_params'].tolist()}")
        assert e_rms < 1e-3 and e_rel < 5e-3
        assert torch.allclose(fp[b, 0].cpu(), s["filter_params"][0], rtol=1e-2) and torch.allclose(fp[b, 1].cpu(), s["filter_params"][1], atol=1.0)
    assert torch.equal(x[0], x[1])                                           # same clip, same noise: bit-identical lanes


def test_full_size_blind_sampler_bf16_bar():
    """The same golden at a bf16 bar (configs[2]'s arithmetic; the T = 3 goldens wrap the network and attenuate UNet error ~1000x,
    this one does not): output RMS error < 5e-3 (signal RMS ~0.1), per-step x_den < 5e-2 relative."""
    s = load("sampler_full_368368.npz")
    L, T = int(s["L"]), int(s["T"])
    net = full_net(L, "bf16")
    smp = _full_size_sampler(net, s)
    it = iter(_noises_full(s, 1))
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp, den, tt, filt = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    for i in range(T):
        e = rel(den[i][:, ::16], s["data_denoised_sub16"][i])
        print(f"  bf16 step {i}: x_den rel {e:.2e}")
        assert e < 5e-2
    print(f"full-size bf16 sampler: RMS err {rms_err(x, s['x']):.2e}, rel {rel(x, s['x']):.2e}")
    assert rms_err(x, s["x"]) < 5e-3


def test_full_size_blind_sampler_on_the_library_evaluation(monkeypatch):
    """The same golden with every score evaluation as ONE library call (babe_score_eval on the UNet and CQT plans, BABE_EVAL_C=1:
    the non-Python host's path) at the benchmark's real size: bit-identical to the Python-sequenced run on the library's CQT plan
    (same kernels, same order), and therefore inside the same bars against the reference."""
    import babe_amd.testing.blind_bwe_sampler as bs
    s = load("sampler_full_368368.npz")
    L, T = int(s["L"]), int(s["T"])
    monkeypatch.setenv("BABE_CQT_C", "1")
    net = full_net(L)
    smp = _full_size_sampler(net, s)
    outs = {}
    for mode in (False, True):
        monkeypatch.setattr(bs, "EVAL_C", mode)
        it = iter(_noises_full(s, 2))
        smp._randn = lambda shape, device: next(it).to(device)
        x, fp = smp.predict_blind_bwe(s["y"].repeat(2, 1).cuda())
        torch.cuda.synchronize()
        outs[mode] = (x.clone(), fp.clone())
    assert smp._ceval, "the library path did not run"
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
    e_rms, e_rel = rms_err(outs[True][0][0:1], s["x"]), rel(outs[True][0][0:1], s["x"])
    print(f"full-size sampler, one C call per evaluation: RMS err {e_rms:.2e}, rel {e_rel:.2e}")
    assert e_rms < 1e-3 and e_rel < 5e-3
