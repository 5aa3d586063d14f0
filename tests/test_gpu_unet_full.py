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


def full_net(L):
    """One full-width HIP network per length for the whole module (packing 1.8 GB of Winograd weights takes seconds)."""
    if L not in _NETS:
        from babe_amd.config import default_args
        from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
        from tests.golden_weights import full_width_sd
        _NETS.clear()                                   # keep one geometry resident at a time
        args = default_args(sample_rate=44100, audio_len=L)
        net = Unet_CQT_oct_with_attention(args, "cuda")
        net.load_state_dict(full_width_sd(0), strict=True)
        _NETS[L] = net
    return _NETS[L]


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
    assert cf["conv53_wino4"] + cf["conv53_direct"] == 82 and cf["conv53_direct"] <= 23 and cf["conv53_wino2"] == 0, cf
    assert cf["conv_bf16"] == 0 and cb["conv_bf16"] == 0


def test_full_width_L368368_vs_reference_golden_all_wino4():
    """The benchmark's segment geometry (conf/exp/maestro44k_8s.yaml:51-52): frames 4096 ... 64."""
    g = load("unet_full_368368.npz")
    L = 368368
    y, gx, cf, cb = run_fwd_vjp(full_net(L), g, L)
    ey, eg = rel(y, g["y"]), rel(gx, g["gx"])
    print(f"full width L={L}: fwd rel {ey:.2e}, vjp rel {eg:.2e}; dispatch fwd {cf} vjp {cb}")
    assert ey < TOL_FWD and eg < TOL_VJP
    # forward: 75 dilated ResnetBlock convs + 7 pyramid projections, all on conv_wino4_kernel
    assert cf["conv53_wino4"] == 82 and cf["conv53_wino2"] == 0 and cf["conv53_direct"] == 0, cf
    # VJP: the 75 dilated convs on F(4,3); the 7 pyramid projections transposed have 2 output channels and run on the
    # vector-ALU kernel (csrc/conv_fewco.hip)
    assert cb["conv53_wino4"] == 75 and cb["conv53_fewco"] == 7 and cb["conv53_wino2"] == 0 and cb["conv53_direct"] == 0, cb
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
    assert cf["conv53_wino4"] == 82 and cb["conv53_wino4"] == 75
