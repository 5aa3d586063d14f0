"""Denoiser pre-pass on the HIP kernels (csrc/denoiser.hip, babe_amd/networks/denoiser.py, babe_amd/testing/denoise.py)
against torch CPU references of each op, the oracle, and golden outputs of the reference (tests/golden/denoiser.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import denoiser as OD

pytestmark = pytest.mark.gpu

CFGS = {
    "full": dict(depth=6, num_tfc=3, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513, T=48),
    "s1": dict(depth=3, num_tfc=2, num_stages=1, use_SAM=False, use_fencoding=False, f_dim=129, T=40),
    "nosam": dict(depth=2, num_tfc=1, num_stages=2, use_SAM=False, use_fencoding=True, f_dim=65, T=21),
}


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "denoiser.npz"))


@pytest.fixture(scope="module")
def dn():
    from babe_amd.networks import denoiser
    return denoiser


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride", [
    (1, 12, 64, 20, 70, 7, 1),        # first feature extractor
    (2, 24, 64, 9, 33, 3, 1),
    (1, 64, 2, 17, 129, 3, 1),        # final block: 2 output channels
    (1, 70, 130, 5, 9, 3, 1),         # ragged channel counts, tiny planes
    (2, 16, 32, 11, 40, 1, 1),
    (1, 64, 64, 21, 65, 4, 2),        # down conv: reflect pad 2, stride 2
    (1, 8, 8, 3, 3, 4, 2),
])
def test_conv_vs_torch(dn, B, Cin, Cout, H, W, k, stride):
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    if stride == 1:
        p = (k - 1) // 2
        xp = F.pad(x.double(), (p, p, p, p), mode="reflect") if p else x.double()
        ref = F.conv2d(xp, w.double(), b.double())
        pad = None
    else:
        ref = F.conv2d(F.pad(x.double(), (2, 2, 2, 2), mode="reflect"), w.double(), b.double(), stride=2)
        pad = (2, 2)
    pc = dn._Packed(w.cuda(), b.cuda())
    out = torch.empty(ref.shape, device="cuda")
    dn._conv(x.cuda(), pc, out, stride=stride, pad=pad)
    assert rel(out, ref) < 2e-6
    # ELU + residual, written into a channel slice of a larger buffer, input read from a channel slice
    res = torch.randn(ref.shape, generator=g)
    big_in = torch.randn(B, Cin + 5, H, W, generator=g).cuda()
    big_in[:, 5:] = x.cuda()
    big_out = torch.zeros(B, Cout + 3, ref.shape[2], ref.shape[3], device="cuda")
    o = big_out[:, 3:]
    o.copy_(res.cuda())
    dn._conv(big_in[:, 5:], pc, o, stride=stride, pad=pad, act=True, res=o)
    assert rel(o, F.elu(ref) + res.double()) < 2e-6
    assert float(big_out[:, :3].abs().max()) == 0.0


@pytest.mark.parametrize("B,Cin,Cout,h,w,H,W", [(1, 16, 8, 5, 6, 9, 11), (2, 64, 64, 8, 9, 15, 17), (1, 70, 33, 3, 3, 6, 6)])
def test_transposed_conv_upsample_merge_vs_torch(dn, B, Cin, Cout, h, w, H, W):
    """D_Block front end (denoiser.py:397-407): ELU(tconv) cropped + nearest-upsampled projection, cropped to the bridge."""
    from babe_amd._lib import check, lib, stream
    g = torch.Generator().manual_seed(Cin + h)
    x = torch.randn(B, Cin, h, w, generator=g)
    wt = torch.randn(Cin, Cout, 4, 4, generator=g) / (Cin * 4) ** 0.5
    bt = torch.randn(Cout, generator=g)
    wp = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    bp = torch.randn(Cout, generator=g)
    up = F.elu(F.conv_transpose2d(x.double(), wt.double(), bt.double(), stride=2))
    x2 = F.conv2d(x.double().repeat_interleave(2, 2).repeat_interleave(2, 3), wp.double(), bp.double())
    y = OD._crop_to(up, x2.shape) + x2
    ref = OD._crop_to(y, (B, Cout, H, W))
    out = torch.empty(B, Cout, H, W, device="cuda")
    low = torch.empty(B, Cout, h, w, device="cuda")
    d2h, d2w = (2 * h - H) // 2, (2 * w - W) // 2
    dn._tconv(x.cuda(), dn._Packed(wt.cuda(), bt.cuda(), tconv=True), out, 1 + d2h, 1 + d2w)
    dn._conv(x.cuda(), dn._Packed(wp.cuda(), bp.cuda()), low)
    p, bs, cs = dn._planes(out)
    lp, lbs, lcs = dn._planes(low)
    check(lib().babe_dn_upsample_add(p, bs, cs, lp, lbs, lcs, B, Cout, H, W, h, w, d2h, d2w, stream()), "upsample_add")
    assert rel(out, ref) < 2e-6


@pytest.mark.parametrize("name", ["s1", "nosam", "full"])
def test_network_vs_reference_golden_and_oracle(dn, gold, name):
    c = CFGS[name]
    cfg = {k: v for k, v in c.items() if k != "T"}
    net = dn.MultiStage_denoise(cfg)
    sd = dn.init_state_dict(cfg, seed=7)
    osd = OD.init_state_dict(c, seed=7)
    assert list(sd.keys()) == list(osd.keys()) and all(torch.equal(sd[k], osd[k]) for k in sd)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net.to("cuda")
    g = torch.Generator().manual_seed(int(gold[f"{name}_seed"]))
    X = torch.randn(1 if name == "full" else 2, 2, c["T"], c["f_dim"], generator=g)
    y = net(X.cuda())
    if c["num_stages"] > 1:
        assert rel(y[0], gold[f"{name}_pred2"]) < 2e-5
        assert rel(y[1], gold[f"{name}_pred1"]) < 2e-5
    else:
        assert rel(y, gold[f"{name}_pred1"]) < 2e-5
    # a second call on the cached buffers gives the same result (no state leaks between calls)
    y2 = net(X.cuda())
    a, b_ = (y[0], y2[0]) if c["num_stages"] > 1 else (y, y2)
    assert torch.equal(a, b_)


def test_stft_istft_vs_torch(dn):
    from babe_amd.testing.denoise import DenoiserPrepass
    pp = DenoiserPrepass(None, dict(stft_win_size=1024, stft_hop_size=256, num_stages=1), "cuda")
    from babe_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1024 + 256 * 37 + 100, generator=g)
    win = torch.hamming_window(1024)
    ref = torch.view_as_real(torch.stft(x.double(), 1024, hop_length=256, window=win.double(), center=False,
                                        return_complex=True)).permute(0, 3, 2, 1)
    frames = ref.shape[2]
    X = torch.empty(2, 2, frames, 513, device="cuda")
    xc = x.cuda()
    check(lib().babe_dn_stft(ptr(xc), xc.stride(0), x.shape[1], ptr(X), 2, 1024, 256, frames, ptr(pp.tw4096), stream()), "stft")
    assert rel(X, ref) < 2e-6
    P = torch.randn(2, 2, frames, 513, generator=g)
    yref = torch.istft(torch.view_as_complex(P.double().permute(0, 3, 2, 1).contiguous()), 1024, hop_length=256,
                       window=win.double(), center=False)
    ws = torch.empty(2, frames, 1024, device="cuda")
    y = torch.empty(2, yref.shape[1], device="cuda")
    Pc = P.cuda()
    check(lib().babe_dn_istft(ptr(Pc), ptr(ws), ptr(y), y.stride(0), yref.shape[1], 2, 1024, 256, frames, ptr(pp.tw4096),
                              stream()), "istft")
    assert rel(y, yref) < 2e-6


def test_segmented_application_vs_reference_golden(dn, gold):
    from babe_amd.testing.denoise import DenoiserPrepass
    cfg = dict(depth=3, num_tfc=1, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
    net = dn.MultiStage_denoise(cfg)
    net.load_state_dict(dn.init_state_dict(cfg, seed=11))
    net.to("cuda")
    pp = DenoiserPrepass(net, dict(sample_rate_denoiser=4000, segment_size=2, stft_win_size=1024, stft_hop_size=256,
                                   num_stages=2), "cuda")
    g = torch.Generator().manual_seed(int(gold["seg_seed"]))
    x = 0.1 * torch.randn(2, 20000, generator=g)
    y1 = pp.apply_denoiser_model(x[:, :8000].cuda())
    assert y1.shape == gold["seg_model"].shape and rel(y1, gold["seg_model"]) < 2e-5
    y = pp.apply_denoiser(x.cuda())
    assert y.shape == gold["seg_full"].shape and rel(y, gold["seg_full"]) < 2e-5
