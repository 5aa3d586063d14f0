"""The library-side UNet sequencer (csrc/unet_engine.hip: babe_unet_fwd / babe_unet_vjp, one C call per direction from a plan handle)
against the Python-sequenced engine (babe_amd/networks/unet_engine.py): same kernels in the same order, so the results must be
IDENTICAL bit for bit - forward and input-VJP, reduced and benchmark width, one state and two states (clip lanes) over one plan."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _small():
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "unet_small.npz")).items()}
    sd = {k[3:]: v.cuda().float() for k, v in g.items() if k.startswith("sd.")}
    return sd, [8, 8, 8, 8, 16, 16, 16], [2, 3, 4, 5, 6, 7, 7]


def _run(eng, C_list, film, gouts):
    outs = eng.forward(C_list, film)
    gC = eng.vjp(gouts)
    torch.cuda.synchronize()
    return outs, gC


@pytest.mark.parametrize("B,T0", [(2, 16), (1, 64)])
def test_c_engine_equals_python_engine_small(monkeypatch, B, T0):
    from babe_amd.networks import unet_engine as ue
    sd, Ns, nd = _small()
    gen = torch.Generator().manual_seed(5 + B)
    C_list = [torch.randn(B, 2, 64, T0 * 2 ** j, generator=gen).cuda() for j in range(7)]
    gouts = [torch.randn(c.shape, generator=gen).cuda() for c in C_list]
    cn = torch.linspace(-1.2, -0.3, B).reshape(B, 1).cuda()
    monkeypatch.setattr(ue, "USE_C", False)
    eng = ue.UnetEngine(sd, Ns, nd)
    film = eng.embed(cn)
    o_py, g_py = _run(eng, C_list, film, gouts)
    monkeypatch.setattr(ue, "USE_C", True)
    eng_c = ue.UnetEngine(sd, Ns, nd)
    assert eng_c._c_engine() is not None
    o_c, g_c = _run(eng_c, C_list, film, gouts)
    for a, b in zip(o_py + g_py, o_c + g_c):
        assert torch.equal(a, b)
    # a second evaluation through the same state and workspace (other inputs), then two states over one plan on two streams
    C2 = [c * 0.5 + 0.1 for c in C_list]
    monkeypatch.setattr(ue, "USE_C", False)
    o_py2, g_py2 = _run(eng, C2, film, gouts)
    monkeypatch.setattr(ue, "USE_C", True)
    o_c2, g_c2 = _run(eng_c, C2, film, gouts)
    for a, b in zip(o_py2 + g_py2, o_c2 + g_c2):
        assert torch.equal(a, b)
    lane = eng_c.clone_state()
    assert lane._c_engine().plan == eng_c._c_engine().plan and lane._c_engine().state != eng_c._c_engine().state
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        oa = eng_c.forward(C_list, film)
    with torch.cuda.stream(s2):
        ob = lane.forward(C2, film)
    with torch.cuda.stream(s1):
        ga = eng_c.vjp(gouts)
    with torch.cuda.stream(s2):
        gb = lane.vjp(gouts)
    torch.cuda.synchronize()
    for a, b in zip(o_py + g_py + o_py2 + g_py2, oa + ga + ob + gb):
        assert torch.equal(a, b)


def test_c_engine_equals_python_engine_full_width(monkeypatch):
    """Benchmark width (Ns = [64, 96, 96, 128, 128, 256, 256]) on a 46046-sample segment's octave lengths: every nested-Winograd,
    (1,1), few-channel and pyramid dispatch goes through babe_conv2d_auto."""
    from babe_amd.networks import unet_engine as ue
    from babe_amd.networks.cqtdiff_plus import init_state_dict
    Ns, nd = [64, 96, 96, 128, 128, 256, 256], [2, 3, 4, 5, 6, 7, 7]
    sd = {k: v.cuda() for k, v in init_state_dict(Ns, nd, seed=3, gate_scale=1.0).items()}
    gen = torch.Generator().manual_seed(11)
    Ts = [16 * 2 ** j for j in range(7)]
    C_list = [torch.randn(1, 2, 64, T, generator=gen).cuda() for T in Ts]
    gouts = [torch.randn(c.shape, generator=gen).cuda() for c in C_list]
    cn = torch.tensor([[-0.7]]).cuda()
    monkeypatch.setattr(ue, "USE_C", False)
    eng = ue.UnetEngine(sd, Ns, nd)
    film = eng.embed(cn)
    o_py, g_py = _run(eng, C_list, film, gouts)
    monkeypatch.setattr(ue, "USE_C", True)
    eng_c = ue.UnetEngine(sd, Ns, nd)
    o_c, g_c = _run(eng_c, C_list, film, gouts)
    for a, b in zip(o_py + g_py, o_c + g_c):
        assert torch.equal(a, b)


def test_groupnorm_sums_from_the_conv_epilogue_leave_the_network_unchanged(monkeypatch):
    """Round 6: with BABE_FUSE_GN_FWD (the default) the forward (5,3) convs on the F(4,5) kernel hand the next layer's GroupNorm its
    sums (babe_conv_args::stat_mode 1) instead of a pass of babe_gn_partial over the tensor: fewer statistics launches, and the
    network's outputs and input-VJP equal to the own-pass form to float rounding of the statistics (the sums are the same numbers
    added in another order, in double)."""
    from babe_amd import ops
    from babe_amd._lib import dispatch_counts
    from babe_amd.networks import unet_engine as ue
    from babe_amd.networks.cqtdiff_plus import init_state_dict
    Ns, nd = [64, 96, 96, 128, 128, 256, 256], [2, 3, 4, 5, 6, 7, 7]
    sd = {k: v.cuda() for k, v in init_state_dict(Ns, nd, seed=3, gate_scale=1.0).items()}
    gen = torch.Generator().manual_seed(12)
    Ts = [16 * 2 ** j for j in range(7)]
    C_list = [torch.randn(1, 2, 64, T, generator=gen).cuda() for T in Ts]
    gouts = [torch.randn(c.shape, generator=gen).cuda() for c in C_list]
    cn = torch.tensor([[-0.7]]).cuda()
    monkeypatch.setattr(ue, "USE_C", False)
    res, launches = [], []
    for fuse in (False, True):
        monkeypatch.setattr(ops, "FUSE_GN_FWD", fuse)
        eng = ue.UnetEngine(sd, Ns, nd)
        film = eng.embed(cn)
        dispatch_counts(reset=True)
        o, g = _run(eng, C_list, film, gouts)
        launches.append(dispatch_counts()["gn_stats"])
        res.append([t.clone() for t in o + g])
    assert launches[1] < launches[0], launches
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), float((a - b).abs().max() / a.abs().max())


def test_workspace_too_small_fails_loudly():
    from babe_amd._lib import lib
    from babe_amd.networks import unet_engine as ue
    from babe_amd.networks.unet_c import CUnet
    import ctypes as C
    sd, Ns, nd = _small()
    eng = ue.UnetEngine(sd, Ns, nd)
    cu = CUnet(eng)
    C_list = [torch.randn(1, 2, 64, 16 * 2 ** j).cuda() for j in range(7)]
    film = eng.embed(torch.tensor([[-0.5]]).cuda())
    outs = [torch.empty_like(c) for c in C_list]
    ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    P = C.c_void_p
    rc = lib().babe_unet_fwd(cu.plan, cu.state, (P * 7)(*[c.data_ptr() for c in C_list]), film.data_ptr(), film.stride(0), 1,
                             (C.c_int * 7)(*[c.shape[-1] for c in C_list]), ws.data_ptr(), ws.numel(),
                             (P * 7)(*[o.data_ptr() for o in outs]), None)
    torch.cuda.synchronize()
    assert rc != 0 and b"workspace too small" in lib().babe_last_error()
