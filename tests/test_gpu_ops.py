"""HIP ops (through the C-ABI) vs the CPU oracle on seeded inputs.  Needs a MI355X."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import unet as UN

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from babe_amd import ops
    return ops


@pytest.mark.parametrize("B,Cin,Cout,Fq,T,kh,kw,dil", [
    (1, 8, 16, 64, 32, 5, 3, 1),
    (2, 16, 16, 128, 16, 5, 3, 4),
    (1, 64, 64, 64, 256, 5, 3, 2),
    (1, 96, 96, 128, 64, 5, 3, 64),
    (2, 2, 8, 64, 24, 1, 1, 1),
    (1, 16, 2, 64, 40, 1, 1, 1),
    (1, 2, 64, 192, 128, 5, 3, 1),
    (1, 128, 256, 70, 100, 5, 3, 8),
    (1, 256, 128, 64, 64, 1, 1, 1),
])
def test_conv2d_fwd_and_vjp(ops, B, Cin, Cout, Fq, T, kh, kw, dil):
    # 2e-6: direct and Winograd F(2,3) kernels; shapes that qualify for F(4,3) (64/96/128k channels, T % 4 == 0) carry its
    # larger transform constants: 1e-6 typical, 3e-6 bound (csrc/conv_wino4.hip header)
    tol = 3e-6 if (kw == 3 and T % 4 == 0) else 2e-6
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g) / math.sqrt(Cin * kh * kw)
    ref = UN.conv_same(x.double(), w.double(), dil)
    pc = ops.PackedConv(w.cuda())
    out = torch.empty(B, Cout, Fq, T, device="cuda")
    ops.conv2d(x.cuda(), pc, out, dil=dil)
    assert rel(out, ref) < tol
    # epilogue: alpha*acc*oscale + rbeta*res
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    out2 = res.cuda().clone()
    ops.conv2d(x.cuda(), pc, out2, dil=dil, res=out2, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
    ref2 = 0.7 * ref * osc[:, :, None, None].double() + 0.3 * res.double()
    assert rel(out2, ref2) < tol
    # input-VJP with per-channel input scale
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    xr = x.double().requires_grad_(True)
    y = UN.conv_same(xr, w.double(), dil)
    gref, = torch.autograd.grad((y * (gy * isc[:, :, None, None]).double()).sum(), xr)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda())
    assert rel(gx, gref) < tol


@pytest.mark.parametrize("precision,tol", [("bf16", 1.5e-2), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("B,Cin,Cout,Fq,T,kh,kw,dil", [
    (1, 8, 16, 64, 32, 5, 3, 1),
    (2, 16, 16, 128, 16, 5, 3, 4),
    (1, 96, 96, 128, 64, 5, 3, 64),
    (2, 2, 8, 64, 24, 1, 1, 1),
    (1, 16, 2, 64, 40, 1, 1, 1),
    (1, 2, 64, 192, 128, 5, 3, 1),
    (1, 128, 256, 70, 100, 5, 3, 8),
    (1, 256, 128, 64, 64, 1, 1, 1),
    (2, 64, 64, 64, 1024, 5, 3, 2),
])
def test_conv2d_bf16_variants(ops, precision, tol, B, Cin, Cout, Fq, T, kh, kw, dil):
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g) / math.sqrt(Cin * kh * kw)
    ref = UN.conv_same(x.double(), w.double(), dil)
    pc = ops.PackedConv(w.cuda(), precision)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    out = res.cuda().clone()
    ops.conv2d(x.cuda(), pc, out, dil=dil, res=out, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
    ref2 = 0.7 * ref * osc[:, :, None, None].double() + 0.3 * res.double()
    assert rel(out, ref2) < tol
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    xr = x.double().requires_grad_(True)
    y = UN.conv_same(xr, w.double(), dil)
    gref, = torch.autograd.grad((y * (gy * isc[:, :, None, None]).double()).sum(), xr)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda())
    assert rel(gx, gref) < tol


def test_conv2d_two_sources_and_views(ops):
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(2, 16, 64, 32, generator=g)
    x2 = torch.randn(2, 16, 64, 32, generator=g)
    w = torch.randn(24, 32, 1, 1, generator=g) / 6
    ref = UN.conv_same(torch.cat((x1, x2), 1).double(), w.double())
    big = torch.zeros(2, 24, 128, 32, device="cuda")
    pc = ops.PackedConv(w.cuda())
    ops.conv2d(x1.cuda(), pc, big[:, :, 64:, :], x2=x2.cuda())
    assert rel(big[:, :, 64:, :], ref) < 2e-6
    assert float(big[:, :, :64, :].abs().max()) == 0.0
    # input as a frequency sub-view
    xin = torch.randn(2, 32, 128, 32, generator=g)
    out = torch.empty(2, 24, 64, 32, device="cuda")
    ops.conv2d(xin.cuda()[:, :, 64:, :], pc, out)
    assert rel(out, UN.conv_same(xin[:, :, 64:, :].double(), w.double())) < 2e-6


@pytest.mark.parametrize("B,C1,C2,Cout,Fq,T,dil", [
    (2, 32, 32, 64, 48, 64, 2),       # two sources, 64co x 256pos variant
    (1, 96, 0, 96, 40, 36, 4),        # 96 channels, T % 8 != 0
    (1, 128, 128, 128, 33, 128, 16),  # 128co x 128pos variant, ragged F
    (2, 5, 0, 20, 17, 20, 1),         # channel padding on both sides
    (1, 72, 0, 256, 9, 272, 2),       # two 128-channel blocks, T not a power of two, Cin % 8 == 0 only
    (1, 20, 44, 64, 12, 16, 8),       # source split inside an 8-channel slab, shortest rows (T = 16), taps skipped
])
def test_conv2d_winograd_vs_direct_and_oracle(ops, B, C1, C2, Cout, Fq, T, dil):
    """Winograd F(2,3)-along-time kernel (csrc/conv_wino.hip) against the direct kernel and the float64 oracle."""
    import ctypes as C
    from babe_amd._lib import lib
    g = torch.Generator().manual_seed(C1 + Cout + T)
    Cin = C1 + C2
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    ref = 0.7 * UN.conv_same(x.double(), w.double(), dil) * osc[:, :, None, None].double() + 0.3 * res.double()
    pc4 = ops.PackedConv(w.cuda())                          # F(4,3) where the shape qualifies, else F(2,3)
    pcw = ops.PackedConv(w.cuda())                          # F(2,3)
    assert pcw.fwd_wino is not None and pc4.fwd_wino4 is not None
    pcw.fwd_wino4 = pcw.bwd_wino4 = None
    pcd = ops.PackedConv(w.cuda())                          # direct
    pcd.fwd_wino = pcd.bwd_wino = pcd.fwd_wino4 = pcd.bwd_wino4 = None
    for pc in (pc4, pcw, pcd):                              # (the nested-Winograd kernels have their own tests below)
        pc.fwd_wino45 = pc.bwd_wino45 = pc.fwd_wino85 = pc.bwd_wino85 = None
    xc = x.cuda()
    x1, x2 = (xc[:, :C1].contiguous(), xc[:, C1:].contiguous()) if C2 else (xc, None)
    outs = []
    for pc in (pcw, pcd, pc4):
        big = torch.zeros(B, Cout, 2 * Fq, T, device="cuda")
        o = big[:, :, Fq:, :]
        o.copy_(res.cuda())
        ops.conv2d(x1, pc, o, dil=dil, x2=x2, res=o, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
        assert float(big[:, :, :Fq, :].abs().max()) == 0.0
        outs.append(o.clone())
    assert rel(outs[0], ref) < 2e-6 and rel(outs[1], ref) < 2e-6 and rel(outs[2], ref) < 3e-6
    assert rel(outs[0], outs[1]) < 2e-6
    # input-VJP weights
    gy = torch.randn(B, Cout, Fq, T, generator=g).cuda()
    gx = [torch.empty(B, Cin, Fq, T, device="cuda") for _ in range(3)]
    ops.conv2d(gy, pcw, gx[0], dil=dil, transpose=True)
    ops.conv2d(gy, pcd, gx[1], dil=dil, transpose=True)
    ops.conv2d(gy, pc4, gx[2], dil=dil, transpose=True)
    xr = x.double().requires_grad_(True)
    gref, = torch.autograd.grad((UN.conv_same(xr, w.double(), dil) * gy.cpu().double()).sum(), xr)
    assert rel(gx[0], gref) < 2e-6 and rel(gx[0], gx[1]) < 2e-6 and rel(gx[2], gref) < 3e-6


@pytest.mark.parametrize("B,C1,C2,Cout,Fq,T", [
    (2, 64, 64, 96, 40, 128),      # two sources (decoder proj_in), 96 output channels (3 row tiles)
    (1, 2, 0, 64, 64, 256),        # init block proj_in: 2 input channels (one zero-padded K-slab)
    (2, 128, 0, 2, 33, 128),       # out block proj_out: 2 output channels, ragged F
    (1, 72, 0, 256, 20, 272),      # Cin % 16 != 0 (last slab half empty), T not a power of two, two 128-channel tiles
    (1, 256, 256, 128, 28, 64),    # decoder level 6: 512 -> 128, short rows (4 rows per tile)
])
def test_conv2d_1x1_pipelined_vs_oracle(ops, B, C1, C2, Cout, Fq, T):
    """(1,1) convs on the pipelined DMA kernel (csrc/conv11p.hip): forward with the fused epilogue (oscale, residual into a
    frequency sub-view) and the input-VJP with in_scale, vs float64."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(C1 + Cout + T)
    Cin = C1 + C2
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    ref = 0.7 * UN.conv_same(x.double(), w.double()) * osc[:, :, None, None].double() + 0.3 * res.double()
    pc = ops.PackedConv(w.cuda())
    xc = x.cuda()
    x1, x2 = (xc[:, :C1].contiguous(), xc[:, C1:].contiguous()) if C2 else (xc, None)
    big = torch.zeros(B, Cout, 2 * Fq, T, device="cuda")
    o = big[:, :, Fq:, :]
    o.copy_(res.cuda())
    ops.conv2d(x1, pc, o, x2=x2, res=o, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
    assert float(big[:, :, :Fq, :].abs().max()) == 0.0
    assert rel(o, ref) < 2e-6
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    ops.conv2d(gy.cuda(), pc, gx, transpose=True, in_scale=isc.cuda(), alpha=0.5)
    xr = x.double().requires_grad_(True)
    gref, = torch.autograd.grad((UN.conv_same(xr, w.double()) * (gy.double() * isc[:, :, None, None].double())).sum(), xr)
    assert rel(gx, 0.5 * gref) < 2e-6


@pytest.mark.parametrize("B,N,Fq,T,dil", [(2, 64, 40, 64, 1), (1, 96, 33, 128, 4), (1, 256, 16, 16, 2)])
def test_conv2d_few_output_channels_vjp_of_pyramid_projection(ops, B, N, Fq, T, dil):
    """Input-VJP of Conv2d(2 -> N, (5,3)) (the pyramid projections): 2 output channels on the vector-ALU kernel
    (csrc/conv_fewco.hip), with alpha and an accumulated residual, vs float64 autograd."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(N + T)
    w = torch.randn(N, 2, 5, 3, generator=g) / math.sqrt(30)
    pc = ops.PackedConv(w.cuda())
    gy = torch.randn(B, N, Fq, T, generator=g)
    acc = torch.randn(B, 2, Fq, T, generator=g)
    x = torch.zeros(B, 2, Fq, T, dtype=torch.float64, requires_grad=True)
    gref, = torch.autograd.grad((UN.conv_same(x, w.double(), dil) * gy.double()).sum(), x)
    out = acc.cuda().clone()
    dispatch_counts(reset=True)
    ops.conv2d(gy.cuda(), pc, out, dil=dil, transpose=True, alpha=0.7, res=out, rbeta=1.0)
    assert dispatch_counts()["conv53_fewco"] == 1
    assert rel(out, 0.7 * gref + acc.double()) < 2e-6


def test_conv2d_winograd_dispatch_rules(ops):
    """Problems the Winograd kernel does not take (T % 4, misaligned views, 1x1) run on the direct kernel."""
    import ctypes as C
    from babe_amd._lib import ConvArgs, lib
    w = torch.randn(32, 32, 5, 3, device="cuda") / 20
    pc = ops.PackedConv(w)
    for T in (30, 18):
        x = torch.randn(1, 32, 16, T, device="cuda")
        out = torch.empty(1, 32, 16, T, device="cuda")
        ops.conv2d(x, pc, out, dil=2)
        assert rel(out, UN.conv_same(x.cpu().double(), w.cpu().double(), 2)) < 2e-6
    a = ConvArgs()
    a.KW, a.KH, a.T, a.Cout = 3, 5, 30, 32
    assert lib().babe_conv2d_wino_supported(C.byref(a)) == 0
    a.T, a.KW = 32, 1
    assert lib().babe_conv2d_wino_supported(C.byref(a)) == 0
    a.KW = 3
    assert lib().babe_conv2d_wino_supported(C.byref(a)) == 1
    a.Cout = 128
    assert lib().babe_conv2d_wino4_supported(C.byref(a)) == 1
    a.Cout = 32     # F(4,3) tiles 64, 96 and multiples of 128 channels only
    assert lib().babe_conv2d_wino4_supported(C.byref(a)) == 0 and lib().babe_conv2d_wino_supported(C.byref(a)) == 1
    a.Cout = 128
    a.out = 8       # F(4,3) stores 16 bytes per lane
    assert lib().babe_conv2d_wino4_supported(C.byref(a)) == 0 and lib().babe_conv2d_wino_supported(C.byref(a)) == 1
    a.out = 0
    a.in_ = 4       # 4-byte aligned but not 16
    assert lib().babe_conv2d_wino_supported(C.byref(a)) == 0 and lib().babe_conv2d_wino4_supported(C.byref(a)) == 0


@pytest.mark.parametrize("B,C,Fq,T", [(2, 16, 64, 24), (1, 64, 128, 256), (1, 8, 5, 8)])
def test_groupnorm_film_gelu_fwd_bwd(ops, B, C, Fq, T):
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, C, Fq, T, generator=g) * 1.7 + 0.4
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    film = 0.3 * torch.randn(B, C, generator=g)
    xr = x.double().requires_grad_(True)
    a_ref = F.gelu(UN.group_norm_nomean(xr, gamma.double().view(1, C, 1, 1)) * (film.double()[:, :, None, None] + 1))
    xc = x.cuda()
    stats, scale = ops.gn_scale(xc, gamma.cuda(), film.cuda())
    a = ops.scale_gelu(xc, scale, torch.empty_like(xc))
    assert rel(a, a_ref) < 2e-6
    da = torch.randn(B, C, Fq, T, generator=g)
    gy = torch.randn(B, C, Fq, T, generator=g)
    gref, = torch.autograd.grad((a_ref * da.double()).sum(), xr)
    gref = gref + 0.5 * gy.double()
    dac = da.cuda()
    gx = gy.cuda().clone()
    ops.gn_bwd(xc, dac, gx, scale, stats, gx, 0.5)
    assert rel(gx, gref) < 5e-6

def test_gn_stats_one_launch_equals_two_launches(ops):
    """gn_partial_kernel<true> (the last workgroup of a group finalises it: one launch) against babe_gn_partial +
    babe_gn_finalize: bit-identical statistics and scales, over shapes with different split counts on the SAME ticket buffer
    (every call must leave its tickets at zero), and from two streams at once."""
    g = torch.Generator().manual_seed(77)
    shapes = [(2, 64, 16, 256), (1, 96, 24, 64), (2, 128, 40, 128), (1, 256, 56, 64), (2, 64, 16, 256)]
    for rep in range(2):
        for (B, C, F, T) in shapes:
            x = torch.randn(B, C, F, T, generator=g).cuda()
            gamma = (torch.rand(C, generator=g) + 0.5).cuda()
            film = torch.randn(B, C, generator=g).cuda()
            ops.GN_FUSED = True
            st1, sc1 = ops.gn_scale(x, gamma, film)
            ops.GN_FUSED = False
            st0, sc0 = ops.gn_scale(x, gamma, film)
            ops.GN_FUSED = True
            assert torch.equal(st1, st0) and torch.equal(sc1, sc0), (B, C, F, T)
    assert all(int(t.abs().sum()) == 0 for t in ops._GN_TICKETS.values())
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.randn(2, 128, 64, 256, generator=g).cuda()
    gamma, film = torch.ones(128, device="cuda"), torch.zeros(2, 128, device="cuda")
    ref = ops.gn_scale(x, gamma, film)
    torch.cuda.synchronize()
    outs = []
    for _ in range(8):
        with torch.cuda.stream(sA):
            outs.append(ops.gn_scale(x, gamma, film))
        with torch.cuda.stream(sB):
            outs.append(ops.gn_scale(x, gamma, film))
    torch.cuda.synchronize()
    assert all(torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) for o in outs)

def test_scale_gelu_with_folded_finalize_is_bit_identical(ops):
    """babe_scale_gelu_fin (gn_finalize's work in the GELU kernel's prologue) against gn_scale + scale_gelu."""
    g = torch.Generator().manual_seed(78)
    for (B, C, F, T) in [(2, 64, 16, 256), (1, 96, 24, 64), (2, 128, 40, 128), (1, 256, 56, 64)]:
        x = torch.randn(B, C, F, T, generator=g).cuda()
        gamma = (torch.rand(C, generator=g) + 0.5).cuda()
        film = torch.randn(B, C, generator=g).cuda()
        st0, sc0 = ops.gn_scale(x, gamma, film)
        a0 = ops.scale_gelu(x, sc0, torch.empty_like(x))
        a1 = torch.empty_like(x)
        keep, ops.GELU_FIN = ops.GELU_FIN, True
        st1, sc1 = ops.gn_scale_gelu(x, gamma, film, a1)
        ops.GELU_FIN = keep
        assert torch.equal(st1, st0) and torch.equal(sc1, sc0) and torch.equal(a1, a0), (B, C, F, T)


@pytest.mark.parametrize("T", [16, 22, 64, 600])
def test_resample_fwd_and_adjoint(ops, T):
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 3, 5, T, generator=g)
    xc = x.cuda()
    dn = ops.resample(xc, torch.empty(2, 3, 5, T // 2, device="cuda"), 0)
    up = ops.resample(xc, torch.empty(2, 3, 5, 2 * T, device="cuda"), 1)
    assert rel(dn, UN.resample_down(x.double())) < 1e-6
    assert rel(up, UN.resample_up(x.double())) < 1e-6
    for mode, fwd, To in ((2, UN.resample_down, T // 2), (3, UN.resample_up, 2 * T)):
        gy = torch.randn(2, 3, 5, To, generator=g)
        xr = x.double().requires_grad_(True)
        gref, = torch.autograd.grad((fwd(xr) * gy.double()).sum(), xr)
        gx = ops.resample(gy.cuda(), torch.empty(2, 3, 5, T, device="cuda"), mode)
        assert rel(gx, gref) < 1e-6
    # alpha/beta + strided output
    big = torch.ones(2, 3, 9, T // 2, device="cuda")
    ops.resample(xc, big[:, :, 4:, :], 0, alpha=2.0, beta=1.0)
    assert rel(big[:, :, 4:, :], 2 * UN.resample_down(x.double()) + 1) < 1e-6
    assert float((big[:, :, :4, :] - 1).abs().max()) == 0.0


@pytest.mark.parametrize("mode,T", [(2, 64), (2, 1024), (3, 64), (0, 256), (1, 64)])
def test_resample_with_a_separate_residual_is_copy_then_accumulate_bit_for_bit(ops, mode, T):
    """babe_resample_res: out = alpha R(x) + beta res in one pass == copy res -> out, then out = alpha R(x) + beta out; res a
    channel-slice view (the encoder VJP's g_skip = gcat[:, N:]), out dense."""
    g = torch.Generator().manual_seed(100 * mode + T)
    Tin = {0: T, 1: T, 2: T // 2, 3: 2 * T}[mode]
    To = {0: T // 2, 1: 2 * T, 2: T, 3: T}[mode]
    x = torch.randn(2, 4, 6, Tin, generator=g).cuda()
    cat = torch.randn(2, 8, 6, To, generator=g).cuda()
    res = cat[:, 4:]
    rs2 = 1.0 / math.sqrt(2.0)
    o1 = torch.empty(2, 4, 6, To, device="cuda")
    ops.axpby(res, o1)
    ops.resample(x, o1, mode, alpha=rs2, beta=1.0)
    o2 = ops.resample(x, torch.empty(2, 4, 6, To, device="cuda"), mode, alpha=rs2, beta=1.0, res=res)
    assert torch.equal(o1, o2)


def test_axpby_linear_rff(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 8, 40, generator=g)
    big = torch.full((2, 3, 20, 40), 2.0, device="cuda")
    ops.axpby(x.cuda(), big[:, :, 12:, :], alpha=0.5, beta=0.25)
    assert rel(big[:, :, 12:, :], 0.5 * x + 0.5) < 1e-7
    xm = torch.randn(2, 256, generator=g)
    W = torch.randn(100, 256, generator=g)
    bias = torch.randn(100, generator=g)
    assert rel(ops.linear(xm.cuda(), W.cuda(), bias.cuda(), relu=True), torch.relu(xm @ W.t() + bias)) < 2e-6
    cn = torch.tensor([[-2.3], [0.1]])
    fr = 16 * torch.randn(1, 32, generator=g)
    tab = 2 * math.pi * cn * fr
    ref = torch.cat([torch.sin(tab), torch.cos(tab)], 1)
    assert float((ops.rff(cn.cuda(), fr.cuda()).cpu() - ref).abs().max()) < 2e-4


def test_gn_bwd_with_merged_block_tail_equals_two_passes_bit_for_bit(ops):
    """babe_gn_bwd_apply_merge: out = ca*acc + cb*gx in the GroupNorm-VJP pass itself == gn_bwd (gx stored) followed by
    axpby2(acc, gx): the tail of an N -> N ResnetBlock's VJP (unet_engine.block_vjp)."""
    g = torch.Generator().manual_seed(23)
    rs2 = 1.0 / math.sqrt(2.0)
    for (B, C, F, T) in [(2, 16, 12, 64), (1, 64, 40, 128)]:
        x = torch.randn(B, C, F, T, generator=g).cuda()
        gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
        film = (0.2 * torch.randn(B, C, generator=g)).cuda()
        a = torch.empty_like(x)
        stats, scale = ops.gn_scale_gelu(x, gamma, film, a)
        da = torch.randn(B, C, F, T, generator=g).cuda()
        gy = torch.randn(B, C, F, T, generator=g).cuda()
        acc = torch.randn(B, C, F, T, generator=g).cuda()
        gx = ops.gn_bwd(x, da, gy, scale, stats, torch.empty_like(x), rs2)
        want = ops.axpby2(acc, gx, torch.empty_like(x), rs2, rs2)
        got = ops.gn_bwd(x, da, gy, scale, stats, torch.empty_like(x), rs2, merge=(acc, rs2, rs2))
        assert torch.equal(want, got), (B, C, F, T)


def test_axpby2_is_the_two_pass_residual_merge_bit_for_bit(ops):
    """babe_axpby2_4d (out = a x + b y in one pass, the (x + h)/sqrt2 merge of cqtdiff+.py:493) == axpby(x -> out, a) followed by
    axpby(y, out, b, 1), incl. a strided frequency sub-view as the destination and the unaligned fallback."""
    g = torch.Generator().manual_seed(19)
    rs2 = 1.0 / math.sqrt(2.0)
    for (B, C, F, T, f0) in [(2, 8, 16, 64, 0), (1, 5, 12, 40, 8), (1, 3, 5, 6, 0)]:          # last: F*T % 4 != 0 -> fallback
        x = torch.randn(B, C, F, T, generator=g).cuda()
        y = torch.randn(B, C, F, T, generator=g).cuda()
        big1 = torch.zeros(B, C, F + f0 + 4, T, device="cuda")
        big2 = torch.zeros_like(big1)
        o1, o2 = big1[:, :, f0:f0 + F, :], big2[:, :, f0:f0 + F, :]
        ops.axpby(x, o1, alpha=rs2)
        ops.axpby(y, o1, alpha=rs2, beta=1.0)
        ops.axpby2(x, y, o2, rs2, rs2)
        assert torch.equal(big1, big2), (B, C, F, T)
        assert rel(o2, rs2 * (x.cpu() + y.cpu())) < 1e-6


def test_unet_body_fwd_and_vjp_small():
    """UNet body (between CQT.fwd and CQT.bwd) on the HIP engine vs the oracle with autograd."""
    import os
    import numpy as np
    from babe_amd.networks.unet_engine import UnetEngine
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "unet_small.npz")).items()}
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=[2, 3, 4, 5, 6, 7, 7])
    Ns = [8, 8, 8, 8, 16, 16, 16]
    gen = torch.Generator().manual_seed(31)
    B = 2
    Ts = [16 * 2 ** j for j in range(7)]
    C_list = [torch.randn(B, 2, 64, T, generator=gen) for T in Ts]
    cn = torch.tensor([[-0.4], [-1.1]])
    Cr = [c.double().requires_grad_(True) for c in C_list]
    sdd = {k: v.double() for k, v in sd.items()}
    emb = UN.embedding(sdd, cn.double())
    outs_ref = UN.unet_body(sdd, cfg, Cr, emb)
    gouts = [torch.randn(o.shape, generator=gen) for o in outs_ref]
    grefs = torch.autograd.grad(sum((o * go.double()).sum() for o, go in zip(outs_ref, gouts)), Cr)
    eng = UnetEngine({k: v.cuda().float() for k, v in sd.items()}, Ns, cfg["num_dils"])
    film = eng.embed(cn.cuda())
    outs = eng.forward([c.cuda() for c in C_list], film)
    for o, r in zip(outs, outs_ref):
        assert rel(o, r) < 2e-5
    gC = eng.vjp([go.cuda() for go in gouts])
    for a, r in zip(gC, grefs):
        assert rel(a, r) < 1e-4


@pytest.mark.parametrize("B,C1,C2,Cout,Fq,T,dil", [
    (2, 64, 0, 64, 64, 1024, 2),       # 64-channel tile, two time tiles per row
    (1, 96, 96, 96, 40, 36, 4),        # two sources (split % 32 == 0), 96 channels on the 128-channel tile, T < tile
    (1, 128, 0, 256, 33, 128, 16),     # two channel blocks, ragged F, taps skipped at the borders
    (2, 72, 0, 128, 9, 272, 1),        # Cin % 32 != 0 (zero channel groups), T not a power of two
    (1, 32, 32, 64, 70, 16, 8),        # shortest rows: 32 rows per tile, 64 halo lanes
    (1, 256, 0, 128, 24, 2048, 1),     # long rows, four time tiles
])
def test_conv2d_bf16_pipelined_vs_rounded_operands(ops, B, C1, C2, Cout, Fq, T, dil):
    """The pipelined bf16 (5,3) kernel (csrc/conv_bf16p.hip) computes EXACTLY conv(bf16(x*in_scale), bf16(w)) with fp32
    accumulation: against float64 on the rounded operands the error is accumulation rounding only (1e-6), which pins
    staging, swizzle, halo, zero padding and the two-source split independently of the bf16 quantisation (1e-2)."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(C1 + Cout + T)
    Cin = C1 + C2
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    rb = lambda t: t.to(torch.bfloat16).double()
    ref = 0.7 * UN.conv_same(rb(x), rb(w), dil) * osc[:, :, None, None].double() + 0.3 * res.double()
    pc = ops.PackedConv(w.cuda(), "bf16")
    xc = x.cuda()
    x1, x2 = (xc[:, :C1].contiguous(), xc[:, C1:].contiguous()) if C2 else (xc, None)
    big = torch.zeros(B, Cout, 2 * Fq, T, device="cuda")
    o = big[:, :, Fq:, :]
    o.copy_(res.cuda())
    c0 = dispatch_counts()["conv_bf16p"]
    ops.conv2d(x1, pc, o, dil=dil, x2=x2, res=o, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
    assert dispatch_counts()["conv_bf16p"] == c0 + 1
    assert rel(o, ref) < 3e-6
    assert float(big[:, :, :Fq, :].abs().max()) == 0.0
    # input-VJP with the folded per-channel scale: bf16(gy * isc) against bf16(w), transposed + flipped
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    gs = (gy * isc[:, :, None, None])
    wt = w.flip(2, 3).transpose(0, 1)
    gref = UN.conv_same(rb(gs), rb(wt), dil)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda())
    assert dispatch_counts()["conv_bf16p"] == c0 + 2
    assert rel(gx, gref) < 3e-6


@pytest.mark.parametrize("B,C1,C2,Cout,Fq,T", [
    (2, 64, 64, 64, 64, 1024),         # two sources, 64-channel tile
    (1, 96, 0, 256, 40, 36),           # two channel blocks, T < tile
    (2, 256, 256, 128, 33, 64),        # wide two-source input (decoder proj_in), ragged F
    (1, 72, 0, 96, 9, 272),            # Cin % 32 != 0, 96 channels on the 128-channel tile
])
def test_conv11_bf16_pipelined_vs_rounded_operands(ops, B, C1, C2, Cout, Fq, T):
    """(1,1) convs on the pipelined bf16 kernel (one tap, no halo): exact conv of the bf16-rounded operands, forward with
    residual / gate epilogue and input-VJP with the folded scale accumulated in place (the decoder's proj_in VJP)."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(C1 + Cout + T)
    Cin = C1 + C2
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    rb = lambda t: t.to(torch.bfloat16).double()
    ref = 0.7 * UN.conv_same(rb(x), rb(w)) * osc[:, :, None, None].double() + 0.3 * res.double()
    pc = ops.PackedConv(w.cuda(), "bf16")
    assert pc.splits == 1
    xc = x.cuda()
    x1, x2 = (xc[:, :C1].contiguous(), xc[:, C1:].contiguous()) if C2 else (xc, None)
    o = res.cuda().clone()
    c0 = dispatch_counts()["conv_bf16p"]
    ops.conv2d(x1, pc, o, x2=x2, res=o, oscale=osc.cuda(), alpha=0.7, rbeta=0.3)
    assert dispatch_counts()["conv_bf16p"] == c0 + 1
    assert rel(o, ref) < 3e-6
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    g0 = torch.randn(B, Cin, Fq, T, generator=g)
    gref = 0.5 * UN.conv_same(rb(gy * isc[:, :, None, None]), rb(w.transpose(0, 1))) + g0.double()
    gx = g0.cuda().clone()
    ops.conv2d(gy.cuda(), pc, gx, transpose=True, in_scale=isc.cuda(), res=gx, alpha=0.5, rbeta=1.0)
    assert dispatch_counts()["conv_bf16p"] == c0 + 2
    assert rel(gx, gref) < 3e-6


def test_bf16_precision_routes_hbm_bound_convs_to_fp32_kernels(ops):
    w = torch.randn(2, 64, 1, 1).cuda()
    assert ops.PackedConv(w, "bf16").splits == 0 and ops.PackedConv(w.transpose(0, 1).contiguous(), "bf16").splits == 0
    assert ops.PackedConv(torch.randn(64, 2, 5, 3).cuda(), "bf16").splits == 0            # pyramid conv
    assert ops.PackedConv(torch.randn(128, 64, 1, 1).cuda(), "bf16").splits == 1
    assert ops.PackedConv(torch.randn(128, 64, 1, 1).cuda(), "bf16x3").splits == 0
    assert ops.PackedConv(torch.randn(128, 64, 5, 3).cuda(), "bf16x3").splits == 2


@pytest.mark.parametrize("B,C,Fq,T,dil", [
    (2, 64, 64, 1024, 2),      # 64-channel shape, two time tiles per row
    (1, 96, 40, 36, 4),        # 96 channels on the 128-channel tile, T < tile (entries beyond T/4 read as zeros)
    (1, 128, 33, 128, 16),     # ragged F, taps skipped at the borders
    (2, 256, 70, 16, 8),       # shortest rows: 32 rows per tile
    (1, 72, 9, 272, 1),        # Cin % 32 != 0, T not a power of two
])
def test_bf16_units_path_is_bit_identical_to_fp32_staging(ops, B, C, Fq, T, dil):
    """precision='bf16' forward: GroupNorm-scale*GELU written as bf16 units (babe_scale_gelu_units) + all-DMA conv
    (babe_conv2d_bf16_units) == scale_gelu to fp32 + the staging kernel, bit for bit (same rounding, same MFMA order)."""
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(B, C, Fq, T, generator=g).cuda()
    sc = (torch.rand(B, C, generator=g) + 0.5).cuda()
    w = (torch.randn(C, C, 5, 3, generator=g) / math.sqrt(C * 15)).cuda()
    res = torch.randn(B, C, Fq, T, generator=g).cuda()
    osc = torch.randn(B, C, generator=g).cuda()
    pc = ops.PackedConv(w, "bf16")
    assert ops.units_ok(pc, C, C, T)
    a = torch.empty_like(x)
    ops.scale_gelu(x, sc, a)
    ref = ops.conv2d(a, pc, torch.empty_like(x), dil=dil, res=res, oscale=osc, alpha=0.7, rbeta=0.3)
    from babe_amd._lib import lib
    au = torch.full((B * lib().babe_units_size(C, Fq, T) * 8,), 0x7fc0, dtype=torch.int16, device="cuda")   # NaN-filled
    ops.scale_gelu_units(x, sc, au)
    out = ops.conv2d_units(au, pc, torch.empty_like(x), C, dil=dil, res=res, oscale=osc, alpha=0.7, rbeta=0.3)
    assert torch.equal(out, ref)
    # the unit tensor itself: plane p entry j = bf16(gelu(x*sc)) at t = 4j + p - 1, zeros outside [0, T)
    u = au.view(torch.bfloat16).view(B, C // 8, Fq, 4, T // 4 + 1, 8).float()
    full = torch.zeros(B, C // 8, Fq, T + 4, 8, device="cuda")
    full[:, :, :, 1:T + 1] = a.to(torch.bfloat16).float().view(B, C // 8, 8, Fq, T).permute(0, 1, 3, 4, 2)
    want = full.view(B, C // 8, Fq, T // 4 + 1, 4, 8).permute(0, 1, 2, 4, 3, 5)
    assert torch.equal(u, want)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_large_lds_kernels_on_a_second_device(ops):
    """hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute (csrc/common.h::babe_lds_optin): the kernels that
    ask for more than 64 KB of LDS must launch on cuda:1 after they have run on cuda:0 in the same process."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 128, 64, 128, generator=g)
    outs = []
    for shape in ((128, 128, 5, 3), (128, 128, 1, 1)):             # conv_wino4p (144 KB) and conv11p
        w = torch.randn(*shape, generator=g) / math.sqrt(128 * shape[2] * shape[3])
        ref = UN.conv_same(x.double(), w.double(), 2 if shape[2] > 1 else 1)
        for dev in ("cuda:0", "cuda:1"):
            with torch.cuda.device(dev):
                pc = ops.PackedConv(w.to(dev))
                out = torch.empty(1, 128, 64, 128, device=dev)
                ops.conv2d(x.to(dev), pc, out, dil=2 if shape[2] > 1 else 1)
                torch.cuda.synchronize()
            assert rel(out, ref) < 3e-6, (dev, shape)


def test_gelu_one_exponential_form_is_within_fp32_roundoff_of_erf(ops):
    """csrc/norm.hip evaluates Phi(u) by Abramowitz-Stegun 7.1.26 sharing one exponential with phi(u): bound the
    deviation from the exact erf form over [-8, 8] - |gelu - exact| <= 1e-6 absolute (|u| Phi error 7.5e-8 * |u|, plus
    fp32 round-off), i.e. below the conv kernels' own rounding at the activations' O(1) scale."""
    u = torch.linspace(-8.0, 8.0, 64 * 4096).view(1, 8, 8, 4096).contiguous()
    out = torch.empty_like(u, device="cuda")
    ops.scale_gelu(u.cuda(), torch.ones(1, 8, device="cuda"), out)
    exact = 0.5 * u.double() * (1.0 + torch.erf(u.double() / math.sqrt(2.0)))
    err = (out.cpu().double() - exact).abs().max().item()
    print(f"GELU (one-exponential form) max abs deviation from erf form on [-8, 8]: {err:.2e}")
    assert err < 1e-6


def test_nested_winograd_wide_and_64_channel_tiles_are_bit_identical(ops):
    """conv_wino45w (128- / 96-channel tiles, transform over 16 input channels at a time) and conv_wino45 (64-channel tiles)
    do the same arithmetic in the same order: a 128- (192-) channel conv on the wide kernel equals, bit for bit, the same
    conv done as two (three) 64-channel convs on the 64-channel-tile kernel."""
    from babe_amd._lib import dispatch_counts
    for Cout, Cin, Fq, T, dil in ((128, 128, 48, 128, 2), (192, 96, 32, 256, 1)):
        g = torch.Generator().manual_seed(Cout + Cin)
        x = torch.randn(2, Cin, Fq, T, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)).cuda()
        out = torch.empty(2, Cout, Fq, T, device="cuda")
        dispatch_counts(reset=True)
        ops.conv2d(x, ops.PackedConv(w), out, dil=dil, force_nested=True)
        parts = torch.empty_like(out)
        for c0 in range(0, Cout, 64):
            ops.conv2d(x, ops.PackedConv(w[c0:c0 + 64].contiguous()), parts[:, c0:c0 + 64], dil=dil, force_nested=True)
        assert dispatch_counts()["conv53_wino45"] == 1 + Cout // 64
        assert torch.equal(out, parts), float((out - parts).abs().max())


@pytest.mark.parametrize("B,C1,C2,Cout,Fq,T,dil", [
    (1, 64, 0, 64, 64, 128, 1),        # whole tiles
    (2, 64, 0, 64, 64, 64, 2),         # two residue classes, T = one tile
    (1, 128, 0, 128, 40, 100, 4),      # ragged: T % 64 != 0, rows per class 10 -> 5 pairs -> 2 groups with empty slots
    (1, 96, 0, 96, 56, 64, 8),         # 96 channels (second 64-channel tile half empty), 7 rows per class (odd)
    (1, 256, 0, 256, 28, 64, 4),       # long reduction (32 slabs per pass)
    (1, 128, 0, 128, 24, 192, 16),     # dilation > rows per class / 2
    (2, 48, 0, 72, 20, 68, 1),         # channel counts that are not multiples of 64
    (1, 128, 0, 64, 448, 64, 64),      # the benchmark's deepest geometry: 7 rows per class
    (2, 256, 0, 128, 32, 128, 2),      # wide kernel (128-channel tiles), B = 2, 1 row pair x 128 steps
    (1, 96, 0, 192, 48, 256, 1),       # wide kernel with two 96-channel tiles (second one at channel offset 96)
    (1, 128, 0, 256, 12, 68, 4),       # wide kernel, two 128-channel tiles, ragged T, 3 rows per class (odd)
])
def test_conv2d_nested_winograd_vs_float64(ops, B, C1, C2, Cout, Fq, T, dil):
    """csrc/conv_wino45.hip - F(2,5) along frequency x F(4,3) along time in three accumulator-carried passes - against the
    float64 direct convolution: forward with the fused epilogue, and the input-VJP form with a per-channel input scale.
    Bound 1e-5 relative (measured 1.5-3.5e-6: 3.4x the F(4,3) kernel's rounding, csrc/conv_wino45.hip header)."""
    from babe_amd._lib import dispatch_counts
    Cin = C1 + C2
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + dil)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    ref = UN.conv_same(x.double(), w.double(), dil)
    pc = ops.PackedConv(w.cuda())
    assert pc.fwd_wino45 is not None
    xs = x.cuda()
    x1, x2 = (xs[:, :C1].contiguous(), xs[:, C1:].contiguous()) if C2 else (xs, None)
    out = torch.empty(B, Cout, Fq, T, device="cuda")
    dispatch_counts(reset=True)
    ops.conv2d(x1, pc, out, dil=dil, x2=x2, force_nested=True)
    assert dispatch_counts()["conv53_wino45"] == 1
    e0 = rel(out, ref)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    out2 = res.cuda().clone()
    ops.conv2d(x1, pc, out2, dil=dil, x2=x2, res=out2, oscale=osc.cuda(), alpha=0.7, rbeta=0.3, force_nested=True)
    ref2 = 0.7 * ref * osc[:, :, None, None].double() + 0.3 * res.double()
    e1 = rel(out2, ref2)
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    xr = x.double().requires_grad_(True)
    y = UN.conv_same(xr, w.double(), dil)
    gref, = torch.autograd.grad((y * (gy * isc[:, :, None, None]).double()).sum(), xr)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    dispatch_counts(reset=True)
    ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda(), force_nested=True)
    n45 = dispatch_counts()["conv53_wino45"]
    e2 = rel(gx, gref)
    print(f"nested Winograd Cin={Cin} Cout={Cout} F={Fq} T={T} dil={dil}: fwd {e0:.2e}, epilogue {e1:.2e}, vjp {e2:.2e} (wino45 launches {n45})")
    assert e0 < 1e-5 and e1 < 1e-5 and e2 < 1e-5


@pytest.mark.parametrize("B,Cin,Cout,Fq,T,dil", [
    (1, 128, 128, 48, 128, 2),         # whole tiles
    (1, 256, 256, 28, 64, 4),          # 7 rows per class: the second row quad has one empty row; two channel blocks
    (2, 128, 256, 40, 100, 1),         # ragged T (100 = 64 + 36), B = 2
    (1, 96, 128, 56, 64, 8),           # 96 input channels = 6 super-slabs per pass; the VJP has one 96-channel tile
    (1, 128, 128, 24, 192, 16),        # dilation above the rows per class: classes of 1 and 2 rows
    (1, 256, 128, 448, 64, 64),        # the benchmark's deepest geometry
    (1, 16, 128, 9, 64, 1),            # ONE 16-channel super-slab per pass; 9 rows = 3 quads, the last with one row
    (3, 144, 384, 21, 132, 3),         # three channel blocks, odd dilation, T % 64 = 4, B = 3
    (1, 96, 96, 40, 128, 2),           # 96-channel tile: six multiplying + two transform waves (conv_wino85s_kernel), forward and VJP
    (2, 128, 192, 20, 68, 1),          # two 96-channel tiles, ragged T; the VJP has one 128-channel tile
    (1, 64, 64, 24, 128, 1),           # 64-channel tile: four multiplying + four transform waves
    (2, 32, 64, 36, 100, 3),           # 64-channel tile, two super-slabs per pass, odd dilation (the VJP's 32 channels are not this kernel's)
    (1, 96, 320, 20, 64, 4),           # 320 = five 64-channel tiles
])
def test_conv2d_nested_winograd_f45_vs_float64(ops, B, Cin, Cout, Fq, T, dil):
    """csrc/conv_wino85.hip - F(4,5) along frequency x F(4,3) along time, two accumulator-carried passes of four frequency phases -
    against the float64 direct convolution: forward with the fused epilogue and the input-VJP form with a per-channel input scale.
    Bound 1e-5 relative, the bar of the F(2,5) x F(4,3) kernel (measured 3.5-5.5e-6: the 8-point set 0, +-1, +-2, +-1/2, inf rounds
    ~1.6x worse than the 6-point one, profiles/r05_wino_f45_estimate.txt)."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + dil)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
    ref = UN.conv_same(x.double(), w.double(), dil)
    pc = ops.PackedConv(w.cuda())
    assert pc.fwd_wino85 is not None
    out = torch.empty(B, Cout, Fq, T, device="cuda")
    dispatch_counts(reset=True)
    ops.conv2d(x.cuda(), pc, out, dil=dil, force_f45=True)
    assert dispatch_counts()["conv53_wino85"] == 1
    e0 = rel(out, ref)
    res = torch.randn(B, Cout, Fq, T, generator=g)
    osc = torch.randn(B, Cout, generator=g)
    out2 = res.cuda().clone()
    ops.conv2d(x.cuda(), pc, out2, dil=dil, res=out2, oscale=osc.cuda(), alpha=0.7, rbeta=0.3, force_f45=True)
    e1 = rel(out2, 0.7 * ref * osc[:, :, None, None].double() + 0.3 * res.double())
    gy = torch.randn(B, Cout, Fq, T, generator=g)
    isc = torch.randn(B, Cout, generator=g)
    xr = x.double().requires_grad_(True)
    y = UN.conv_same(xr, w.double(), dil)
    gref, = torch.autograd.grad((y * (gy * isc[:, :, None, None]).double()).sum(), xr)
    gx = torch.empty(B, Cin, Fq, T, device="cuda")
    dispatch_counts(reset=True)
    ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda(), force_f45=True)
    n85 = dispatch_counts()["conv53_wino85"]
    assert n85 == (1 if Cin % 64 == 0 or Cin % 96 == 0 else 0)           # (the transposed conv has Cin output channels)
    e2 = rel(gx, gref)
    print(f"F(4,5) x F(4,3) Cin={Cin} Cout={Cout} F={Fq} T={T} dil={dil}: fwd {e0:.2e}, epilogue {e1:.2e}, vjp {e2:.2e} (wino85 launches {n85})")
    assert e0 < 1e-5 and e1 < 1e-5 and e2 < 1e-5


def test_conv2d_f45_random_shapes_vs_float64(ops):
    """Twelve seeded random problems across the kernel's supported domain - Cin 16 .. 272 in steps of 16, Cout from the three tile
    widths and their multiples, 5 .. 70 rows, T a multiple of 4 from 64, dilation 1 .. 9, B 1 .. 3, with and without the fused
    epilogue / input scale - forward and (where the transposed problem is supported) input-VJP against the float64 direct conv."""
    import random
    from babe_amd._lib import dispatch_counts
    rnd = random.Random(4543)
    for case in range(12):
        Cin = 16 * rnd.randint(1, 17)
        Cout = rnd.choice([64, 96, 128, 192, 256, 320])
        Fq, T, dil, B = rnd.randint(5, 70), 4 * rnd.randint(16, 50), rnd.randint(1, 9), rnd.randint(1, 3)
        g = torch.Generator().manual_seed(case)
        x = torch.randn(B, Cin, Fq, T, generator=g)
        w = torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)
        ref = UN.conv_same(x.double(), w.double(), dil)
        pc = ops.PackedConv(w.cuda())
        out = torch.empty(B, Cout, Fq, T, device="cuda")
        fused = case % 2 == 1
        res = torch.randn(B, Cout, Fq, T, generator=g)
        osc = torch.randn(B, Cout, generator=g)
        dispatch_counts(reset=True)
        if fused:
            out.copy_(res)
            ops.conv2d(x.cuda(), pc, out, dil=dil, res=out, oscale=osc.cuda(), alpha=1.3, rbeta=-0.4, force_f45=True)
            ref = 1.3 * ref * osc[:, :, None, None].double() - 0.4 * res.double()
        else:
            ops.conv2d(x.cuda(), pc, out, dil=dil, force_f45=True)
        assert dispatch_counts()["conv53_wino85"] == 1
        e0 = rel(out, ref)
        e1 = float("nan")
        if pc.bwd_wino85 is not None:
            gy = torch.randn(B, Cout, Fq, T, generator=g)
            isc = torch.randn(B, Cout, generator=g) if fused else None
            xr = x.double().requires_grad_(True)
            y = UN.conv_same(xr, w.double(), dil)
            gsrc = gy.double() * isc[:, :, None, None].double() if fused else gy.double()
            gref, = torch.autograd.grad((y * gsrc).sum(), xr)
            gx = torch.empty(B, Cin, Fq, T, device="cuda")
            dispatch_counts(reset=True)
            ops.conv2d(gy.cuda(), pc, gx, dil=dil, transpose=True, in_scale=isc.cuda() if fused else None, force_f45=True)
            assert dispatch_counts()["conv53_wino85"] == 1
            e1 = rel(gx, gref)
        print(f"case {case}: B={B} Cin={Cin} Cout={Cout} F={Fq} T={T} dil={dil} fused={fused}: fwd {e0:.2e} vjp {e1:.2e}")
        assert e0 < 1e-5 and not (e1 > 1e-5)


def test_f45_tile_widths_are_bit_identical(ops):
    """conv_wino85_kernel (128-channel tiles, every wave transforms and multiplies) and conv_wino85s_kernel (96- / 64-channel tiles,
    specialised waves) do the same arithmetic in the same order for a given output: a 128-channel conv on the first equals, bit for
    bit, the same conv done as two 64-channel convs on the second; a 192-channel conv (two 96-channel tiles) equals three
    64-channel convs."""
    from babe_amd._lib import dispatch_counts, lib
    for Cout, Cin, Fq, T, dil, waves in ((128, 128, 48, 128, 2, 12), (128, 128, 48, 128, 2, 8), (192, 96, 32, 196, 1, 12), (256, 128, 56, 64, 8, 12)):
        # (round 6: the 128-channel tile as 8 multiplying + 4 transform waves - the default - and as the 8-wave kernel)
        assert lib().babe_conv2d_wino85_set_waves(waves) == 0
        g = torch.Generator().manual_seed(Cout + Cin)
        x = torch.randn(2, Cin, Fq, T, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 5, 3, generator=g) / math.sqrt(Cin * 15)).cuda()
        isc = torch.randn(2, Cin, generator=g).cuda()
        out = torch.empty(2, Cout, Fq, T, device="cuda")
        dispatch_counts(reset=True)
        ops.conv2d(x, ops.PackedConv(w), out, dil=dil, in_scale=isc, force_f45=True)
        parts = torch.empty_like(out)
        for c0 in range(0, Cout, 64):
            ops.conv2d(x, ops.PackedConv(w[c0:c0 + 64].contiguous()), parts[:, c0:c0 + 64], dil=dil, in_scale=isc, force_f45=True)
        assert dispatch_counts()["conv53_wino85"] == 1 + Cout // 64
        assert torch.equal(out, parts), float((out - parts).abs().max())
    assert lib().babe_conv2d_wino85_set_waves(12) == 0


def test_conv2d_f45_dispatch_rule_and_tile_order(ops):
    """Which launches take the F(4,5) x F(4,3) kernel by default: 128-channel output tiles whose row quads x time tiles are >= 80 %
    full (babe_conv2d_wino85_preferred); force_nested keeps the F(2,5) x F(4,3) kernel; two-source convs never take it.  The
    XCD-contiguous tile order only renumbers the workgroups: results are bit-identical for every batch item."""
    from babe_amd._lib import dispatch_counts
    g = torch.Generator().manual_seed(85)
    w = (torch.randn(128, 128, 5, 3, generator=g) / math.sqrt(128 * 15)).cuda()
    pc = ops.PackedConv(w)
    for Fq, T, dil, want in ((64, 128, 4, 1), (40, 64, 8, 0), (48, 64, 8, 0), (56, 64, 2, 1), (64, 68, 1, 0)):
        # rows per class / quads: 16/4 full; 5 -> 2 quads = 0.625; 6 -> 2 quads = 0.75; 28 -> 7 quads full; T 68 of 128 = 0.53
        x = torch.randn(2, 128, Fq, T, generator=g).cuda()
        out = torch.empty(2, 128, Fq, T, device="cuda")
        dispatch_counts(reset=True)
        ops.conv2d(x, pc, out, dil=dil)
        c = dispatch_counts()
        assert c["conv53_wino85"] == want and c["conv53_wino85"] + c["conv53_wino45"] + c["conv53_wino4"] == 1, (Fq, T, dil, c)
        dispatch_counts(reset=True)
        out45 = torch.empty_like(out)
        ops.conv2d(x, pc, out45, dil=dil, force_nested=True)
        assert dispatch_counts()["conv53_wino45"] == 1
        assert rel(out, out45) < 1e-5
        x1 = torch.empty(1, 128, Fq, T, device="cuda").copy_(x[1:])
        o1 = torch.empty(1, 128, Fq, T, device="cuda")
        ops.conv2d(x1, pc, o1, dil=dil, force_f45=True)
        o2 = torch.empty_like(out)
        ops.conv2d(x, pc, o2, dil=dil, force_f45=True)
        assert torch.equal(o2[1:], o1)                                    # batch items are independent launches of the same tiles
    xa, xb = torch.randn(1, 64, 64, 128, generator=g).cuda(), torch.randn(1, 64, 64, 128, generator=g).cuda()
    out = torch.empty(1, 128, 64, 128, device="cuda")
    dispatch_counts(reset=True)
    ops.conv2d(xa, pc, out, dil=4, x2=xb)
    assert dispatch_counts()["conv53_wino85"] == 0


def test_f45_epilogue_forms_the_next_groupnorm_sums(ops, monkeypatch):
    """Round 6: the forward (5,3) conv that writes a layer's output also forms, in its epilogue, the sum and sum of squares per group
    that the NEXT layer's GroupNorm needs (babe_conv_args::stat_mode 1) - for the 128- (12- and 8-wave), 96-, 64- and 256-channel
    forms, ragged time tiles and partly filled row quads included: output bit-identical with and without the reduction, group totals
    against float64 sums of the output, and gn_scale_gelu on the fused sums against its own pass (statistics to float rounding,
    the activation to 1e-5)."""
    from babe_amd._lib import dispatch_counts, lib
    monkeypatch.setattr(ops, "FUSE_GN_FWD", True)
    RS2 = 1.0 / math.sqrt(2.0)
    for Cc, Fq, T, dil, waves in ((128, 48, 128, 2, 12), (128, 48, 128, 2, 8), (96, 40, 192, 1, 12), (64, 36, 100, 3, 12), (256, 28, 64, 4, 12),
                                  (128, 5, 64, 1, 12)):
        assert lib().babe_conv2d_wino85_set_waves(waves) == 0
        g = torch.Generator().manual_seed(Cc + Fq + T + 1)
        B = 2
        a_in = torch.randn(B, Cc, Fq, T, generator=g).cuda()
        z = torch.randn(B, Cc, Fq, T, generator=g).cuda()
        w = (torch.randn(Cc, Cc, 5, 3, generator=g) / math.sqrt(Cc * 15)).cuda()
        gate = torch.randn(B, Cc, generator=g).cuda()
        gamma = (0.5 + torch.rand(Cc, generator=g)).cuda()
        film = (0.1 * torch.randn(B, Cc, generator=g)).cuda()
        pc = ops.PackedConv(w)
        o0, o1 = torch.empty_like(z), torch.empty_like(z)
        dispatch_counts(reset=True)
        ops.conv2d(a_in, pc, o0, dil=dil, res=z, oscale=gate, alpha=RS2, rbeta=RS2, force_f45=True)
        fs = ops.conv2d(a_in, pc, o1, dil=dil, res=z, oscale=gate, alpha=RS2, rbeta=RS2, force_f45=True, fwd_stat=Cc // 8)
        assert dispatch_counts()["conv53_wino85"] == 2 and fs is not None
        assert torch.equal(o0, o1)
        part, S = fs
        assert S == lib().babe_conv2d_wino85_stat_slots(C.byref(_stat_args(Cc // 8, Fq, T, dil)))
        got = part.view(B, 8, S, 2).sum(2)
        x64 = o0.double().view(B, 8, -1)
        want = torch.stack([x64.sum(-1), (x64 * x64).sum(-1)], -1)
        err = float(((got - want).abs() / torch.stack([x64.abs().sum(-1), (x64 * x64).sum(-1)], -1)).max())
        assert err < 1e-12, (Cc, waves, err)
        act0, act1 = torch.empty_like(z), torch.empty_like(z)
        st0, sc0 = ops.gn_scale_gelu(o0, gamma, film, act0)
        st1, sc1 = ops.gn_scale_gelu(o0, gamma, film, act1, fused=fs)
        assert float((st0 - st1).abs().max()) < 1e-6 and float(((sc0 - sc1) / sc0).abs().max()) < 1e-6
        assert float((act0 - act1).abs().max()) < 1e-5
    assert lib().babe_conv2d_wino85_set_waves(12) == 0


def _stat_args(cg, F, T, dil):
    from babe_amd._lib import ConvArgs
    a = ConvArgs()
    a.stat_cg, a.F, a.T, a.dil = cg, F, T, dil
    return a


def test_f45_epilogue_forms_the_groupnorm_vjp_partial_sums(ops, monkeypatch):
    """(Opt-in, BABE_FUSE_GN=1: measured 0.5 % slower than the separate pass - profiles/r06_gn_fusion_experiment.txt.)  Round 6: the transposed (5,3) conv that writes da also forms, in its epilogue, the per-group sums S_g = sum da * u * gelu'(u)
    (u = scale * z) that babe_gn_bwd_partial would re-read z and da for (babe_conv_args::stat_mode 2) - for the 128- (12- and 8-wave),
    96- and 64-channel tile forms, ragged time tiles and partly filled row quads included: group totals against the stand-alone pass
    (both sum in double, in different orders), da bit-identical with and without the reduction, and gn_bwd on either set of partial
    sums to float rounding."""
    from babe_amd._lib import dispatch_counts, lib
    monkeypatch.setattr(ops, "FUSE_GN", True)
    for Cc, Fq, T, dil, waves in ((128, 48, 128, 2, 12), (128, 48, 128, 2, 8), (96, 40, 192, 1, 12), (64, 36, 100, 3, 12), (256, 28, 64, 4, 12)):
        assert lib().babe_conv2d_wino85_set_waves(waves) == 0
        g = torch.Generator().manual_seed(Cc + Fq + T)
        B = 2
        z = torch.randn(B, Cc, Fq, T, generator=g).cuda()
        src = torch.randn(B, Cc, Fq, T, generator=g).cuda()
        w = (torch.randn(Cc, Cc, 5, 3, generator=g) / math.sqrt(Cc * 15)).cuda()
        gate = torch.randn(B, Cc, generator=g).cuda()
        scale = (0.5 + torch.rand(B, Cc, generator=g)).cuda()
        pc = ops.PackedConv(w)
        da0, da1 = torch.empty_like(z), torch.empty_like(z)
        dispatch_counts(reset=True)
        ops.conv2d(src, pc, da0, dil=dil, transpose=True, in_scale=gate, alpha=0.7, force_f45=True)
        fs = ops.conv2d(src, pc, da1, dil=dil, transpose=True, in_scale=gate, alpha=0.7, force_f45=True, vjp_stat=(z, scale, Cc // 8))
        assert dispatch_counts()["conv53_wino85"] == 2 and fs is not None
        assert torch.equal(da0, da1)
        part, S = fs
        got = part.view(B, 8, S).sum(-1)
        n = (Cc // 8) * Fq * T
        Sp = ops._splits(n, B, 8)
        ref = torch.empty(B * 8 * Sp, device="cuda", dtype=torch.float64)
        from babe_amd._lib import check, ptr, stream
        check(lib().babe_gn_bwd_partial(ptr(z), ptr(da0), ptr(scale), ptr(ref), B, Cc, 8, Fq * T, Sp, stream()), "gn_bwd_partial")
        want = ref.view(B, 8, Sp).sum(-1)
        err = float(((got - want).abs() / (want.abs() + 1e-9 * want.abs().max())).max())
        assert err < 1e-9, (Cc, waves, err)
        stats = torch.stack([torch.zeros(B, 8), torch.ones(B, 8), torch.ones(B, 8)], -1).cuda().contiguous()
        gx0, gx1 = torch.empty_like(z), torch.empty_like(z)
        ops.gn_bwd(z, da0, src, scale, stats, gx0, 0.7)
        ops.gn_bwd(z, da0, src, scale, stats, gx1, 0.7, fused=fs)
        assert float((gx0 - gx1).abs().max() / gx0.abs().max()) < 1e-6
    assert lib().babe_conv2d_wino85_set_waves(12) == 0
