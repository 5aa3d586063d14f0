"""Python views of the babe_hip C-ABI ops (UNet building blocks).  Device tensors only."""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ConvArgs, check, lib, ptr, stream

RSQRT2 = 1.0 / math.sqrt(2.0)


def _view(t):
    """(ptr, batch stride, channel stride) of a [B,C,F,T] tensor whose rows are contiguous."""
    assert t.dim() == 4 and t.dtype == torch.float32
    assert t.stride(3) == 1 and t.stride(2) == t.shape[3], f"rows must be contiguous, got strides {t.stride()}"
    return ptr(t), t.stride(0), t.stride(1)


PRECISIONS = {"f32": 0, "bf16": 1, "bf16x3": 2}
BF16_HBM_F32 = os.environ.get("BABE_BF16_HBM_F32", "1") != "0"
# debug switch for the (5,3) fp32 convs: 0 = direct kernel only, 2 = Winograd F(2,3) only, 4 (default) = F(4,3) where the
# problem qualifies, F(2,3) otherwise
_W = os.environ.get("BABE_CONV_WINO", "4")
FEWCO = os.environ.get("BABE_CONV_FEWCO", "1") != "0"
WINOGRAD = _W != "0"
WINOGRAD4 = _W not in ("0", "2", "1")
# nested Winograd F(2,5) x F(4,3) (csrc/conv_wino45.hip) for the (5,3) layers it supports; BABE_CONV_WINO45=0 switches it off
WINOGRAD45 = WINOGRAD4 and os.environ.get("BABE_CONV_WINO45", "1") != "0"
# nested Winograd F(4,5) x F(4,3) (csrc/conv_wino85.hip) for the (5,3) layers with 128-channel output tiles whose row quads are
# at least 80 % full (babe_conv2d_wino85_preferred); BABE_CONV_F45=0 leaves them to the F(2,5) x F(4,3) kernel
WINOGRAD85 = WINOGRAD45 and os.environ.get("BABE_CONV_F45", "1") != "0"


# (1,1) convs: 64-channel output tiles per workgroup (two 32-row tiles) wherever both directions' tile counts are even, instead of
# the library default of up to 128: more workgroups on the small planes of the deep levels, where a launch has a few hundred
# (whole-job A/B, same box: 2.431 / 2.432 vs 2.423 / 2.424 audio-sec/s; one tile: 2.399 / 2.395).  0 = library default.
_C11_NT = int(os.environ.get("BABE_CONV11_NT", "2"))


class PackedConv:
    """Conv2d weights packed for babe_conv2d, forward and input-VJP (flipped/transposed) versions.
    precision: 'f32' (exact fp32 MFMA), 'bf16' or 'bf16x3' (bf16 MFMA, see csrc/conv_bf16.hip)."""

    def __init__(self, w, precision="f32", nt=0):
        """nt: row tiles (x32 output channels) per workgroup of the direct kernel, 0 = default (include/babe_hip.h)."""
        self.precision = precision
        self.nt = nt
        if nt == 0 and _C11_NT and w.shape[2] * w.shape[3] == 1 and ((w.shape[0] + 31) // 32) % _C11_NT == 0 \
                and ((w.shape[1] + 31) // 32) % _C11_NT == 0:
            self.nt = _C11_NT
        self.splits = PRECISIONS[precision]
        # Shapes that gain nothing from bf16 MFMA run on the fp32 kernels whatever the requested precision (exact AND at
        # least as fast): convs with <= 4 channels on one side (few-channel kernels), (1,1) kernels with fewer than 32
        # channels on one side (HBM-bound: all-DMA kernel), and every (1,1) kernel of the bf16x3 mode.  Wide (1,1) kernels are
        # MFMA-bound in fp32 (up to 85 flop/B) and take the pipelined bf16 kernel under 'bf16'.
        k11 = w.shape[2] * w.shape[3] == 1
        if self.splits and BF16_HBM_F32 and (min(w.shape[0], w.shape[1]) <= 4 or
                                             (k11 and (self.splits == 2 or min(w.shape[0], w.shape[1]) < 32))):
            self.splits = 0
        if self.splits:
            self._init_bf16(w)
            return
        self._init_f32(w)

    def _init_bf16(self, w):
        assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 4
        w = w.contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        L = lib()
        nf = L.babe_conv_packed_size_bf16(self.Cout, self.Cin, self.KH, self.KW, 0, self.splits)
        nb = L.babe_conv_packed_size_bf16(self.Cout, self.Cin, self.KH, self.KW, 1, self.splits)
        self.fwd = torch.empty(nf, device=w.device, dtype=torch.int16)
        self.bwd = torch.empty(nb, device=w.device, dtype=torch.int16)
        for tf, dst in ((0, self.fwd), (1, self.bwd)):
            check(L.babe_conv_pack_weights_bf16(ptr(w), ptr(dst), self.Cout, self.Cin, self.KH, self.KW, tf, self.splits,
                                                stream()), "pack_bf16")

    def _init_f32(self, w):
        assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 4
        w = w.contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        # raw weights for the few-output-channel kernel (the input-VJP of a 2..4-input-channel conv, csrc/conv_fewco.hip)
        self.w_raw = w if (self.KW == 3 and min(self.Cout, self.Cin) <= 4) else None
        L = lib()
        nf = L.babe_conv_packed_size(self.Cout, self.Cin, self.KH, self.KW, 0)
        nb = L.babe_conv_packed_size(self.Cout, self.Cin, self.KH, self.KW, 1)
        self.fwd = torch.empty(nf, device=w.device, dtype=torch.float32)
        self.bwd = torch.empty(nb, device=w.device, dtype=torch.float32)
        check(L.babe_conv_pack_weights_nt(ptr(w), ptr(self.fwd), self.Cout, self.Cin, self.KH, self.KW, 0, self.nt, stream()), "pack")
        check(L.babe_conv_pack_weights_nt(ptr(w), ptr(self.bwd), self.Cout, self.Cin, self.KH, self.KW, 1, self.nt, stream()), "pack")
        # Winograd F(2,3)-along-time images for the 3-tap kernels (used whenever the problem qualifies)
        self.fwd_wino = self.bwd_wino = None
        if self.KW == 3 and WINOGRAD:
            self.fwd_wino = torch.empty(L.babe_conv_packed_size_wino(self.Cout, self.Cin, self.KH, 0), device=w.device)
            self.bwd_wino = torch.empty(L.babe_conv_packed_size_wino(self.Cout, self.Cin, self.KH, 1), device=w.device)
            check(L.babe_conv_pack_weights_wino(ptr(w), ptr(self.fwd_wino), self.Cout, self.Cin, self.KH, self.KW, 0, stream()), "pack_wino")
            check(L.babe_conv_pack_weights_wino(ptr(w), ptr(self.bwd_wino), self.Cout, self.Cin, self.KH, self.KW, 1, stream()), "pack_wino")
        self.fwd_wino4 = self.bwd_wino4 = None
        if self.KW == 3 and WINOGRAD4:
            self.fwd_wino4 = torch.empty(L.babe_conv_packed_size_wino4(self.Cout, self.Cin, self.KH, 0), device=w.device)
            self.bwd_wino4 = torch.empty(L.babe_conv_packed_size_wino4(self.Cout, self.Cin, self.KH, 1), device=w.device)
            check(L.babe_conv_pack_weights_wino4(ptr(w), ptr(self.fwd_wino4), self.Cout, self.Cin, self.KH, self.KW, 0, stream()), "pack_wino4")
            check(L.babe_conv_pack_weights_wino4(ptr(w), ptr(self.bwd_wino4), self.Cout, self.Cin, self.KH, self.KW, 1, stream()), "pack_wino4")
        # nested-Winograd images (36 floats per weight pair and direction): only for the direction(s) the kernel can take -
        # babe_conv2d_wino45_supported needs the EXECUTED op's input channels % 16 == 0 and more than 32 output channels
        self.fwd_wino45 = self.bwd_wino45 = None
        if self.KH == 5 and self.KW == 3 and WINOGRAD45:
            if self.Cin % 16 == 0 and self.Cout > 32:
                self.fwd_wino45 = torch.empty(L.babe_conv_packed_size_wino45(self.Cout, self.Cin, 0), device=w.device)
                check(L.babe_conv_pack_weights_wino45(ptr(w), ptr(self.fwd_wino45), self.Cout, self.Cin, self.KH, self.KW, 0, stream()), "pack_wino45")
            if self.Cout % 16 == 0 and self.Cin > 32:
                self.bwd_wino45 = torch.empty(L.babe_conv_packed_size_wino45(self.Cout, self.Cin, 1), device=w.device)
                check(L.babe_conv_pack_weights_wino45(ptr(w), ptr(self.bwd_wino45), self.Cout, self.Cin, self.KH, self.KW, 1, stream()), "pack_wino45")
        self.fwd_wino85 = self.bwd_wino85 = None
        if self.KH == 5 and self.KW == 3 and WINOGRAD85:
            if self.Cin % 16 == 0 and self.Cout % 32 == 0 and (self.Cout % 128 == 0 or self.Cout % 96 == 0 or self.Cout % 64 == 0):   # its tile widths
                self.fwd_wino85 = torch.empty(L.babe_conv_packed_size_wino85(self.Cout, self.Cin, 0), device=w.device)
                check(L.babe_conv_pack_weights_wino85(ptr(w), ptr(self.fwd_wino85), self.Cout, self.Cin, self.KH, self.KW, 0, stream()), "pack_wino85")
            if self.Cout % 16 == 0 and (self.Cin % 128 == 0 or self.Cin % 96 == 0 or self.Cin % 64 == 0):
                self.bwd_wino85 = torch.empty(L.babe_conv_packed_size_wino85(self.Cout, self.Cin, 1), device=w.device)
                check(L.babe_conv_pack_weights_wino85(ptr(w), ptr(self.bwd_wino85), self.Cout, self.Cin, self.KH, self.KW, 1, stream()), "pack_wino85")


# GroupNorm-VJP partial sums formed in the F(4,5) transposed conv's epilogue instead of babe_gn_bwd_partial's own pass: OFF by
# default - measured 0.5 % slower than the separate pass (profiles/r06_gn_fusion_experiment.txt: the GELU' arithmetic runs on the
# multiply waves with nothing to overlap it); BABE_FUSE_GN=1 turns it on (tests/test_gpu_ops.py keeps it parity-checked)
FUSE_GN = os.environ.get("BABE_FUSE_GN", "0") != "0"
# The NEXT layer's GroupNorm sums (sum, sum of squares of the output) formed in the forward F(4,5) conv's epilogue instead of
# babe_gn_partial's pass over the freshly written output: two double additions per output, no extra loads.  BABE_FUSE_GN_FWD=0: own pass.
FUSE_GN_FWD = os.environ.get("BABE_FUSE_GN_FWD", "1") != "0"


def conv2d(x, pc, out, *, dil=1, transpose=False, x2=None, res=None, in_scale=None, oscale=None, alpha=1.0, rbeta=0.0,
           force_nested=False, force_f45=False, vjp_stat=None, fwd_stat=None):
    """out = alpha*conv(x[,x2]; W)*oscale + rbeta*res   (transpose=True: input-VJP weights).
    force_nested: take the nested-Winograd F(2,5) x F(4,3) kernel whenever it CAN run the problem (tests), not only when it is
    preferred; force_f45: the same for the F(4,5) x F(4,3) kernel.
    vjp_stat=(z, scale, cg): if the launch takes the F(4,5) kernel, its epilogue also forms the partial sums of the GroupNorm /
    FiLM / GELU input-VJP for the gradient `out` it writes (z: the layer's saved input, dense like out; scale [B,C]; cg channels
    per group) and (part, S) is RETURNED for gn_bwd(part=, S=); otherwise None is returned and gn_bwd runs its own pass.
    fwd_stat=cg: likewise the sums of the output itself - the next layer's GroupNorm partial sums - for gn_scale_gelu(fused=);
    (part, S) or None is returned."""
    a = ConvArgs()
    B, C1, F, T = x.shape
    Cin = pc.Cout if transpose else pc.Cin
    Cout = pc.Cin if transpose else pc.Cout
    a.in_, a.in_bs, a.in_cs = _view(x)
    if x2 is not None:
        assert x2.shape[0] == B and x2.shape[2:] == x.shape[2:]
        a.in2, a.in2_bs, a.in2_cs = _view(x2)
        a.cin_split = C1
        assert C1 + x2.shape[1] == Cin
    else:
        a.in2, a.in2_bs, a.in2_cs, a.cin_split = None, 0, 0, Cin
        assert C1 == Cin, (C1, Cin)
    wq = pc.bwd if transpose else pc.fwd
    a.w_packed = None if pc.splits else ptr(wq)
    assert out.shape == (B, Cout, F, T), (out.shape, (B, Cout, F, T))
    a.out, a.out_bs, a.out_cs = _view(out)
    if res is not None:
        assert res.shape == out.shape
        a.res, a.res_bs, a.res_cs = _view(res)
    else:
        a.res, a.res_bs, a.res_cs = None, 0, 0
    if in_scale is not None:
        assert in_scale.is_contiguous() and in_scale.shape == (B, Cin)
    if oscale is not None:
        assert oscale.is_contiguous() and oscale.shape == (B, Cout)
    a.in_scale, a.oscale = ptr(in_scale), ptr(oscale)
    a.alpha, a.rbeta = alpha, rbeta
    a.B, a.Cin, a.Cout, a.F, a.T = B, Cin, Cout, F, T
    a.KH, a.KW, a.dil = pc.KH, pc.KW, dil
    if pc.splits:
        check(lib().babe_conv2d_bf16(C.byref(a), ptr(wq), pc.splits, stream()), "conv2d_bf16")
    elif FEWCO and getattr(pc, "w_raw", None) is not None and Cout <= 4 and lib().babe_conv2d_fewco_supported(C.byref(a)):
        check(lib().babe_conv2d_fewco(C.byref(a), ptr(pc.w_raw), int(transpose), stream()), "conv2d_fewco")
    elif getattr(pc, "bwd_wino85" if transpose else "fwd_wino85", None) is not None and x2 is None and not force_nested and (
            lib().babe_conv2d_wino85_supported(C.byref(a)) if force_f45 else lib().babe_conv2d_wino85_preferred(C.byref(a))):
        fused = None
        if vjp_stat is not None and FUSE_GN:
            z, scale, cg = vjp_stat
            assert z.is_contiguous() and out.is_contiguous() and z.shape == out.shape and scale.is_contiguous()
            a.stat_mode, a.stat_cg, a.stat_x, a.stat_scale = 2, cg, ptr(z), ptr(scale)
            S = lib().babe_conv2d_wino85_stat_slots(C.byref(a))
            part = torch.empty(B * (Cout // cg) * S, device=x.device, dtype=torch.float64)
            a.stat_part = ptr(part)
            fused = (part, S)
        elif fwd_stat is not None and FUSE_GN_FWD and out.is_contiguous():
            a.stat_mode, a.stat_cg = 1, fwd_stat
            S = lib().babe_conv2d_wino85_stat_slots(C.byref(a))
            part = torch.empty(B * (Cout // fwd_stat) * S * 2, device=x.device, dtype=torch.float64)
            a.stat_part = ptr(part)
            fused = (part, S)
        check(lib().babe_conv2d_wino85(C.byref(a), ptr(pc.bwd_wino85 if transpose else pc.fwd_wino85), stream()), "conv2d_wino85")
        return fused if (vjp_stat is not None or fwd_stat is not None) else out
    elif getattr(pc, "bwd_wino45" if transpose else "fwd_wino45", None) is not None and (lib().babe_conv2d_wino45_supported(C.byref(a)) if force_nested
                                                          else lib().babe_conv2d_wino45_preferred(C.byref(a))):
        check(lib().babe_conv2d_wino45(C.byref(a), ptr(pc.bwd_wino45 if transpose else pc.fwd_wino45), stream()), "conv2d_wino45")
    elif getattr(pc, "fwd_wino4", None) is not None and lib().babe_conv2d_wino4_supported(C.byref(a)):
        check(lib().babe_conv2d_wino4(C.byref(a), ptr(pc.bwd_wino4 if transpose else pc.fwd_wino4), stream()), "conv2d_wino4")
    elif getattr(pc, "fwd_wino", None) is not None and lib().babe_conv2d_wino_supported(C.byref(a)):
        check(lib().babe_conv2d_wino(C.byref(a), ptr(pc.bwd_wino if transpose else pc.fwd_wino), stream()), "conv2d_wino")
    else:
        check(lib().babe_conv2d_nt(C.byref(a), pc.nt, stream()), "conv2d")
    return None if (vjp_stat is not None or fwd_stat is not None) else out


def _splits(n, B, G):
    s = max(1, min(64, n // 16384))
    return int(s)


_GN_TICKETS = {}
# Measured (round 4, same box, A/B/A/B): the one-launch form is SLOWER on the whole job - 1.992 / 1.991 vs 2.080 / 2.079 audio-sec/s -
# every workgroup of a 10 us streaming kernel pays a device-scope fence + an atomic before it retires, and the last one a
# serial tail; the separate 3.5 us finalize launch overlaps the other lane's kernels instead.  Off by default (BABE_GN_FUSED=1).
GN_FUSED = os.environ.get("BABE_GN_FUSED", "0") == "1"


def _gn_ticket(dev, n):
    """Zero-initialised ticket buffer of the fused statistics kernel, one per (device, current stream): calls on different
    streams may run concurrently and must not share tickets; every call leaves its tickets at zero."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    t = _GN_TICKETS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 1024), device=dev, dtype=torch.int32)
        _GN_TICKETS[key] = t
    return t


def gn_scale(x, gamma, film, G=8, eps=1e-7):
    """Returns (stats [B,G,3], scale [B,C]) with scale = gamma*(film+1)/(std+eps).  x dense [B,C,F,T].
    Two launches (partial sums, finalize); BABE_GN_FUSED=1 = one launch (csrc/norm.hip gn_partial_kernel<true>: the last
    workgroup of a group finalises it; bit-identical, measured slower on the whole job - see GN_FUSED above)."""
    assert x.is_contiguous()
    B, Cc, F, T = x.shape
    n = (Cc // G) * F * T
    S = _splits(n, B, G)
    part = torch.empty(B * G * S * 2, device=x.device, dtype=torch.float64)
    stats = torch.empty(B, G, 3, device=x.device, dtype=torch.float32)
    scale = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
    L = lib()
    assert film.stride(1) == 1
    if GN_FUSED:
        check(L.babe_gn_stats(ptr(x), ptr(part), ptr(_gn_ticket(x.device, B * G)), ptr(gamma), ptr(film), film.stride(0),
                              ptr(stats), ptr(scale), B, Cc, G, n, S, eps, stream()), "gn_stats")
        return stats, scale
    check(L.babe_gn_partial(ptr(x), ptr(part), B, G, n, S, stream()), "gn_partial")
    check(L.babe_gn_finalize(ptr(part), ptr(gamma), ptr(film), film.stride(0), ptr(stats), ptr(scale), B, Cc, G, n, S,
                             eps, stream()), "gn_finalize")
    return stats, scale


GELU_FIN = os.environ.get("BABE_GELU_FIN", "1") != "0"


def gn_scale_gelu(x, gamma, film, out, G=8, eps=1e-7, fused=None):
    """gn_scale + scale_gelu with the finalize folded into the GELU kernel's prologue: out = gelu(x * scale); returns
    (stats [B,G,3], scale [B,C]) for the VJP.  Bit-identical to gn_scale followed by scale_gelu (BABE_GELU_FIN=0).
    fused=(part, S): the partial sums of x already formed by the conv that wrote it (conv2d(fwd_stat=)): no pass over x for them."""
    if fused is not None:
        assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and film.stride(1) == 1
        B, Cc, F, T = x.shape
        part, S = fused
        stats = torch.empty(B, G, 3, device=x.device, dtype=torch.float32)
        scale = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
        check(lib().babe_scale_gelu_fin(ptr(x), ptr(part), ptr(gamma), ptr(film), film.stride(0), ptr(stats), ptr(scale), ptr(out),
                                        B, Cc, G, F * T, S, eps, stream()), "scale_gelu_fin")
        return stats, scale
    if not GELU_FIN:
        stats, scale = gn_scale(x, gamma, film, G, eps)
        scale_gelu(x, scale, out)
        return stats, scale
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and film.stride(1) == 1
    B, Cc, F, T = x.shape
    n = (Cc // G) * F * T
    S = _splits(n, B, G)
    part = torch.empty(B * G * S * 2, device=x.device, dtype=torch.float64)
    stats = torch.empty(B, G, 3, device=x.device, dtype=torch.float32)
    scale = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
    L = lib()
    check(L.babe_gn_partial(ptr(x), ptr(part), B, G, n, S, stream()), "gn_partial")
    check(L.babe_scale_gelu_fin(ptr(x), ptr(part), ptr(gamma), ptr(film), film.stride(0), ptr(stats), ptr(scale), ptr(out),
                                B, Cc, G, F * T, S, eps, stream()), "scale_gelu_fin")
    return stats, scale


def scale_gelu(x, scale, out):
    B, Cc, F, T = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
    check(lib().babe_scale_gelu(ptr(x), ptr(scale), ptr(out), B, Cc, F * T, stream()), "scale_gelu")
    return out


UNITS = os.environ.get("BABE_CONV_BF16U", "1") != "0"


def units_ok(pc, Cin, Cout, T):
    """True if the forward of this (5,3) conv can take its input as bf16 units (csrc/conv_bf16p.hip, UNITS variant)."""
    return UNITS and pc.splits == 1 and pc.KH == 5 and pc.KW == 3 and T % 4 == 0 and Cin % 8 == 0 and Cout > 32


def scale_gelu_units(x, scale, au):
    """GroupNorm-scale * GELU of x [B,C,F,T], written as bf16 units into the int16 buffer `au` (>= B*units_size*8)."""
    B, Cc, F, T = x.shape
    assert x.is_contiguous() and au.dtype == torch.int16 and au.is_contiguous()
    assert au.numel() >= B * lib().babe_units_size(Cc, F, T) * 8
    check(lib().babe_scale_gelu_units(ptr(x), ptr(scale), ptr(au), B, Cc, F, T, stream()), "scale_gelu_units")
    return au


def units_args(au, pc, out, Cin, dil=1, res=None, oscale=None, alpha=1.0, rbeta=1.0):
    """ConvArgs of out = alpha * conv(units, w) * oscale + rbeta * res with the input given as bf16 units."""
    B, Cout, F, T = out.shape
    a = ConvArgs()
    nu = lib().babe_units_size(Cin, F, T)
    assert au.dtype == torch.int16 and au.numel() >= B * nu * 8, "units buffer too small for this (Cin, F, T)"
    a.in_, a.in_bs, a.in_cs = ptr(au), nu, nu // (Cin // 8)
    a.in2, a.in2_bs, a.in2_cs, a.cin_split = None, 0, 0, Cin
    a.w_packed = None
    a.out, a.out_bs, a.out_cs = _view(out)
    if res is not None:
        assert res.shape == out.shape
        a.res, a.res_bs, a.res_cs = _view(res)
    else:
        a.res, a.res_bs, a.res_cs = None, 0, 0
    if oscale is not None:
        assert oscale.is_contiguous() and oscale.shape == (B, Cout)
    a.in_scale, a.oscale = None, ptr(oscale)
    a.alpha, a.rbeta = alpha, rbeta
    a.B, a.Cin, a.Cout, a.F, a.T = B, Cin, Cout, F, T
    a.KH, a.KW, a.dil = pc.KH, pc.KW, dil
    return a


def units_supported(a):
    """The library's own verdict (alignment, 2 GiB descriptor limits, BABE_CONV_BF16U): ask BEFORE writing units."""
    return bool(lib().babe_conv2d_bf16_units_supported(C.byref(a)))


def conv2d_units(au, pc, out, Cin, dil=1, res=None, oscale=None, alpha=1.0, rbeta=1.0, args=None):
    a = args if args is not None else units_args(au, pc, out, Cin, dil, res, oscale, alpha, rbeta)
    check(lib().babe_conv2d_bf16_units(C.byref(a), ptr(pc.fwd), stream()), "conv2d_bf16_units")
    return out


def gn_bwd(x, da, gy, scale, stats, gx, rbeta, G=8, eps=1e-7, merge=None, fused=None):
    """gx = rbeta*gy + GN/FiLM/GELU input-VJP of da (gx may alias gy; da is only read).
    merge=(acc, ca, cb): gx = ca*acc + cb*(that result) in the same pass (a block's VJP tail, babe_gn_bwd_apply_merge).
    fused=(part, S): the partial sums already formed by the conv that wrote da (conv2d(..., vjp_stat=)): no babe_gn_bwd_partial."""
    B, Cc, F, T = x.shape
    assert x.is_contiguous() and da.is_contiguous() and gx.is_contiguous() and (gy is None or gy.is_contiguous())
    n = (Cc // G) * F * T
    L = lib()
    if fused is not None:
        part, S = fused
    else:
        S = _splits(n, B, G)
        part = torch.empty(B * G * S, device=x.device, dtype=torch.float64)
        check(L.babe_gn_bwd_partial(ptr(x), ptr(da), ptr(scale), ptr(part), B, Cc, G, F * T, S, stream()), "gn_bwd_partial")
    if merge is not None:
        acc, ca, cb = merge
        assert acc.is_contiguous() and acc.shape == x.shape
        check(L.babe_gn_bwd_apply_merge(ptr(x), ptr(da), ptr(gy), ptr(scale), ptr(stats), ptr(part), ptr(gx), rbeta, B, Cc, G,
                                        F * T, S, eps, stream(), ptr(acc), ca, cb), "gn_bwd_apply_merge")
        return gx
    check(L.babe_gn_bwd_apply(ptr(x), ptr(da), ptr(gy), ptr(scale), ptr(stats), ptr(part), ptr(gx), rbeta, B, Cc, G,
                              F * T, S, eps, stream()), "gn_bwd_apply")
    return gx


AXPBY2 = os.environ.get("BABE_AXPBY2", "1") != "0"          # 0: the two-pass forms of the merged element-wise passes (A/B switch)


def resample(x, out, mode, alpha=1.0, beta=0.0, res=None):
    """mode 0 down, 1 up, 2 down^T, 3 up^T. T argument is the forward op's input length.
    out = alpha*R(x) + beta*out, or with res: out = alpha*R(x) + beta*res (one pass; unaligned views: copy, then accumulate)."""
    B, Cc, F, Tin = x.shape
    T = {0: Tin, 1: Tin, 2: Tin * 2, 3: Tin // 2}[mode]
    Tout = {0: T // 2, 1: 2 * T, 2: T, 3: T}[mode]
    assert out.shape == (B, Cc, F, Tout), (out.shape, (B, Cc, F, Tout))
    xp, xbs, xcs = _view(x)
    op, obs, ocs = _view(out)
    if res is not None:
        assert res.shape == out.shape
        rp, rbs, rcs = _view(res)
        if AXPBY2 and all(v % 4 == 0 for v in (rbs, rcs, obs, ocs)) and res.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0:
            check(lib().babe_resample_res(xp, xbs, xcs, rp, rbs, rcs, op, obs, ocs, B, Cc, F, T, mode, alpha, beta, stream()),
                  "resample_res")
            return out
        axpby(res, out)
    check(lib().babe_resample(xp, xbs, xcs, op, obs, ocs, B, Cc, F, T, mode, alpha, beta, stream()), "resample")
    return out


def axpby(x, out, alpha=1.0, beta=0.0):
    assert x.shape == out.shape
    B, Cc, F, T = x.shape
    xp, xbs, xcs = _view(x)
    op, obs, ocs = _view(out)
    check(lib().babe_axpby4d(xp, xbs, xcs, op, obs, ocs, B, Cc, F, T, alpha, beta, stream()), "axpby4d")
    return out


def axpby2(x, y, out, alpha, beta):
    """out = alpha*x + beta*y in one pass over [B,C,F,T] views (falls back to two axpby calls for unaligned views)."""
    assert x.shape == out.shape and y.shape == out.shape
    B, Cc, F, T = x.shape
    xp, xbs, xcs = _view(x)
    yp, ybs, ycs = _view(y)
    op, obs, ocs = _view(out)
    aligned = (F * T) % 4 == 0 and all(v % 4 == 0 for v in (xbs, xcs, ybs, ycs, obs, ocs)) and \
        all(t.data_ptr() % 16 == 0 for t in (x, y, out))
    if not (aligned and AXPBY2):
        axpby(x, out, alpha=alpha)
        return axpby(y, out, alpha=beta, beta=1.0)
    check(lib().babe_axpby2_4d(xp, xbs, xcs, yp, ybs, ycs, op, obs, ocs, B, Cc, F, T, alpha, beta, stream()), "axpby2_4d")
    return out


def linear(x, W, bias, relu=False, out=None):
    B, K = x.shape
    J = W.shape[0]
    assert x.is_contiguous() and W.is_contiguous()
    if out is None:
        out = torch.empty(B, J, device=x.device, dtype=torch.float32)
    check(lib().babe_linear(ptr(x), ptr(W), ptr(bias), ptr(out), B, K, J, int(relu), stream()), "linear")
    return out


def rff(cnoise, freq):
    B = cnoise.shape[0]
    R = freq.numel()
    out = torch.empty(B, 2 * R, device=cnoise.device, dtype=torch.float32)
    check(lib().babe_rff(ptr(cnoise.contiguous()), ptr(freq.contiguous()), ptr(out), B, R, stream()), "rff")
    return out
