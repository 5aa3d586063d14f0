"""The C-ABI library loads without a GPU and exports every symbol include/babe_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "babe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(babe_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 10
    L = ctypes.CDLL(os.path.join(ROOT, "babe_amd", "libbabe_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    L.babe_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.babe_version()


def test_unet_plan_structs_have_the_headers_layout(tmp_path):
    """babe_amd/networks/unet_c.py mirrors babe_packed_conv / babe_unet_block / babe_unet_plan_desc with ctypes: sizes and a few
    field offsets must be what a C compiler gives the header (a silent mismatch would hand the library garbage pointers)."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from babe_amd.networks import unet_c as uc
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   ' sizeof(babe_packed_conv), sizeof(babe_unet_block), sizeof(babe_unet_plan_desc), offsetof(babe_packed_conv, w_raw),'
                   ' offsetof(babe_packed_conv, bwd_wino85),'
                   ' offsetof(babe_unet_block, gamma), offsetof(babe_unet_block, film_gate), offsetof(babe_unet_plan_desc, pyr_conv));return 0;}\n'
                   % os.path.join(ROOT, "include", "babe_hip.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    want = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    got = [ctypes.sizeof(uc.CPackedConv), ctypes.sizeof(uc.CBlock), ctypes.sizeof(uc.CPlanDesc), uc.CPackedConv.w_raw.offset,
           uc.CPackedConv.bwd_wino85.offset, uc.CBlock.gamma.offset, uc.CBlock.film_gate.offset, uc.CPlanDesc.pyr_conv.offset]
    assert got == want, (got, want)


def test_cqt_bands_struct_has_the_headers_layout(tmp_path):
    """babe_amd/cqt.py mirrors babe_cqt_bands with ctypes (passed BY VALUE to the band kernels): size and the offsets of the
    fields round 6 appended (analytic Kaiser window) must be the header's."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from babe_amd.cqt import _BandsStruct
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   ' sizeof(babe_cqt_bands), offsetof(babe_cqt_bands, coef), offsetof(babe_cqt_bands, wg_rec), offsetof(babe_cqt_bands, sum_T),'
                   ' offsetof(babe_cqt_bands, kdeg), offsetof(babe_cqt_bands, kpoly));return 0;}\n'
                   % os.path.join(ROOT, "include", "babe_hip.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    want = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    got = [ctypes.sizeof(_BandsStruct), _BandsStruct.coef.offset, _BandsStruct.wg_rec.offset, _BandsStruct.sum_T.offset,
           _BandsStruct.kdeg.offset, _BandsStruct.kpoly.offset]
    assert got == want, (got, want)


def test_kaiser_poly_reproduces_the_window_table():
    """The truncated I0 series the band kernels evaluate (babe_cqt_bands::kpoly) against the float64 Kaiser window the tables
    are built from: below fp32 resolution for the betas it accepts; large betas fall back to the table."""
    import numpy as np
    from babe_amd.cqt import _kaiser, kaiser_poly
    for beta in (0.5, 1.0, 2.0, 3.0):
        deg, co = kaiser_poly(beta)
        assert 1 <= deg <= 11 and (co[deg + 1:] == 0).all()
        for M in (4, 7, 64, 1001, 4096):
            m = np.arange(-(M // 2), M - M // 2)
            a = np.maximum(1 - (2.0 * m / M) ** 2, 0)
            p = sum(co[j] * a ** j for j in range(12))
            assert np.abs(p - _kaiser(M, beta)).max() < 3e-9
    assert kaiser_poly(8.6) is None


def test_unet_plan_create_validates_every_block():
    """babe_unet_plan_create takes descriptors from any C-ABI host: a dilation-layer count outside 0..8 (it indexes saved[..][8],
    H[8], gamma[8]), a channel count that is not a positive multiple of the 8 GroupNorm groups, or a missing gamma / packed image
    in ANY block - init, main, up, out, middle - is refused with a message instead of overrunning.  Host-side
    validation only: no GPU call is made."""
    from babe_amd._lib import lib
    from babe_amd.networks import unet_c as uc
    uc._register()
    L = lib()
    FAKE = 0x1000                                         # never dereferenced by plan_create (pointers stay the caller's)

    def good():
        d = uc.CPlanDesc()
        d.nocts, d.bpo = 2, 64
        for i in range(2):
            d.Ns[i] = 16
        def blk(b, nd):
            b.N, b.nd, b.k53 = 16, nd, 1
            for k in range(nd):
                b.H[k].Cout = b.H[k].Cin = 16
                b.H[k].KH, b.H[k].KW = 5, 3
                b.H[k].fwd = FAKE
                b.gamma[k] = FAKE
        for i in range(2):
            for arr in (d.init_blk, d.main_blk, d.up_out, d.up_blk):
                blk(arr[i], 2)
        blk(d.mid_blk, 3)
        blk(d.mid_out, 1)
        return d

    def create(d):
        p = L.babe_unet_plan_create(ctypes.byref(d))
        if p:
            L.babe_unet_plan_destroy(p)
        return bool(p), L.babe_last_error().decode()

    assert create(good())[0]
    for name, mutate in {
        "init_blk nd 9": lambda d: setattr(d.init_blk[1], "nd", 9),
        "up_out nd -1": lambda d: setattr(d.up_out[0], "nd", -1),
        "mid_blk nd 12": lambda d: setattr(d.mid_blk, "nd", 12),
        "mid_out nd 0": lambda d: setattr(d.mid_out, "nd", 0),
        "main_blk N 12": lambda d: setattr(d.main_blk[0], "N", 12),
        "up_blk gamma NULL": lambda d: d.up_blk[1].gamma.__setitem__(1, None),
        "main_blk H without image": lambda d: setattr(d.main_blk[1].H[0], "fwd", None),
        "Ns not a multiple of 8": lambda d: d.Ns.__setitem__(1, 20),
    }.items():
        d = good()
        mutate(d)
        ok, msg = create(d)
        assert not ok and "bad descriptor" in msg, (name, ok, msg)


def test_f45_dispatch_rule_is_host_logic():
    """babe_conv2d_wino85_supported / _preferred are pure host functions (no launch, no device pointer dereferenced): the dispatch
    rule of the F(4,5) x F(4,3) kernels - output channels a multiple of 128 / 96 / 64, Cin % 16 == 0, one source, T % 4 == 0 and
    >= 64, 16-byte aligned views; preferred = supported and row quads x time tiles >= 80 % full - checked without a GPU."""
    import ctypes as C
    from babe_amd._lib import ConvArgs, lib
    L = lib()

    def args(Cin=128, Cout=128, F=64, T=128, dil=4, in2=None, in_=0x10000, B=1):
        a = ConvArgs()
        a.in_, a.in_bs, a.in_cs = in_, Cin * F * T, F * T
        a.in2, a.in2_bs, a.in2_cs, a.cin_split = in2, 0, 0, Cin
        a.out, a.out_bs, a.out_cs = 0x20000, Cout * F * T, F * T
        a.res, a.res_bs, a.res_cs = None, 0, 0
        a.in_scale = a.oscale = None
        a.alpha, a.rbeta = 1.0, 0.0
        a.B, a.Cin, a.Cout, a.F, a.T, a.KH, a.KW, a.dil = B, Cin, Cout, F, T, 5, 3, dil
        return a

    sup = lambda **k: L.babe_conv2d_wino85_supported(C.byref(args(**k)))
    pre = lambda **k: L.babe_conv2d_wino85_preferred(C.byref(args(**k)))
    for co in (64, 96, 128, 192, 256, 320, 384):
        assert sup(Cout=co) == 1, co
    for co in (16, 32, 48, 80, 112, 144):
        assert sup(Cout=co) == 0, co
    assert sup(Cin=16) == 1 and sup(Cin=24) == 0 and sup(Cin=8) == 0
    assert sup(T=64) == 1 and sup(T=60) == 0 and sup(T=66) == 0 and sup(T=68) == 1
    assert sup(in2=0x30000) == 0 and sup(in_=0x10004) == 0
    a = args()
    a.KH = 3
    assert L.babe_conv2d_wino85_supported(C.byref(a)) == 0
    # fill = F / (4 * quads * dil) * T / (64 * tiles): 16 rows per class = 4 full quads; 5 rows -> 2 quads = 0.625; 6 -> 0.75; 7 -> 0.875
    assert pre(F=64, dil=4) == 1 and pre(F=40, dil=8) == 0 and pre(F=48, dil=8) == 0 and pre(F=56, dil=8) == 1
    assert pre(F=64, dil=4, T=68) == 0 and pre(F=64, dil=4, T=120) == 1          # 68 of 128 steps / 120 of 128
    # the benchmark's part-filled geometries: 320 rows at dilation 32 (0.83: taken), 384 rows at dilation 64 (0.75: left to F(2,5) x F(4,3))
    assert pre(F=320, T=256, dil=32) == 1 and pre(F=384, T=128, dil=64, Cin=256, Cout=256) == 0
    assert pre(F=448, T=64, dil=64, Cin=256, Cout=256) == 1 and pre(F=320, T=256, dil=16) == 1


def test_plan_entry_points_refuse_bad_arguments_without_a_gpu():
    """Host-side validation of the round-6 plan entry points (no GPU call is made on these paths): NULL handles, NULL buffers and
    an empty descriptor are refused with a message instead of being dereferenced."""
    from babe_amd._lib import lib
    from babe_amd.testing import eval_c
    eval_c._register()
    L = lib()
    L.babe_last_error.restype = ctypes.c_char_p
    L.babe_cqt_workspace_bytes.restype = ctypes.c_long
    L.babe_cqt_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.babe_cqt_workspace_bytes(None, 4) == -1
    for name in ("babe_cqt_fwd", "babe_cqt_bwd", "babe_cqt_fwd_adjoint", "babe_cqt_bwd_adjoint", "babe_cqt_hpf"):
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
        assert fn(None, None, None, None, 1, None) < 0 and b"bad arguments" in L.babe_last_error()
    d = eval_c.EvalDesc()
    assert L.babe_eval_workspace_bytes(ctypes.byref(d), 1) == -1 and b"bad descriptor" in L.babe_last_error()
    assert L.babe_score_eval(ctypes.byref(d), None, 0.1, 1.0, 1.0, 1.0, 0.0, None, None, None, None, None, None, None, 0, 1, None) < 0
    L.babe_filter_loss_grad.restype = ctypes.c_int
    L.babe_filter_loss_grad.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert L.babe_filter_loss_grad(None, 0, None, None, 1, 1, 2049, 44100.0, 4096, None, None) < 0
