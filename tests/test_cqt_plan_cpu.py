"""The library-side band design of the CQT (csrc/cqt_plan.hip: babe_cqt_design_create, host only - no GPU call) against the numpy
design the Python class uses (babe_amd/cqt.py::design_bands, factor_len, small_radices, kaiser_poly): integer tables exactly,
the float64 tables to 4 ulp (numpy's array power and exp are its own SIMD routines, libm's differ from them by an ulp on some
arguments, and a window is a quotient of two I0 values; 16 ulp on the dual window, which divides by a sum
of ~10^5 squared window samples; the high-pass response 1 - lp / diag to 1e-15 absolute), and the float32 images that reach the device equal
except where a float64 difference of that size straddles a float32 rounding boundary (< 1e-4 of the entries, 1 float32 ulp)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(44100, 368368, 7, 64, 1.0), (22050, 92092, 7, 64, 1.0), (16000, 184184, 7, 64, 1.0), (44100, 46046, 7, 64, 1.0),
         (22050, 65536, 6, 32, 2.0)]


def _lib():
    L = C.CDLL(os.path.join(ROOT, "babe_amd", "libbabe_hip.so"))
    L.babe_cqt_design_create.restype = C.c_void_p
    L.babe_cqt_design_create.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]
    L.babe_cqt_design_destroy.argtypes = [C.c_void_p]
    L.babe_cqt_design_get.restype = C.c_long
    L.babe_cqt_design_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
    L.babe_last_error.restype = C.c_char_p
    return L


def _get(L, d, name, dtype):
    n = L.babe_cqt_design_get(d, name.encode(), None, 0)
    assert n >= 0, name
    out = np.empty(n // np.dtype(dtype).itemsize, dtype=dtype)
    assert L.babe_cqt_design_get(d, name.encode(), out.ctypes.data_as(C.c_void_p), n) == n
    return out


def _ulps(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a.view(np.int64) - b.view(np.int64)).max() if a.size else 0


@pytest.mark.parametrize("fs,Ls,numocts,binsoct,beta", CASES)
def test_library_design_equals_numpy_design(fs, Ls, numocts, binsoct, beta):
    from babe_amd.cqt import design_bands, factor_len, kaiser_poly, small_radices
    L = _lib()
    d = L.babe_cqt_design_create(float(fs), Ls, numocts, binsoct, float(beta))
    assert d, L.babe_last_error()
    try:
        ref = design_bands(fs, Ls, numocts, binsoct, beta)
        for k in ("M", "c", "T", "woff", "rowptr"):
            assert np.array_equal(_get(L, d, k, np.int64), np.asarray(ref[k], dtype=np.int64)), k
        assert np.array_equal(_get(L, d, "src", np.int64), np.asarray(ref["src"], dtype=np.int64))
        assert int(_get(L, d, "nwin", np.int64)[0]) == ref["nwin"] and int(_get(L, d, "M_dc", np.int64)[0]) == ref["M_dc"]
        worst = {}
        for k in ("f", "Om", "g", "gdual", "Tw", "hpf"):
            got = _get(L, d, k, np.float64)
            assert got.shape == np.asarray(ref[k]).shape, k
            if k == "hpf":                                   # 1 - lp / diag: values near 0 make ulps meaningless; absolute
                worst[k] = float(np.abs(got - ref[k]).max())
                assert worst[k] < 1e-15, worst[k]
                continue
            worst[k] = int(_ulps(got, ref[k]))
            assert worst[k] <= {"Tw": 0, "gdual": 16}.get(k, 4), (k, worst[k])
        # the float32 images the device gets (cqt.py: tf(g / Tw), tf(gd * Tw), tf(gd * Tw * (2.0 / L)))
        g, gd, Tw = (_get(L, d, k, np.float64) for k in ("g", "gdual", "Tw"))
        for got, want in ((g / Tw, ref["g"] / ref["Tw"]), (gd * Tw, ref["gdual"] * ref["Tw"]),
                          (gd * Tw * (2.0 / Ls), ref["gdual"] * ref["Tw"] * (2.0 / Ls))):
            a, b = got.astype(np.float32), want.astype(np.float32)
            diff = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
            assert diff.max() <= 1 and (diff != 0).mean() < 1e-4
        # FFT plan, workgroup table, analytic window
        N1, N2 = factor_len(Ls)
        assert (int(_get(L, d, "N1", np.int64)[0]), int(_get(L, d, "N2", np.int64)[0])) == (N1, N2)
        assert list(_get(L, d, "rad1", np.int32)) == small_radices(N1) and list(_get(L, d, "rad2", np.int32)) == small_radices(N2)
        assert int(_get(L, d, "KX", np.int64)[0]) == ((Ls // 2) // N1 + 1) * N1
        T_oct = [int(ref["T"][j * binsoct]) for j in range(numocts)]
        assert list(_get(L, d, "T_oct", np.int32)) == T_oct
        wgf, wgc = [], []
        for j in range(numocts):
            bpw = int(min(binsoct, max(1, 4096 // T_oct[j])))
            for s0 in range(0, binsoct, bpw):
                wgf.append(j * binsoct + s0)
                wgc.append(min(bpw, binsoct - s0))
        assert list(_get(L, d, "wg_first", np.int32)) == wgf and list(_get(L, d, "wg_count", np.int32)) == wgc
        deg, co = kaiser_poly(beta)
        assert int(_get(L, d, "kdeg", np.int64)[0]) == deg and _ulps(_get(L, d, "kpoly", np.float64), co) <= 2
        print(f"fs={fs} L={Ls}: float64 tables within {worst} ulp of the numpy design")
    finally:
        L.babe_cqt_design_destroy(d)


def test_design_refuses_what_it_cannot_build():
    L = _lib()
    assert not L.babe_cqt_design_create(44100.0, 368369, 7, 64, 1.0)              # odd length
    assert b"even audio length" in L.babe_last_error()
    assert not L.babe_cqt_design_create(44100.0, 2 * 100003, 7, 64, 1.0)           # 2 x prime: no balanced factorisation
    assert not L.babe_cqt_design_create(44100.0, 368368, 9, 64, 1.0)               # more octaves than coef[8] holds
    assert L.babe_cqt_design_get(None, b"M", None, 0) == -1
