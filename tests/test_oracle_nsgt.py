"""Invariants that pin the oracle CQT (parity with cqt_nsgt_pytorch is UNPINNED, see oracle/nsgt.py)."""
import pytest
import torch

from oracle.nsgt import CQT_nsgt


@pytest.fixture(scope="module")
def cqt64():
    return CQT_nsgt(7, 64, "oct", ("kaiser", 1), 22050, 92092, dtype=torch.float64)


def test_shapes(cqt64):
    x = torch.randn(2, 1, 92092, dtype=torch.float64)
    C = cqt64.fwd(x)
    assert [tuple(c.shape) for c in C] == [(2, 1, 64, 16 * 2 ** j) for j in range(7)]
    c44 = CQT_nsgt(7, 64, "oct", ("kaiser", 1), 44100, 368368).design
    assert c44["T_oct"] == [64, 128, 256, 512, 1024, 2048, 4096]
    assert abs(c44["f"][0] - (22050 - 1e-6) / 128) < 1e-9


def test_perfect_reconstruction_white_noise(cqt64):
    torch.manual_seed(0)
    x = torch.randn(2, 1, 92092, dtype=torch.float64)
    xr = cqt64.bwd(cqt64.fwd(x))
    xh = cqt64.apply_hpf_DC(x)
    assert float((xr - xh).norm() / xh.norm()) < 1e-12
    # float32 instance
    c32 = CQT_nsgt(7, 64, "oct", ("kaiser", 1), 22050, 92092)
    x32 = x.float()
    assert float((c32.bwd(c32.fwd(x32)) - c32.apply_hpf_DC(x32)).norm() / x32.norm()) < 1e-5


def test_hpf_zero_phase_unit_passband(cqt64):
    H = cqt64.Hhpf_full
    L = cqt64.Ls
    assert torch.allclose(H[1:], H[1:].flip(0))                      # real symmetric -> zero phase
    d = cqt64.design
    lo = int(d["Om"][0] * 2 ** (1 / 128)) + d["M"][0]
    hi = int(L / 2 / 2 ** (1 / 128)) - 8
    assert float((H[d["M_dc"] // 2 + 2: hi] - 1).abs().max()) < 1e-12
    assert float(H[0]) < 1e-12


def test_adjoint(cqt64):
    torch.manual_seed(1)
    x = torch.randn(1, 1, 92092, dtype=torch.float64, requires_grad=True)
    C = cqt64.fwd(x)
    G = [torch.randn_like(c) for c in C]
    s = sum((c.real * g.real + c.imag * g.imag).sum() for c, g in zip(C, G))
    gx, = torch.autograd.grad(s, x)
    # <fwd x, G> == <x, fwd^T G> with a second random probe
    x2 = torch.randn_like(x)
    C2 = cqt64.fwd(x2)
    lhs = sum((c.real * g.real + c.imag * g.imag).sum() for c, g in zip(C2, G))
    assert abs(float(lhs) - float((x2 * gx).sum())) < 1e-8 * abs(float(lhs)) + 1e-8


# ---- the pin that becomes active the day the library is present (tests/golden/make_cqt_golden.py) ----------------------------
import os                                                                    # noqa: E402

import numpy as np                                                           # noqa: E402

CQT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cqt_lib.npz")


def lib_probe(L, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.1 * torch.randn(2, 1, L, generator=g)


@pytest.mark.skipif(not os.path.exists(CQT_LIB), reason="tests/golden/cqt_lib.npz absent: cqt_nsgt_pytorch is not available in "
                    "the build container (run tests/golden/make_cqt_golden.py where it is); the CQT is 'parity unpinned'")
@pytest.mark.parametrize("fs,L", [(22050, 92092), (44100, 368368)])
def test_oracle_vs_cqt_nsgt_pytorch_library(fs, L):
    """oracle/nsgt.py against outputs of the reference's real dependency.  If this fails, the first suspect is the synthesis
    convention (oracle/nsgt.py bwd: conjugate-mirrored bands added before the inverse FFT; the library is believed to irfft
    the positive half only, SURVEY App. B 'pitfall'), then the relocated top bin / Nyquist band."""
    z = np.load(CQT_LIB)
    tag = f"{fs}_{L}"
    sub = 8
    cqt = CQT_nsgt(7, 64, "oct", ("kaiser", 1), fs, L)
    x = lib_probe(L, int(z[f"{tag}.seed"]))
    X = cqt.fwd(x)
    assert [int(c.shape[-1]) for c in X] == [int(v) for v in z[f"{tag}.T_oct"]]
    for j, c in enumerate(X):
        got = torch.view_as_real(c.squeeze(1))[..., ::max(1, sub // 2), :]
        ref = torch.from_numpy(z[f"{tag}.fwd{j}"])
        assert float((got - ref).norm() / ref.norm()) < 1e-4, f"fwd octave {j}"
    for name, y in (("bwd", cqt.bwd(X)), ("hpf", cqt.apply_hpf_DC(x.squeeze(1)))):
        ref = torch.from_numpy(z[f"{tag}.{name}"])
        got = y.reshape(2, -1)[:, ::sub]
        assert float((got - ref).norm() / ref.norm()) < 1e-4, name
