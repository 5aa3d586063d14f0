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
