"""Oracle (CPU restatement) vs golden vectors generated from the imported reference."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import bwe_utils as U
from oracle import edm as E
from oracle import unet as UN

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name)).items()}


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("name", ["formal", "brass", "train"])
def test_edm_scalars(name):
    g = load("edm.npz")
    c = {k: float(g[f"{name}_cfg_{k}"]) for k in ("sigma_data", "sigma_min", "sigma_max", "ro", "Schurn", "Stmin", "Stmax", "Snoise")}
    p = E.EDMParams(**c)
    for N in (3, 35):
        t = E.schedule(p, N)
        assert torch.equal(t, g[f"{name}_sched_{N}"])
        assert torch.equal(E.schedule(p, N, 0.2), g[f"{name}_sched0_{N}"])
        assert torch.equal(E.gamma(p, t), g[f"{name}_gamma_{N}"])
    s = g[f"{name}_sig"]
    for fn in ("cskip", "cout", "cin", "cnoise"):
        assert torch.allclose(getattr(E, fn)(p, s), g[f"{name}_{fn}"], rtol=1e-6, atol=0)


def test_edm_known_answers():
    # SURVEY §8 a3 known-answer values
    p = E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10)
    t = E.schedule(p, 35)
    assert abs(float(t[1]) - 0.849993) < 1e-5 and abs(float(t[34]) - 1e-4) < 1e-9 and float(t[35]) == 0
    assert abs(float(E.gamma(p, t)[0]) - 10 / 36) < 1e-6 and float(E.gamma(p, t)[35]) == 0


@pytest.mark.parametrize("nfft", [1024, 4096])
def test_stft_and_filter(nfft):
    g = load("stft_filter.npz")
    gen = torch.Generator().manual_seed(int(g["stft_seed"]))
    x = torch.randn(2, 20000, generator=gen) * 0.1
    X = torch.view_as_real(U.stft(x, nfft))
    assert rel(X, g[f"stft_{nfft}"]) < 2e-6
    f = U.bin_freqs(nfft, 44100)
    H = U.design_filter(torch.tensor([3000.0, 5000.0]), torch.tensor([-20.0, -40.0]), f)
    assert rel(U.apply_filter(x, H, nfft), g[f"filt_{nfft}"]) < 5e-6
    assert rel(U.apply_filter(x, torch.ones_like(H), nfft), g[f"ident_{nfft}"]) < 5e-6


@pytest.mark.parametrize("fs", [44100, 22050])
@pytest.mark.parametrize("case", ["k1", "k5", "k4_onbin", "k2_nyq"])
def test_design_filter(fs, case):
    g = load("stft_filter.npz")
    p = g[f"df_{fs}_{case}_p"].clone().requires_grad_(True)
    f = U.bin_freqs(4096, fs)
    H = U.design_filter(p[0], p[1], f)
    assert torch.equal(H.detach(), g[f"df_{fs}_{case}_H"])          # bit-exact incl. bin masks
    wv = torch.linspace(0.5, 1.5, H.shape[0])
    gr, = torch.autograd.grad((H * wv).sum(), p)
    assert rel(gr, g[f"df_{fs}_{case}_g"]) < 1e-5


def test_weighted_losses():
    g = load("stft_filter.npz")
    gen = torch.Generator().manual_seed(int(g["stft_seed"]))
    x = torch.randn(2, 20000, generator=gen) * 0.1
    Xm = U.stft(x, 4096).abs()
    Ym = U.stft(x.flip(0) * 0.7, 4096).abs()
    H = U.design_filter(torch.tensor([1000.0, 3000.0]), torch.tensor([-10.0, -30.0]), U.bin_freqs(4096, 44100))
    for w in ("sqrt", "linear", "None", "log"):
        assert abs(float(U.mag_loss(Xm, Ym, H, U.freq_weight(2049, w))) / float(g[f"loss_{w}"]) - 1) < 2e-6


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_fit_params(ci):
    g = load("fit_params.npz")
    seed, B, n = int(g[f"fit{ci}_seed"]), int(g[f"fit{ci}_B"]), int(g[f"fit{ci}_n"])
    fc_true, A_true = [float(v) for v in g[f"fit{ci}_true"]]
    gen = torch.Generator().manual_seed(seed)
    xd = torch.randn(B, n, generator=gen) * 0.1
    f = U.bin_freqs(4096, 44100)
    y = U.apply_filter(xd, U.design_filter(torch.tensor([fc_true]), torch.tensor([A_true]), f), 4096) \
        + 1e-3 * torch.randn(B, n, generator=gen)
    p0 = torch.tensor([[280.0, 285.0, 290.0, 295.0, 300.0], [-15.0, -17.0, -20.0, -25.0, -30.0]])
    for mi in (1, 2, 5):
        p, _ = U.fit_params(xd, y, p0, fs=44100, max_iter=mi)
        assert torch.allclose(p, g[f"fit{ci}_it{mi}"], rtol=2e-5, atol=2e-4), (mi, p, g[f"fit{ci}_it{mi}"])
    p, nit = U.fit_params(xd, y, p0, fs=44100)
    assert torch.allclose(p, g[f"fit{ci}_final"], rtol=1e-3, atol=5e-2), (p, g[f"fit{ci}_final"], nit)


@pytest.mark.parametrize("name,nd,after", [("b53", 3, False), ("b11", 1, False), ("bout", 1, True), ("bsame", 2, False)])
def test_resnet_block(name, nd, after):
    g = load("blocks.npz")
    sd = {k[len(name) + 4:]: v for k, v in g.items() if k.startswith(name + ".sd.")}
    x = g[f"{name}.x"].clone().requires_grad_(True)
    y = UN.resnet_block(sd, "", x, g["emb"], nd, proj_after=after)
    assert rel(y, g[f"{name}.y"]) < 2e-6
    gx, = torch.autograd.grad((y * g[f"{name}.wv"]).sum(), x)
    assert rel(gx, g[f"{name}.gx"]) < 5e-6


def test_small_ops():
    g = load("blocks.npz")
    assert rel(UN.group_norm_nomean(g["gn.x"], torch.ones(1, 16, 1, 1)), g["gn.y"]) < 1e-6
    for T in (16, 22):
        assert rel(UN.resample_down(g[f"rs.x{T}"]), g[f"rs.down{T}"]) < 1e-6
        assert rel(UN.resample_up(g[f"rs.x{T}"]), g[f"rs.up{T}"]) < 1e-6
    sd = {"embedding." + k[len("rff.sd."):]: v for k, v in g.items() if k.startswith("rff.sd.")}
    assert rel(UN.embedding(sd, g["rff.s"]), g["rff.y"]) < 1e-6


def test_compute_sweep_vs_reference_golden():
    """oracle compute_sweep (fit objective + autograd gradient on the 15 x 12 (fc, A) grid) vs the reference's own
    BlindSampler.compute_sweep (testing/blind_bwe_sampler.py:598-616; tests/golden/make_golden.py::g25)."""
    s = load("sweep_helpers.npz")
    norms, grads = U.compute_sweep(s["den"], s["y"], 22050)
    assert float((norms - s["norms"]).abs().max() / s["norms"].abs().max()) < 1e-5
    assert float((grads - s["grads"]).abs().max() / s["grads"].abs().max()) < 1e-4
