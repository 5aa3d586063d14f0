"""ORACLE (test infrastructure) — STFT-domain degradation model.

Follows /root/reference/utils/blind_bwe_utils.py: apply_stft :15-26,
apply_filter_istft :28-39, apply_filter :6-13, design_filter :82-119,
apply_filter_and_norm_STFTmag_fweighted :250-296 ("sqrt"/"linear"/"None"
weightings), and BlindSampler.fit_params
/root/reference/testing/blind_bwe_sampler.py:533-595.
Pinned by tests/golden (G2-G5).
"""
import math

import torch


def stft(x, nfft):
    """[B,L] -> complex [B, nfft/2+1, frames]; NFFT zeros appended, hop nfft/2, periodic hamming, no centering."""
    w = torch.hamming_window(nfft, dtype=x.dtype, device=x.device)
    xp = torch.cat((x, torch.zeros(x.shape[0], nfft, dtype=x.dtype, device=x.device)), dim=1)
    fr = xp.unfold(-1, nfft, nfft // 2)                      # [B, frames, nfft]
    return torch.fft.rfft(fr * w, dim=-1).transpose(1, 2)


def istft(X, nfft):
    """inverse of stft() with torch.istft(center=False) semantics: window-squared envelope normalisation."""
    w = torch.hamming_window(nfft, dtype=X.real.dtype, device=X.device)
    fr = torch.fft.irfft(X.transpose(1, 2), n=nfft, dim=-1) * w    # [B, frames, nfft]
    B, nfr, _ = fr.shape
    hop = nfft // 2
    n = nfft + hop * (nfr - 1)
    out = torch.zeros(B, n, dtype=fr.dtype, device=fr.device)
    env = torch.zeros(n, dtype=fr.dtype, device=fr.device)
    for t in range(nfr):
        out[:, t * hop: t * hop + nfft] = out[:, t * hop: t * hop + nfft] + fr[:, t]
        env[t * hop: t * hop + nfft] += w * w
    return out / env


def apply_filter(x, H, nfft):
    X = stft(x, nfft)
    return istft(X * H[None, :, None], nfft)[:, : x.shape[-1]]


def bin_freqs(nfft, fs):
    """float32 k*fs/nfft, identical to torch.fft.rfftfreq(nfft, d=1/fs)."""
    return torch.fft.rfftfreq(nfft, d=1 / fs)


def design_filter(fc, A, f):
    """Piecewise log-linear low-pass; closed form of the reference's in-place loop."""
    fc = torch.atleast_1d(fc)
    A = torch.atleast_1d(A)
    K = fc.shape[0]
    LOG = math.log(10.0) / 20.0
    H = torch.ones_like(f)
    m0 = f >= fc[0]
    # masked-out bins use a dummy ratio of 1 so that no inf/nan enters autograd
    seg = 10 ** (A[0] * torch.log2(torch.where(m0, f, fc[0].detach()) / fc[0]) / 20)
    H = torch.where(m0, seg, H)
    for i in range(1, K):
        mi = f >= fc[i]
        kstar = int(torch.nonzero(mi)[0, 0])
        anchor = H[kstar]
        seg = 10 ** (A[i] * torch.log2(torch.where(mi, f, fc[i].detach()) / fc[i]) / 20) * anchor
        H = torch.where(mi, seg, H)
    return H


def freq_weight(nbins, kind, dtype=torch.float32):
    fr = torch.linspace(0, 1, nbins, dtype=dtype)
    if kind == "sqrt":
        return torch.sqrt(fr)
    if kind == "linear":
        return fr
    if kind == "None":
        return torch.ones_like(fr)
    if kind == "log":
        return torch.log2(1 + fr)
    if kind == "quadratic":
        return fr ** 2
    if kind == "cubic":
        return fr ** 3
    if kind == "squared":
        return fr ** 4
    if kind == "logquadratic":
        return torch.log2(1 + fr ** 2)
    if kind == "logcubic":
        return torch.log2(1 + fr ** 3)
    raise NotImplementedError(kind)


def stft_distance(y, rec, nfft, weight="None", mag=False, logmag=False):
    """apply_norm_STFT_fweighted :148-196 (mag=False) / apply_norm_STFTmag_fweighted :198-247: ONE scalar over the whole
    batch, ||w (S(rec) - S(y))||_2, ||w|S(rec)| - w|S(y)|||_2 or the same on log10(w|S| + 1e-8)."""
    X, Xr = torch.view_as_real(stft(rec, nfft)), torch.view_as_real(stft(y, nfft))        # [B, bins, frames, 2]
    w = freq_weight(X.shape[1], weight, X.dtype).to(X.device)
    if not mag:
        return torch.linalg.norm((X * w[None, :, None, None] - Xr * w[None, :, None, None]).reshape(-1), ord=2)
    Xm = torch.sqrt(X[..., 0] ** 2 + X[..., 1] ** 2) * w[None, :, None]
    Xrm = torch.sqrt(Xr[..., 0] ** 2 + Xr[..., 1] ** 2) * w[None, :, None]
    if logmag:
        return torch.linalg.norm(torch.log10(Xm.reshape(-1) + 1e-8) - torch.log10(Xrm.reshape(-1) + 1e-8), ord=2)
    return torch.linalg.norm(Xm.reshape(-1) - Xrm.reshape(-1), ord=2)


def mag_loss(Xmag, Ymag, H, w):
    """|| w_f (H_f |X| - |Y|) ||_2 over all b,f,t  (:250-296)."""
    d = (Xmag * H[None, :, None] - Ymag) * w[None, :, None]
    return torch.linalg.norm(d.reshape(-1), ord=2)


def fit_params(x_den, y, params, fs, nfft=4096, mu=(1000.0, 10.0), tol=(5e-3, 5e-3), max_iter=100,
               fcmin=20.0, fcmax=None, Amin=-50.0, weighting="sqrt", trajectory=None):
    """Projected gradient descent on (fc, A), blind_bwe_sampler.py:533-595 (clamp_fc, clamp_A, only_negative_A)."""
    fcmax = fs // 2 if fcmax is None else fcmax
    f = bin_freqs(nfft, fs)
    Xm = stft(x_den, nfft).abs()
    Ym = stft(y, nfft).abs()
    w = freq_weight(Xm.shape[1], weighting)
    mu_t = torch.tensor(mu, dtype=torch.float32)
    p = params.clone().float()
    prev = None
    n_it = 0
    for it in range(max_iter):
        p = p.detach().requires_grad_(True)
        loss = mag_loss(Xm, Ym, design_filter(p[0], p[1], f), w)
        g, = torch.autograd.grad(loss, p)
        p = (p - mu_t[:, None] * g).detach()
        K = p.shape[1]
        p[0, 0] = torch.clamp(p[0, 0], min=fcmin, max=fcmax)
        for k in range(1, K):
            p[0, k] = torch.clamp(p[0, k], min=p[0, k - 1] + 1, max=fcmax)
        p[1, 0] = torch.clamp(p[1, 0], min=Amin, max=-1)
        for k in range(1, K):
            p[1, k] = torch.clamp(p[1, k], min=Amin, max=p[1, k - 1])
        n_it = it + 1
        if trajectory is not None:
            trajectory.append(p.clone())
        if it > 0 and (p[0] - prev[0]).abs().mean() < tol[0] and (p[1] - prev[1]).abs().mean() < tol[1]:
            break
        prev = p.clone()
    return p, n_it


def compute_sweep(x_den, y, fs, nfft=4096, weighting="sqrt", fc_s=None, A_s=None):
    """BlindSampler.compute_sweep (testing/blind_bwe_sampler.py:598-616): the fit objective and its autograd gradient w.r.t.
    (fc, A) for every point of the grid fc_s x A_s (one break point).  Returns (norms [nf, na], grads [nf, na, 2])."""
    fc_s = torch.logspace(2.5, 4, 15) if fc_s is None else fc_s
    A_s = torch.linspace(-80, -5, 12) if A_s is None else A_s
    f = bin_freqs(nfft, fs)
    Xm, Ym = stft(x_den, nfft).abs(), stft(y, nfft).abs()
    w = freq_weight(Xm.shape[1], weighting)
    norms = torch.zeros(len(fc_s), len(A_s))
    grads = torch.zeros(len(fc_s), len(A_s), 2)
    for i, fc in enumerate(fc_s):
        for j, A in enumerate(A_s):
            p = torch.tensor([float(fc), float(A)]).requires_grad_(True)
            loss = mag_loss(Xm, Ym, design_filter(p[0:1], p[1:2], f), w)
            g, = torch.autograd.grad(loss, p)
            norms[i, j], grads[i, j] = loss.detach(), g
    return norms, grads

