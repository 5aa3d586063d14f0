"""ORACLE (test infrastructure) — invertible constant-Q transform, mode "oct".

Restates the non-stationary Gabor transform (Holighaus, Doerfler, Velasco,
Grill 2013; Balazs et al. 2011) with the call surface the reference uses from
the un-vendored dependency ``cqt_nsgt_pytorch.CQT_nsgt``:

  construction  /root/reference/networks/cqtdiff+.py:620
  .fwd          /root/reference/networks/cqtdiff+.py:743
  .bwd          /root/reference/networks/cqtdiff+.py:841
  .apply_hpf_DC /root/reference/testing/blind_bwe_sampler.py:156

PARITY UNPINNED: the dependency (PyPI cqt-nsgt-pytorch, no version pin in the
reference) is not available in this image, so this file cannot be checked
against it.  It is pinned by invariants (tests/test_oracle_nsgt.py):
output structure 7 x [B,1,64,T_j] with T_j halving per octave (required by the
UNet, cqtdiff+.py:750,768-794,830), bwd(fwd(x)) == apply_hpf_DC(x) on white
noise, apply_hpf_DC zero-phase with unit pass-band gain, and adjointness.

Definition used here (and by the HIP kernels, babe_amd/cqt_plan.py):
  * scale: fmax = fs/2 - 1e-6, fmin = fmax/2^numocts, nb = numocts*binsoct
    log-spaced centres f_k = fmin * r^k, r = 2^(numocts/(nb-1)),
    Q = sqrt(r)/(r-1)/2  (NSGT ``LogScale``).
  * in FFT-bin units Om_k = f_k * L / fs.  Band lengths M_k =
    round(Om_{k+1}-Om_{k-1}) (interior), round(Om_k/Q) (first and last bin),
    M_DC = round(2*Om_0), M_Nyq = 4.  Centres c_k = round(Om_k) except the top
    bin, relocated to round((Om_{nb-2} + L/2)/2) as the NSGT reference code
    does; c_DC = 0, c_Nyq = L/2.
  * window g_k[m] = I0(beta*sqrt(1-(2m/M_k)^2))/I0(beta) for
    m = -floor(M_k/2) .. M_k-floor(M_k/2)-1  (Kaiser, peak on the centre bin);
    the DC and Nyquist bands are their own mirror image and use the symmetric
    support m = -floor(M/2) .. floor(M/2) so that d[n] = d[-n].
  * "oct" rasterisation: every bin of octave j has T_j = nextpow2(max M_k)
    coefficients; DC and Nyquist bands are not output.
  * analysis: X = FFT_L(x);  c_k = IFFT_{T_j}(fold(X[(c_k+m) mod L] g_k[m])),
    fold puts offset m at index m mod T_j.
  * dual frame (painless case): d[n] = sum over ALL bands (DC, bins, Nyquist
    and the mirrored negative-frequency bins) of T_k g_k[n-c_k]^2,
    gd_k[m] = g_k[m] / d[(c_k+m) mod L].
  * synthesis: P[(c_k+m) mod L] += T_j * FFT_{T_j}(c_k)[m mod T_j] * gd_k[m];
    the mirrored bands contribute conj(P[-n]); x = Re IFFT_L(P + conj(P[-n])).
  * apply_hpf_DC: x -> Re IFFT(FFT(x) * H), H = 1 - (DC and Nyquist bands'
    T g^2 / d)  ==  response of bwd(fwd(.)).
"""
import math

import numpy as np
import torch


def _next_pow2(v):
    return 1 << int(math.ceil(math.log2(max(int(v), 1))))


def nsgt_design(fs, L, numocts=7, binsoct=64, beta=1.0):
    """Band geometry in float64 numpy. Returns dict of per-band tables."""
    assert L % 2 == 0
    nb = numocts * binsoct
    fmax = fs / 2.0 - 1e-6
    fmin = fmax / 2.0 ** numocts
    r = 2.0 ** (numocts / (nb - 1.0))
    f = fmin * r ** np.arange(nb, dtype=np.float64)
    Q = math.sqrt(r) / (r - 1.0) / 2.0
    Om = f * L / fs
    M = np.zeros(nb, dtype=np.int64)
    M[1:-1] = np.round(Om[2:] - Om[:-2]).astype(np.int64)
    M[0] = int(np.round(Om[0] / Q))
    M[-1] = int(np.round(Om[-1] / Q))
    M = np.maximum(M, 4)
    c = np.round(Om).astype(np.int64)
    c[-1] = int(np.round((Om[-2] + L / 2.0) / 2.0))
    M_dc = max(int(np.round(2.0 * Om[0])), 4)
    M_ny = 4
    T = np.zeros(nb, dtype=np.int64)
    T_oct = []
    for j in range(numocts):
        sl = slice(j * binsoct, (j + 1) * binsoct)
        t = _next_pow2(M[sl].max())
        T[sl] = t
        T_oct.append(t)
    return dict(fs=fs, L=L, nb=nb, numocts=numocts, binsoct=binsoct, beta=beta,
                f=f, Om=Om, M=M, c=c, T=T, T_oct=T_oct, M_dc=M_dc, M_ny=M_ny)


def kaiser_centered(Mk, beta, symmetric=False):
    """g[m], m=-floor(M/2)..M-floor(M/2)-1 (float64); symmetric=True (DC/Nyquist,
    which are their own mirror image) uses m=-floor(M/2)..floor(M/2)."""
    hi = (Mk // 2) + 1 if symmetric else Mk - (Mk // 2)
    m = np.arange(-(Mk // 2), hi, dtype=np.float64)
    arg = 1.0 - (2.0 * m / Mk) ** 2
    arg = np.maximum(arg, 0.0)
    return np.i0(beta * np.sqrt(arg)) / np.i0(beta)


class CQT_nsgt:
    """Drop-in for cqt_nsgt_pytorch.CQT_nsgt (mode='oct' only), autograd-transparent."""

    def __init__(self, numocts, binsoct, mode="oct", window=("kaiser", 1), fs=44100,
                 audio_len=44100, device="cpu", dtype=torch.float32):
        assert mode == "oct"
        if isinstance(window, (tuple, list)):
            assert window[0] == "kaiser"
            beta = float(window[1])
        else:
            raise NotImplementedError("only ('kaiser', beta) windows")
        self.Ls = int(audio_len)
        self.fs = fs
        self.numocts = numocts
        self.binsoct = binsoct
        self.dtype = dtype
        self.cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.device = torch.device(device)
        d = nsgt_design(fs, self.Ls, numocts, binsoct, beta)
        self.design = d
        L = self.Ls
        nb = d["nb"]
        # dual-frame diagonal over the full circle
        diag = np.zeros(L, dtype=np.float64)
        g_list = []
        for k in range(nb):
            g = kaiser_centered(int(d["M"][k]), beta)
            g_list.append(g)
            m = np.arange(-(len(g) // 2), len(g) - (len(g) // 2))
            idx = (d["c"][k] + m) % L
            np.add.at(diag, idx, d["T"][k] * g * g)
            idxm = (-(d["c"][k] + m)) % L           # mirrored band
            np.add.at(diag, idxm, d["T"][k] * g * g)
        g_dc = kaiser_centered(d["M_dc"], beta, symmetric=True)
        m = np.arange(-(d["M_dc"] // 2), d["M_dc"] // 2 + 1)
        lp = np.zeros(L, dtype=np.float64)
        np.add.at(lp, m % L, d["M_dc"] * g_dc * g_dc)
        g_ny = kaiser_centered(d["M_ny"], beta, symmetric=True)
        m = np.arange(-(d["M_ny"] // 2), d["M_ny"] // 2 + 1)
        np.add.at(lp, (L // 2 + m) % L, d["M_ny"] * g_ny * g_ny)
        diag += lp
        assert diag.min() > 0
        self.diag = diag
        self.Hhpf_full = torch.tensor(1.0 - lp / diag, dtype=dtype, device=self.device)
        # per-octave gather/scatter tables
        self.octs = []
        for j in range(numocts):
            Tj = int(d["T_oct"][j])
            ks = range(j * binsoct, (j + 1) * binsoct)
            Mmax = max(int(d["M"][k]) for k in ks)
            idx = np.zeros((binsoct, Mmax), dtype=np.int64)
            pos = np.zeros((binsoct, Mmax), dtype=np.int64)
            win = np.zeros((binsoct, Mmax), dtype=np.float64)
            dwin = np.zeros((binsoct, Mmax), dtype=np.float64)
            for i, k in enumerate(ks):
                g = g_list[k]
                Mk = len(g)
                m = np.arange(-(Mk // 2), Mk - (Mk // 2))
                ii = (d["c"][k] + m) % L
                idx[i, :Mk] = ii
                pos[i, :Mk] = m % Tj
                win[i, :Mk] = g
                dwin[i, :Mk] = g / diag[ii] * Tj
                # padding entries: point at a unique unused fold slot with zero window
                free = np.setdiff1d(np.arange(Tj), pos[i, :Mk])
                pos[i, Mk:] = free[: Mmax - Mk]
                idx[i, Mk:] = 0
            self.octs.append(dict(
                T=Tj,
                idx=torch.tensor(idx, device=self.device),
                pos=torch.tensor(pos, device=self.device),
                win=torch.tensor(win, dtype=dtype, device=self.device),
                dwin=torch.tensor(dwin, dtype=dtype, device=self.device)))

    # -- analysis -----------------------------------------------------------
    def fwd(self, x):
        """x [B,1,L] real -> list of numocts complex tensors [B,1,binsoct,T_j] (index 0 = lowest octave)."""
        assert x.shape[-1] == self.Ls
        X = torch.fft.fft(x.to(self.dtype), dim=-1)          # [B,1,L]
        out = []
        for o in self.octs:
            vals = X[..., o["idx"]] * o["win"]                # [B,1,64,Mmax]
            buf = torch.zeros(*vals.shape[:-1], o["T"], dtype=vals.dtype, device=vals.device)
            buf = buf.scatter(-1, o["pos"].expand(vals.shape), vals)
            out.append(torch.fft.ifft(buf, dim=-1))
        return out

    # -- synthesis ----------------------------------------------------------
    def bwd(self, clist):
        L = self.Ls
        B = clist[0].shape[0]
        P = torch.zeros(B, 1, L, dtype=self.cdtype, device=clist[0].device)
        for c, o in zip(clist, self.octs):
            fc = torch.fft.fft(c.to(self.cdtype), dim=-1)
            vals = torch.gather(fc, -1, o["pos"].expand(fc.shape[:-1] + o["pos"].shape[-1:])) * o["dwin"]
            flat_idx = o["idx"].reshape(-1)
            P = P.index_add(-1, flat_idx, vals.reshape(B, 1, -1))
        # (the builder's own choice: the conjugate-mirrored bands are ADDED before a full inverse FFT; the library is believed
        # to irfft the positive half only - SURVEY App. B "pitfall".  FIRST SUSPECT if tests/golden/cqt_lib.npz
        # (tests/golden/make_cqt_golden.py, written only where cqt_nsgt_pytorch is importable) ever disagrees.)
        Pm = torch.conj(torch.roll(torch.flip(P, dims=(-1,)), 1, dims=-1))   # P[-n]
        x = torch.fft.ifft(P + Pm, dim=-1).real
        return x.to(self.dtype)

    def apply_hpf_DC(self, x):
        L = self.Ls
        if x.shape[-1] < L:
            x = torch.nn.functional.pad(x, (0, L - x.shape[-1]))
        elif x.shape[-1] > L:
            raise ValueError("Input signal is longer than the maximum length")
        X = torch.fft.fft(x.to(self.dtype), dim=-1)
        return torch.fft.ifft(X * self.Hhpf_full, dim=-1).real.to(self.dtype)
