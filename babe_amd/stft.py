"""STFT-domain degradation model on the babe_hip kernels (host wrappers).

Mirrors /root/reference/utils/blind_bwe_utils.py (apply_stft :15-26, apply_filter :6-13,
apply_filter_istft :28-39, design_filter :82-119, apply_filter_and_norm_STFTmag_fweighted :250-296)
and BlindSampler.fit_params (/root/reference/testing/blind_bwe_sampler.py:533-595).
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from ._lib import check, lib, ptr, stream

_WEIGHTS = {"None": 0, "sqrt": 1, "linear": 2, "log": 3}


class FitCfg(C.Structure):
    _fields_ = [("mu_fc", C.c_float), ("mu_A", C.c_float), ("tol_fc", C.c_float), ("tol_A", C.c_float),
                ("fcmin", C.c_float), ("fcmax", C.c_float), ("Amin", C.c_float), ("Amax", C.c_float),
                ("max_iter", C.c_int), ("clamp_fc", C.c_int), ("clamp_A", C.c_int), ("only_negative_A", C.c_int),
                ("weighting", C.c_int), ("kernel", C.c_int)]


_registered = False


def _register():
    global _registered
    if _registered:
        return
    L = lib()
    P, I, F, Lg = C.c_void_p, C.c_int, C.c_float, C.c_long
    sig = {
        "babe_stft_fwd": [P, Lg, I, P, P, I, I, I, P, P],
        "babe_spec_filter_istft": [P, P, Lg, P, I, I, I, P, P],
        "babe_ola": [P, P, P, Lg, P, Lg, P, I, I, I, I, I, P],
        "babe_residual_seed": [P, Lg, P, I, P, P, Lg, I, I, P],
        "babe_stft_mag_stats": [P, P, P, I, I, I, I, P],
        "babe_design_filter": [P, P, I, I, I, F, I, P],
        "babe_filter_fit": [P, P, P, I, I, I, F, I, C.POINTER(FitCfg), P],
        "babe_filter_loss_grad": [P, Lg, P, P, I, I, I, F, I, C.POINTER(FitCfg), P],
        "babe_lincomb3": [P, F, P, F, P, F, P, Lg, P],
        "babe_add_obs_noise": [P, Lg, P, Lg, F, I, Lg, P],
        "babe_sumsq_partial": [P, Lg, P, I, I, Lg, P],
        "babe_cos_partial": [P, Lg, P, Lg, P, I, I, Lg, P],
        "babe_stft_dist_partial": [P, P, P, P, I, I, I, I, I, P],
        "babe_stft_dist_grad": [P, P, P, P, I, P, I, I, I, I, I, P],
        "babe_residual_seed_alt": [P, Lg, P, Lg, P, I, P, P, Lg, I, I, I, F, P],
        "babe_score_direction": [P, P, P, P, I, P, F, F, F, I, I, I, Lg, P],
        "babe_fir_same": [P, Lg, P, I, P, Lg, I, I, I, P],
        "babe_mask_blend": [P, P, Lg, P, P, I, Lg, P],
    }
    for n, s in sig.items():
        fn = getattr(L, n)
        fn.argtypes = s
        fn.restype = C.c_int
    _registered = True


def lincomb(out, a, x, b=0.0, y=None, c=0.0, z=None):
    _register()
    n = x.numel()
    assert x.is_contiguous() and out.is_contiguous() and (y is None or y.is_contiguous()) and (z is None or z.is_contiguous())
    check(lib().babe_lincomb3(ptr(out), a, ptr(x), b, ptr(y), c, ptr(z), n, stream()), "lincomb3")
    return out


def add_obs_noise(y, noise, snr):
    """y[b] += sqrt(var(y[b]) / snr) * noise[b] IN PLACE (get_rec_grads :80-86, fit_params :542-548); snr linear."""
    _register()
    B, L = y.shape
    assert y.stride(1) == 1 and noise.stride(1) == 1 and noise.shape == y.shape
    check(lib().babe_add_obs_noise(ptr(y), y.stride(0), ptr(noise), noise.stride(0), float(snr), B, L, stream()), "add_obs_noise")
    return y


def fir_same(x, taps, adjoint=False):
    """F.conv1d(x[:,None], taps[None,None], padding="same") or its transpose; x [B,L] device, taps [ntaps] device."""
    _register()
    B, L = x.shape
    out = torch.empty_like(x)
    taps = taps.reshape(-1).contiguous()
    check(lib().babe_fir_same(ptr(x), x.stride(0), ptr(taps), taps.numel(), ptr(out), out.stride(0), B, L,
                              int(adjoint), stream()), "fir_same")
    return out


def mask_blend(mask, a=None, b=None):
    """mask*a + (1-mask)*b with a/b optional (None = 0); mask [L] (shared) or [B,L]."""
    _register()
    ref = a if a is not None else b
    B, n = ref.shape
    out = torch.empty_like(ref)
    mbs = 0 if mask.dim() == 1 or mask.shape[0] == 1 else mask.stride(0)
    check(lib().babe_mask_blend(ptr(out), ptr(mask.contiguous()), mbs, ptr(a), ptr(b), B, n, stream()), "mask_blend")
    return out


def freq_weights(nbins, kind):
    """Frequency weighting of the STFT-domain guidance distances (utils/blind_bwe_utils.py:159-196): w(f), f = linspace(0, 1)."""
    fr = torch.linspace(0, 1, nbins, dtype=torch.float32)
    table = {"None": lambda: torch.ones_like(fr), "linear": lambda: fr, "sqrt": lambda: torch.sqrt(fr),
             "log": lambda: torch.log2(1 + fr), "quadratic": lambda: fr ** 2, "cubic": lambda: fr ** 3,
             "squared": lambda: fr ** 4, "logquadratic": lambda: torch.log2(1 + fr ** 2),
             "logcubic": lambda: torch.log2(1 + fr ** 3)}
    if kind not in table:
        raise NotImplementedError(f"freq_weighting={kind!r} (implemented: {sorted(table)}; 'log2' / 'log10' are -inf at DC)")
    return table[kind]().contiguous()


class STFTOps:
    """Plan for one (nfft, L, fs): window-envelope table, FFT twiddles, scratch."""

    NBLK = 64

    def __init__(self, nfft, L, fs, device):
        _register()
        self.nfft, self.L, self.fs, self.dev = int(nfft), int(L), float(fs), torch.device(device)
        self.hop = self.nfft // 2
        self.frames = 1 + self.L // self.hop
        self.nbins = self.hop + 1
        q = np.arange(2048, dtype=np.float64)
        self.tw4096 = torch.tensor(np.stack([np.cos(2 * np.pi * q / 4096), -np.sin(2 * np.pi * q / 4096)], -1),
                                   dtype=torch.float32, device=self.dev).contiguous()
        w = torch.hamming_window(self.nfft, dtype=torch.float32)           # periodic, like the reference (:19)
        ntot = self.nfft + self.hop * (self.frames - 1)
        env = torch.zeros(ntot, dtype=torch.float32)
        for t in range(self.frames):
            env[t * self.hop: t * self.hop + self.nfft] += w * w
        self.env_inv = (1.0 / env).to(self.dev)
        self.freqs = torch.fft.rfftfreq(self.nfft, d=1 / fs).to(self.dev)

    # -- kernels
    def stft(self, x):
        B = x.shape[0]
        assert x.shape[1] == self.L and x.stride(1) == 1
        spec = torch.empty(B, self.frames, self.nbins, 2, device=self.dev)
        check(lib().babe_stft_fwd(ptr(x), x.stride(0), self.L, None, ptr(spec), B, self.nfft, self.frames,
                                  ptr(self.tw4096), stream()), "stft_fwd")
        return spec

    def filter_frames(self, spec, H):
        """H: [nbins] (shared) or [B,nbins]."""
        B = spec.shape[0]
        fr = torch.empty(B, self.frames, self.nfft, device=self.dev)
        H_bs = 0 if H.dim() == 1 else H.stride(0)
        check(lib().babe_spec_filter_istft(ptr(spec), ptr(H), H_bs, ptr(fr), B, self.nfft, self.frames,
                                           ptr(self.tw4096), stream()), "spec_filter_istft")
        return fr

    def ola(self, fr, normalise, y=None):
        """Returns overlap-add (cropped to L) or, with y, (residual y-ola, partial sums of squares)."""
        B = fr.shape[0]
        out = torch.empty(B, self.L, device=self.dev)
        part = torch.empty(B, self.NBLK, device=self.dev, dtype=torch.float64) if y is not None else None
        check(lib().babe_ola(ptr(fr), ptr(self.env_inv) if normalise else None, ptr(y), y.stride(0) if y is not None else 0,
                             ptr(out), out.stride(0), ptr(part), self.NBLK, B, self.L, self.nfft, self.frames, stream()), "ola")
        return (out, part) if y is not None else out

    def residual_seed(self, r, part, post=True, norm=2, y=None, beta=1.0):
        """d(distance(y, rec))/d(rec) from r = y - rec, times the overlap-add normalisation if post.
        norm 2 (default): -r/||r|| with the partial sums `part` of ||r||^2; 'smoothl1' / 'cosine': the alternative
        distances of get_rec_grads (testing/blind_bwe_sampler.py:99-103)."""
        B = r.shape[0]
        out = torch.empty_like(r)
        if norm != 2:
            mode = {"smoothl1": 1, "cosine": 2, "ready": 3}[norm]        # 'ready': r already is the gradient w.r.t. rec
            cpart = None
            if mode == 2:
                assert y is not None and y.shape == r.shape
                cpart = torch.empty(B, self.NBLK, 3, device=self.dev, dtype=torch.float64)
                check(lib().babe_cos_partial(ptr(r), r.stride(0), ptr(y), y.stride(0), ptr(cpart), self.NBLK, B, r.shape[1],
                                             stream()), "cos_partial")
            check(lib().babe_residual_seed_alt(ptr(r), r.stride(0), ptr(y), y.stride(0) if y is not None else 0, ptr(cpart),
                                               self.NBLK, ptr(self.env_inv) if post else None, ptr(out), out.stride(0), B,
                                               r.shape[1], mode, float(beta), stream()), "residual_seed_alt")
            return out
        check(lib().babe_residual_seed(ptr(r), r.stride(0), ptr(part), self.NBLK, ptr(self.env_inv) if post else None, ptr(out),
                                       out.stride(0), B, self.L, stream()), "residual_seed")
        return out

    def mag_stats(self, specX, specY, shared=False):
        B = specX.shape[0]
        stats = torch.empty(1 if shared else B, 3, self.nbins, device=self.dev, dtype=torch.float64)
        check(lib().babe_stft_mag_stats(ptr(specX), ptr(specY), ptr(stats), B, self.nbins, self.frames, int(shared),
                                        stream()), "stft_mag_stats")
        return stats

    def design_filter(self, params):
        """params [2,K] or [P,2,K] -> H [nbins] or [P,nbins]."""
        single = params.dim() == 2
        p = (params.unsqueeze(0) if single else params).contiguous().float()
        P_, _, K = p.shape
        H = torch.empty(P_, self.nbins, device=self.dev)
        check(lib().babe_design_filter(ptr(p), ptr(H), P_, K, self.nbins, self.fs, self.nfft, stream()), "design_filter")
        return H[0] if single else H

    def filter_fit(self, stats, params, cfg):
        """params [P,2,K] float32 device, updated in place. Returns n_iter [P] int32."""
        P_, _, K = params.shape
        assert params.is_contiguous() and stats.shape[0] == P_
        nit = torch.empty(P_, dtype=torch.int32, device=self.dev)
        check(lib().babe_filter_fit(ptr(stats), ptr(params), ptr(nit), P_, K, self.nbins, self.fs, self.nfft,
                                    C.byref(cfg), stream()), "filter_fit")
        return nit

    def filter_loss_grad(self, stats, params, cfg):
        """The fit's objective and its gradient at params [P,2,K] WITHOUT a descent step (BlindSampler.optimizer_func + autograd,
        as compute_sweep evaluates them on a grid): returns [P, 1 + 2K] = loss, d/dfc_j, d/dA_j.  stats [P,3,nbins], or [1,3,nbins]
        shared by all P parameter sets."""
        P_, _, K = params.shape
        assert params.is_contiguous() and stats.shape[0] in (1, P_)
        out = torch.empty(P_, 1 + 2 * K, device=self.dev)
        check(lib().babe_filter_loss_grad(ptr(stats), 0 if (stats.shape[0] == 1 and P_ > 1) else 3 * self.nbins, ptr(params), ptr(out),
                                          P_, K, self.nbins, self.fs, self.nfft, C.byref(cfg), stream()), "filter_loss_grad")
        return out

    def distance_grad(self, rec, y, weight, mode, shared=False):
        """d D(y, rec) / d rec for the STFT-domain guidance distances (get_rec_grads :105-115): mode 0 complex, 1 magnitude,
        2 log-magnitude (utils/blind_bwe_utils.py:148-247); weight [nbins].  STFT^T is the overlap-add of windowed inverse
        FFTs of G * nfft * [1, 1/2, ..., 1/2, 1]."""
        B = rec.shape[0]
        X, R = self.stft(rec), self.stft(y)
        part = torch.empty(B, self.NBLK, device=self.dev, dtype=torch.float64)
        check(lib().babe_stft_dist_partial(ptr(X), ptr(R), ptr(weight), ptr(part), self.NBLK, B, self.nbins, self.frames, mode,
                                           stream()), "stft_dist_partial")
        G = torch.empty_like(X)
        check(lib().babe_stft_dist_grad(ptr(X), ptr(R), ptr(weight), ptr(part), self.NBLK, ptr(G), B, self.nbins, self.frames,
                                        mode, int(shared), stream()), "stft_dist_grad")
        if getattr(self, "_Hadj", None) is None:
            h = torch.full((self.nbins,), 0.5 * self.nfft, device=self.dev)
            h[0] = h[-1] = float(self.nfft)
            self._Hadj = h
        return self.ola(self.filter_frames(G, self._Hadj), normalise=False)

    # -- composites
    def apply_filter(self, x, H):
        """x -> crop(istft(stft(x) * H))   (blind_bwe_utils.apply_filter)."""
        return self.ola(self.filter_frames(self.stft(x), H), normalise=True)


def make_fit_cfg(mu=(1000.0, 10.0), tol=(5e-3, 5e-3), max_iter=100, fcmin=20.0, fcmax=22050.0, Amin=-50.0, Amax=30.0,
                 clamp_fc=True, clamp_A=True, only_negative_A=True, weighting="sqrt", kernel=None):
    """kernel: 0 = filter_fit_fast_kernel (default), 1 = the first, reference-order kernel; None takes the process default
    (0 unless the A/B scripts under tools/ set BABE_FIT_FAST=0 - the choice is a field of babe_fit_cfg, not library state)."""
    c = FitCfg()
    if kernel is None:
        kernel = 1 if os.environ.get("BABE_FIT_FAST", "1") == "0" else 0
    c.kernel = int(kernel)
    c.mu_fc, c.mu_A, c.tol_fc, c.tol_A = float(mu[0]), float(mu[1]), float(tol[0]), float(tol[1])
    c.fcmin, c.fcmax, c.Amin, c.Amax = float(fcmin), float(fcmax), float(Amin), float(Amax)
    c.max_iter, c.clamp_fc, c.clamp_A, c.only_negative_A = int(max_iter), int(clamp_fc), int(clamp_A), int(only_negative_A)
    if weighting not in _WEIGHTS:
        raise NotImplementedError(f"freq_weighting_filter={weighting!r} (implemented: {sorted(_WEIGHTS)})")
    c.weighting = _WEIGHTS[weighting]
    return c
