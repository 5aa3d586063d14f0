"""Constant-Q transform (NSGT, mode "oct") on the babe_hip kernels.

Stands in for the reference's third-party ``cqt_nsgt_pytorch.CQT_nsgt`` (constructed at
/root/reference/networks/cqtdiff+.py:620; ``.fwd`` :743, ``.bwd`` :841, ``.apply_hpf_DC``
testing/blind_bwe_sampler.py:156).  The mathematical definition is stated in full in
DESIGN.md §CQT (the dependency's source is not available, so this is a restatement of the
published NSGT algorithm; parity with the dependency is unpinned).

Pipeline (all device work is C-ABI calls):
  rfft_L   : four-step N1 x N2 DFT, both dense DFT stages are 1x1 "convs" on the fp32 MFMA conv
             kernel (weights = DFT matrices), with a twiddle+transpose kernel in between
  analysis : per band gather * window -> fold -> IFFT_T in LDS -> planar coefficients
  synthesis: per band FFT_T in LDS * dual window -> band spectra -> CSR gather (overlap-add in
             frequency, incl. the conjugate-mirrored bands) -> irfft_L (= transposed four-step)
Every linear map has its exact adjoint (same kernels, other window table), used by the VJP.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, stream


# ----------------------------------------------------------------------------- band design
def _next_pow2(v):
    return 1 << int(math.ceil(math.log2(max(int(v), 1))))


def _kaiser(Mk, beta, symmetric=False):
    hi = (Mk // 2) + 1 if symmetric else Mk - (Mk // 2)
    m = np.arange(-(Mk // 2), hi, dtype=np.float64)
    arg = np.maximum(1.0 - (2.0 * m / Mk) ** 2, 0.0)
    return np.i0(beta * np.sqrt(arg)) / np.i0(beta)


def design_bands(fs, L, numocts=7, binsoct=64, beta=1.0):
    """Band geometry + windows + dual windows + high-pass response (float64)."""
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
    for j in range(numocts):
        sl = slice(j * binsoct, (j + 1) * binsoct)
        T[sl] = _next_pow2(M[sl].max())
    assert T.max() <= 4096, "band longer than the LDS FFT supports"
    woff = np.concatenate(([0], np.cumsum(M)[:-1]))
    nwin = int(M.sum())
    g = np.zeros(nwin)
    idx = np.zeros(nwin, dtype=np.int64)              # spectral index (mod L) of every window sample
    diag = np.zeros(L)
    for k in range(nb):
        gk = _kaiser(int(M[k]), beta)
        m = np.arange(-(M[k] // 2), M[k] - (M[k] // 2))
        ii = (c[k] + m) % L
        g[woff[k]: woff[k] + M[k]] = gk
        idx[woff[k]: woff[k] + M[k]] = ii
        np.add.at(diag, ii, T[k] * gk * gk)
        np.add.at(diag, (-ii) % L, T[k] * gk * gk)
    lp = np.zeros(L)
    gd_ = _kaiser(M_dc, beta, True)
    np.add.at(lp, np.arange(-(M_dc // 2), M_dc // 2 + 1) % L, M_dc * gd_ * gd_)
    gn_ = _kaiser(M_ny, beta, True)
    np.add.at(lp, (L // 2 + np.arange(-(M_ny // 2), M_ny // 2 + 1)) % L, M_ny * gn_ * gn_)
    diag += lp
    assert diag.min() > 0
    Tw = np.repeat(T, M).astype(np.float64)
    gdual = g / diag[idx]
    hpf = 1.0 - lp / diag
    # CSR over n in [0, L/2]: entries = window samples landing on n directly or through the mirror
    tgt = np.where(idx <= L // 2, idx, L - idx)
    conj = idx > L // 2
    order = np.argsort(tgt, kind="stable")
    rowptr = np.zeros(L // 2 + 2, dtype=np.int64)
    np.add.at(rowptr, tgt + 1, 1)
    rowptr = np.cumsum(rowptr)
    src = order.astype(np.int64)
    src_signed = np.where(conj[order], src - (1 << 31), src)
    return dict(nb=nb, numocts=numocts, binsoct=binsoct, L=L, fs=fs, M=M, c=c, T=T, woff=woff, nwin=nwin, g=g,
                gdual=gdual, Tw=Tw, hpf=hpf[: L // 2 + 1], rowptr=rowptr, src=src_signed, f=f, Om=Om, M_dc=M_dc)


def factor_len(L):
    """L = N1*N2, N1 <= N2 as balanced as possible; among the balanced pairs (N2 <= 1.25 N1) one with BOTH factors
    multiples of 4 is preferred: each factor is the row length ("positions") of one dense DFT stage, and 16-byte aligned
    rows put all four stages on the pipelined all-DMA (1,1) kernel (368368 = 572 x 644 instead of 598 x 616: the stage over
    598 positions ran 144 us on the generic kernel)."""
    best = best4 = None
    for a in range(1, int(math.isqrt(L)) + 1):
        if L % a == 0:
            best = (a, L // a)
            if a % 4 == 0 and (L // a) % 4 == 0 and 4 * (L // a) <= 5 * a:
                best4 = (a, L // a)
    N1, N2 = best4 or best
    if N2 > 4096:
        raise ValueError(f"audio_len={L} has no balanced factorisation (got {N1}x{N2}); unsupported length")
    return N1, N2


def small_radices(N):
    """N as a product of at most 6 radices of csrc/fft_mixed.hip (4 preferred over 2 x 2, large radices last), or None."""
    out, n = [], N
    for r in (4, 2, 3, 5, 7, 11, 13, 23):
        while n % r == 0 and not (r == 2 and n % 4 == 0):
            out.append(r)
            n //= r
    return out if (n == 1 and 1 <= len(out) <= 6) else None


DFT_SLOT = 5        # BABE_SLOT_DFT_STAGE (csrc/prof.h)


class _BandsStruct(C.Structure):
    _fields_ = [("nbands", C.c_int), ("L", C.c_int), ("KX", C.c_int),
                ("c", C.c_void_p), ("M", C.c_void_p), ("woff", C.c_void_p), ("log2T", C.c_void_p),
                ("oct", C.c_void_p), ("binoct", C.c_void_p), ("tw4096", C.c_void_p),
                ("nocts", C.c_int), ("binsoct", C.c_int), ("coef", C.c_void_p * 8),
                ("wg_first", C.c_void_p), ("wg_count", C.c_void_p), ("nwg", C.c_int),
                ("wg_rec", C.c_void_p), ("band_rec", C.c_void_p), ("abl", C.c_int),
                ("max_wg_count", C.c_int), ("min_log2T", C.c_int), ("max_log2T", C.c_int),
                ("sum_T", C.c_long), ("sum_M", C.c_long), ("sum_TlogT", C.c_double),
                ("kdeg", C.c_int), ("kpoly", C.c_float * 12)]


def kaiser_poly(beta, tol=1e-9, max_deg=11):
    """Coefficients of the truncated power series I0(beta sqrt(a)) / I0(beta) = sum_j (beta^2/4)^j / (j!)^2 / I0(beta) a^j for
    a in [0, 1] (include/babe_hip.h, babe_cqt_bands::kpoly): smallest degree whose first dropped term is below `tol`
    (beta = 1: degree 5); None when that needs more than `max_deg` terms (the kernels then read the window table)."""
    q = beta * beta / 4.0
    terms, t, j = [1.0], 1.0, 0
    while True:
        j += 1
        t = t * q / (j * j)
        if t < tol:
            break
        terms.append(t)
        if j > max_deg:
            return None
    deg = max(len(terms) - 1, 1)
    co = np.zeros(12)
    co[:len(terms)] = np.asarray(terms) / float(np.i0(beta))
    return deg, co


def _register_sigs():
    L = lib()
    P, I, F, Lg = C.c_void_p, C.c_int, C.c_float, C.c_long
    L.babe_cqt_plan_create.restype = P
    L.babe_cqt_plan_create.argtypes = [C.c_double, I, I, I, C.c_double]
    L.babe_cqt_plan_destroy.argtypes = [P]
    L.babe_cqt_workspace_bytes.restype = Lg
    L.babe_cqt_workspace_bytes.argtypes = [P, I]
    for n in ("babe_cqt_fwd", "babe_cqt_bwd", "babe_cqt_fwd_adjoint", "babe_cqt_bwd_adjoint", "babe_cqt_hpf"):
        getattr(L, n).restype = I
        getattr(L, n).argtypes = [P, P, P, P, I, P]
    L.babe_fft_twiddle_transpose.argtypes = [P, P, P, I, I, I, I, P]
    L.babe_rfft_mixed.argtypes = [P, P, P, P, P, I, I, I, I, P, I, P, I, P, P, P, I, P]
    L.babe_rfft_mixed.restype = C.c_int
    L.babe_cqt_band_analysis.argtypes = [C.POINTER(_BandsStruct), P, P, I, P]
    L.babe_cqt_band_synthesis.argtypes = [C.POINTER(_BandsStruct), P, P, Lg, I, P]
    L.babe_cqt_gather.argtypes = [P, Lg, P, P, P, P, I, I, F, P, I, P]
    L.babe_spec_scale.argtypes = [P, P, P, P, I, I, F, F, I, P]
    for n in ("babe_fft_twiddle_transpose", "babe_cqt_band_analysis", "babe_cqt_band_synthesis", "babe_cqt_gather",
              "babe_spec_scale"):
        getattr(L, n).restype = C.c_int


class RealFFT:
    """Length-L real DFT / its transpose on the GPU (four-step, dense DFT stages on the MFMA conv kernel)."""

    def __init__(self, L, device):
        _register_sigs()                     # (a RealFFT used on its own: ctypes must know the pointer arguments are 64-bit)
        self.L = L
        N1, N2 = factor_len(L)
        self.N1, self.N2 = N1, N2
        K2 = (L // 2) // N1 + 1
        self.K2 = K2
        self.KX = K2 * N1
        k1 = np.arange(N1)[:, None].astype(np.float64)
        n1 = np.arange(N1)[None, :].astype(np.float64)
        ang = 2 * np.pi * ((k1 * n1) % N1) / N1
        W1 = np.concatenate([np.cos(ang), -np.sin(ang)], 0)                       # [2N1][N1]
        k2 = np.arange(K2)[:, None].astype(np.float64)
        n2 = np.arange(N2)[None, :].astype(np.float64)
        ang = 2 * np.pi * ((k2 * n2) % N2) / N2
        Fr, Fi = np.cos(ang), -np.sin(ang)
        W3 = np.block([[Fr, -Fi], [Fi, Fr]])                                       # [2K2][2N2]
        kk = np.arange(N1)[:, None].astype(np.int64) * np.arange(N2)[None, :].astype(np.int64)
        ang = 2 * np.pi * (kk % L).astype(np.float64) / L
        tw = np.stack([np.cos(ang), -np.sin(ang)], -1)                            # [N1][N2][2]
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        # Two packings: a stage has only N2 (1001) or N1 (368) "positions" per clip, so for a few clips 32-channel row
        # tiles (nt=1: the most workgroups, 100 vs 129 us per stage at B <= 2) win, from B = 4 on the default tiling does
        self.W1s, self.W3s = ops.PackedConv(t(W1).reshape(2 * N1, N1, 1, 1), nt=1), ops.PackedConv(t(W3).reshape(2 * K2, 2 * N2, 1, 1), nt=1)
        self.W1b, self.W3b = ops.PackedConv(t(W1).reshape(2 * N1, N1, 1, 1)), ops.PackedConv(t(W3).reshape(2 * K2, 2 * N2, 1, 1))
        self.tw = t(tw).contiguous()
        self.dev = device
        # Mixed-radix plan (csrc/fft_mixed.hip): both factors as products of the radices the Stockham kernel has; otherwise
        # (or with BABE_FFT_MIXED=0) the dense DFT stages above are used
        self.rad1, self.rad2 = small_radices(N1), small_radices(N2)
        self.mixed = (self.rad1 is not None and self.rad2 is not None and max(N1, N2) <= 1077 and
                      os.environ.get("BABE_FFT_MIXED", "1") != "0")
        if self.mixed:
            wN = lambda N: t(np.stack([np.cos(2 * np.pi * np.arange(N) / N), -np.sin(2 * np.pi * np.arange(N) / N)], -1)).contiguous()
            self.w1, self.w2 = wN(N1), wN(N2)
            self.rad1_c = (C.c_int * len(self.rad1))(*self.rad1)
            self.rad2_c = (C.c_int * len(self.rad2))(*self.rad2)

    def rfft(self, x, out=None):
        """x [B,L] -> planar spectrum [B,2,KX] (bins above L/2 hold valid but redundant values)."""
        lib().babe_prof_conv_slot(DFT_SLOT)
        try:
            return self._rfft(x, out)
        finally:
            lib().babe_prof_conv_slot(-1)

    def rfft_T(self, spec, out=None):
        """Transpose (real-linear adjoint) of rfft: planar [B,2,KX] (entries above L/2 must be 0) -> [B,L].
        irfft(X) = rfft_T(c*X/L) with c = 1 at DC/Nyquist and 2 elsewhere."""
        lib().babe_prof_conv_slot(DFT_SLOT)
        try:
            return self._rfft_T(spec, out)
        finally:
            lib().babe_prof_conv_slot(-1)

    def _mixed(self, x_in, spec_out, spec_in, x_out, B, direction):
        work = torch.empty(B, 2, self.L, device=self.dev)
        check(lib().babe_rfft_mixed(ptr(x_in) if x_in is not None else None, ptr(spec_out) if spec_out is not None else None,
                                    ptr(spec_in) if spec_in is not None else None, ptr(x_out) if x_out is not None else None,
                                    ptr(work), B, self.N1, self.N2, self.K2, self.rad1_c, len(self.rad1), self.rad2_c,
                                    len(self.rad2), ptr(self.w1), ptr(self.w2), ptr(self.tw), direction, stream()), "rfft_mixed")

    def _rfft(self, x, out=None):
        B = x.shape[0]
        N1, N2, K2 = self.N1, self.N2, self.K2
        if self.mixed:
            x = x.reshape(B, self.L)
            x = x if x.is_contiguous() else x.contiguous()
            if out is None:
                out = torch.empty(B, 2, self.KX, device=self.dev)
            self._mixed(x, out, None, None, B, 0)
            return out
        self.W1, self.W3 = (self.W1s, self.W3s) if B < 4 else (self.W1b, self.W3b)
        A = torch.empty(B, 2 * N1, 1, N2, device=self.dev)
        ops.conv2d(x.reshape(B, N1, 1, N2), self.W1, A)
        At = torch.empty(B, 2 * N2, 1, N1, device=self.dev)
        check(lib().babe_fft_twiddle_transpose(ptr(A), ptr(At), ptr(self.tw), B, N1, N2, 0, stream()), "twiddle")
        if out is None:
            out = torch.empty(B, 2, self.KX, device=self.dev)
        ops.conv2d(At, self.W3, out.view(B, 2 * K2, 1, N1))
        return out

    def _rfft_T(self, spec, out=None):
        B = spec.shape[0]
        N1, N2, K2 = self.N1, self.N2, self.K2
        if self.mixed:
            spec = spec if spec.is_contiguous() else spec.contiguous()
            if out is None:
                out = torch.empty(B, self.L, device=self.dev)
            self._mixed(None, None, spec, out, B, 1)
            return out
        self.W1, self.W3 = (self.W1s, self.W3s) if B < 4 else (self.W1b, self.W3b)
        At = torch.empty(B, 2 * N2, 1, N1, device=self.dev)
        ops.conv2d(spec.view(B, 2 * K2, 1, N1), self.W3, At, transpose=True)
        A = torch.empty(B, 2 * N1, 1, N2, device=self.dev)
        check(lib().babe_fft_twiddle_transpose(ptr(At), ptr(A), ptr(self.tw), B, N1, N2, 1, stream()), "twiddle^T")
        if out is None:
            out = torch.empty(B, self.L, device=self.dev)
        ops.conv2d(A, self.W1, out.view(B, N1, 1, N2), transpose=True)
        return out


class CQT_nsgt:
    """HIP-backed drop-in for cqt_nsgt_pytorch.CQT_nsgt(numocts, binsoct, mode="oct", window=("kaiser",beta), ...)."""

    def __init__(self, numocts, binsoct, mode="oct", window=("kaiser", 1), fs=44100, audio_len=44100,
                 device="cuda", dtype=torch.float32):
        if mode != "oct":
            raise NotImplementedError("only mode='oct' is implemented")
        if not (isinstance(window, (tuple, list)) and window[0] == "kaiser"):
            raise NotImplementedError("only ('kaiser', beta) windows are implemented")
        if dtype != torch.float32:
            raise NotImplementedError("fp32 only")
        _register_sigs()
        self.Ls, self.fs, self.numocts, self.binsoct = int(audio_len), fs, numocts, binsoct
        self.device = torch.device(device)
        d = design_bands(fs, self.Ls, numocts, binsoct, float(window[1]))
        d["beta"] = float(window[1])
        self.design = d
        L = self.Ls
        dev = self.device
        self.fft = RealFFT(L, dev)
        KX = self.fft.KX
        ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=dev)
        tf = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
        self.T_oct = [int(d["T"][j * binsoct]) for j in range(numocts)]
        self._tabs = dict(c=ti(d["c"]), M=ti(d["M"]), woff=ti(d["woff"]), log2T=ti(np.log2(d["T"]).astype(np.int64)),
                          oct=ti(np.arange(d["nb"]) // binsoct), binoct=ti(np.arange(d["nb"]) % binsoct))
        # workgroup table of the band-FFT kernels: 4096 points (4096/T bands of one octave, at most the whole octave) each
        wgf, wgc = [], []
        for j in range(numocts):
            bpw = int(min(binsoct, max(1, 4096 // self.T_oct[j])))
            for s0 in range(0, binsoct, bpw):
                wgf.append(j * binsoct + s0)
                wgc.append(min(bpw, binsoct - s0))
        self._tabs["wg_first"], self._tabs["wg_count"] = ti(wgf), ti(wgc)
        # packed copies for the kernel: one 16-byte record per workgroup / per band
        l2b = np.log2(d["T"]).astype(np.int64)
        self._tabs["wg_rec"] = ti(np.stack([np.asarray(wgf), np.asarray(wgc), l2b[wgf],
                                            (np.asarray(wgf) // binsoct) | ((np.asarray(wgf) % binsoct) << 8)], 1).reshape(-1))
        self._tabs["band_rec"] = ti(np.stack([d["c"], d["M"], d["woff"], np.zeros_like(d["c"])], 1).reshape(-1))
        self.nwg = len(wgf)
        self._wg_max = int(max(wgc))
        l2 = np.log2(d["T"]).astype(np.int64)
        self._l2_range = (int(l2.min()), int(l2.max()))
        q = np.arange(2048, dtype=np.float64)
        self.tw4096 = tf(np.stack([np.cos(2 * np.pi * q / 4096), -np.sin(2 * np.pi * q / 4096)], -1)).contiguous()
        g, gd, Tw = d["g"], d["gdual"], d["Tw"]
        # the Kaiser window itself (analysis window of fwd, synthesis-side window of fwd's adjoint) is evaluated INSIDE the band
        # kernels (win = None below; BABE_CQT_ANALYTIC_WIN=0 reads the table); the dual-window tables cannot be
        self.kaiser = kaiser_poly(float(window[1])) if os.environ.get("BABE_CQT_ANALYTIC_WIN", "1") != "0" else None
        self.win_fwd = tf(g / Tw)                         # analysis window incl. the IFFT's 1/T
        self.win_bwd = tf(gd * Tw)                        # synthesis: T * dual window
        self.win_fwd_adj = tf(g / Tw)                     # adjoint of analysis (synthesis-type kernel)
        self.win_bwd_adj = tf(gd * Tw * (2.0 / L))        # adjoint of synthesis (analysis-type kernel, FFT^H = unnormalised IFFT)
        self.rowptr = ti(d["rowptr"])
        self.src = torch.tensor(np.asarray(d["src"], dtype=np.int64).astype(np.int32), dtype=torch.int32, device=dev)
        self.nwin = d["nwin"]
        # the CSR as fixed 16-byte records {src0, src1, src2, count} when no bin has more than three sources
        rp, sr = np.asarray(d["rowptr"]), np.asarray(d["src"], dtype=np.int64)
        cnt = np.diff(rp)
        self.rec = None
        if cnt.max() <= 3:
            rec = np.zeros((L // 2 + 1, 4), dtype=np.int64)
            for e in range(3):
                has = cnt > e
                rec[has, e] = sr[rp[:-1][has] + e]
            rec[:, 3] = cnt
            self.rec = torch.tensor(rec.astype(np.int32).reshape(-1), dtype=torch.int32, device=dev)
        self.hpf = tf(d["hpf"])                           # real response on bins 0..L/2
        cw = np.full(L // 2 + 1, 2.0 / L)
        cw[0] = cw[-1] = 1.0 / L
        self.hpf_irfft = tf(d["hpf"] * cw)                # hpf * c_k / L   (irfft weights folded in)
        self.irfft_w = tf(cw)
        # Every transform below is ONE call into the library's own plan (csrc/cqt_plan.hip: band design in C++, device tables,
        # sequencing) - the path a non-Python host takes, and since round 6 the default, so that this class, babe_score_eval and a
        # C host all run on the SAME tables.  BABE_CQT_C=0: this class sequences the kernels from the numpy design (kept as the
        # cross-check of the library's design: tests/test_cqt_plan_cpu.py, tests/test_gpu_cqt.py).
        self._plan = None
        if os.environ.get("BABE_CQT_C", "1") == "1":
            # (NULL for a length whose factors the mixed-radix FFT does not cover: this class then sequences the kernels itself,
            # with the dense-DFT form of the length-L transform)
            self._plan = lib().babe_cqt_plan_create(float(fs), self.Ls, numocts, binsoct, float(window[1])) or None
            self._ws = {}

    def __del__(self):
        try:
            for k in ("_plan", "_plan_eval"):                                 # (_plan_eval: created by testing/eval_c.py)
                if getattr(self, k, None):
                    lib().babe_cqt_plan_destroy(getattr(self, k))
                    setattr(self, k, None)
        except Exception:                                                     # interpreter shutdown  noqa: BLE001
            pass

    def _plan_call(self, name, a, b, B):
        """a / b: a device tensor or a list of per-octave tensors (passed as an array of pointers), in the entry point's order."""
        key = (B, torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(lib().babe_cqt_workspace_bytes(self._plan, B) // 4 + 1, device=self.device)
            self._ws[key] = ws
        conv = lambda v: (C.c_void_p * len(v))(*[ptr(t) for t in v]) if isinstance(v, (list, tuple)) else ptr(v)
        check(getattr(lib(), name)(self._plan, conv(a), conv(b), ptr(ws), B, stream()), name)

    # ------------------------------------------------------------------ internals
    def _bands(self, coefs):
        s = _BandsStruct()
        d = self.design
        s.nbands, s.L, s.KX = d["nb"], self.Ls, self.fft.KX
        for k in ("c", "M", "woff", "log2T", "oct", "binoct", "wg_first", "wg_count", "wg_rec", "band_rec"):
            setattr(s, k, ptr(self._tabs[k]))
        s.nwg = self.nwg
        s.abl = 0                                         # (ablation builds only: tools/ab/abl_build.sh -DBABE_CQT_ABL)
        s.max_wg_count, (s.min_log2T, s.max_log2T) = self._wg_max, self._l2_range
        s.tw4096 = ptr(self.tw4096)
        s.nocts, s.binsoct = self.numocts, self.binsoct
        s.sum_T, s.sum_M = int(np.sum(d["T"])), int(np.sum(d["M"]))
        s.sum_TlogT = float(np.sum(d["T"] * np.log2(d["T"])))
        if self.kaiser is not None:
            s.kdeg = int(self.kaiser[0])
            for j in range(12):
                s.kpoly[j] = float(self.kaiser[1][j])
        for j, cf in enumerate(coefs):
            assert cf.is_contiguous() and cf.shape[1:] == (2, self.binsoct, self.T_oct[j]), cf.shape
            s.coef[j] = ptr(cf)
        return s

    def alloc_coefs(self, B):
        return [torch.empty(B, 2, self.binsoct, T, device=self.device) for T in self.T_oct]

    def analysis(self, spec, win, coefs=None):
        B = spec.shape[0]
        coefs = coefs or self.alloc_coefs(B)
        s = self._bands(coefs)
        win = None if (win is self.win_fwd and self.kaiser is not None) else win
        check(lib().babe_cqt_band_analysis(C.byref(s), ptr(spec), ptr(win), B, stream()), "cqt_band_analysis")
        return coefs

    def synthesis_spec(self, coefs, win, scale, mul=None, spec=None):
        B = coefs[0].shape[0]
        s = self._bands(coefs)
        bs = torch.empty(B, self.nwin, 2, device=self.device)
        win = None if (win is self.win_fwd_adj and self.kaiser is not None) else win
        check(lib().babe_cqt_band_synthesis(C.byref(s), ptr(bs), ptr(win), self.nwin, B, stream()), "cqt_band_synthesis")
        if spec is None:
            spec = torch.empty(B, 2, self.fft.KX, device=self.device)
        check(lib().babe_cqt_gather(ptr(bs), self.nwin, ptr(self.rowptr), ptr(self.src), ptr(self.rec), ptr(spec),
                                    self.fft.KX, self.Ls, scale, ptr(mul), B, stream()), "cqt_gather")
        return spec

    def spec_scale(self, s1, mul, sc1=1.0, s2=None, sc2=0.0, out=None):
        B = s1.shape[0]
        out = out if out is not None else torch.empty_like(s1)
        check(lib().babe_spec_scale(ptr(s1), ptr(s2), ptr(out), ptr(mul), self.fft.KX, self.Ls, sc1, sc2, B, stream()),
              "spec_scale")
        return out

    # ------------------------------------------------------------------ planar API (used by the UNet)
    def fwd_planar(self, x):
        """x [B,L] -> list of planar coefficient tensors [B,2,binsoct,T_j] (index 0 = lowest octave)."""
        if self._plan:
            x = x.contiguous()
            co = self.alloc_coefs(x.shape[0])
            self._plan_call("babe_cqt_fwd", x, co, x.shape[0])
            return co
        return self.analysis(self.fft.rfft(x), self.win_fwd)

    def bwd_planar(self, coefs):
        if self._plan:
            x = torch.empty(coefs[0].shape[0], self.Ls, device=self.device)
            self._plan_call("babe_cqt_bwd", [c.contiguous() for c in coefs], x, x.shape[0])
            return x
        return self.fft.rfft_T(self.synthesis_spec(coefs, self.win_bwd, 2.0 / self.Ls))

    def fwd_adjoint(self, gcoefs):
        """(fwd_planar)^T : gradients w.r.t. coefficients -> gradient w.r.t. x."""
        if self._plan:
            gx = torch.empty(gcoefs[0].shape[0], self.Ls, device=self.device)
            self._plan_call("babe_cqt_fwd_adjoint", [c.contiguous() for c in gcoefs], gx, gx.shape[0])
            return gx
        return self.fft.rfft_T(self.synthesis_spec(gcoefs, self.win_fwd_adj, 1.0))

    def bwd_adjoint(self, gx):
        """(bwd_planar)^T : gradient w.r.t. x -> gradients w.r.t. coefficients."""
        if self._plan:
            gx = gx.contiguous()
            co = self.alloc_coefs(gx.shape[0])
            self._plan_call("babe_cqt_bwd_adjoint", gx, co, gx.shape[0])
            return co
        return self.analysis(self.fft.rfft(gx), self.win_bwd_adj)

    # ------------------------------------------------------------------ reference API (complex tensors)
    def fwd(self, x):
        """x [B,1,L] -> list of complex tensors [B,1,binsoct,T_j]."""
        assert x.shape[-1] == self.Ls
        co = self.fwd_planar(x.reshape(-1, self.Ls).contiguous().float())
        return [torch.complex(c[:, 0], c[:, 1]).unsqueeze(1) for c in co]

    def bwd(self, clist):
        co = [torch.stack((c.squeeze(1).real, c.squeeze(1).imag), 1).contiguous().float() for c in clist]
        return self.bwd_planar(co).unsqueeze(1)

    def apply_hpf_DC(self, x):
        """Zero-phase removal of the DC and Nyquist bands (self-adjoint). x [B,L'] with L' <= L."""
        L = self.Ls
        if x.shape[-1] > L:
            raise ValueError("Input signal is longer than the maximum length")
        if x.shape[-1] < L:
            xp = torch.zeros(x.shape[0], L, device=self.device)
            xp[:, : x.shape[-1]] = x
            x = xp
        if self._plan:
            x = x.contiguous()
            out = torch.empty_like(x)
            self._plan_call("babe_cqt_hpf", x, out, x.shape[0])
            return out
        spec = self.fft.rfft(x.contiguous())
        return self.fft.rfft_T(self.spec_scale(spec, self.hpf_irfft))
