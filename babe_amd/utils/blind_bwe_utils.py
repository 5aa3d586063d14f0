"""Function-level mirror of /root/reference/utils/blind_bwe_utils.py on the babe_hip kernels:
apply_stft :15-26, apply_filter :6-13, design_filter :82-119.  (Plans are cached per (nfft, L, fs, device).)"""
import torch

from ..stft import STFTOps

_plans = {}


def _plan(nfft, L, fs, device):
    key = (int(nfft), int(L), float(fs), str(device))
    if key not in _plans:
        _plans[key] = STFTOps(nfft, L, fs, device)
    return _plans[key]


def apply_stft(x, NFFT, fs=44100):
    """[B,L] -> [B, NFFT/2+1, frames, 2] like torch.view_as_real(torch.stft(...))."""
    st = _plan(NFFT, x.shape[-1], fs, x.device)
    return st.stft(x.contiguous().float()).permute(0, 2, 1, 3)


def design_filter(fc, A, f):
    """f must be torch.fft.rfftfreq(NFFT, d=1/fs) (float32), as at every call site of the reference."""
    nbins = f.shape[0]
    nfft = 2 * (nbins - 1)
    fs = float(f[1]) * nfft
    st = _plan(nfft, 2 * nfft, fs, f.device)
    p = torch.stack([torch.atleast_1d(torch.as_tensor(fc)), torch.atleast_1d(torch.as_tensor(A))]).float().to(f.device)
    return st.design_filter(p)


def apply_filter(x, H, NFFT, fs=44100):
    st = _plan(NFFT, x.shape[-1], fs, x.device)
    return st.apply_filter(x.contiguous().float(), H.contiguous().float())
