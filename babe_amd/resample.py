"""Sample-rate conversion of the file-level flows on the babe_hip kernel (csrc/resample_sinc.hip).

Mirrors `torchaudio.functional.resample(waveform, orig_freq, new_freq)` - what the reference calls at
/root/reference/testing/blind_bwe_tester.py:410,744,930 and /root/reference/testing/denoise_and_bwe_tester.py:282-289 - as that
function is PUBLISHED (torchaudio >= 0.9, `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`): Hann-windowed sinc
interpolation, lowpass_filter_width = 6, rolloff = 0.99, one kernel row per output phase after dividing both rates by their gcd.
torchaudio is neither in /root/reference nor installed here, so parity with the library is UNPINNED (DESIGN.md says the same;
tests/golden/make_resample_golden.py pins it the day `import torchaudio` works).

The kernel TABLE ([new][2 width + orig] float32) is built on the host with the published sequence of float32 operations; the
convolution - all the arithmetic on audio - is the HIP kernel.  Device tensors only, no CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch

from ._lib import check, lib, ptr, stream

_tables = {}
_registered = False


def _register():
    global _registered
    if not _registered:
        L = lib()
        L.babe_resample_sinc.restype = C.c_int
        L.babe_resample_sinc.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_long, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _registered = True


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """(kernel [new][2 width + orig] float32, width, orig, new, support mask) for rates already divided by their gcd - the published
    `_get_sinc_resample_kernel` with resampling_method='sinc_interp_hann' and dtype = float32 (functional.resample passes the
    waveform's dtype, so the table arithmetic is float32, in this order)."""
    orig, new = int(orig_freq), int(new_freq)
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float32)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float32)[:, None, None] / new + idx
    t *= base_freq
    inside = (t.abs() < lowpass_filter_width)[:, 0, :]        # taps the clamp below leaves alone (the window's support)
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    return kernels[:, 0, :].contiguous(), width, orig, new, inside


def _table(orig_freq, new_freq, lowpass_filter_width, rolloff, device):
    key = (int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff), str(device))
    tb = _tables.get(key)
    if tb is None:
        g = math.gcd(int(orig_freq), int(new_freq))
        k, width, orig, new, inside = sinc_resample_kernel(int(orig_freq) // g, int(new_freq) // g, lowpass_filter_width, rolloff)
        # first / one-past-last tap of every phase inside the window's support.  Where |t| was clamped the published table holds
        # cos(fl32(pi / 2))^2 * sin(fl32(6 pi)) / (6 pi) * scale: |value| < 1e-22 next to taps of order 1 - far below half an ulp of
        # any fp32 partial sum of audio-scale samples, so leaving those taps out changes no result bit (checked against the dense
        # sum in the CPU test suite); it makes the pass 2 * width * ... / taps ~ 25 x shorter for 441 -> 320.
        nz = inside.numpy()
        first = nz.argmax(1)
        last = nz.shape[1] - nz[:, ::-1].argmax(1)
        empty = ~nz.any(1)
        first[empty], last[empty] = 0, 0
        kr = np.stack([first, last], 1).astype(np.int32)
        tb = (k.to(device), torch.from_numpy(kr).contiguous().to(device), width, orig, new)
        _tables[key] = tb
    return tb


def resampled_length(length, orig_freq, new_freq):
    """ceil(new * length / orig) as the published code computes it: the quotient goes through a float32 tensor."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    return int(np.ceil(np.float32(new * length / orig)))


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann"):
    """waveform [..., L] float32 device tensor -> [..., ceil(new L / orig)].  Equal rates return the input itself, as the
    reference's call does."""
    if resampling_method != "sinc_interp_hann":
        raise NotImplementedError(f"resampling_method={resampling_method!r} (the reference uses the default, sinc_interp_hann)")
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Original frequency and desired frequecy should be positive")
    if int(orig_freq) == int(new_freq):
        return waveform
    if not waveform.is_cuda:
        raise RuntimeError("babe_amd.resample runs on the GPU only (no CPU fallback)")
    _register()
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).contiguous().float()
    B, L = x.shape
    kern, krange, width, orig, new = _table(orig_freq, new_freq, lowpass_filter_width, rolloff, x.device)
    Lo = resampled_length(L, orig_freq, new_freq)
    out = torch.empty(B, Lo, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(lib().babe_resample_sinc(ptr(x), x.stride(0), ptr(out), out.stride(0), B, L, Lo, ptr(kern), ptr(krange), orig, new,
                                       width, stream(x)), "resample_sinc")
    return out.reshape(*shape[:-1], Lo)
