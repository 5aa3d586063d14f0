"""Known low-pass degradations used by config #1.  Mirrors /root/reference/utils/bandwidth_extension.py:
get_FIR_lowpass :59-74 (scipy.signal.firwin, Kaiser window, host side exactly like the reference) and
apply_low_pass_firwin :76-95 (here: the babe_fir_same HIP kernel instead of F.conv1d)."""
import scipy.signal
import torch

from ..stft import fir_same


def get_FIR_lowpass(order, fc, beta, sr):
    B = scipy.signal.firwin(numtaps=order, cutoff=fc, width=beta, window="kaiser", fs=sr)
    return torch.FloatTensor(B).unsqueeze(0).unsqueeze(0)


def apply_low_pass_firwin(y, filter):
    return fir_same(y.contiguous().float(), filter.to(y.device))
