"""ORACLE (test infrastructure) - sample-rate conversion of the file-level flows.

Restates `torchaudio.functional.resample` (PyPI `torchaudio`; the reference pins no version and ships no requirements file), the
function the reference calls at /root/reference/testing/blind_bwe_tester.py:410,744,930 and
/root/reference/testing/denoise_and_bwe_tester.py:282-289, from its PUBLISHED algorithm (torchaudio >= 0.9,
`functional.functional._get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`): windowed-sinc interpolation, Hann window
cos^2(pi t / (2 w)) with w = lowpass_filter_width = 6, cut-off rolloff = 0.99 of the lower Nyquist, rates divided by their gcd,
one kernel row per output phase, zero padding (width, width + orig), strided convolution, crop to ceil(new L / orig).

**PARITY UNPINNED**: torchaudio is neither under /root/reference nor installed in the build container, and the reference has no
test or fixture at this boundary.  What pins this file instead: the invariants in tests/test_oracle_resample.py (identity at equal
rates, length rule, unit DC gain, pass-band tones preserved, tones above the new Nyquist rejected, agreement with
scipy.signal.resample_poly in the pass band) and, the day `import torchaudio` works, tests/golden/make_resample_golden.py ->
tests/golden/resample_lib.npz (compared in the same test file, skipped until then).
"""
import math

import torch


def sinc_kernel(orig, new, lowpass_filter_width=6, rolloff=0.99, dtype=torch.float32):
    """[new, 1, 2 width + orig] interpolation kernel and width, for rates already reduced by their gcd."""
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    taps = torch.arange(-width, width + orig, dtype=dtype)[None, None] / orig            # input sample times [s at rate 1]
    phase = torch.arange(0, -new, -1, dtype=dtype)[:, None, None] / new                   # minus the output sample times
    t = phase + taps
    t *= base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    win = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    k = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    k *= win * (base / orig)
    return k, width


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """[..., L] -> [..., ceil(new L / orig)] (CPU, the waveform's dtype)."""
    if int(orig_freq) == int(new_freq):
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    k, width = sinc_kernel(orig, new, lowpass_filter_width, rolloff, waveform.dtype)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    n, L = x.shape
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], k, stride=orig)                            # [n, new, frames]
    y = y.transpose(1, 2).reshape(n, -1)
    target = int(torch.ceil(torch.as_tensor(new * L / orig)).long())
    return y[..., :target].reshape(*shape[:-1], target)
