"""babe_resample_sinc (csrc/resample_sinc.hip, babe_amd/resample.py) against the oracle restatement of
torchaudio.functional.resample (oracle/resample.py; PARITY UNPINNED - the library is absent, tests/golden/make_resample_golden.py
is the hook), the flows that use it, and the command line with a 48 kHz file.  Needs a MI355X."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import resample as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
RATES = [(44100, 22050), (22050, 16000), (16000, 22050), (48000, 44100), (22050, 44100), (44100, 16000)]


@pytest.mark.parametrize("fo,fn", RATES)
def test_hip_resample_equals_oracle(fo, fn):
    from babe_amd.resample import resample
    gen = torch.Generator().manual_seed(fo + fn)
    for shape in [(1, 1), (1, 7), (2, 441), (3, 10007), (1, 2, 4410), (2, 88200)]:        # ragged lengths, batch shapes, 1 sample
        x = 0.1 * torch.randn(*shape, generator=gen)
        y = resample(x.cuda(), fo, fn)
        ref = R.resample(x, fo, fn)
        assert y.shape == ref.shape, (shape, y.shape, ref.shape)
        err = float((y.cpu() - ref).abs().max())
        assert err < 2e-7, (shape, err)                                                    # fp32 sums of ~17 terms of size 0.1


def test_equal_rates_return_the_input_and_cpu_tensors_raise():
    from babe_amd.resample import resample
    x = torch.randn(2, 100).cuda()
    assert resample(x, 16000, 16000) is x
    with pytest.raises(RuntimeError):
        resample(torch.randn(2, 100), 44100, 16000)                                        # no CPU fallback


def test_round_trip_of_a_band_limited_signal():
    """22050 -> 16000 -> 22050 (what config #5 does around the denoiser, the other way round): a signal below 0.6 of the
    narrower Nyquist comes back to 1 %."""
    from babe_amd.resample import resample
    n = 44100
    spec = torch.fft.rfft(torch.randn(n, generator=torch.Generator().manual_seed(2), dtype=torch.float64))
    spec[int(0.6 * 8000 / 11025 * (n // 2)):] = 0
    x = torch.fft.irfft(spec, n=n).float()[None]
    x = (x / x.std() * 0.1).cuda()
    z = resample(resample(x, 22050, 16000), 16000, 22050)
    m = min(z.shape[-1], n)
    e = float((z[0, 500:m - 500] - x[0, 500:m - 500]).pow(2).mean().sqrt()) / 0.1
    assert z.shape[-1] in (n, n + 1) and e < 1e-2, (z.shape, e)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "resample_lib.npz")),
                    reason="tests/golden/resample_lib.npz needs torchaudio (make_resample_golden.py): parity unpinned until then")
def test_hip_resample_vs_torchaudio_golden():
    import importlib.util
    from babe_amd.resample import resample
    spec = importlib.util.spec_from_file_location("mrg", os.path.join(G, "make_resample_golden.py"))
    mrg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mrg)
    g = np.load(os.path.join(G, "resample_lib.npz"))
    for i, (fo, fn, n) in enumerate(mrg.CASES):
        y = resample(mrg.case_input(i, n).cuda(), fo, fn).cpu().numpy()
        assert y.shape == g[f"case{i}"].shape and np.abs(y - g[f"case{i}"]).max() < 1e-6


def test_restore_cli_accepts_a_48_khz_file(tmp_path):
    """python -m babe_amd.restore on a 48 kHz wav: resampled to the model's 22.05 kHz (blind_bwe_tester.py:410), restored (random
    weights, T = 2: the path is what is exercised) and written at the model's rate."""
    from scipy.io import wavfile
    fs_in, fs_m, segL = 48000, 22050, 92092
    n = int(2.2 * fs_in)
    t = np.arange(n) / fs_in
    x = (0.2 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    wav = str(tmp_path / "in48k.wav")
    wavfile.write(wav, fs_in, x)
    out_dir = str(tmp_path / "out")
    code = ("import sys; sys.argv = ['restore', %r, %r, '--T', '2', '--sample-rate', '%d', '--audio-len', '%d']; "
            "import babe_amd.config as c; _d = c.default_args; "
            "c.default_args = lambda **k: _d(**dict(k, Ns=[8, 8, 8, 8, 16, 16, 16])); "
            "from babe_amd import restore; restore.main()") % (wav, out_dir, fs_m, segL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    sr, y = wavfile.read(os.path.join(out_dir, "in48k.wav"))
    assert sr == fs_m and y.shape[0] == math.ceil(n * 147 / 320) and np.isfinite(y).all() and y.std() > 0
