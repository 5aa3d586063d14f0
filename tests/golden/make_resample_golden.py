"""Pin for the sample-rate conversion (oracle/resample.py, babe_amd/resample.py): outputs of the LIBRARY the reference calls,
`torchaudio.functional.resample`, on seeded inputs -> tests/golden/resample_lib.npz.  torchaudio is not installed in the build
container (and not vendored by the reference), so today this script prints why it cannot run and exits 3; the tests that
compare against the file are skipped until it exists.  Data only: inputs are re-derived from seeds, outputs stored.

    python tests/golden/make_resample_golden.py
"""
import os
import sys

import numpy as np
import torch

CASES = [(44100, 22050, 30011), (22050, 16000, 44100), (16000, 22050, 32000), (48000, 44100, 24000), (22050, 44100, 10007)]


def case_input(i, n):
    g = torch.Generator().manual_seed(900 + i)
    return 0.1 * torch.randn(2, n, generator=g)


def main():
    try:
        import torchaudio
    except Exception as e:                                       # noqa: BLE001
        print(f"cannot import torchaudio ({type(e).__name__}: {e}): the resampler stays 'parity unpinned'")
        return 3
    out = {"version": np.array(torchaudio.__version__)}
    for i, (fo, fn, n) in enumerate(CASES):
        out[f"case{i}"] = torchaudio.functional.resample(case_input(i, n), fo, fn).numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_lib.npz"), **out)
    print("wrote resample_lib.npz with torchaudio", torchaudio.__version__)
    return 0


if __name__ == "__main__":
    sys.exit(main())
