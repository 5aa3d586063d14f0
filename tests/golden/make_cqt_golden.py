"""Pin for the CQT, to be run the day the third-party package is available.

The reference's transform is `cqt_nsgt_pytorch.CQT_nsgt` (PyPI `cqt-nsgt-pytorch`, upstream eloimoliner/CQT_pytorch; call
sites /root/reference/networks/cqtdiff+.py:9,620,743,841 and testing/blind_bwe_sampler.py:156,169).  It is neither in
/root/reference nor installable here (no network), so oracle/nsgt.py restates it from the NSGT literature and is checked by
invariants only: "parity unpinned" for this one sub-component.

    python tests/golden/make_cqt_golden.py

IF `import cqt_nsgt_pytorch` succeeds, this script writes tests/golden/cqt_lib.npz: the library's fwd / bwd / apply_hpf_DC on
seeded white noise at the two geometries the build uses (22.05 kHz / 92092 samples and 44.1 kHz / 368368 samples, 7 octaves x 64
bins, Kaiser beta = 1 - the reference's cqtdiff+.yaml), outputs subsampled so that the fixture stays small.  tests/
test_oracle_nsgt.py (CPU oracle) and tests/test_gpu_cqt.py (HIP) compare against that file when it exists and skip otherwise.
Without the package it prints why it cannot run and exits with status 3.  Data only: no library source is copied.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GEOMETRIES = ((22050, 92092), (44100, 368368))
SUB = 8                                   # keep every 8th time sample of the coefficients / signals


def probe(L, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.1 * torch.randn(2, 1, L, generator=g)


def main():
    try:
        from cqt_nsgt_pytorch import CQT_nsgt
    except Exception as e:                                   # noqa: BLE001
        print(f"cqt_nsgt_pytorch is not importable here ({type(e).__name__}: {e}): tests/golden/cqt_lib.npz NOT written; "
              "the CQT stays 'parity unpinned' (oracle/nsgt.py header)")
        return 3
    out = {}
    for fs, L in GEOMETRIES:
        cqt = CQT_nsgt(7, 64, mode="oct", window=("kaiser", 1), fs=fs, audio_len=L, dtype=torch.float32, device="cpu")
        x = probe(L, 1000 + fs)
        X = cqt.fwd(x)                                       # list of 7 complex tensors [B, 1, 64, T_j], lowest octave first
        xr = cqt.bwd(X)
        xh = cqt.apply_hpf_DC(x.squeeze(1))
        tag = f"{fs}_{L}"
        out[f"{tag}.seed"] = 1000 + fs
        out[f"{tag}.T_oct"] = np.array([int(c.shape[-1]) for c in X])
        for j, c in enumerate(X):
            c = torch.view_as_real(c.squeeze(1) if c.dim() == 4 else c)[..., ::max(1, SUB // 2), :]
            out[f"{tag}.fwd{j}"] = c.numpy()
        out[f"{tag}.bwd"] = xr.reshape(2, -1)[:, ::SUB].numpy()
        out[f"{tag}.hpf"] = xh.reshape(2, -1)[:, ::SUB].numpy()
    np.savez_compressed(os.path.join(HERE, "cqt_lib.npz"), **out)
    print("wrote cqt_lib.npz", {k: np.asarray(v).shape for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
