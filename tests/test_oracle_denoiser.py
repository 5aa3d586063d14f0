"""Oracle restatement of the denoiser pre-pass (oracle/denoiser.py) against golden outputs of the reference
(tests/golden/denoiser.npz, produced by tests/golden/make_golden.py::g11 importing networks/denoiser.py and
testing/denoise_and_bwe_tester.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as OD

CFGS = {
    "full": dict(depth=6, num_tfc=3, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513, T=48),
    "s1": dict(depth=3, num_tfc=2, num_stages=1, use_SAM=False, use_fencoding=False, f_dim=129, T=40),
    "nosam": dict(depth=2, num_tfc=1, num_stages=2, use_SAM=False, use_fencoding=True, f_dim=65, T=21),
}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "denoiser.npz"))


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("name", ["s1", "nosam", "full"])
def test_forward_vs_reference_golden(gold, name):
    c = CFGS[name]
    sd = OD.init_state_dict(c, seed=7)
    g = torch.Generator().manual_seed(int(gold[f"{name}_seed"]))
    X = torch.randn(1 if name == "full" else 2, 2, c["T"], c["f_dim"], generator=g)
    torch.set_num_threads(8)
    with torch.no_grad():
        y = OD.denoiser_forward(sd, c, X)
    if c["num_stages"] > 1:
        assert rel(y[0], gold[f"{name}_pred2"]) < 2e-5
        assert rel(y[1], gold[f"{name}_pred1"]) < 2e-5
    else:
        assert rel(y, gold[f"{name}_pred1"]) < 2e-5


def test_param_table_matches_shipped_config():
    c = CFGS["full"]
    shapes = OD.param_shapes(c)
    assert len(shapes) == 293                                   # reference state_dict size (72.59 M parameters)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 72591502
    assert torch.allclose(OD.freq_embeddings(513)[:, 0], torch.cos(torch.pi * torch.arange(513) / 512), atol=1e-6)


def test_segmented_application_vs_reference_golden(gold):
    c = dict(depth=3, num_tfc=1, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)
    sd = OD.init_state_dict(c, seed=11)
    g = torch.Generator().manual_seed(int(gold["seg_seed"]))
    x = 0.1 * torch.randn(2, 20000, generator=g)
    with torch.no_grad():
        y1 = OD.apply_denoiser_model(sd, c, x[:, :8000])
        y = OD.apply_denoiser(sd, c, x, segment_size=8000)
    assert y1.shape == gold["seg_model"].shape and rel(y1, gold["seg_model"]) < 2e-5
    assert y.shape == gold["seg_full"].shape and rel(y, gold["seg_full"]) < 2e-5
