"""csrc/score_eval.hip: one score evaluation as ONE C-ABI call (babe_score_eval: UNet plan + CQT plan + STFT tables in a
babe_eval_desc) against BlindSampler.evaluate sequencing the same kernels from Python.  Needs a MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(monkeypatch, B, semantics):
    import babe_amd.testing.blind_bwe_sampler as bs
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    Ns, L, fs = [8, 8, 8, 8, 16, 16, 16], 92092, 22050
    args = default_args(sample_rate=fs, audio_len=L, Ns=Ns, T=3, start_sigma=0.05)
    monkeypatch.setenv("BABE_CQT_C", "1")                       # the Python sequencer on the library's CQT plan (same tables)
    net = Unet_CQT_oct_with_attention(args, "cuda")
    net.load_state_dict(init_state_dict(Ns, args.network.num_dils, seed=0, gate_scale=1.0))
    smp = bs.BlindSampler(net, EDM(args), args, batch_semantics=semantics)
    g = torch.Generator().manual_seed(5)
    y = (0.1 * torch.randn(B, L, generator=g)).cuda()
    x = (y.cpu() + 0.05 * torch.randn(B, L, generator=g)).cuda()
    st = smp.stft_ops(L, y.device)
    specY = st.stft(y)
    P = 1 if semantics == "reference" else B
    ic = smp.args.tester.blind_bwe.initial_conditions
    fp = torch.tensor([list(ic.fc), list(ic.A)], dtype=torch.float32).unsqueeze(0).repeat(P, 1, 1).cuda()
    return bs, smp, x, y, specY, fp


@pytest.mark.parametrize("B,semantics,blind", [(1, "per_clip", True), (2, "per_clip", True), (2, "reference", True), (1, "per_clip", False)])
def test_score_eval_equals_the_python_sequencer_bit_for_bit(monkeypatch, B, semantics, blind):
    bs, smp, x, y, specY, fp = _setup(monkeypatch, B, semantics)
    outs = {}
    for mode in (False, True, False, True):                    # twice each: the saved UNet state of one path must not leak into the other
        monkeypatch.setattr(bs, "EVAL_C", mode)
        res = []
        for t in (0.05, 0.011):
            d, x_den, p = smp.evaluate(x, t, y, specY, fp, blind, lane=0)
            res.append((d.clone(), x_den.clone(), p.clone()))
        torch.cuda.synchronize()
        if mode in outs:
            for a, b in zip(outs[mode], res):
                assert all(torch.equal(u, v) for u, v in zip(a, b)), "not deterministic"
        outs[mode] = res
    assert smp._ceval, "the library path did not run"
    for (d0, x0, p0), (d1, x1, p1) in zip(outs[False], outs[True]):
        assert torch.isfinite(d1).all()
        assert torch.equal(x0, x1), float((x0 - x1).abs().max())
        assert torch.equal(p0, p1), (p0, p1)
        assert torch.equal(d0, d1), float((d0 - d1).abs().max())


def test_whole_sampler_run_on_the_library_evaluation(monkeypatch):
    """predict_blind_bwe T = 3 with BABE_EVAL_C on = off, bit for bit (single stream and two clip lanes)."""
    bs, smp, x, y, specY, fp = _setup(monkeypatch, 2, "per_clip")
    L = y.shape[1]
    g = torch.Generator().manual_seed(9)
    noises = [torch.randn(2, L, generator=g) for _ in range(4)]
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(bs, "EVAL_C", mode)
        it = iter(noises)
        smp._randn = lambda shape, device: next(it).to(device)
        xr, fpr = smp.predict_blind_bwe(y)
        torch.cuda.synchronize()
        res[mode] = (xr.clone(), fpr.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])


def test_eval_desc_has_the_headers_layout(tmp_path):
    import ctypes
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from babe_amd.testing.eval_c import EvalDesc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   ' sizeof(babe_eval_desc), offsetof(babe_eval_desc, emb_dim), offsetof(babe_eval_desc, film_J), offsetof(babe_eval_desc, fit),'
                   ' offsetof(babe_eval_desc, blind), offsetof(babe_eval_desc, audio_len_norm));return 0;}\n'
                   % os.path.join(root, "include", "babe_hip.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    want = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    got = [ctypes.sizeof(EvalDesc), EvalDesc.emb_dim.offset, EvalDesc.film_J.offset, EvalDesc.fit.offset, EvalDesc.blind.offset,
           EvalDesc.audio_len_norm.offset]
    assert got == want, (got, want)
