"""Host-side logic of the product (no GPU needed): band design vs the oracle, EDM host class vs golden,
config loader, FFT factorisation, parameter inventory."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, name)).items()}


@pytest.mark.parametrize("fs,L", [(22050, 92092), (44100, 368368), (16000, 184184)])
def test_band_design_matches_oracle(fs, L):
    from babe_amd.cqt import design_bands, factor_len
    from oracle.nsgt import CQT_nsgt
    d = design_bands(fs, L)
    o = CQT_nsgt(7, 64, "oct", ("kaiser", 1), fs, L, dtype=torch.float64)
    assert np.array_equal(d["M"], o.design["M"]) and np.array_equal(d["c"], o.design["c"]) and np.array_equal(d["T"], o.design["T"])
    assert np.allclose(d["hpf"], o.Hhpf_full[: L // 2 + 1].numpy(), atol=1e-14)
    # window tables: band k of the product == row of the oracle's per-octave table
    for k in (0, 63, 64, 200, 447):
        j, i = divmod(k, 64)
        Mk = int(d["M"][k])
        w = d["g"][d["woff"][k]: d["woff"][k] + Mk]
        assert np.allclose(w, o.octs[j]["win"][i, :Mk].numpy(), atol=1e-14)
        gd = d["gdual"][d["woff"][k]: d["woff"][k] + Mk] * d["T"][k]
        assert np.allclose(gd, o.octs[j]["dwin"][i, :Mk].numpy(), rtol=1e-12)
    # CSR covers every window sample exactly once
    assert d["rowptr"][-1] == d["nwin"] and len(np.unique(d["src"] & 0x7FFFFFFF)) == d["nwin"]
    N1, N2 = factor_len(L)
    assert N1 * N2 == L and N2 <= 4096


def test_factor_len_rejects_unbalanced():
    from babe_amd.cqt import factor_len
    with pytest.raises(ValueError):
        factor_len(2 * 100003)       # 2 x prime


@pytest.mark.parametrize("name", ["formal", "brass", "train"])
def test_edm_host_class_vs_golden(name):
    from babe_amd.config import to_attr
    from babe_amd.diff_params.edm import EDM
    g = load("edm.npz")
    c = {k: float(g[f"{name}_cfg_{k}"]) for k in ("sigma_data", "sigma_min", "sigma_max", "ro", "Schurn", "Stmin", "Stmax", "Snoise")}
    e = EDM(to_attr(dict(diff_params=dict(c, P_mean=-1.2, P_std=1.2, ro_train=10, aweighting=dict(use_aweighting=False)))))
    for N in (3, 35):
        t = e.create_schedule(N)
        assert torch.equal(t, g[f"{name}_sched_{N}"])
        assert torch.equal(e.create_schedule_from_initial_t(0.2, N), g[f"{name}_sched0_{N}"])
        assert torch.equal(e.get_gamma(t), g[f"{name}_gamma_{N}"])
    s = g[f"{name}_sig"]
    for fn in ("cskip", "cout", "cin", "cnoise"):
        assert torch.allclose(getattr(e, fn)(s), g[f"{name}_{fn}"], rtol=1e-6, atol=0)


def test_config_loader_coerces_exponent_floats(tmp_path):
    from babe_amd.config import default_args, load_yaml
    p = tmp_path / "c.yaml"
    p.write_text("a: 1e-4\nb: {c: 5e-3, d: 'None', e: nyquist}\nl: [1e-8, 2]\n")
    c = load_yaml(str(p))
    assert c.a == 1e-4 and c.b.c == 5e-3 and c.b.d == "None" and c.b.e == "nyquist" and c.l == [1e-8, 2]
    a = default_args()
    assert a.tester.blind_bwe.optimization.mu == [1000, 10] and a.exp.audio_len == 368368 and a.tester.T == 35


def test_parameter_inventory_matches_reference_counts():
    from babe_amd.networks.cqtdiff_plus import init_state_dict, param_specs
    specs = param_specs([64, 96, 96, 128, 128, 256, 256], [2, 3, 4, 5, 6, 7, 7])
    assert len(specs) == 606
    n = sum(int(np.prod(s)) for k, s, _ in specs if not k.endswith(".kernel"))
    assert abs(n / 1e6 - 44.499) < 0.01                             # 44.50 M parameters
    g = load("unet_small.npz")
    ref = {k[3:]: tuple(v.shape) for k, v in g.items() if k.startswith("sd.")}
    mine = {k: tuple(v.shape) for k, v in init_state_dict([8, 8, 8, 8, 16, 16, 16], [2, 3, 4, 5, 6, 7, 7]).items()}
    assert ref == mine


def test_product_refuses_cpu_device():
    from babe_amd.config import default_args
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    with pytest.raises(RuntimeError):
        Unet_CQT_oct_with_attention(default_args(sample_rate=22050, audio_len=92092), "cpu")


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "babe_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f


@pytest.mark.parametrize("L", [441000, 368368 * 3 + 777, 368368, 100000])
def test_long_file_segmentation_and_crossfade(L):
    """Identity 'restoration' must reproduce the file: the Hann halves of the cross-fade sum to one."""
    from babe_amd.testing.long_file import assemble, cut_segments, plan_segments
    segL = 368368
    plan = plan_segments(L, segL)
    hop = segL - 200 - 256
    assert plan[0][0] == 0 and all(b[0] - a[0] == hop for a, b in zip(plan, plan[1:]))
    assert plan[-1][0] + plan[-1][1] == L or len(plan) == 1
    if L == 441000:
        assert [s for s, _ in plan] == [0, hop]          # a 10 s clip is 2 segments (bench.py workload)
    y = torch.randn(L, generator=torch.Generator().manual_seed(L % 1000))
    segs = cut_segments(y, segL, plan)
    out = assemble(segs, plan, L, segL)
    assert torch.allclose(out, y, atol=1e-6)

def test_long_file_plan_and_assemble_vs_reference_formal_test_bwe():
    """plan_segments / assemble against the reference's OWN driver, BlindTester.formal_test_bwe (non-AR, blind;
    /root/reference/testing/blind_bwe_tester.py:421-566), run with its file I/O stubbed (tests/golden/make_golden.py::g21) on a
    2.3-segment file: the segment starts it recorded in its filter pickle, and its output file re-assembled from the three
    per-segment predictions it produced - must be bit-exact (slicing, two Hann halves, overlap-add: no arithmetic but that)."""
    from babe_amd.testing.long_file import assemble, cut_segments, plan_segments
    s = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formal_test_bwe.npz"))
    L, segL, ola = int(s["L"]), 92092, int(s["OLA"])
    plan = plan_segments(L, segL, 200, ola)
    assert [p[0] for p in plan] == [int(v) for v in s["seg_starts"]]
    segs = cut_segments(torch.from_numpy(s["degraded"])[0], segL, plan)
    assert torch.equal(segs[:, ::16], torch.from_numpy(s["seg_in_sub16"]))      # what the reference fed predict_blind_bwe
    out = assemble(torch.from_numpy(s["seg_pred"]), plan, L, segL, 200, ola)
    assert torch.equal(out, torch.from_numpy(s["final"])[0])


def test_checkpoint_conventions_and_wav_io(tmp_path):
    from babe_amd.io import ema_state_dict, read_audio_file, write_audio_file, write_filter_data
    model = {"a.weight": torch.ones(2, 2, requires_grad=True), "b.kernel": torch.zeros(3), "c.bias": torch.ones(2, requires_grad=True)}
    ema = {"a.weight": torch.full((2, 2), 5.0), "b.kernel": torch.zeros(3), "c.bias": torch.full((2,), 7.0)}
    assert torch.equal(ema_state_dict({"ema": ema, "it": 3})["a.weight"], ema["a.weight"])
    sd2 = ema_state_dict({"model": model, "ema_weights": [ema[k] for k in model]})
    assert all(torch.equal(sd2[k], ema[k]) for k in model)
    sd3 = ema_state_dict({"model": model, "ema_weights": [ema["a.weight"], ema["c.bias"]]})
    assert torch.equal(sd3["a.weight"], ema["a.weight"]) and torch.equal(sd3["c.bias"], ema["c.bias"]) and torch.equal(sd3["b.kernel"], model["b.kernel"])
    x = 0.1 * torch.randn(5000)
    p = write_audio_file(x, 44100, "t", str(tmp_path))
    y, sr = read_audio_file(p)
    assert sr == 44100 and torch.allclose(x, y)
    fp = write_filter_data([((0, 10), torch.ones(2, 5))], str(tmp_path), "t")
    import pickle
    assert pickle.load(open(fp, "rb"))[0][0] == (0, 10)


def test_long_file_AR_driver_sequence():
    """Segment bookkeeping of the AR driver with a stand-in sampler that records what it is asked to do."""
    from babe_amd.config import default_args
    from babe_amd.testing.long_file import restore_file_AR

    class Fake:
        def __init__(self):
            self.args = default_args(sample_rate=1000, audio_len=4000)
            self.calls = []

        def predict_bwe(self, seg, filt, filt_type):
            self.calls.append(("bwe", seg.shape))
            return seg.clone()

        def predict_bwe_AR(self, seg, y_masked, filt, filt_type, mask=None):
            self.calls.append(("ar", int(mask.sum())))
            return mask * y_masked + (1 - mask) * seg

    f = Fake()
    y = torch.arange(11000, dtype=torch.float32)
    out = restore_file_AR(f, y, None)
    # identity "restoration": every sample is reproduced (known overlap regions carry the previous result) ...
    hop = 4000 - 250 - 200
    last = 2 * hop
    assert torch.equal(out[:last], y[:last]) and torch.equal(out[last + 250:], y[last + 250:])
    # ... except the known region of the LAST segment, which the reference fills from the tail of the previous
    # prediction (pred[-overlap:], blind_bwe_tester.py:842) - 200 samples (discard_end) later than where that
    # segment starts.  Reproduced as is.
    assert torch.equal(out[last:last + 250], y[last + 200:last + 450])
    assert f.calls[0][0] == "bwe" and all(c[0] == "ar" for c in f.calls[1:]) and len(f.calls) >= 3
    assert f.calls[1][1] == 250                      # 0.25 s overlap at 1 kHz is the known region


def test_committed_bench_line_follows_the_contract():
    """The newest bench line under profiles/ carries every field the driver / judge reads (format regression guard)."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r0[0-9]_bench_v[0-9]*.json")),
                   key=lambda p: (os.path.basename(p)[:3], int("".join(ch for ch in os.path.basename(p).split("_v")[1].split(".")[0].split("_")[0] if ch.isdigit()))))
    headline = [f for f in files if os.path.basename(f).count("_") == 2]          # rNN_bench_vN.json (no suffix)
    d = json.loads(open(headline[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-sec/s" and d["dtype"] == "f32" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["headline"] is True
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and 0 < r["frac"] <= 1.0
    if os.path.basename(headline[-1]).startswith("r01"):
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    else:
        # round 2 on: frac = EXECUTED MFMA flops / peak for the named kernel, algorithmic_frac = achieved / peak; the hbm
        # block carries the per-kernel HBM fractions of the CQT / STFT / GroupNorm / resample / (1,1)-conv kernels
        assert abs(r["algorithmic_frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(r["frac"] - r["executed_tflops"] / r["peak"]) < 1e-3
        assert r["conv_dispatch_counts_timed_region"]["conv53_direct"] == 0
        for k in ("cqt_band_analysis", "cqt_band_synthesis", "stft_fwd", "gn_stats", "resample", "conv11"):
            assert k in d["hbm"]["timed_region"], k
            assert 0 < d["hbm"]["timed_region"][k]["frac_of_hbm_peak"] < 1
    assert r["traffic"] is None or isinstance(r["traffic"], (int, float))
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1


def test_bench_gpus_flag_without_launcher_spawns_ranks_or_fails_loudly(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset must start N ranks (never silently run one); with a launcher whose
    WORLD_SIZE disagrees it must exit non-zero.  No GPU is touched: the spawn is intercepted."""
    import sys
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: calls.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert calls == [4] and e.value.code == 0
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code not in (0, None) and calls == [4]


def test_rate_plan_is_the_references_rule_for_both_entry_points():
    """testing/blind_bwe_tester.py:410 and testing/denoise_and_bwe_tester.py:279-289: without a denoiser fs -> model rate; with
    one, both conversions hang on fs != sample_rate_denoiser (a file at the denoiser's rate goes to the model unconverted)."""
    from babe_amd.testing.long_file import rate_plan
    assert rate_plan(44100, 44100) == []
    assert rate_plan(48000, 44100) == [(48000, 44100)]
    assert rate_plan(48000, 16000, 22050) == [(48000, 22050), "denoise", (22050, 16000)]
    assert rate_plan(22050, 16000, 22050) == ["denoise"]                       # the reference's quirk, kept
    assert rate_plan(22050, 22050, 22050) == ["denoise"]
    assert rate_plan(16000, 16000, 22050) == [(16000, 22050), "denoise", (22050, 16000)]
