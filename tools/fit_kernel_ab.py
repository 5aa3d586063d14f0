#!/usr/bin/env python3
"""Round-6 review item 5: is the default filter-fit kernel (filter_fit_fast_kernel, babe_fit_cfg.kernel = 0) harmless ON THE SAMPLER,
or should the reference-order kernel (kernel = 1) be the default?  Runs the reference goldens `sampler_T35.npz` (T = 35, 69 fits)
and, when present, `sampler_full_368368.npz` (full width, L = 368368, T = 2) with both kernels and prints, per kernel: output RMS
error vs the imported reference, worst per-step x_den error, and the MASK FLIPS - STFT bins that land in a different filter segment
than in the reference's run (design_filter's `f >= fc` comparisons, utils/blind_bwe_utils.py:82-119) summed over the recorded steps -
plus the time of the fit inside an evaluation chain (HIP events, library hook)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from babe_amd import _lib                                                    # noqa: E402
from babe_amd.diff_params.edm import EDM                                     # noqa: E402
from babe_amd.testing.blind_bwe_sampler import BlindSampler                  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
load = lambda n: {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, n)).items()}
rel = lambda a, b: float((a.detach().double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))
rms = lambda a, b: float((a.detach().double().cpu() - b.double()).pow(2).mean().sqrt())


def mask_flips(fp_run, fp_ref, fs, nfft=4096):
    """bins whose `f >= fc_k` verdict differs between the two parameter sets, summed over the break frequencies"""
    f = torch.fft.rfftfreq(nfft, d=1.0 / fs).double()
    a, b = fp_run.detach().double().cpu().reshape(2, -1)[0], fp_ref.double().reshape(2, -1)[0]
    return int(sum(((f >= x) != (f >= y)).sum() for x, y in zip(a, b)))


def run(s, net, args, L, kernel, pre_draws, T):
    gen = torch.Generator().manual_seed(int(s["seed"]))
    for _ in range(pre_draws):
        torch.randn(L, generator=gen)
    noises = [torch.randn(1, L, generator=gen) for _ in range(T + 1)]
    smp = BlindSampler(net, EDM(args), args)
    smp.fit_cfg.kernel = kernel
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    _lib.prof_read()
    _lib.prof_enable(True)
    x, fp, dden, t, dfil = smp.predict_blind_bwe(s["y"].cuda(), rid=True)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_read()
    fit = pr.get("filter_fit") or {"ms": float("nan"), "launches": 0}
    worst = max(rel(dden[i][:, ::16], s["data_denoised_sub16"][i]) for i in range(T))
    flips = sum(mask_flips(dfil[i], s["data_filters"][i], args.exp.sample_rate) for i in range(T))
    total_ms = sum(v["ms"] for v in pr.values())
    print(f"  kernel={kernel}: output RMS err {rms(x, s['x']):.3e} (rel {rel(x, s['x']):.3e}), worst per-step x_den rel {worst:.3e}, "
          f"mask flips over {T} recorded steps {flips}, final fc {fp.reshape(2, -1)[0].tolist()} (ref {s['filter_params'].reshape(2, -1)[0].tolist()}), "
          f"fit {fit['ms'] / max(fit['launches'], 1) * 1e3:.0f} us x {fit['launches']} launches = {fit['ms']:.1f} ms of {total_ms:.0f} ms kernel time", flush=True)
    return x


def main():
    from test_gpu_sampler import small_net
    s = load("sampler_T35.npz")
    g, args, net = small_net(T=35, start_sigma=float(s["start_sigma"]))
    print("sampler_T35.npz (reduced width, L = 92092, T = 35, mu = default [1000, 10]):")
    xs = [run(s, net, args, 92092, k, 1, 35) for k in (0, 1)]
    print(f"  kernel 0 vs kernel 1 outputs: RMS diff {rms(xs[0], xs[1].cpu()):.3e}")
    p = os.path.join(G, "sampler_full_368368.npz")
    if os.path.exists(p):
        from babe_amd.config import default_args
        from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention
        from golden_weights import full_width_sd
        s = load("sampler_full_368368.npz")
        L, T = int(s["L"]), int(s["T"])
        args = default_args(sample_rate=44100, audio_len=L, T=T, start_sigma=float(s["start_sigma"]))
        args.tester.blind_bwe.optimization.mu = [float(v) for v in s["mu"]]
        net = Unet_CQT_oct_with_attention(args, "cuda")
        net.load_state_dict(full_width_sd(int(s["wseed"])))
        print(f"sampler_full_368368.npz (full width, L = {L}, T = {T}, mu = {args.tester.blind_bwe.optimization.mu}):")
        xs = [run(s, net, args, L, k, 1, T) for k in (0, 1)]
        print(f"  kernel 0 vs kernel 1 outputs: RMS diff {rms(xs[0], xs[1].cpu()):.3e}")


if __name__ == "__main__":
    main()
