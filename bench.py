#!/usr/bin/env python3
"""Benchmark of the blind-BWE guided reverse-diffusion hot path on MI355X.

Metric (BASELINE.json): audio-seconds restored per wall-second, 10 s @ 44.1 kHz clips, 35 EDM
steps (2nd order, 69 score evaluations), blind low-pass estimation; whole-job aggregate over
all ranks.  One "step" = ONE 10 s clip per GPU through BlindSampler.predict_blind_bwe: the clip is
cut into two 368368-sample segments (the reference's model length, conf/exp/maestro44k_8s.yaml:52;
segmentation and cross-fade as formal_test_bwe, testing/blind_bwe_tester.py:421-566, restated in
babe_amd/testing/long_file.py) that run as one batch of 2 with per-clip semantics.  Weak scaling: every rank restores its own clips; one RCCL all_gather of
the restored audio + filters closes each step.

Usage:  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 44100
SEG = 368368            # exp.audio_len of the 44.1 kHz model
CLIP_SEC = 10.0
CLIP = int(CLIP_SEC * FS)
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def synth_clip(clip_id, n=CLIP, fs=FS):
    """Piano-like synthetic clip: decaying harmonic partials on a random note sequence."""
    g = torch.Generator().manual_seed(1000 + clip_id)
    t = torch.arange(n, dtype=torch.float64) / fs
    x = torch.zeros(n, dtype=torch.float64)
    onset = 0.0
    while onset < n / fs:
        midi = int(torch.randint(40, 88, (1,), generator=g))
        f0 = 440.0 * 2 ** ((midi - 69) / 12)
        dur = 0.25 + 1.5 * float(torch.rand(1, generator=g))
        npart = int(torch.randint(8, 17, (1,), generator=g))
        tt = (t - onset).clamp(min=0)
        gate = (t >= onset).double()
        for k in range(1, npart + 1):
            fk = f0 * k * math.sqrt(1 + 4e-4 * k * k)
            if fk > fs / 2 * 0.98:
                break
            x += gate * (1.0 / k) * torch.exp(-tt * (1.5 + 0.6 * k)) * torch.sin(2 * math.pi * fk * tt)
        onset += dur * 0.5
    x = x + 1e-3 * torch.randn(n, generator=g, dtype=torch.float64)
    return x.float()


def cpu_baseline(threads):
    """Oracle (CPU restatement of the reference) timed on one score evaluation of a 1/8-length segment of the
    same workload: fs=44100, L=46046, full-width network. Cost per audio-second is identical to the
    L=368368 segment (work is proportional to the number of CQT frames, i.e. to L).  Bounded to <=16
    threads: with hundreds of threads the many small autograd ops of the reference path get slower."""
    from oracle import edm as E
    from oracle import unet as UN
    from oracle.nsgt import CQT_nsgt as OracleCQT
    from oracle.sampler import OracleBlindSampler
    from babe_amd.networks.cqtdiff_plus import init_state_dict
    torch.set_num_threads(threads)
    L = 46046
    Ns, nd = [64, 96, 96, 128, 128, 256, 256], [2, 3, 4, 5, 6, 7, 7]
    sd = init_state_dict(Ns, nd, seed=0, gate_scale=1.0)
    cqt = OracleCQT(7, 64, "oct", ("kaiser", 1), FS, L)
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=nd)
    net = lambda x, cn: UN.unet_forward(sd, cfg, cqt, x, cn)
    smp = OracleBlindSampler(net, cqt, E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10), fs=FS, audio_len=L, T=35)
    g = torch.Generator().manual_seed(0)
    y = 0.1 * torch.randn(1, L, generator=g)
    x = y + 0.2 * torch.randn(1, L, generator=g)
    params = torch.tensor([list(smp.fc_init), list(smp.A_init)], dtype=torch.float32)
    t0 = time.time()
    smp.evaluate(x, torch.tensor(0.2), y, params, blind=True)
    dt = time.time() - t0
    evals_per_segment = 69
    value = (L / FS) / (evals_per_segment * dt)
    return {"value": value, "unit": "audio-sec/s", "cores": threads, "kind": "port",
            "sample": f"1 of 69 score evaluations (UNet fwd+input-VJP, filter fit, guidance) of a {L}-sample "
                      f"segment at 44.1 kHz (1/8 of the 368368-sample segment, same cost per audio-second), "
                      f"full-width network, {dt:.1f} s on {threads} threads, extrapolated x69"}


def conv_traffic(precision):
    """HBM-side bytes per conv launch from the committed PMC passes (rocprofv3 cannot run inside the bench):
    profiles/r01_conv_traffic.json, FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE averaged over the same launches."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_conv_traffic.json")
    if precision != "f32" or not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return {"bytes_per_launch": round(t["bytes_per_launch"]), "fetch": round(t["fetch_bytes_per_launch"]),
            "write": round(t["write_bytes_per_launch"]), "unit": "bytes per launch",
            "source": "profiles/r01_conv_traffic.json (PMC passes, not live)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=35, help="EDM steps (35 = the benchmark; anything else is a debug run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="10 s clips restored per GPU per step (1 = the benchmark's single-clip workload; larger values "
                         "batch more independent segments per kernel launch, cf. configs[2])")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"],
                    help="conv arithmetic: f32 = exact fp32 MFMA (the benchmark's dtype); bf16x3 / bf16 = bf16 MFMA with "
                         "fp32 storage+accumulation (reported with their own dtype string, never as f32)")
    ap.add_argument("--profile-convs", type=int, default=1)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % max(ndev, 1)          # (== local_rank on a real N-GPU node; lets 2 ranks share 1 GPU in tests)
    torch.cuda.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BABE_DIST_BACKEND", "nccl")      # "gloo" only for the shared-GPU functional test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    ge.build()
    from babe_amd._lib import lib
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.dist import gather_results
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from babe_amd.stft import STFTOps
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.long_file import assemble, cut_segments, plan_segments

    args = default_args(sample_rate=FS, audio_len=SEG, T=a.T)
    net = Unet_CQT_oct_with_attention(args, dev, precision=a.precision)
    net.load_state_dict(init_state_dict(args.network.Ns, args.network.num_dils, seed=0, gate_scale=1.0))
    sampler = BlindSampler(net, EDM(args), args, batch_semantics="per_clip", noise_device="cuda")
    st = STFTOps(4096, SEG, FS, dev)
    Hlp = st.design_filter(torch.tensor([[10000.0], [-60.0]], device=dev))

    plan = plan_segments(CLIP, SEG)                           # a 10 s clip = 2 segments (reference segmentation)
    nseg = len(plan)
    C_ = a.clips_per_gpu

    def make_inputs(step):
        ys = []
        for c in range(C_):
            clip_id = (step * world + rank) * C_ + c
            x = synth_clip(clip_id).to(dev)
            x = x * (0.1 / x.std())
            ys.append(st.apply_filter(cut_segments(x, SEG, plan), Hlp))   # "22.05 kHz content": nothing above ~11 kHz
        return torch.cat(ys, 0).contiguous()

    torch.manual_seed(2000 + rank)
    torch.cuda.manual_seed(2000 + rank)
    nsteps = a.warmup + a.steps
    inputs = [make_inputs(s) for s in range(nsteps)]          # resident in HBM before the timed region

    def one_step(s):
        x, fp = sampler.predict_blind_bwe(inputs[s])
        clips = torch.stack([assemble(x[c * nseg:(c + 1) * nseg], plan, CLIP, SEG) for c in range(C_)])
        fpc = fp.reshape(C_, -1)
        return gather_results(clips, fpc) if world > 1 else (clips, fpc)

    L_ = lib()
    L_.babe_conv_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    # Warm-up.  The LAST warm-up step also measures the conv launches with every batch item on one stream (kernels
    # serialised, so a launch duration is that kernel alone); the timed region below runs batch items on two streams and
    # its per-launch durations include the overlap with the other stream's kernels.
    serial = None
    for s in range(a.warmup):
        last = s == a.warmup - 1 and a.profile_convs and rank == 0
        if last:
            lanes_keep, net.MAX_LANES = net.MAX_LANES, 1
            torch.cuda.synchronize()
            L_.babe_conv_prof_enable(1)
        one_step(s)
        if last:
            torch.cuda.synchronize()
            ms0, fl0, nl0 = C.c_double(0), C.c_double(0), C.c_long(0)
            L_.babe_conv_prof_read(C.byref(ms0), C.byref(fl0), C.byref(nl0))
            L_.babe_conv_prof_enable(0)
            net.MAX_LANES = lanes_keep
            if ms0.value > 0:
                serial = (fl0.value / (ms0.value * 1e-3) / 1e12, ms0.value * 1e3 / max(nl0.value, 1))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if a.profile_convs:
        L_.babe_conv_prof_enable(1)
    t0 = time.perf_counter()
    for s in range(a.warmup, nsteps):
        out = one_step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, fl, nl = C.c_double(0), C.c_double(0), C.c_long(0)
    if a.profile_convs:
        L_.babe_conv_prof_read(C.byref(ms), C.byref(fl), C.byref(nl))
        L_.babe_conv_prof_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    finite = bool(torch.isfinite(out[0]).all())

    if rank == 0:
        value = world * a.steps * C_ * CLIP_SEC / dt
        roof = None
        dtype = {"f32": "f32", "bf16x3": "bf16x3 (bf16 MFMA on hi/lo-split operands, fp32 storage+accumulate)",
                 "bf16": "bf16 (bf16 MFMA, fp32 storage+accumulate)"}[a.precision]
        peak = PEAK_FP32_MFMA_TFLOPS if a.precision == "f32" else 2500.0
        kname = ("babe_conv2d launches of the UNet: conv_wino4_kernel (Winograd F(4,3)-along-time, fp32 "
                 "v_mfma_f32_32x32x2_f32; F(2,3) / direct fallbacks) for the (5,3) layers, conv_mfma_kernel for (1,1); fwd + "
                 "input-VJP; achieved counts the ALGORITHMIC (direct-convolution) flops - the matrix pipe executes half of "
                 "them on the (5,3) layers"
                 if a.precision == "f32" else
                 "conv_bf16_kernel (v_mfma_f32_32x32x16_bf16; %s products per k-block; achieved counts ALGORITHMIC flops)"
                 % ("3" if a.precision == "bf16x3" else "1"))
        if a.profile_convs and ms.value > 0:
            tr = conv_traffic(a.precision)
            ach = fl.value / (ms.value * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,
                    "kernel": kname,
                    "launches": nl.value, "avg_launch_us": round(ms.value * 1e3 / max(nl.value, 1), 2),
                    "algorithmic_tflop_per_launch_avg": round(fl.value / max(nl.value, 1) / 1e12, 5),
                    "kernel_time_share_of_step": round(ms.value * 1e-3 / dt, 4)}
            roof["concurrency"] = ("batch items run on %d streams: launch durations in the timed region include overlap with "
                                   "the other stream's kernels (sum of durations / wall = kernel_time_share_of_step)"
                                   % max(1, min(net.MAX_LANES, nseg * C_)))
            if serial is not None:
                roof["serial_achieved"] = round(serial[0], 2)          # same launches, one stream (last warm-up step)
                roof["serial_frac"] = round(serial[0] / peak, 4)
                roof["serial_avg_launch_us"] = round(serial[1], 2)
            if a.precision == "f32":
                # 96.6 % of the conv flops of this workload are (5,3) layers (3.897 of 4.034 TFLOP per forward), all
                # of which qualify for the F(4,3) kernel here; it multiplies half as often as the direct convolution
                ex = (serial[0] if serial is not None else ach) * (0.966 * 0.5 + 0.034)
                roof["executed_mfma_tflops_estimate"] = round(ex, 2)
                roof["executed_frac_of_peak_estimate"] = round(ex / peak, 4)
        rec = {
            "metric": "audio-sec/s (blind BWE, 10 s @ 44.1 kHz clips, 35 EDM steps 2nd order), whole job",
            "value": round(value, 5), "unit": "audio-sec/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: one 10 s 44.1 kHz clip per GPU per step = 2 segments x 368368 samples, "
                                   "blind LPF estimation, T=%d EDM steps (order 2, %d score evaluations), %s, "
                                   "CQTDiff+ Ns=[64,96,96,128,128,256,256], random-init weights" % (a.T, 2 * a.T - 1, a.precision),
                       "segments_per_clip": nseg, "clips_per_gpu_per_step": C_, "segment_len": SEG, "sample_rate": FS, "T": a.T,
                       "parallelism": "clips sharded over %d GPU(s), RCCL all_gather at end of step" % world,
                       "headline": a.T == 35 and a.precision == "f32" and C_ == 1},
            "per_gpu_realtime_factor": round(value / world, 5),
            "output_finite": finite,
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
