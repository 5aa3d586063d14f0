#!/usr/bin/env python3
"""Benchmark of the blind-BWE guided reverse-diffusion hot path on MI355X.

Metric (BASELINE.json): audio-seconds restored per wall-second, 10 s @ 44.1 kHz clips, 35 EDM
steps (2nd order, 69 score evaluations), blind low-pass estimation; whole-job aggregate over
all ranks.  One "step" = ONE 10 s clip per GPU through BlindSampler.predict_blind_bwe: the clip is
cut into two 368368-sample segments (the reference's model length, conf/exp/maestro44k_8s.yaml:52;
segmentation and cross-fade as formal_test_bwe, testing/blind_bwe_tester.py:421-566, restated in
babe_amd/testing/long_file.py) that run as one batch of 2 with per-clip semantics.  Weak scaling: every rank restores its own clips; one RCCL all_gather of
the restored audio + filters closes each step.

Usage:  python bench.py --gpus N --steps K --warmup W
N>1: either launched by torch.distributed.run (one rank per GPU, RANK/LOCAL_RANK/WORLD_SIZE in the env), or run plainly,
in which case this process spawns the N ranks itself (as a child torch.distributed.run, before touching the GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
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


def physical_cores():
    """Physical core count of the host (unique (physical id, core id) pairs of /proc/cpuinfo); None if unknown."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        return len(pairs) or None
    except OSError:
        return None


def cpu_baseline():
    """SURVEY 8d protocol.  The oracle (CPU restatement of the reference, pinned at FULL width against the imported
    reference by tests/golden/unet_full_46046.npz) runs T=2 of the sampler = 3 score evaluations after one warm-up
    evaluation, on a 46046-sample segment at 44.1 kHz (1/8 of the 368368-sample segment: the work is proportional to
    the number of CQT frames, i.e. identical cost per audio-second), full-width network; extrapolated x69/3.  Three thread
    counts: ALL PHYSICAL CORES of the host (count stated; what 8d asks for), 16 (the fastest setting measured on the 128-core
    hosts of this pool: the many small ops of the reference path do not scale past a few tens of threads - the all-cores
    leg is the number that shows it) and 8 (comparable with the survey's numbers), each with the per-component split.
    The headline `value` is ONE WARMED evaluation at the benchmark's own geometry (368368 samples, no extrapolation in length),
    timed at the best short-sample thread count and at all physical cores (best of the two), x69; the short-sample legs and
    their x8 extrapolation are reported beside it."""
    from oracle import edm as E
    from oracle import unet as UN
    from oracle.nsgt import CQT_nsgt as OracleCQT
    from oracle.sampler import OracleBlindSampler
    from babe_amd.networks.cqtdiff_plus import init_state_dict
    Ns, nd = [64, 96, 96, 128, 128, 256, 256], [2, 3, 4, 5, 6, 7, 7]
    sd = init_state_dict(Ns, nd, seed=0, gate_scale=1.0)
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=nd)

    def setup(L):
        cqt = OracleCQT(7, 64, "oct", ("kaiser", 1), FS, L)
        net = lambda x, cn: UN.unet_forward(sd, cfg, cqt, x, cn)
        smp = OracleBlindSampler(net, cqt, E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10), fs=FS, audio_len=L, T=2)
        g = torch.Generator().manual_seed(0)
        y = 0.1 * torch.randn(1, L, generator=g)
        x = y + 0.2 * torch.randn(1, L, generator=g)
        return smp, x, y, E.schedule(smp.p, 2, 0.2)

    L = 46046
    smp, x, y, sched = setup(L)

    def run(threads):
        torch.set_num_threads(threads)
        params = torch.tensor([list(smp.fc_init), list(smp.A_init)], dtype=torch.float32)
        t0 = time.perf_counter()
        smp.evaluate(x, sched[0], y, params, blind=True)                    # warm-up evaluation (untimed)
        # (a leg whose warm-up evaluation already took seconds - hundreds of threads on small ops - times ONE evaluation: the
        # default bench run has to finish within minutes)
        evals = (sched[0], sched[1], sched[1]) if time.perf_counter() - t0 < 3.0 else (sched[0],)
        timers = {}
        t0 = time.perf_counter()
        for tt in evals:                                                    # the 3 evaluations of a T=2 run
            _, _, params = smp.evaluate(x, tt, y, params, blind=True, timers=timers)
        n = len(evals)
        return (time.perf_counter() - t0) / n, {k: round(v / n, 4) for k, v in timers.items()}

    logical = os.cpu_count() or 1
    phys = physical_cores() or logical
    legs = {}
    for n in sorted({min(phys, logical), min(16, logical), min(8, logical)}):
        dt, split = run(n)
        legs[n] = {"threads": n, "seconds_per_evaluation": round(dt, 4), "split_seconds_per_evaluation": split,
                   "value": (L / FS) / (69 * dt)}
    best = max(legs, key=lambda n: legs[n]["value"])
    dt = legs[best]["seconds_per_evaluation"]
    out = {"value": legs[best]["value"], "unit": "audio-sec/s", "cores": best, "kind": "port",
           "physical_cores": physical_cores(), "logical_cpus": logical,
           "seconds_per_evaluation": dt, "split_seconds_per_evaluation": legs[best]["split_seconds_per_evaluation"],
           "sample": f"T=2 (3 score evaluations after 1 warm-up evaluation: UNet fwd + input-VJP, filter fit, filter apply) "
                     f"of a {L}-sample segment at 44.1 kHz (1/8 of the 368368-sample segment, same cost per audio-second), "
                     f"full-width network, {3 * dt:.1f} s on {best} threads (the best of {sorted(legs)} threads; all-physical-cores "
                     f"leg = {min(phys, logical)} threads), extrapolated x69/3",
           "legs": [legs[n] for n in sorted(legs)]}
    if 8 in legs and best != 8:
        out["value_8_threads"] = legs[8]["value"]
    pc = min(phys, logical)
    if pc in legs:
        out["value_all_physical_cores"] = legs[pc]["value"]
    # The benchmark's own segment length (no length extrapolation): ONE untimed warm-up evaluation (first touch of the ~26 GB
    # working set, lazy tables), then one timed evaluation at each of two thread counts - the best short-sample count and all
    # physical cores (the thread choice is re-made at full length, not inherited from the 46046-sample legs); best one counts.
    try:
        smpF, xF, yF, schedF = setup(SEG)
        params0 = torch.tensor([list(smpF.fc_init), list(smpF.A_init)], dtype=torch.float32)
        # second thread count: all physical cores unless the short-sample leg already showed them >= 3x slower than the best
        # count (128-core hosts of this pool: 10x - a 200 s leg would not fit the default run), then twice the best count
        pcn = min(phys, logical)
        second = pcn if legs[pcn]["seconds_per_evaluation"] < 3.0 * dt else min(2 * best, pcn)
        cand = sorted({best, second})
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        smpF.evaluate(xF, schedF[0], yF, params0.clone(), blind=True)       # warm-up (untimed)
        warm_s = time.perf_counter() - t0
        full_legs = []
        for n in cand:
            if n != best and warm_s > 60.0:
                continue                                                    # (a slow host: keep the default run within minutes)
            torch.set_num_threads(n)
            timers = {}
            t0 = time.perf_counter()
            smpF.evaluate(xF, schedF[0], yF, params0.clone(), blind=True, timers=timers)
            dtn = time.perf_counter() - t0
            full_legs.append({"threads": n, "seconds": round(dtn, 3), "split_seconds": {k: round(v, 4) for k, v in timers.items()},
                              "value": (SEG / FS) / (69 * dtn)})
        bl = max(full_legs, key=lambda e: e["value"])
        out["full_segment_evaluation"] = dict(bl, segment_len=SEG, warmup_evaluation_seconds=round(warm_s, 3), warmed=True,
                                              legs=full_legs, ratio_to_8x_short_sample=round(bl["seconds"] / (8 * dt), 3))
    except Exception as e:                                                   # (host memory: ~26 GB)  noqa: BLE001
        out["full_segment_evaluation"] = {"error": f"{type(e).__name__}: {e}"}
    # The headline `value` is the MEASURED, WARMED full-geometry evaluation (one score evaluation of a 368368-sample segment,
    # x69) at its best thread count; the short-sample legs and their x8 length extrapolation stay beside it.
    fse = out["full_segment_evaluation"]
    out["value_extrapolated_from_short_sample"] = out["value"]
    if "value" in fse:
        out["value"] = fse["value"]
        out["cores"] = fse["threads"]
        out["seconds_per_evaluation"] = fse["seconds"]
        out["split_seconds_per_evaluation"] = fse["split_seconds"]
        out["sample"] = (f"ONE score evaluation (UNet fwd + input-VJP, filter fit, filter apply) of a full {SEG}-sample segment at "
                         f"44.1 kHz, full-width network, WARMED (one untimed evaluation first, {fse['warmup_evaluation_seconds']:.1f} s), "
                         f"{fse['seconds']:.1f} s on {fse['threads']} threads = the best of {[e['threads'] for e in fse['legs']]} threads "
                         f"timed at full length, x69 evaluations per segment; the short-sample legs {sorted(legs)} (46046-sample "
                         f"segment, T=2) are beside this value under `legs`; their x8 length extrapolation is "
                         f"{fse['ratio_to_8x_short_sample']}x of this measurement")
    return out


PMC_TRAFFIC_FILE = "r06_conv_traffic.json"      # falls back to the newest committed file



def conv_traffic(precision):
    """HBM-side bytes per launch of the dominant conv kernel (conv_wino85_kernel / conv_wino85s_kernel) from the committed PMC passes (rocprofv3 cannot run inside the bench; the
    passes are separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this command, profiles/README.md)."""
    global PMC_TRAFFIC_FILE
    path = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
    if not os.path.exists(path):
        import glob
        older = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_conv_traffic.json")))
        if older:
            path, PMC_TRAFFIC_FILE = older[-1], os.path.basename(older[-1])
    if precision != "f32" or not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return {"bytes_per_launch": round(t["bytes_per_launch"]), "fetch": round(t["fetch_bytes_per_launch"]),
            "write": round(t["write_bytes_per_launch"]), "unit": "bytes per launch",
            "source": f"profiles/{PMC_TRAFFIC_FILE} (separate PMC passes of this command, not live)"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks under torch.distributed.run as a CHILD process
    (this parent never touches the GPU) and relay their output and exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
HBM_SLOTS = ["conv11", "gn_stats", "scale_gelu", "gn_bwd_partial", "gn_bwd_apply", "resample", "axpby", "cqt_band_analysis",
             "cqt_band_synthesis", "cqt_gather", "stft_fwd", "istft", "mag_stats", "sampler"]
CONV_SLOTS = ["conv53_wino85", "conv53_wino45", "conv53_wino4", "conv53_wino2", "conv53_direct", "conv53_fewco", "conv11", "conv_bf16", "conv_bf16p"]


def slot_table(prof, names, wall_s=None):
    out = {}
    for k in names:
        r = prof.get(k)
        if not r or not r["launches"]:
            continue
        sec = r["ms"] * 1e-3
        e = {"launches": r["launches"], "avg_us": round(r["ms"] * 1e3 / r["launches"], 2),
             "GB_per_s": round(r["bytes"] / sec / 1e9, 1), "frac_of_hbm_peak": round(r["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4),
             "MB_per_launch": round(r["bytes"] / r["launches"] / 1e6, 3)}
        if r["flops"]:
            e["TFLOP_per_s"] = round(r["flops"] / sec / 1e12, 2)
        if wall_s:
            e["share_of_step"] = round(sec / wall_s, 4)
        out[k] = e
    return out


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
    ap.add_argument("--dry-run", action="store_true",
                    help="print the rank -> device -> clip-range table of this command (JSON) and exit: no GPU, no process group")
    ap.add_argument("--no-eval-c", action="store_true",
                    help="the Python kernel sequencer (~1200 C-ABI calls per evaluation) instead of ONE library call per score "
                         "evaluation (babe_score_eval on the UNet / CQT plans, the default; bit-identical results): what the "
                         "sequencing itself costs")
    ap.add_argument("--no-pin", action="store_true", help="do not pin this rank's host threads to its own block of CPUs")
    ap.add_argument("--profile-steps", type=int, default=1,
                    help="timed steps whose launches are bracketed by HIP events (0 = no roofline/hbm blocks)")
    a = ap.parse_args()

    if a.dry_run:
        # the plan only: which rank drives which device and restores which clips of one step (block partition, babe_amd/dist.py);
        # nothing below touches a GPU or opens a process group
        from babe_amd.dist import rank_table
        from babe_amd.testing.long_file import plan_segments
        n_clips = a.gpus * a.clips_per_gpu
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        per = ncpu // a.gpus
        rows = [{"rank": r, "device": f"cuda:{d}", "clips": [lo, hi], "n_clips": hi - lo,
                 "segments": (hi - lo) * len(plan_segments(CLIP, SEG)),
                 "host_cpus": (f"{r * per}..{(r + 1) * per - 1} of the {ncpu} this process may use" if per >= 2 and not a.no_pin else "unpinned (fewer than 2 CPUs per rank on this host)")}
                for r, d, lo, hi in rank_table(n_clips, a.gpus, a.gpus)]
        print(json.dumps({"dry_run": True, "n_gpus": a.gpus, "clips_per_step": n_clips, "partition": "static block (babe_amd.dist.shard_range)",
                          "collective": "one all_gather_into_tensor of restored audio + filters per step (RCCL); none per EDM step",
                          "launch": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P "
                                    "bench.py --gpus %d ...  (or plain `python bench.py --gpus %d`, which spawns exactly that as a child)"
                                    % (a.gpus, a.gpus, a.gpus), "ranks": rows}, indent=1))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))                  # before anything touches the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                 f"--nproc-per-node {a.gpus} (or without a launcher, which spawns the ranks itself)")
    import torch.distributed as dist
    pinned = None
    if world > 1 and not a.no_pin:
        # one block of host CPUs per rank, before the first GPU call: 8 Python enqueue loops must not migrate over each other
        from babe_amd.dist import pin_host_threads
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        pinned = pin_host_threads(local_rank % lw, lw)
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % max(ndev, 1)          # (== local_rank on a real N-GPU node; lets 2 ranks share 1 GPU in tests)
    torch.cuda.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    # A process group exists whenever a launcher started this rank (torch.distributed.run sets WORLD_SIZE) - also for ONE rank,
    # so that `torchrun --nproc-per-node 1 bench.py --gpus 1` drives the RCCL init + all_gather_into_tensor branch on a one-GPU
    # box (tests/test_gpu_dist.py); a plain `python bench.py` (the driver's N = 1 command) creates none.
    launched = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BABE_DIST_BACKEND", "nccl")      # "gloo" only for the shared-GPU functional test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if a.no_eval_c:
        os.environ["BABE_EVAL_C"] = "0"                    # (read when babe_amd.testing.blind_bwe_sampler is imported, below)
    import __graft_entry__ as ge
    ge.build()
    from babe_amd import _lib
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
        return gather_results(clips, fpc, force_collective=True) if (world > 1 or launched) else (clips, fpc)

    do_prof = a.profile_steps > 0 and rank == 0
    # Warm-up.  The LAST warm-up step also measures every launch with all batch items on ONE stream (kernels serialised,
    # so a launch duration is that kernel alone - the figure a rocprofv3 kernel trace of this command agrees with, since
    # tracing serialises the lanes too); the timed region below runs batch items on two streams and its per-launch
    # durations include the overlap with the other stream's kernels.
    serial = None
    for s in range(a.warmup):
        last = s == a.warmup - 1 and do_prof
        if last:
            lanes_keep, net.MAX_LANES = net.MAX_LANES, 1
            slanes_keep, sampler.LANES = sampler.LANES, 1
            torch.cuda.synchronize()
            _lib.prof_read()
            _lib.prof_enable(True)
            t_ser = time.perf_counter()
        one_step(s)
        if last:
            torch.cuda.synchronize()
            t_ser = time.perf_counter() - t_ser
            _lib.prof_enable(False)
            serial = _lib.prof_read()
            net.MAX_LANES = lanes_keep
            sampler.LANES = slanes_keep
    if world > 1 or launched:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.dispatch_counts(reset=True)
    if do_prof:
        _lib.prof_read()
        _lib.prof_enable(True)
    n_prof = min(a.profile_steps, a.steps) if do_prof else 0
    t0 = time.perf_counter()
    for i, s in enumerate(range(a.warmup, nsteps)):
        out = one_step(s)
        if do_prof and i + 1 == n_prof:
            _lib.prof_enable(False)                 # host-side switch only: no sync inside the timed region
    torch.cuda.synchronize()
    if world > 1 or launched:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = _lib.prof_read() if do_prof else None
    counts = _lib.dispatch_counts()
    if world > 1 or launched:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    finite = bool(torch.isfinite(out[0]).all())

    if rank == 0:
        value = world * a.steps * C_ * CLIP_SEC / dt
        roof = hbm = None
        dtype = {"f32": "f32", "bf16x3": "bf16x3 (bf16 MFMA on hi/lo-split operands, fp32 storage+accumulate)",
                 "bf16": "bf16 (bf16 MFMA, fp32 storage+accumulate)"}[a.precision]
        peak = PEAK_FP32_MFMA_TFLOPS if a.precision == "f32" else 2500.0
        dom = "conv53_wino4" if a.precision == "f32" else "conv_bf16"
        if a.precision == "f32" and timed is not None:
            dom = max(("conv53_wino85", "conv53_wino45", "conv53_wino4"), key=lambda k: timed[k]["ms"])   # the (5,3) kernel with the most time
        if a.precision == "bf16" and timed is not None and timed["conv_bf16p"]["launches"]:
            dom = "conv_bf16p"                           # the pipelined kernel takes the (5,3) layers of the bf16 build
        if timed is not None and timed[dom]["launches"]:
            wall_prof = dt * n_prof / a.steps            # wall time of the profiled steps (steps are identical work)
            r = timed[dom]
            sec = r["ms"] * 1e-3
            conv_all = {k: timed[k] for k in CONV_SLOTS}
            sum_conv_ms = sum(v["ms"] for v in conv_all.values())
            sum_all_ms = sum(v["ms"] for v in timed.values())
            tr = conv_traffic(a.precision)
            roof = {
                "bound": "mfma",
                "kernel": (("conv_wino85_kernel / conv_wino85s_kernel (128- / 96- and 64-channel tiles of the same algorithm; one "
                            "measurement slot): nested Winograd F(4,5) along frequency x F(4,3) along time, fp32 "
                            "v_mfma_f32_16x16x4_f32 (executes 0.2 of the algorithmic flops), fwd + input-VJP launches of the "
                            "(5,3) layers of the UNet with >= 64 channels whose row quads are >= 80 % full (all_conv_kernels "
                            "has the rest)" if dom == "conv53_wino85" else
                            "conv_wino45x_kernel / conv_wino45_kernel (128- / 96- and 64-channel tiles of the same algorithm; one "
                            "measurement slot): nested Winograd F(2,5) along frequency x F(4,3) along time, fp32 "
                            "v_mfma_f32_16x16x4_f32 (executes 0.3 of the algorithmic flops), fwd + input-VJP launches of every "
                            "(5,3) layer of the UNet with >= 64 channels (all_conv_kernels has the rest)" if dom == "conv53_wino45" else
                            "conv_wino4p_kernel: pipelined Winograd F(4,3)-along-time (5,3) conv, fp32 v_mfma_f32_32x32x2_f32, "
                            "fwd + input-VJP launches of the UNet") if a.precision == "f32" else
                           "%s (v_mfma_f32_32x32x16_bf16; %s products per k-block)"
                           % ("conv_bf16p_kernel" if dom == "conv_bf16p" else "conv_bf16_kernel",
                              "3" if a.precision == "bf16x3" else "1")),
                "achieved": round(r["flops"] / sec / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(r["exec_flops"] / sec / 1e12 / peak, 4),
                "frac_definition": "EXECUTED MFMA flops / duration / peak (F(4,3) executes 1/2, the nested F(2,5)xF(4,3) "
                                   "kernel 3/10, the nested F(4,5)xF(4,3) kernel 2/10 of the algorithmic direct-convolution "
                                   "flops that `achieved` counts: a kernel that executes FEWER flops for the same outputs is "
                                   "faster at a LOWER frac - rounds 3-5 read 0.59-0.60 on the F(2,5) kernel at 310 algorithmic "
                                   "TFLOP/s); algorithmic_frac = achieved / peak.  "
                                   "`achieved` / `frac` / `avg_launch_us` are the kernel ALONE on the GPU (the `serial` "
                                   "block: all batch items on one stream, what the rocprofv3 summary under profiles/ "
                                   "reproduces) when that block exists; the same launches inside the two-lane timed region, "
                                   "whose durations include the other lane's kernels, are under `timed_region`",
                "algorithmic_frac": round(r["flops"] / sec / 1e12 / peak, 4),
                "executed_tflops": round(r["exec_flops"] / sec / 1e12, 2),
                "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,
                "algorithmic_MB_per_launch": round(r["bytes"] / r["launches"] / 1e6, 2),
                "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                "algorithmic_gflop_per_launch_avg": round(r["flops"] / r["launches"] / 1e9, 3),
                "profiled_steps": n_prof,
                "timing": "HIP events on each launch's own stream over the first %d step(s) of the timed region; the clips' "
                          "whole evaluation chains run on %d streams (BlindSampler._sample_lanes), so these durations include "
                          "overlap with the other stream's kernels: overlap_factor = sum of all kernel durations / wall time "
                          "of those steps" % (n_prof, max(1, min(sampler.LANES, nseg * C_))),
                "overlap_factor": round(sum_all_ms * 1e-3 / wall_prof, 4),
                "conv_time_share_of_kernel_time": round(sum_conv_ms / sum_all_ms, 4),
                "conv_dispatch_counts_timed_region": {k: counts[k] for k in CONV_SLOTS + ["dft_stage", "gn_stats", "scale_gelu", "gn_bwd_partial", "gn_bwd_apply"]},
                "all_conv_kernels": slot_table(timed, CONV_SLOTS),
            }
            if serial is not None and serial[dom]["launches"]:
                q = serial[dom]
                qs = q["ms"] * 1e-3
                roof["serial"] = {
                    "what": "same launches with all batch items on ONE stream (last warm-up step): the kernel alone on the "
                            "GPU; this is what a rocprofv3 kernel trace (which serialises the lanes) agrees with",
                    "achieved": round(q["flops"] / qs / 1e12, 2), "frac": round(q["exec_flops"] / qs / 1e12 / peak, 4),
                    "algorithmic_frac": round(q["flops"] / qs / 1e12 / peak, 4),
                    "avg_launch_us": round(q["ms"] * 1e3 / q["launches"], 2), "launches": q["launches"],
                    "sum_kernel_time_over_wall": round(sum(v["ms"] for v in serial.values()) * 1e-3 / t_ser, 4),
                    "all_conv_kernels": slot_table(serial, CONV_SLOTS, t_ser)}
                # headline figures = the kernel alone (reproducible from profiles/); the overlapped ones move beside them.  EVERY
                # per-launch field moves with them, so that achieved = algorithmic_gflop_per_launch_avg / avg_launch_us holds
                # inside the line (the serial step launches each layer ONCE for the whole batch, the timed region once per lane)
                per_launch = ("achieved", "frac", "algorithmic_frac", "executed_tflops", "avg_launch_us", "launches",
                              "algorithmic_gflop_per_launch_avg", "algorithmic_MB_per_launch")
                roof["timed_region"] = {k: roof[k] for k in per_launch}
                sr = roof["serial"]
                if dom == "conv53_wino85":               # cross-round comparison only: the same outputs per second on the round-4 algorithm (3/10 executed)
                    roof["cross_round_comparison"] = {"frac_if_this_rate_executed_3_10": round(0.3 * q["flops"] / qs / 1e12 / peak, 4),
                                                      "note": "hypothetical, NOT an achieved utilisation: rounds 3-5a read 0.59-0.60 on the "
                                                              "F(2,5)xF(4,3) kernel, which executes 3/10 of the algorithmic flops"}
                roof.update(achieved=sr["achieved"], frac=sr["frac"], algorithmic_frac=sr["algorithmic_frac"],
                            executed_tflops=round(q["exec_flops"] / qs / 1e12, 2), avg_launch_us=sr["avg_launch_us"],
                            launches=q["launches"], algorithmic_gflop_per_launch_avg=round(q["flops"] / q["launches"] / 1e9, 3),
                            algorithmic_MB_per_launch=round(q["bytes"] / q["launches"] / 1e6, 2))
                if tr:
                    # the PMC passes trace one-segment launches; a serial launch carries the whole batch: same unit as `achieved`
                    nb = nseg * C_
                    roof["traffic"] = round(tr["bytes_per_launch"] * nb)
                    roof["traffic_over_algorithmic"] = round(roof["traffic"] / (q["bytes"] / q["launches"]), 3)
                    roof["traffic_detail"] = dict(tr, segments_per_launch_here=nb,
                                                  note="bytes_per_launch is per ONE-segment launch; `traffic` = x segments per launch")
            hbm = {"peak_GB_per_s": HBM_PEAK_GBS, "bytes": "ALGORITHMIC bytes per launch (each operand touched once), DESIGN.md 3",
                   "timed_region": slot_table(timed, HBM_SLOTS, wall_prof)}
            if serial is not None:
                hbm["serial"] = slot_table(serial, HBM_SLOTS, t_ser)
        rec = {
            "metric": "audio-sec/s (blind BWE, 10 s @ 44.1 kHz clips, 35 EDM steps 2nd order), whole job",
            "value": round(value, 5), "unit": "audio-sec/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": ("configs[1]: one 10 s 44.1 kHz clip per GPU per step = 2 segments x 368368 samples, "
                                    if C_ == 1 else
                                    "configs[2]-style batch: %d x 10 s 44.1 kHz clips per GPU per step (%d segments x 368368 "
                                    "samples, restored in sub-batches of %d segments), " % (C_, C_ * nseg, sampler.max_in_flight)) +
                                   "blind LPF estimation, T=%d EDM steps (order 2, %d score evaluations), %s, "
                                   "CQTDiff+ Ns=[64,96,96,128,128,256,256], random-init weights" % (a.T, 2 * a.T - 1, a.precision),
                       "segments_per_clip": nseg, "clips_per_gpu_per_step": C_, "segment_len": SEG, "sample_rate": FS, "T": a.T,
                       "parallelism": "clips sharded over %d GPU(s), one process per GPU, %s" % (
                           world, "no collective (single rank, no launcher)" if not (world > 1 or launched) else
                           "%s all_gather at end of step" % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend())),
                       "host_cpus_rank0": (f"pinned to {len(pinned)} CPUs ({pinned[0]}..{pinned[-1]})" if pinned else "unpinned"),
                       "hip_graphs": "none: eager launch loop, no host sync inside a step (the opt-in graph replay of rounds 2-4 "
                                     "measured 2.129 vs 2.140 audio-sec/s on this command and was removed in round 5)",
                       "sequencer": ("Python: ~1200 C-ABI calls per evaluation (--no-eval-c / BABE_EVAL_C=0)"
                                     if (a.no_eval_c or os.environ.get("BABE_EVAL_C", "1") != "1" or a.precision != "f32") else
                                     "library: one babe_score_eval call per evaluation from Python (the product default)"),
                       "headline": a.T == 35 and a.precision == "f32" and C_ == 1},
            "per_gpu_realtime_factor": round(value / world, 5),
            "output_finite": finite,
            "roofline": roof,
            "hbm": hbm,
        }
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if world > 1 or launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
