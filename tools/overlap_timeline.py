"""Where does the two-lane step's wall time go?  

A rocprofv3 kernel trace serialises the dispatches of different streams (profiles/README.md, round 1 note), so it cannot show
what runs beside what.  This tool uses the library's own measurement hook instead - HIP events around every launch on the
launch's own stream, read back as a timeline on the common device clock (babe_prof_timeline) - for
  (A) ONE segment alone on the GPU (B = 1, one stream): every kernel's duration with nothing beside it;
  (B) the benchmark's step: the two segments of a 10 s clip on two lanes.
The launch sequence of a segment is deterministic, so launch i of a lane in (B) is the same kernel on the same shapes as launch
i of (A): stretch = duration in (B) / duration in (A).  Reduced to
  * wall-time split by (what lane 0 runs, what lane 1 runs), idle included;
  * per kernel family: time alone (A), time in (B), stretch, and the stretch split by what the other lane ran meanwhile;
  * stream gaps (stream idle between two launches of a lane = host enqueue not keeping up / event overhead).
Usage: python3 tools/overlap_timeline.py [--T 35] > profiles/r05_overlap.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN                                        # noqa: E402  (synthetic clip, constants)

FAMILIES = {"conv53_wino85": "conv53", "conv53_wino45": "conv53", "conv53_wino4": "conv53", "conv53_wino2": "conv53", "conv53_direct": "conv53",
            "conv53_fewco": "conv_small", "conv11": "conv11", "conv_bf16": "conv53", "conv_bf16p": "conv53",
            "gn_stats": "gn_fwd", "scale_gelu": "gn_fwd", "gn_bwd_partial": "gn_vjp", "gn_bwd_apply": "gn_vjp",
            "resample": "ew", "axpby": "ew"}
ORDER = ["conv53", "conv11", "conv_small", "gn_fwd", "gn_vjp", "ew", "other", "idle"]


def fam_of(names, slot):
    return FAMILIES.get(names[slot], "other")


def run(sampler, y, names):
    from babe_amd import _lib
    torch.cuda.synchronize()
    _lib.prof_read()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    sampler.predict_blind_bwe(y)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    _lib.prof_enable(False)
    tl = _lib.prof_timeline()
    _lib.prof_read()
    tl["fam"] = np.array([fam_of(names, s) for s in tl["slot"]])
    return wall, tl


def coactivity(tl, nl):
    """sweep line: seconds spent in every (state of lane 0, state of lane 1) pair; state = family or idle"""
    pts = []
    for i in range(len(tl["t0_ms"])):
        pts.append((tl["t0_ms"][i], 1, int(tl["lane"][i]), tl["fam"][i]))
        pts.append((tl["t1_ms"][i], 0, int(tl["lane"][i]), tl["fam"][i]))
    pts.sort(key=lambda p: (p[0], p[1]))
    state = ["idle"] * nl
    acc = {}
    last = pts[0][0]
    for t, kind, ln, fam in pts:
        if t > last:
            key = tuple(state)
            acc[key] = acc.get(key, 0.0) + (t - last) * 1e-3
            last = t
        state[ln] = fam if kind == 1 else "idle"
    return acc, (pts[-1][0] - pts[0][0]) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=35)
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from babe_amd import _lib
    from babe_amd.config import default_args
    from babe_amd.diff_params.edm import EDM
    from babe_amd.networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from babe_amd.stft import STFTOps
    from babe_amd.testing.blind_bwe_sampler import BlindSampler
    from babe_amd.testing.long_file import cut_segments, plan_segments
    dev = torch.device("cuda", 0)
    args = default_args(sample_rate=BN.FS, audio_len=BN.SEG, T=a.T)
    net = Unet_CQT_oct_with_attention(args, dev)
    net.load_state_dict(init_state_dict(args.network.Ns, args.network.num_dils, seed=0, gate_scale=1.0))
    sampler = BlindSampler(net, EDM(args), args, batch_semantics="per_clip", noise_device="cuda")
    st = STFTOps(4096, BN.SEG, BN.FS, dev)
    Hlp = st.design_filter(torch.tensor([[10000.0], [-60.0]], device=dev))
    plan = plan_segments(BN.CLIP, BN.SEG)
    x = BN.synth_clip(0).to(dev)
    x = x * (0.1 / x.std())
    y = st.apply_filter(cut_segments(x, BN.SEG, plan), Hlp).contiguous()          # [2, SEG]
    names = _lib.prof_slot_names()
    torch.manual_seed(2000)
    torch.cuda.manual_seed(2000)
    sampler.predict_blind_bwe(y)                                                   # warm-up (tables, allocator)
    # (A) goes through the SAME lane code as (B) - one lane, so that launch i of (A) is launch i of each lane of (B)
    sampler._use_lanes = lambda *aa, **kk: True
    sampler.predict_blind_bwe(y[:1])
    torch.cuda.synchronize()

    wA, A = run(sampler, y[:1], names)                                             # (A) one segment alone
    del sampler._use_lanes
    wB, B = run(sampler, y, names)                                                 # (B) the benchmark's step, two lanes
    # streams of (B): the caller's stream (STFT of the observation, first noise, final cross-fade: a handful of launches) and
    # the two lane streams.  The analysis below is about the two lanes; the caller's records are reported and set aside.
    cnt = np.bincount(B["lane"])
    keep = np.argsort(-cnt)[:2]
    side = ~np.isin(B["lane"], keep)
    if side.any():
        print(f"# (B): {int(side.sum())} launches ({((B['t1_ms'] - B['t0_ms'])[side]).sum() * 1e-3:.4f} s) on the caller's stream set aside")
    remap = {int(k): i for i, k in enumerate(sorted(keep))}
    B = {k: v[~side] for k, v in B.items()}
    B["lane"] = np.array([remap[int(v)] for v in B["lane"]], np.int32)
    cntA = np.bincount(A["lane"])
    sideA = A["lane"] != int(np.argmax(cntA))
    A = {k: v[~sideA] for k, v in A.items()}
    nlB = 2
    print(f"# T = {a.T}; (A) one segment alone: wall {wA:.3f} s, {len(A['slot'])} launches;  (B) two lanes: wall {wB:.3f} s, "
          f"{len(B['slot'])} launches on {nlB} streams  ->  {10.0 / wB:.3f} audio-sec/s for this step (HIP events around every "
          f"launch cost a few %: the unprofiled step is faster)")
    dA = (A["t1_ms"] - A["t0_ms"]) * 1e-3
    dB = (B["t1_ms"] - B["t0_ms"]) * 1e-3
    # (A): kernel time by family, stream gaps
    order = np.argsort(A["t0_ms"], kind="stable")
    gapsA = np.clip(A["t0_ms"][order][1:] - A["t1_ms"][order][:-1], 0, None).sum() * 1e-3
    print(f"\n## (A) one segment alone (B = 1 launches, one stream): sum of kernel time {dA.sum():.3f} s, stream gaps {gapsA:.3f} s")
    print(f"{'family':12s} {'launches':>9s} {'seconds':>9s} {'share':>7s}")
    for f in ORDER[:-1]:
        m = A["fam"] == f
        if m.any():
            print(f"{f:12s} {int(m.sum()):9d} {dA[m].sum():9.3f} {dA[m].sum() / dA.sum():7.3f}")
    print(f"(two segments one after the other would take {2 * wA:.3f} s; the MFMA-bound families alone "
          f"{2 * dA[np.isin(A['fam'], ['conv53', 'conv11'])].sum():.3f} s)")

    # (B): co-activity
    acc, span = coactivity(B, nlB)
    print(f"\n## (B) wall-time split by what the two lanes run (span of the timeline {span:.3f} s)")
    tot = sum(acc.values())
    rows = {}
    for (s0, s1), v in acc.items():
        k = tuple(sorted((s0, s1), key=ORDER.index))
        rows[k] = rows.get(k, 0.0) + v
    print(f"{'lane x':12s} {'lane y':12s} {'seconds':>9s} {'share':>7s}")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]):
        if v / tot >= 0.002:
            print(f"{k[0]:12s} {k[1]:12s} {v:9.3f} {v / tot:7.3f}")
    both_mfma = sum(v for k, v in rows.items() if k[0] in ("conv53", "conv11") and k[1] in ("conv53", "conv11"))
    one_mfma = sum(v for k, v in rows.items() if (k[0] in ("conv53", "conv11")) != (k[1] in ("conv53", "conv11")))
    none_mfma = tot - both_mfma - one_mfma
    print(f"both lanes in an MFMA-bound kernel {both_mfma:.3f} s ({both_mfma / tot:.3f}), exactly one {one_mfma:.3f} s "
          f"({one_mfma / tot:.3f}), none {none_mfma:.3f} s ({none_mfma / tot:.3f});  a lane idle (stream gap) "
          f"{sum(v for k, v in rows.items() if 'idle' in k):.3f} s")

    # per-launch stretch: match launch i of each lane of (B) with launch i of (A)
    print("\n## stretch of the same launches: duration in (B) / duration alone in (A), by family and by what the other lane ran "
          "for most of the launch")
    idxA = np.argsort(A["t0_ms"], kind="stable")
    famA, durA = A["fam"][idxA], dA[idxA]
    lanes = []
    for ln in range(nlB):
        m = np.where(B["lane"] == ln)[0]
        m = m[np.argsort(B["t0_ms"][m], kind="stable")]
        lanes.append(m)
    ok = all(len(m) == len(idxA) and (B["fam"][m] == famA).all() for m in lanes)
    if not ok:
        print("(launch sequences of (A) and the lanes of (B) differ: no per-launch matching; lens", len(idxA), [len(m) for m in lanes], ")")
        return
    # what the OTHER lane ran during each launch: overlap-weighted majority family
    other_of = {}
    for ln in range(nlB):
        o = lanes[1 - ln] if nlB == 2 else lanes[ln]
        ot0, ot1, ofam = B["t0_ms"][o], B["t1_ms"][o], B["fam"][o]
        res = []
        j = 0
        for i in lanes[ln]:
            s, e = B["t0_ms"][i], B["t1_ms"][i]
            while j > 0 and ot1[j - 1] > s:
                j -= 1
            while j < len(o) and ot1[j] <= s:
                j += 1
            w = {}
            k = j
            while k < len(o) and ot0[k] < e:
                ov = min(e, ot1[k]) - max(s, ot0[k])
                if ov > 0:
                    w[ofam[k]] = w.get(ofam[k], 0.0) + ov
                k += 1
            idle = (e - s) - sum(w.values())
            if idle > 0:
                w["idle"] = idle
            res.append(max(w, key=w.get) if w else "idle")
        other_of[ln] = np.array(res)
    print(f"{'family':12s} {'other lane':12s} {'launches':>9s} {'alone s':>9s} {'in (B) s':>9s} {'stretch':>8s} {'extra s':>8s}")
    tot_extra = 0.0
    for f in ORDER[:-1]:
        fa = fb = 0.0
        lines = []
        for of in ORDER:
            al = bl = 0.0
            n = 0
            for ln in range(nlB):
                m = (famA == f) & (other_of[ln] == of)
                if m.any():
                    al += durA[m].sum()
                    bl += dB[lanes[ln]][m].sum()
                    n += int(m.sum())
            if n:
                lines.append(f"{f:12s} {of:12s} {n:9d} {al:9.3f} {bl:9.3f} {bl / max(al, 1e-12):8.2f} {bl - al:8.3f}")
                fa += al
                fb += bl
        if lines:
            print("\n".join(lines))
            print(f"{f:12s} {'(all)':12s} {'':9s} {fa:9.3f} {fb:9.3f} {fb / max(fa, 1e-12):8.2f} {fb - fa:8.3f}")
            tot_extra += fb - fa
    sumA2 = 2 * dA.sum()
    print(f"\nsum of alone durations of both segments {sumA2:.3f} s; sum of durations in (B) {dB.sum():.3f} s (extra {tot_extra:.3f} s); "
          f"wall (B) {wB:.3f} s = {dB.sum() / wB:.2f} kernels in flight on average; wall (B) / (sum alone / 1) = {wB / sumA2:.3f}")
    print("Reading: CU-time is additive on this GPU for this workload (a conv workgroup owns its CU: 2 x 248 registers per SIMD, "
          "144 KB LDS), so wall (B) ~ sum of alone durations minus what tails / stream gaps of one lane the other lane fills.")


if __name__ == "__main__":
    main()
