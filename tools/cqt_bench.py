"""HBM-roofline check of the CQT kernels (north star: >= 60 % of HBM bandwidth on the CQT kernel).
Algorithmic bytes per clip: analysis reads the 1.47 MB half spectrum and writes 4.16 MB of
coefficients (520192 complex); synthesis the reverse."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import _lib
from babe_amd.cqt import CQT_nsgt

L, fs = 368368, 44100
cq = CQT_nsgt(7, 64, "oct", ("kaiser", 1), fs, L, device="cuda")
ncoef = sum(64 * T for T in cq.T_oct)


def event_bracket_floor():
    """What the HIP-event bracket of the measurement hook reads for a kernel that does (almost) nothing: one 256-thread workgroup
    of babe_spec_scale.  The bracket [event, launch, event] contains the dispatch latency between the first event's timestamp and
    the kernel's start; a rocprofv3 kernel trace (tools/cqt_trace_summary.py) reports the kernel's own begin / end."""
    t = torch.zeros(1, 2, 256, device="cuda")
    m = torch.ones(256, device="cuda")
    from babe_amd._lib import check, lib, ptr, stream
    for _ in range(20):                                      # (warm: the first launches carry module loading)
        check(lib().babe_spec_scale(ptr(t), None, ptr(t), ptr(m), 256, 2, 1.0, 0.0, 1, stream()), "spec_scale")
    torch.cuda.synchronize()
    _lib.prof_read()
    _lib.prof_enable(True)
    for _ in range(50):
        check(lib().babe_spec_scale(ptr(t), None, ptr(t), ptr(m), 256, 2, 1.0, 0.0, 1, stream()), "spec_scale")
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_read()["cqt_gather"]
    return pr["ms"] * 1e3 / max(pr["launches"], 1)


print(f"# HIP-event bracket of a one-workgroup kernel (the floor every [GPU time, HIP events] figure below contains): {event_bracket_floor():.1f} us; "
      f"kernel-only durations: tools/cqt_trace_summary.py over a rocprofv3 --kernel-trace of this script")
for B in [int(v) for v in os.environ.get("BS", "1,2,8,32,64").split(",")]:
    x = torch.randn(B, L, device="cuda")
    spec = cq.fft.rfft(x)
    coefs = cq.alloc_coefs(B)
    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_an = timeit(lambda: cq.analysis(spec, cq.win_fwd, coefs))
    t_sy = timeit(lambda: cq.synthesis_spec(coefs, cq.win_bwd, 2.0 / L))
    t_fft = timeit(lambda: cq.fft.rfft(x))
    by = B * ((L // 2 + 1) * 8 + ncoef * 8)
    # GPU-side durations (HIP events around each launch, the library's measurement hook), one operation at a time - the
    # slots are shared (twiddle_transpose / spec_scale also report into "cqt_gather").  The Python loop above is bound by
    # the call rate (~20 us per call) at small B.
    def gpu_us(fn, slots, n=20):
        _lib.prof_read()
        _lib.prof_enable(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        pr = _lib.prof_read()
        return [pr[k]["ms"] * 1e3 / max(pr[k]["launches"], 1) for k in slots]
    g_an, = gpu_us(lambda: cq.analysis(spec, cq.win_fwd, coefs), ["cqt_band_analysis"])
    g_sy, g_ga = gpu_us(lambda: cq.synthesis_spec(coefs, cq.win_bwd, 2.0 / L), ["cqt_band_synthesis", "cqt_gather"])
    g_dft, g_tt = gpu_us(lambda: cq.fft.rfft(x), ["dft_stage", "cqt_gather"])
    tb = lambda us: by / us / 1e6          # algorithmic bytes (5.63 MB per clip) / GPU time -> TB/s
    print(f"B={B:3d} [GPU time, HIP events] band_analysis {g_an:7.1f} us = {tb(g_an):5.2f} TB/s ({tb(g_an) / 8 * 100:4.1f}% of 8 TB/s) | "
          f"band_synthesis {g_sy:7.1f} us + gather {g_ga:6.1f} us = {tb(g_sy + g_ga):5.2f} TB/s ({tb(g_sy + g_ga) / 8 * 100:4.1f}%) "
          f"[band_synthesis alone {tb(g_sy):5.2f} TB/s] | rfft_L: " +
          (f"mixed-radix, 2 launches {g_tt:6.1f} us" if getattr(cq.fft, "mixed", False) else f"2 dense DFT stages {g_dft:6.1f} us each + twiddle_transpose {g_tt:6.1f} us"))
    t_fftT = timeit(lambda: cq.fft.rfft_T(spec))
    print(f"B={B:3d} [Python loop, wall]    analysis {t_an*1e3:8.1f} us | synthesis(+gather) {t_sy*1e3:8.1f} us | rfft_L {t_fft*1e3:8.1f} us | rfft_L^T {t_fftT*1e3:8.1f} us  (mixed-radix: {getattr(cq.fft, 'mixed', False)})")
    # whole transforms (what the UNet calls): fwd = rfft_L + band analysis; algorithmic bytes = signal in + coefficients out
    t_fwd = timeit(lambda: cq.fwd_planar(x))
    by_fwd = B * (L * 4 + ncoef * 8)
    print(f"B={B:3d} [whole CQT.fwd, wall] {t_fwd*1e3:8.1f} us = {by_fwd / t_fwd / 1e9:6.3f} TB/s on {by_fwd / B / 1e6:.2f} MB per clip ({by_fwd / t_fwd / 1e9 / 8 * 100:4.1f}% of 8 TB/s)")
