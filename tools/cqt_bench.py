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
    # GPU-side durations (HIP events around each launch, the library's measurement hook): the loop above is bound by the
    # Python call rate (~20 us) at small B
    _lib.prof_read()
    _lib.prof_enable(True)
    for _ in range(20):
        cq.analysis(spec, cq.win_fwd, coefs)
        cq.synthesis_spec(coefs, cq.win_bwd, 2.0 / L)
        cq.fft.rfft(x)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_read()
    ev = {k: (pr[k]["ms"] * 1e3 / max(pr[k]["launches"], 1), pr[k]["bytes"] / max(pr[k]["ms"], 1e-9) / 1e6) for k in
          ("cqt_band_analysis", "cqt_band_synthesis", "cqt_gather", "dft_stage")}
    print(f"B={B:3d} [HIP events] band_analysis {ev['cqt_band_analysis'][0]:7.1f} us {ev['cqt_band_analysis'][1]:7.0f} GB/s "
          f"({ev['cqt_band_analysis'][1] / 80:5.1f}% of 8 TB/s) | band_synthesis {ev['cqt_band_synthesis'][0]:7.1f} us "
          f"{ev['cqt_band_synthesis'][1]:7.0f} GB/s ({ev['cqt_band_synthesis'][1] / 80:5.1f}%) | gather "
          f"{ev['cqt_gather'][0]:6.1f} us | DFT stages (2 per rfft_L) {ev['dft_stage'][0]:7.1f} us each")
    print(f"B={B:3d} analysis {t_an*1e3:8.1f} us {by/t_an/1e6:8.1f} GB/s ({by/t_an/1e6/8000*100:5.1f}% of 8 TB/s) | "
          f"synthesis(+gather) {t_sy*1e3:8.1f} us {by/t_sy/1e6:8.1f} GB/s | rfft_L {t_fft*1e3:8.1f} us")
