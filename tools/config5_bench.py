"""Builder-side timing of BASELINE configs[4] on ONE MI355X: a 30 s 16 kHz recording through the whole flow - 16000 -> 22050 ->
full denoiser (depth 6, num_tfc 3) -> 16000 -> blind filter estimate on 2 segments (one coupled batch) -> AR bandwidth extension
over the file (3 segments of 184184 samples, sequential by construction) - with the FULL-width network, T = 35, order 2,
sigma_data 0.15 / sigma_max 2 / rho 9 / start_sigma 0.6, random weights.  Usage: python3 tools/config5_bench.py [bf16|f32]"""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_flows as tf                                  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tm = {}
out, filt, pred, pre, rec, net = tf._config5_flow(T=35, start_sigma=0.6, precision=prec, timing=tm)
print(json.dumps({"workload": "configs[4] on one GPU: 30 s @ 16 kHz, denoise (22.05 kHz) + blind step (2 segments) + AR BWE (3 "
                              "segments), T = 35, full width, " + prec,
                  "seconds": round(tm["seconds"], 3), "audio_sec_per_s": round(tm["audio_seconds"] / tm["seconds"], 4),
                  "score_evaluations": 69 * 4, "output_finite": bool(out.isfinite().all())}))
