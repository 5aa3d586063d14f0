#!/usr/bin/env python3
"""Per-batch-size kernel durations of the CQT kernels from a rocprofv3 --kernel-trace of tools/cqt_bench.py (the kernel's own
begin/end timestamps: no event-bracket overhead).  usage: cqt_trace_summary.py <trace dir>"""
import collections
import csv
import glob
import sys

L, NCOEF = 368368, 520192
BYTES = (L // 2 + 1) * 8 + NCOEF * 8          # algorithmic bytes per clip of one band launch: 5.63 MB
rows = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        for key in ("band_fft_kernel<0", "band_fft_kernel<1", "gather_rec_kernel", "colfft_kernel<false>", "colfft_kernel<true>"):
            if key in n:
                gy = int(r.get("Grid_Size_Y") or r.get("Grid_Size_y") or 1)
                gz = int(r.get("Grid_Size_Z") or r.get("Grid_Size_z") or 1)
                wy = int(r.get("Workgroup_Size_Y") or 1)
                rows[(key, gy // max(wy, 1), gz)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("# rocprofv3 --kernel-trace: average kernel duration per (kernel, grid.y, grid.z); band kernels: grid.y = clips")
for (k, gy, gz), v in sorted(rows.items()):
    v = sorted(v)[len(v) // 10: len(v) - len(v) // 10 or None] or v          # trimmed mean (first launches are cold)
    us = sum(v) / len(v)
    extra = ""
    if k.startswith("band_fft"):
        extra = f"  = {gy * BYTES / us / 1e6:5.2f} TB/s algorithmic = {gy * BYTES / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s"
    print(f"{k:24s} grid.y={gy:4d} grid.z={gz:3d}  n={len(v):4d}  {us:8.2f} us{extra}")
