#!/usr/bin/env python3
"""Side-by-side per-kernel totals of two rocprofv3 --kernel-trace --stats runs.  usage: kstats_diff.py <dir A> <dir B> [top N]"""
import csv
import glob
import re
import sys


def load(d):
    out = {}
    for p in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            c, t = out.get(n, (0, 0))
            out[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
    return out


A, B = load(sys.argv[1]), load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
names = sorted(set(A) | set(B), key=lambda n: -(A.get(n, (0, 0))[1] + B.get(n, (0, 0))[1]))
ta, tb = sum(v[1] for v in A.values()), sum(v[1] for v in B.values())
print(f"{'kernel':58s} {'calls A':>8s} {'ms A':>9s} {'us/call':>8s} | {'calls B':>8s} {'ms B':>9s} {'us/call':>8s} | {'B-A ms':>8s}")
for n in names[:top]:
    ca, xa = A.get(n, (0, 0))
    cb, xb = B.get(n, (0, 0))
    print(f"{n[:58]:58s} {ca:8d} {xa / 1e6:9.2f} {xa / 1e3 / max(ca, 1):8.2f} | {cb:8d} {xb / 1e6:9.2f} {xb / 1e3 / max(cb, 1):8.2f} | {(xb - xa) / 1e6:8.2f}")
print(f"{'all kernels':58s} {'':8s} {ta / 1e6:9.2f} {'':8s} | {'':8s} {tb / 1e6:9.2f} {'':8s} | {(tb - ta) / 1e6:8.2f}")
