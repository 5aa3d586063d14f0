#!/usr/bin/env python3
"""Aggregate the two PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, each with --kernel-trace only, run
separately as MI355X_MICROARCH.md prescribes) into per-kernel HBM-side bytes per launch and write
profiles/rNN_pmc_hbm_traffic.csv + profiles/rNN_conv_traffic.json (read by bench.py's roofline.traffic).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out_prefix> [kernel substring, default conv_wino4p_kernel]

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B for wide coalesced reads,
so it is doubled (guide: HBM / rocprofv3 section).  The counters sit at the L2's memory side: Infinity-Cache hits are
included, i.e. this is L2-miss traffic, an upper bound of the HBM traffic."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(\w+<[^>]*>|\w+)\(", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            k = m.group(1) if m else r["Kernel_Name"][:60]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    fd, wd, prefix = sys.argv[1:4]
    sel = sys.argv[4] if len(sys.argv) > 4 else "conv_wino4p_kernel"
    f, fc = load(fd, "FETCH_SIZE")
    w, wc = load(wd, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, 0) * 2 + w.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        rows.append((k, n, f.get(k, 0) * 2 * 1024 / n, w.get(k, 0) * 1024 / n))
    with open(prefix + "_pmc_hbm_traffic.csv", "w") as o:
        o.write("kernel,launches,fetch_bytes_per_launch_x2_corrected,write_bytes_per_launch\n")
        for k, n, fb, wb in rows:
            o.write(f"\"{k}\",{n},{fb:.0f},{wb:.0f}\n")
    n = sum(r[1] for r in rows if sel in r[0])
    fb = sum(r[2] * r[1] for r in rows if sel in r[0]) / max(n, 1)
    wb = sum(r[3] * r[1] for r in rows if sel in r[0]) / max(n, 1)
    json.dump({"kernel": sel, "launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
               "bytes_per_launch": fb + wb,
               "source": "tools/pmc_traffic.py over separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python3 "
                         "bench.py --T 2 --steps 1 --warmup 0 --no-cpu-baseline` (one launch = one segment); FETCH_SIZE doubled "
                         "(gfx950 correction); L2-miss traffic incl. Infinity-Cache hits"},
              open(prefix + "_conv_traffic.json", "w"), indent=1)
    print(open(prefix + "_conv_traffic.json").read())


if __name__ == "__main__":
    main()
