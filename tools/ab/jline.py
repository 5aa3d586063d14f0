"""usage: python3 bench.py ... | python3 tools/ab/jline.py LABEL: one line of the bench JSON's value, finiteness and dispatch counts."""
import json
import sys


def find(d, key):
    if isinstance(d, dict):
        if key in d:
            return d[key]
        for v in d.values():
            r = find(v, key)
            if r is not None:
                return r
    return None


d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = find(d, "conv_dispatch_counts_timed_region") or {}
print(sys.argv[1], d["value"], d["output_finite"], {k: c[k] for k in ("conv53_wino85", "gn_stats", "scale_gelu", "gn_bwd_partial") if k in c})
