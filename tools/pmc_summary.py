import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "conv_bf16"
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        key = (k.split("(")[0][-60:], r.get("Grid_Size", ""))
        rows[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(key, r["Counter_Name"])] += 1
for key, c in rows.items():
    print(key, {k: round(v / n[(key, k)]) for k, v in c.items()})
