import csv, glob, sys, collections, re
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if pat and pat not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        nd = len(cnt[(k, n)])
        print(f"   {n:34s} {v/nd:16.1f} per dispatch ({nd} dispatches)")
