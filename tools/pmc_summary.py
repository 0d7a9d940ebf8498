import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "conv"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:60], r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k)
    for c, val in sorted(v.items()):
        print("   %-28s %16.0f  (per launch %14.1f)" % (c, val, val / cnt[(k, c)]))
