import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:58]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "pair" in k or "buf_kernel" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:32s} {sum(v)/len(v):16.0f}")
