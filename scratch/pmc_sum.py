import sys, csv, glob, collections
# usage: pmc_sum.py dir  -> per kernel per counter average
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "raygen" not in k: continue
    print(k[:110])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
