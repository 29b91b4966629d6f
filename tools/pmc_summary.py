import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "vhr::" not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (tot, n) in sorted(cs.items()):
        print(f"   {c:34s} per-dispatch avg {tot / n:16.1f}   (dispatches {n})")
