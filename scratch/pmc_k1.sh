cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_pmc_k1; rm -rf $O; mkdir -p $O
B="--steps 8 --warmup 2 --no-cpu-baseline --no-extras --min-seconds 0"
for v in 1 0; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pre$v -- python3 $R/bench.py $B --option raygen_precompute=$v > $O/pre$v.log 2>&1 || { echo fail $v; tail -3 $O/pre$v.log; }
done
python3 - <<PY
import csv, glob, collections
for v in (1, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/pre%d/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "raygen_queue" in k or "redo" in k:
            print("precompute", v, k, {c: round(sum(x) / len(x)) for c, x in cs.items()}, len(next(iter(cs.values()))))
PY
