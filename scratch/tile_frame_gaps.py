import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "vhr::" in r["Kernel_Name"] and "gbuffer" not in r["Kernel_Name"] and "k0_" not in r["Kernel_Name"] and "bvh" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames start with the any-hit queue kernel
starts = [i for i, r in enumerate(rows) if "raygen_queue_kernel" in r["Kernel_Name"]]
frames = [rows[a:b] for a, b in zip(starts[-20:-1], starts[-19:])]
import collections
busy_by = collections.defaultdict(float); span = 0.0; busy = 0.0; union = 0.0
for fr in frames:
    s0 = int(fr[0]["Start_Timestamp"]); e1 = max(int(r["End_Timestamp"]) for r in fr)
    span += (e1 - s0)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in fr)
    cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e: union += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    union += cur_e - cur_s
    for r in fr:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
        busy_by[r["Kernel_Name"].split("(")[0][:70]] += d
n = len(frames)
nxt = [int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(frames[:-1], frames[1:])]
print(f"{n} frames: frame period {sum(nxt) / len(nxt) / 1e3:.1f} us, first launch to last end {span / n / 1e3:.1f} us, some kernel running {union / n / 1e3:.1f} us, sum of kernel times {busy / n / 1e3:.1f} us, launches per frame {sum(len(f) for f in frames) / n:.1f}")
for k, v in sorted(busy_by.items(), key=lambda kv: -kv[1]): print(f"   {v / n / 1e3:8.1f} us  {k}")
