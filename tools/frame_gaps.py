"""Gaps between consecutive kernels of the context's stream in a rocprofv3 --kernel-trace CSV (the steady-state middle of the run):
   python tools/frame_gaps.py <dir with *_kernel_trace.csv>
Prints, per (kernel, next kernel) pair, the median and mean idle time between the end of one and the start of the next on the busiest queue."""
import collections
import csv
import glob
import sys


def short(n):
    if "raygen_queue" in n: return "any-hit"
    if "reflection_queue" in n: return "mirror"
    if "temporal" in n: return "svgf.comp"
    if "atrous" in n: return "a-trous " + n.split("<")[1].split(",")[0]
    return n.split("(")[0][-24:]


def main():
    files = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=lambda f: -len(open(f).read()))
    rows = [r for r in csv.DictReader(open(files[0])) if "vhr::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    mid = rows[len(rows) // 3: 2 * len(rows) // 3]
    queues = collections.Counter(r["Queue_Id"] for r in mid)
    main_q = queues.most_common(1)[0][0]
    seq = [r for r in mid if r["Queue_Id"] == main_q]
    gaps = collections.defaultdict(list)
    for a, b in zip(seq, seq[1:]):
        gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    frames = sum(1 for r in seq if "raygen_queue" in r["Kernel_Name"])
    total = 0.0
    print(f"{files[0]}: {len(seq)} launches of {frames} frames on queue {main_q} (queues: {dict(queues)})")
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < frames // 2:
            continue
        v.sort()
        print(f"  {k[0]:>12} -> {k[1]:<12} n {len(v):5d}   median {v[len(v) // 2] / 1e3:7.2f} us   mean {sum(v) / len(v) / 1e3:7.2f} us")
        total += sum(v) / len(v)
    period = (int(seq[-1]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / max(1, frames) / 1e3
    print(f"  idle between kernels per frame (sum of means): {total / 1e3:.2f} us of a {period:.1f} us frame period")


if __name__ == "__main__":
    main()
