import sys, csv, glob
# usage: gaps.py dir : per-kernel durations and the idle gaps between consecutive dispatches of the steady-state frames
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
rows = rows[len(rows) // 2:]          # steady state
# find frame boundaries: raygen_queue_kernel starts a frame
frames = []
cur = []
for r in rows:
    if "raygen_queue_kernel" in r[2] and cur:
        frames.append(cur); cur = []
    cur.append(r)
frames = [f for f in frames if len(f) == len(frames[len(frames)//2])]
import statistics
n = len(frames[0])
print("kernels per frame", n, "frames", len(frames))
for i in range(n):
    dur = statistics.median((f[i][1] - f[i][0]) / 1e3 for f in frames)
    gap = statistics.median(((f[i + 1][0] if i + 1 < n else None) or f[i][1]) / 1e3 - f[i][1] / 1e3 for f in frames) if i + 1 < n else float("nan")
    print(f"{frames[0][i][2]:60s} dur {dur:8.2f} us   gap after {gap:6.2f} us")
per = statistics.median((frames[j + 1][0][0] - frames[j][0][0]) / 1e3 for j in range(len(frames) - 1))
print("frame period", per, "us; sum of durations", sum(statistics.median((f[i][1] - f[i][0]) / 1e3 for f in frames) for i in range(n)))
