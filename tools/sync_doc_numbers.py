#!/usr/bin/env python3
"""After `tools/finalize_round.sh <tag> collect`: write the numbers of profiles/<tag>_bench_1gpu.json and <tag>_other_configs_1gpu.json into the
places of DESIGN.md, README.md and profiles/README.md that quote them (the frame, Grays/s, the frame with the mirror ray, configs 2-5).
usage: python tools/sync_doc_numbers.py [tag]"""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
d = json.loads(open(os.path.join(root, "profiles", f"{tag}_bench_1gpu.json")).read().strip().splitlines()[-1])
o = json.load(open(os.path.join(root, "profiles", f"{tag}_other_configs_1gpu.json")))
ms, val, wm = d["ms_per_step"], d["value"], d["ms_per_step_with_mirror_ray"]
c2, c3, c4, c5 = [o[k]["ms_per_step"] for k in ("cfg2_1080p_shadows_only", "cfg3_4k_4spp", "cfg4_bistro_1080p_full_hybrid", "cfg5_bistro_4k_16spp_2bounce")]
g3 = o["cfg3_4k_4spp"]["value"] / 1000
subs = {
    "profiles/README.md": [(r"0\.4\d\d\d ms / 1\d \d\d\d Mrays/s", f"{ms:.4f} ms / {int(val) // 1000} {int(val) % 1000:03d} Mrays/s"),
                           (r"0\.3\d\d / 2\.\d\d / 0\.9\d\d / 9\.\d\d ms", f"{c2:.3f} / {c3:.2f} / {c4:.3f} / {c5:.2f} ms")],
    "DESIGN.md": [(r"\| \*\*0\.4\d\d\d\*\* \|", f"| **{ms:.4f}** |"),
                  (r"config 3 \(4K, 4 AO\) 2\.\d\d ms = 1\d\.\d Grays/s", f"config 3 (4K, 4 AO) {c3:.2f} ms = {g3:.1f} Grays/s"),
                  (r"config 4 \(bistro_proc 1080p, full hybrid\) 0\.9\d\d ms", f"config 4 (bistro_proc 1080p, full hybrid) {c4:.3f} ms"),
                  (r"config 5 \(bistro_proc 4K, 16 AO, two bounces\) 9\.\d\d ms", f"config 5 (bistro_proc 4K, 16 AO, two bounces) {c5:.2f} ms"),
                  (r"its frame 1\.12 → 0\.9\d\d ms", f"its frame 1.12 → {c4:.3f} ms"),
                  (r"\| 0\.6\d\d \(mirror-ray launch", f"| {wm:.3f} (mirror-ray launch"),
                  (r"NOT met \(0\.6\d\d\) and config 4", f"NOT met ({wm:.3f}) and config 4")],
    "README.md": [(r"0\.4\d\d\d ms/frame", f"{ms:.4f} ms/frame"), (r"With the mirror ray 0\.6\d\d ms", f"With the mirror ray {wm:.3f} ms"),
                  (r"config 3 \(4K, 4 AO samples\) 2\.\d\d ms / 1\d\.\d Grays/s; config 4 \(bistro_proc 1080p, full hybrid\) 0\.9\d+ ms;\n  config 5 \(bistro_proc 4K, 16 AO, two bounces\) 9\.\d\d ms",
                   f"config 3 (4K, 4 AO samples) {c3:.2f} ms / {g3:.1f} Grays/s; config 4 (bistro_proc 1080p, full hybrid) {c4:.3f} ms;\n  config 5 (bistro_proc 4K, 16 AO, two bounces) {c5:.2f} ms")],
}
for path, pairs in subs.items():
    text = open(os.path.join(root, path)).read()
    for pattern, new in pairs:
        if not re.search(pattern, text):
            print(f"{path}: nothing matches {pattern!r}", file=sys.stderr)
            continue
        text = re.sub(pattern, new, text, count=1)
    open(os.path.join(root, path), "w").write(text)
print(f"frame {ms} ms, {val} Mrays/s, with the mirror ray {wm} ms, configs 2-5 {c2} / {c3} / {c4} / {c5} ms; DESIGN.md {os.path.getsize(os.path.join(root, 'DESIGN.md'))} bytes")
