"""Known-answer pairs for decision (vi), one set per class of tools/audit_decision_vi.py, taken from an audit of the oracle's own rays
(the raytraced render path on the rotated 0.3-detail sponza_proc of tests/test_raytraced_path.py, 232 x 220, frames 0-1):
-> tests/golden/kat_decision_vi.json.  TEST INFRASTRUCTURE; run here (CPU only), the file is committed.

Each entry holds the ray and the triangle as fp32 hex strings, the fp32 rule mask (bit 0 Moeller-Trumbore's comparisons pass, bit 1 the
solution is consistent, bit 3 the rule in force accepts), the exact decision (python Fractions), and the class."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob                     # noqa: E402
from vulkanhybridrenderer_amd import camera, scenes   # noqa: E402
from tests import exact_rational                     # noqa: E402


def main():
    scene = scenes.rotated(scenes.sponza_proc(0.3), rot_y=0.6, rot_x=0.25)
    osc = ob.Scene(scene)
    W, H = 232, 220
    recs = []
    for pfd in camera.dolly_frames(scene, W, H, 2):
        with ob.Audit(max_records=400000) as a:
            osc.raytraced(pfd, W, H)
        recs.append(a.records)
    recs = np.concatenate(recs)
    out = []

    def take(name, sel, n=3):
        chosen = recs[sel][:n]
        assert len(chosen), name
        for r in chosen:
            hit, _ = exact_rational.ray_triangle(r["o"], r["d"], r["v0"], r["e1"], r["e2"], r["tmin"], r["tmax"])
            assert int(hit) == int(r["exact"])
            out.append(dict(cls=name, mask=int(r["pass_mask"]) & 0b1011, exact_hit=bool(hit), tmin=float(r["tmin"]).hex(), tmax=float(r["tmax"]).hex(),
                            **{k: [float(x).hex() for x in r[k]] for k in ("o", "d", "v0", "e1", "e2")}))
    m = recs["pass_mask"]
    take("B: comparisons pass, the solution contradicts itself, binary64 and exact arithmetic miss", (recs["cls"] == b"B") & ((m & 0b1011) == 0b0001))
    take("C(r5): comparisons pass, the solution contradicts itself, binary64 and exact arithmetic HIT (round 5 rejected these)", (recs["cls"] == b"C") & ((m & 0b1011) == 0b1001))
    take("D: accepted as consistent, exact arithmetic misses (fp32's own edge band)", (recs["cls"] == b"D") & ((m & 0b1011) == 0b1011))
    take("E: the fp32 comparisons fail on an exact hit (fp32's own edge band)", recs["cls"] == b"E")
    path = os.path.join(ROOT, "tests", "golden", "kat_decision_vi.json")
    json.dump(out, open(path, "w"), indent=1)
    print(len(out), "entries ->", path)


if __name__ == "__main__":
    main()
