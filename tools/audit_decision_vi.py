"""The audit of decision (vi): every fp32 ray / triangle decision of the oracle against exact arithmetic.

TEST INFRASTRUCTURE (it drives the oracle; the product is not involved).  For each scene and each of two dolly frames the
oracle produces the G-buffer (primary rays, closest hit), the shadow rays, the AO rays (any hit) and the mirror rays (closest
hit) -- each kind in its own audit session (oracle/vhr_oracle.h orc_audit_begin / _end) -- and every (ray, triangle) pair the
binary64 walk meets is classified:

    A  fp32 accepts, exact arithmetic hits          D  fp32 accepts, exact arithmetic misses
    B  Moeller-Trumbore's comparisons pass, the      C  Moeller-Trumbore's comparisons pass, the second half of decision (vi)
       second half rejects, exact misses                rejects, exact arithmetic HITS (a hit the rule costs)
    E  Moeller-Trumbore's comparisons themselves fail on an exact hit (fp32's own edge band; no rule involved)

for four forms of the second half: none; round 5's rule (a candidate whose fp32 solution contradicts itself is rejected); a rule weighed in
round 6 and dropped (the reported point lies in the triangle's world-axes box: not independent of the hierarchy's frame); and the rule in
force (a candidate whose fp32 solution contradicts itself is decided again in binary64).  Pairs the binary64 filter leaves
undecided are settled here with python Fractions (tests/exact_rational.py).  Members of C, D and E are "explained" when the exact
barycentric point lies within fp32 Moeller-Trumbore's own forward error bound of the triangle's border or of the t interval's ends:
no fp32 evaluation of these formulas can promise that decision.

usage: python tools/audit_decision_vi.py [--width 1920 --height 1080] [--scenes sponza_proc,...] [--out profiles/r6_decision_vi.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as ob                     # noqa: E402
from vulkanhybridrenderer_amd import abi, camera, scenes   # noqa: E402
from tests import exact_rational                     # noqa: E402

EPS = 2.0 ** -24


def mt_error_margin(rec):
    """How far (in units of fp32 Moeller-Trumbore's forward error bound) the exact (u, v, t) sits from the nearest decision border.
    <= 1 means: inside the band where the rounding of the fp32 evaluation decides."""
    o, d, v0, e1, e2 = (np.asarray(rec[k], np.float64) for k in ("o", "d", "v0", "e1", "e2"))
    tv = np.abs(o) + np.abs(v0)
    ad, a1, a2 = np.abs(d), np.abs(e1), np.abs(e2)

    def pcross(a, b):
        return np.array([a[1] * b[2] + a[2] * b[1], a[2] * b[0] + a[0] * b[2], a[0] * b[1] + a[1] * b[0]])
    ppv, pqv = pcross(ad, a2), pcross(tv, a1)
    pdet, pun, pvn, ptn = a1 @ ppv, tv @ ppv, ad @ pqv, a2 @ pqv
    det = abs(rec["xdet"])
    if det == 0:
        return 0.0
    k = 8 * EPS                                       # 7 roundings to a numerator + the reciprocal and the product
    u, v, t = rec["xu"], rec["xv"], rec["xt"]
    eu = k * (pun + abs(u) * pdet) / det
    ev = k * (pvn + abs(v) * pdet) / det
    et = k * (ptn + abs(t) * pdet) / det
    margins = [abs(u) / eu, abs(1 - u) / eu, abs(v) / ev, abs(1 - u - v) / (eu + ev + EPS), abs(t - rec["tmin"]) / et, abs(rec["tmax"] - t) / et]
    return float(min(margins))


def settle_undecided(records):
    """exact rationals for the pairs the binary64 filter left open -> list of (record, exact hit)"""
    out = []
    for r in records[records["cls"] == b"U"]:
        hit, _ = exact_rational.ray_triangle(r["o"], r["d"], r["v0"], r["e1"], r["e2"], r["tmin"], r["tmax"])
        out.append((r, hit))
    return out


def trace_kinds(tp_default):
    kinds = []
    for name, shadow, ao, refl in (("shadow (any hit)", 1, 0, 0), ("AO (any hit)", 0, 2, 0), ("mirror (closest hit)", 0, 0, 1)):
        tp = tp_default.copy()
        tp["shadow_enable"], tp["ao_spp"], tp["reflections"] = shadow, ao, refl
        kinds.append((name, tp, bool(refl)))
    return kinds


def fmt_counts(c, log):
    log(f"      rays {int(c['rays'].sum()):>10}   pairs {int(c['pairs']):>12}   exact hits {int(c['exact_hits']):>10}   fp32 comparisons pass {int(c['mt_hits']):>10}   sent to binary64 {int(c['escalated']):>7}"
        f"   E (fp32 comparisons miss an exact hit) {int(c['mt_miss_exact_hit'])}   undecided in binary64: {int(c['undecided'])} pairs / {int(c['rays_undecided'])} rays")
    for r, name in enumerate(ob.AUDIT_RULES):
        a, b, cc, d = (int(x) for x in c["cls"][r])
        per_ray = []
        if c["rays"][0]:
            per_ray.append(f"any-hit rays: leaks {int(c['any_leak'][r])}, spurious occlusions {int(c['any_spurious'][r])}")
        if c["rays"][1]:
            per_ray.append(f"closest-hit rays: hit/miss differs {int(c['closest_hit_miss'][r])}, another triangle wins {int(c['closest_differs'][r])} (of them farther than 1e-4 t: {int(c['closest_differs_far'][r])})")
        per_ray = "; ".join(per_ray)
        log(f"      {name:<48} A {a:>10}  B {b:>6}  C {cc:>6}  D {d:>6}   {per_ray}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--scenes", default="sponza_proc,bistro_proc,sponza_proc_rot,sponza_hard_rot,bistro_proc_rot")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--raytraced", action="store_true", help="also audit the raytraced render path's rays (row f4: shadow rays leave the hit point without a bias)")
    ap.add_argument("--brute", action="store_true", help="the raytraced path's session tests every triangle instead of walking the boxes (complete B / D counts)")
    ap.add_argument("--append", action="store_true", help="append to --out instead of replacing it")
    args = ap.parse_args()
    lines = []

    def log(s=""):
        print(s, flush=True)
        lines.append(s)

    W, H = args.width, args.height
    log(f"audit of decision (vi): {W} x {H}, {args.frames} dolly frames per scene, every primary / shadow / AO / mirror ray of the oracle; {ob.max_threads()} host threads")
    log("classes per (ray, triangle) pair -- A: accepted, exact hit; B: comparisons pass, rule rejects, exact miss; C: comparisons pass, rule rejects, EXACT HIT;")
    log("D: accepted, exact miss; E: fp32 comparisons fail on an exact hit.  'explained' = the exact (u, v, t) lies within fp32 Moeller-Trumbore's forward error bound of a decision border.")
    totals = {}
    unexplained = []
    tp_default = np.zeros((), abi.trace_params_dtype)
    ob.lib().orc_default_trace_params(ob._p(tp_default))
    for scene_name in args.scenes.split(","):
        if args.detail != 1.0:                        # (the *_rot makers take no detail: turn the detailed scene here)
            rot = scene_name.endswith("_rot")
            scene = getattr(scenes, scene_name[:-4] if rot else scene_name)(args.detail)
            if rot:
                scene = scenes.rotated(scene, name=scene_name)
        else:
            scene = getattr(scenes, scene_name)()
        osc = ob.Scene(scene)
        log(f"\n== {scene_name}: {osc.triangle_count} triangles")
        for fi, pfd in enumerate(camera.dolly_frames(scene, W, H, args.frames)):
            sessions = []
            t0 = time.time()
            with ob.Audit(brute_force=args.brute and not args.raytraced) as a:
                gbuf = osc.gbuffer(pfd, W, H)
            sessions.append(("primary (closest hit)", a))
            for name, tp, refl in trace_kinds(tp_default):
                with ob.Audit() as a:
                    osc.raygen(pfd, tp, gbuf[0], gbuf[2], want_reflections=refl)
                sessions.append((name, a))
            if args.raytraced:
                with ob.Audit(brute_force=args.brute) as a:
                    osc.raytraced(pfd, W, H)
                sessions.append(("raytraced render path: primary (closest hit) + shadow rays from the hit point itself (any hit)" + (", BRUTE FORCE over every triangle" if args.brute else ""), a))
            log(f"  frame {fi} ({time.time() - t0:.0f} s)")
            for name, a in sessions:
                c = a.counts
                log(f"    {name}")
                fmt_counts(c, log)
                if int(c["records_dropped"]):
                    log(f"      ({int(c['records_dropped'])} records dropped: raise max_records)")
                tot = totals.setdefault(name, np.zeros((), ob.audit_counts_dtype))
                for f in ob.audit_counts_dtype.names:
                    tot[f] += c[f]
                settled = settle_undecided(a.records)
                if settled:
                    log(f"      undecided pairs settled with exact rationals: {len(settled)} ({sum(1 for _, h in settled if h)} hits); fp32 masks (comparisons, r5, r6) of them: "
                        + ", ".join(f"{int(r['pass_mask']):03b}/{'hit' if h else 'miss'}" for r, h in settled[:12]))
                for cls in (b"C", b"D", b"E"):
                    recs = a.records[a.records["cls"] == cls]
                    if not len(recs):
                        continue
                    m = np.array([mt_error_margin(r) for r in recs])
                    # which rules does the record concern?  C: a rule rejected it; D: a rule accepted it
                    r6 = ((recs["pass_mask"] & 8) == 0) if cls == b"C" else ((recs["pass_mask"] & 8) != 0) if cls == b"D" else np.ones(len(recs), bool)
                    tag = {b"C": "C under the rule in force", b"D": "D under the rule in force", b"E": "E"}[cls]
                    if r6.any():
                        mm = m[r6]
                        log(f"      {tag}: {len(mm)} members, distance to the nearest decision border in units of the fp32 error bound: max {mm.max():.3g}, median {np.median(mm):.3g}; beyond the bound: {int((mm > 1).sum())}")
                        for r in recs[r6][mm > 1][:50]:
                            unexplained.append((scene_name, fi, name, cls.decode(), r.copy()))
                    if cls == b"C":
                        r5 = (recs["pass_mask"] & 2) == 0
                        if r5.any():
                            mm = m[r5]
                            log(f"      C under round 5's rule: {len(mm)} members, distance to the nearest border / error bound: max {mm.max():.3g}, median {np.median(mm):.3g}; beyond the bound (true hits lost with no rounding excuse): {int((mm > 1).sum())}")
    log("\n== totals over all scenes and frames")
    for name, tot in totals.items():
        log(f"  {name}")
        fmt_counts(tot, log)
    log(f"\nmembers of C / D / E under the rule in force that lie beyond fp32's own error bound: {len(unexplained)}")
    for scene_name, fi, name, cls, r in unexplained[:40]:
        log(f"  {scene_name} frame {fi} {name} class {cls}: flat {int(r['flat'])} fp32 (t, u, v, det) = ({r['t']:.7g}, {r['u']:.7g}, {r['v']:.7g}, {r['det']:.7g}) exact (t, u, v) = ({r['xt']:.12g}, {r['xu']:.12g}, {r['xv']:.12g}) mask {int(r['pass_mask']):03b}")
        log("     o " + " ".join(float(x).hex() for x in r["o"]) + "  d " + " ".join(float(x).hex() for x in r["d"]))
        log("     v0 " + " ".join(float(x).hex() for x in r["v0"]) + "  e1 " + " ".join(float(x).hex() for x in r["e1"]) + "  e2 " + " ".join(float(x).hex() for x in r["e2"]))
    if args.out:
        with open(args.out, "a" if args.append else "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
