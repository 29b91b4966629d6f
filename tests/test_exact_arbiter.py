"""The arbiter behind decision (vi) (oracle/vhr_exact.h, tests/exact_rational.py) and the classes of tools/audit_decision_vi.py.  CPU only."""
import json
import os

import numpy as np
import pytest

from tests import exact_rational
from vulkanhybridrenderer_amd import camera, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.build()
    return binding


def h(xs):
    return np.array([float.fromhex(x) for x in xs], f32)


def test_known_answer_pairs_of_every_class(ob):
    """tests/golden/kat_decision_vi.json (tools/make_decision_vi_kats.py): per pair the fp32 mask of the oracle (bit 0 Moeller-Trumbore's comparisons
    pass, bit 1 the solution is consistent, bit 3 the rule in force accepts), the binary64 filter's verdict and exact rationals -- class B: a hit
    that is not there, removed; class C(r5): a true hit at grazing incidence that round 5's rule threw away and binary64 keeps; classes D and E:
    fp32 Moeller-Trumbore's own edge band, where it disagrees with exact arithmetic whatever the second half says."""
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_decision_vi.json")))
    seen = set()
    for k in kats:
        o, d, v0, e1, e2 = (h(k[n]) for n in ("o", "d", "v0", "e1", "e2"))
        tmin, tmax = float.fromhex(k["tmin"]), float.fromhex(k["tmax"])
        mask, _ = ob.ray_triangle_rules(o, d, v0, e1, e2, tmin, tmax)
        assert mask & 0b1011 == k["mask"], k["cls"]
        verdict, _ = ob.ray_triangle_exact(o, d, v0, e1, e2, tmin, tmax)
        exact, _ = exact_rational.ray_triangle(o, d, v0, e1, e2, tmin, tmax)
        assert exact == k["exact_hit"] and verdict == int(exact), k["cls"]
        # the oracle's decision itself
        import ctypes as C
        t, u, v = C.c_float(), C.c_float(), C.c_float()
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        got = ob.lib().orc_ray_triangle(p(o), p(d), p(v0), p(e1), p(e2), tmin, tmax, C.byref(t), C.byref(u), C.byref(v))
        assert bool(got) == bool(mask & 0b1000)
        letter = k["cls"][0]
        seen.add(letter)
        if letter == "B":
            assert mask & 1 and not mask & 2 and not got and not exact
        elif letter == "C":
            assert mask & 1 and not mask & 2 and got and exact              # round 5: rejected.  round 6: binary64 confirms the hit
        elif letter == "D":
            assert mask & 2 and got and not exact
        elif letter == "E":
            assert not mask & 1 and not got and exact
    assert seen == set("BCDE")


def test_binary64_filter_never_contradicts_exact_rationals(ob):
    """Random and adversarial pairs: wherever oracle/vhr_exact.h commits itself (hit / miss) python Fractions agree; exact ties -- a ray through a
    vertex or along an edge of axis-aligned geometry -- are what it hands back as undecided, and rationals settle them (closed triangle)."""
    rng = np.random.default_rng(7)
    undecided = 0
    for i in range(6000):
        v0 = (rng.normal(size=3) * 10).astype(f32)
        e1 = (rng.normal(size=3) * rng.choice([0.01, 1.0, 40.0])).astype(f32)
        e2 = (rng.normal(size=3) * rng.choice([0.01, 1.0, 40.0])).astype(f32)
        a, b = rng.random(2)
        mode = i % 5
        if mode == 0: a = 0.0
        if mode == 1: b = 0.0
        if mode == 2: b = 1.0 - a
        target = v0.astype(np.float64) + a * e1 + b * e2
        o = (rng.normal(size=3) * 20).astype(f32)
        if mode == 3:                                   # grazing: the origin (nearly) in the triangle's plane
            o = (target + (rng.normal() * e1 + rng.normal() * e2)).astype(f32)
        d = target - o
        d = (d / np.linalg.norm(d)).astype(f32)
        verdict, _ = ob.ray_triangle_exact(o, d, v0, e1, e2, 0.01, 1e4)
        exact, _ = exact_rational.ray_triangle(o, d, v0, e1, e2, 0.01, 1e4)
        if verdict < 0: undecided += 1
        else: assert bool(verdict) == exact, (i, o, d, v0, e1, e2)
    assert undecided < 30
    # exact ties on a unit right triangle in z = 0: through the corner, along an edge's line, through the hypotenuse's midpoint, at t == tmax
    v0, e1, e2 = f32([0, 0, 0]), f32([1, 0, 0]), f32([0, 1, 0])
    for o, tmax, want in (([0, 0, 1], 100.0, True), ([0.5, 0, 1], 100.0, True), ([0.5, 0.5, 1], 100.0, True), ([1, 1, 1], 100.0, False),
                          ([0.25, 0.25, 1], 1.0, False), ([1, 0, 1], 100.0, True), ([-0.0, 1, 1], 100.0, True)):
        o, d = f32(o), f32([0, 0, -1])
        verdict, _ = ob.ray_triangle_exact(o, d, v0, e1, e2, 0.01, tmax)
        exact, _ = exact_rational.ray_triangle(o, d, v0, e1, e2, 0.01, tmax)
        assert exact == want and verdict in (-1, int(want)), (o, tmax, verdict)
    # a ray in the triangle's plane: the determinant is exactly 0 -> a miss, decided
    assert ob.ray_triangle_exact(f32([-1, 0.25, 0]), f32([1, 0, 0]), v0, e1, e2, 0.01, 100.0)[0] == 0


def test_audit_of_a_small_frame(ob):
    """The audit end to end on a small rotated scene, every ray kind of the hybrid path and the raytraced path's (whose shadow rays leave the hit
    point itself -- where the self-contradicting candidates live): the rule in force rejects no exact hit (class C empty by construction up to
    binary64's own band), every hit it removes is a miss in exact arithmetic, and per ray it differs from exact arithmetic no more often than
    Moeller-Trumbore alone; round 5's rule loses hits.  The BRUTE-FORCE session (every triangle, no boxes) must count the same exact hits as the walk
    through the boxes: the binary64 walk visits every pair that matters."""
    from vulkanhybridrenderer_amd import abi
    scene = scenes.rotated(scenes.sponza_proc(0.15), rot_y=0.6, rot_x=0.25)
    osc = ob.Scene(scene)
    W, H = 96, 64
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    tp = abi.default_trace_params()
    with ob.Audit() as a:
        gbuf = osc.gbuffer(pfd, W, H)
        osc.raygen(pfd, tp, gbuf[0], gbuf[2])
        osc.raytraced(pfd, W, H)
    c = a.counts
    assert int(c["rays"].sum()) > 20000 and int(c["pairs"]) > int(c["exact_hits"]) > 10000
    mt, r5, _, r6 = (c["cls"][i] for i in range(4))            # columns A, B, C, D
    undecided = a.records[a.records["cls"] == b"U"]
    assert len(undecided) == int(c["undecided"]) <= 4
    assert int(r6[2]) == 0                                      # C: no exact hit rejected
    assert int(r6[0]) == int(mt[0])                             # A: every true hit Moeller-Trumbore finds is kept ...
    assert int(r6[1]) + int(r6[3]) == int(mt[3])                # ... and of its false ones B are removed, D stay
    assert int(r6[1]) == int(r5[1])                             # the hits round 5 removed rightly are removed still
    assert int(r5[2]) >= int(r6[2])
    wrong = lambda i: int(c["any_leak"][i] + c["any_spurious"][i] + c["closest_hit_miss"][i] + c["closest_differs_far"][i])   # noqa: E731
    assert wrong(3) <= wrong(0) + 2 and wrong(3) <= wrong(1)
    with ob.Audit(brute_force=True) as b:
        osc.raytraced(pfd, W, H, rows=(20, 28))
    with ob.Audit() as w:
        osc.raytraced(pfd, W, H, rows=(20, 28))
    assert int(b.counts["exact_hits"]) == int(w.counts["exact_hits"]) and int(b.counts["pairs"]) > 50 * int(w.counts["pairs"])
    assert (b.counts["cls"][3] == w.counts["cls"][3]).all() or int(b.counts["cls"][3][1]) >= int(w.counts["cls"][3][1])      # (B may grow: hits that are not there live outside the boxes too)
