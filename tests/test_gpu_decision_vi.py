"""Decision (vi) as the DEVICE computes it (vhr_debug_ray_triangle: the walkers' ray_triangle() on explicit pairs) against the oracle's orc_ray_triangle, bit for
bit, and against exact arithmetic's known answers -- no scene, no tree in between.  The reference leaves the test to the driver (raygen.rgen:39,51,64)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from vulkanhybridrenderer_amd import lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def _oracle_pairs(ob, pairs):
    """orc_ray_triangle and the rules' mask per pair -> hit (n,), tuv (n, 3), mask (n,)"""
    L = ob.lib()
    n = len(pairs)
    hit, tuv, mask = np.zeros(n, bool), np.zeros((n, 3), f32), np.zeros(n, np.uint32)
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    for i, p in enumerate(pairs):
        a = [np.ascontiguousarray(p[k:k + 3]) for k in (0, 3, 6, 9, 12)]
        ptr = [x.ctypes.data_as(C.c_void_p) for x in a]
        hit[i] = bool(L.orc_ray_triangle(*ptr, float(p[15]), float(p[16]), C.byref(t), C.byref(u), C.byref(v)))
        if hit[i]:
            tuv[i] = (t.value, u.value, v.value)
        mask[i], _ = ob.ray_triangle_rules(*a, float(p[15]), float(p[16]))
    return hit, tuv, mask


def test_known_answer_pairs_on_the_device(oracle):
    """tests/golden/kat_decision_vi.json: every class of the audit (B: a hit that is not there; C: a true hit at grazing incidence round 5 threw away; D, E: fp32's
    own edge band).  The device's decision is the rule in force (mask bit 3) and its (t, u, v) are the oracle's bits."""
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_decision_vi.json")))
    h = lambda xs: [float.fromhex(x) for x in xs]     # noqa: E731
    pairs = np.array([h(k["o"]) + h(k["d"]) + h(k["v0"]) + h(k["e1"]) + h(k["e2"]) + [float.fromhex(k["tmin"]), float.fromhex(k["tmax"])] for k in kats], f32)
    ctx = lib.Context(64, 64)
    try:
        hit, tuv = ctx.ray_triangle(pairs)
    finally:
        ctx.close()
    ohit, otuv, _ = _oracle_pairs(oracle, pairs)
    for i, k in enumerate(kats):
        assert bool(hit[i]) == bool(k["mask"] & 0b1000), k["cls"]
        if k["cls"][0] in "BC":
            assert bool(hit[i]) == k["exact_hit"], k["cls"]                   # where the second half decides, it decides like exact arithmetic
    assert np.array_equal(hit, ohit) and np.array_equal(tuv.view(np.uint32), otuv.view(np.uint32))


def test_grazing_pairs_device_equals_oracle_bit_for_bit(oracle):
    """40 000 seeded pairs aimed at where the rule matters: rays in or within 1e-7..1e-1 of a triangle's plane, through its interior, its edges and its
    corners, at scene scale (coordinates up to 50, edges 0.05..5).  Hit / miss and the bits of (t, u, v) equal the oracle's; thousands of the pairs take the
    binary64 path (fp32 comparisons pass, solution contradicts itself)."""
    rng = np.random.default_rng(20260605)
    n = 40000
    v0 = rng.uniform(-50, 50, (n, 3))
    scale = 10.0 ** rng.uniform(-1.3, 0.7, (n, 1))
    e1 = rng.normal(size=(n, 3)) * scale
    e2 = rng.normal(size=(n, 3)) * scale
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    kind = rng.integers(0, 4, n)
    bu, bv = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    flip = bu + bv > 1
    bu[flip], bv[flip] = 1 - bu[flip], 1 - bv[flip]
    bu[kind == 1] = 0.0                                                        # an edge
    bv[kind == 2] = 1.0 - bu[kind == 2]                                        # the other edge
    bu[kind == 3], bv[kind == 3] = 0.0, 0.0                                    # a corner
    point = v0 + bu[:, None] * e1 + bv[:, None] * e2
    along = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
    along /= np.linalg.norm(along, axis=1, keepdims=True)
    eps = rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1], n)[:, None]
    d = along + eps * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0 = 10.0 ** rng.uniform(-2, 1.5, (n, 1))
    o = point - t0 * d + nrm * rng.choice([0.0, 0.0, 1e-6, 1e-4], n)[:, None]
    pairs = np.concatenate([o, d, v0, e1, e2, np.full((n, 1), 1e-3), np.full((n, 1), 1e4)], axis=1).astype(f32)
    ctx = lib.Context(64, 64)
    try:
        hit, tuv = ctx.ray_triangle(pairs)
    finally:
        ctx.close()
    ohit, otuv, mask = _oracle_pairs(oracle, pairs)
    binary64 = ((mask & 1) != 0) & ((mask & 2) == 0)
    assert binary64.sum() > 1000 and ohit.sum() > 5000 and (~ohit).sum() > 5000, (int(binary64.sum()), int(ohit.sum()))
    assert (ohit & binary64).sum() > 100 and (~ohit & binary64).sum() > 100      # binary64 confirms some and rejects some
    bad = np.nonzero(hit != ohit)[0]
    assert len(bad) == 0, f"{len(bad)} decisions differ, first pair {pairs[bad[0]].tolist()}"
    assert np.array_equal(tuv.view(np.uint32), otuv.view(np.uint32))
