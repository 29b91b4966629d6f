"""Pins for the CPU oracle (no GPU): known-answer values derived by hand from the reference's formulas
(committed as tests/golden/kat.json), numpy's IEEE fp16 conversion, an independent numpy
restatement of the SVGF kernels, and brute-force ray casting.  The reference itself has no tests (PARITY UNPINNED)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import numpy_restatement as npr
from tests.helpers import simple_pfd, synthetic_svgf_inputs, ulp16_diff
from vulkanhybridrenderer_amd import abi, camera, scenes

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def test_struct_sizes(oracle):
    out = (C.c_uint32 * 8)()
    oracle.lib().orc_struct_sizes(out)
    assert list(out)[:6] == [56, 44, 120, 112, 584, 32]


def test_rng_known_answers(oracle, kat):
    for case in kat["rng"]:
        seed, states, vals = oracle.rng_sequence(case["input"], len(case["random01"]))
        assert seed == int(case["seed"], 16)
        assert [f"{s:#010x}" for s in states[:len(case.get("states", []))]] == case.get("states", [])
        assert np.allclose(vals, case["random01"], rtol=0, atol=5e-8)


def test_rng_matches_numpy_restatement(oracle):
    for x in [0, 1, 7, 12345, 8170673, 0xffffffff, 0x80000000]:
        seed, states, vals = oracle.rng_sequence(x, 16)
        s = npr.seed_thread(x)
        assert s == seed
        for st, v in zip(states, vals):
            s, f = npr.random01(s)
            assert s == st and f == v


def test_raygen_seed_formula(oracle, kat):
    """(y * LaunchSize.y + x) * frame_index in uint32 (raygen.rgen:17)."""
    c = kat["raygen_seed"]
    v = ((c["y"] * c["H"] + c["x"]) * c["frame"]) & 0xffffffff
    assert v == c["product"]
    assert oracle.lib().orc_seed_thread(v) == int(c["seed"], 16)


def test_fp16_known_answers_and_numpy(oracle, kat):
    L = oracle.lib()
    for k, v in kat["fp16"].items():
        assert L.orc_f32_to_f16(float(np.float32(float(k)))) == int(v, 16), k
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    special = np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e-8, 5.9604645e-8, 2.9802322e-8, 2.98023224e-8 * 1.0001,
                        6.1035156e-5, 6.0975552e-5, np.inf, -np.inf, 2049.0, 2051.0, 1.00048828125, 1.000732421875], np.float32)
    vals = np.concatenate([vals, special])
    with np.errstate(over="ignore", invalid="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_f32_to_f16(float(v)) for v in vals], np.uint16)
    nan = np.isnan(vals)
    assert np.array_equal(got[~nan], ref[~nan])
    assert ((got[nan] & 0x7c00) == 0x7c00).all() and ((got[nan] & 0x3ff) != 0).all()
    all16 = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    back = np.array([L.orc_f16_to_f32(int(h)) for h in all16], np.float32)
    ref32 = all16.view(np.float16).astype(np.float32)
    ok = np.isnan(ref32) | (back == ref32)
    assert ok.all()


def test_sincos_accuracy_and_quadrants(oracle):
    worst = 0.0
    for phi in np.linspace(0.0, 2 * np.pi, 4001, dtype=np.float32):
        s, c = oracle.sincos(phi)
        worst = max(worst, abs(float(s) - np.sin(np.float64(phi))), abs(float(c) - np.cos(np.float64(phi))))
    assert worst < 2.5e-7
    s, c = oracle.sincos(0.0)
    assert s == 0.0 and c == 1.0


def test_sampling_functions(oracle):
    L = oracle.lib()
    out = np.zeros(3, np.float32)
    rng = np.random.default_rng(1)
    for _ in range(200):
        u = rng.random(2).astype(np.float32)
        L.orc_uniform_sample_cone(float(u[0]), float(u[1]), 0.999995, out.ctypes.data_as(C.c_void_p))
        assert abs(np.linalg.norm(out) - 1) < 1e-3 and out[2] >= 0.999994        # fp32 cancellation in sin_theta, like the shader
        L.orc_cosine_hemisphere(float(u[0]), float(u[1]), out.ctypes.data_as(C.c_void_p))
        assert abs(np.linalg.norm(out) - 1) < 1e-6 and out[2] >= 0
    n = np.array([0, 0, -1], np.float32)
    M = np.zeros(9, np.float32)
    L.orc_onb(n.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p))
    assert M.tolist() == [0, -1, 0, -1, 0, 0, 0, 0, -1]                            # common.glsl:83-87


def test_ray_triangle_rules(oracle):
    L = oracle.lib()
    f = lambda *a: np.array(a, np.float32)   # noqa: E731
    v0, e1, e2 = f(0, 0, 0), f(1, 0, 0), f(0, 1, 0)
    t, u, v = C.c_float(), C.c_float(), C.c_float()

    def hit(o, d, tmin=0.01, tmax=100.0):
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        return L.orc_ray_triangle(p(o), p(d), p(v0), p(e1), p(e2), tmin, tmax, C.byref(t), C.byref(u), C.byref(v))

    assert hit(f(0.25, 0.25, 1), f(0, 0, -1)) and t.value == 1.0 and u.value == 0.25 and v.value == 0.25
    assert hit(f(0.25, 0.25, -1), f(0, 0, 1))                    # two-sided (TRIANGLE_FACING_CULL_DISABLE)
    assert not hit(f(0.25, 0.25, 1), f(1, 0, 0))                 # parallel: det == 0 -> miss
    assert not hit(f(0.25, 0.25, 1), f(0, 0, -1), tmax=1.0)      # t < tmax strictly
    assert not hit(f(0.25, 0.25, 1), f(0, 0, -1), tmin=1.0)      # t > tmin strictly
    assert hit(f(0.0, 0.0, 1), f(0, 0, -1))                      # u = v = 0 corner is inside
    assert not hit(f(0.75, 0.75, 1), f(0, 0, -1))                # u + v > 1
    # Decision (vi), second half.  A shadow ray of the raytraced path's rotated test scene (round 5), within rounding of this triangle's plane:
    # det = -2^-20 * 1.25 is rounding noise, (t, u, v) = (0.6, -0, 0.887) pass every comparison, and the point o + t d is 31 cm beside
    # v0 + u e1 + v e2 -- outside every box around the triangle.  The solution contradicts itself, so the pair is decided again in binary64
    # (round 6; round 5 rejected it outright): not a hit, and exact arithmetic agrees (tests/test_exact_arbiter.py has a pair of every class).
    h = lambda *a: np.array([float.fromhex(x) for x in a], np.float32)   # noqa: E731
    v0, e1, e2 = (h("0x1.287664p+4", "0x1.2d010ap+3", "-0x1.0ac8dcp+3"), h("0x1.40202p+0", "-0x1.21888p-1", "0x1.d3edp+0"), h("0x1.87a64p+0", "0x1.5f511p+0", "0x1.1e3c6p+1"))
    o, d = h("0x1.4216bp+4", "0x1.42c918p+3", "-0x1.7fbd2p+2"), h("-0x1.bcaf8cp-5", "0x1.fda0e4p-1", "-0x1.44ff66p-4")
    assert not hit(o, d, tmin=0.1, tmax=10000.0)
    centre = (v0 + 0.25 * e1 + 0.25 * e2).astype(np.float32)
    normal = np.cross(e1, e2).astype(np.float32)
    assert hit((centre + normal).astype(np.float32), (-normal).astype(np.float32), tmin=0.1, tmax=10000.0) and abs(u.value - 0.25) < 1e-5      # (a ray that does cross it is a hit as ever)


def test_projection_and_frame_fill(oracle):
    m = np.zeros(16, np.float32)
    oracle.lib().orc_infinite_reverse_depth_projection(0.9, 16 / 9, 0.1, m.ctypes.data_as(C.c_void_p))
    ref = abi.mat_to_glm(camera.infinite_reverse_depth_projection(0.9, 16 / 9, 0.1))
    assert np.allclose(m, ref, rtol=1e-6)
    assert m[11] == -1.0 and m[14] == np.float32(0.1) and m[10] == 0.0       # vulkan_utils.h:497-502
    sc = scenes.tiny_scene()
    f0, f1 = camera.dolly_frames(sc, 64, 48, 2)
    assert f0["frame_index"] == 0 and f1["frame_index"] == 1                 # renderer.cpp:202 post-increment
    assert not f0["camera_view_prev_frame"].any() and not f0["camera_proj_prev_frame"].any()   # zero on frame 0 (:188)
    assert np.array_equal(f1["camera_view_prev_frame"], f0["camera_view"])
    assert np.allclose(abi.glm_to_mat(f1["camera_view"]) @ abi.glm_to_mat(f1["camera_view_inverse"]), np.eye(4), atol=1e-5)


def test_bvh_equals_brute_force(oracle):
    """Box culling must never change a hit: the oracle's BVH walk equals the all-triangles loop."""
    sc = scenes.sponza_proc(detail=0.12)
    osc = oracle.Scene(sc)
    rng = np.random.default_rng(3)
    lo, hi = np.array([-19, 0.2, -7.5]), np.array([19, 13, 7.5])
    n_hit = 0
    for _ in range(1500):
        o = (lo + rng.random(3) * (hi - lo)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        assert osc.occluded(o, d, 0.01, 5.0, True) == osc.occluded(o, d, 0.01, 5.0, False)
        a, b = osc.closest(o, d, 0.01, 1e4, True), osc.closest(o, d, 0.01, 1e4, False)
        assert a == b
        n_hit += a is not None
    assert n_hit > 1000


def test_axis_aligned_rays_and_zero_direction_components(oracle):
    sc = scenes.tiny_scene()
    osc = oracle.Scene(sc)
    for d in [(0, -1, 0), (0, 1, 0), (1, 0, 0), (0, 0, -1), (0, -1, 1e-30), (-0.0, -1, 0.0)]:
        for o in [(0.1, 3.0, 0.1), (-1.4, 4.0, -0.5), (0.0, 0.5, 0.0)]:
            o32, d32 = np.array(o, np.float32), np.array(d, np.float32)
            assert osc.closest(o32, d32, 0.01, 1e4, True) == osc.closest(o32, d32, 0.01, 1e4, False)


@pytest.mark.parametrize("step", [1, 2, 8])
def test_atrous_matches_numpy_restatement(oracle, step):
    W, H = 96, 64
    normals, motion, rt = synthetic_svgf_inputs(W, H, seed=11)
    rng = np.random.default_rng(12)
    integ = np.stack([rng.random((H, W)), rng.random((H, W)), 0.2 * rng.random((H, W)), 0.2 * rng.random((H, W))], -1)
    integ = integ.astype(np.float16).view(np.uint16)
    got = oracle.svgf_atrous(simple_pfd(W, H), normals, integ, step)
    ref = npr.atrous(normals, integ, step)
    d = ulp16_diff(got, ref)
    assert d.max() <= 1 and (d == 0).mean() > 0.999          # numpy's SIMD expf may differ from libm in the last ulp


@pytest.mark.parametrize("motion", [(0.0, 0.0), (1.25, -0.5), (40.0, 7.0)])
def test_temporal_matches_numpy_restatement(oracle, motion):
    W, H = 96, 64
    normals, motion_img, rt = synthetic_svgf_inputs(W, H, seed=21, motion=motion)
    prev, _, _ = synthetic_svgf_inputs(W, H, seed=21)
    rng = np.random.default_rng(22)
    history = rng.random((H, W, 4)).astype(np.float16).view(np.uint16)
    moments = rng.random((H, W, 2)).astype(np.float16).view(np.uint16)
    gi, gm = oracle.svgf_temporal(simple_pfd(W, H), normals, motion_img, rt, prev, history, moments)
    ri, rm = npr.temporal(W, H, normals, motion_img, rt, prev, history, moments)
    assert np.array_equal(gi, ri) and np.array_equal(gm, rm)


def test_temporal_shipped_quirks(oracle):
    """(a) RG16F moments read as vec4 give (r, g, 0, 1): with a valid reprojection ao_var = 0.8 + 0.16 ao^2
    (SURVEY.md a4); (b) NaN motion (frame 0) falls through to 'current sample'."""
    W, H = 32, 16
    normals, motion_img, rt = synthetic_svgf_inputs(W, H, seed=31, n_ids=1)
    normals[..., :3] = np.array([0, 0, 1], np.float16).view(np.uint16)      # identical normals: every tap valid
    history = np.zeros((H, W, 4), np.uint16)
    moments = np.zeros((H, W, 2), np.uint16)
    gi, _ = oracle.svgf_temporal(simple_pfd(W, H), normals, motion_img, rt, normals, history, moments)
    ao = npr.h2f(rt)[..., 1]
    exp_var = np.float32(0.8) - (np.float32(0.2) * ao) ** 2 + np.float32(0.2) * ao * ao   # m2 - m1^2 with m1 = .2 ao, m2 = .8 + .2 ao^2
    assert np.allclose(npr.h2f(gi)[..., 3], exp_var, atol=2e-3)
    nanmv = motion_img.copy()
    nanmv[..., :2] = 0x7e00
    gi, gm = oracle.svgf_temporal(simple_pfd(W, H, 0), normals, nanmv, rt, normals, history, moments)
    assert np.array_equal(gi[..., :2], rt) and not gi[..., 2:].any()


def test_schedule_publishes_second_to_last_iteration(oracle):
    """hybrid_render_path.cpp:299-328: with 5 a-trous steps the published image is iteration 3's output (step 8);
    the step-16 pass is dead work; the history is iteration 0's output."""
    W, H = 48, 32
    normals, motion, rt = synthetic_svgf_inputs(W, H, seed=41)
    pfd = simple_pfd(W, H)
    st = oracle.SVGF(W, H)
    den = st.frame(pfd, normals, motion, rt)
    z4, z2 = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 2), np.uint16)
    a, _ = oracle.svgf_temporal(pfd, normals, motion, rt, z4, z4, z2)
    chain = [a]
    for i in range(5):
        chain.append(oracle.svgf_atrous(pfd, normals, chain[-1], 1 << i))
    assert np.array_equal(den, chain[4])                 # output of iteration 3
    assert np.array_equal(st.image(3), chain[1])         # history = output of iteration 0
    assert np.array_equal(st.image(2), normals)          # previous normals
    assert np.array_equal(st.image(1), chain[5])         # the dead iteration's output sits in the other ping-pong image


def test_golden_crops(oracle):
    """Committed fixtures (tests/golden/make_golden.py) keep the oracle itself from drifting."""
    g = np.load(os.path.join(GOLDEN, "svgf_crops.npz"))
    W, H = int(g["W"]), int(g["H"])
    pfd = simple_pfd(W, H)
    assert np.array_equal(oracle.svgf_atrous(pfd, g["normals"], g["integrated"], 2), g["atrous_step2"])
    ti, tm = oracle.svgf_temporal(pfd, g["normals"], g["motion"], g["raytraced"], g["prev_normals"], g["history"], g["moments"])
    assert np.array_equal(ti, g["temporal_integrated"]) and np.array_equal(tm, g["temporal_moments"])
    t = np.load(os.path.join(GOLDEN, "trace_tiny.npz"))
    sc = scenes.tiny_scene()
    osc = oracle.Scene(sc)
    pfd = camera.dolly_frames(sc, int(t["W"]), int(t["H"]), 2)[1]
    n, m, d = osc.gbuffer(pfd, int(t["W"]), int(t["H"]))
    sa, refl, mask, rays = osc.raygen(pfd, abi.default_trace_params(), n, d)
    assert np.array_equal(sa, t["shadow_ao"]) and np.array_equal(mask, t["mask"]) and rays == int(t["rays"])


def test_composition_golden_and_orientation(oracle):
    """composition.frag restatement: pinned by a committed crop; the presented image is the G-buffer flipped vertically
    (pipeline.cpp:175-178) and sky pixels come out black."""
    t = np.load(os.path.join(GOLDEN, "trace_tiny.npz"))
    c = np.load(os.path.join(GOLDEN, "composition_tiny.npz"))
    pfd = np.frombuffer(t["pfd"].tobytes(), abi.per_frame_dtype)[0]
    out = oracle.composition(pfd, (0, 0, 0), c["albedo"], t["normals"], t["motion"], t["depth"], t["shadow_ao"], t["reflections"])
    assert np.array_equal(out, c["composition"])
    sky = (t["depth"] == 0)[::-1]
    assert sky.any() and not out[sky][:, :3].any() and (out[..., 3] == 255).all()
    off = oracle.composition(pfd, (2, 2, 2), c["albedo"], t["normals"], t["motion"], t["depth"], t["shadow_ao"], t["reflections"])
    lit = oracle.composition(pfd, (0, 2, 2), c["albedo"], t["normals"], t["motion"], t["depth"], t["shadow_ao"], t["reflections"])
    assert (lit[..., :3].astype(int) <= off[..., :3].astype(int)).all() and (lit != off).any()      # shadows only darken


def test_reflection_hit_matches_float64_restatement(oracle):
    """K2 (raygen.rgen:59-65 + reflection_hit.rchit:10-72 + common.glsl:116-150) against an independent float64 numpy
    derivation from the GLSL, on the untextured tiny scene: same hit, payload equal at fp16 resolution."""
    from tests import numpy_restatement as nr
    from tests.helpers import f16
    sc = scenes.tiny_scene()
    osc = oracle.Scene(sc)
    W, H = 48, 30
    pfd = camera.dolly_frames(sc, W, H, 3)[2]
    n, m, d = osc.gbuffer(pfd, W, H)
    sa, refl, mask, rays = osc.raygen(pfd, abi.default_trace_params(), n, d)
    normals, got = f16(n), f16(refl)
    checked = 0
    mismatched = 0
    for y in range(H):
        for x in range(W):
            if d[y, x] == 0:
                assert not refl[y, x].any()                                   # raygen.rgen:20-24: sky writes vec4(0)
                continue
            o, r = nr.reflection_ray(pfd, d[y, x], normals[y, x, :3], x, y, W, H)
            hit = osc.closest(o, r, 0.01, 1e4)
            if hit is None:
                if not (mask[y, x] & 0x80):
                    assert not refl[y, x].any()                               # reflection_miss.rmiss:7
                continue
            if not (mask[y, x] & 0x80):
                continue                                                      # float64 vs float32 ray at a silhouette
            t, u, v, prim, tri = hit
            want = nr.reflection_hit(sc, pfd, prim, tri, float(u), float(v))
            tol = 3 * 2.0 ** -10 * np.maximum(np.abs(want), 2.0 ** -14)        # 3 fp16 steps: fp32 chain + the final fp16 store
            if (np.abs(got[y, x, :3] - want) <= tol).all() and got[y, x, 3] == 1.0:
                checked += 1
            else:
                mismatched += 1           # the float64 ray may pick the neighbouring triangle at an edge; a formula error would hit every pixel
    assert checked > 300 and mismatched <= checked // 100, (checked, mismatched)
