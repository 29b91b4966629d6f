"""GPU parity of the ray-tracing pass (K0 BVH build, K1 raygen, K2 closest-hit shading) against the oracle.

Bar: the RG16F shadow/AO image -- integer visibility decisions -- is BIT-EXACT; the RGBA16F reflection
colour agrees within 2 fp16 ulps (float tolerance: shading uses the same formulas but the hardware's
division / sqrt-free paths may round differently in the last fp32 bit before the fp16 store)."""
import numpy as np
import pytest

from vulkanhybridrenderer_amd import abi, camera, lib, scenes
from tests.helpers import GpuHybrid, assert_reflections_identical, f16, oracle_frames

pytestmark = pytest.mark.gpu


def _compare_trace(ob, scene, W, H, n_frames, tp):
    frames, osc, _ = oracle_frames(ob, scene, W, H, n_frames, tp, denoise=False)
    g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
    try:
        for i, fr in enumerate(frames):
            g.frame(fr["pfd"], fr["gbuf"])
            sa = g.ctx.download(lib.RAYTRACED)
            assert np.array_equal(sa, fr["shadow_ao"]), \
                f"frame {i}: visibility differs at {np.argwhere(sa != fr['shadow_ao'])[:8]} ({(sa != fr['shadow_ao']).any(-1).sum()} px)"
            assert_reflections_identical(g.ctx.download(lib.REFLECTIONS), fr["reflections"], f"frame {i}: reflections")
    finally:
        g.close()


def test_tiny_scene_all_ray_kinds(oracle):
    tp = abi.default_trace_params()
    _compare_trace(oracle, scenes.tiny_scene(), 96, 64, 3, tp)


def test_tiny_scene_matches_brute_force_oracle(oracle):
    """The GPU (SAH BVH) result equals the oracle's brute-force result: BVH culling changes nothing."""
    scene = scenes.tiny_scene()
    W, H = 64, 48
    tp = abi.default_trace_params()
    osc = oracle.Scene(scene)
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    gbuf = osc.gbuffer(pfd, W, H)
    sa, refl, _, _ = osc.raygen(pfd, tp, gbuf[0], gbuf[2], use_bvh=False)
    g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
    try:
        g.frame(pfd, gbuf)
        assert np.array_equal(g.ctx.download(lib.RAYTRACED), sa)
    finally:
        g.close()


def test_sponza_proc_quarter_res(oracle):
    tp = abi.default_trace_params()
    _compare_trace(oracle, scenes.sponza_proc(), 480, 270, 3, tp)


def test_ao_spp_extension(oracle):
    tp = abi.default_trace_params(ao_spp=4, reflections=False)
    _compare_trace(oracle, scenes.tiny_scene(), 80, 48, 2, tp)


def test_ray_statistics(oracle):
    scene = scenes.tiny_scene()
    W, H = 96, 64
    tp = abi.default_trace_params()
    frames, _, _ = oracle_frames(oracle, scene, W, H, 2, tp, denoise=False)
    g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
    try:
        g.ctx.set_ray_statistics(True)
        g.frame(frames[1]["pfd"], frames[1]["gbuf"])
        st = g.ctx.ray_statistics()
        covered = int((frames[1]["gbuf"][2] != 0).sum())
        assert st["covered_pixels"] == covered
        assert st["unique_rays"] == frames[1]["rays"] == covered * 4
        assert st["reference_issued_rays"] == covered * 7      # 4 duplicate shadow + 2 AO + 1 reflection
        assert st["stack_overflows"] == 0
    finally:
        g.close()


@pytest.mark.parametrize("scene_name,detail,size", [("tiny_scene", None, (96, 64)), ("sponza_proc", 0.3, (480, 270)), ("bistro_proc", 0.2, (480, 270))])
def test_standin_gbuffer_is_the_oracles_bit_for_bit(oracle, scene_name, detail, size):
    """The stand-in G-buffer producer (gbuf.vert:19-28, gbuf.frag:17-59 through primary rays) against the oracle's producer: normals / object ids,
    motion / metallic / roughness, depth and albedo IDENTICAL, every texel, signed zeros included -- the producer shares the intersector, decision
    (vi) and every rounding with the oracle.  (Rounds 1-5 allowed 0.5 % of the pixels on another surface and 4e-3 on the normals; what was left in
    round 6 were 142 of 2 M normals one fp16 step off: the compiler had folded the normalisation's last multiply into the fp16 conversion --
    one rounding instead of decision (iii)'s two; device_math.hpp float_to_half_bits pins the fp32 value now.)"""
    maker = getattr(scenes, scene_name)
    scene = maker() if detail is None else maker(detail)
    W, H = size
    osc = oracle.Scene(scene)
    pfds = camera.dolly_frames(scene, W, H, 2)
    ctx = lib.Context(W, H)
    ctx.upload_scene(scene)
    path = lib.HybridRenderPath(ctx, 0, 0, 2, False, 5, lambda c: c.standin_gbuffer_with_albedo(0))
    path.build()
    try:
        for pfd in pfds:
            ctx.update_per_frame_ubo(0, pfd)
            ctx.execute(0, 0)
            ctx.synchronize()
        want = osc.gbuffer(pfds[1], W, H, with_albedo=True)
        got = [ctx.download(k) for k in (lib.NORMALS, lib.MOTION, lib.DEPTH, lib.ALBEDO)]
        for name, g_img, w_img in zip(("normals / ids", "motion / metallic / roughness", "depth", "albedo"), got, want):
            g_img, w_img = np.asarray(g_img), np.asarray(w_img)
            same = g_img.view(np.uint32) == w_img.view(np.uint32) if g_img.dtype == np.float32 else g_img == w_img
            assert same.all(), f"{name}: {int((~same).reshape(H, W, -1).any(-1).sum())} texels differ, first at {np.argwhere(~same)[:3].tolist()}"
    finally:
        path.destroy()
        ctx.close()


@pytest.mark.parametrize("scene_name,W,H", [("tiny", 100, 37), ("sponza", 480, 270), ("bistro_small", 200, 120)])
def test_mirror_ray_kernels_bit_identical(oracle, scene_name, W, H):
    """The work-queue mirror-ray kernel (reflection_variant 1, default) and the one-pixel-per-thread kernel (0) trace the same
    rays with the same arithmetic and shade with the same code: their RGBA16F images are equal bit for bit, at sizes that are
    not multiples of the 16x8 tile, with and without an LDS-only stack."""
    scene = {"tiny": scenes.tiny_scene, "sponza": scenes.sponza_proc,
             "bistro_small": lambda: scenes.bistro_proc(detail=0.02, n_primitives=300, n_textures=8, texture_size=64)}[scene_name]()
    g = GpuHybrid(scene, W, H, denoise=False, trace_params=abi.default_trace_params(), gbuffer="standin")
    try:
        g.ctx.set_ray_statistics(True)
        for bounces in (1, 2):                                    # 2 = the two-bounce extension (tests/test_two_bounce.py)
            g.ctx.set_trace_params(abi.default_trace_params(reflections=bounces))
            for pfd in camera.dolly_frames(scene, W, H, 3):
                out, rays = {}, {}
                for variant, levels in ((0, 8), (1, 8), (1, 32), (1, 2)):
                    g.ctx.set_option("reflection_variant", variant)
                    g.ctx.set_option("lds_stack_levels", levels)
                    g.ctx.set_option("reflection_lds_stack_levels", levels)       # the mirror ray's walk has its own count (r5)
                    g.frame(pfd)
                    out[(variant, levels)] = g.ctx.download(lib.REFLECTIONS)
                    st = g.ctx.ray_statistics()
                    assert st["stack_overflows"] == 0
                    rays[(variant, levels)] = st["unique_rays"]
                for k, img in out.items():
                    assert np.array_equal(img, out[(0, 8)]), f"{bounces} bounce(s), reflection_variant {k}: {(img != out[(0, 8)]).any(-1).sum()} pixels differ"
                    assert rays[k] == rays[(0, 8)]
            assert (f16(out[(0, 8)])[..., 3] > 0).mean() > 0.1
    finally:
        g.ctx.set_option("lds_stack_levels", 8)
        g.ctx.set_option("reflection_lds_stack_levels", lib.option_table()["reflection_lds_stack_levels"][0])
        g.close()
