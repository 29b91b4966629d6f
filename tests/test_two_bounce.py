"""BASELINE config 5's `2-bounce mirror reflections` (trace parameter reflections = 2): a documented extension -- the
reference traces one bounce (raygen.rgen:59-65) and only declares recursion depth 2 (pipeline.cpp:285).  Semantics (oracle
header): a mirror ray from the first hit about reflection_hit.rchit's N, shaded by the same closest-hit shader without
recursion, blended into the first hit's specular term the way composition.frag:141-149 blends the first bounce."""
import numpy as np
import pytest

from tests.helpers import GpuHybrid, assert_reflections_identical, f16
from vulkanhybridrenderer_amd import abi, camera, lib, scenes
from vulkanhybridrenderer_amd.camera import directional_light
from vulkanhybridrenderer_amd.scenes import _Builder, plane

W, H = 96, 60
MIRROR_ALBEDO = (0.9, 0.6, 0.3)


def mirror_scene():
    """A floor and a metallic == 1 wall facing the camera; everything else is sky, so a second bounce off the wall escapes."""
    b = _Builder()
    b.add(plane([-6, 0, 6], [12, 0, 0], [0, 0, -9], 2, 2), base_color=(0.7, 0.7, 0.7, 1.0), roughness=0.5)          # floor, faces +y
    b.add(plane([-6, 0, -3], [12, 0, 0], [0, 6, 0], 2, 2), base_color=MIRROR_ALBEDO + (1.0,), metallic=1.0, roughness=0.2)   # wall, faces +z
    cam = dict(position=(0.0, 1.5, 5.0), yaw=0.0, pitch=-0.25, yfov=0.9, znear=0.1, dolly=(0.0, 0.0, -0.05))
    return b.finish("mirror", cam, directional_light((0.1, -0.9, -0.4)), [])


def _render(oracle, sc, bounces, use_bvh=True):
    osc = oracle.Scene(sc)
    pfd = camera.dolly_frames(sc, W, H, 2)[1]
    gb = osc.gbuffer(pfd, W, H)
    tp = abi.default_trace_params(reflections=bounces)
    sa, refl, mask, rays = osc.raygen(pfd, tp, gb[0], gb[2], use_bvh=use_bvh)
    return pfd, gb, sa, refl, mask, rays


def test_oracle_second_bounce_semantics(oracle):
    sc = mirror_scene()
    pfd, gb, sa1, r1, m1, rays1 = _render(oracle, sc, 1)
    _, _, sa2, r2, m2, rays2 = _render(oracle, sc, 2)
    assert np.array_equal(sa1, sa2)                                           # visibility rays are untouched
    hit1 = (m1 & 0x80) != 0
    assert rays2 == rays1 + int(hit1.sum())                                   # one more ray per first-bounce hit
    assert np.array_equal(r1[~hit1], r2[~hit1])                               # a first bounce that escapes stays reflection_miss (0)
    # floor pixels whose mirror ray meets the metallic wall: the second bounce leaves towards the sky (payload 0), so the
    # specular term is replaced by 0 (metallic == 1 -> specular = reflections) and diffuse vanishes ((1 - metallic) == 0):
    # payload = albedo * 0.2 / pi exactly
    ids = np.where(gb[2] != 0, f16(gb[0])[..., 3], -1)
    sel = (ids == 0) & hit1
    assert sel.sum() > 200
    want = (np.asarray(MIRROR_ALBEDO, np.float32) * np.float32(0.31830988618379067 * 0.2)).astype(np.float16).astype(np.float32)
    got = f16(r2)[sel][:, :3]
    assert np.abs(got - want).max() <= 2.0 ** -11 * want.max() * 2
    assert (f16(r2)[sel][:, 3] == 1.0).all()
    assert (f16(r1)[sel][:, :3].sum(-1) > got.sum(-1)).mean() > 0.9            # with one bounce the wall also carries its specular highlight
    # brute force == BVH for the chained rays too
    _, _, _, r2b, _, rays2b = _render(oracle, sc, 2, use_bvh=False)
    assert np.array_equal(r2, r2b) and rays2 == rays2b


def test_oracle_second_bounce_sees_geometry(oracle):
    """tiny_scene: the second bounce finds surfaces, so two-bounce payloads differ from one-bounce ones where the first hit is rough."""
    sc = scenes.tiny_scene()
    _, _, _, r1, m1, _ = _render(oracle, sc, 1)
    _, _, _, r2, m2, _ = _render(oracle, sc, 2)
    hit = (m1 & 0x80) != 0
    assert np.array_equal(m1, m2)
    assert (r1[hit] != r2[hit]).any(-1).mean() > 0.2                          # equal where the second ray escapes AND the first hit is unlit
    assert np.isfinite(f16(r2)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["mirror", "tiny", "bistro_small"])
def test_gpu_two_bounces_match_oracle(oracle, scene_name):
    sc = {"mirror": mirror_scene, "tiny": scenes.tiny_scene,
          "bistro_small": lambda: scenes.bistro_proc(detail=0.02, n_primitives=300, n_textures=8, texture_size=64)}[scene_name]()
    osc = oracle.Scene(sc)
    tp = abi.default_trace_params(reflections=2)
    g = GpuHybrid(sc, W, H, denoise=False, trace_params=tp)
    g.ctx.set_ray_statistics(True)
    try:
        for variant in (1, 0):                                                 # work-queue + reflection_kernel, literal raygen_kernel
            g.ctx.set_option("raygen_variant", variant)
            for pfd in camera.dolly_frames(sc, W, H, 3)[1:]:
                gb = osc.gbuffer(pfd, W, H)
                sa, refl, mask, rays = osc.raygen(pfd, tp, gb[0], gb[2])
                g.frame(pfd, gb)
                assert np.array_equal(g.ctx.download(lib.RAYTRACED), sa)
                assert_reflections_identical(g.ctx.download(lib.REFLECTIONS), refl)        # both bounces: the second ray's origin is the first hit's, bit for bit
                st = g.ctx.ray_statistics()
                assert abs(int(st["unique_rays"]) - rays) <= 2 and st["stack_overflows"] == 0
    finally:
        g.close()
