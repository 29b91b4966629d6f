"""Next row f3: the composition stand-in (composition.frag:60-161, ray-traced modes) on the GPU vs the oracle, fed by
the full hybrid path; also the stand-in G-buffer's albedo attachment.  Tolerance: the sRGB attachment is 8 bits, the
GPU encodes with powf in fp32 and the oracle in double -> at most 1 code value apart, >= 99 % identical."""
import numpy as np
import pytest

from tests.helpers import oracle_frames
from vulkanhybridrenderer_amd import abi, camera, lib, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("modes,denoise", [((0, 0, 0), True), ((0, 2, 2), False), ((2, 0, 0), True), ((2, 2, 2), False)])
def test_composition_matches_oracle(oracle, modes, denoise):
    scene = scenes.bistro_proc(detail=0.02, n_primitives=300, n_textures=6, texture_size=32)
    W, H = 128, 72
    tp = abi.default_trace_params(shadow=modes[0] == 0, ao_spp=2 if modes[1] == 0 else 0, reflections=modes[2] == 0)
    any_rt = 0 in modes
    osc = oracle.Scene(scene)
    svgf = oracle.SVGF(W, H)
    ctx = lib.Context(W, H)
    ctx.upload_scene(scene)
    ctx.set_trace_params(tp)
    state = {}

    def gbuffer_pass(c):
        n, m, d, al = state["gbuf"]
        c.upload(lib.NORMALS, n); c.upload(lib.MOTION, m); c.upload(lib.DEPTH, d); c.upload(lib.ALBEDO, al)

    out_img = ctx.upload_new_storage_image(W, H, abi.FORMAT_B8G8R8A8_SRGB)

    def composition_pass(c):
        src = lib.DENOISED if (denoise and any_rt) else lib.RAYTRACED
        c.standin_composition(out_img, modes[0], modes[1], modes[2], shadow_ao=src, reflections=lib.REFLECTIONS if any_rt else None)

    path = lib.HybridRenderPath(ctx, modes[0], modes[1], modes[2], denoise, 5, gbuffer_pass, composition_pass)
    path.build()
    try:
        for pfd in camera.dolly_frames(scene, W, H, 4):
            state["gbuf"] = osc.gbuffer(pfd, W, H, with_albedo=True)
            n, m, d, al = state["gbuf"]
            sa, refl, _, _ = osc.raygen(pfd, tp, n, d)
            den = svgf.frame(pfd, n, m, sa)
            ctx.update_per_frame_ubo(0, pfd)
            ctx.execute(0, 0)
            ctx.synchronize()
            if not any_rt:
                sa = np.zeros_like(sa)
            ref = oracle.composition(pfd, modes, al, n, m, d, den if (denoise and any_rt) else sa, refl)
            got = ctx.download(out_img)
            diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1, f"composition off by {diff.max()} code values at {np.argwhere(diff > 1)[:4]}"
            assert (diff == 0).mean() > 0.99
        assert got[..., :3].mean() > 5          # an actual picture, not black
    finally:
        path.destroy()
        ctx.close()


def test_standin_gbuffer_albedo_matches_oracle(oracle):
    scene = scenes.bistro_proc(detail=0.02, n_primitives=300, n_textures=6, texture_size=32)
    W, H = 96, 64
    osc = oracle.Scene(scene)
    ctx = lib.Context(W, H)
    ctx.upload_scene(scene)
    path = lib.HybridRenderPath(ctx, 0, 2, 2, False, 5, lambda c: c.standin_gbuffer_with_albedo(0))
    path.build()
    try:
        for pfd in camera.dolly_frames(scene, W, H, 2):
            ctx.update_per_frame_ubo(0, pfd)
            ctx.execute(0, 0)
            ctx.synchronize()
        n0, m0, d0, al0 = osc.gbuffer(pfd, W, H, with_albedo=True)
        al, n = ctx.download(lib.ALBEDO), ctx.download(lib.NORMALS)
        same = n[..., 3] == n0[..., 3]                      # same primitive hit (silhouette pixels may differ)
        assert same.mean() > 0.99
        d = np.abs(al.astype(np.int32) - al0.astype(np.int32))[same]
        assert (d <= 1).mean() > 0.995 and np.median(d) == 0      # textured surfaces: bilinear taps may round differently
    finally:
        path.destroy()
        ctx.close()
