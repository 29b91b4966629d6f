"""SURVEY.md section 8 row f4, second half: the screen-space alternatives of the hybrid render path -- ssao.comp,
ssao_blur.comp, ssr.comp (hybrid_render_path.cpp:138-243) and their use by composition.frag:114-156.

CPU: the oracle against hand-derived values and against the independent float64 numpy restatement; the host graph.
GPU (-m gpu): the HIP kernels through the C ABI against the oracle.  Bar: the kernels use the oracle's operation order
with contraction off, so every RGBA16F texel is expected to be bit-identical; the tests allow ssao / ssr a last-place
disagreement on < 0.1 % of the texels (none observed) and demand exact equality for the blur."""
import numpy as np
import pytest

from tests import numpy_restatement as nr
from tests.helpers import f16
from vulkanhybridrenderer_amd import abi, camera, lib, scenes

F4 = abi.FORMAT_R16G16B16A16_SFLOAT


def _small_scene():
    return scenes.bistro_proc(detail=0.02, n_primitives=300, n_textures=6, texture_size=32)


# ---------------------------------------------------------------------------------------------------------
# oracle: known answers
# ---------------------------------------------------------------------------------------------------------
def test_default_sampler_known_values(oracle):
    """decision (x): LINEAR / REPEAT at LOD 0 -- a texel centre returns the texel, a texel corner the mean of the four
    texels around it, coordinates outside [0, 1) wrap."""
    W, H = 4, 3
    d = (np.arange(W * H, dtype=np.float32).reshape(H, W) + 1.0) / 16.0        # exactly representable
    for y in range(H):
        for x in range(W):
            assert oracle.sample_linear_repeat(d, (x + 0.5) / W, (y + 0.5) / H)[0] == d[y, x]
    assert oracle.sample_linear_repeat(d, 1.0 / W, 1.0 / H)[0] == (d[0, 0] + d[0, 1] + d[1, 0] + d[1, 1]) / 4
    # the image corner (0, 0): the four corner texels through the wrap; what ssao.comp / ssr.comp fetch for pixel (0, 0)
    assert oracle.sample_linear_repeat(d, 0.0, 0.0)[0] == (d[0, 0] + d[0, W - 1] + d[H - 1, 0] + d[H - 1, W - 1]) / 4
    assert oracle.sample_linear_repeat(d, 1.0 + 1.5 / W, -1.0 + 0.5 / H)[0] == d[0, 1]
    rgba = np.zeros((H, W, 4), np.uint16)
    rgba[..., 0] = nr.f2h(d)
    rgba[..., 3] = nr.f2h(np.ones_like(d))
    assert np.array_equal(oracle.sample_linear_repeat(rgba, 2.0 / W, 1.0 / H), [(d[0, 1] + d[0, 2] + d[1, 1] + d[1, 2]) / 4, 0, 0, 1])
    bgra = np.zeros((H, W, 4), np.uint8)
    bgra[..., 0], bgra[..., 2] = 255, 51                                         # b = 1.0, r = 0.2
    assert np.allclose(oracle.sample_linear_repeat(bgra, 0.3, 0.7), [0.2, 0, 1.0, 0], atol=1e-7)


def _facing_plane(W, H, frame_index=3):
    """A G-buffer the shaders can be evaluated on by hand: one plane of constant view depth (constant NDC depth under the
    infinite reverse projection) whose normal points at the camera, the top rows sky."""
    drv = camera.FrameDriver(W, H, 0.9, 0.1, camera.directional_light((0.0, -0.97, 0.35)))
    drv.frame_index = frame_index
    pfd = drv.next((0.0, 1.0, 0.0))
    view_inv = np.asarray(pfd["camera_view_inverse"], np.float32).reshape(4, 4).T
    toward_camera = view_inv[:3, 2]                         # the camera looks down -z of its own frame
    normals = np.zeros((H, W, 4), np.uint16)
    normals[..., :3] = nr.f2h(np.broadcast_to(toward_camera, (H, W, 3)))
    depth = np.full((H, W), 0.05, np.float32)               # znear / 0.05 = 2 m in front of the camera
    return pfd, normals, depth


def test_ssao_of_a_plane_facing_the_camera_is_one(oracle):
    """Every sample of ssao.comp lies in the plane of P, so dot(V, N) = 0, max(0 - beta, 0) = 0 and ao = 1 exactly; pixels
    whose bilinear depth footprint touches no geometry write 0 (ssao.comp:17-24); a footprint half in the sky sees a depth
    discontinuity and darkens."""
    W, H = 48, 32
    pfd, normals, depth = _facing_plane(W, H)
    raw = f16(oracle.ssao(pfd, normals, depth))
    assert (raw == 1.0).all()
    depth[:8] = 0.0
    depth[H - 1] = 0.0                                      # row 0 samples rows H-1 and 0 (REPEAT): sky only when both are
    raw = f16(oracle.ssao(pfd, normals, depth))
    assert (raw[0] == 0.0).all() and (raw[1:8] == 0.0).all()
    assert (raw[12:H - 8] >= 0.0).all() and np.isfinite(raw).all()
    assert (raw[..., 0] == raw[..., 3]).all()               # decision (xi): vec4(ao) lands in all four components
    blurred = f16(oracle.ssao_blur(pfd, oracle.ssao(pfd, normals, np.full((H, W), 0.05, np.float32))))
    assert blurred[16, 24, 0] == 1.0                        # 169 ones / 169
    assert blurred[0, 0, 0] == np.float16(49.0 / 169.0)     # a 7 x 7 corner of ones, still divided by 169 (ssao_blur.comp:25)


def test_oracle_matches_the_numpy_restatement(oracle):
    scene = _small_scene()
    W, H = 120, 68
    pfd = camera.dolly_frames(scene, W, H, 3)[2]
    normals, motion, depth, albedo = oracle.Scene(scene).gbuffer(pfd, W, H, with_albedo=True)
    raw = oracle.ssao(pfd, normals, depth)
    d = np.abs(f16(raw)[..., 0] - nr.ssao(pfd, normals, depth))
    assert d.max() < 2e-2 and (d > 2e-3).mean() < 2e-3       # fp32 + polynomial sin/cos there, float64 + libm here
    blurred = oracle.ssao_blur(pfd, raw)
    mine = nr.ssao_blur_f32(raw, float(pfd["display_size"][0]), float(pfd["display_size"][1]))
    assert np.array_equal(nr.f2h(mine), blurred[..., 0])     # same fp32 summation order: the same bits
    refl = f16(oracle.ssr(pfd, albedo, normals, motion, depth))
    found, lit, _ = nr.ssr(pfd, albedo, normals, motion, depth)
    assert 0.1 < found.mean() < 0.9
    assert ((refl[..., 3] == 1.0) == found).mean() > 0.998   # a march step may flip where delta sits on a threshold
    both = (refl[..., 3] == 1.0) & found
    rel = np.abs(refl[..., :3] - lit)[both] / np.maximum(np.abs(lit[both]), 1e-2)
    assert np.median(rel) < 1e-3 and (rel > 2e-2).mean() < 5e-3
    assert (refl[~(refl[..., 3] == 1.0)] == 0.0).all()       # ssr.comp:62-66: misses keep the cleared texel


def _golden():
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    t, c, s = (np.load(os.path.join(g, f)) for f in ("trace_tiny.npz", "composition_tiny.npz", "screen_space_tiny.npz"))
    pfd = np.frombuffer(t["pfd"].tobytes(), abi.per_frame_dtype)[0]
    return pfd, (t["normals"], t["motion"], t["depth"], c["albedo"]), s


def test_golden_fixture(oracle):
    """tests/golden/screen_space_tiny.npz (make_golden.py) keeps the restatement from drifting."""
    pfd, (n, m, d, al), s = _golden()
    raw = oracle.ssao(pfd, n, d)
    assert np.array_equal(raw, s["ssao_raw"]) and np.array_equal(oracle.ssao_blur(pfd, raw), s["ssao"])
    assert np.array_equal(oracle.ssr(pfd, al, n, m, d), s["ssr"])
    assert 0.05 < (f16(s["ssr"])[..., 3] == 1.0).mean() < 0.95 and 0.2 < f16(s["ssao"])[..., 0].mean() < 1.0


def test_oracle_rows_are_independent(oracle):
    scene = _small_scene()
    W, H = 64, 40
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    normals, motion, depth, albedo = oracle.Scene(scene).gbuffer(pfd, W, H, with_albedo=True)
    whole = oracle.ssao(pfd, normals, depth), oracle.ssr(pfd, albedo, normals, motion, depth)
    part = oracle.ssao(pfd, normals, depth, rows=(10, 23)), oracle.ssr(pfd, albedo, normals, motion, depth, rows=(10, 23))
    for w, p in zip(whole, part):
        assert np.array_equal(w[10:23], p[10:23]) and not p[:10].any() and not p[23:].any()


# ---------------------------------------------------------------------------------------------------------
# host graph (no GPU)
# ---------------------------------------------------------------------------------------------------------
def test_hybrid_path_registers_the_screen_space_passes(vhr):
    ctx = lib.Context(64, 64, host_only=True)
    try:
        p = lib.HybridRenderPath(ctx, shadow_mode=2, ambient_occlusion_mode=1, reflection_mode=1, denoise=False)
        p.build()
        order = ctx.execution_order()
        # FindExecutionOrder walks the composition's dependency list back to front (render_graph.cpp:686-720)
        assert order[0] == "G-Buffer Pass" and order[-1] == "Composition Pass"
        assert order.index("SSAO Pass") < order.index("SSAO Blur Pass") and "SSR Pass" in order and "Raytrace Pass" not in order
        for name in (lib.SSAO_RAW, lib.SSAO, lib.SSR):
            assert ctx.image_format(name) == F4               # R16G16B16A16 although the shaders say r16f (:149,176,219)
        p.rebuild(shadow_mode=0, denoise_shadow_and_ao=1)   # ray-traced shadows + SVGF next to SSAO / SSR
        order = ctx.execution_order()
        assert {"Raytrace Pass", "SVGF Denoise Pass", "SSAO Pass", "SSAO Blur Pass", "SSR Pass"} <= set(order)
        p.rebuild(ambient_occlusion_mode=2, reflection_mode=2)
        assert "SSAO Pass" not in ctx.execution_order() and not ctx.contains_image(lib.SSAO_RAW)
        p.destroy()
    finally:
        ctx.close()


def test_dispatch_rules_of_the_screen_space_kernels(vhr):
    ctx = lib.Context(32, 32, host_only=True)
    try:
        noop = lambda ec: None   # noqa: E731
        ctx.add_compute_pass("SSAO Pass", [lib.transient(lib.NORMALS, F4, 0, lib.SAMPLED_IMAGE), lib.transient(lib.DEPTH, abi.FORMAT_D32_SFLOAT, 1, lib.SAMPLED_IMAGE)],
                             [lib.transient(lib.SSAO_RAW, F4, 2)], [lib.SSAO_SHADER], 0, noop)
        with pytest.raises(lib.VhrError, match="already registered by pass"):
            ctx.add_compute_pass("again", [], [], [lib.SSAO_SHADER], 0, noop)
        ctx.add_compute_pass("SSR Pass", [], [lib.transient(lib.SSR, F4, 4)], [lib.SSR_SHADER], 16, noop)
        with pytest.raises(lib.VhrError, match="no HIP kernel"):
            ctx.add_compute_pass("C", [], [], ["hybrid_render_path/depth_prepass.comp"], 0, noop)
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------
class ScreenSpacePath:
    """HybridRenderPath with SSAO + SSR (and optionally ray-traced shadows) on host-supplied G-buffers."""

    def __init__(self, scene, W, H, shadow_mode=2, denoise=False, composition=True):
        self.ctx = lib.Context(W, H)
        self.ctx.upload_scene(scene)
        self.gbuf = None
        self.modes = (shadow_mode, 1, 1)
        self.out_img = self.ctx.upload_new_storage_image(W, H, abi.FORMAT_B8G8R8A8_SRGB)
        self.denoise = denoise

        def gbuffer_pass(c):
            n, m, d, al = self.gbuf
            c.upload(lib.NORMALS, n); c.upload(lib.MOTION, m); c.upload(lib.DEPTH, d); c.upload(lib.ALBEDO, al)

        def composition_pass(c):
            src = lib.DENOISED if (denoise and shadow_mode == 0) else lib.RAYTRACED
            c.standin_composition(self.out_img, shadow_mode, 1, 1, shadow_ao=src, reflections=lib.SSR, ssao=lib.SSAO)

        self.path = lib.HybridRenderPath(self.ctx, shadow_mode, 1, 1, denoise, 5, gbuffer_pass, composition_pass if composition else None)
        self.path.build()

    def frame(self, pfd, gbuf):
        self.gbuf = gbuf
        self.ctx.update_per_frame_ubo(0, pfd)
        self.ctx.execute(0, 0)
        self.ctx.synchronize()

    def close(self):
        self.path.destroy()
        self.ctx.close()


def _bits_close(got, want, what, exact=False):
    same = (got == want).all(-1)
    if exact:
        assert same.all(), f"{what}: {(~same).sum()} texels differ"
        return
    assert same.mean() > 0.999, f"{what}: only {same.mean():.5f} of the texels identical"
    d = np.abs(f16(got).astype(np.float64) - f16(want))
    assert np.nanmax(d[same == False], initial=0.0) < 0.05 or (~same).mean() < 2e-4, what   # noqa: E712


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,W,H", [("bistro_small", 200, 120), ("sponza", 150, 77), ("tiny", 64, 40)])
def test_gpu_kernels_match_the_oracle(oracle, scene_name, W, H):
    scene = {"bistro_small": _small_scene, "sponza": scenes.sponza_proc, "tiny": scenes.tiny_scene}[scene_name]()
    osc = oracle.Scene(scene)
    g = ScreenSpacePath(scene, W, H)
    try:
        for pfd in camera.dolly_frames(scene, W, H, 3)[1:]:
            gbuf = osc.gbuffer(pfd, W, H, with_albedo=True)
            n, m, d, al = gbuf
            g.frame(pfd, gbuf)
            raw = oracle.ssao(pfd, n, d)
            _bits_close(g.ctx.download(lib.SSAO_RAW), raw, "ssao.comp")
            got_raw = g.ctx.download(lib.SSAO_RAW)
            # the blur of the GPU's own raw image must equal the oracle's blur of it bit for bit (same summation order)
            assert np.array_equal(g.ctx.download(lib.SSAO), oracle.ssao_blur(pfd, got_raw)), "ssao_blur.comp"
            _bits_close(g.ctx.download(lib.SSR), oracle.ssr(pfd, al, n, m, d), "ssr.comp")
            ref = oracle.composition(pfd, (2, 1, 1), al, n, m, d, np.zeros((H, W, 2), np.uint16), g.ctx.download(lib.SSR), ssao=g.ctx.download(lib.SSAO))
            diff = np.abs(g.ctx.download(g.out_img).astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1 and (diff == 0).mean() > 0.99
        refl = f16(g.ctx.download(lib.SSR))
        assert 0.05 < (refl[..., 3] == 1.0).mean() < 0.95 and f16(g.ctx.download(lib.SSAO))[..., 0].mean() > 0.3
    finally:
        g.close()


@pytest.mark.gpu
def test_gpu_reproduces_the_golden_fixture():
    """The committed fixture through the C ABI (no oracle at run time): every texel of the three images."""
    pfd, gbuf, s = _golden()
    H, W = gbuf[2].shape
    g = ScreenSpacePath(scenes.tiny_scene(), W, H, composition=False)
    try:
        g.frame(pfd, gbuf)
        for name, key in ((lib.SSAO_RAW, "ssao_raw"), (lib.SSAO, "ssao"), (lib.SSR, "ssr")):
            assert np.array_equal(g.ctx.download(name), s[key]), name
    finally:
        g.close()


@pytest.mark.gpu
def test_gpu_settings_and_the_radius_quirk(oracle):
    """SSRPushConstants reach ssr.comp at dispatch time; ssao.comp is dispatched without constants and sees the radius the
    blur pass pushed last (0.75 before that): a caller that registers its own passes can set it that way."""
    scene = _small_scene()
    W, H = 96, 64
    osc = oracle.Scene(scene)
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    n, m, d, al = osc.gbuffer(pfd, W, H, with_albedo=True)
    ctx = lib.Context(W, H)
    state = {"radius": 0.75, "ssr": (25.0, 0.1, 0.5, 10)}
    try:
        def gbuffer_pass(c):
            c.upload(lib.NORMALS, n); c.upload(lib.MOTION, m); c.upload(lib.DEPTH, d); c.upload(lib.ALBEDO, al)
        ctx.add_graphics_pass("G-Buffer Pass", [], [lib.transient(lib.ALBEDO, abi.FORMAT_B8G8R8A8_UNORM, 0, lib.ATTACHMENT_IMAGE), lib.transient(lib.NORMALS, F4, 1, lib.ATTACHMENT_IMAGE),
                                                    lib.transient(lib.MOTION, F4, 2, lib.ATTACHMENT_IMAGE), lib.transient(lib.DEPTH, abi.FORMAT_D32_SFLOAT, 3, lib.ATTACHMENT_IMAGE)], gbuffer_pass)
        ctx.add_compute_pass("SSAO Pass", [lib.transient(lib.NORMALS, F4, 0, lib.SAMPLED_IMAGE), lib.transient(lib.DEPTH, abi.FORMAT_D32_SFLOAT, 1, lib.SAMPLED_IMAGE)],
                             [lib.transient(lib.SSAO_RAW, F4, 2)], [lib.SSAO_SHADER], 0, lambda ec: ec.dispatch(lib.SSAO_SHADER, W // 8, H // 8, 1))
        ctx.add_compute_pass("SSAO Blur Pass", [lib.transient(lib.SSAO_RAW, F4, 0)], [lib.transient(lib.SSAO, F4, 1)], [lib.SSAO_BLUR_SHADER], 4,
                             lambda ec: ec.dispatch(lib.SSAO_BLUR_SHADER, W // 8, H // 8, 1, np.float32([state["radius"]])))
        ctx.add_compute_pass("SSR Pass", [lib.transient(lib.ALBEDO, abi.FORMAT_B8G8R8A8_UNORM, 0, lib.SAMPLED_IMAGE), lib.transient(lib.NORMALS, F4, 1, lib.SAMPLED_IMAGE),
                                          lib.transient(lib.MOTION, F4, 2, lib.SAMPLED_IMAGE), lib.transient(lib.DEPTH, abi.FORMAT_D32_SFLOAT, 3, lib.SAMPLED_IMAGE)],
                             [lib.transient(lib.SSR, F4, 4)], [lib.SSR_SHADER], 16,
                             lambda ec: ec.dispatch(lib.SSR_SHADER, W // 8, H // 8, 1, np.array([state["ssr"]], dtype=[("a", "f4"), ("b", "f4"), ("c", "f4"), ("d", "i4")])))
        ctx.add_graphics_pass("Sink", [lib.transient(lib.SSAO, F4, 5, lib.SAMPLED_IMAGE), lib.transient(lib.SSR, F4, 6, lib.SAMPLED_IMAGE)], [lib.render_output(0)])
        ctx.build()
        ctx.update_per_frame_ubo(0, pfd)
        ctx.execute(0, 0); ctx.synchronize()
        assert np.array_equal(ctx.download(lib.SSAO_RAW), oracle.ssao(pfd, n, d, radius=0.75))
        state["radius"], state["ssr"] = 2.0, (8.0, 0.25, 1.5, 4)
        ctx.execute(0, 0); ctx.synchronize()               # this frame's ssao still ran with 0.75: the blur pushes after it
        assert np.array_equal(ctx.download(lib.SSAO_RAW), oracle.ssao(pfd, n, d, radius=0.75))
        assert np.array_equal(ctx.download(lib.SSR), oracle.ssr(pfd, al, n, m, d, 8.0, 0.25, 1.5, 4))
        ctx.execute(0, 0); ctx.synchronize()
        assert np.array_equal(ctx.download(lib.SSAO_RAW), oracle.ssao(pfd, n, d, radius=2.0))
        assert not np.array_equal(oracle.ssao(pfd, n, d, radius=2.0), oracle.ssao(pfd, n, d, radius=0.75))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_strips_compose_the_whole_frame(oracle):
    """Row strips (vhr_set_strip) on whole input images: the owned rows of every strip equal the single-context image."""
    scene = _small_scene()
    W, H = 128, 72
    osc = oracle.Scene(scene)
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    gbuf = osc.gbuffer(pfd, W, H, with_albedo=True)
    whole = ScreenSpacePath(scene, W, H, composition=False)
    try:
        whole.frame(pfd, gbuf)
        want = {k: whole.ctx.download(k) for k in (lib.SSAO, lib.SSR)}
    finally:
        whole.close()
    got = {k: np.zeros_like(v) for k, v in want.items()}
    for r0, r1 in ((0, 19), (19, 50), (50, 72)):
        part = ScreenSpacePath(scene, W, H, composition=False)
        try:
            part.ctx.set_strip(r0, r1, 0, 0)
            part.frame(pfd, gbuf)
            for k in got:
                got[k][r0:r1] = part.ctx.download(k)[r0:r1]
        finally:
            part.close()
    for k in got:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.gpu
def test_gpu_config0_512_square_screen_space_whole_frame(oracle):
    """BASELINE.json configs[0] ("Sponza 512x512, raster-only path: shadow map + Alchemy SSAO", the reference's CPU-runnable
    plumbing case) as far as this build reaches: sponza_proc at 512x512 with SSAO + blur (+ SSR) and nothing ray traced --
    whole frames against the oracle; the shadow map is raster work and stays outside (composition with shadows off)."""
    scene = scenes.sponza_proc()
    W = H = 512
    osc = oracle.Scene(scene)
    g = ScreenSpacePath(scene, W, H)
    try:
        assert "Raytrace Pass" not in g.ctx.execution_order()
        for pfd in camera.dolly_frames(scene, W, H, 2):
            gbuf = osc.gbuffer(pfd, W, H, with_albedo=True)
            n, m, d, al = gbuf
            g.frame(pfd, gbuf)
            raw = g.ctx.download(lib.SSAO_RAW)
            _bits_close(raw, oracle.ssao(pfd, n, d), "ssao.comp")
            assert np.array_equal(g.ctx.download(lib.SSAO), oracle.ssao_blur(pfd, raw))
            _bits_close(g.ctx.download(lib.SSR), oracle.ssr(pfd, al, n, m, d), "ssr.comp")
            ref = oracle.composition(pfd, (2, 1, 1), al, n, m, d, np.zeros((H, W, 2), np.uint16), g.ctx.download(lib.SSR), ssao=g.ctx.download(lib.SSAO))
            diff = np.abs(g.ctx.download(g.out_img).astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1 and (diff == 0).mean() > 0.99
    finally:
        g.close()


@pytest.mark.gpu
def test_gpu_1080p_with_raytraced_shadows_and_svgf(oracle):
    """Full size, mixed modes (ray-traced shadows + SVGF, SSAO, SSR): a band of every image against the oracle, properties
    on the rest, and the kernel times."""
    scene = scenes.sponza_proc()
    W, H = 1920, 1080
    osc = oracle.Scene(scene)
    g = ScreenSpacePath(scene, W, H, shadow_mode=0, denoise=True)
    try:
        g.ctx.set_kernel_timing(["ssao", "ssao_blur", "ssr"])
        pfds = camera.dolly_frames(scene, W, H, 3)
        for pfd in pfds:
            gbuf = osc.gbuffer(pfd, W, H, with_albedo=True)
            g.frame(pfd, gbuf)
        n, m, d, al = gbuf
        band = (500, 540)
        raw = g.ctx.download(lib.SSAO_RAW)
        assert (raw[band[0]:band[1]] == oracle.ssao(pfd, n, d, rows=band)[band[0]:band[1]]).all(-1).mean() > 0.999
        assert np.array_equal(g.ctx.download(lib.SSAO)[band[0]:band[1]], oracle.ssao_blur(pfd, raw, rows=band)[band[0]:band[1]])
        refl = g.ctx.download(lib.SSR)
        assert (refl[band[0]:band[1]] == oracle.ssr(pfd, al, n, m, d, rows=band)[band[0]:band[1]]).all(-1).mean() > 0.999
        sky = d == 0
        assert np.isfinite(f16(raw)).all() and np.isfinite(f16(refl)).all()
        all_sky = sky[1:, 1:] & sky[:-1, 1:] & sky[1:, :-1] & sky[:-1, :-1]      # the four texels pixel (x, y) blends (x, y >= 1)
        assert (f16(refl)[1:, 1:, 3][all_sky] == 0).all() and (f16(raw)[1:, 1:, 0][all_sky] == 0).all()
        for k in ("ssao", "ssao_blur", "ssr"):
            ms, launches = g.ctx.kernel_time(k)
            assert launches == 3 and ms / launches < 50.0, (k, ms / launches)
            print(f"{k}: {ms / launches * 1e3:.1f} us per 1080p launch")
        assert g.ctx.download(g.out_img)[..., :3].mean() > 5
    finally:
        g.close()
