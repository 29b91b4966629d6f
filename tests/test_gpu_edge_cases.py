"""Edge cases of the hot path on the GPU, each checked against the oracle: textured closest-hit shading and fp16
object-id aliasing (> 2048 primitives), empty and single-triangle scenes, image sizes that are not multiples of
the tile sizes, axis-parallel rays (zero direction components), the BASELINE ao_spp extensions, sky-only frames."""
import numpy as np
import pytest

from tests.helpers import GpuHybrid, assert_reflections_identical, f16, oracle_frames
from vulkanhybridrenderer_amd import abi, camera, lib, scenes

pytestmark = pytest.mark.gpu


def _check(oracle, scene, W, H, n_frames, tp, denoise=True):
    frames, osc, _ = oracle_frames(oracle, scene, W, H, n_frames, tp, denoise=denoise)
    g = GpuHybrid(scene, W, H, denoise=denoise, trace_params=tp)
    try:
        for i, fr in enumerate(frames):
            g.frame(fr["pfd"], fr["gbuf"])
            assert np.array_equal(g.ctx.download(lib.RAYTRACED), fr["shadow_ao"]), f"frame {i}: visibility"
            if tp["reflections"]:
                assert_reflections_identical(g.ctx.download(lib.REFLECTIONS), fr["reflections"], f"frame {i}: reflections")
            if denoise:
                den, ref = f16(g.ctx.download(lib.DENOISED)), f16(fr["denoised"])
                assert float(np.sqrt(np.mean((den - ref) ** 2))) <= 1e-4, f"frame {i}: denoised RMSE"
        return frames
    finally:
        g.close()


def test_textured_scene_with_id_aliasing(oracle):
    """bistro_proc in miniature: 2200 primitives (ids above 2048 alias in fp16, gbuf.frag:43), sRGB base-colour
    textures with REPEAT/LINEAR samplers sampled by the closest-hit shader (reflection_hit.rchit:27-39)."""
    scene = scenes.bistro_proc(detail=0.02, n_primitives=2200, n_textures=8, texture_size=64)
    frames = _check(oracle, scene, 160, 90, 3, abi.default_trace_params())
    ids = f16(frames[1]["gbuf"][0])[..., 3]
    assert ids.max() > 2048 and (ids[ids > 2048] % 2 == 0).all()          # odd ids above 2048 are not representable
    assert (f16(frames[1]["reflections"])[..., 3] > 0).mean() > 0.5


def test_empty_scene(oracle):
    scene = scenes.tiny_scene()
    scene.primitives = scene.primitives[:0]
    W, H = 40, 24
    tp = abi.default_trace_params()
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    # a G-buffer that claims coverage although nothing can be hit: every ray escapes
    n = np.zeros((H, W, 4), np.float16); n[..., 1] = 1.0
    gbuf = (n.view(np.uint16), np.zeros((H, W, 4), np.uint16), np.full((H, W), 0.5, np.float32))
    osc = oracle.Scene(scene)
    sa, refl, _, _ = osc.raygen(pfd, tp, gbuf[0], gbuf[2])
    assert (f16(sa) == 1.0).all() and not refl.any()
    for variant in (0, 1):
        g = GpuHybrid(scene, W, H, denoise=False, trace_params=tp)
        try:
            g.ctx.set_option("raygen_variant", variant)
            g.frame(pfd, gbuf)
            assert np.array_equal(g.ctx.download(lib.RAYTRACED), sa)
            assert not g.ctx.download(lib.REFLECTIONS).any()
            assert g.ctx.bvh_statistics()["nodes"] == 0
        finally:
            g.close()


def test_single_triangle_scene(oracle):
    scene = scenes.tiny_scene()
    p = scene.primitives[:1].copy()
    p["index_count"] = 3
    scene.primitives = p
    _check(oracle, scene, 64, 40, 2, abi.default_trace_params(), denoise=False)


@pytest.mark.parametrize("W,H", [(67, 45), (130, 9), (8, 8)])
def test_odd_image_sizes(oracle, W, H):
    _check(oracle, scenes.tiny_scene(), W, H, 3, abi.default_trace_params())


def test_axis_parallel_light_and_rays(oracle):
    """Light straight down: shadow rays have exactly-zero direction components on the frame where every pixel
    draws the same cone sample; the slab test must not cull on 0 * inf."""
    scene = scenes.tiny_scene()
    scene.light = camera.directional_light((0.0, -1.0, 0.0))
    frames = _check(oracle, scene, 96, 64, 3, abi.default_trace_params())
    assert 0.01 < (f16(frames[1]["shadow_ao"])[..., 0] == 0).mean() < 0.95


@pytest.mark.parametrize("ao_spp", [0, 1, 4, 16, 20, 31, 32, 64])       # (up to 31: one blocked bit per ray in the pixel's word; beyond: a count)
def test_ao_sample_counts(oracle, ao_spp):
    """BASELINE.json configs 3 and 5 use 4 and 16 AO samples."""
    tp = abi.default_trace_params(ao_spp=ao_spp, reflections=False)
    frames, _, _ = oracle_frames(oracle, scenes.tiny_scene(), 64, 40, 2, tp, denoise=False)
    g = GpuHybrid(scenes.tiny_scene(), 64, 40, denoise=False, trace_params=tp, reflections=False)
    try:
        for fr in frames:
            g.frame(fr["pfd"], fr["gbuf"])
            assert np.array_equal(g.ctx.download(lib.RAYTRACED), fr["shadow_ao"])
    finally:
        g.close()


def test_shadows_disabled_extension(oracle):
    tp = abi.default_trace_params(shadow=False, reflections=False)
    frames = _check(oracle, scenes.tiny_scene(), 64, 40, 2, tp, denoise=False)
    assert (f16(frames[1]["shadow_ao"])[..., 0] == 1.0).all()


def _option_cases():
    """Every option of the library's table (vhr_option_info) with its smallest, its largest and its default value (+ a value in between for
    the wide ranges).  frames_in_flight is read at graph build (its own tests: test_gpu_frames_in_flight.py); kernel_timing_stride only
    thins event pairs.  No device needed to list them."""
    table = lib.option_table()
    cases = []
    for name, (default, lo, hi) in sorted(table.items()):
        if name == "frames_in_flight":
            continue
        values = sorted({lo, hi, default} | ({(lo + hi) // 2} if hi - lo > 2 and hi - lo < 100 else set()) | ({6} if name == "kernel_timing_stride" else set()))
        if name == "kernel_timing_stride":
            values = [1, 6]
        cases.append((name, tuple(values)))
    return cases


def test_the_option_table_is_what_the_neutrality_test_covers():
    table = lib.option_table()
    assert set(n for n, _ in _option_cases()) == set(table) - {"frames_in_flight"}
    for name, (default, lo, hi) in table.items():
        assert lo <= default <= hi, name


def test_options_outside_their_range_are_refused():
    c = lib.Context(32, 32)
    try:
        for name, (default, lo, hi) in lib.option_table().items():
            assert c.get_option(name) == default
            for bad in (lo - 1, hi + 1):
                with pytest.raises(lib.VhrError):
                    c.set_option(name, bad)
            assert c.get_option(name) == default
        with pytest.raises(lib.VhrError):
            c.set_option("no_such_option", 1)
    finally:
        c.close()


@pytest.mark.parametrize("option,values", _option_cases())
def test_every_tuning_option_is_result_neutral(oracle, option, values):
    scene = scenes.tiny_scene()
    W, H = 72, 56
    tp = abi.default_trace_params(reflections=False)
    frames, _, _ = oracle_frames(oracle, scene, W, H, 3, tp)
    outs = []
    for v in values:
        g = GpuHybrid(scene, W, H, trace_params=tp, reflections=False)
        try:
            g.ctx.set_option(option, v)
            for fr in frames:
                g.frame(fr["pfd"], fr["gbuf"])
                assert np.array_equal(g.ctx.download(lib.RAYTRACED), fr["shadow_ao"]), (option, v)
            outs.append(g.ctx.download(lib.DENOISED))
            if option in ("lds_stack_levels", "compact_nodes", "raygen_waves_per_block", "raygen_tile_rows"):
                g.ctx.set_ray_statistics(True)
                g.frame(frames[-1]["pfd"], frames[-1]["gbuf"])
                assert g.ctx.ray_statistics()["stack_overflows"] == 0
        finally:
            g.close()
    if option != "atrous_variant":        # the a-trous variants may differ in the last fp16 step (dot2 vs three FMAs)
        assert all(np.array_equal(o, outs[0]) for o in outs)
    else:
        assert all(float(np.sqrt(np.mean((f16(o) - f16(frames[-1]["denoised"])) ** 2))) <= 1e-4 for o in outs)


_TRACE_OPTIONS = ("raygen_variant", "reflection_variant", "refill_threshold", "lds_stack_levels", "reflection_lds_stack_levels", "raygen_waves_per_block", "compact_nodes", "raygen_early_exit",
                  "reflection_early_exit", "raygen_tile_rows", "raygen_cost_order", "raygen_steal", "reflection_async", "fuse_temporal", "svgf_async_unread")


@pytest.mark.parametrize("shadow,ao_spp,bounces", [(True, 0, 0), (True, 16, 0), (True, 33, 1), (False, 2, 1), (True, 2, 2), (False, 64, 0)])
def test_tuning_options_crossed_with_trace_parameters(oracle, shadow, ao_spp, bounces):
    """The neutrality test above sweeps every option at the default trace parameters; a bug of round 4 (raygen_steal's one blocked bit per
    ray with more than 31 AO samples) lived in the cross.  Here every option that touches the ray-tracing launches or their place in the
    frame takes its smallest, largest and a middle value under: no AO rays, 16 and 33 and 64 AO samples (the visibility word's two
    encodings), shadows off, one and two mirror bounces.  Per setting, two frames: the visibility image bit-exact against the oracle,
    Reflections and Denoised bit-identical to the same context's run with every option at its default (frame 0 carries NaN motion
    vectors, so it restarts the SVGF history: frame 1 is comparable between runs)."""
    scene = scenes.tiny_scene()
    W, H = 72, 56
    tp = abi.default_trace_params(shadow=shadow, ao_spp=ao_spp, reflections=bounces)
    frames, _, _ = oracle_frames(oracle, scene, W, H, 2, tp)
    table = lib.option_table()
    g = GpuHybrid(scene, W, H, shadow=shadow, ao=bool(ao_spp), reflections=bool(bounces), trace_params=tp)
    try:
        def run(tag):
            out = None
            for fr in frames:
                g.frame(fr["pfd"], fr["gbuf"])
                got = g.ctx.download(lib.RAYTRACED)
                assert np.array_equal(got, fr["shadow_ao"]), (tag, int((got != fr["shadow_ao"]).any(-1).sum()))
                out = (g.ctx.download(lib.DENOISED), g.ctx.download(lib.REFLECTIONS) if bounces else None)
            return out
        base = run("defaults")
        if bounces:
            _refl_equal(base[1], frames[-1]["reflections"])
        for name in _TRACE_OPTIONS:
            default, lo, hi = table[name]
            for v in sorted({lo, hi, (lo + hi) // 2} - {default}):
                g.ctx.set_option(name, v)
                got = run((name, v))
                g.ctx.set_option(name, default)
                assert np.array_equal(got[0], base[0]), (name, v, "Denoised differs from the default options' run")
                if bounces and name != "reflection_variant":
                    assert np.array_equal(got[1], base[1]), (name, v, "Reflections differ from the default options' run")
                elif bounces:
                    _refl_equal(got[1], frames[-1]["reflections"])
    finally:
        g.close()


def _refl_equal(got_bits, want_bits):
    """mirror-ray payloads against the oracle: bit-identical"""
    assert_reflections_identical(got_bits, want_bits)


def test_sky_only_frame(oracle):
    scene = scenes.tiny_scene()
    scene.camera = dict(scene.camera, pitch=1.45)        # look straight up: nothing but sky
    frames = _check(oracle, scene, 64, 40, 2, abi.default_trace_params())
    assert not frames[1]["gbuf"][2].any()
    assert (f16(frames[1]["shadow_ao"]) == 1.0).all()


def test_scene_through_the_gltf_host(oracle, tmp_path):
    """Row f1: a scene that went through the glTF container (export, then the scene_loader.cpp-equivalent import with
    PNG-decoded textures) renders with the same parity as the in-memory one: visibility bit-exact, mirror ray shaded
    from the loaded sRGB textures."""
    from vulkanhybridrenderer_amd import gltf
    src = scenes.bistro_proc(detail=0.004, n_primitives=60, n_textures=4, texture_size=32)
    path = str(tmp_path / "scene.glb")
    gltf.save(src, path)
    scene = gltf.load(path)
    assert np.array_equal(scene.vertices, src.vertices) and len(scene.textures) == 4
    _check(oracle, scene, 96, 64, 2, abi.default_trace_params())


def test_odd_sizes_and_ao_only_queues(oracle):
    """Image sizes that leave partial tiles, with and without shadow rays (the AO-only queue's cut is pruned to the rays' reach; with shadow
    rays in the queue the AO rays skip the cut's far entries), 1 and 4 AO samples, every tile height: visibility bit-identical to the oracle."""
    scene = scenes.sponza_proc(detail=0.25) if hasattr(scenes, "sponza_proc") else scenes.tiny_scene()
    W, H = 150, 93
    for shadow, ao in ((True, 1), (False, 4), (True, 0)):
        tp = abi.default_trace_params(shadow=shadow, ao_spp=ao, reflections=False)
        frames, _, _ = oracle_frames(oracle, scene, W, H, 2, tp, denoise=False)
        g = GpuHybrid(scene, W, H, shadow=shadow, ao=bool(ao), trace_params=tp, reflections=False, denoise=False)
        try:
            for key, val in (("raygen_tile_rows", 8), ("raygen_tile_rows", 5), ("raygen_waves_per_block", 4), ("lds_stack_levels", 2)):
                g.ctx.set_option(key, val)
                for fr in frames:
                    g.frame(fr["pfd"], fr["gbuf"])
                    got = g.ctx.download(lib.RAYTRACED)
                    assert np.array_equal(got, fr["shadow_ao"]), (shadow, ao, key, val, int((got != fr["shadow_ao"]).any(-1).sum()))
        finally:
            g.close()


def test_non_finite_geometry_is_rejected():
    """vhr_update_geometry refuses NaN / Inf positions and transforms (they would reach the builder's bin index and comparators)."""
    import copy
    scene = scenes.tiny_scene()
    c = lib.Context(32, 32)
    try:
        bad = copy.copy(scene)
        bad.vertices = scene.vertices.copy()
        bad.vertices["pos"][3, 1] = np.nan
        with pytest.raises(lib.VhrError, match="non-finite position"):
            c.upload_scene(bad)
        bad = copy.copy(scene)
        bad.primitives = scene.primitives.copy()
        bad.primitives["transform"][0, 5] = np.inf
        with pytest.raises(lib.VhrError, match="non-finite transform"):
            c.upload_scene(bad)
        c.upload_scene(scene)                                   # the context is still usable
        build_ms, upload_ms = c.build_times_ms()
        assert build_ms > 0 and upload_ms > 0
    finally:
        c.close()


def test_every_builder_makes_the_same_tree():
    """The host builder's thread pool ("bvh_builder" 0 with "bvh_build_threads" 1, 3 and the default) and the device's builder (the
    default): the same algorithm, so the same TREE -- vhr_get_bvh_tree_fingerprint hashes every inner node's child boxes (their bits) and
    every leaf's set of triangles, whatever the order of nodes, leaves and triangles in memory --, the same counts and depth, and the same
    image bit for bit, on a scene large enough to be split among threads and to run the device builder's level passes (sponza_proc, 257 k
    triangles).  The array hash (vhr_get_bvh_fingerprint) is equal among the host's thread counts and differs for the device's tree: its
    leaves lie in depth-first order, the host's by treelets."""
    scene = scenes.sponza_proc()
    W, H = 256, 144
    tp = abi.default_trace_params(reflections=False)
    pfd = camera.dolly_frames(scene, W, H, 2)[1]
    ref = None
    for threads in (1, 3, 0, None):
        c = lib.Context(W, H)
        try:
            if threads is not None:
                c.set_option("bvh_builder", 0)
                c.set_option("bvh_build_threads", threads)
            c.upload_scene(scene)
            assert c.bvh_builder_used() == (1 if threads is None else 0)
            c.set_trace_params(tp)
            path = lib.HybridRenderPath(c, 0, 0, 2, False, 5, lambda ctx: ctx.standin_gbuffer(0))
            path.build()
            c.update_per_frame_ubo(0, pfd)
            c.execute(0, 0)
            c.synchronize()
            got = (c.bvh_statistics(), c.download(lib.RAYTRACED), c.bvh_tree_fingerprint(), c.bvh_fingerprint())
            path.destroy()
        finally:
            c.close()
        if ref is None:
            ref = got
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and got[2] == ref[2] and got[2] != 0, threads
        assert (got[3] == ref[3]) == (threads is not None), threads


def test_host_and_device_builders_make_the_same_tree_of_the_large_scene():
    """bistro_proc at half its size (0.7 M triangles, 21 levels of the device builder's level passes): one tree fingerprint from both builders."""
    scene = scenes.bistro_proc(0.5)
    prints = []
    c = lib.Context(64, 64)
    try:
        for builder in (1, 0):
            c.set_option("bvh_builder", builder)
            c.upload_scene(scene)
            assert c.bvh_builder_used() == builder
            prints.append((c.bvh_tree_fingerprint(), c.bvh_statistics()["nodes"], c.bvh_statistics()["max_depth"]))
    finally:
        c.close()
    assert prints[0] == prints[1] and prints[0][0] != 0


def test_every_node_form_contains_its_box():
    """The walkers read three derived node forms (centre / half extent, the same in 48 bytes with 16-bit half extents, half-precision
    compact nodes); box tests only cull, so bit-identity with the oracle needs every derived box to CONTAIN the builder's (lo, hi) box in
    exact arithmetic.  The library checks that on the host after every build (vhr_get_bvh_form_checks); here: the two benchmark scenes,
    the tiny scene, and a scene far from the origin with very small and very large triangles (where fp32 centres lose bits and the
    half extents span 12 orders of magnitude)."""
    import dataclasses
    rng = np.random.default_rng(7)
    tiny = scenes.tiny_scene()
    v = tiny.vertices.copy()
    v["pos"] = v["pos"] * rng.choice(np.array([1e-4, 1.0, 3e3], np.float32), size=(len(v), 1)).astype(np.float32) + np.float32(12345.678)
    far = dataclasses.replace(tiny, name="tiny_far", vertices=v)
    for scene in (tiny, scenes.sponza_proc(0.5), scenes.bistro_proc(0.25, n_primitives=400, n_textures=4, texture_size=16), far):
        c = lib.Context(64, 64)
        try:
            # the device-built tree is checked where it is (k0_check_forms_kernel); "bvh_host_checks" 1 fetches it and repeats the checks with
            # the host's code: the same counts, and the same fingerprint whether it is taken right away or when first asked for;
            # then the host builder's tree
            seen = []
            for builder, host_checks in ((1, 0), (1, 1), (0, 0)):
                c.set_option("bvh_builder", builder)
                c.set_option("bvh_host_checks", host_checks)
                c.upload_scene(scene)
                assert c.bvh_builder_used() == builder
                boxes, ch_bad, n48_bad, n16_bad = c.bvh_form_checks()
                assert boxes == 2 * c.bvh_statistics()["nodes"] and boxes > 0
                assert (ch_bad, n48_bad, n16_bad) == (0, 0, 0), (scene.name, builder, host_checks, ch_bad, n48_bad, n16_bad)
                seen.append((boxes, c.bvh_fingerprint()))
            assert seen[0] == seen[1]
        finally:
            c.close()


def test_scene_far_from_the_origin_with_mixed_scales(oracle):
    """Coordinates around 12 345 with vertices scaled by 1/100 .. 100 at random (triangles from millimetres to half a kilometre, many of
    them slivers): where the fp32 centres of the 48-byte nodes lose bits and their 16-bit half extents span many orders of magnitude.
    Visibility must still equal the oracle's bit for bit, with its BVH and by brute force (the boxes only cull), frame after frame.
    (With a 1 000-fold mix culling stops being neutral for ANY box set -- the oracle's own BVH then differs from its brute force in a few
    texels: Moeller-Trumbore on a 5 km sliver accepts points decimetres outside the triangle's padded box.  DESIGN.md section 4.)"""
    import dataclasses
    rng = np.random.default_rng(9)
    tiny = scenes.tiny_scene()
    v = tiny.vertices.copy()
    centre = np.float32(12345.678)
    scale = rng.choice(np.array([1e-2, 1.0, 1.0, 1.0, 1e2], np.float32), size=(len(v), 1)).astype(np.float32)
    v["pos"] = v["pos"] * scale + centre
    cam = dict(tiny.camera)
    cam["position"] = tuple(float(c) + float(centre) for c in cam["position"])
    far = dataclasses.replace(tiny, name="tiny_far", vertices=v, camera=cam)
    tp = abi.default_trace_params(reflections=False)
    frames = _check(oracle, far, 96, 64, 3, tp, denoise=False)
    osc = oracle.Scene(far)
    brute = osc.raygen(frames[1]["pfd"], tp, frames[1]["gbuf"][0], frames[1]["gbuf"][2], use_bvh=False, want_reflections=False)[0]
    assert np.array_equal(brute, frames[1]["shadow_ao"])
    shadow = f16(frames[1]["shadow_ao"])[..., 0]
    assert 0.0 < (shadow == 0).mean() < 1.0          # something is lit, something is in shadow


def test_scene_wider_than_the_half_range_falls_back_to_the_48_byte_nodes():
    """The 32-byte nodes (compact_nodes, the default) hold centres and half extents as IEEE halves around the scene centre: a scene wider
    than +-65 504 has no such form.  The build then marks it unusable (no violation is counted), the queue kernel walks the 48-byte nodes,
    and the frame is the one compact_nodes 0 gives, bit for bit."""
    import dataclasses
    tiny = scenes.tiny_scene()
    v = tiny.vertices.copy()
    far_corner = np.zeros(len(v), bool)
    far_corner[:3] = True                                  # three vertices of the floor go 200 km out: triangles 200 km long
    v["pos"][far_corner] *= np.float32(5e4)
    wide = dataclasses.replace(tiny, name="tiny_wide", vertices=v)
    W, H = 96, 64
    pfds = camera.dolly_frames(wide, W, H, 2)
    images = []
    for compact in (1, 0):
        g = GpuHybrid(wide, W, H, reflections=False, trace_params=abi.default_trace_params(reflections=False), gbuffer="standin", denoise=False)
        try:
            g.ctx.set_option("compact_nodes", compact)
            assert g.ctx.bvh_form_checks()[1:] == (0, 0, 0)
            for pfd in pfds:
                g.frame(pfd)
            images.append(g.ctx.download(lib.RAYTRACED))
        finally:
            g.close()
    assert np.array_equal(images[0], images[1])
    assert 0.0 < (f16(images[0])[..., 0] == 0).mean() < 1.0
