"""ctypes binding of oracle/libvhr_oracle.so (TEST INFRASTRUCTURE ONLY -- see oracle/vhr_oracle.h).

PARITY UNPINNED for the shaders' arithmetic: the oracle is pinned only by hand-derived known-answer values
(tests/golden/) and an independent numpy restatement (tests/numpy_restatement.py), because the reference ships no
tests and cannot be built in this image.  Its struct layouts are pinned against the reference's own glsl_common.h
(tests/test_reference_pins.py, oracle/ref_probes/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libvhr_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("vhr_oracle.c", "vhr_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvhr_oracle.so")
        if os.environ.get("VHR_ORACLE_LIB"):          # tests/test_sanitizers.py: the AddressSanitizer build (make -C oracle asan)
            so = os.path.abspath(os.environ["VHR_ORACLE_LIB"])
        elif not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u32, f32, vp, i32 = C.c_uint32, C.c_float, C.c_void_p, C.c_int
        L.orc_seed_thread.restype = u32
        L.orc_seed_thread.argtypes = [u32]
        L.orc_random.restype = u32
        L.orc_random.argtypes = [C.POINTER(u32)]
        L.orc_random01.restype = f32
        L.orc_random01.argtypes = [C.POINTER(u32)]
        L.orc_random_range.restype = u32
        L.orc_random_range.argtypes = [C.POINTER(u32), u32, u32]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [f32]
        L.orc_f16_to_f32.restype = f32
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_sincos.argtypes = [f32, C.POINTER(f32), C.POINTER(f32)]
        L.orc_uniform_sample_cone.argtypes = [f32, f32, f32, vp]
        L.orc_cosine_hemisphere.argtypes = [f32, f32, vp]
        L.orc_onb.argtypes = [vp, vp]
        L.orc_ray_triangle.restype = i32
        L.orc_ray_triangle.argtypes = [vp, vp, vp, vp, vp, f32, f32, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
        L.orc_infinite_reverse_depth_projection.argtypes = [f32, f32, f32, vp]
        L.orc_mat4_inverse.argtypes = [vp, vp]
        L.orc_mat4_mul.argtypes = [vp, vp, vp]
        L.orc_default_trace_params.argtypes = [vp]
        L.orc_struct_sizes.restype = i32
        L.orc_struct_sizes.argtypes = [vp]
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, u32, vp, u32, vp, u32]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_add_texture.restype = i32
        L.orc_scene_add_texture.argtypes = [vp, u32, u32, vp, i32, i32, i32, i32, i32]
        L.orc_scene_triangle_count.restype = u32
        L.orc_scene_triangle_count.argtypes = [vp]
        L.orc_scene_occluded.restype = i32
        L.orc_scene_occluded.argtypes = [vp, vp, vp, f32, f32, i32]
        L.orc_scene_closest.restype = i32
        L.orc_scene_closest.argtypes = [vp, vp, vp, f32, f32, i32, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32),
                                        C.POINTER(u32), C.POINTER(u32)]
        L.orc_gbuffer.argtypes = [vp, vp, u32, u32, vp, vp, vp]
        L.orc_gbuffer_albedo.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp]
        L.orc_composition.argtypes = [vp, u32, u32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, u32, vp]
        L.orc_shadow_map.argtypes = [vp, vp, u32, u32, u32, vp, i32]
        L.orc_raygen.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, i32]
        L.orc_raytraced.argtypes = [vp, vp, u32, u32, u32, u32, i32, vp, vp, i32]
        L.orc_raytraced_composition.argtypes = [u32, u32, vp, vp]
        L.orc_sample_linear_repeat.argtypes = [i32, vp, u32, u32, f32, f32, vp]
        L.orc_ssao.argtypes = [vp, u32, u32, u32, u32, vp, vp, f32, vp]
        L.orc_ssao_blur.argtypes = [vp, u32, u32, u32, u32, vp, vp]
        L.orc_ssr.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, vp, f32, f32, f32, C.c_int32, vp]
        L.orc_svgf_temporal.argtypes = [vp, u32, u32] + [vp] * 8
        L.orc_svgf_atrous.argtypes = [vp, u32, u32, vp, vp, vp, C.c_int32]
        L.orc_svgf_create.restype = vp
        L.orc_svgf_create.argtypes = [u32, u32]
        L.orc_svgf_destroy.argtypes = [vp]
        L.orc_svgf_frame.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_svgf_image.restype = vp
        L.orc_svgf_image.argtypes = [vp, i32]
        L.orc_max_threads.restype = i32
        L.orc_audit_begin.argtypes = [i32, u32]
        L.orc_audit_end.restype = u32
        L.orc_audit_end.argtypes = [vp, vp]
        L.orc_ray_triangle_exact.restype = i32
        L.orc_ray_triangle_exact.argtypes = [vp, vp, vp, vp, vp, f32, f32, vp]
        L.orc_ray_triangle_rules.restype = i32
        L.orc_ray_triangle_rules.argtypes = [vp, vp, vp, vp, vp, f32, f32, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype=None):
    a = np.ascontiguousarray(a) if dtype is None else np.ascontiguousarray(a, dtype=dtype)
    return a


def f32_to_f16_bits(x):
    L = lib()
    return np.array([L.orc_f32_to_f16(float(np.float32(v))) for v in np.ravel(x)], np.uint16).reshape(np.shape(x))


def rng_sequence(seed_input, n):
    """seed_thread(seed_input) followed by n random01 draws -> (seed, states, floats)."""
    L = lib()
    st = C.c_uint32(L.orc_seed_thread(seed_input))
    seed = st.value
    states, vals = [], []
    for _ in range(n):
        vals.append(np.float32(L.orc_random01(C.byref(st))))
        states.append(st.value)
    return seed, states, vals


def sincos(phi):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(float(np.float32(phi)), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


class Scene:
    def raytraced(self, pfd, W, H, use_anyhit_shader=False, rows=None, use_bvh=True):
        """The raytraced render path's "Raytracing Pass": (B8G8R8A8_UNORM image (H, W, 4), rays traced)."""
        r0, r1 = rows if rows is not None else (0, H)
        out = np.zeros((H, W, 4), np.uint8)
        rays = C.c_uint64()
        pfd = _c(pfd)
        lib().orc_raytraced(self.handle, _p(pfd), W, H, r0, r1, int(bool(use_anyhit_shader)), _p(out), C.byref(rays), int(use_bvh))
        return out, rays.value

    def __init__(self, scene):
        L = lib()
        self._v = _c(scene.vertices)
        self._i = _c(scene.indices, np.uint32)
        self._p = _c(scene.primitives)
        self.handle = L.orc_scene_create(_p(self._v), len(self._v), _p(self._i), len(self._i), _p(self._p), len(self._p))
        for t in scene.textures:
            img = _c(t["rgba8"], np.uint8)
            L.orc_scene_add_texture(self.handle, img.shape[1], img.shape[0], _p(img), t["format"], t["mag"], t["min"],
                                    t["address_u"], t["address_v"])

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_scene_destroy(self.handle)
            self.handle = None

    @property
    def triangle_count(self):
        return lib().orc_scene_triangle_count(self.handle)

    def occluded(self, o, d, tmin, tmax, use_bvh=True):
        o, d = _c(o, np.float32), _c(d, np.float32)
        return bool(lib().orc_scene_occluded(self.handle, _p(o), _p(d), tmin, tmax, int(use_bvh)))

    def closest(self, o, d, tmin, tmax, use_bvh=True):
        o, d = _c(o, np.float32), _c(d, np.float32)
        t, u, v = C.c_float(), C.c_float(), C.c_float()
        pr, tr = C.c_uint32(), C.c_uint32()
        hit = lib().orc_scene_closest(self.handle, _p(o), _p(d), tmin, tmax, int(use_bvh), C.byref(t), C.byref(u),
                                      C.byref(v), C.byref(pr), C.byref(tr))
        return (np.float32(t.value), np.float32(u.value), np.float32(v.value), pr.value, tr.value) if hit else None

    def shadow_map(self, pfd, size=4096, rows=None, use_bvh=True):
        """Stand-in for the rasterised Shadow Map Pass (decision xiv): (size, size) float32."""
        out = np.zeros((size, size), np.float32)
        pfd = _c(pfd)
        r0, r1 = (0, size) if rows is None else rows
        lib().orc_shadow_map(self.handle, _p(pfd), size, r0, r1, _p(out), int(use_bvh))
        return out

    def gbuffer(self, pfd, W, H, with_albedo=False):
        normals = np.zeros((H, W, 4), np.uint16)
        motion = np.zeros((H, W, 4), np.uint16)
        depth = np.zeros((H, W), np.float32)
        albedo = np.zeros((H, W, 4), np.uint8) if with_albedo else None
        pfd = _c(pfd)
        lib().orc_gbuffer_albedo(self.handle, _p(pfd), W, H, _p(normals), _p(motion), _p(depth), _p(albedo))
        return (normals, motion, depth, albedo) if with_albedo else (normals, motion, depth)

    def raygen(self, pfd, tp, normals, depth, rows=None, use_bvh=True, want_reflections=True):
        H, W = depth.shape
        r0, r1 = rows if rows is not None else (0, H)
        shadow_ao = np.zeros((H, W, 2), np.uint16)
        refl = np.zeros((H, W, 4), np.uint16) if want_reflections else None
        mask = np.zeros((H, W), np.uint8)
        rays = C.c_uint64()
        pfd, tp = _c(pfd), _c(tp)
        normals, depth = _c(normals, np.uint16), _c(depth, np.float32)
        lib().orc_raygen(self.handle, _p(pfd), _p(tp), W, H, r0, r1, _p(normals), _p(depth), _p(shadow_ao), _p(refl),
                         _p(mask), C.byref(rays), int(use_bvh))
        return shadow_ao, refl, mask, rays.value


def raytraced_composition(raytraced_bgra8):
    """raytraced_render_path/composition.frag: B8G8R8A8_UNORM image -> presented B8G8R8A8_SRGB texels (H, W, 4)."""
    H, W = raytraced_bgra8.shape[:2]
    src = _c(raytraced_bgra8, np.uint8)
    out = np.zeros((H, W, 4), np.uint8)
    lib().orc_raytraced_composition(W, H, _p(src), _p(out))
    return out


def composition(pfd, modes, albedo, normals, motion, depth, shadow_ao, reflections, ssao=None, shadow_map=None):
    """composition.frag with (shadow_mode, ao_mode, reflection_mode); returns B8G8R8A8_SRGB texels (H, W, 4)."""
    H, W = depth.shape
    out = np.zeros((H, W, 4), np.uint8)
    pfd = _c(pfd)
    albedo, normals, motion = _c(albedo, np.uint8), _c(normals, np.uint16), _c(motion, np.uint16)
    depth, shadow_ao = _c(depth, np.float32), _c(shadow_ao, np.uint16)
    reflections = _c(reflections, np.uint16) if reflections is not None else np.zeros((H, W, 4), np.uint16)
    lib().orc_composition(_p(pfd), W, H, modes[0], modes[1], modes[2], _p(albedo), _p(normals), _p(motion), _p(depth), _p(shadow_ao),
                          shadow_ao.shape[-1], _p(reflections), _p(_c(ssao, np.uint16)) if ssao is not None else None,
                          _p(_c(shadow_map, np.float32)) if shadow_map is not None else None, 0 if shadow_map is None else shadow_map.shape[0], _p(out))
    return out


def sample_linear_repeat(img, u, v):
    """texture() through the default sampler (decision x) on an RGBA16F (H, W, 4) uint16, D32F (H, W) float32 or
    B8G8R8A8 (H, W, 4) uint8 image -> 4 floats."""
    img = _c(img)
    kind = {np.dtype(np.uint16): 0, np.dtype(np.float32): 1, np.dtype(np.uint8): 2}[img.dtype]
    H, W = img.shape[:2]
    out = np.zeros(4, np.float32)
    lib().orc_sample_linear_repeat(kind, _p(img), W, H, float(np.float32(u)), float(np.float32(v)), _p(out))
    return out


def _rows(rows, H):
    return (0, H) if rows is None else (int(rows[0]), int(rows[1]))


def ssao(pfd, normals, depth, radius=0.75, rows=None):
    """ssao.comp -> "Screen Space Ambient Occlusion Raw" (H, W, 4) uint16 (RGBA16F bits)."""
    H, W = depth.shape
    out = np.zeros((H, W, 4), np.uint16)
    pfd, normals, depth = _c(pfd), _c(normals, np.uint16), _c(depth, np.float32)
    r0, r1 = _rows(rows, H)
    lib().orc_ssao(_p(pfd), W, H, r0, r1, _p(normals), _p(depth), float(radius), _p(out))
    return out


def ssao_blur(pfd, ssao_raw, rows=None):
    """ssao_blur.comp -> "Screen Space Ambient Occlusion"."""
    H, W = ssao_raw.shape[:2]
    out = np.zeros((H, W, 4), np.uint16)
    pfd, ssao_raw = _c(pfd), _c(ssao_raw, np.uint16)
    r0, r1 = _rows(rows, H)
    lib().orc_ssao_blur(_p(pfd), W, H, r0, r1, _p(ssao_raw), _p(out))
    return out


def ssr(pfd, albedo, normals, motion, depth, ray_distance=25.0, step_size=0.1, thickness=0.5, bsearch_steps=10, rows=None):
    """ssr.comp -> "Screen Space Reflections" (defaults: hybrid_render_path.cpp:203-208)."""
    H, W = depth.shape
    out = np.zeros((H, W, 4), np.uint16)
    pfd, albedo, normals, motion, depth = _c(pfd), _c(albedo, np.uint8), _c(normals, np.uint16), _c(motion, np.uint16), _c(depth, np.float32)
    r0, r1 = _rows(rows, H)
    lib().orc_ssr(_p(pfd), W, H, r0, r1, _p(albedo), _p(normals), _p(motion), _p(depth), float(ray_distance), float(step_size),
                  float(thickness), int(bsearch_steps), _p(out))
    return out


def svgf_temporal(pfd, normals, motion, raytraced, prev_normals, history, moments_in):
    H, W = normals.shape[:2]
    integrated = np.zeros((H, W, 4), np.uint16)
    moments = np.zeros((H, W, 2), np.uint16)
    a = [_c(x, np.uint16) for x in (normals, motion, raytraced, prev_normals, history, moments_in)]
    pfd = _c(pfd)
    lib().orc_svgf_temporal(_p(pfd), W, H, *[_p(x) for x in a], _p(integrated), _p(moments))
    return integrated, moments


def svgf_atrous(pfd, normals, integrated_in, step):
    H, W = normals.shape[:2]
    out = np.zeros((H, W, 4), np.uint16)
    normals, integrated_in, pfd = _c(normals, np.uint16), _c(integrated_in, np.uint16), _c(pfd)
    lib().orc_svgf_atrous(_p(pfd), W, H, _p(normals), _p(integrated_in), _p(out), int(step))
    return out


class SVGF:
    """Host schedule of hybrid_render_path.cpp:245-331 with its five persistent images."""

    def __init__(self, W, H):
        self.W, self.H = W, H
        self.handle = lib().orc_svgf_create(W, H)

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_svgf_destroy(self.handle)
            self.handle = None

    def frame(self, pfd, normals, motion, raytraced):
        out = np.zeros((self.H, self.W, 4), np.uint16)
        a = [_c(x, np.uint16) for x in (normals, motion, raytraced)]
        pfd = _c(pfd)
        lib().orc_svgf_frame(self.handle, _p(pfd), *[_p(x) for x in a], _p(out))
        return out

    def image(self, which):
        ch = 2 if which == 4 else 4
        ptr = lib().orc_svgf_image(self.handle, which)
        buf = (C.c_uint16 * (self.W * self.H * ch)).from_address(ptr)
        return np.frombuffer(buf, np.uint16).reshape(self.H, self.W, ch).copy()


def max_threads():
    return lib().orc_max_threads()


# ---- the audit of decision (vi) (round 6; oracle/vhr_exact.h) ----
AUDIT_RULES = ("Moeller-Trumbore alone", "round 5: reject what contradicts itself", "(dropped) point in the triangle's box", "IN FORCE: binary64 for what contradicts itself")
audit_counts_dtype = np.dtype([
    ("rays", "<u8", (2,)), ("rays_undecided", "<u8"), ("rays_not_finite", "<u8"), ("pairs", "<u8"), ("undecided", "<u8"),
    ("exact_hits", "<u8"), ("mt_hits", "<u8"), ("mt_miss_exact_hit", "<u8"), ("cls", "<u8", (4, 4)),
    ("any_leak", "<u8", (4,)), ("any_spurious", "<u8", (4,)), ("closest_hit_miss", "<u8", (4,)),
    ("closest_differs", "<u8", (4,)), ("closest_differs_far", "<u8", (4,)), ("records_dropped", "<u8"), ("escalated", "<u8")])
audit_record_dtype = np.dtype([
    ("o", "<f4", (3,)), ("d", "<f4", (3,)), ("tmin", "<f4"), ("tmax", "<f4"), ("v0", "<f4", (3,)), ("e1", "<f4", (3,)), ("e2", "<f4", (3,)),
    ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("det", "<f4"), ("xdet", "<f8"), ("xu", "<f8"), ("xv", "<f8"), ("xt", "<f8"),
    ("flat", "<u4"), ("mt", "u1"), ("pass_mask", "u1"), ("any_hit", "u1"), ("exact", "i1"), ("cls", "S1"), ("pad_", "u1", (3,))], align=True)


class Audit:
    """with Audit(brute_force=False) as a: <oracle calls that trace rays>; then a.counts (audit_counts_dtype scalar), a.records."""

    def __init__(self, brute_force=False, max_records=200000):
        self.brute, self.cap = brute_force, max_records
        self.counts, self.records = None, None

    def __enter__(self):
        lib().orc_audit_begin(int(self.brute), self.cap)
        return self

    def __exit__(self, *exc):
        counts = np.zeros((), audit_counts_dtype)
        records = np.zeros(self.cap, audit_record_dtype)
        n = lib().orc_audit_end(_p(counts), _p(records))
        self.counts, self.records = counts, records[:n].copy()
        return False


def ray_triangle_exact(o, d, v0, e1, e2, tmin, tmax):
    """(decision, (det, u, v, t) in binary64): 1 hit, 0 miss, -1 undecided by the binary64 filter."""
    a = [_c(x, np.float32) for x in (o, d, v0, e1, e2)]
    out = np.zeros(4, np.float64)
    r = lib().orc_ray_triangle_exact(*[_p(x) for x in a], float(np.float32(tmin)), float(np.float32(tmax)), _p(out))
    return r, out


def ray_triangle_rules(o, d, v0, e1, e2, tmin, tmax):
    """(mask, (t, u, v, det) fp32): bit 0 Moeller-Trumbore's comparisons pass, bit 1 round 5's rule accepts, bit 2 the rule in force accepts."""
    a = [_c(x, np.float32) for x in (o, d, v0, e1, e2)]
    out = np.zeros(4, np.float32)
    r = lib().orc_ray_triangle_rules(*[_p(x) for x in a], float(np.float32(tmin)), float(np.float32(tmax)), _p(out))
    return r, out
