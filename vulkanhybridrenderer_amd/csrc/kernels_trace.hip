// K1 + K2 + stand-in G-buffer producer: software BVH2 ray tracing for gfx950.
//
// Replaces, for the hybrid render path's "Raytrace Pass" (hybrid_render_path.cpp:101-136):
//   data/shaders/hybrid_render_path/raygen.rgen:14-66        -> raygen_kernel
//   data/shaders/hybrid_render_path/miss.rmiss:6-8            -> payload 1.0 when the any-hit walk finds nothing
//   data/shaders/hybrid_render_path/reflection_miss.rmiss:6-8 -> payload 0 when the closest-hit walk finds nothing
//   data/shaders/hybrid_render_path/reflection_hit.rchit:10-72 -> shade_reflection_hit
//   vkCmdTraceRaysKHR (raytracing_execution_context.cpp:4-13) -> launch_raygen
// and the driver's acceleration-structure traversal (traceRayEXT) by traverse<>() below.
//
// Compiled with -ffp-contract=off: ray setup and Moeller-Trumbore follow the exact-arithmetic contract of
// device_math.hpp so visibility results are bit-reproducible.
#include <cstring>

#include "device_math.hpp"
#include "vhr_internal.hpp"
#include "svgf_temporal.hpp"

namespace vhr {

__constant__ float c_srgb_lut[256];

#ifndef VHR_K1_POSTPONE
#define VHR_K1_POSTPONE 0            // 1: a scratch build of the any-hit queue kernel WITH the postponed-leaf step (measured slower: profiles/r6_k1_postponed_leaf.txt)
#endif
#ifndef VHR_REDO_INLINE
#define VHR_REDO_INLINE __attribute__((noinline))
#endif
#ifndef VHR_K1_WAVES_MIN
#define VHR_K1_WAVES_MIN 8          // waves per SIMD the any-hit queue kernel is allocated for (7: <= 72 registers, 8: <= 64; redo_pixel_visibility() inherits it)
#endif
constexpr uint32_t kRedoPixel = 0x80000000u;   // raygen_queue_kernel's visibility word: the pixel is computed again by redo_pixel_visibility (decision (vi))
constexpr int kTraceBlock = 256;          // 4 waves; each wave owns an 8x8 pixel tile of a 16x16 block tile

struct Hit {
    float t, u, v;
    uint32_t tri_index;    // index into DeviceScene::tris
    uint32_t flat;
};

// Decision (vi), second half.  A candidate of fp32 Moeller-Trumbore whose solution is CONSISTENT -- the ray's point
// o + t d and the triangle's point v0 + u e1 + v e2 agree per axis to within 5e-4 + 5e-6 |coordinate|, half the padding of any box around the
// triangle -- is accepted as it is: whatever it is, it lies inside every box that leads to the triangle, in any frame.  For a ray within rounding
// of the triangle's plane the determinant is rounding noise and (t, u, v) contradict themselves: round 5 rejected such a candidate, which removed
// the hits that are not there (a point centimetres beside the triangle) and, at grazing incidence on large triangles, true ones with them
// (profiles/r6_decision_vi.txt: 21 059 of 387 896 exact hits on the raytraced path's terminator rays).  Since round 6 it is DECIDED AGAIN IN
// BINARY64 (mt_binary64 below): the audit against exact arithmetic counts no lost and no invented hit among them.
// Individually rounded operations in the oracle's order; a NaN is inconsistent.
__device__ __forceinline__ bool solution_consistent(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float t, float u, float v) {
    const float px = o.x + d.x * t, py = o.y + d.y * t, pz = o.z + d.z * t;
    const float qx = (v0.x + e1.x * u) + e2.x * v, qy = (v0.y + e1.y * u) + e2.y * v, qz = (v0.z + e1.z * u) + e2.z * v;
    return fabsf(px - qx) <= 5e-4f + 5e-6f * fabsf(qx) && fabsf(py - qy) <= 5e-4f + 5e-6f * fabsf(qy) && fabsf(pz - qz) <= 5e-4f + 5e-6f * fabsf(qz);
}

// Moeller-Trumbore in binary64 (the oracle's mt_binary64, operation for operation): the fp32 operands and every product of two of them are exact,
// every other operation rounds once in the order written (this unit is built with -ffp-contract=off), the quotients are IEEE divisions; the comparisons
// are ray_triangle()'s and (t, u, v) come back rounded to fp32.  Written for few live registers -- operands are widened where they are used, pvec and
// tvec are the only vectors kept.  It sits inside the leaf tests of the per-pixel walkers (traverse<>) and of the raytraced path's queue kernel; the two
// queue kernels of the hybrid path keep it OUT of their loops: a self-contradicting candidate marks the pixel, and the tile's epilogue computes the pixel
// again through traverse<> behind one call (redo_pixel_visibility / redo_pixel_reflection).  Inside the any-hit queue kernel's leaf test it cost 13
// registers = a wave per SIMD = 1.5 % of the frame, inside the two-bounce mirror kernel's 45 spilled registers = 13 % of the launch; the forms measured on
// the way (a call from the leaf test, a second launch, a list per ray decided at the ray's commit) are in profiles/r6_decision_vi_cost.txt.
__device__ __forceinline__ bool mt_binary64(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float tmin, float tmax, float &t, float &u, float &v) {
    const double px = double(d.y) * double(e2.z) - double(d.z) * double(e2.y);
    const double py = double(d.z) * double(e2.x) - double(d.x) * double(e2.z);
    const double pz = double(d.x) * double(e2.y) - double(d.y) * double(e2.x);
    const double det = (double(e1.x) * px + double(e1.y) * py) + double(e1.z) * pz;
    if (det == 0.0) return false;
    const double tx = double(o.x) - double(v0.x), ty = double(o.y) - double(v0.y), tz = double(o.z) - double(v0.z);
    const double uu = ((tx * px + ty * py) + tz * pz) / det;
    if (!(uu >= 0.0) || uu > 1.0) return false;
    const double qx = ty * double(e1.z) - tz * double(e1.y), qy = tz * double(e1.x) - tx * double(e1.z), qz = tx * double(e1.y) - ty * double(e1.x);
    const double vv = ((double(d.x) * qx + double(d.y) * qy) + double(d.z) * qz) / det;
    if (!(vv >= 0.0) || uu + vv > 1.0) return false;
    const double tt = ((double(e2.x) * qx + double(e2.y) * qy) + double(e2.z) * qz) / det;
    if (!(tt > double(tmin) && tt < double(tmax))) return false;
    t = float(tt); u = float(uu); v = float(vv);
    return true;
}

// Moeller-Trumbore, two-sided, det == 0 -> miss, accept iff tmin < t < tmax; a candidate whose solution contradicts itself is decided again in
// binary64 (decision vi in DESIGN.md).
__device__ __forceinline__ bool ray_triangle(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float tmin, float tmax,
                                             float &t, float &u, float &v) {
    f3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    f3 tvec = o - v0;
    float uu = dot3(tvec, pvec) * inv;
    if (!(uu >= 0.0f) || uu > 1.0f) return false;
    f3 qvec = cross3(tvec, e1);
    float vv = dot3(d, qvec) * inv;
    if (!(vv >= 0.0f) || uu + vv > 1.0f) return false;
    float tt = dot3(e2, qvec) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    t = tt; u = uu; v = vv;
    if (solution_consistent(o, d, v0, e1, e2, tt, uu, vv)) return true;
    return mt_binary64(o, d, v0, e1, e2, tmin, tmax, t, u, v);
}

// Moeller-Trumbore's comparisons without ray_triangle()'s early returns: the same operations in the same order on the same operands (a lane the
// branching form would have sent home early computes on and fails the same comparison at the end; det == 0 gives inf / NaN quotients, which fail
// every comparison, and is tested explicitly as well).  Used by the queue kernels' leaf stage, where the early returns buy nothing (some lane of
// the wave always goes on) and cost a second memory round trip: the compiler sinks the load of v0 behind the `det == 0` return, so every triangle
// test waited for memory twice.  true = a CANDIDATE; the caller accepts it if solution_consistent() and decides it again with mt_binary64() if not.
__device__ __forceinline__ bool mt_candidate(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float tmin, float tmax, float &t, float &u, float &v) {
    const f3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    const float inv = 1.0f / det;
    const f3 tvec = o - v0;
    const float uu = dot3(tvec, pvec) * inv;
    const f3 qvec = cross3(tvec, e1);
    const float vv = dot3(d, qvec) * inv;
    const float tt = dot3(e2, qvec) * inv;
    t = tt; u = uu; v = vv;
    return det != 0.0f && uu >= 0.0f && !(uu > 1.0f) && vv >= 0.0f && !(uu + vv > 1.0f) && tt > tmin && tt < tmax;
}

// Slab test of one child box against [tmin, tlimit]; NaNs from 0 * inf drop out of fminf/fmaxf
// (IEEE minNum/maxNum), which can only enlarge the interval, i.e. stays conservative.
__device__ __forceinline__ bool box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv,
                                         float tmin, float tlimit, float &tnear) {
    float t0 = (lox - o.x) * inv.x, t1 = (hix - o.x) * inv.x;
    float tn = fmaxf(tmin, fminf(t0, t1)), tf = fminf(tlimit, fmaxf(t0, t1));
    t0 = (loy - o.y) * inv.y; t1 = (hiy - o.y) * inv.y;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    t0 = (loz - o.z) * inv.z; t1 = (hiz - o.z) * inv.z;
    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    tnear = tn;
    return tn <= tf;
}

// "bvh_frame": what the slab tests see of a ray.  The boxes of all node forms live in the frame DeviceScene::frame (row i = axis i in world
// coordinates); a walker rotates origin and direction once per ray for them and intersects triangles in world space as ever (boxes only cull:
// the rotation's rounding, ~1e-6 |x|, is two orders below the boxes' padding).  frame_on is uniform: a scalar branch around 18 FMAs.
__device__ __forceinline__ f3 frame_rotate(const float *R, f3 p) {
    return f3{ (R[0] * p.x + R[1] * p.y) + R[2] * p.z, (R[3] * p.x + R[4] * p.y) + R[5] * p.z, (R[6] * p.x + R[7] * p.y) + R[8] * p.z };
}
__device__ __forceinline__ void box_ray(const DeviceScene &sc, f3 ro, f3 rd, f3 &bo, f3 &bd) {
    bo = ro; bd = rd;
    if (sc.frame_on) { bo = frame_rotate(sc.frame, ro); bd = frame_rotate(sc.frame, rd); }
}
// the bounds of a tile's ray origins in the frame: the box of the rotated box (centre R c, half extent |R| h -- a superset of the rotated origins)
__device__ __forceinline__ void box_bounds(const DeviceScene &sc, f3 &omin, f3 &omax) {
    if (!sc.frame_on || !(omin.x <= omax.x)) return;
    const float *R = sc.frame;
    const f3 c = f3{ 0.5f * omin.x + 0.5f * omax.x, 0.5f * omin.y + 0.5f * omax.y, 0.5f * omin.z + 0.5f * omax.z };
    const f3 h = f3{ (omax.x - c.x) * 1.000001f + 1e-6f, (omax.y - c.y) * 1.000001f + 1e-6f, (omax.z - c.z) * 1.000001f + 1e-6f };
    const f3 rc = frame_rotate(R, c);
    const f3 rh = f3{ (fabsf(R[0]) * h.x + fabsf(R[1]) * h.y) + fabsf(R[2]) * h.z, (fabsf(R[3]) * h.x + fabsf(R[4]) * h.y) + fabsf(R[5]) * h.z,
                      (fabsf(R[6]) * h.x + fabsf(R[7]) * h.y) + fabsf(R[8]) * h.z };
    omin = f3{ rc.x - rh.x, rc.y - rh.y, rc.z - rh.z };
    omax = f3{ rc.x + rh.x, rc.y + rh.y, rc.z + rh.z };
}

// Per-lane BVH2 walk with the traversal stack in LDS (stack[level * kTraceBlock + thread]: conflict free; STRIDE 1: a private array).
// ANY_HIT: gl_RayFlagsTerminateOnFirstHitEXT | SkipClosestHitShader (raygen.rgen:39,51) -- returns at the
// first accepted triangle; the boolean result does not depend on the visiting order.
// !ANY_HIT: closest hit = min t, ties broken by the smaller flat triangle index; subtrees are pruned with
// tnear > best t only (strict), so equal-t candidates are always examined.
// ALPHA: rays of the raytraced render path traced with gl_RayFlagsNoOpaqueEXT (raygen_test_alpha.rgen:20,
// closesthit_test_alpha.rchit:42): every candidate first runs shadow_anyhit.rahit, an ignored candidate does not exist.
__device__ bool alpha_ignored(const DeviceScene &sc, uint32_t tri_index, float u, float v);

template <bool ANY_HIT, bool ALPHA = false, int STRIDE = kTraceBlock>
__device__ __forceinline__ bool traverse(const DeviceScene &sc, f3 o, f3 d, float tmin, float tmax, int *stack, Hit &best,
                                         uint32_t &overflow) {
    if (sc.node_count == 0) return false;
    f3 bo, bd;
    box_ray(sc, o, d, bo, bd);                                  // "bvh_frame": the slab tests' ray; the triangle tests below keep (o, d)
    const f3 inv = f3{ 1.0f / bd.x, 1.0f / bd.y, 1.0f / bd.z };
    bool found = false;
    float tbest = tmax;
    int sp = 0;
    int cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
            const float4 q0 = np[0], q1 = np[1], q2 = np[2];
            const int4 q3 = reinterpret_cast<const int4 *>(np)[3];
            float tn0, tn1;
            const bool h0 = box_test(q0.x, q0.z, q1.x, q0.y, q0.w, q1.y, bo, inv, tmin, tbest, tn0);
            const bool h1 = box_test(q1.z, q2.x, q2.z, q1.w, q2.y, q2.w, bo, inv, tmin, tbest, tn1);
            if (h0 && h1) {
                const bool first0 = tn0 <= tn1;
                const int nearc = first0 ? q3.x : q3.y, farc = first0 ? q3.y : q3.x;
                if (sp < kTraceStack) { stack[sp * STRIDE] = farc; ++sp; } else { overflow = 1; }
                cur = nearc;
                continue;
            }
            if (h0) { cur = q3.x; continue; }
            if (h1) { cur = q3.y; continue; }
        } else {
            const uint32_t v = ~uint32_t(cur);
            const uint32_t first = v >> 2, count = (v & 3u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + first + i);
                const float4 a = tp[0], b = tp[1];
                const float4 c = tp[2];
                float t, u, w;
                if (ray_triangle(o, d, f3{ a.x, a.y, a.z }, f3{ a.w, b.x, b.y }, f3{ b.z, b.w, c.x }, tmin, tmax, t, u, w)) {
                    if (ALPHA && alpha_ignored(sc, first + i, u, w)) continue;
                    if (ANY_HIT) return true;
                    const uint32_t flat = __float_as_uint(c.w);
                    if (!found || t < best.t || (t == best.t && flat < best.flat)) {
                        found = true;
                        best.t = t; best.u = u; best.v = w; best.tri_index = first + i; best.flat = flat;
                        tbest = t;
                    }
                }
            }
        }
        if (sp == 0) break;
        --sp;
        cur = stack[sp * STRIDE];
    }
    return found;
}

// ---------------------------------------------------------------------------------------------
// image helpers (linear, row-major, tightly packed)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f4 load_rgba16f(const void *img, uint32_t W, uint32_t x, uint32_t y) {
    const uint2 raw = reinterpret_cast<const uint2 *>(img)[size_t(y) * W + x];
    return f4{ half_bits_to_float(uint16_t(raw.x & 0xffffu)), half_bits_to_float(uint16_t(raw.x >> 16)),
               half_bits_to_float(uint16_t(raw.y & 0xffffu)), half_bits_to_float(uint16_t(raw.y >> 16)) };
}
__device__ __forceinline__ void store_rgba16f(void *img, uint32_t W, uint32_t x, uint32_t y, float a, float b, float c, float d) {
    uint2 raw;
    raw.x = uint32_t(float_to_half_bits(a)) | (uint32_t(float_to_half_bits(b)) << 16);
    raw.y = uint32_t(float_to_half_bits(c)) | (uint32_t(float_to_half_bits(d)) << 16);
    reinterpret_cast<uint2 *>(img)[size_t(y) * W + x] = raw;
}
__device__ __forceinline__ void store_rg16f(void *img, uint32_t W, uint32_t x, uint32_t y, float a, float b) {
    reinterpret_cast<uint32_t *>(img)[size_t(y) * W + x] =
        uint32_t(float_to_half_bits(a)) | (uint32_t(float_to_half_bits(b)) << 16);
}

// glsl_common.h:118-122
__device__ __forceinline__ f3 get_world_space_position(const vhr_per_frame_data &pfd, float depth, float u, float v) {
    const f4 r = mat4_mul(pfd.camera_viewproj_inverse, f4{ u * 2.0f - 1.0f, v * 2.0f - 1.0f, depth, 1.0f });
    return f3{ r.x / r.w, r.y / r.w, r.z / r.w };
}

// ---------------------------------------------------------------------------------------------
// texture(): LOD 0, per-texture sampler, software bilinear (float tolerance, not bit-exact)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_coord(int i, int n, int mode) {
    if (mode == 2) return min(max(i, 0), n - 1);
    if (mode == 1) {
        const int p = 2 * n;
        int m = i % p;
        if (m < 0) m += p;
        return m < n ? m : p - 1 - m;
    }
    int m = i % n;
    if (m < 0) m += n;
    return m;
}
__device__ __forceinline__ f4 fetch_texel(const DeviceTexture &t, int x, int y) {
    const uchar4 p = reinterpret_cast<const uchar4 *>(t.texels)[size_t(y) * t.width + x];
    f4 r;
    if (t.format == VHR_FORMAT_R8G8B8A8_SRGB) { r.x = c_srgb_lut[p.x]; r.y = c_srgb_lut[p.y]; r.z = c_srgb_lut[p.z]; }
    else { r.x = p.x * (1.0f / 255.0f); r.y = p.y * (1.0f / 255.0f); r.z = p.z * (1.0f / 255.0f); }
    r.w = p.w * (1.0f / 255.0f);
    return r;
}
__device__ f4 sample_texture(const DeviceScene &sc, int idx, float u, float v) {
    if (idx < 0 || uint32_t(idx) >= sc.texture_count) return f4{ 0, 0, 0, 0 };
    const DeviceTexture t = sc.textures[idx];
    float x = u * float(t.width), y = v * float(t.height);
    if (t.mag_filter == 0)
        return fetch_texel(t, wrap_coord(int(floorf(x)), int(t.width), t.address_u), wrap_coord(int(floorf(y)), int(t.height), t.address_v));
    x -= 0.5f; y -= 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float fx = x - fx0, fy = y - fy0;
    const int x0 = wrap_coord(int(fx0), int(t.width), t.address_u), x1 = wrap_coord(int(fx0) + 1, int(t.width), t.address_u);
    const int y0 = wrap_coord(int(fy0), int(t.height), t.address_v), y1 = wrap_coord(int(fy0) + 1, int(t.height), t.address_v);
    const f4 a = fetch_texel(t, x0, y0), b = fetch_texel(t, x1, y0), c = fetch_texel(t, x0, y1), d = fetch_texel(t, x1, y1);
    f4 r;
    r.x = (a.x * (1.0f - fx) + b.x * fx) * (1.0f - fy) + (c.x * (1.0f - fx) + d.x * fx) * fy;
    r.y = (a.y * (1.0f - fx) + b.y * fx) * (1.0f - fy) + (c.y * (1.0f - fx) + d.y * fx) * fy;
    r.z = (a.z * (1.0f - fx) + b.z * fx) * (1.0f - fy) + (c.z * (1.0f - fx) + d.z * fx) * fy;
    r.w = (a.w * (1.0f - fx) + b.w * fx) * (1.0f - fy) + (c.w * (1.0f - fx) + d.w * fx) * fy;
    return r;
}

// ---------------------------------------------------------------------------------------------
// K2: reflection_hit.rchit:10-72
// ---------------------------------------------------------------------------------------------
struct TriAttributes { float uvx, uvy; f3 normal; f3 object_pos; };

__device__ __forceinline__ TriAttributes interpolate(const DeviceScene &sc, const vhr_primitive &prim, uint32_t tri, float u, float v) {
    const uint32_t i0 = sc.indices[prim.index_offset + 3 * tri + 0];
    const uint32_t i1 = sc.indices[prim.index_offset + 3 * tri + 1];
    const uint32_t i2 = sc.indices[prim.index_offset + 3 * tri + 2];
    const vhr_vertex &a = sc.vertices[prim.vertex_offset + i0];
    const vhr_vertex &b = sc.vertices[prim.vertex_offset + i1];
    const vhr_vertex &c = sc.vertices[prim.vertex_offset + i2];
    const float bx = 1.0f - u - v, by = u, bz = v;                                     // rchit:21
    TriAttributes r;
    r.uvx = a.uv0[0] * bx + b.uv0[0] * by + c.uv0[0] * bz;                             // rchit:22
    r.uvy = a.uv0[1] * bx + b.uv0[1] * by + c.uv0[1] * bz;
    r.normal = f3{ a.normal[0] * bx + b.normal[0] * by + c.normal[0] * bz,            // rchit:23 (object space)
                   a.normal[1] * bx + b.normal[1] * by + c.normal[1] * bz,
                   a.normal[2] * bx + b.normal[2] * by + c.normal[2] * bz };
    r.object_pos = f3{ a.pos[0] * bx + b.pos[0] * by + c.pos[0] * bz, a.pos[1] * bx + b.pos[1] * by + c.pos[1] * bz,
                       a.pos[2] * bx + b.pos[2] * by + c.pos[2] * bz };
    return r;
}

// gbuf.vert:21 passes the vertex tangent through; the rasteriser interpolates it like the normal
__device__ __forceinline__ f4 interpolate_tangent(const DeviceScene &sc, const vhr_primitive &prim, uint32_t tri, float u, float v) {
    const vhr_vertex &a = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * tri + 0]];
    const vhr_vertex &b = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * tri + 1]];
    const vhr_vertex &c = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * tri + 2]];
    const float bx = 1.0f - u - v, by = u, bz = v;
    return f4{ a.tangent[0] * bx + b.tangent[0] * by + c.tangent[0] * bz, a.tangent[1] * bx + b.tangent[1] * by + c.tangent[1] * bz,
               a.tangent[2] * bx + b.tangent[2] * by + c.tangent[2] * bz, a.tangent[3] * bx + b.tangent[3] * by + c.tangent[3] * bz };
}

// second_bounce (may be nullptr): the documented 2-bounce extension (BASELINE config 5; the reference traces one bounce and
// declares recursion depth 2, pipeline.cpp:285): the payload of a mirror ray traced from this hit replaces / blends into the
// specular term exactly like composition.frag:141-149 blends the first bounce at the primary hit.  hit_position / hit_normal
// (optional) return the world-space hit point and the shader's N for the caller to build that ray.
__device__ f4 shade_reflection_hit(const DeviceScene &sc, const vhr_per_frame_data &pfd, const Hit &h, const f4 *second_bounce = nullptr,
                                   f3 *hit_position = nullptr, f3 *hit_normal = nullptr) {
    const BvhTri &bt = sc.tris[h.tri_index];
    const vhr_primitive &prim = sc.primitives[bt.prim];                                 // rchit:11
    const TriAttributes at = interpolate(sc, prim, bt.tri, h.u, h.v);
    const f3 position = mat4_mul_point(prim.transform, at.object_pos);                  // rchit:24
    f3 albedo;
    if (prim.material.base_color_texture == -1) {                                       // rchit:27-32
        albedo = f3{ prim.material.base_color[0], prim.material.base_color[1], prim.material.base_color[2] };
    } else {
        const f4 t = sample_texture(sc, prim.material.base_color_texture, at.uvx, at.uvy);
        albedo = f3{ t.x, t.y, t.z };
    }
    float metallic = prim.material.metallic_factor, roughness = prim.material.roughness_factor;
    if (prim.material.metallic_roughness_texture != -1) {                               // rchit:35-39
        const f4 mr = sample_texture(sc, prim.material.metallic_roughness_texture, at.uvx, at.uvy);
        metallic *= mr.y;
        roughness *= mr.z;
    }
    const f3 cam = f3{ pfd.camera_view_inverse[12], pfd.camera_view_inverse[13], pfd.camera_view_inverse[14] };
    const f3 V = normalize3(cam - position);                                            // rchit:42
    const f3 L = -f3{ pfd.directional_light.direction[0], pfd.directional_light.direction[1], pfd.directional_light.direction[2] };
    const f3 N = at.normal;                                                             // rchit:44 (not normalised)
    const f3 H = normalize3(L + V);
    roughness = fminf(fmaxf(roughness, 0.04f), 1.0f);                                   // rchit:53-55
    metallic = fminf(fmaxf(metallic, 0.0f), 1.0f);
    const float ambient_factor = VHR_PI_INVERSE * 0.2f;                                 // rchit:59
    const f3 li = f3{ pfd.directional_light.intensity[0], pfd.directional_light.intensity[1], pfd.directional_light.intensity[2] };
    const f3 lc = f3{ pfd.directional_light.color[0], pfd.directional_light.color[1], pfd.directional_light.color[2] };
    const f3 f0 = f3{ 0.04f * (1.0f - metallic) + albedo.x * metallic, 0.04f * (1.0f - metallic) + albedo.y * metallic,
                      0.04f * (1.0f - metallic) + albedo.z * metallic };                // rchit:63-64
    const f3 F = fresnel_schlick(f0, H, V);
    const f3 ambient = albedo * ambient_factor;                                         // rchit:67
    const f3 dp = f3{ (1.0f - F.x) * (1.0f - metallic), (1.0f - F.y) * (1.0f - metallic), (1.0f - F.z) * (1.0f - metallic) };
    const f3 diffuse = f3{ dp.x * albedo.x / VHR_PI, dp.y * albedo.y / VHR_PI, dp.z * albedo.z / VHR_PI };
    const float dg = D_GGX(roughness, N, H) * G_GGX(roughness, N, V, L);
    const float denom = 4.0f * fmaxf(dot3(N, V), 0.0f) * fmaxf(dot3(N, L), 0.0f);
    const float invd = 1.0f / fmaxf(denom, 1e-6f);
    const f3 specular = f3{ dg * F.x * invd, dg * F.y * invd, dg * F.z * invd };
    const float nl = fmaxf(dot3(N, L), 0.0f);
    if (hit_position) *hit_position = position;
    if (hit_normal) *hit_normal = N;
    if (second_bounce) {
        const f3 dl = mul3(mul3(diffuse * nl, li), lc);                                 // composition.frag:138 without the shadow factor
        f3 sl = mul3(mul3(specular * nl, li), lc);                                      // :139
        const f3 refl = f3{ second_bounce->x, second_bounce->y, second_bounce->z };
        if (metallic == 1.0f) sl = refl;                                                // :141-149
        else sl = f3{ sl.x * (1.0f - roughness) + refl.x * roughness, sl.y * (1.0f - roughness) + refl.y * roughness,
                      sl.z * (1.0f - roughness) + refl.z * roughness };
        const f3 lighting2 = (ambient + dl) + sl;                                       // :160
        return f4{ lighting2.x, lighting2.y, lighting2.z, 1.0f };
    }
    const f3 lit = mul3(mul3((diffuse + specular) * nl, li), lc);                       // rchit:70
    const f3 lighting = ambient + lit;
    return f4{ lighting.x, lighting.y, lighting.z, 1.0f };
}

// world-space hit point and the shader's N of reflection_hit.rchit:11-24,44 (what shade_reflection_hit returns through
// hit_position / hit_normal, without the shading)
__device__ __forceinline__ void hit_position_normal(const DeviceScene &sc, const Hit &h, f3 &position, f3 &normal) {
    const BvhTri &bt = sc.tris[h.tri_index];
    const vhr_primitive &prim = sc.primitives[bt.prim];
    const TriAttributes at = interpolate(sc, prim, bt.tri, h.u, h.v);
    position = mat4_mul_point(prim.transform, at.object_pos);
    normal = at.normal;
}

// The mirror ray of raygen.rgen:59-65 with the optional second bounce: a mirror ray from the first hit about the shader's N
// (normalised, facing the incoming ray), origin biased like raygen.rgen:29, shaded by reflection_hit.rchit without recursion.
template <int STRIDE = kTraceBlock>
__device__ __forceinline__ f4 trace_reflection(const DeviceScene &sc, const vhr_per_frame_data &pfd, const vhr_trace_params &tp, f3 origin,
                                               f3 rdir, int *stack, uint32_t &overflow, bool &second_ray) {
    Hit hit;
    second_ray = false;
    if (!traverse<false, false, STRIDE>(sc, origin, rdir, tp.tmin, tp.tmax, stack, hit, overflow)) return f4{ 0.0f, 0.0f, 0.0f, 0.0f };   // reflection_miss.rmiss:7
    if (tp.reflections < 2) return shade_reflection_hit(sc, pfd, hit);
    f3 hp, hn;
    (void)shade_reflection_hit(sc, pfd, hit, nullptr, &hp, &hn);
    const f3 nn = normalize3(hn);
    const float ni = dot3(nn, rdir);
    const f3 nf = ni < 0.0f ? nn : -nn;
    const f3 d2 = rdir - nn * (2.0f * ni);
    const f3 o2 = hp + nf * tp.normal_bias;
    second_ray = true;
    Hit hit2;
    f4 second = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
    if (traverse<false, false, STRIDE>(sc, o2, d2, tp.tmin, tp.tmax, stack, hit2, overflow)) second = shade_reflection_hit(sc, pfd, hit2);
    return shade_reflection_hit(sc, pfd, hit, &second);
}

// ---------------------------------------------------------------------------------------------
// K1: raygen.rgen:14-66
// ---------------------------------------------------------------------------------------------
// "raygen_cost_order", what a queue kernel's launch gets: wave_cost != nullptr -> every wave leaves its lifetime there (index = its block's tiles *
// WAVES + wave); block_order != nullptr -> block b works on the tiles of block block_order[b] (the launch before last's blocks, longest-lived
// first); order_out != nullptr -> the launch's first block sorts the previous launch's `order_blocks` blocks by `cost_prev` into it before its own tile
struct CostOrderArgs {
    uint32_t *wave_cost = nullptr;
    const uint32_t *block_order = nullptr;
    const uint32_t *cost_prev = nullptr;
    uint32_t *order_out = nullptr;
    uint32_t order_blocks = 0;
};

struct RaygenArgs {
    DeviceScene scene;
    vhr_per_frame_data pfd;
    vhr_trace_params tp;
    const void *normals;     // RGBA16F
    const float *depth;      // D32F
    void *shadow_ao;         // RG16F
    void *reflections;       // RGBA16F or nullptr
    uint32_t width, height;  // launch size == image size
    uint32_t row_begin, row_end;
    uint32_t col_begin, col_end;     // screen tiles (vhr_set_tile): the columns the queue kernels trace -- col_begin a multiple of the tile
                                     // width, [0, width) otherwise; the per-pixel kernels trace whole rows (a superset)
    RayStats *stats;         // nullptr = off
    // "fuse_temporal": the queue kernel's tile epilogue runs svgf.comp for the tile's pixels (the dispatch the SVGF pass records next)
    uint32_t fuse_temporal;  // 0 = off
    TemporalArgs temporal;
    CostOrderArgs co;        // "raygen_cost_order" (the default queue kernel, the mirror-ray queue kernel)
};

__device__ __forceinline__ void pixel_of_thread(uint32_t &x, uint32_t &y, uint32_t row_begin) {
    // 16x16 pixel tile per block; wave w covers the 8x8 sub-tile (w & 1, w >> 1)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    y = row_begin + blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// raygen.rgen:26-55 for one covered pixel: its shadow ray and its AO rays one after another (the walker with the whole of decision (vi): traverse<> ->
// ray_triangle).  STRIDE: the stack's (kTraceBlock: the per-pixel kernels' LDS columns; 1: a private array).
template <int STRIDE>
__device__ __forceinline__ void pixel_visibility(const RaygenArgs &a, const uint32_t x, const uint32_t y, const float depth, int *stack, uint32_t &overflow,
                                                 f3 &P, f3 &N, f3 &origin, float &shadow_payload, float &ao_payload) {
    const uint32_t W = a.width, H = a.height;
    const float u = (float(x) + 0.5f) / float(W);                                        // rgen:15-16
    const float v = (float(y) + 0.5f) / float(H);
    uint32_t rng = seed_thread((y * H + x) * a.pfd.frame_index);                         // rgen:17 (LaunchSize.y)
    P = get_world_space_position(a.pfd, depth, u, v);                                    // rgen:26
    const f3 L = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1],
                      a.pfd.directional_light.direction[2] };                            // rgen:27
    const f4 nid = load_rgba16f(a.normals, W, x, y);                                     // rgen:28
    N = f3{ nid.x, nid.y, nid.z };
    origin = P + N * a.tp.normal_bias;                                                   // rgen:29
    Hit hit;

    float rnd1 = random01(rng);                                                          // rgen:32-33
    float rnd2 = random01(rng);
    shadow_payload = 1.0f;
    if (a.tp.shadow_enable) {
        const f3 cone_dir = normalize3(uniform_sample_cone(rnd1, rnd2, a.tp.cone_cos_max));   // rgen:34
        const f3 dir = onb_transform(L, cone_dir);                                       // rgen:35,40
        // rgen:37-41 issues this trace four times with identical arguments; once is equivalent
        const bool occluded = traverse<true, false, STRIDE>(a.scene, origin, dir, a.tp.tmin, a.tp.tmax, stack, hit, overflow);
        shadow_payload = occluded ? 0.0f : 1.0f;                                         // miss.rmiss:7
    }
    ao_payload = 0.0f;                                                                   // rgen:44-55
    for (uint32_t i = 0; i < a.tp.ao_spp; ++i) {
        rnd1 = random01(rng);
        rnd2 = random01(rng);
        const f3 rnd_dir = cosine_hemisphere(rnd1, rnd2);
        const f3 dir = onb_transform(N, rnd_dir);
        const bool occluded = traverse<true, false, STRIDE>(a.scene, origin, dir, a.tp.tmin, a.tp.ao_tmax, stack, hit, overflow);
        ao_payload += occluded ? 0.0f : 1.0f;
    }
    if (a.tp.ao_spp) ao_payload /= float(a.tp.ao_spp); else ao_payload = 1.0f;
}

// Decision (vi) in the any-hit queue kernel: a pixel one of whose rays met a candidate that contradicts itself is computed again, whole, by the per-pixel
// kernel's code (binary64 decisions inline) when its tile is done -- a call, so that the queue kernel's loops carry none of this (inlined in its leaf
// test the binary64 arithmetic cost the kernel 13 registers = a wave per SIMD, 1.3-1.5 % of the frame: profiles/r6_decision_vi_cost.txt).  `a` points at
// the launch's arguments where they lie in memory (the address of a by-value argument would copy all of it to every lane's scratch).  At 1080p:
// none to three pixels of a frame on the BASELINE stand-ins, up to ~70 on sponza_hard_rot (profiles/r6_decision_vi_cost.txt).
__device__ VHR_REDO_INLINE float2 redo_pixel_visibility(const RaygenArgs *a, const uint32_t x, const uint32_t y) {
    int st[kTraceStack];
    uint32_t overflow = 0;
    f3 P, N, origin;
    float shadow_payload, ao_payload;
    pixel_visibility<1>(*a, x, y, a->depth[size_t(y) * a->width + x], st, overflow, P, N, origin, shadow_payload, ao_payload);
    return float2{ shadow_payload, ao_payload };
}

// raygen.rgen:14-66 for one pixel
__device__ __forceinline__ void raygen_pixel(const RaygenArgs &a, const uint32_t x, const uint32_t y, int *stack, uint32_t &overflow, bool &covered, bool &second_ray) {
    const uint32_t W = a.width;
    const float depth = a.depth[size_t(y) * W + x];                                      // rgen:19
    if (depth == 0.0f) {                                                                 // rgen:20-24
        store_rg16f(a.shadow_ao, W, x, y, 1.0f, 1.0f);
        if (a.reflections) store_rgba16f(a.reflections, W, x, y, 0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    covered = true;
    f3 P, N, origin;
    float shadow_payload, ao_payload;
    pixel_visibility<kTraceBlock>(a, x, y, depth, stack, overflow, P, N, origin, shadow_payload, ao_payload);
    store_rg16f(a.shadow_ao, W, x, y, shadow_payload, ao_payload);                       // rgen:57

    if (a.reflections) {
        f4 payload = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
        if (a.tp.reflections) {                                                          // rgen:60-65
            const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };
            const f3 I = normalize3(P - cam);
            const float ni2 = 2.0f * dot3(N, I);
            const f3 rdir = I - N * ni2;                                                 // reflect(I, N)
            payload = trace_reflection(a.scene, a.pfd, a.tp, origin, rdir, stack, overflow, second_ray);
        }
        store_rgba16f(a.reflections, W, x, y, payload.x, payload.y, payload.z, payload.w);
    }
}

__global__ __launch_bounds__(kTraceBlock) void raygen_kernel(const RaygenArgs a, const Stamps st) {
    vhr_stamp(st);
    __shared__ int s_stack[kTraceStack * kTraceBlock];
    int *stack = s_stack + threadIdx.x;
    uint32_t x, y;
    pixel_of_thread(x, y, a.row_begin);
    uint32_t overflow = 0;
    bool covered = false, second_ray = false;
    if (x < a.width && y < a.row_end) raygen_pixel(a, x, y, stack, overflow, covered, second_ray);
    if (a.stats) {
        const unsigned long long cov = __ballot(covered), ovf = __ballot(overflow != 0), sec = __ballot(second_ray);
        if ((threadIdx.x & 63u) == 0) {
            if (cov) atomicAdd(&a.stats->covered_pixels, (unsigned long long)__popcll(cov));
            if (ovf) atomicAdd(&a.stats->stack_overflows, (unsigned long long)__popcll(ovf));
            if (sec) atomicAdd(&a.stats->second_bounce_rays, (unsigned long long)__popcll(sec));
        }
    }
}


// ---------------------------------------------------------------------------------------------
// K1 (work-queue form): the same visibility rays as raygen_kernel, scheduled for wave64 occupancy.
//
// raygen.rgen traces its rays one after another per pixel, so on a SIMD machine every lane of a wave waits
// for the slowest lane of EACH ray (measured: 43 % of VALU lanes active).  Here a 16x16-pixel block first
// compacts its covered pixels (wave ballot + one LDS atomic per wave), which defines a block-local queue of
// `covered * (1 + ao_spp)` any-hit rays, kind-major (all shadow rays, then AO sample 0, AO sample 1, ...) so
// that rays in flight together are of one kind.  Lanes pull rays from the queue whenever at least
// `refill_threshold` lanes of the wave are idle -- one LDS atomic per refill, ranks from the idle ballot
// ("wavefront-ballot compaction") -- regenerate the ray from (pixel, kind) with raygen.rgen's exact arithmetic
// (the RNG state is recomputed from the seed), and walk the BVH one node or leaf per loop trip.  Visibility
// results are integers accumulated in LDS (order independent), so the output is bit-identical to the
// sequential form: shadow = !any_hit, ao = float(visible) / float(spp).  The mirror ray (closest hit +
// shading, a different register budget) runs in reflection_kernel.
// ---------------------------------------------------------------------------------------------
// Slab test of one child box, (lo, hi) pairs per axis, as three packed FMAs against precomputed 1/d and -o/d.
// Box tests only cull (boxes are padded, NaNs drop out of min/max), so they are outside the exact-arithmetic
// contract: 1/d may come from v_rcp_f32 and the FMA may round differently from (lo - o) * inv without changing
// any result.
typedef float f2v __attribute__((ext_vector_type(2)));

// v_min / v_max / v_min3 / v_max3 spelled as instructions: fminf / fmaxf lower to llvm.minnum / maxnum, which under the
// kernel's IEEE mode get a canonicalising v_max_f32 x, x, x in front of every operand the compiler cannot prove quiet
// (14 extra instructions per node here).  The hardware ops already return the non-NaN operand, which is all the
// cull needs (and no NaN can arise: see cull_reciprocal).
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// A comparison as the 64-bit lane mask it is, and a select on such a mask.  Spelled out because a ballot of a bool the compiler holds as a
// lane mask comes back through a vector register when it is handed to an asm statement (v_cndmask 0 / 1 + v_cmp_ne: two instructions per
// mask, four per node visit, r5); masks combine on the scalar unit.  (Lanes that are switched off read 0.)
__device__ __forceinline__ unsigned long long cmp_le_mask(float a, float b) { unsigned long long m; asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cmp_gt_i32_mask_s(int uniform_a, int b) { unsigned long long m; asm("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(m) : "s"(uniform_a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cmp_eq_i32_mask_s(int uniform_a, int b) { unsigned long long m; asm("v_cmp_eq_i32_e64 %0, %1, %2" : "=s"(m) : "s"(uniform_a), "v"(b)); return m; }
__device__ __forceinline__ int select_mask(int if_clear, int if_set, unsigned long long m) { int d; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(m)); return d; }

__device__ __forceinline__ bool box_test_pk(f2v bx, f2v by, f2v bz, f3 inv, f3 noi, float tmin, float tlimit, float &tnear) {
    const f2v tx = __builtin_elementwise_fma(bx, f2v{ inv.x, inv.x }, f2v{ noi.x, noi.x });
    const f2v ty = __builtin_elementwise_fma(by, f2v{ inv.y, inv.y }, f2v{ noi.y, noi.y });
    const f2v tz = __builtin_elementwise_fma(bz, f2v{ inv.z, inv.z }, f2v{ noi.z, noi.z });
    const float tn = hw_max3(hw_min(tx.x, tx.y), hw_min(ty.x, ty.y), hw_max(hw_min(tz.x, tz.y), tmin));
    const float tf = hw_min3(hw_max(tx.x, tx.y), hw_max(ty.x, ty.y), hw_min(hw_max(tz.x, tz.y), tlimit));
    tnear = tn;
    return tn <= tf;
}

constexpr int kQueueBlock = 64;
constexpr int kStackSentinel = int(0x80000000u);   // not a node (>= 0) and not a leaf code the builder can emit

// One box in centre / half-extent form (a cut entry, build_tile_cut) against one ray: `ainv` = |1/d|.  The three centre terms are one packed
// FMA + one plain one, an axis's (near, far) pair is ONE packed FMA (-h and +h through neg_lo on the same register) that needs no min / max:
// 5 FMAs + 4 min / max + the compare instead of box_test_pk's 3 + 10 + 1.  Culling only: the box is the (lo, hi) box grown by a few ulp.
__device__ __forceinline__ unsigned long long box_test_ch1(float cx, float cy, float cz, float hx, float hy, float hz, f3 inv, f3 ainv, f3 noi, float tmin, float tlimit) {
    const f2v cxy = __builtin_elementwise_fma(f2v{ cx, cy }, f2v{ inv.x, inv.y }, f2v{ noi.x, noi.y });
    const float ciz = __builtin_fmaf(cz, inv.z, noi.z);
    const f2v x = __builtin_elementwise_fma(f2v{ -hx, hx }, f2v{ ainv.x, ainv.x }, f2v{ cxy.x, cxy.x });
    const f2v y = __builtin_elementwise_fma(f2v{ -hy, hy }, f2v{ ainv.y, ainv.y }, f2v{ cxy.y, cxy.y });
    const f2v z = __builtin_elementwise_fma(f2v{ -hz, hz }, f2v{ ainv.z, ainv.z }, f2v{ ciz, ciz });
    const float tn = hw_max3(x.x, y.x, hw_max(z.x, tmin));
    const float tf = hw_min3(x.y, y.y, hw_min(z.y, tlimit));
    return cmp_le_mask(tn, tf);
}

// A refilled ray against the tile's cut (build_tile_cut): the subtrees it hits go on its (empty) stack, the deepest -- the one closest to the
// origins -- on top; those that do not fit the LDS levels are remembered in `emask`.  Written without branches like the node step: the link is
// stored above the top whatever the test says (a slot above the top may hold anything) and the test's mask is the carry that moves the top.
// 13 vector instructions per entry (r5; 19 with box_test_pk and a predicated push); the masks are lane masks in scalar registers (cmp_le_mask).  Ends with the top entry popped into `cur`.
__device__ __forceinline__ void cut_to_stack(const float4 (*cut)[2], const uint32_t cut_n, int *stack, const uint32_t stack_levels, f3 inv, f3 noi, float tmin_v, float tlimit,
                                             int &cur, int &sp, uint32_t &emask) {
    const f3 ainv = f3{ fabsf(inv.x), fabsf(inv.y), fabsf(inv.z) };
    emask = 0;
    sp = 0;
    for (uint32_t e = 0; e < cut_n; ++e) {
        const float4 b0 = cut[e][0], b1 = cut[e][1];              // (cx, cy, cz, hx), (hy, hz, link, -): LDS broadcasts
        const unsigned long long hit = box_test_ch1(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, inv, ainv, noi, tmin_v, tlimit);
        const unsigned long long fits = cmp_gt_i32_mask_s(int(stack_levels) - 2, sp);          // sp + 2 < stack_levels (so sp + 1 <= stack_levels - 1: the store below stays inside the lane's rows)
        stack[(uint32_t(sp) + 1u) * kQueueBlock] = __float_as_int(b1.z);
        {
            unsigned long long carry_out;
            asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(sp), "=s"(carry_out) : "v"(sp), "s"(hit & fits));
        }
        if (hit & ~fits) emask = uint32_t(select_mask(int(emask), int(emask | (1u << e)), hit & ~fits));      // (wave-uniform and rare: a ray that hits more entries than the LDS levels hold)
    }
    if (sp > 0) { cur = stack[uint32_t(sp) * kQueueBlock]; --sp; } else cur = kStackSentinel;
}

// Both child boxes of a centre / half-extent node (BvhNodeCH) against one ray: `ainv` = |1/d|.  The centre terms of the two boxes
// share one packed FMA per axis; a box's (near, far) pair of an axis is ONE packed FMA (-h and +h through neg_lo on the same
// register), so no per-axis min / max is needed: 9 FMAs + 8 min / max per node instead of 6 + 20.  Culling only (see BvhNodeCH).
template <typename V4>
__device__ __forceinline__ void box_pair_ch(const V4 q0, const V4 q1, const V4 q2, f3 inv, f3 ainv, f3 noi, float tmin, float tlimit,
                                            float &tn0, float &tn1, float &tf0, float &tf1) {
    const f2v cix = __builtin_elementwise_fma(f2v{ q0.x, q0.y }, f2v{ inv.x, inv.x }, f2v{ noi.x, noi.x });
    const f2v ciy = __builtin_elementwise_fma(f2v{ q0.z, q0.w }, f2v{ inv.y, inv.y }, f2v{ noi.y, noi.y });
    const f2v ciz = __builtin_elementwise_fma(f2v{ q1.x, q1.y }, f2v{ inv.z, inv.z }, f2v{ noi.z, noi.z });
    const f2v x0 = __builtin_elementwise_fma(f2v{ -q1.z, q1.z }, f2v{ ainv.x, ainv.x }, f2v{ cix.x, cix.x });
    const f2v y0 = __builtin_elementwise_fma(f2v{ -q1.w, q1.w }, f2v{ ainv.y, ainv.y }, f2v{ ciy.x, ciy.x });
    const f2v z0 = __builtin_elementwise_fma(f2v{ -q2.x, q2.x }, f2v{ ainv.z, ainv.z }, f2v{ ciz.x, ciz.x });
    const f2v x1 = __builtin_elementwise_fma(f2v{ -q2.y, q2.y }, f2v{ ainv.x, ainv.x }, f2v{ cix.y, cix.y });
    const f2v y1 = __builtin_elementwise_fma(f2v{ -q2.z, q2.z }, f2v{ ainv.y, ainv.y }, f2v{ ciy.y, ciy.y });
    const f2v z1 = __builtin_elementwise_fma(f2v{ -q2.w, q2.w }, f2v{ ainv.z, ainv.z }, f2v{ ciz.y, ciz.y });
    tn0 = hw_max3(x0.x, y0.x, hw_max(z0.x, tmin));
    tn1 = hw_max3(x1.x, y1.x, hw_max(z1.x, tmin));
    tf0 = hw_min3(x0.y, y0.y, hw_min(z0.y, tlimit));
    tf1 = hw_min3(x1.y, y1.y, hw_min(z1.y, tlimit));
}
template <typename V4>
__device__ __forceinline__ void box_pair_ch(const V4 q0, const V4 q1, const V4 q2, f3 inv, f3 ainv, f3 noi, float tmin, float tlimit,
                                            bool &h0, bool &h1, float &tn0, float &tn1) {
    float tf0, tf1;
    box_pair_ch(q0, q1, q2, inv, ainv, noi, tmin, tlimit, tn0, tn1, tf0, tf1);
    h0 = tn0 <= tf0;
    h1 = tn1 <= tf1;
}

// Both child boxes of a 32-byte node (BvhNode16, r3c): centres and half extents are HALVES, child 0 in the low and child 1 in the high half of
// each word, and v_fma_mix_f32 widens the half operand inside the instruction -- 18 plain FMAs, no unpacking (a packed fp32 FMA
// occupies the SIMD twice as long as a plain one: the 9 packed FMAs of the fp32 form are the same lane operations).  `noi` is
// -(o - scene centre) / d: the centres are relative to the scene centre.  Culling only (see BvhNode16).
#define VHR_MIX(name, mods, b_open, b_close, sel)                                                                                      \
    __device__ __forceinline__ float name(uint32_t h, float b, float c) {                                                             \
        float r;                                                                                                                       \
        asm("v_fma_mix_f32 %0, " mods "%1, " b_open "%2" b_close ", %3 op_sel:[" sel ",0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b), "v"(c));      \
        return r;                                                                                                                      \
    }
VHR_MIX(mix_lo, "", "", "", "0")
VHR_MIX(mix_hi, "", "", "", "1")
// h * |b| + c and -h * |b| + c: the ray's |1 / d| as an operand modifier of the instruction (no register copy of it)
VHR_MIX(mix_lo_abs, "", "|", "|", "0")
VHR_MIX(mix_hi_abs, "", "|", "|", "1")
VHR_MIX(mix_lo_neg_abs, "-", "|", "|", "0")
VHR_MIX(mix_hi_neg_abs, "-", "|", "|", "1")
#undef VHR_MIX

__device__ __forceinline__ void box_pair_ch16(const uint32_t cx, const uint32_t cy, const uint32_t cz, const uint32_t hx, const uint32_t hy, const uint32_t hz,
                                              f3 inv, f3 noi, float tmin, float tlimit, float &tn0, float &tn1, float &tf0, float &tf1) {
    const float cx0 = mix_lo(cx, inv.x, noi.x), cx1 = mix_hi(cx, inv.x, noi.x);
    const float cy0 = mix_lo(cy, inv.y, noi.y), cy1 = mix_hi(cy, inv.y, noi.y);
    const float cz0 = mix_lo(cz, inv.z, noi.z), cz1 = mix_hi(cz, inv.z, noi.z);
    tn0 = hw_max3(mix_lo_neg_abs(hx, inv.x, cx0), mix_lo_neg_abs(hy, inv.y, cy0), hw_max(mix_lo_neg_abs(hz, inv.z, cz0), tmin));
    tn1 = hw_max3(mix_hi_neg_abs(hx, inv.x, cx1), mix_hi_neg_abs(hy, inv.y, cy1), hw_max(mix_hi_neg_abs(hz, inv.z, cz1), tmin));
    tf0 = hw_min3(mix_lo_abs(hx, inv.x, cx0), mix_lo_abs(hy, inv.y, cy0), hw_min(mix_lo_abs(hz, inv.z, cz0), tlimit));
    tf1 = hw_min3(mix_hi_abs(hx, inv.x, cx1), mix_hi_abs(hy, inv.y, cy1), hw_min(mix_hi_abs(hz, inv.z, cz1), tlimit));
}
__device__ __forceinline__ void box_pair_ch16(const uint32_t cx, const uint32_t cy, const uint32_t cz, const uint32_t hx, const uint32_t hy, const uint32_t hz,
                                              f3 inv, f3 noi, float tmin, float tlimit, bool &h0, bool &h1, float &tn0, float &tn1) {
    float tf0, tf1;
    box_pair_ch16(cx, cy, cz, hx, hy, hz, inv, noi, tmin, tlimit, tn0, tn1, tf0, tf1);
    h0 = tn0 <= tf0;
    h1 = tn1 <= tf1;
}

// One visit's worth of a 48-byte node (BvhNode48): three 16-byte loads, the half extents widened back to fp32 words (first of a
// pair = the word itself, second = one shift), links from the third load.  Feeds box_pair_ch unchanged.
struct Node48Words { float4 q0, q1, q2; int2 links; };
__device__ __forceinline__ Node48Words load_node48(const BvhNode48 *nodes, int cur) {
    // `cur` is the node's BYTE offset (index * 48: what the 48-byte nodes' inner links hold, r3 -- one v_mul_lo_u32, a quarter-rate
    // instruction, less per visit)
    const float4 *np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + uint32_t(cur));
    const float4 r0 = np[0], r1 = np[1], r2 = np[2];
    Node48Words n;
    n.q0 = r0;
    n.q1 = make_float4(r1.x, r1.y, r1.z, __uint_as_float(__float_as_uint(r1.z) << 16));
    n.q2 = make_float4(r1.w, __uint_as_float(__float_as_uint(r1.w) << 16), r2.x, __uint_as_float(__float_as_uint(r2.x) << 16));
    n.links = int2{ __float_as_int(r2.y), __float_as_int(r2.z) };
    return n;
}

// 1/d for the slab test.  A zero (or denormal) component must not become inf: fma(lo, inf, -o*inf) is NaN on one
// side of the slab only, which would cull boxes the ray is inside of.  1e30 keeps lo * inv finite for any scene
// coordinate and classifies "parallel to the slab" correctly: inside -> (-huge, +huge), outside -> both beyond tmax.
// (The exact direction d itself is untouched: Moeller-Trumbore never sees this value.)
__device__ __forceinline__ float cull_reciprocal(float d) {
    // v_rcp_f32 (1 ulp) instead of the correctly rounded division (12 instructions, three of them per ray): at scene scale
    // (boxes reach t of a few tens) an ulp of 1/d moves a slab distance by ~1e-5, two orders below the boxes' padding
    const float r = __builtin_amdgcn_rcpf(d);
    return fabsf(d) < 1e-30f ? copysignf(1e30f, d) : r;
}

// Rank of this lane among the set bits of a wave mask: v_mbcnt_lo / v_mbcnt_hi (no per-lane (1 << lane) - 1 mask to keep in registers)
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}


// raygen.rgen:32-53 for one (pixel, kind): the ray direction, exact arithmetic
__device__ __forceinline__ f3 ray_direction(const vhr_trace_params &tp, uint32_t seed, uint32_t kind, f3 L, f3 N) {
    uint32_t rng = seed;
    float rnd1 = random01(rng), rnd2 = random01(rng);                                        // rgen:32-33
    if (kind == 0) {                                                                         // rgen:34-41
        const f3 cone_dir = normalize3(uniform_sample_cone(rnd1, rnd2, tp.cone_cos_max));
        return onb_transform(L, cone_dir);
    }
    for (uint32_t i = 0; i < kind; ++i) { rnd1 = random01(rng); rnd2 = random01(rng); }       // rgen:46-48
    return onb_transform(N, cosine_hemisphere(rnd1, rnd2));                                  // rgen:49-51
}

// Every wave owns one 8x8-pixel tile and runs its own queue; a block is WAVES such waves side by side (a CU
// accepts at most 16 workgroups, so single-wave blocks cap occupancy at 4 waves per SIMD: measured).  Waves of a
// block share nothing and never synchronise with each other.
template <int WAVES>
__device__ __forceinline__ void tile_pixel(uint32_t block_tile, uint32_t tiles_x, uint32_t wave, uint32_t local, uint32_t row_begin, uint32_t tile_rows,
                                           uint32_t &x, uint32_t &y, uint32_t col_begin = 0u) {
    const uint32_t by = block_tile / tiles_x, bx = block_tile - by * tiles_x;
    x = col_begin + (bx * WAVES + wave) * 8u + (local & 7u);
    y = row_begin + by * tile_rows + (local >> 3);          // tile_rows < 8: the lanes of the tile's missing rows stay out of range
}

// orders this wave's LDS writes before its later LDS reads by other lanes (no cross-wave communication exists)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The shared descent ("the cut"): the rays of a tile all start within centimetres of each other, and each of them would spend most of its
// ~16 node visits walking from the root down to the boxes around that spot -- every box on the way contains the origin, so every ray
// hits it whatever its direction.  The wave therefore makes that descent ONCE per tile (uniformly: follow the inner child whose
// box contains the bounding box of the tile's ray origins, keep the other child) and leaves a CUT of the tree in LDS: up to
// kCutMax subtrees that together cover all geometry.  A ray then starts by testing the cut's boxes (a short uniform loop over
// LDS broadcasts, every refilled lane busy) and walks only the subtrees it hits.  Box tests only cull, so results are unchanged.
constexpr int kCutMax = 16;          // (12 and 24 entries measured flat around 16)

// An upper bound of |onb_transform(n, v)| / |v|, i.e. of the largest singular value of the Frisvad basis (c0, c1, n) that
// common.glsl:80-93 builds around the G-buffer normal.  The normal is a rounded half vector, not a unit vector, and near
// n.z = -1 the basis amplifies that error by 1 / (1 + n.z): AO directions are NOT unit vectors, so the reach of an AO ray is
// tmax * |d|, not tmax.  Gershgorin on the Gram matrix of the three columns, 2 % of slack for this function's own rounding
// and for |v| of the cosine-hemisphere sample (1 within a few ulp).
__device__ __forceinline__ float onb_norm_bound(f3 n) {
    f3 c0, c1;
    if (n.z < -0.9999999f) {
        c0 = f3{ 0.0f, -1.0f, 0.0f };
        c1 = f3{ -1.0f, 0.0f, 0.0f };
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b = ((-n.x) * n.y) * a;
        c0 = f3{ 1.0f - (n.x * n.x) * a, b, -n.x };
        c1 = f3{ b, 1.0f - (n.y * n.y) * a, -n.y };
    }
    const float g00 = dot3(c0, c0), g11 = dot3(c1, c1), g22 = dot3(n, n);
    const float g01 = fabsf(dot3(c0, c1)), g02 = fabsf(dot3(c0, n)), g12 = fabsf(dot3(c1, n));
    const float row = fmaxf(fmaxf(g00 + g01 + g02, g01 + g11 + g12), g02 + g12 + g22);
    const float bound = sqrtf(row) * 1.02f;
    return bound == bound ? bound : 3.0e38f;            // a NaN normal prunes nothing
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f *uniform_f4_ptr;         // constant address space + uniform address = SMEM loads
typedef const __attribute__((address_space(4))) v4i *uniform_i4_ptr;

// The same descent written on wave-uniform values (rounds 2-4): every comparison and every move of a box is a vector instruction for the whole
// wave, ~100 per level.  Kept for the closest-hit walks and the raytraced path, whose launches are not bound by vector issue (and whose kernels
// the lane-parallel form below does not compile for: the backend's verifier rejects a private-to-flat cast next to it).
__device__ __forceinline__ uint32_t build_tile_cut_uniform(const DeviceScene &sc, f3 omin, f3 omax, float4 (*s_cut)[2], uint32_t lane, float reach = 3.0e38f,
                                                   const int max_entries = kCutMax, const int link_bytes = int(sizeof(BvhNode48)), const f3 centre = f3{ 0.0f, 0.0f, 0.0f }) {
    // ---- bounds of the origins (wave reduction), then the descent; every lane computes the same thing ----
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        omin.x = fminf(omin.x, __shfl_xor(omin.x, off)); omin.y = fminf(omin.y, __shfl_xor(omin.y, off)); omin.z = fminf(omin.z, __shfl_xor(omin.z, off));
        omax.x = fmaxf(omax.x, __shfl_xor(omax.x, off)); omax.y = fmaxf(omax.y, __shfl_xor(omax.y, off)); omax.z = fmaxf(omax.z, __shfl_xor(omax.z, off));
        reach = fmaxf(reach, __shfl_xor(reach, off));
    }
    // wave-uniform from here on, and told so: the descent then runs on scalar registers and scalar branches
    auto uni = [](float f) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(f))); };
    omin = f3{ uni(omin.x), uni(omin.y), uni(omin.z) }; omax = f3{ uni(omax.x), uni(omax.y), uni(omax.z) };
    box_bounds(sc, omin, omax);                          // "bvh_frame": the nodes' boxes are in the frame
    reach = uni(reach);
    const float reach2 = reach * reach;                 // inf for "no pruning" (and for anything that overflows)
    // The cut while it is being built: entry e lives in lane e (box, link).
    float e_lx = 0.0f, e_hx = 0.0f, e_ly = 0.0f, e_hy = 0.0f, e_lz = 0.0f, e_hz = 0.0f;
    int e_link = 0;
    uint32_t cut_n = 0;
    auto gap2_of = [&](float lx, float hx, float ly, float hy, float lz, float hz) {
        const float gx = fmaxf(fmaxf(lx - omax.x, omin.x - hx), 0.0f), gy = fmaxf(fmaxf(ly - omax.y, omin.y - hy), 0.0f),
                    gz = fmaxf(fmaxf(lz - omax.z, omin.z - hz), 0.0f);
        return (gx * gx + gy * gy) + gz * gz;
    };
    auto put = [&](uint32_t slot, float lx, float hx, float ly, float hy, float lz, float hz, int link) {
        if (lane == slot) { e_lx = lx; e_hx = hx; e_ly = ly; e_hy = hy; e_lz = lz; e_hz = hz; e_link = link; }
    };
    auto add_entry = [&](float lx, float hx, float ly, float hy, float lz, float hz, int link) {
        const float g2 = gap2_of(lx, hx, ly, hy, lz, hz);
        if (g2 > reach2) return;                                                             // out of every ray's reach
        put(cut_n, lx, hx, ly, hy, lz, hz, link);
        ++cut_n;
    };
    int node = 0;
    float fb[6] = { -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f };          // box of `node` (the root: everything)
    bool open = true;                                                                      // `node` still waits for its entry
    for (int it = 0; it < max_entries - 2; ++it) {
        // a uniform address in the constant address space: the node arrives through the scalar cache (s_load), not through the
        // vector memory path the walk itself is bound by
        const uniform_f4_ptr np = (uniform_f4_ptr)(uintptr_t)(sc.nodes + node);
        const v4f q0 = np[0], q1 = np[1], q2 = np[2];
        const v4i vl = ((uniform_i4_ptr)np)[3];
        const int2 links = int2{ vl.x, vl.y };
        const bool in0 = q0.x <= omin.x && omax.x <= q0.y && q0.z <= omin.y && omax.y <= q0.w && q1.x <= omin.z && omax.z <= q1.y;
        const bool in1 = q1.z <= omin.x && omax.x <= q1.w && q2.x <= omin.y && omax.y <= q2.y && q2.z <= omin.z && omax.z <= q2.w;
        const bool follow0 = links.x >= 0 && in0, follow1 = !follow0 && links.y >= 0 && in1;
        if (!(follow0 || follow1)) {                                                      // the descent ends here: both children join the cut
            add_entry(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, links.x);
            add_entry(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, links.y);
            open = false;
            break;
        }
        if (follow0) {
            add_entry(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, links.y);
            fb[0] = q0.x; fb[1] = q0.y; fb[2] = q0.z; fb[3] = q0.w; fb[4] = q1.x; fb[5] = q1.y;
            node = links.x;
        } else {
            add_entry(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, links.x);
            fb[0] = q1.z; fb[1] = q1.w; fb[2] = q2.x; fb[3] = q2.y; fb[4] = q2.z; fb[5] = q2.w;
            node = links.y;
        }
    }
    if (open) add_entry(fb[0], fb[1], fb[2], fb[3], fb[4], fb[5], node);                  // the budget ran out: the subtree itself
    if (lane < cut_n) {
        // centre / half extent (box_test_ch1), the centre relative to `centre` (the walkers of the half-precision nodes keep their ray origins
        // relative to the scene's centre): c +- h contains [lo, hi] -- h carries 4 ulp of the magnitudes involved, the roundings of c, of
        // hi - c and of the shift are below one each.  The root's "everything" box (+-3e38) stays finite: c = 0, h = 3e38 (1 + 2.4e-7).
        auto ch = [](float lo, float hi, float shift, float &c, float &h) {
            const float mid = 0.5f * lo + 0.5f * hi;
            c = mid - shift;
            h = fmaxf(hi - mid, mid - lo);
            h += (fabsf(mid) + fabsf(shift) + h) * 2.4e-7f;
        };
        float cx, cy, cz, hx, hy, hz;
        ch(e_lx, e_hx, centre.x, cx, hx); ch(e_ly, e_hy, centre.y, cy, hy); ch(e_lz, e_hz, centre.z, cz, hz);
        s_cut[lane][0] = make_float4(cx, cy, cz, hx);
        s_cut[lane][1] = make_float4(hy, hz, __int_as_float(e_link >= 0 ? e_link * link_bytes : e_link), 0.0f);
    }
    wave_lds_sync();
    return cut_n;
}


// One entry of a tile's cut (build_tile_cut): what a lane knows about itself, and the store of a child's box in centre / half-extent form.
struct CutLane {
    f3 omin, omax;
    float reach2, shift;
    uint32_t axis;
    bool lo_lane;
    uint32_t my_child, lane;
    int link_bytes;
};
__device__ __forceinline__ void cut_add_entry(const CutLane cl, float4 (*s_cut)[2], uint32_t &cut_n, const float word, const int child, const int link) {
    if (cl.reach2 < 3.0e38f) {                                                               // (uniform; launches without shadow rays only)
        const int b0 = 6 * child;
        auto lane_word = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        const float lx = lane_word(word, b0), hx = lane_word(word, b0 + 1), ly = lane_word(word, b0 + 2), hy = lane_word(word, b0 + 3), lz = lane_word(word, b0 + 4),
                    hz = lane_word(word, b0 + 5);
        const float gx = fmaxf(fmaxf(lx - cl.omax.x, cl.omin.x - hx), 0.0f), gy = fmaxf(fmaxf(ly - cl.omax.y, cl.omin.y - hy), 0.0f),
                    gz = fmaxf(fmaxf(lz - cl.omax.z, cl.omin.z - hz), 0.0f);
        if ((gx * gx + gy * gy) + gz * gz > cl.reach2) return;                                // out of every ray's reach
    }
    // centre / half extent (box_test_ch1), the centre relative to `centre` (the walkers of the half-precision nodes keep their ray origins
    // relative to the scene's centre): c +- h contains [lo, hi] -- h carries 4 ulp of the magnitudes involved, the roundings of c, of
    // hi - c and of the shift are below one each.  The root's "everything" box (+-3e38) stays finite: c = 0, h = 3e38 (1 + 2.4e-7).
    const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(word), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]: the pair's other word
    const float mid = 0.5f * word + 0.5f * other;                                          // (lo lanes: word = lo, other = hi)
    const float c = mid - cl.shift;
    float h = fmaxf(other - mid, mid - word);
    h += (fabsf(mid) + fabsf(cl.shift) + h) * 2.4e-7f;
    // (the cut is an LDS array at every call site; said so, so that the stores are ds_write and not flat stores behind an address-space test)
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *const entry = (lds_float *)(&s_cut[0][0]) + cut_n * 8u;                       // (cx, cy, cz, hx), (hy, hz, link, -)
    if (cl.lo_lane && cl.my_child == uint32_t(child)) { entry[cl.axis] = c; entry[3u + cl.axis] = h; }
    if (cl.lane == 0u) entry[6] = __int_as_float(link >= 0 ? link * cl.link_bytes : link);
    ++cut_n;
}

// The shared descent of a tile (see CUT above): `omin` / `omax` are this lane's contribution to the bounds of the tile's ray
// origins (+-3e38 for lanes without one).  Leaves the cut in s_cut[0 .. n) -- centre / half extent: (cx, cy, cz, hx), (hy, hz, link, -) --
// and returns n, wave-uniform.  Entries are in path order: the deeper an entry, the closer its box to the origins.
// `reach` (this lane's contribution, 0 for lanes without rays; +inf = no pruning): an upper bound of how far any of the tile's
// rays can get from its origin, tmax * |d|.  A subtree whose box lies farther than that from the bounds of the origins cannot
// hold a hit of any of them and is left out of the cut -- decided once per tile instead of by a box test per ray.
// `link_bytes`: inner links of the finished cut are multiplied by it (48 for the walkers of the 48-byte nodes, whose links are byte offsets).
__device__ __forceinline__ uint32_t build_tile_cut(const DeviceScene &sc, f3 omin, f3 omax, float4 (*s_cut)[2], uint32_t lane, float reach = 3.0e38f,
                                                   const int max_entries = kCutMax, const int link_bytes = int(sizeof(BvhNode48)), const f3 centre = f3{ 0.0f, 0.0f, 0.0f }) {
    // ---- bounds of the origins (wave reduction), then the descent; every lane computes the same thing ----
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        omin.x = fminf(omin.x, __shfl_xor(omin.x, off)); omin.y = fminf(omin.y, __shfl_xor(omin.y, off)); omin.z = fminf(omin.z, __shfl_xor(omin.z, off));
        omax.x = fmaxf(omax.x, __shfl_xor(omax.x, off)); omax.y = fmaxf(omax.y, __shfl_xor(omax.y, off)); omax.z = fmaxf(omax.z, __shfl_xor(omax.z, off));
        reach = fmaxf(reach, __shfl_xor(reach, off));
    }
    auto uni = [](float f) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(f))); };
    omin = f3{ uni(omin.x), uni(omin.y), uni(omin.z) }; omax = f3{ uni(omax.x), uni(omax.y), uni(omax.z) };
    box_bounds(sc, omin, omax);                          // "bvh_frame": the nodes' boxes are in the frame
    reach = uni(reach);
    const float reach2 = reach * reach;                 // inf for "no pruning" (and for anything that overflows)
    // The descent is wave-uniform, but this chip's scalar unit has no float arithmetic: written on uniform values, every comparison, every
    // min / max and every move of a box is a vector instruction for the whole wave -- about a hundred per level, 800 per tile, a seventh of the
    // any-hit launch's instructions (r5).  So the lanes take a WORD of the 64-byte node each: lane w (of every sixteen) loads word w -- child 0's
    // box in words 0-5 as (lo, hi) pairs per axis, child 1's in 6-11, the links in 12 and 13 -- and tests it against its own bound of the
    // origins; the answers come back as a lane mask, the links by v_readlane, and everything that steers the descent is scalar.  An entry's
    // centre / half-extent form is computed by the three even lanes that hold its lo words (the hi word comes from the neighbour by DPP) and
    // stored straight to the cut in LDS.  ~20 vector instructions per level.
    const uint32_t w = lane & 15u;
    const bool is_hi = (w & 1u) != 0u;
    const uint32_t axis = (w % 6u) >> 1;
    const bool lo_lane = lane < 12u && !is_hi;          // the lanes that hold a box's lo words (one sixteen of the wave writes)
    const uint32_t my_child = w >= 6u ? 1u : 0u;
    // a lo word passes iff word <= omin[axis], a hi word iff omax[axis] <= word, i.e. -word <= -omax[axis]: one comparison with the sign flipped
    const uint32_t flip = is_hi ? 0x80000000u : 0u;
    const float ref = is_hi ? -(axis == 0u ? omax.x : axis == 1u ? omax.y : omax.z) : (axis == 0u ? omin.x : axis == 1u ? omin.y : omin.z);
    const float shift = axis == 0u ? centre.x : axis == 1u ? centre.y : centre.z;
    const uint32_t word_offset = w * 4u;
    uint32_t cut_n = 0;
    const CutLane cl{ omin, omax, reach2, shift, axis, lo_lane, my_child, lane, link_bytes };
    // `word`: a node's words, one per lane; `child`'s box joins the cut with `link`
#define add_entry(word, child, link) cut_add_entry(cl, s_cut, cut_n, word, child, link)
    int node = 0;
    float pword = is_hi ? 3.0e38f : -3.0e38f;                                              // the words `node`'s own box came in: for the root, "everything" as child 0
    int pchild = 0;
    bool open = true;                                                                      // `node` still waits for its entry
    const char *const base = reinterpret_cast<const char *>(sc.nodes);
    for (int it = 0; it < max_entries - 2; ++it) {
        const float word = *reinterpret_cast<const float *>(base + (uint32_t(node) * uint32_t(sizeof(BvhNode)) + word_offset));
        const uint32_t inside = uint32_t(cmp_le_mask(__uint_as_float(__float_as_uint(word) ^ flip), ref));      // bit w: word w keeps the origins inside
        const bool in0 = (inside & 0x3fu) == 0x3fu, in1 = (inside & 0xfc0u) == 0xfc0u;
        const int link0 = __builtin_amdgcn_readlane(__float_as_int(word), 12), link1 = __builtin_amdgcn_readlane(__float_as_int(word), 13);
        const bool follow0 = link0 >= 0 && in0, follow1 = !follow0 && link1 >= 0 && in1;
        if (!(follow0 || follow1)) {                                                      // the descent ends here: both children join the cut
            add_entry(word, 0, link0);
            add_entry(word, 1, link1);
            open = false;
            break;
        }
        if (follow0) { add_entry(word, 1, link1); pchild = 0; node = link0; }
        else { add_entry(word, 0, link0); pchild = 1; node = link1; }
        pword = word;
    }
    if (open) {                                                                            // the budget ran out: the subtree itself
        if (pchild == 0) add_entry(pword, 0, node); else add_entry(pword, 1, node);
    }
#undef add_entry
    wave_lds_sync();
    return cut_n;
}

// "raygen_cost_order": the blocks of an earlier launch sorted by cost, heaviest first, by ONE block of the ray-tracing launch (its first: it starts at
// once and has the whole launch to finish its own tile afterwards).  A block's cost is its longest-lived wave's lifetime; the blocks fall into 8
// classes of cost relative to the maximum, and a stable counting sort puts the heaviest class first -- inside a class the blocks keep their
// row-major order (neighbouring tiles share nodes; a full sort gives that up: round 2's "longest tiles first").  Whatever the lifetimes hold, the
// result is a permutation of 0 .. n_blocks - 1: the order is a speed hint, never correctness.  `lds`: 8 * 64 * WAVES words of scratch.
template <int WAVES>
__device__ __forceinline__ void order_blocks_by_cost(const uint32_t *__restrict__ wave_cost, const uint32_t n_blocks, uint32_t *__restrict__ order, uint32_t *lds) {
    constexpr uint32_t NT = 64u * WAVES, C = 8u;
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_blocks + NT - 1u) / NT, b0 = min(t * per, n_blocks), b1 = min(b0 + per, n_blocks);     // a contiguous chunk per thread
    auto cost = [&](uint32_t b) { uint32_t c = 0; for (uint32_t w = 0; w < uint32_t(WAVES); ++w) c = max(c, wave_cost[b * uint32_t(WAVES) + w]); return c; };
    uint32_t mx = 0;
    for (uint32_t b = b0; b < b1; ++b) mx = max(mx, cost(b));
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, uint32_t(__shfl_xor(int(mx), off)));
    if ((t & 63u) == 0u) lds[t >> 6] = mx;
    __syncthreads();
    for (uint32_t w = 0; w < uint32_t(WAVES); ++w) mx = max(mx, lds[w]);
    __syncthreads();
    const float scale = float(C) / float(max(1u, mx));
    auto cls = [&](uint32_t c) { return (C - 1u) - min(C - 1u, uint32_t(float(c) * scale)); };    // 0 = the heaviest (any deterministic map will do)
    uint32_t mine[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) mine[c] = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t k = cls(cost(b));
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) mine[c] += k == c ? 1u : 0u;
    }
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) lds[c * NT + t] = mine[c];
    __syncthreads();
    for (uint32_t off = 1; off < C * NT; off <<= 1) {         // inclusive scan over (class-major, thread-minor)
        uint32_t v[C];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) { const uint32_t i = c * NT + t; v[c] = i >= off ? lds[i - off] : 0u; }
        __syncthreads();
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) lds[c * NT + t] += v[c];
        __syncthreads();
    }
    uint32_t pos[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) pos[c] = lds[c * NT + t] - mine[c];       // inclusive -> exclusive
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t k = cls(cost(b));
        uint32_t p = 0;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) { p = k == c ? pos[c] : p; pos[c] += k == c ? 1u : 0u; }
        // (rotated by one: the LIGHTEST block goes to the front -- the next launch's first block, which does this sort before its own tile)
        order[p + 1u == n_blocks ? 0u : p + 1u] = b;
    }
    __syncthreads();                                        // the scratch is the waves' traversal stacks from here on
}

// raygen_queue_kernel<WAVES, COMPACT, SPILL, STATS>: every wave owns one 8x8-pixel tile (tile_rows < 8: fewer rows) and runs its own queue of
// `covered x (1 + ao_spp)` rays, kind-major, the shadow rays last.  COMPACT: the walk reads the 32-byte half-precision nodes (BvhNode16, two loads
// per visit; trees whose boxes do not fit the half range walk the 48-byte fp32 nodes instead).  SPILL: stack entries beyond the LDS levels
// live in scratch.  STATS: in-kernel counters and timers (vhr_set_ray_statistics).  Per covered pixel the wave keeps 5 words in LDS -- the ray
// origin and the G-buffer normal as the halves it is -- and recomputes the pixel's seed and the ray's direction at refill with raygen.rgen's
// exact arithmetic.
template <int WAVES, bool COMPACT, bool SPILL, bool STATS, bool FUSE = false>
__global__ __launch_bounds__(kQueueBlock *WAVES) __attribute__((amdgpu_waves_per_eu(VHR_K1_WAVES_MIN, 8))) void raygen_queue_kernel(const RaygenArgs a, const uint32_t stack_levels, const uint32_t refill_threshold,
                                                                          const uint32_t block_tiles_x, const uint32_t early_exit, const uint32_t tile_rows, const uint32_t steal_threshold, const Stamps st) {
    vhr_stamp(st);
    RayStats *const stats = STATS ? a.stats : nullptr;    // !STATS: counters and timers below are dead code (fewer VGPRs)
    extern __shared__ int s_dyn[];                    // per wave: (stack_levels + 3) x 64 ints
    const unsigned long long t_start = stats ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_setup = 0, t_refill = 0, t_nodes = 0, t_leaves = 0, n_refills = 0;
    __shared__ uint32_t s_vis_all[WAVES][kQueueBlock];    // bit k: the pixel's ray of kind k (0 = shadow, 1.. = AO) found an occluder
    __shared__ float s_ray_all[WAVES][5][kQueueBlock];    // per covered pixel: ray origin (3), the normal's half bits (2)
    __shared__ uint8_t s_list_all[WAVES][kQueueBlock];    // compacted covered pixels
    __shared__ float4 s_cut_all[WAVES][kCutMax][2];       // (lo.x, hi.x, lo.y, hi.y), (lo.z, hi.z, link, -)
    // (the compiler cannot know that threadIdx.x >> 6 is the same in every lane of a wave: said explicitly, what derives
    // from it -- the LDS bases, the wave's tile -- stays in scalar registers)
    const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    uint32_t (&s_vis)[kQueueBlock] = s_vis_all[wave];
    float (&s_ray)[5][kQueueBlock] = s_ray_all[wave];
    uint8_t (&s_list)[kQueueBlock] = s_list_all[wave];
    float4 (&s_cut)[kCutMax][2] = s_cut_all[wave];
    // LDS stack rows: [0] = sentinel (what a pop of the empty stack returns), [1 .. stack_levels] = entries
    // 0 .. stack_levels-1, [stack_levels+1, +2] = dummies that absorb the accesses of entries living in scratch
    const unsigned long long t_cost0 = a.co.wave_cost ? __builtin_readcyclecounter() : 0ull;
    if constexpr (!STATS) {
        // "raygen_cost_order": the launch's first block orders the previous launch's blocks for the next one (8 * 64 * WAVES words <= the block's
        // (stack_levels + 3) * 64 * WAVES words of stack: the host asks for it only with >= 5 stack levels)
        if (a.co.order_out && blockIdx.x == 0u) order_blocks_by_cost<WAVES>(a.co.cost_prev, a.co.order_blocks, a.co.order_out, reinterpret_cast<uint32_t *>(s_dyn));
    }
    int *stack = s_dyn + wave * (stack_levels + 3u) * kQueueBlock + lane;
    stack[0] = kStackSentinel;
    const uint32_t W = a.width, H = a.height;
    uint32_t x, y;
    // "raygen_cost_order": the blocks that lived longest two launches ago start first (a launch ends with its last wave; a long-lived wave that
    // starts late is what the launch's end waits for)
    const uint32_t block_tile = a.co.block_order ? a.co.block_order[blockIdx.x] : blockIdx.x;
    tile_pixel<WAVES>(block_tile, block_tiles_x, wave, lane, a.row_begin, tile_rows, x, y, a.col_begin);
    const bool in_range = x < a.col_end && y < a.row_end && (lane >> 3) < tile_rows;
    bool covered = false;
    float depth = 0.0f;
    if (in_range) {
        depth = a.depth[size_t(y) * W + x];                                                  // rgen:19
        covered = depth != 0.0f;
        if (!covered) store_rg16f(a.shadow_ao, W, x, y, 1.0f, 1.0f);                         // rgen:20-21
    }
    s_vis[lane] = 0;
    const uint32_t first_kind = a.tp.shadow_enable ? 0u : 1u;
    const uint32_t last_kind = a.tp.ao_spp;           // kinds first_kind .. last_kind
    // without shadow rays the queue holds AO rays only and the cut is pruned to their reach (with them in the queue, letting the AO rays skip
    // the entries beyond their reach was measured: the test per entry costs sponza_proc what it saves bistro_proc, r4)
    const bool ao_only = first_kind != 0u;
    // A pixel's visibility word holds one "blocked" bit per ray kind (what lets several lanes share a ray: "raygen_steal") while the kinds fit a
    // word; with more than 30 AO samples it holds bit 0 for the shadow ray and, from bit 8 up, the COUNT of AO rays that escaped, and a ray is
    // walked by the one lane that fetched it.  Bit 31 (kRedoPixel) in either form: decision (vi) wants the pixel computed again (below).
    const bool kind_bits = last_kind <= 30u;
    const f3 L = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1], a.pfd.directional_light.direction[2] };
    f3 omin = f3{ 3.0e38f, 3.0e38f, 3.0e38f }, omax = f3{ -3.0e38f, -3.0e38f, -3.0e38f };   // bounds of the tile's ray origins
    float ao_reach = 0.0f;                                                                   // bound of tmax * |d| over this pixel's AO rays
    if (covered) {
        // ---- raygen.rgen:15-29 once per pixel (shared by all of the pixel's rays) ----
        const float u = (float(x) + 0.5f) / float(W);
        const float v = (float(y) + 0.5f) / float(H);
        const f3 P = get_world_space_position(a.pfd, depth, u, v);                           // rgen:26
        const uint2 nraw = reinterpret_cast<const uint2 *>(a.normals)[size_t(y) * W + x];    // rgen:28 (R16G16B16A16: nx ny | nz id)
        const f3 N = f3{ half_bits_to_float(uint16_t(nraw.x & 0xffffu)), half_bits_to_float(uint16_t(nraw.x >> 16)), half_bits_to_float(uint16_t(nraw.y & 0xffffu)) };
        const f3 origin = P + N * a.tp.normal_bias;                                          // rgen:29
        s_ray[0][lane] = origin.x; s_ray[1][lane] = origin.y; s_ray[2][lane] = origin.z;
        s_ray[3][lane] = __uint_as_float(nraw.x); s_ray[4][lane] = __uint_as_float(nraw.y);
        if (ao_only) ao_reach = a.tp.ao_tmax * onb_norm_bound(N);      // (wave-uniform condition)
        omin = origin; omax = origin;
    }
    const unsigned long long cov_mask = __ballot(covered);
    const uint32_t ncov = uint32_t(__popcll(cov_mask));
    if (covered) s_list[lane_rank(cov_mask)] = uint8_t(lane);
    wave_lds_sync();
    const uint32_t total = (a.scene.node_count == 0) ? 0u : ncov * (1u + last_kind - first_kind);
    uint32_t cut_n = 0;
    if (total) cut_n = build_tile_cut(a.scene, omin, omax, s_cut, lane, ao_only ? ao_reach : 3.0e38f, kCutMax, COMPACT ? int(sizeof(BvhNode16)) : int(sizeof(BvhNode48)),
                                      COMPACT ? f3{ a.scene.centre[0], a.scene.centre[1], a.scene.centre[2] } : f3{ 0.0f, 0.0f, 0.0f });
    const uint32_t n_cut_entries = cut_n;
    uint32_t emask = 0;                               // cut entries this lane's ray hits that did not fit its LDS stack
    if (stats) t_setup = __builtin_readcyclecounter() - t_start;

    f3 ro = f3{ 0, 0, 0 }, rd = f3{ 0, 0, 1 }, rinv = f3{ 0, 0, 0 }, noi = f3{ 0, 0, 0 }, ainv = f3{ 0, 0, 0 };
    float tmax = 0.0f;
    int cur = 0, sp = 0;
#if VHR_K1_POSTPONE
    int parked = kStackSentinel;                      // "postponed leaf": a leaf this lane has reached and not tested yet (kStackSentinel = none)
#endif
    int sbase = 0;                                    // "raygen_steal": stack rows 1 .. sbase have been taken by other lanes (row sbase holds a sentinel)
    uint32_t pix = 0, kind = 0;
    bool has = false;
    uint32_t next = 0;                                // queue head: wave-uniform, lives in a register
    uint32_t overflow = 0;
    uint32_t n_nodes = 0, n_leaves = 0, n_tris = 0, n_wave_trips = 0, n_drain_trips = 0;      // statistics (only flushed when stats)
    uint32_t n_drain_le4 = 0, n_drain_le8 = 0, n_drain_le16 = 0;
    // Stack entries beyond the LDS levels spill to a small private (scratch) array: any-hit walks rarely hold more than
    // a dozen pending subtrees, so the LDS part can be much shallower than the tree -- more waves per CU -- without
    // giving up the guarantee that kTraceStack entries can never overflow (the builder bounds the depth).
    int spill[SPILL ? kSpillStack : 1];
    const float tmin = a.tp.tmin;
    float tmin_v = tmin;                              // one VGPR copy for the asm-operand min/max of the slab test
    asm volatile("" : "+v"(tmin_v));
    for (;;) {
        // ---- refill idle lanes from the tile's ray queue (ranks from the idle ballot) ----
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = uint32_t(__popcll(idle));
        const unsigned long long t0 = stats ? __builtin_readcyclecounter() : 0ull;
        const bool queue_dry = next >= total;         // (before this trip's refill: the steal below reads the idle ballot taken above)
        if (next < total && (n_idle >= refill_threshold || n_idle == 64u)) {     // wave-uniform condition
            __builtin_amdgcn_s_setprio(0);            // (see below)
            ++n_refills;
            const uint32_t r = next + lane_rank(idle);
            next += n_idle;
            if (!has && r < total) {
                // k = r / ncov without the integer division (~25 instructions): the queue is kind-major, k < kinds
                uint32_t k = 0, rr = r;
                while (rr >= ncov) { rr -= ncov; ++k; }
                kind = last_kind - k;                 // the shadow rays (kind 0) at the END of the queue (r2: the AO rays' long drains overlap them)
                pix = s_list[rr];
                ro = f3{ s_ray[0][pix], s_ray[1][pix], s_ray[2][pix] };
                const uint32_t nxy = __float_as_uint(s_ray[3][pix]), nzw = __float_as_uint(s_ray[4][pix]);
                const uint32_t px = x - (lane & 7u) + (pix & 7u), py = y - (lane >> 3) + (pix >> 3);     // the tile's origin + the pixel's place in it
                rd = ray_direction(a.tp, seed_thread((py * H + px) * a.pfd.frame_index), kind, L,                            // rgen:17
                                   f3{ half_bits_to_float(uint16_t(nxy & 0xffffu)), half_bits_to_float(uint16_t(nxy >> 16)), half_bits_to_float(uint16_t(nzw & 0xffffu)) });
                tmax = kind == 0 ? a.tp.tmax : a.tp.ao_tmax;                                 // rgen:40,52
                f3 bo, bd;
                box_ray(a.scene, ro, rd, bo, bd);             // "bvh_frame": the slab tests' ray (ro, rd stay the triangle tests')
                rinv = f3{ cull_reciprocal(bd.x), cull_reciprocal(bd.y), cull_reciprocal(bd.z) };
                // COMPACT boxes are relative to the scene centre: shift the origin used by the slab test (only)
                const f3 oc = COMPACT ? f3{ bo.x - a.scene.centre[0], bo.y - a.scene.centre[1], bo.z - a.scene.centre[2] } : bo;
                noi = f3{ -(oc.x * rinv.x), -(oc.y * rinv.y), -(oc.z * rinv.z) };
                if (!COMPACT) ainv = f3{ fabsf(rinv.x), fabsf(rinv.y), fabsf(rinv.z) };
                sbase = 0;
#if VHR_K1_POSTPONE
                parked = kStackSentinel;
#endif
                cut_to_stack(s_cut, cut_n, stack, stack_levels, rinv, noi, tmin_v, tmax, cur, sp, emask);      // (its boxes are relative to the centre `noi` is)
                has = true;
            }
        }
        // ---- the queue is dry: idle lanes take pending subtrees off the busy lanes' stacks ----
        // Any hit = OR over the subtrees a ray touches, in any order and by any lane: a lane with nothing left to fetch takes the top stack entry
        // of a busy lane and walks it for the same ray (the ray's registers come over by ds_bpermute).  One entry per busy lane and trip: the
        // walkers of a long ray double from trip to trip.  A lane whose ray another lane has meanwhile found blocked stops.
        if (steal_threshold && kind_bits && queue_dry && n_idle >= steal_threshold && n_idle != 64u) {         // wave-uniform
            if (has && ((s_vis[pix] >> kind) & 1u)) has = false;                                  // (the idle ballot above is one trip old for such a lane: it steals next trip)
            const bool donor = has && sp > sbase && uint32_t(sbase) < stack_levels;               // rows sbase + 1 .. sp are its own; row sbase + 1 is in LDS
            const unsigned long long dmask = __ballot(donor);
            if (dmask) {
                if (donor) s_list[lane_rank(dmask)] = uint8_t(lane);                              // (the list of covered pixels is not needed any more: the queue is dry)
                wave_lds_sync();
                const uint32_t nd = uint32_t(__popcll(dmask));
                const uint32_t r = lane_rank(idle);
                const bool thief = ((idle >> lane) & 1ull) && r < nd;
                const uint32_t d = thief ? uint32_t(s_list[r]) : lane;
                const int drow = __shfl(sbase, int(d)) + 1;                                       // the donor's LOWEST entry: the far child pushed first, the largest subtree it holds
                const int link = (stack - lane + d)[(thief ? uint32_t(drow) : 0u) * kQueueBlock];
                const float t_ox = __shfl(ro.x, int(d)), t_oy = __shfl(ro.y, int(d)), t_oz = __shfl(ro.z, int(d));
                const float t_dx = __shfl(rd.x, int(d)), t_dy = __shfl(rd.y, int(d)), t_dz = __shfl(rd.z, int(d));
                const float t_ix = __shfl(rinv.x, int(d)), t_iy = __shfl(rinv.y, int(d)), t_iz = __shfl(rinv.z, int(d));
                const float t_nx = __shfl(noi.x, int(d)), t_ny = __shfl(noi.y, int(d)), t_nz = __shfl(noi.z, int(d));
                const float t_ax = COMPACT ? 0.0f : __shfl(ainv.x, int(d)), t_ay = COMPACT ? 0.0f : __shfl(ainv.y, int(d)), t_az = COMPACT ? 0.0f : __shfl(ainv.z, int(d));
                const float t_tmax = __shfl(tmax, int(d));
                const uint32_t t_pk = uint32_t(__shfl(int(pix | (kind << 8)), int(d)));
                wave_lds_sync();
                if (donor && lane_rank(dmask) < n_idle) {                                         // its lowest entry has a taker: a sentinel in its place ends the donor's walk there
                    ++sbase;
                    stack[uint32_t(sbase) * kQueueBlock] = kStackSentinel;
                }
                if (thief) {
                    ro = f3{ t_ox, t_oy, t_oz }; rd = f3{ t_dx, t_dy, t_dz }; rinv = f3{ t_ix, t_iy, t_iz }; noi = f3{ t_nx, t_ny, t_nz };
                    if (!COMPACT) ainv = f3{ t_ax, t_ay, t_az };
                    tmax = t_tmax; pix = t_pk & 0xffu; kind = t_pk >> 8;
                    cur = link; sp = 0; sbase = 0; emask = 0;
#if VHR_K1_POSTPONE
                    parked = kStackSentinel;          // (a lane that dropped a ray another lane had found blocked may still hold one)
#endif
                    has = true;
                }
                if (stats) ++n_refills;
            }
        }
        if (!__any(has)) break;                       // nothing in flight and (since all lanes were idle) nothing left to fetch
#ifdef VHR_K1_COUNT_TRIPS
        if (stats) ++n_drain_trips;                   // (scratch builds: the statistics' drain counter counts the outer loop's trips instead)
#endif
        // The walk is a chain of dependent round trips, the refill a block of arithmetic that nothing waits for: a wave in the walk goes first when
        // both want the SIMD (r3c: -1.4 % on sponza_proc, nothing on bistro_proc)
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long t1 = stats ? __builtin_readcyclecounter() : 0ull;
        // ---- inner nodes: descend until this lane holds a leaf or its ray has run out of subtrees ----
        // The step is written without branches (selects + one unconditional LDS write and read per trip): divergent
        // if/else chains here cost more scalar exec-mask bookkeeping than the box tests themselves.  The write goes to
        // slot `sp` (level stack_levels - 1 at most: the builder bounds the depth), the read takes the current top.
        // Early exit: lanes leave this loop one by one (ray finished, or a leaf reached) and then idle until the LAST
        // walker leaves it.  With few leaf visits per any-hit ray that wait dominates (measured: 3.7 refills per 192-ray
        // tile, 44 % of the lanes active), so once the walkers have shrunk to a fraction of those that entered, the
        // loop is left: waiting lanes test their leaves, finished ones are refilled, the walkers resume where they were.
        bool found = false;
        const uint32_t nodes_before = n_nodes, tris_before = n_tris;
        const uint32_t walkers_in = uint32_t(__popcll(__ballot(has && cur >= 0)));
        while (has && cur >= 0) {
            if (uint32_t(__popcll(__ballot(true))) * 16u <= walkers_in * early_exit) break;   // ballot(true) = the walkers left; early_exit in 0..15 sixteenths, so the first trip always runs
            ++n_nodes;
            float tn0, tn1, tf0, tf1;
            int2 links;
            if (COMPACT) {
                // `cur` is the node's BYTE offset (index * 32); two 16-byte loads per visit
                const uint4 *np = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.scene.nodes16) + uint32_t(cur));
                const uint4 c0 = np[0], c1 = np[1];
                box_pair_ch16(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, rinv, noi, tmin_v, tmax, tn0, tn1, tf0, tf1);
                links = int2{ int(c1.z), int(c1.w) };
            } else {
                const Node48Words nw = load_node48(a.scene.nodes48, cur);
                links = nw.links;
                box_pair_ch(nw.q0, nw.q1, nw.q2, rinv, ainv, noi, tmin_v, tmax, tn0, tn1, tf0, tf1);
            }
            // The hit tests as lane masks in scalar registers (cmp_le_mask): what combines them is scalar work, what uses them takes them as they are.
            const unsigned long long h0 = cmp_le_mask(tn0, tf0), h1 = cmp_le_mask(tn1, tf1);
            // Child 0 is entered if it is hit and child 1 is not, or is not nearer; the child NOT entered goes to the stack slot (a
            // meaningful entry only when both are hit).  Three selects and two carries per visit (r3; the (near, far) form took ten
            // vector instructions: masks are scalar work, selects are not).
            const unsigned long long enter0 = h0 & (~h1 | cmp_le_mask(tn0, tn1));
            const int nearc = select_mask(links.y, links.x, enter0), farc = select_mask(links.x, links.y, enter0);
            // One address serves both accesses: row min(sp, L+1) holds the top entry (sp - 1; the sentinel when the stack
            // is empty), the row after it is where entry sp goes.  The write is harmless when !both (above the top).
            int *const row = stack + min(uint32_t(sp), stack_levels + 1u) * kQueueBlock;
            int top = row[0];
            row[kQueueBlock] = farc;
            // deep entries: behind a wave-uniform test, so that the hot path keeps plain ds_read / ds_write (an
            // if-converted "LDS or scratch" access becomes a flat load plus ten instructions of pointer selection)
            if (__any(uint32_t(sp) >= stack_levels)) {
                if (SPILL && uint32_t(sp) > stack_levels) top = spill[(uint32_t(sp) - 1u - stack_levels) & uint32_t(kSpillStack - 1)];
                if (uint32_t(sp) >= stack_levels) {
                    if (SPILL && uint32_t(sp) - stack_levels < uint32_t(kSpillStack)) spill[uint32_t(sp) - stack_levels] = farc;
                    else overflow |= uint32_t((h0 & h1) >> lane) & 1u;            // cannot happen (builder depth bound); counted
                }
            }
#if VHR_K1_POSTPONE
            // Postponed leaf (round 6): a lane whose next stop would be a LEAF parks it -- one slot per lane -- and goes on with what the stack holds
            // instead of sitting out the rest of the loop (10 % of the loop's lane-trips on sponza_proc, 15 % on bistro_proc, were lanes holding a
            // leaf: profiles/r6_k1_postponed_leaf.txt).  Both children hit: the far child is entered at once (it was just written above the top, sp stays);
            // one child hit: the old top is popped.  The parked leaf is tested in the leaf stage below; any hit = OR over the leaves a ray touches, in
            // any order.  Five vector instructions per trip, the masks on the scalar unit.
            const unsigned long long any_hit = h0 | h1;
            const unsigned long long park = cmp_gt_i32_mask_s(0, nearc) & any_hit & cmp_eq_i32_mask_s(kStackSentinel, parked);
            parked = select_mask(parked, nearc, park);
            cur = select_mask(select_mask(top, nearc, any_hit & ~park), farc, park & h0 & h1);
            {   // sp += h0 + h1 - 1 - park
                int t;
                unsigned long long carry_out;
                asm("v_addc_co_u32_e64 %0, %1, %2, -1, %3" : "=v"(t), "=s"(carry_out) : "v"(sp), "s"(h0));
                asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(t), "=s"(carry_out) : "v"(t), "s"(h1));
                asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(sp), "=s"(carry_out) : "v"(t), "s"(park));
            }
#else
            cur = select_mask(top, nearc, h0 | h1);                               // no child hit: popping the empty stack yields the sentinel
            {   // sp += h0 + h1 - 1: +1 both, 0 one, -1 none (with the sentinel) -- two add-with-carry, the hit masks as the carries
                int t;
                unsigned long long carry_out;
                asm("v_addc_co_u32_e64 %0, %1, %2, -1, %3" : "=v"(t), "=s"(carry_out) : "v"(sp), "s"(h0));
                asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(sp), "=s"(carry_out) : "v"(t), "s"(h1));
            }
#endif
        }
#ifdef VHR_K1_COUNT_IDLE
        // (scratch builds, profiles/r6_k1_postponed_leaf.txt) who sat out how many trips of this pass through the node loop: lanes holding a leaf -- what a
        // postponed-leaf step could have kept walking -- and lanes without a ray.  [drain_le4, drain_le8, drain_le16] = lane-trips of walkers, holders, the rest.
        if (stats) {
            const uint32_t mine = n_nodes - nodes_before;
            uint32_t tot = mine;
            for (int off = 32; off > 0; off >>= 1) tot = max(tot, uint32_t(__shfl_xor(int(tot), off)));
            const bool holder = has && cur < 0 && cur != kStackSentinel;
            uint32_t w = mine, hl = holder ? tot - mine : 0u, fr = holder ? 0u : tot - mine;
            for (int off = 32; off > 0; off >>= 1) { w += uint32_t(__shfl_xor(int(w), off)); hl += uint32_t(__shfl_xor(int(hl), off)); fr += uint32_t(__shfl_xor(int(fr), off)); }
            n_drain_le4 += w; n_drain_le8 += hl; n_drain_le16 += fr;
        }
#endif
        const unsigned long long t2 = stats ? __builtin_readcyclecounter() : 0ull;
        // ---- leaf ----
        // one memory round trip per triangle: its three loads are issued together and the test has no early return (with
        // ray_triangle() the compiler sinks the load of v0 behind the `det == 0` return: two dependent round trips per test).
        // (r3c-r5 the NEXT triangle's loads were in flight while this one was tested: -1.3 % on sponza_proc, +0.7 % on bistro_proc.  With the
        // consistency test of decision (vi) the triangle's nine floats live to the end of the test, and next to a second triangle's nine the
        // kernel needed 68 registers -- a wave per SIMD; without the prefetch 59.  Measured equal: 0.4558 / 0.4569 ms.  Verifying candidates
        // outside the loop on a triangle fetched again kept the prefetch at 62 registers and cost more: 0.469 ms -- a found ray is no rarity.)
        auto test_leaf = [&](const int code) {
            const uint32_t vv = ~uint32_t(code);
            const uint32_t first = vv >> 2, count = (vv & 3u) + 1u;
            ++n_leaves;
            const BvhTri *const leaf = a.scene.tris + first;
            for (uint32_t i = 0; i < count; ++i) {
                ++n_tris;
                const float4 ta = reinterpret_cast<const float4 *>(leaf + i)[0], tb = reinterpret_cast<const float4 *>(leaf + i)[1];
                const float tcx = reinterpret_cast<const float *>(leaf + i)[8];
                const f3 v0 = f3{ ta.x, ta.y, ta.z }, e1 = f3{ ta.w, tb.x, tb.y }, e2 = f3{ tb.z, tb.w, tcx };
                float ct, cu, cv;
                // (the consistency test behind the candidates only: few tests get this far, and a wave whose lanes all failed skips it)
                if (mt_candidate(ro, rd, v0, e1, e2, tmin, tmax, ct, cu, cv)) {
                    if (solution_consistent(ro, rd, v0, e1, e2, ct, cu, cv)) {
                        found = true;
                        break;
                    }
                    // decision (vi): a candidate that contradicts itself is decided again in binary64 -- not here: the pixel is marked and computed
                    // again when the tile is done (redo_pixel_visibility), the walk goes on as if the candidate had missed
                    atomicOr(&s_vis[pix], kRedoPixel);
                }
            }
        };
#if VHR_K1_POSTPONE
        if (has && parked != kStackSentinel) {                                      // the leaf the lane parked on its way
            test_leaf(parked);
            parked = kStackSentinel;
        }
        if (has && !found && cur < 0 && cur != kStackSentinel) {                    // ... and the one it stopped at
#else
        if (has && cur < 0 && cur != kStackSentinel) {
#endif
            test_leaf(cur);
            if (!found) {                                                          // pop (the sentinel if nothing is pending)
                cur = stack[min(uint32_t(sp), stack_levels + 1u) * kQueueBlock];
                if (SPILL && __any(uint32_t(sp) > stack_levels)) {
                    if (uint32_t(sp) > stack_levels) cur = spill[(uint32_t(sp) - 1u - stack_levels) & uint32_t(kSpillStack - 1)];
                }
                --sp;
            }
        }
        if (has && !found && cur == kStackSentinel && emask) {                 // overflowed cut entries: next subtree
            const int e = __ffs(int(emask)) - 1;
            emask &= emask - 1u;
            cur = __float_as_int(s_cut[e][1].z);
            sp = 0;                                   // the pop of the empty stack left it at -1
            sbase = 0;                                // (every row is this lane's own again)
        }
        const bool finished = found || cur == kStackSentinel;
        if (has && finished) {
            has = false;
            if (kind_bits) { if (found) atomicOr(&s_vis[pix], 1u << kind); }                  // miss.rmiss:7 leaves 1.0 where nothing is found
            else if (kind == 0) { if (found) atomicOr(&s_vis[pix], 1u); }
            else if (!found) atomicAdd(&s_vis[pix], 256u);
        }
        if (stats) {       // wave-level trip counts of the two inner loops = the slowest lane's (for lane utilisation)
            const unsigned long long t3 = __builtin_readcyclecounter();
            t_refill += t1 - t0; t_nodes += t2 - t1; t_leaves += t3 - t2;
            uint32_t tn = n_nodes - nodes_before, tt = n_tris - tris_before;
            for (int off = 32; off > 0; off >>= 1) { tn = max(tn, uint32_t(__shfl_xor(int(tn), off))); tt = max(tt, uint32_t(__shfl_xor(int(tt), off))); }
            n_wave_trips += tn + tt;
            if (next >= total) {
#ifndef VHR_K1_COUNT_TRIPS
                n_drain_trips += tn + tt;    // trips made after the tile's queue ran dry (nothing left to refill with)
#endif
                const uint32_t live = uint32_t(__popcll(__ballot(has)));   // rays still in flight after this round of trips
#ifndef VHR_K1_COUNT_IDLE
                if (live <= 4u) n_drain_le4 += tn + tt;
                if (live <= 8u) n_drain_le8 += tn + tt;
                if (live <= 16u) n_drain_le16 += tn + tt;
#endif
            }
        }
    }
    wave_lds_sync();
    float shadow_payload = 1.0f, ao_payload = 1.0f;                                          // rgen:20-21 for a pixel without geometry
    uint32_t n_redo = 0;
    if (covered) {
        const uint32_t vis = s_vis[lane];
        shadow_payload = (vis & 1u) ? 0.0f : 1.0f;
        if (a.tp.ao_spp) ao_payload = float(a.scene.node_count == 0 ? a.tp.ao_spp : (kind_bits ? a.tp.ao_spp - uint32_t(__popc(vis >> 1)) : (vis >> 8))) / float(a.tp.ao_spp);   // rgen:55: the AO rays that escaped
        if (vis & kRedoPixel) {                           // decision (vi): one of this pixel's rays met a candidate that contradicts itself
            static_assert(offsetof(RaygenArgs, scene) == 0, "the launch's arguments start with `a`");
            const float2 again = redo_pixel_visibility(reinterpret_cast<const RaygenArgs *>((const void *)__builtin_amdgcn_kernarg_segment_ptr()), x, y);
            shadow_payload = again.x; ao_payload = again.y;
            n_redo = 1;
        }
        store_rg16f(a.shadow_ao, W, x, y, shadow_payload, ao_payload);                       // rgen:57
    }
    if (a.co.wave_cost && lane == 0) a.co.wave_cost[block_tile * uint32_t(WAVES) + wave] = uint32_t(min(__builtin_readcyclecounter() - t_cost0, 0xffffffffull));
    // ---- "fuse_temporal": svgf.comp for this tile's pixels, right here ----
    // svgf.comp reads of the CURRENT frame only the pixel's own texels; everything else it gathers is the previous frame's.  So the
    // wave that has just finished a tile can run it for the tile: the visibility goes from LDS into the filter (rounded to the halves
    // the image holds, which is also what is stored), the normals are the ones the set-up loaded, and the gathers of 16 200 x 2 waves
    // spread over the launch instead of forming a kernel of their own that waits on memory.
    if constexpr (FUSE) {                             // (an instantiation of its own: the epilogue's registers and code stay out of the plain launch)
        if (a.fuse_temporal) {                                                               // (uniform)
            const TemporalArgs &t = a.temporal;
            if (in_range && x >= t.col_begin && x < t.limit_x && y >= t.row_begin && y < t.row_end && y < t.limit_y) {
                const uint2 nraw = reinterpret_cast<const uint2 *>(a.normals)[size_t(y) * W + x];
                const float2 cur = unpack_rg16f(pack_rg16f(shadow_payload, ao_payload));     // what the RG16F image holds
                svgf_temporal_pixel(t, x, y, unpack_rgba16f(nraw), cur.x, cur.y);
            }
        }
    }
    if (stats) {
        const unsigned long long ovf = __ballot(overflow != 0);
        if (lane == 0) {
            if (cov_mask) atomicAdd(&stats->covered_pixels, (unsigned long long)__popcll(cov_mask));
            if (ovf) atomicAdd(&stats->stack_overflows, (unsigned long long)__popcll(ovf));
            atomicAdd(&stats->wave_iterations, (unsigned long long)n_wave_trips);
            atomicAdd(&stats->drain_iterations, (unsigned long long)n_drain_trips);
            atomicAdd(&stats->drain_le4, (unsigned long long)n_drain_le4);
            atomicAdd(&stats->drain_le8, (unsigned long long)n_drain_le8);
            atomicAdd(&stats->drain_le16, (unsigned long long)n_drain_le16);
            atomicAdd(&stats->cycles_total, __builtin_readcyclecounter() - t_start);
            atomicAdd(&stats->cycles_setup, t_setup);
            atomicAdd(&stats->cycles_refill, t_refill);
            atomicAdd(&stats->cycles_nodes, t_nodes);
            atomicAdd(&stats->cycles_leaves, t_leaves);
            atomicAdd(&stats->refills, n_refills);
            atomicAdd(&stats->waves, 1ull);
            atomicAdd(&stats->cut_entries, (unsigned long long)n_cut_entries);
        }
        const unsigned long long redo = __ballot(n_redo != 0);
        if (redo && lane == 0) atomicAdd(&stats->pending_rays, (unsigned long long)__popcll(redo));
        // wave-reduced first: 64 same-address atomics per wave serialise at the memory side (the diagnostic launch took 3.4 ms)
        for (int off = 32; off > 0; off >>= 1) { n_nodes += uint32_t(__shfl_xor(int(n_nodes), off)); n_leaves += uint32_t(__shfl_xor(int(n_leaves), off)); n_tris += uint32_t(__shfl_xor(int(n_tris), off)); }
        if (lane == 0) {
            atomicAdd(&stats->node_visits, (unsigned long long)n_nodes);
            atomicAdd(&stats->leaf_visits, (unsigned long long)n_leaves);
            atomicAdd(&stats->triangle_tests, (unsigned long long)n_tris);
        }
    }
}

// Mirror ray of raygen.rgen:59-65 (closest hit, reflection_hit.rchit / reflection_miss.rmiss) for one pixel of the image
template <int STRIDE>
__device__ __forceinline__ f4 reflection_payload(const RaygenArgs &a, const uint32_t x, const uint32_t y, int *stack, bool &second_ray) {
    const uint32_t W = a.width, H = a.height;
    f4 payload = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const float depth = a.depth[size_t(y) * W + x];
    if (depth != 0.0f && a.tp.reflections) {
        const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
        const f3 P = get_world_space_position(a.pfd, depth, u, v);
        const f4 nid = load_rgba16f(a.normals, W, x, y);
        const f3 N = f3{ nid.x, nid.y, nid.z };
        const f3 origin = P + N * a.tp.normal_bias;
        const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };
        const f3 I = normalize3(P - cam);
        const float ni2 = 2.0f * dot3(N, I);
        const f3 rdir = I - N * ni2;
        uint32_t overflow = 0;
        payload = trace_reflection<STRIDE>(a.scene, a.pfd, a.tp, origin, rdir, stack, overflow, second_ray);
    }
    return payload;
}
__device__ __forceinline__ void reflection_pixel(const RaygenArgs &a, const uint32_t x, const uint32_t y, int *stack, bool &second_ray) {
    const f4 payload = reflection_payload<kTraceBlock>(a, x, y, stack, second_ray);
    store_rgba16f(a.reflections, a.width, x, y, payload.x, payload.y, payload.z, payload.w);
}

// Decision (vi) in the mirror ray's queue kernel: a pixel whose ray (first or second bounce) met a candidate whose fp32 solution contradicts itself is
// computed again, whole, by the per-pixel kernel's code (binary64 decisions inline) when its tile is shaded -- a call, so that the queue's walk carries
// none of this: inlined in its leaf test the binary64 arithmetic made the two-bounce kernel spill 45 registers (launch +13 %), a list of the candidates
// per ray decided by a call at the ray's commit still cost +7 % (profiles/r6_decision_vi_cost.txt).  `a` points at the launch's arguments where they lie
// in memory (the address of a by-value argument would copy all of it to every lane's scratch: 1.4 KB a lane, and the launch took four times as long for
// the waves the scratch ring then had room for).  About one pixel of a 1080p frame on the BASELINE stand-ins, ~70-100 on sponza_hard_rot.  `second`: a second-bounce ray was
// traced (the launch's ray count).
struct RedoReflection { f4 payload; uint32_t second; };
__device__ __attribute__((noinline)) RedoReflection redo_pixel_reflection(const RaygenArgs *a, const uint32_t x, const uint32_t y) {
    int st[kTraceStack];
    bool second_ray = false;
    const f4 payload = reflection_payload<1>(*a, x, y, st, second_ray);
    return RedoReflection{ payload, second_ray ? 1u : 0u };
}

// ... one pixel per thread
__global__ __launch_bounds__(kTraceBlock) void reflection_kernel(const RaygenArgs a, const Stamps st) {
    vhr_stamp(st);
    __shared__ int s_refl_stack[kTraceStack * kTraceBlock];
    int *stack = s_refl_stack + threadIdx.x;
    uint32_t x, y;
    pixel_of_thread(x, y, a.row_begin);
    bool second_ray = false;
    if (x < a.width && y < a.row_end) reflection_pixel(a, x, y, stack, second_ray);
    if (a.stats) {
        const unsigned long long sec = __ballot(second_ray);
        if ((threadIdx.x & 63u) == 0 && sec) atomicAdd(&a.stats->second_bounce_rays, (unsigned long long)__popcll(sec));
    }
}

// ---------------------------------------------------------------------------------------------
// One pass of a wave over its ray queue (`total` rays; fetch(r, pix, origin, direction) delivers the r-th one and the id of
// its pixel, commit(pix, triangle, u, v) takes its result, kNoHit = miss).  Lanes pull rays whenever `refill_threshold` of
// them are idle and walk the BVH "while-while" with the node step of raygen_queue_kernel: packed-FMA slabs against 1/d and
// -o/d, near child first, far child pushed, boxes culled against the closest t so far (tn <= tbest keeps equal-t candidates:
// decision vi), early exit of the node loop, LDS stack + scratch spill.  Leaves: every triangle, Moeller-Trumbore against the
// full [tmin, tmax] interval, closest = min t then smaller flat index; with `any_hit` (wave-uniform) the first accepted
// triangle ends the ray (gl_RayFlagsTerminateOnFirstHitEXT -- the boolean does not depend on the order).  ALPHA: every
// candidate first runs shadow_anyhit.rahit (alpha_ignored).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kNoHit = 0xffffffffu;

// what a walk did (STATS builds only): node visits, leaf visits and triangle tests summed over lanes, and the trips of the two inner loops
// counted once per wave (the slowest lane's) -- lane utilisation = (nodes + triangles) / (64 x wave_trips), as for raygen_queue_kernel
struct WalkCounters { uint32_t nodes = 0, leaves = 0, triangles = 0, wave_trips = 0, refills = 0; };

struct NoFlag { __device__ __forceinline__ void operator()(uint32_t) const {} };
template <bool SPILL, bool ALPHA, bool DEFER, bool STATS = false, typename Fetch, typename Commit, typename Flag = NoFlag>
__device__ __forceinline__ void wave_queue_walk(const DeviceScene &sc, int *stack, const uint32_t stack_levels, const uint32_t lane,
                                                const uint32_t total, const uint32_t refill_threshold, const uint32_t early_exit,
                                                const float tmin, const float tmax, const bool any_hit, uint32_t &overflow,
                                                const float4 (*cut)[2], const uint32_t cut_n, Fetch fetch, Commit commit, WalkCounters *wc = nullptr,
                                                Flag flag = Flag{}) {
    f3 ro = f3{ 0, 0, 0 }, rd = f3{ 0, 0, 1 }, rinv = f3{ 0, 0, 0 }, noi = f3{ 0, 0, 0 }, ainv = f3{ 0, 0, 0 };
    float tbest = 0.0f, best_u = 0.0f, best_v = 0.0f;
    uint32_t best_tri = kNoHit, best_flat = 0;
    int cur = 0, sp = 0;
    uint32_t pix = 0, next = 0;
    uint32_t emask = 0;                                   // cut entries this lane's ray hits that did not fit its LDS stack
    bool has = false;
    // volatile: keeps the array in scratch.  Left alone, the compiler promotes it to 32 VGPRs with indirect indexing, which
    // pushes the kernels over their register budget (55 spilled VGPRs, 38 spilled SGPRs, 1.5x slower: measured)
    volatile int spill[SPILL ? kSpillStack : 1];
    float tmin_v = tmin;
    asm volatile("" : "+v"(tmin_v));
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = uint32_t(__popcll(idle));
        if (next < total && (n_idle >= refill_threshold || n_idle == 64u)) {                 // wave-uniform
            const uint32_t r = next + lane_rank(idle);
            next += n_idle;
            if (STATS && lane == 0) ++wc->refills;
            if (!has && r < total) {
                fetch(r, pix, ro, rd);
                f3 bo, bd;
                box_ray(sc, ro, rd, bo, bd);                  // "bvh_frame": the slab tests' ray (ro, rd stay the triangle tests')
                rinv = f3{ cull_reciprocal(bd.x), cull_reciprocal(bd.y), cull_reciprocal(bd.z) };
                noi = f3{ -(bo.x * rinv.x), -(bo.y * rinv.y), -(bo.z * rinv.z) };
                ainv = f3{ fabsf(rinv.x), fabsf(rinv.y), fabsf(rinv.z) };
                tbest = tmax; best_tri = kNoHit; best_flat = 0; best_u = 0.0f; best_v = 0.0f;
                cur = 0; sp = 0;
                // the ray against the tile's cut: the (t, flat index) order of the commit makes the result independent of the order the subtrees are walked in
                if (cut_n) cut_to_stack(cut, cut_n, stack, stack_levels, rinv, noi, tmin_v, tmax, cur, sp, emask);
                has = true;
            }
        }
        if (!__any(has)) break;
        // ---- inner nodes ----
        const uint32_t walkers_in = uint32_t(__popcll(__ballot(has && cur >= 0)));
        uint32_t my_nodes = 0, my_tris = 0;               // (STATS) this lane's trips of the two inner loops in this round
        while (has && cur >= 0) {
            if (uint32_t(__popcll(__ballot(true))) * 16u <= walkers_in * early_exit) break;
            if (STATS) ++my_nodes;
            // (the 48-byte fp32 nodes: the 32-byte half-precision ones were measured here too -- r3c, and r4 with the walk as a kernel of its
            // own at 58 registers -- and make no difference to this walk)
            const Node48Words nw = load_node48(sc.nodes48, cur);
            const int2 links = nw.links;
            float tn0, tn1;
            bool h0, h1;
            box_pair_ch(nw.q0, nw.q1, nw.q2, rinv, ainv, noi, tmin_v, tbest, h0, h1, tn0, tn1);
            const bool both = h0 && h1, none = !(h0 || h1);
            const bool first0 = tn0 <= tn1;
            const int nearc = first0 ? links.x : links.y, farc = first0 ? links.y : links.x;
            int *const row = stack + min(uint32_t(sp), stack_levels + 1u) * kQueueBlock;
            int top = row[0];
            row[kQueueBlock] = farc;
            if (__any(uint32_t(sp) >= stack_levels)) {
                if (SPILL && uint32_t(sp) > stack_levels) top = spill[(uint32_t(sp) - 1u - stack_levels) & uint32_t(kSpillStack - 1)];
                if (uint32_t(sp) >= stack_levels) {
                    if (SPILL && uint32_t(sp) - stack_levels < uint32_t(kSpillStack)) spill[uint32_t(sp) - stack_levels] = farc;
                    else overflow |= both ? 1u : 0u;
                }
            }
            cur = both ? nearc : (none ? top : (h0 ? links.x : links.y));
            sp += (both ? 1 : 0) - (none ? 1 : 0);
        }
        // ---- leaf ----
        if (has && cur < 0 && cur != kStackSentinel) {
            const uint32_t vv = ~uint32_t(cur);
            const uint32_t first = vv >> 2, count = (vv & 3u) + 1u;
            bool done = false;
            if (STATS) ++wc->leaves;
            for (uint32_t i = 0; i < count; ++i) {
                const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + first + i);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                float t, uu, ww;
                if (STATS) ++my_tris;
                const f3 v0 = f3{ ta.x, ta.y, ta.z }, e1 = f3{ ta.w, tb.x, tb.y }, e2 = f3{ tb.z, tb.w, tc.x };
                if (mt_candidate(ro, rd, v0, e1, e2, tmin, tmax, t, uu, ww)) {
                    // decision (vi): a candidate that contradicts itself is decided again in binary64.  DEFER (the mirror ray's kernels, where about one ray of a
                    // 1080p frame has one): not here, where the walk's registers are all alive -- the ray's pixel is flagged and computed again by the per-pixel
                    // code when the tile is shaded (redo_pixel_reflection); the walk goes on as if the candidate had missed.  !DEFER (the raytraced path,
                    // whose shadow rays leave the hit point itself: 6 % of its rays have one): inline.
                    if (!solution_consistent(ro, rd, v0, e1, e2, t, uu, ww)) {
                        if (DEFER) { flag(pix); continue; }
                        if (!mt_binary64(ro, rd, v0, e1, e2, tmin, tmax, t, uu, ww)) continue;
                    }
                    if (ALPHA && alpha_ignored(sc, first + i, uu, ww)) continue;
                    const uint32_t flat = __float_as_uint(tc.w);
                    if (best_tri == kNoHit || t < tbest || (t == tbest && flat < best_flat)) {
                        tbest = t; best_tri = first + i; best_flat = flat; best_u = uu; best_v = ww;
                    }
                    if (any_hit) { done = true; break; }
                }
            }
            if (done) {
                cur = kStackSentinel;
                emask = 0;
            } else {
                cur = stack[min(uint32_t(sp), stack_levels + 1u) * kQueueBlock];             // pop (the sentinel if nothing is pending)
                if (SPILL && __any(uint32_t(sp) > stack_levels)) {
                    if (uint32_t(sp) > stack_levels) cur = spill[(uint32_t(sp) - 1u - stack_levels) & uint32_t(kSpillStack - 1)];
                }
                --sp;
            }
        }
        if (has && cur == kStackSentinel && emask) {          // overflowed cut entries: the next subtree
            const int e = __ffs(int(emask)) - 1;
            emask &= emask - 1u;
            cur = __float_as_int(cut[e][1].z);
            sp = 0;                                           // (the pop of the empty stack left it at -1)
        }
        if (has && cur == kStackSentinel) {
            has = false;
            commit(pix, best_tri, best_u, best_v);
        }
        if (STATS) {
            wc->nodes += my_nodes; wc->triangles += my_tris;
            uint32_t tn = my_nodes, tt = my_tris;
            for (int off = 32; off > 0; off >>= 1) { tn = max(tn, uint32_t(__shfl_xor(int(tn), off))); tt = max(tt, uint32_t(__shfl_xor(int(tt), off))); }
            wc->wave_trips += tn + tt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Mirror ray, work-queue form (the default for one bounce): closest-hit traversal with the node step of raygen_queue_kernel.
//
// Every wave owns a 16x8-pixel tile = a queue of up to 128 mirror rays.  Phase 1 (whole wave, twice): raygen.rgen:15-29,60-63
// per pixel -- origin and reflect(I, N) parked in LDS, covered pixels compacted with a ballot.  Phase 2: lanes pull rays from
// the queue whenever `refill_threshold` of them are idle and walk the BVH "while-while": the branch-free inner-node step
// (packed-FMA slabs against 1/d and -o/d, near child first, far child pushed, boxes culled against the closest t so far),
// then the leaf's <= 4 Moeller-Trumbore tests with the (t, flat index) order of decision (vi).  A finished ray leaves its
// hit record (triangle, u, v) in the LDS slot its origin came from.  Phase 3 (whole wave, twice): reflection_hit.rchit on
// the records -- the texture fetches and the BRDF run with all lanes active instead of inside the divergent walk.
// Results are those of reflection_kernel bit for bit: the same rays, the same intersection arithmetic, the same shader.
// ---------------------------------------------------------------------------------------------
constexpr int kReflRays = 128;

// raygen.rgen:15-16, 26-29, 60-61 for one covered pixel: the mirror ray's origin and direction
__device__ __forceinline__ void mirror_ray_of_pixel(const RaygenArgs &a, f3 cam, uint32_t x, uint32_t y, float depth, f3 &origin, f3 &rdir) {
    const uint32_t W = a.width, H = a.height;
    const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);          // rgen:15-16
    const f3 P = get_world_space_position(a.pfd, depth, u, v);                               // rgen:26
    const f4 nid = load_rgba16f(a.normals, W, x, y);                                         // rgen:28
    const f3 N = f3{ nid.x, nid.y, nid.z };
    origin = P + N * a.tp.normal_bias;                                                       // rgen:29
    const f3 I = normalize3(P - cam);                                                        // rgen:60
    const float ni2 = 2.0f * dot3(N, I);
    rdir = I - N * ni2;                                                                      // rgen:61 reflect(I, N)
}

template <bool SPILL, int BOUNCES, bool STATS = false>
__global__ __launch_bounds__(kQueueBlock * 2) __attribute__((amdgpu_waves_per_eu(7, 7))) void reflection_queue_kernel(
    const RaygenArgs a, const uint32_t stack_levels, const uint32_t refill_threshold, const uint32_t tiles_x, const uint32_t tiles_total,
    const uint32_t early_exit, const Stamps st) {
    vhr_stamp(st);
    extern __shared__ int s_dyn[];                        // per wave: (stack_levels + 3) x 64 ints, see raygen_queue_kernel
    // rows 0-2 origin -> hit record (triangle, u, v), rows 3-5 direction; two bounces: rows 6-8 second origin -> second record,
    // and rows 3-5 are rewritten with the second direction between the two walks.
    // A wave's tile is 8 x 8 pixels, one ray per lane (r5; 16 x 8 before).  The launch waits on the latency of its walks -- 4 KB of LDS padding per
    // workgroup, 5.5 -> 4.5 waves per SIMD, cost it 16-23 % -- and its LDS, not its registers, set the occupancy: with 128 rays per wave 14 080 B
    // per workgroup of two waves (11 workgroups per CU = 5.5 waves per SIMD; two bounces 17 152 B = 4.5 waves).  With 64 rays 10 880 / 12 416 B:
    // 7 / 6.5 waves (69-72 registers admit 7).  One bounce 261-265 -> 244-250 us (sponza_proc 1080p), 444-450 -> 435 us (bistro_proc); two bounces
    // 626 -> 508 us, 963 -> 854 us.  (Recomputing a ray from its pixel at the queue's fetch instead of parking it in LDS -- 9 344 B -- was equal
    // or slower than parking it: the ~120 instructions per ray eat what the last half wave per SIMD buys; 8 waves at 64 registers bought nothing.)
    constexpr int ROWS = BOUNCES > 1 ? 9 : 6;
    constexpr uint32_t SUBS = 1u;                         // 8-pixel-wide sub-tiles per wave (the loops below are written for any number)
    constexpr int RAYS = 64 * int(SUBS);
    __shared__ float s_ray_all[2][ROWS][RAYS];
    __shared__ uint8_t s_list_all[2][RAYS];               // compacted covered pixels
    __shared__ float4 s_cut_all[2][kCutMax][2];           // the tile's shared descent (build_tile_cut), once per walk
    __shared__ uint32_t s_redo_all[2][RAYS / 32];         // decision (vi): bit p = pixel p is computed again in phase 3 (redo_pixel_reflection)
    const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    // "raygen_cost_order" for this launch (see raygen_queue_kernel): the first block sorts the previous launch's blocks before its own tiles
    const unsigned long long t_cost0 = a.co.wave_cost ? __builtin_readcyclecounter() : 0ull;
    if (a.co.order_out && blockIdx.x == 0u) order_blocks_by_cost<2>(a.co.cost_prev, a.co.order_blocks, a.co.order_out, reinterpret_cast<uint32_t *>(s_dyn));
    const uint32_t block = a.co.block_order ? a.co.block_order[blockIdx.x] : blockIdx.x;
    const uint32_t tile = block * 2u + wave;
    if (tile >= tiles_total) return;                      // waves of a block share nothing and never synchronise
    float (&s_ray)[ROWS][RAYS] = s_ray_all[wave];
    uint8_t (&s_list)[RAYS] = s_list_all[wave];
    uint32_t (&s_redo)[RAYS / 32] = s_redo_all[wave];
    if (lane < uint32_t(RAYS / 32)) s_redo[lane] = 0u;
    int *stack = s_dyn + wave * (stack_levels + 3u) * kQueueBlock + lane;
    stack[0] = kStackSentinel;
    const uint32_t W = a.width;
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };

    // ---- phase 1: per-pixel ray setup, whole wave ----
    unsigned long long covered_mask[SUBS];
    uint32_t ncov = 0;
    f3 omin = f3{ 3.0e38f, 3.0e38f, 3.0e38f }, omax = f3{ -3.0e38f, -3.0e38f, -3.0e38f };   // bounds of this walk's ray origins
    auto grow = [&](f3 o) {
        omin = f3{ fminf(omin.x, o.x), fminf(omin.y, o.y), fminf(omin.z, o.z) };
        omax = f3{ fmaxf(omax.x, o.x), fmaxf(omax.y, o.y), fmaxf(omax.z, o.z) };
    };
#pragma unroll
    for (uint32_t sub = 0; sub < SUBS; ++sub) {
        const uint32_t x = a.col_begin + tile_x * (8u * SUBS) + sub * 8u + (lane & 7u), y = a.row_begin + tile_y * 8u + (lane >> 3);
        const bool in_range = x < a.col_end && y < a.row_end;
        const float depth = in_range ? a.depth[size_t(y) * W + x] : 0.0f;                    // rgen:19
        const bool covered = depth != 0.0f;
        if (in_range && !covered) store_rgba16f(a.reflections, W, x, y, 0.0f, 0.0f, 0.0f, 0.0f);   // rgen:22
        const uint32_t p = sub * 64u + lane;
        if (covered) {
            f3 origin, rdir;
            mirror_ray_of_pixel(a, cam, x, y, depth, origin, rdir);
            s_ray[0][p] = origin.x; s_ray[1][p] = origin.y; s_ray[2][p] = origin.z;
            s_ray[3][p] = rdir.x; s_ray[4][p] = rdir.y; s_ray[5][p] = rdir.z;
            grow(origin);
        }
        const unsigned long long m = __ballot(covered);
        covered_mask[sub] = m;
        if (covered) s_list[ncov + lane_rank(m)] = uint8_t(p);
        ncov += uint32_t(__popcll(m));
    }
    wave_lds_sync();
    uint32_t total = a.scene.node_count == 0 ? 0u : ncov;
    const bool traced = total != 0;

    // ---- phase 2: the queue (once per bounce) ----
    uint32_t overflow = 0, second_rays = 0;
    WalkCounters wc;
    const unsigned long long t_walk0 = STATS ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_walk = 0;
#pragma unroll 1
    for (int bounce = 0; bounce < BOUNCES; ++bounce) {
    const int orow = bounce ? 6 : 0;                      // where this bounce's origins sit and its hit records go
    // (first bounce only: the second bounce's origins are scattered over the scene, their descent ends at once -- measured: no gain)
    const uint32_t cut_n = total && bounce == 0 ? build_tile_cut_uniform(a.scene, omin, omax, s_cut_all[wave], lane) : 0u;
    const unsigned long long tw0 = STATS ? __builtin_readcyclecounter() : 0ull;
    wave_queue_walk<SPILL, false, true, STATS>(
        a.scene, stack, stack_levels, lane, total, refill_threshold, early_exit, a.tp.tmin, a.tp.tmax, false, overflow, s_cut_all[wave], cut_n,
        [&](uint32_t r, uint32_t &pix, f3 &ro, f3 &rd) {
            pix = s_list[r];
            ro = f3{ s_ray[orow][pix], s_ray[orow + 1][pix], s_ray[orow + 2][pix] };
            rd = f3{ s_ray[3][pix], s_ray[4][pix], s_ray[5][pix] };
        },
        [&](uint32_t pix, uint32_t tri, float u, float v) {                                  // the hit record replaces the ray's origin
            s_ray[orow][pix] = __uint_as_float(tri); s_ray[orow + 1][pix] = u; s_ray[orow + 2][pix] = v;
        }, &wc,
        [&](uint32_t pix) { atomicOr(&s_redo[pix >> 5], 1u << (pix & 31u)); });                   // decision (vi): the pixel is computed again in phase 3
    wave_lds_sync();
    if (STATS) t_walk += __builtin_readcyclecounter() - tw0;
    if constexpr (BOUNCES > 1) if (bounce == 0) {
        // ---- second-bounce rays (trace_reflection's arithmetic), whole wave: a mirror ray from every first hit ----
        uint32_t n2 = 0;
#pragma unroll
        for (uint32_t sub = 0; sub < SUBS; ++sub) {
            const uint32_t p = sub * 64u + lane;
            const bool was_covered = traced && ((covered_mask[sub] >> lane) & 1ull);
            const uint32_t tri = was_covered ? __float_as_uint(s_ray[0][p]) : kNoHit;
            const bool hit1 = tri != kNoHit;
            if (hit1) {
                Hit h;
                h.t = 0.0f; h.u = s_ray[1][p]; h.v = s_ray[2][p]; h.tri_index = tri; h.flat = 0;
                f3 hp, hn;
                hit_position_normal(a.scene, h, hp, hn);
                const f3 rdir = f3{ s_ray[3][p], s_ray[4][p], s_ray[5][p] };
                const f3 nn = normalize3(hn);
                const float ni = dot3(nn, rdir);
                const f3 nf = ni < 0.0f ? nn : -nn;
                const f3 d2 = rdir - nn * (2.0f * ni);
                const f3 o2 = hp + nf * a.tp.normal_bias;
                s_ray[6][p] = o2.x; s_ray[7][p] = o2.y; s_ray[8][p] = o2.z;
                s_ray[3][p] = d2.x; s_ray[4][p] = d2.y; s_ray[5][p] = d2.z;
            }
            const unsigned long long m = __ballot(hit1);
            if (hit1) s_list[n2 + lane_rank(m)] = uint8_t(p);
            n2 += uint32_t(__popcll(m));
        }
        wave_lds_sync();
        total = n2;
        second_rays = n2;
    }
    }

    // ---- phase 3: reflection_hit.rchit / reflection_miss.rmiss on the records, whole wave ----
    uint32_t n_redo = 0;
    int redo_second = 0;
#pragma unroll
    for (uint32_t sub = 0; sub < SUBS; ++sub) {
        if (!((covered_mask[sub] >> lane) & 1ull)) continue;
        const uint32_t x = a.col_begin + tile_x * (8u * SUBS) + sub * 8u + (lane & 7u), y = a.row_begin + tile_y * 8u + (lane >> 3);
        const uint32_t p = sub * 64u + lane;
        f4 payload = f4{ 0.0f, 0.0f, 0.0f, 0.0f };                                           // reflection_miss.rmiss:7
        const uint32_t tri = traced ? __float_as_uint(s_ray[0][p]) : kNoHit;
        if (traced && ((s_redo[p >> 5] >> (p & 31u)) & 1u)) {                                // decision (vi): this pixel's ray asked for binary64
            static_assert(offsetof(RaygenArgs, scene) == 0, "the launch's arguments start with `a`");
            const RedoReflection again = redo_pixel_reflection(reinterpret_cast<const RaygenArgs *>((const void *)__builtin_amdgcn_kernarg_segment_ptr()), x, y);
            payload = again.payload;
            ++n_redo;
            if (BOUNCES > 1) redo_second += int(again.second) - int(tri != kNoHit);          // (the launch's count of second-bounce rays)
        } else if (tri != kNoHit) {
            Hit h;
            h.t = 0.0f; h.u = s_ray[1][p]; h.v = s_ray[2][p]; h.tri_index = tri; h.flat = 0;
            if constexpr (BOUNCES > 1) {
                f4 second = f4{ 0.0f, 0.0f, 0.0f, 0.0f };                                    // reflection_miss.rmiss:7
                const uint32_t tri2 = __float_as_uint(s_ray[6][p]);
                if (tri2 != kNoHit) {
                    Hit h2;
                    h2.t = 0.0f; h2.u = s_ray[7][p]; h2.v = s_ray[8][p]; h2.tri_index = tri2; h2.flat = 0;
                    second = shade_reflection_hit(a.scene, a.pfd, h2);
                }
                payload = shade_reflection_hit(a.scene, a.pfd, h, &second);
            } else {
                payload = shade_reflection_hit(a.scene, a.pfd, h);
            }
        }
        store_rgba16f(a.reflections, W, x, y, payload.x, payload.y, payload.z, payload.w);   // rgen:65
    }
    if (a.stats) {
        for (int off = 32; off > 0; off >>= 1) { n_redo += uint32_t(__shfl_xor(int(n_redo), off)); redo_second += __shfl_xor(redo_second, off); }
        second_rays = uint32_t(int(second_rays) + redo_second);
        if (lane == 0) {
            if (overflow) atomicAdd(&a.stats->stack_overflows, 1ull);
            if (second_rays) atomicAdd(&a.stats->second_bounce_rays, (unsigned long long)second_rays);
            if (n_redo) atomicAdd(&(a.stats + 1)->pending_rays, (unsigned long long)n_redo);
        }
    }
    if constexpr (STATS) {
        // the mirror-ray launch's own counters: the second RayStats of the buffer (vhr_get_reflection_statistics)
        RayStats *const rs = a.stats + 1;
        uint32_t n_nodes = wc.nodes, n_leaves = wc.leaves, n_tris = wc.triangles;
        for (int off = 32; off > 0; off >>= 1) { n_nodes += uint32_t(__shfl_xor(int(n_nodes), off)); n_leaves += uint32_t(__shfl_xor(int(n_leaves), off)); n_tris += uint32_t(__shfl_xor(int(n_tris), off)); }
        if (lane == 0) {
            atomicAdd(&rs->node_visits, (unsigned long long)n_nodes);
            atomicAdd(&rs->leaf_visits, (unsigned long long)n_leaves);
            atomicAdd(&rs->triangle_tests, (unsigned long long)n_tris);
            atomicAdd(&rs->wave_iterations, (unsigned long long)wc.wave_trips);
            atomicAdd(&rs->unique_rays, (unsigned long long)(ncov + second_rays));
            atomicAdd(&rs->covered_pixels, (unsigned long long)ncov);
            atomicAdd(&rs->second_bounce_rays, (unsigned long long)second_rays);
            atomicAdd(&rs->refills, (unsigned long long)wc.refills);
            atomicAdd(&rs->waves, 1ull);
            atomicAdd(&rs->cycles_total, __builtin_readcyclecounter() - t_walk0);          // set-up + walks + shading, this wave
            atomicAdd(&rs->cycles_nodes, t_walk);                                           // the walks alone (both bounces)
        }
    }
    if (a.co.wave_cost && lane == 0) a.co.wave_cost[tile] = uint32_t(min(__builtin_readcyclecounter() - t_cost0, 0xffffffffull));
}

// The shadow / AO launch itself, by the options in force (everything launch_raygen decided is in `a`).
// "raygen_cost_order": the cost / order pointers of a queue-kernel launch of `n_blocks` blocks of `wv` waves (see vhr_context::CostOrder).
// 1 (default) = launches of at least 2 048 blocks (a full round of waves or more), 2 = any launch (tests); the two launches an order connects must
// have been issued on the same stream -- the order is written and read in stream order, nothing else guards it.
static void prepare_cost_order(vhr_context *ctx, vhr_context::CostOrder &co, const uint32_t n_blocks, const uint32_t wv, const uint32_t key, CostOrderArgs &out,
                               const vhr_context::CostOrder::Shape &shape) {
    const int mode = ctx->options[kOptRaygenCostOrder];
    if (!mode) return;
    if (co.stream != ctx->stream) {
        // another stream than the last launch's (frames in flight switched on or off, say): whatever of that stream is still in flight may be
        // writing an order -- wait once, forget both
        if (co.capacity) (void)hipDeviceSynchronize();
        co.stream = ctx->stream;
        co.order_blocks[0] = co.order_blocks[1] = co.cost_blocks[0] = co.cost_blocks[1] = 0;
    }
    if (n_blocks < (mode >= 2 ? 2u : 2048u)) return;
    const uint32_t n_waves = n_blocks * wv;
    if (n_waves > co.capacity) {
        (void)hipDeviceSynchronize();              // (first use / a larger launch: nothing may still read the old buffers)
        for (int i = 0; i < 2; ++i) { (void)hipFree(co.cost[i]); (void)hipFree(co.order[i]); co.cost[i] = co.order[i] = nullptr; }
        co.capacity = 0;
        co.order_blocks[0] = co.order_blocks[1] = co.cost_blocks[0] = co.cost_blocks[1] = 0;
        bool ok = true;
        for (int i = 0; i < 2; ++i)
            ok = ok && hipMalloc(reinterpret_cast<void **>(&co.cost[i]), size_t(n_waves) * 4) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&co.order[i]), size_t(n_waves) * 4) == hipSuccess;
        if (!ok) return;
        co.capacity = n_waves;
    }
    const uint32_t prev = co.slot, slot = prev ^ 1u;
    co.slot = slot;
    out.wave_cost = co.cost[slot];
    if (co.order_blocks[slot] == n_blocks && co.order_key[slot] == key) out.block_order = co.order[slot];
    if (co.cost_blocks[prev] == n_blocks && co.cost_key[prev] == key) {       // the previous launch had this shape: its blocks get ordered
        out.cost_prev = co.cost[prev]; out.order_out = co.order[prev]; out.order_blocks = n_blocks;
        co.order_blocks[prev] = n_blocks; co.order_key[prev] = key;
    } else {
        co.order_blocks[prev] = 0;
    }
    co.cost_blocks[slot] = n_blocks; co.cost_key[slot] = key;
    co.cost_waves[slot] = n_waves;
    co.shape[slot] = shape;
}

static void issue_raygen(vhr_context *ctx, const RaygenArgs &a_in, const uint32_t width, const uint32_t height) {
    RaygenArgs a = a_in;
    a.co = CostOrderArgs{};
    (void)height;
    ctx->time_begin(kKernelRaygen);
    if (ctx->options[kOptRaygenVariant] == 0) {
        launch(ctx, raygen_kernel, dim3((width + 15) / 16, (a.row_end - a.row_begin + 15) / 16), dim3(kTraceBlock), 0, a);
    } else {
        // LDS part of the traversal stack, sized by the tree actually built (depth <= kMaxBvhDepth): less LDS, more waves per CU; deeper
        // entries spill to scratch (see the kernel)
        const uint32_t levels = std::max<uint32_t>(1u, std::min<uint32_t>(ctx->bvh_depth + 1u, uint32_t(std::max(1, ctx->options[kOptLdsStackLevels]))));
        const uint32_t threshold = uint32_t(std::max(1, std::min(64, ctx->options[kOptRefillThreshold])));
        const size_t stack_bytes = size_t(levels + 3) * kQueueBlock * sizeof(int);
        const uint32_t rows_traced = a.row_end - a.row_begin;
        // rows of a wave's tile: 8, or ("raygen_tile_rows" 0 = auto, the default) 6 for a launch whose 8x8 tiles would fill less than 70 % of the
        // chip's wave slots -- a single partial round of waves lasts as long as its slowest wave, and a wave with three quarters of the rays
        // lives shorter: the 540 x 570 rectangle of a 1080p / 8 screen tile 97.7 -> 89.9 us (4 rows: 90.2); whole frames and larger tiles keep
        // 8 rows (a 990 x 570 tile: 122 us either way, 128 with 4 rows)
        uint32_t tile_rows = uint32_t(std::max(0, std::min(8, ctx->options[kOptRaygenTileRows])));
        if (tile_rows == 0u) {
            const uint64_t tiles8 = uint64_t((a.col_end - a.col_begin + 7) / 8) * ((rows_traced + 7) / 8), slots = uint64_t(ctx->cu_count) * 32u;
            tile_rows = tiles8 * 10u < slots * 7u ? 6u : 8u;
        }
        const uint32_t tiles_x = (a.col_end - a.col_begin + 7) / 8, tiles_y = (rows_traced + tile_rows - 1) / tile_rows;
        const int waves = ctx->options[kOptWavesPerBlock];
        const uint32_t wv = waves >= 4 ? 4u : (waves >= 2 ? 2u : 1u);
        const uint32_t early_exit = uint32_t(std::max(0, std::min(15, ctx->options[kOptEarlyExit])));
        const bool compact = ctx->options[kOptCompactNodes] != 0 && ctx->nodes16_valid;      // the 32-byte half-precision nodes (two loads per visit instead of three)
        const bool spill = levels < ctx->bvh_depth + 1u;           // the whole stack in LDS (no scratch) whenever the tree's depth fits the configured LDS levels
        const uint32_t n_blocks = ((tiles_x + wv - 1u) / wv) * tiles_y;
        {   // "raygen_cost_order" (see vhr_context::CostOrder)
            const uint32_t key = (tiles_x * 2654435761u) ^ (tiles_y * 40503u) ^ (wv << 28) ^ (tile_rows << 24) ^ (a.row_begin * 97u) ^ (a.col_begin * 193u);
            if (!a.stats && levels >= 5u) prepare_cost_order(ctx, ctx->cost_order_raygen, n_blocks, wv, key, a.co, { tiles_x, (tiles_x + wv - 1u) / wv, wv, 8u, tile_rows, a.col_begin, a.row_begin });
        }
        const uint32_t steal = uint32_t(std::max(0, std::min(63, ctx->options[kOptRaygenSteal])));
        auto go = [&](auto kernel) { launch(ctx, kernel, dim3(n_blocks), dim3(kQueueBlock * wv), stack_bytes * wv, a, levels, threshold, (tiles_x + wv - 1u) / wv, early_exit, tile_rows, steal); };
        auto by_flags = [&](auto waves_c) {
            constexpr int WV = decltype(waves_c)::value;
            const int sel = (compact ? 4 : 0) | (spill ? 2 : 0) | (a.stats ? 1 : 0);
            if (a.fuse_temporal && compact && !a.stats) {       // "fuse_temporal": svgf.comp in the tiles' epilogues (the default node form only)
                if (spill) go(raygen_queue_kernel<WV, true, true, false, true>); else go(raygen_queue_kernel<WV, true, false, false, true>);
                return;
            }
            switch (sel) {
                case 0: go(raygen_queue_kernel<WV, false, false, false>); break;
                case 1: go(raygen_queue_kernel<WV, false, false, true>); break;
                case 2: go(raygen_queue_kernel<WV, false, true, false>); break;
                case 3: go(raygen_queue_kernel<WV, false, true, true>); break;
                case 4: go(raygen_queue_kernel<WV, true, false, false>); break;
                case 5: go(raygen_queue_kernel<WV, true, false, true>); break;
                case 6: go(raygen_queue_kernel<WV, true, true, false>); break;
                default: go(raygen_queue_kernel<WV, true, true, true>); break;
            }
        };
        if (wv == 4u) by_flags(std::integral_constant<int, 4>{});
        else if (wv == 2u) by_flags(std::integral_constant<int, 2>{});
        else by_flags(std::integral_constant<int, 1>{});
    }
    ctx->time_end(kKernelRaygen);
}

// "fuse_temporal": a TraceRays whose shadow / AO launch was held back (launch_raygen) is issued now -- with svgf.comp fused into the
// tiles' epilogues if `fuse` is the dispatch the SVGF pass has just recorded for the very images this launch works on, alone otherwise.
// Called by flush_recorded (with or without `fuse`) and, without, by everything that enqueues on or waits for the stream in between
// (vhr::launch, sync_streams, image copies, pass epilogues and external callbacks, the end of vhr_graph_execute).
struct DeferredRaygen { RaygenArgs a; uint32_t width, height; };
int flush_deferred_raygen(vhr_context *ctx, const TemporalArgs *fuse) {
    if (!ctx->deferred_raygen) return VHR_OK;
    DeferredRaygen d;
    std::memcpy(&d, ctx->deferred_raygen_blob.data(), sizeof(d));
    ctx->deferred_raygen = false;                      // (first: the launch below goes through vhr::launch, which would flush again)
    if (fuse) { d.a.fuse_temporal = 1u; d.a.temporal = *fuse; }
    PassDescription *const running = ctx->cur_pass;
    ctx->cur_pass = ctx->deferred_pass;               // the stamps (and the time) are the ray-tracing pass's
    issue_raygen(ctx, d.a, d.width, d.height);
    if (ctx->deferred_pass && ctx->deferred_pass->stamped_in_kernel) {
        ctx->pending_end = &ctx->d_stamps[ctx->deferred_pass->stamp_index].end;      // stored by the next kernel on the stream
        ctx->deferred_pass->timed = true;
    }
    ctx->cur_pass = running;
    ctx->deferred_pass = nullptr;
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "raygen kernel launch failed");
    return VHR_OK;
}
bool deferred_raygen_matches(const vhr_context *ctx, const TemporalArgs &t) {
    if (!ctx->deferred_raygen) return false;
    DeferredRaygen d;
    std::memcpy(&d, ctx->deferred_raygen_blob.data(), sizeof(d));
    // the dispatch reads this launch's two images and covers exactly the pixels the launch covers (whole-image work)
    return t.raytraced == d.a.shadow_ao && static_cast<const void *>(t.normals) == d.a.normals && t.width == d.a.width && t.height == d.a.height &&
           t.col_begin == 0 && t.row_begin == 0 && t.limit_x == d.a.width && t.row_end >= d.a.height && t.limit_y >= d.a.height &&
           d.a.col_begin == 0 && d.a.row_begin == 0 && d.a.col_end == d.a.width && d.a.row_end == d.a.height;
}

int launch_raygen(vhr_context *ctx, const vhr_per_frame_data &pfd, uint32_t width, uint32_t height, const Image &normals,
                  const Image &depth, Image &shadow_ao, Image *reflections) {
    if (width != normals.width || height != normals.height || width != depth.width || height != depth.height ||
        width != shadow_ao.width || height != shadow_ao.height || (reflections && (reflections->width != width || reflections->height != height)))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "TraceRays: launch size must equal the extent of the pass images");
    RaygenArgs a;
    a.fuse_temporal = 0u;
    a.temporal = TemporalArgs{};
    a.co = CostOrderArgs{};
    a.scene = ctx->device_scene();
    a.pfd = pfd;
    a.tp = ctx->trace_params;
    a.normals = normals.ptr;
    a.depth = static_cast<const float *>(depth.ptr);
    a.shadow_ao = shadow_ao.ptr;
    a.reflections = reflections ? reflections->ptr : nullptr;
    a.width = width;
    a.height = height;
    const uint32_t owned_begin = std::min(ctx->row_begin, height), owned_end = std::min(ctx->row_end, height);
    a.row_begin = owned_begin;
    a.row_end = owned_end;
    // screen tiles: the owned columns, on 16-pixel tile boundaries (a few columns more than owned are harmless: rays are per pixel)
    const uint32_t owned_col_begin = std::min(ctx->col_begin, width), owned_col_end = std::min(ctx->col_end, width);
    a.col_begin = owned_col_begin & ~15u;
    a.col_end = owned_col_end;
    a.stats = ctx->ray_stats_enabled ? ctx->d_ray_stats : nullptr;
    ctx->raytraced_pixels = 0;                 // ray statistics are the hybrid path's again
    if (a.row_end <= a.row_begin || a.col_end <= a.col_begin) return VHR_OK;
    if (ctx->options[kOptTraceOverlap]) {      // strips / tiles: trace the margin the denoiser recomputes too (no exchange of raw visibility)
        a.row_begin = owned_begin > ctx->overlap ? owned_begin - ctx->overlap : 0u;
        a.row_end = uint32_t(std::min<uint64_t>(height, uint64_t(owned_end) + ctx->overlap));
        a.col_begin = (owned_col_begin > ctx->overlap ? owned_col_begin - ctx->overlap : 0u) & ~15u;
        a.col_end = uint32_t(std::min<uint64_t>(width, uint64_t(owned_col_end) + ctx->overlap));
    }
    if (a.stats) {
        if (hipMemsetAsync(ctx->d_ray_stats, 0, 2 * sizeof(RayStats), ctx->stream) != hipSuccess)     // [0] shadow / AO launch, [1] mirror-ray launch
            return ctx->fail(VHR_ERROR_DEVICE, "hipMemsetAsync(ray stats) failed");
    }
    // "fuse_temporal" (opt-in): hold the launch back until the next pass shows its first command -- if that is svgf.comp on this launch's
    // images, the queue kernel runs it in its tiles' epilogues (flush_deferred_raygen).  Only the default kernel has that epilogue, only
    // whole-image work on one stream qualifies, and only a pass nobody hooked an epilogue to (its owner expects the image when it runs).
    {
        // (the epilogue exists in the queue kernel on the 32-byte nodes only: issue_raygen)
        const bool default_kernel = ctx->options[kOptRaygenVariant] != 0 && ctx->options[kOptCompactNodes] != 0 && ctx->nodes16_valid;
        const bool whole = a.row_begin == 0 && a.row_end == height && a.col_begin == 0 && a.col_end == width;
        const bool mirror = a.reflections && a.tp.reflections;
        if (ctx->options[kOptFuseTemporal] && ctx->may_defer_raygen && default_kernel && whole && !mirror && !a.stats && ctx->frames_in_flight == 1 &&
            (ctx->in_kernel_stamps() || !ctx->options[kOptPassTimestamps]) && a.scene.node_count != 0) {
            a.fuse_temporal = 0u;
            DeferredRaygen d;
            std::memset(static_cast<void *>(&d), 0, sizeof(d));
            d.a = a; d.width = width; d.height = height;
            ctx->deferred_raygen_blob.resize(sizeof(d));
            std::memcpy(ctx->deferred_raygen_blob.data(), &d, sizeof(d));
            ctx->deferred_raygen = true;
            ctx->deferred_pass = ctx->cur_pass;
            return VHR_OK;
        }
    }
    issue_raygen(ctx, a, width, height);
    // The mirror ray's launch (raygen.rgen:59-65): not denoised, so owned rows (and columns) only.  It runs BEHIND the shadow / AO launch: beside it
    // (a second stream) the two take as long as one after the other, and with walk and shading in two launches the walk is no faster
    // (profiles/r4_reflection_concurrent.txt, r4d/r4e logs in profiles/r4_reflection_split.txt).
    if (ctx->options[kOptRaygenVariant] != 0 && a.reflections && a.tp.reflections) {
        RaygenArgs m = a;
        m.row_begin = owned_begin;
        m.row_end = owned_end;
        m.col_begin = owned_col_begin & ~15u;
        m.col_end = owned_col_end;
        // "reflection_async": on a stream of its own, behind the shadow / AO launch and beside whatever the caller's stream does next (the SVGF
        // pass, which reads nothing of it).  One stream and event pair per context; the caller's stream joins at the next external pass, at the
        // end of the frame, and wherever the library waits or hands images out.
        hipStream_t const main_stream = ctx->stream;
        PassDescription *const main_pass = ctx->cur_pass;
        bool on_own_stream = false;
        if (ctx->options[kOptReflectionAsync] && ctx->frames_in_flight == 1 && !m.stats) {
            bool ok = true;
            if (!ctx->refl_stream)
                ok = hipStreamCreateWithFlags(&ctx->refl_stream, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&ctx->refl_ready, VHR_JOIN_EVENT_FLAGS) == hipSuccess &&
                     hipEventCreateWithFlags(&ctx->refl_done, VHR_JOIN_EVENT_FLAGS) == hipSuccess;
            if (ok && ctx->refl_pending) ok = ctx->join_refl() == VHR_OK;                    // (one launch at a time on that stream)
            ok = ok && hipEventRecord(ctx->refl_ready, ctx->stream) == hipSuccess && hipStreamWaitEvent(ctx->refl_stream, ctx->refl_ready, 0) == hipSuccess;
            if (ok) {
                on_own_stream = true;
                ctx->stream = ctx->refl_stream;
                ctx->cur_pass = nullptr;             // the pass's time stamps stay on the caller's stream
                ctx->no_stamps = true;
            }
        }
        ctx->time_begin(kKernelReflection);
        if (m.tp.reflections <= 2 && ctx->options[kOptReflectionVariant] != 0) {
            const uint32_t levels = std::max<uint32_t>(1u, std::min<uint32_t>(ctx->bvh_depth + 1u, uint32_t(std::max(1, ctx->options[kOptReflectionLdsStackLevels]))));
            const uint32_t threshold = uint32_t(std::max(1, std::min(64, ctx->options[kOptRefillThreshold])));
            const uint32_t early_exit = uint32_t(std::max(0, std::min(15, ctx->options[kOptReflectionEarlyExit])));
            const uint32_t tile_w = 8u;                                    // (reflection_queue_kernel: 8 x 8 pixels per wave)
            const uint32_t tiles_x = (m.col_end - m.col_begin + tile_w - 1u) / tile_w, tiles_total = tiles_x * ((owned_end - owned_begin + 7) / 8);
            const size_t lds = size_t(levels + 3) * kQueueBlock * sizeof(int) * 2;
            m.co = CostOrderArgs{};
            if (levels >= 5u && !m.stats)                  // "raygen_cost_order" for the mirror-ray launch (its own lifetimes and orders)
                prepare_cost_order(ctx, ctx->cost_order_reflection, (tiles_total + 1u) / 2u, 2u,
                                   (tiles_x * 2654435761u) ^ (tiles_total * 40503u) ^ (uint32_t(m.tp.reflections) << 28) ^ (m.row_begin * 97u) ^ (m.col_begin * 193u), m.co,
                                   { tiles_x, tiles_x, 1u, 8u, 8u, m.col_begin, m.row_begin });
#define VHR_LAUNCH_REFL(SP, B, ST) launch(ctx, (reflection_queue_kernel<SP, B, ST>), dim3((tiles_total + 1) / 2), dim3(kQueueBlock * 2), lds, m, levels, threshold, tiles_x, tiles_total, early_exit)
#define VHR_LAUNCH_REFL_S(SP, B) do { if (m.stats) VHR_LAUNCH_REFL(SP, B, true); else VHR_LAUNCH_REFL(SP, B, false); } while (0)
            const bool spill = levels < ctx->bvh_depth + 1u;
            if (m.tp.reflections == 2) { if (spill) VHR_LAUNCH_REFL_S(true, 2); else VHR_LAUNCH_REFL_S(false, 2); }
            else { if (spill) VHR_LAUNCH_REFL_S(true, 1); else VHR_LAUNCH_REFL_S(false, 1); }
#undef VHR_LAUNCH_REFL_S
#undef VHR_LAUNCH_REFL
        } else {
            launch(ctx, reflection_kernel, dim3((width + 15) / 16, (owned_end - owned_begin + 15) / 16), dim3(kTraceBlock), 0, m);
        }
        ctx->time_end(kKernelReflection);
        if (on_own_stream) {
            ctx->no_stamps = false;
            ctx->cur_pass = main_pass;
            const bool rec = hipEventRecord(ctx->refl_done, ctx->refl_stream) == hipSuccess;
            ctx->stream = main_stream;
            ctx->refl_pending = true;
            ctx->refl_writes = m.reflections;
            ctx->refl_reads[0] = m.normals; ctx->refl_reads[1] = m.depth;
            if (!rec) return ctx->fail(VHR_ERROR_DEVICE, "hipEventRecord(mirror-ray stream) failed");
        }
    }
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "raygen kernel launch failed");
    if (a.stats) {
        if (hipMemcpyAsync(&ctx->h_ray_stats, ctx->d_ray_stats, sizeof(RayStats), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&ctx->h_refl_stats, ctx->d_ray_stats + 1, sizeof(RayStats), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            return ctx->fail(VHR_ERROR_DEVICE, "hipMemcpyAsync(ray stats) failed");
    }
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// next row f4: the raytraced render path's "Raytracing Pass" (raytraced_render_path.cpp:11-47)
//   raytraced_render_path/raygen.rgen:10-23 (+ miss.rmiss:6-8, shadow_miss.rmiss:6-8, closesthit.rchit:10-58), or with the
//   alpha test for shadows switched on raygen_test_alpha.rgen:10-23 + closesthit_test_alpha.rchit:10-51 +
//   shadow_anyhit.rahit:8-27.  One pixel per lane, 8x8 pixels per wave: the primary rays of a tile and the shadow rays
//   towards the directional light are both coherent, so the per-lane walk keeps most lanes on the same nodes.
// ---------------------------------------------------------------------------------------------
// shadow_anyhit.rahit:8-27: true = ignoreIntersectionEXT.  textures[-1] (no base colour texture) reads (0, 0, 0, 0)
// (decision ix of the oracle; out of bounds in the reference).
__device__ bool alpha_ignored(const DeviceScene &sc, uint32_t tri_index, float u, float v) {
    const BvhTri &bt = sc.tris[tri_index];
    const vhr_primitive &prim = sc.primitives[bt.prim];                                  // rahit:9
    if (prim.material.alpha_mask != 1) return false;                                     // rahit:24 (the texture fetch has no other effect)
    const vhr_vertex &a = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * bt.tri + 0]];
    const vhr_vertex &b = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * bt.tri + 1]];
    const vhr_vertex &c = sc.vertices[prim.vertex_offset + sc.indices[prim.index_offset + 3 * bt.tri + 2]];
    const float bx = 1.0f - u - v, by = u, bz = v;                                       // rahit:19
    const float uvx = a.uv0[0] * bx + b.uv0[0] * by + c.uv0[0] * bz;                     // rahit:20
    const float uvy = a.uv0[1] * bx + b.uv0[1] * by + c.uv0[1] * bz;
    const f4 albedo = sample_texture(sc, prim.material.base_color_texture, uvx, uvy);    // rahit:23
    return albedo.w < prim.material.alpha_cutoff;                                        // rahit:24-26
}

// closesthit.rchit:26-57 (ALPHA: closesthit_test_alpha.rchit:26-50) once the shadow ray's answer is known
template <bool ALPHA>
__device__ f4 raytraced_hit_payload(const DeviceScene &sc, const vhr_per_frame_data &pfd, const Hit &h, bool shadowed) {
    const BvhTri &bt = sc.tris[h.tri_index];                                             // rchit:11-24
    const vhr_primitive &prim = sc.primitives[bt.prim];
    const TriAttributes at = interpolate(sc, prim, bt.tri, h.u, h.v);
    f3 albedo;
    if (!ALPHA && prim.material.base_color_texture == -1) {                              // rchit:26-32 (alpha variant: :26, unconditional)
        albedo = f3{ prim.material.base_color[0], prim.material.base_color[1], prim.material.base_color[2] };
    } else {
        const f4 t = sample_texture(sc, prim.material.base_color_texture, at.uvx, at.uvy);
        albedo = f3{ t.x, t.y, t.z };
    }
    const f3 normal = at.normal;
    f3 N = normal;                                                                       // rchit:34-41
    if (prim.material.normal_map >= 0) {
        const f4 tg = interpolate_tangent(sc, prim, bt.tri, h.u, h.v);
        const f3 T = f3{ tg.x, tg.y, tg.z };
        const f4 tx = sample_texture(sc, prim.material.normal_map, at.uvx, at.uvy);
        const f3 tsn = normalize3(f3{ tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f });
        const f3 bitangent = cross3(tsn, T) * tg.w;                                      // sic
        const f3 tangent = normalize3(T - normal * dot3(T, normal));
        N = (tangent * tsn.x + bitangent * tsn.y) + normal * tsn.z;
    }
    const f3 light_dir = -f3{ pfd.directional_light.direction[0], pfd.directional_light.direction[1], pfd.directional_light.direction[2] };
    const f3 lc = f3{ pfd.directional_light.color[0], pfd.directional_light.color[1], pfd.directional_light.color[2] };
    const f3 li = f3{ pfd.directional_light.intensity[0], pfd.directional_light.intensity[1], pfd.directional_light.intensity[2] };
    const f3 albedo_lighting = ALPHA ? albedo * 0.2f : albedo * VHR_PI_INVERSE;          // alpha :39 / :46
    f3 col = albedo_lighting;
    if (!shadowed) {                                                                     // rchit:52-54 / alpha :45-47
        const float nl = fmaxf(dot3(N, light_dir), 0.0f);
        f3 lit = albedo * nl;
        if (!ALPHA) lit = mul3(lit, li);                                                 // the alpha variant drops light_intensity
        lit = mul3(lit, lc);
        col = albedo_lighting + lit;
    }
    return f4{ col.x, col.y, col.z, 1.0f };
}

struct RaytracedArgs {
    DeviceScene scene;
    vhr_per_frame_data pfd;
    uchar4 *out;             // "RaytracedOutput", B8G8R8A8_UNORM
    uint32_t width, height;
    uint32_t row_begin, row_end;
    RayStats *stats;         // nullptr = off; covered_pixels counts the primary hits (= shadow rays)
    CostOrderArgs co;        // "raygen_cost_order" (the queue kernel)
};

__device__ __forceinline__ uint32_t unorm8(float f);

// raytraced_render_path/raygen.rgen:10-23 for one pixel
template <bool ALPHA>
__device__ __forceinline__ void raytraced_pixel(const RaytracedArgs &a, const uint32_t x, const uint32_t y, int *stack, uint32_t &overflow, bool &hit_any) {
    const uint32_t W = a.width, H = a.height;
    const float ux = ((float(x) + 0.5f) / float(W)) * 2.0f - 1.0f;                   // rgen:11-13
    const float uy = ((float(y) + 0.5f) / float(H)) * 2.0f - 1.0f;
    const f4 origin = mat4_mul(a.pfd.camera_view_inverse, f4{ 0.0f, 0.0f, 0.0f, 1.0f });       // rgen:15
    const f4 target = mat4_mul(a.pfd.camera_proj_inverse, f4{ ux, uy, 1.0f, 1.0f });           // rgen:16
    const f3 tn = normalize3(f3{ target.x, target.y, target.z });
    const f4 direction = mat4_mul(a.pfd.camera_view_inverse, f4{ tn.x, tn.y, tn.z, 0.0f });    // rgen:17
    f4 payload = f4{ 0.3f, 0.8f, 0.2f, 1.0f };                                       // miss.rmiss:7
    Hit h;
    if (traverse<false, ALPHA>(a.scene, f3{ origin.x, origin.y, origin.z }, f3{ direction.x, direction.y, direction.z }, 0.1f, 10000.0f,
                               stack, h, overflow)) {                                // rgen:20
        hit_any = true;
        f3 position, unused_normal;
        hit_position_normal(a.scene, h, position, unused_normal);                    // rchit:24
        const f3 light_dir = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1], a.pfd.directional_light.direction[2] };
        Hit sh;
        // shadow ray, rchit:48-50 (alpha :41-43): shadow_payload stays true unless shadow_miss.rmiss:7 runs
        const bool shadowed = traverse<true, ALPHA>(a.scene, position, light_dir, 0.1f, 10000.0f, stack, sh, overflow);
        payload = raytraced_hit_payload<ALPHA>(a.scene, a.pfd, h, shadowed);
    }
    a.out[size_t(y) * W + x] = make_uchar4(uint8_t(unorm8(payload.z)), uint8_t(unorm8(payload.y)), uint8_t(unorm8(payload.x)),
                                           uint8_t(unorm8(payload.w)));             // rgen:22 imageStore, B8G8R8A8
}

template <bool ALPHA>
__global__ __launch_bounds__(kTraceBlock) void raytraced_kernel(const RaytracedArgs a, const Stamps st) {
    vhr_stamp(st);
    __shared__ int s_rt_stack[kTraceStack * kTraceBlock];
    int *stack = s_rt_stack + threadIdx.x;
    uint32_t x, y;
    pixel_of_thread(x, y, a.row_begin);
    bool hit_any = false;
    uint32_t overflow = 0;
    if (x < a.width && y < a.row_end) raytraced_pixel<ALPHA>(a, x, y, stack, overflow, hit_any);
    if (a.stats) {
        const unsigned long long cov = __ballot(hit_any), ovf = __ballot(overflow != 0);
        if ((threadIdx.x & 63u) == 0) {
            if (cov) atomicAdd(&a.stats->covered_pixels, (unsigned long long)__popcll(cov));
            if (ovf) atomicAdd(&a.stats->stack_overflows, (unsigned long long)__popcll(ovf));
        }
    }
}

// Work-queue form (default, `raytraced_variant` 1): a wave owns a 16x8-pixel tile and runs wave_queue_walk twice -- the
// primary rays (closest hit), then one shadow ray per primary hit towards the light (any hit) -- with the ray setup, the
// shadow-ray origins (rchit:24) and closesthit.rchit's shading done by the whole wave in between and after.  Same rays, same
// intersection arithmetic, same shader as raytraced_kernel: bit-identical output.
template <bool SPILL, bool ALPHA>
__global__ __launch_bounds__(kQueueBlock * 2) __attribute__((amdgpu_waves_per_eu(5, 6))) void raytraced_queue_kernel(
    const RaytracedArgs a, const uint32_t stack_levels, const uint32_t refill_threshold, const uint32_t tiles_x, const uint32_t tiles_total,
    const uint32_t early_exit, const Stamps st) {
    vhr_stamp(st);
    extern __shared__ int s_dyn[];                        // per wave: (stack_levels + 3) x 64 ints
    // rows 0-2: primary direction -> primary hit record (triangle, u, v); rows 3-5: shadow-ray origin -> row 3 = its answer
    __shared__ float s_ray_all[2][6][kReflRays];
    __shared__ uint8_t s_list_all[2][kReflRays];
    __shared__ float4 s_cut_all[2][kCutMax][2];           // the tile's shared descent (build_tile_cut), once per walk
    const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    // "raygen_cost_order" for this launch (see raygen_queue_kernel): the first block sorts the previous launch's blocks before its own tiles
    const unsigned long long t_cost0 = a.co.wave_cost ? __builtin_readcyclecounter() : 0ull;
    if (a.co.order_out && blockIdx.x == 0u) order_blocks_by_cost<2>(a.co.cost_prev, a.co.order_blocks, a.co.order_out, reinterpret_cast<uint32_t *>(s_dyn));
    const uint32_t tile = (a.co.block_order ? a.co.block_order[blockIdx.x] : blockIdx.x) * 2u + wave;
    if (tile >= tiles_total) return;                      // waves of a block share nothing and never synchronise
    float (&s_ray)[6][kReflRays] = s_ray_all[wave];
    uint8_t (&s_list)[kReflRays] = s_list_all[wave];
    int *stack = s_dyn + wave * (stack_levels + 3u) * kQueueBlock + lane;
    stack[0] = kStackSentinel;
    const uint32_t W = a.width, H = a.height;
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const f4 origin4 = mat4_mul(a.pfd.camera_view_inverse, f4{ 0.0f, 0.0f, 0.0f, 1.0f });           // rgen:15
    const f3 origin = f3{ origin4.x, origin4.y, origin4.z };
    const f3 light_dir = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1], a.pfd.directional_light.direction[2] };

    // ---- primary rays, whole wave (rgen:11-17) ----
    unsigned long long in_mask[2];
    uint32_t total = 0;
#pragma unroll
    for (uint32_t sub = 0; sub < 2; ++sub) {
        const uint32_t x = tile_x * 16u + sub * 8u + (lane & 7u), y = a.row_begin + tile_y * 8u + (lane >> 3);
        const bool in_range = x < W && y < a.row_end;
        const uint32_t p = sub * 64u + lane;
        if (in_range) {
            const float ux = ((float(x) + 0.5f) / float(W)) * 2.0f - 1.0f;
            const float uy = ((float(y) + 0.5f) / float(H)) * 2.0f - 1.0f;
            const f4 target = mat4_mul(a.pfd.camera_proj_inverse, f4{ ux, uy, 1.0f, 1.0f });
            const f3 tn = normalize3(f3{ target.x, target.y, target.z });
            const f4 direction = mat4_mul(a.pfd.camera_view_inverse, f4{ tn.x, tn.y, tn.z, 0.0f });
            s_ray[0][p] = direction.x; s_ray[1][p] = direction.y; s_ray[2][p] = direction.z;
        }
        const unsigned long long m = __ballot(in_range);
        in_mask[sub] = m;
        if (in_range) s_list[total + lane_rank(m)] = uint8_t(p);
        total += uint32_t(__popcll(m));
    }
    wave_lds_sync();
    const bool traced = a.scene.node_count != 0;
    uint32_t overflow = 0;
    // ---- walk 1: closest hit of the primary rays (rgen:20; ALPHA: gl_RayFlagsNoOpaqueEXT -> the any-hit filter) ----
    // (one origin for every ray: the shared descent follows the boxes around the camera)
    uint32_t cut_n = traced && total ? build_tile_cut_uniform(a.scene, origin, origin, s_cut_all[wave], lane) : 0u;
    wave_queue_walk<SPILL, ALPHA, false>(
        a.scene, stack, stack_levels, lane, traced ? total : 0u, refill_threshold, early_exit, 0.1f, 10000.0f, false, overflow, s_cut_all[wave], cut_n,
        [&](uint32_t r, uint32_t &pix, f3 &ro, f3 &rd) {
            pix = s_list[r];
            ro = origin;
            rd = f3{ s_ray[0][pix], s_ray[1][pix], s_ray[2][pix] };
        },
        [&](uint32_t pix, uint32_t tri, float u, float v) {
            s_ray[0][pix] = __uint_as_float(tri); s_ray[1][pix] = u; s_ray[2][pix] = v;
        });
    wave_lds_sync();
    // ---- shadow rays from the primary hits, whole wave (rchit:24,48-50) ----
    uint32_t nhit = 0;
    f3 omin = f3{ 3.0e38f, 3.0e38f, 3.0e38f }, omax = f3{ -3.0e38f, -3.0e38f, -3.0e38f };   // bounds of the shadow rays' origins
#pragma unroll
    for (uint32_t sub = 0; sub < 2; ++sub) {
        const uint32_t p = sub * 64u + lane;
        const bool inside = traced && ((in_mask[sub] >> lane) & 1ull);
        const uint32_t tri = inside ? __float_as_uint(s_ray[0][p]) : kNoHit;
        const bool hit = tri != kNoHit;
        if (hit) {
            Hit h;
            h.t = 0.0f; h.u = s_ray[1][p]; h.v = s_ray[2][p]; h.tri_index = tri; h.flat = 0;
            f3 position, unused_normal;
            hit_position_normal(a.scene, h, position, unused_normal);
            s_ray[3][p] = position.x; s_ray[4][p] = position.y; s_ray[5][p] = position.z;
            omin = f3{ fminf(omin.x, position.x), fminf(omin.y, position.y), fminf(omin.z, position.z) };
            omax = f3{ fmaxf(omax.x, position.x), fmaxf(omax.y, position.y), fmaxf(omax.z, position.z) };
        }
        const unsigned long long m = __ballot(hit);
        if (hit) s_list[nhit + lane_rank(m)] = uint8_t(p);
        nhit += uint32_t(__popcll(m));
    }
    wave_lds_sync();
    // ---- walk 2: any hit towards the light; the answer (an occluder's triangle or kNoHit) lands in row 3 ----
    cut_n = nhit ? build_tile_cut_uniform(a.scene, omin, omax, s_cut_all[wave], lane) : 0u;
    wave_queue_walk<SPILL, ALPHA, false>(
        a.scene, stack, stack_levels, lane, nhit, refill_threshold, early_exit, 0.1f, 10000.0f, true, overflow, s_cut_all[wave], cut_n,
        [&](uint32_t r, uint32_t &pix, f3 &ro, f3 &rd) {
            pix = s_list[r];
            ro = f3{ s_ray[3][pix], s_ray[4][pix], s_ray[5][pix] };
            rd = light_dir;
        },
        [&](uint32_t pix, uint32_t tri, float, float) { s_ray[3][pix] = __uint_as_float(tri); });
    wave_lds_sync();
    // ---- closesthit.rchit / miss.rmiss and the image store, whole wave ----
#pragma unroll
    for (uint32_t sub = 0; sub < 2; ++sub) {
        if (!((in_mask[sub] >> lane) & 1ull)) continue;
        const uint32_t x = tile_x * 16u + sub * 8u + (lane & 7u), y = a.row_begin + tile_y * 8u + (lane >> 3);
        const uint32_t p = sub * 64u + lane;
        f4 payload = f4{ 0.3f, 0.8f, 0.2f, 1.0f };                                       // miss.rmiss:7
        const uint32_t tri = traced ? __float_as_uint(s_ray[0][p]) : kNoHit;
        if (tri != kNoHit) {
            Hit h;
            h.t = 0.0f; h.u = s_ray[1][p]; h.v = s_ray[2][p]; h.tri_index = tri; h.flat = 0;
            payload = raytraced_hit_payload<ALPHA>(a.scene, a.pfd, h, __float_as_uint(s_ray[3][p]) != kNoHit);
        }
        a.out[size_t(y) * W + x] = make_uchar4(uint8_t(unorm8(payload.z)), uint8_t(unorm8(payload.y)), uint8_t(unorm8(payload.x)),
                                               uint8_t(unorm8(payload.w)));             // rgen:22 imageStore, B8G8R8A8
    }
    if (a.stats && lane == 0) {
        if (nhit) atomicAdd(&a.stats->covered_pixels, (unsigned long long)nhit);
        if (overflow) atomicAdd(&a.stats->stack_overflows, 1ull);
    }
    if (a.co.wave_cost && lane == 0) a.co.wave_cost[tile] = uint32_t(min(__builtin_readcyclecounter() - t_cost0, 0xffffffffull));
}

int launch_raytraced(vhr_context *ctx, const vhr_per_frame_data &pfd, uint32_t width, uint32_t height, Image &out, bool alpha_test) {
    if (width != out.width || height != out.height) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "TraceRays: launch size must equal the extent of RaytracedOutput");
    if (out.bpp != 4) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "TraceRays: RaytracedOutput must be B8G8R8A8_UNORM");
    RaytracedArgs a;
    a.scene = ctx->device_scene();
    a.pfd = pfd;
    a.out = static_cast<uchar4 *>(out.ptr);
    a.width = width;
    a.height = height;
    a.row_begin = std::min(ctx->row_begin, height);          // strips: per-pixel independent, owned rows only
    a.row_end = std::min(ctx->row_end, height);
    a.stats = ctx->ray_stats_enabled ? ctx->d_ray_stats : nullptr;
    if (a.row_end <= a.row_begin) return VHR_OK;
    if (a.stats && hipMemsetAsync(ctx->d_ray_stats, 0, sizeof(RayStats), ctx->stream) != hipSuccess)
        return ctx->fail(VHR_ERROR_DEVICE, "hipMemsetAsync(ray stats) failed");
    const dim3 grid((width + 15) / 16, (a.row_end - a.row_begin + 15) / 16);
    ctx->time_begin(kKernelRaygen);
    if (ctx->options[kOptRaytracedVariant] != 0) {
        const uint32_t levels = std::max<uint32_t>(1u, std::min<uint32_t>(ctx->bvh_depth + 1u, uint32_t(std::max(1, ctx->options[kOptLdsStackLevels]))));
        const uint32_t threshold = uint32_t(std::max(1, std::min(64, ctx->options[kOptRefillThreshold])));
        const uint32_t early_exit = uint32_t(std::max(0, std::min(15, ctx->options[kOptEarlyExit])));
        const uint32_t tiles_x = (width + 15) / 16, tiles_total = tiles_x * ((a.row_end - a.row_begin + 7) / 8);
        const size_t lds = size_t(levels + 3) * kQueueBlock * sizeof(int) * 2;
        const bool spill = levels < ctx->bvh_depth + 1u;
        a.co = CostOrderArgs{};
        if (levels >= 5u && !a.stats)                      // "raygen_cost_order" for this path's launch (its own lifetimes and orders)
            prepare_cost_order(ctx, ctx->cost_order_raytraced, (tiles_total + 1u) / 2u, 2u,
                               (tiles_x * 2654435761u) ^ (tiles_total * 40503u) ^ (uint32_t(alpha_test) << 28) ^ (a.row_begin * 97u), a.co, { tiles_x, tiles_x, 1u, 16u, 8u, 0u, a.row_begin });
#define VHR_LAUNCH_RT(SP, AL) launch(ctx, (raytraced_queue_kernel<SP, AL>), dim3((tiles_total + 1) / 2), dim3(kQueueBlock * 2), lds, a, levels, threshold, tiles_x, tiles_total, early_exit)
        if (alpha_test) { if (spill) VHR_LAUNCH_RT(true, true); else VHR_LAUNCH_RT(false, true); }
        else { if (spill) VHR_LAUNCH_RT(true, false); else VHR_LAUNCH_RT(false, false); }
#undef VHR_LAUNCH_RT
    } else if (alpha_test) {
        launch(ctx, raytraced_kernel<true>, grid, dim3(kTraceBlock), 0, a);
    } else {
        launch(ctx, raytraced_kernel<false>, grid, dim3(kTraceBlock), 0, a);
    }
    ctx->time_end(kKernelRaygen);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "raytraced kernel launch failed");
    if (a.stats && hipMemcpyAsync(&ctx->h_ray_stats, ctx->d_ray_stats, sizeof(RayStats), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return ctx->fail(VHR_ERROR_DEVICE, "hipMemcpyAsync(ray stats) failed");
    ctx->raytraced_pixels = uint64_t(width) * (a.row_end - a.row_begin);
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// stand-in G-buffer producer (gbuf.vert:19-28, gbuf.frag:17-59 encodings) -- primary rays
// ---------------------------------------------------------------------------------------------
struct GbufferArgs {
    DeviceScene scene;
    vhr_per_frame_data pfd;
    float projview[16], prev_projview[16];
    void *normals, *motion;
    float *depth;
    uchar4 *albedo;          // B8G8R8A8_UNORM, optional
    uint32_t width, height;
};

constexpr int kGbufferMaxLayers = 32;      // discarded surfaces a primary ray may step through

__device__ __forceinline__ uint32_t unorm8(float f) { return uint32_t(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f + 0.5f); }

__global__ __launch_bounds__(kTraceBlock) void gbuffer_kernel(const GbufferArgs a, const Stamps st) {
    vhr_stamp(st);
    __shared__ int s_stack[kTraceStack * kTraceBlock];
    int *stack = s_stack + threadIdx.x;
    uint32_t x, y;
    pixel_of_thread(x, y, 0);
    if (x >= a.width || y >= a.height) return;
    const uint32_t W = a.width, H = a.height;
    const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
    const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };
    const f3 pnear = get_world_space_position(a.pfd, 1.0f, u, v);      // reverse-Z: depth 1 is the near plane
    const f3 dir = pnear - cam;
    // Row f2: gbuf.frag:27-32 discards alpha-masked / fully transparent fragments, so the surface behind shows.  A
    // primary-ray caster gets the same picture by stepping past a discarded hit (tmin = its t) and casting again.
    Hit h;
    uint32_t overflow = 0;
    bool visible = false;
    float tmin = 1.0f;
    f4 al = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
    float uvx = 0.0f, uvy = 0.0f;
    for (int layer = 0; layer < kGbufferMaxLayers; ++layer) {
        if (!traverse<false>(a.scene, cam, dir, tmin, 3.0e38f, stack, h, overflow)) break;
        const BvhTri &bt = a.scene.tris[h.tri_index];
        const vhr_primitive &prim = a.scene.primitives[bt.prim];
        const TriAttributes at = interpolate(a.scene, prim, bt.tri, h.u, h.v);
        uvx = at.uvx; uvy = at.uvy;
        al = f4{ prim.material.base_color[0], prim.material.base_color[1], prim.material.base_color[2], prim.material.base_color[3] };
        if (prim.material.base_color_texture != -1) al = sample_texture(a.scene, prim.material.base_color_texture, uvx, uvy);   // :19-26
        if ((prim.material.alpha_mask == 1 && al.w < prim.material.alpha_cutoff) || al.w == 0.0f) { tmin = h.t; continue; }    // :27-32
        visible = true;
        break;
    }
    if (!visible) {                                                                          // clears: hybrid_render_path.cpp:16-19
        store_rgba16f(a.normals, W, x, y, 0.0f, 0.0f, 0.0f, 0.0f);
        store_rgba16f(a.motion, W, x, y, 0.0f, 0.0f, -1.0f, -1.0f);
        a.depth[size_t(y) * W + x] = 0.0f;
        if (a.albedo) a.albedo[size_t(y) * W + x] = make_uchar4(0, 0, 0, 0);
        return;
    }
    const BvhTri &bt = a.scene.tris[h.tri_index];
    const vhr_primitive &prim = a.scene.primitives[bt.prim];
    const f3 P = cam + dir * h.t;
    const f4 clip = mat4_mul(a.projview, f4{ P.x, P.y, P.z, 1.0f });
    a.depth[size_t(y) * W + x] = clip.z / clip.w;
    const TriAttributes at = interpolate(a.scene, prim, bt.tri, h.u, h.v);
    const float *M = a.scene.normal_matrices + 9 * size_t(bt.prim);
    const f3 n = at.normal;
    f3 N = n;
    if (prim.material.normal_map >= 0) {                                                     // gbuf.frag:35-41
        const f4 tx = sample_texture(a.scene, prim.material.normal_map, uvx, uvy);
        const f3 tsn = normalize3(f3{ tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f });
        const f4 tg = interpolate_tangent(a.scene, prim, bt.tri, h.u, h.v);
        const f3 T = f3{ tg.x, tg.y, tg.z };
        const f3 bitangent = cross3(tsn, T) * tg.w;                  // sic: cross(tangent_space_normal, in_tangent.xyz)
        const f3 tangent = normalize3(T - n * dot3(T, n));
        N = (tangent * tsn.x + bitangent * tsn.y) + n * tsn.z;
    }
    const f3 wn = normalize3(f3{ (M[0] * N.x + M[3] * N.y) + M[6] * N.z, (M[1] * N.x + M[4] * N.y) + M[7] * N.z,
                                 (M[2] * N.x + M[5] * N.y) + M[8] * N.z });                  // gbuf.frag:43
    store_rgba16f(a.normals, W, x, y, wn.x, wn.y, wn.z, float(bt.prim));
    const float cx = (float(x) + 0.5f) * a.pfd.display_size_inverse[0];                      // gbuf.frag:46
    const float cy = (float(y) + 0.5f) * a.pfd.display_size_inverse[1];
    const f4 rp = mat4_mul(a.prev_projview, f4{ P.x, P.y, P.z, 1.0f });
    const float px = (rp.x / rp.w) * 0.5f + 0.5f, py = (rp.y / rp.w) * 0.5f + 0.5f;          // gbuf.frag:47
    float metallic = prim.material.metallic_factor, roughness = prim.material.roughness_factor;
    if (prim.material.metallic_roughness_texture != -1) {                                    // gbuf.frag:50-56
        const f4 mr = sample_texture(a.scene, prim.material.metallic_roughness_texture, uvx, uvy);
        metallic *= mr.y;
        roughness *= mr.z;
    }
    store_rgba16f(a.motion, W, x, y, cx - px, cy - py, metallic, roughness);                 // gbuf.frag:58
    if (a.albedo)                                                                            // gbuf.frag:33
        a.albedo[size_t(y) * W + x] = make_uchar4(uint8_t(unorm8(al.z)), uint8_t(unorm8(al.y)), uint8_t(unorm8(al.x)), uint8_t(unorm8(al.w)));
}

static void host_mat4_mul(const float *a, const float *b, float *out) {
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i)
            out[c * 4 + i] = ((a[0 * 4 + i] * b[c * 4 + 0] + a[1 * 4 + i] * b[c * 4 + 1]) + a[2 * 4 + i] * b[c * 4 + 2]) + a[3 * 4 + i] * b[c * 4 + 3];
}

int launch_standin_gbuffer(vhr_context *ctx, const vhr_per_frame_data &pfd, Image &normals, Image &motion, Image &depth, Image *albedo) {
    if (albedo && (albedo->width != depth.width || albedo->height != depth.height || albedo->bpp != 4))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_gbuffer: albedo image must be B8G8R8A8 of the same extent");
    if (normals.width != depth.width || normals.height != depth.height || motion.width != depth.width || motion.height != depth.height)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_gbuffer: image extents differ");
    GbufferArgs a;
    a.scene = ctx->device_scene();
    a.pfd = pfd;
    host_mat4_mul(pfd.camera_proj, pfd.camera_view, a.projview);
    host_mat4_mul(pfd.camera_proj_prev_frame, pfd.camera_view_prev_frame, a.prev_projview);
    a.normals = normals.ptr;
    a.motion = motion.ptr;
    a.depth = static_cast<float *>(depth.ptr);
    a.albedo = albedo ? static_cast<uchar4 *>(albedo->ptr) : nullptr;
    a.width = depth.width;
    a.height = depth.height;
    const dim3 grid((a.width + 15) / 16, (a.height + 15) / 16);
    launch(ctx, gbuffer_kernel, grid, dim3(kTraceBlock), 0, a);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "gbuffer kernel launch failed");
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// Stand-in for the rasterised "Shadow Map Pass" (hybrid_render_path.cpp:58-99, depth_prepass.vert:16-19; BASELINE configs[0]):
// the closest hit of the orthographic ray through every texel centre of directional_light.projview's frustum, from the near
// plane (NDC z = 1, reverse Z) to the far plane; depth = 1 - t, misses keep the clear value 0 (oracle decision xiv).
// ---------------------------------------------------------------------------------------------
struct ShadowMapArgs {
    DeviceScene scene;
    float inv_projview[16];
    float *out;
    uint32_t size, row_begin, row_end;
};

__global__ __launch_bounds__(kTraceBlock) void shadow_map_kernel(const ShadowMapArgs a, const Stamps st) {
    vhr_stamp(st);
    __shared__ int s_stack[kTraceStack * kTraceBlock];
    int *stack = s_stack + threadIdx.x;
    uint32_t x, y;
    pixel_of_thread(x, y, a.row_begin);
    if (x >= a.size || y >= a.row_end) return;
    const float nx = ((float(x) + 0.5f) / float(a.size)) * 2.0f - 1.0f, ny = ((float(y) + 0.5f) / float(a.size)) * 2.0f - 1.0f;
    const f4 pa = mat4_mul(a.inv_projview, f4{ nx, ny, 1.0f, 1.0f }), pb = mat4_mul(a.inv_projview, f4{ nx, ny, 0.0f, 1.0f });
    const f3 o = f3{ pa.x / pa.w, pa.y / pa.w, pa.z / pa.w }, f = f3{ pb.x / pb.w, pb.y / pb.w, pb.z / pb.w };
    Hit h;
    uint32_t overflow = 0;
    float depth = 0.0f;
    if (a.scene.node_count != 0 && traverse<false>(a.scene, o, f - o, 0.0f, 1.0f, stack, h, overflow)) depth = 1.0f - h.t;
    a.out[size_t(y) * a.size + x] = depth;
}

// general 4x4 inverse by cofactors in double, rounded once (the light's projview has no inverse in PerFrameData)
static bool host_mat4_inverse(const float *m, float *out) {
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = double(m[i]);
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return false;
    for (int i = 0; i < 16; ++i) out[i] = float(inv[i] / det);
    return true;
}

int launch_standin_shadow_map(vhr_context *ctx, const vhr_per_frame_data &pfd, Image &shadow_map) {
    if (shadow_map.format != VHR_FORMAT_D32_SFLOAT || shadow_map.width != shadow_map.height)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_shadow_map: a square D32_SFLOAT image is expected (4096 x 4096, hybrid_render_path.cpp:62)");
    ShadowMapArgs a;
    a.scene = ctx->device_scene();
    if (!host_mat4_inverse(pfd.directional_light.projview, a.inv_projview))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "standin_shadow_map: directional_light.projview is singular");
    a.out = static_cast<float *>(shadow_map.ptr);
    a.size = shadow_map.width;
    a.row_begin = 0;
    a.row_end = shadow_map.height;
    launch(ctx, shadow_map_kernel, dim3((a.size + 15) / 16, (a.row_end - a.row_begin + 15) / 16), dim3(kTraceBlock), 0, a);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "shadow map kernel launch failed");
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// next row f3: stand-in for the composition stage (composition.vert:5-8, composition.frag:60-161)
// ---------------------------------------------------------------------------------------------
struct CompositionArgs {
    vhr_per_frame_data pfd;
    const uchar4 *albedo;        // B8G8R8A8_UNORM
    const void *normals, *motion;
    const float *depth;
    const void *shadow_ao;       // RGBA16F (denoised) or RG16F (raw)
    const void *reflections;     // RGBA16F or nullptr: "Raytraced Reflections" (mode 0) / "Screen Space Reflections" (mode 1)
    const void *ssao;            // RGBA16F or nullptr: "Screen Space Ambient Occlusion" (ambient occlusion mode 1)
    const float *shadow_map;     // D32F, shadow_size^2, or nullptr: "Shadow Map" (shadow mode 1)
    float bias_projview[16];     // SHADOW_BIAS_MATRIX * directional_light.projview (composition.frag:82, the matrix product first)
    uint32_t shadow_size;
    uchar4 *out;                 // B8G8R8A8_SRGB
    uint32_t width, height;
    int shadow_mode, ao_mode, reflection_mode, shadow_ao_is_rgba;
};

__device__ __forceinline__ uint8_t srgb8(float c) {       // sRGB attachment store: NaN -> 0, clamp, encode, round
    if (!(c > 0.0f)) return 0;
    if (c >= 1.0f) return 255;
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return uint8_t(e * 255.0f + 0.5f);
}

__global__ __launch_bounds__(256) void composition_kernel(const CompositionArgs a, const Stamps st) {
    vhr_stamp(st);
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.width || j >= a.height) return;
    const uint32_t W = a.width, H = a.height, gy = H - 1 - j;         // flipped presentation viewport (pipeline.cpp:175-178)
    const float u = (float(x) + 0.5f) / float(W), v = (float(gy) + 0.5f) / float(H);
    const uchar4 ab = a.albedo[size_t(gy) * W + x];
    const f3 albedo = f3{ ab.z * (1.0f / 255.0f), ab.y * (1.0f / 255.0f), ab.x * (1.0f / 255.0f) };             // :61
    const float depth = a.depth[size_t(gy) * W + x];                                                               // :62
    const f3 P = get_world_space_position(a.pfd, depth, u, v);                                                     // :63
    const f4 nid = load_rgba16f(a.normals, W, x, gy);                                                              // :64
    const f3 N = f3{ nid.x, nid.y, nid.z };
    const f4 mm = load_rgba16f(a.motion, W, x, gy);                                                                // :65
    float rs = 1.0f, ra = 1.0f;                                                                                    // :67-70
    if (a.shadow_mode == 0 || a.ao_mode == 0) {
        if (a.shadow_ao_is_rgba) { const f4 t = load_rgba16f(a.shadow_ao, W, x, gy); rs = t.x; ra = t.y; }
        else {
            const uint32_t raw = reinterpret_cast<const uint32_t *>(a.shadow_ao)[size_t(gy) * W + x];
            rs = half_bits_to_float(uint16_t(raw & 0xffffu)); ra = half_bits_to_float(uint16_t(raw >> 16));
        }
    }
    const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };
    const f3 V = normalize3(cam - P);                                                                              // :72-75
    const f3 L = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1], a.pfd.directional_light.direction[2] };
    const f3 Hh = normalize3(L + V);
    float shadow = a.shadow_mode == 0 ? rs : 1.0f;                                                                 // :77-80
    if (a.shadow_mode == 1) {                                                                                      // :81-107: 16-tap PCF
        const f4 pl = mat4_mul(a.bias_projview, f4{ P.x, P.y, P.z, 1.0f });
        const float sx = pl.x / pl.w, sy = pl.y / pl.w, sz = pl.z / pl.w;
        const float scale = 1.0f / 4096.0f;
        float lit = 0.0f;
        for (int i = 0; i < 16; ++i) {
            const float ox = (float(i >> 2) - 1.5f) * scale, oy = (float(i & 3) - 1.5f) * scale;                  // offsets[i], :88-93
            const float ds = sample_depth(a.shadow_map, a.shadow_size, a.shadow_size, sx + ox, sy + oy);
            lit += (sz < ds - 1e-4f) ? 0.0f : 1.0f;
        }
        shadow = lit / 16.0f;
    }
    float ao = a.ao_mode == 0 ? ra : 1.0f;                                                                         // :114-121
    if (a.ao_mode == 1) ao = load_rgba16f(a.ssao, W, x, gy).x;                                                     // :117-119 (in_uv is the texel centre)
    const float metallic = fminf(fmaxf(mm.z, 0.0f), 1.0f), roughness = fminf(fmaxf(mm.w, 0.04f), 1.0f);            // :123-125
    const f3 li = f3{ a.pfd.directional_light.intensity[0], a.pfd.directional_light.intensity[1], a.pfd.directional_light.intensity[2] };
    const f3 lc = f3{ a.pfd.directional_light.color[0], a.pfd.directional_light.color[1], a.pfd.directional_light.color[2] };
    const f3 f0 = f3{ 0.04f * (1.0f - metallic) + albedo.x * metallic, 0.04f * (1.0f - metallic) + albedo.y * metallic,
                      0.04f * (1.0f - metallic) + albedo.z * metallic };                                           // :131-132
    const f3 F = fresnel_schlick(f0, Hh, V);
    const float ndl = fmaxf(dot3(N, L), 0.0f);                                                                     // :135
    const f3 ambient = albedo * (ao * VHR_PI_INVERSE);                                                             // :137
    const f3 dp = f3{ (1.0f - F.x) * (1.0f - metallic), (1.0f - F.y) * (1.0f - metallic), (1.0f - F.z) * (1.0f - metallic) };
    const f3 diffuse = mul3(mul3(f3{ dp.x * albedo.x / VHR_PI, dp.y * albedo.y / VHR_PI, dp.z * albedo.z / VHR_PI } * ndl, li), lc) * shadow;   // :138
    const float dg = D_GGX(roughness, N, Hh) * G_GGX(roughness, N, V, L);
    const float invd = 1.0f / fmaxf(4.0f * fmaxf(dot3(N, V), 0.0f) * fmaxf(dot3(N, L), 0.0f), 1e-6f);
    f3 spec = mul3(mul3(f3{ dg * F.x * invd, dg * F.y * invd, dg * F.z * invd } * ndl, li), lc) * shadow;          // :139
    if ((a.reflection_mode == 0 || a.reflection_mode == 1) && a.reflections) {                                     // :139-156 (the same blend for both sources)
        const f4 r = load_rgba16f(a.reflections, W, x, gy);
        const f3 refl = f3{ r.x, r.y, r.z } * shadow;
        if (metallic == 1.0f) spec = refl;
        else spec = f3{ spec.x * (1.0f - roughness) + refl.x * roughness, spec.y * (1.0f - roughness) + refl.y * roughness,
                        spec.z * (1.0f - roughness) + refl.z * roughness };
    }
    const f3 lighting = ambient + diffuse + spec;                                                                  // :160-162
    a.out[size_t(j) * W + x] = make_uchar4(srgb8(lighting.z), srgb8(lighting.y), srgb8(lighting.x), 255);
}

int launch_composition(vhr_context *ctx, const vhr_per_frame_data &pfd, const vhr_composition_desc &d, const Image &albedo, const Image &normals,
                       const Image &motion, const Image &depth, const Image &shadow_ao, const Image *reflections, const Image *ssao,
                       const Image *shadow_map, Image &out) {
    const uint32_t W = depth.width, H = depth.height;
    const Image *all[] = { &albedo, &normals, &motion, &shadow_ao, &out };
    for (const Image *im : all)
        if (im->width != W || im->height != H) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: image extents differ");
    if (reflections && (reflections->width != W || reflections->height != H)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: image extents differ");
    if (albedo.bpp != 4 || out.bpp != 4 || normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || motion.format != VHR_FORMAT_R16G16B16A16_SFLOAT ||
        depth.format != VHR_FORMAT_D32_SFLOAT || (shadow_ao.format != VHR_FORMAT_R16G16B16A16_SFLOAT && shadow_ao.format != VHR_FORMAT_R16G16_SFLOAT))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: unexpected image format");
    if (d.shadow_mode == 1 && (!shadow_map || shadow_map->format != VHR_FORMAT_D32_SFLOAT || shadow_map->width != shadow_map->height))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: shadow_mode 1 needs the square D32 \"Shadow Map\" image");
    for (int m : { d.shadow_mode, d.ambient_occlusion_mode, d.reflection_mode })
        if (m < 0 || m > 2) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: modes are 0 (ray traced), 1 (screen space) or 2 (off)");
    if (d.ambient_occlusion_mode == 1 && (!ssao || ssao->width != W || ssao->height != H || ssao->format != VHR_FORMAT_R16G16B16A16_SFLOAT))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "composition: ambient_occlusion_mode 1 needs the R16G16B16A16 \"Screen Space Ambient Occlusion\" image");
    CompositionArgs a;
    a.ssao = ssao ? ssao->ptr : nullptr;
    a.shadow_map = shadow_map ? static_cast<const float *>(shadow_map->ptr) : nullptr;
    a.shadow_size = shadow_map ? shadow_map->width : 0;
    static const float kShadowBias[16] = { 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.5f, 0.5f, 0.0f, 1.0f };   // common.glsl:6-11
    host_mat4_mul(kShadowBias, pfd.directional_light.projview, a.bias_projview);
    a.pfd = pfd;
    a.albedo = static_cast<const uchar4 *>(albedo.ptr);
    a.normals = normals.ptr; a.motion = motion.ptr;
    a.depth = static_cast<const float *>(depth.ptr);
    a.shadow_ao = shadow_ao.ptr;
    a.reflections = reflections ? reflections->ptr : nullptr;
    a.out = static_cast<uchar4 *>(out.ptr);
    a.width = W; a.height = H;
    a.shadow_mode = d.shadow_mode; a.ao_mode = d.ambient_occlusion_mode; a.reflection_mode = d.reflection_mode;
    a.shadow_ao_is_rgba = shadow_ao.format == VHR_FORMAT_R16G16B16A16_SFLOAT;
    launch(ctx, composition_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, a);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "composition kernel launch failed");
    return VHR_OK;
}

int upload_srgb_lut(const float *lut) {
    return hipMemcpyToSymbol(HIP_SYMBOL(c_srgb_lut), lut, 256 * sizeof(float)) == hipSuccess ? 0 : -1;
}

// raytraced_render_path/composition.vert:5-8 + composition.frag:11-13: "RaytracedOutput" sampled at the texel centre,
// written to the B8G8R8A8_SRGB swapchain through the flipped presentation viewport (pipeline.cpp:175-178).
__global__ __launch_bounds__(256) void raytraced_composition_kernel(const uchar4 *in, uchar4 *out, uint32_t W, uint32_t H, const Stamps st) {
    vhr_stamp(st);
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || j >= H) return;
    const uchar4 p = in[size_t(H - 1 - j) * W + x];
    out[size_t(j) * W + x] = make_uchar4(srgb8(p.x * (1.0f / 255.0f)), srgb8(p.y * (1.0f / 255.0f)), srgb8(p.z * (1.0f / 255.0f)), p.w);
}

int launch_raytraced_composition(vhr_context *ctx, const Image &in, Image &out) {
    if (in.width != out.width || in.height != out.height) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "raytraced composition: image extents differ");
    if (in.bpp != 4 || out.bpp != 4) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "raytraced composition: 4-byte texels expected");
    launch(ctx, raytraced_composition_kernel, dim3((in.width + 63) / 64, (in.height + 3) / 4), dim3(256), 0,
                       static_cast<const uchar4 *>(in.ptr), static_cast<uchar4 *>(out.ptr), in.width, in.height);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "raytraced composition kernel launch failed");
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// vhr_debug_ray_triangle: decision (vi) as the walkers' triangle test computes it, on explicit (ray, triangle) pairs -- what tests/ hold against the oracle's
// orc_ray_triangle and against exact arithmetic (tests/golden/kat_decision_vi.json), without a scene or a tree in between.  17 floats per pair:
// o, d, v0, e1, e2, tmin, tmax; out: hit (0 / 1) and (t, u, v).  One pair per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ray_triangle_pairs_kernel(const float *pairs, const uint32_t n, uint32_t *hit, float *tuv, const Stamps st) {
    vhr_stamp(st);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *p = pairs + size_t(i) * 17u;
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const bool h = ray_triangle(f3{ p[0], p[1], p[2] }, f3{ p[3], p[4], p[5] }, f3{ p[6], p[7], p[8] }, f3{ p[9], p[10], p[11] }, f3{ p[12], p[13], p[14] }, p[15], p[16], t, u, v);
    hit[i] = h ? 1u : 0u;
    tuv[size_t(i) * 3u] = h ? t : 0.0f; tuv[size_t(i) * 3u + 1u] = h ? u : 0.0f; tuv[size_t(i) * 3u + 2u] = h ? v : 0.0f;
}

int launch_ray_triangle_pairs(vhr_context *ctx, const float *pairs, uint32_t n, uint32_t *hit, float *tuv) {
    if (n == 0) return VHR_OK;
    float *d_pairs = nullptr, *d_tuv = nullptr;
    uint32_t *d_hit = nullptr;
    const size_t pb = size_t(n) * 17u * sizeof(float), tb = size_t(n) * 3u * sizeof(float), hb = size_t(n) * sizeof(uint32_t);
    int rc = VHR_OK;
    if (hipMalloc(&d_pairs, pb) != hipSuccess || hipMalloc(&d_tuv, tb) != hipSuccess || hipMalloc(&d_hit, hb) != hipSuccess)
        rc = ctx->fail(VHR_ERROR_DEVICE, "vhr_debug_ray_triangle: device allocation failed");
    if (rc == VHR_OK && hipMemcpyAsync(d_pairs, pairs, pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = ctx->fail(VHR_ERROR_DEVICE, "vhr_debug_ray_triangle: upload failed");
    if (rc == VHR_OK) {
        launch(ctx, ray_triangle_pairs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, static_cast<const float *>(d_pairs), n, d_hit, d_tuv);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hit, d_hit, hb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(tuv, d_tuv, tb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = ctx->fail(VHR_ERROR_DEVICE, "vhr_debug_ray_triangle: launch or download failed");
    }
    (void)hipFree(d_pairs); (void)hipFree(d_tuv); (void)hipFree(d_hit);
    return rc;
}

}  // namespace vhr
