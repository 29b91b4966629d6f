// K3, svgf.comp:16-145 for ONE pixel -- shared by svgf_temporal_kernel (csrc/kernels_svgf.hip) and by the ray-tracing queue kernel's tile
// epilogue (csrc/kernels_trace.hip, "fuse_temporal": the pixel's visibility goes from the wave's LDS straight into the temporal filter).
// No FMA contraction anywhere in here (the pragma below; kernels_trace.hip is compiled with -ffp-contract=off as a whole): with the
// shader's (= the oracle's) roundings the output is bit-identical to the oracle's.  (A fused mix(prev, cur, 0.2) lands on the other
// side of an fp16 tie in ~2 % of the pixels.)
#pragma once

#include "device_math.hpp"
#include "vhr_internal.hpp"

namespace vhr {

#pragma clang fp contract(off)

__device__ __forceinline__ f4 unpack_rgba16f(uint2 raw) {
    const float2 lo = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
    const float2 hi = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
    return f4{ lo.x, lo.y, hi.x, hi.y };
}
__device__ __forceinline__ uint2 pack_rgba16f(float a, float b, float c, float d) {
    const __half2 lo = __floats2half2_rn(a, b), hi = __floats2half2_rn(c, d);
    uint2 r;
    r.x = *reinterpret_cast<const uint32_t *>(&lo);
    r.y = *reinterpret_cast<const uint32_t *>(&hi);
    return r;
}
__device__ __forceinline__ float2 unpack_rg16f(uint32_t raw) { return __half22float2(*reinterpret_cast<const __half2 *>(&raw)); }
__device__ __forceinline__ uint32_t pack_rg16f(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *reinterpret_cast<const uint32_t *>(&h);
}
// GLSL int(float): truncation toward zero; v_cvt_i32_f32 maps NaN to 0 and saturates
__device__ __forceinline__ int f2i(float f) { return int(f); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// ---------------------------------------------------------------------------------------------
// K3: svgf.comp
// ---------------------------------------------------------------------------------------------
// No FMA contraction in K3 (the pragma sits above the shared helpers, mixf included): the kernel is bound by its
// gathers, not by arithmetic, and with the shader's (= the oracle's) roundings its output is bit-identical to the
// oracle's.  (A fused mix(prev, cur, 0.2) lands on the other side of an fp16 tie in ~2 % of the pixels.)

// svgf.comp:16-39
__device__ __forceinline__ bool is_valid_reprojection(const TemporalArgs &a, int px, int py, int current_object_id, f3 current_normal) {
    if (px < 0 || py < 0 || float(px) >= a.display_w || float(py) >= a.display_h) return false;
    if (uint32_t(px) >= a.width || uint32_t(py) >= a.height) return false;     // imageLoad outside the image returns 0
    const f4 pn = unpack_rgba16f(a.prev_normals[size_t(py) * a.width + px]);
    if (current_object_id != f2i(pn.w)) return false;
    if (dot3(current_normal, f3{ pn.x, pn.y, pn.z }) < 0.70710678118654752440084f) return false;
    return true;
}

// everything of svgf.comp behind the reads of the pixel's own normal / id and raw visibility (:46-145): `nid` = the unpacked normals / id
// texel, (current_shadow, current_ao) = the RG16F texel of "Raytraced Shadows and Ambient Occlusion" as read back from the image
__device__ __forceinline__ void svgf_temporal_pixel(const TemporalArgs &a, const uint32_t cx, const uint32_t cy, const f4 nid, const float current_shadow, const float current_ao) {
    const uint32_t idx = cy * a.width + cx;
    auto at8 = [](const uint2 *base, uint32_t i) { return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + i * 8u); };
    const f3 current_normal = f3{ nid.x, nid.y, nid.z };
    const int current_object_id = f2i(nid.w);
    const f4 mv = unpack_rgba16f(at8(a.motion, idx));                                       // :46

    const float pcx = (float(cx) - mv.x * a.display_w) + 0.5f;                              // :52
    const float pcy = (float(cy) - mv.y * a.display_h) + 0.5f;
    const float x = pcx - floorf(pcx), y = pcy - floorf(pcy);                               // :53-54
    const int ax = f2i(pcx), ay = f2i(pcy);                                                 // :55
    const float bw[4] = { (1.0f - x) * (1.0f - y), x * (1.0f - y), (1.0f - x) * y, x * y };  // :57

    float prev_shadow = 0.0f, prev_ao = 0.0f, sum = 0.0f;
    float psm0 = 0.0f, psm1 = 0.0f, pam0 = 0.0f, pam1 = 0.0f;
    {   // :65-77 -- the four bilinear taps.  The two taps of a row are neighbouring texels: ONE load fetches both (16 bytes of an
        // RGBA16F image at 8-byte alignment, 8 bytes of the RG16F one at 4 -- the CU's address unit charges per load instruction,
        // not per byte: profiles/r2_pmc_memory.txt), six gathers instead of twelve, all issued before any is consumed.  The pair
        // starts at column clamp(ax, 0, W - 2); a tap outside the image is rejected below whatever was loaded for it.
        struct __attribute__((aligned(8))) Pair8 { uint2 t[2]; };
        struct __attribute__((aligned(4))) Pair4 { uint32_t t[2]; };
        uint2 pn[4], hs[4];
        uint32_t mo[4];
        bool inb[4];
        if (a.width >= 2u) {
            const int x0 = min(max(ax, 0), int(a.width) - 2);
            Pair8 pnp[2], hsp[2];
            Pair4 mop[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t sidx = uint32_t(min(max(ay + r, 0), int(a.height) - 1)) * a.width + uint32_t(x0);
                pnp[r] = *reinterpret_cast<const Pair8 *>(reinterpret_cast<const char *>(a.prev_normals) + sidx * 8u);
                hsp[r] = *reinterpret_cast<const Pair8 *>(reinterpret_cast<const char *>(a.history) + sidx * 8u);
                mop[r] = *reinterpret_cast<const Pair4 *>(reinterpret_cast<const char *>(a.moments_in) + sidx * 4u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sx = ax + (i & 1), sy = ay + (i >> 1);
                inb[i] = sx >= 0 && sy >= 0 && float(sx) < a.display_w && float(sy) < a.display_h && uint32_t(sx) < a.width && uint32_t(sy) < a.height;
                const bool second = sx > x0;            // (in bounds: sx is x0 or x0 + 1)
                pn[i] = second ? pnp[i >> 1].t[1] : pnp[i >> 1].t[0];
                hs[i] = second ? hsp[i >> 1].t[1] : hsp[i >> 1].t[0];
                mo[i] = second ? mop[i >> 1].t[1] : mop[i >> 1].t[0];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sx = ax + (i & 1), sy = ay + (i >> 1);
                inb[i] = sx >= 0 && sy >= 0 && float(sx) < a.display_w && float(sy) < a.display_h && uint32_t(sx) < a.width && uint32_t(sy) < a.height;
                const size_t sidx = size_t(min(max(sy, 0), int(a.height) - 1)) * a.width + size_t(min(max(sx, 0), int(a.width) - 1));
                pn[i] = a.prev_normals[sidx];
                hs[i] = a.history[sidx];
                mo[i] = a.moments_in[sidx];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f4 n4 = unpack_rgba16f(pn[i]);
            const bool ok = inb[i] && current_object_id == f2i(n4.w) &&
                            !(dot3(current_normal, f3{ n4.x, n4.y, n4.z }) < 0.70710678118654752440084f);      // :16-39
            if (ok) {
                const f4 h4 = unpack_rgba16f(hs[i]);
                const float2 m = unpack_rg16f(mo[i]);                // RG16F read as vec4 = (r, g, 0, 1)
                prev_shadow += bw[i] * h4.x;
                prev_ao += bw[i] * h4.y;
                psm0 += bw[i] * m.x; psm1 += bw[i] * m.y;
                pam0 += bw[i] * 0.0f; pam1 += bw[i] * 1.0f;
                sum += bw[i];
            }
        }
    }
    bool valid = sum > 1e-6f;                                                               // :78
    if (!valid) {                                                                           // :81-97
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int sx = ax + dx, sy = ay + dy;
                if (is_valid_reprojection(a, sx, sy, current_object_id, current_normal)) {
                    const size_t sidx = size_t(sy) * a.width + sx;
                    const f4 hs = unpack_rgba16f(a.history[sidx]);
                    const float2 m = unpack_rg16f(a.moments_in[sidx]);
                    prev_shadow += hs.x; prev_ao += hs.y;
                    psm0 += m.x; psm1 += m.y; pam0 += 0.0f; pam1 += 1.0f;
                    sum += 1.0f;
                }
            }
        valid = sum > 1e-6f;
    }
    float sm0 = current_shadow, sm1 = current_shadow * current_shadow;                      // :99-102
    float am0 = current_ao, am1 = current_ao * current_ao;
    float out_s = current_shadow, out_a = current_ao;
    if (valid) {                                                                            // :106-126
        prev_shadow /= sum; psm0 /= sum; psm1 /= sum;
        prev_ao /= sum; pam0 /= sum; pam1 /= sum;
        sm0 = mixf(psm0, sm0, 0.2f); sm1 = mixf(psm1, sm1, 0.2f);
        am0 = mixf(pam0, am0, 0.2f); am1 = mixf(pam1, am1, 0.2f);
        out_s = mixf(prev_shadow, current_shadow, 0.2f);
        out_a = mixf(prev_ao, current_ao, 0.2f);
    }
    const float sv = fmaxf(0.0f, sm1 - sm0 * sm0);
    const float av = fmaxf(0.0f, am1 - am0 * am0);
    *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(a.integrated_out) + idx * 8u) = pack_rgba16f(out_s, out_a, sv, av);      // :118-135
    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.moments_out) + idx * 4u) = pack_rg16f(sm0, sm1);                    // :138-144 (RG16F image keeps .xy)
}

}  // namespace vhr
