// Device-side scalar / vector helpers for the trace kernels.
//
// "Exact arithmetic" contract: everything that feeds a ray origin, a ray direction or a
// ray/triangle decision uses only individually rounded IEEE fp32 +, -, *, /, sqrt in the operation order
// written here (this translation unit is compiled with -ffp-contract=off; hipcc's default
// correctly-rounded fp32 divide/sqrt and fp32 denormal support are kept), so results are reproducible
// bit for bit on any IEEE machine.  sin/cos use a fixed polynomial instead of the ocml routines.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

namespace vhr {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 make_f3(float x, float y, float z) { return f3{ x, y, z }; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{ -a.x, -a.y, -a.z }; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ f3 mul3(f3 a, f3 b) { return f3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return f3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
__device__ __forceinline__ f3 normalize3(f3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return a * inv;
}

// GLSL mat4 * vec4 on a column-major matrix; columns accumulated left to right
__device__ __forceinline__ f4 mat4_mul(const float *m, f4 v) {
    f4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
__device__ __forceinline__ f3 mat4_mul_point(const float *m, f3 p) {
    return f3{ ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
               ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
               ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] };
}

// ---- fp16 image storage: stores round to nearest even, loads widen exactly ----
__device__ __forceinline__ float half_bits_to_float(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// (the empty asm pins the fp32 VALUE: left alone, the compiler folds the multiply that produced f into the conversion -- v_fma_mixlo_f16 rounds the
// unrounded product to a half ONCE, and where the fp32 product is a tie between two halves that lands one step off the store of the rounded fp32
// value (decision iii; found in round 6 on 142 of 2 M normals of the stand-in G-buffer, the one image whose stored values are products))
__device__ __forceinline__ uint16_t float_to_half_bits(float f) { asm("" : "+v"(f)); return __half_as_ushort(__float2half_rn(f)); }

// ---- data/shaders/common.glsl:47-76 RNG (integer exact) ----
__device__ __forceinline__ uint32_t seed_thread(uint32_t seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
__device__ __forceinline__ uint32_t xorshift(uint32_t &state) {
    state ^= (state << 13);
    state ^= (state >> 17);
    state ^= (state << 5);
    return state;
}
__device__ __forceinline__ float random01(uint32_t &state) {
    return __uint_as_float(0x3f800000u | (xorshift(state) >> 9)) - 1.0f;
}

// ---- sin/cos on [0, 2*pi]: quadrant reduction (3-term Cody-Waite split of pi/2) + Cephes minimax
//      polynomials on [-pi/4, pi/4]; every operation individually rounded ----
__device__ __forceinline__ void exact_sincos(float phi, float &s_out, float &c_out) {
    float k = rintf(phi * 0.636619772367581343f);
    float r = ((phi - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188e-8f;
    float z = r * r;
    float s = ((((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z) * r) + r;
    float c = ((((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z) * z) + (1.0f - 0.5f * z);
    int q = int(k) & 3;
    float ss = (q & 1) ? c : s;
    float cc = (q & 1) ? s : c;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    s_out = ss;
    c_out = cc;
}

#define VHR_TWO_PI 6.28318530717958647692528f
#define VHR_PI 3.14159265358979323846264f
#define VHR_PI_INVERSE 0.31830988618379067153776f

// common.glsl:29-34
__device__ __forceinline__ f3 uniform_sample_cone(float ux, float uy, float cos_theta_max) {
    float cos_theta = (1.0f - ux) + ux * cos_theta_max;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = uy * VHR_TWO_PI;
    float s, c;
    exact_sincos(phi, s, c);
    return f3{ c * sin_theta, s * sin_theta, cos_theta };
}
// common.glsl:37-42
__device__ __forceinline__ f3 cosine_hemisphere(float ux, float uy) {
    float s, c;
    exact_sincos(VHR_TWO_PI * uy, s, c);
    float sq = sqrtf(ux);
    return f3{ sq * c, sq * s, sqrtf(1.0f - ux) };
}
// common.glsl:80-93 Frisvad basis; returns R * v for R = onb_from_unit_vector(n)
__device__ __forceinline__ f3 onb_transform(f3 n, f3 v) {
    f3 c0, c1;
    if (n.z < -0.9999999f) {
        c0 = f3{ 0.0f, -1.0f, 0.0f };
        c1 = f3{ -1.0f, 0.0f, 0.0f };
    } else {
        float a = 1.0f / (1.0f + n.z);
        float b = ((-n.x) * n.y) * a;
        c0 = f3{ 1.0f - (n.x * n.x) * a, b, -n.x };
        c1 = f3{ b, 1.0f - (n.y * n.y) * a, -n.y };
    }
    return f3{ (c0.x * v.x + c1.x * v.y) + n.x * v.z, (c0.y * v.x + c1.y * v.y) + n.y * v.z,
               (c0.z * v.x + c1.z * v.y) + n.z * v.z };
}

// common.glsl:116-150
__device__ __forceinline__ f3 fresnel_schlick(f3 f0, f3 H, f3 V) {
    const float hv = fmaxf(dot3(H, V), 0.0f);
    const float om = 1.0f - hv;
    const float p5 = om * om * om * om * om;
    return f3{ f0.x + (1.0f - f0.x) * p5, f0.y + (1.0f - f0.y) * p5, f0.z + (1.0f - f0.z) * p5 };
}
__device__ __forceinline__ float D_GGX(float roughness, f3 N, f3 H) {
    const float a2 = roughness * roughness;
    const float nh = fmaxf(dot3(N, H), 0.0f);
    const float f = nh * nh * (a2 - 1.0f) + 1.0f;
    return a2 / (VHR_PI * f * f);
}
__device__ __forceinline__ float G_GGX(float roughness, f3 N, f3 V, f3 L) {
    const float k = ((roughness + 1.0f) * (roughness + 1.0f)) * 0.125f;
    const float nv = fmaxf(dot3(N, V), 0.0f), nl = fmaxf(dot3(N, L), 0.0f);
    return (nv / (nv * (1.0f - k) + k)) * (nl / (nl * (1.0f - k) + k));
}
// common.glsl:140-150
__device__ __forceinline__ f3 specular_brdf(float roughness, f3 F, f3 V, f3 L, f3 N, f3 H) {
    const float dg = D_GGX(roughness, N, H) * G_GGX(roughness, N, V, L);
    const float denom = 4.0f * fmaxf(dot3(N, V), 0.0f) * fmaxf(dot3(N, L), 0.0f);
    const float invd = 1.0f / fmaxf(denom, 1e-6f);
    return f3{ dg * F.x * invd, dg * F.y * invd, dg * F.z * invd };
}
__device__ __forceinline__ f3 diffuse_brdf(float metallic, f3 albedo, f3 F) {
    const f3 dp = f3{ (1.0f - F.x) * (1.0f - metallic), (1.0f - F.y) * (1.0f - metallic), (1.0f - F.z) * (1.0f - metallic) };
    return f3{ dp.x * albedo.x / VHR_PI, dp.y * albedo.y / VHR_PI, dp.z * albedo.z / VHR_PI };
}

// ---- texture() through the default sampler (LINEAR / REPEAT, resource_manager.cpp:58-69) on linear images ----
// REPEAT addressing.  Nearly every coordinate is on screen: the integer modulo (~20 instructions for a run-time divisor, two per
// sample) sits behind a test that whole waves usually skip.
__device__ __forceinline__ int wrap_repeat(int i, int n) {
    if (__builtin_expect(uint32_t(i) < uint32_t(n), 1)) return i;
    const int m = i % n;
    return m < 0 ? m + n : m;
}

struct Taps { int x0, x1, y0, y1; float ax, ay, bx, by; };
// oracle decision (x): u * W - 0.5, floor, wrapped texels, fp32 weights
__device__ __forceinline__ Taps bilinear_taps(uint32_t W, uint32_t H, float u, float v) {
    const float fx = u * float(W) - 0.5f, fy = v * float(H) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    Taps t;
    t.ax = fx - x0f; t.ay = fy - y0f;
    t.bx = 1.0f - t.ax; t.by = 1.0f - t.ay;
    t.x0 = wrap_repeat(int(x0f), int(W));          // v_cvt_i32_f32: saturates, NaN -> 0 (decision xiii)
    t.y0 = wrap_repeat(int(y0f), int(H));
    t.x1 = t.x0 + 1 == int(W) ? 0 : t.x0 + 1;
    t.y1 = t.y0 + 1 == int(H) ? 0 : t.y0 + 1;
    return t;
}
__device__ __forceinline__ float blend(const Taps &t, float t00, float t10, float t01, float t11) {
    return (t00 * t.bx + t10 * t.ax) * t.by + (t01 * t.bx + t11 * t.ax) * t.ay;
}
__device__ __forceinline__ float sample_depth(const float *img, uint32_t W, uint32_t H, float u, float v) {
    const Taps t = bilinear_taps(W, H, u, v);
    const float *r0 = img + size_t(t.y0) * W, *r1 = img + size_t(t.y1) * W;
    const float t00 = r0[t.x0], t10 = r0[t.x1], t01 = r1[t.x0], t11 = r1[t.x1];
    return blend(t, t00, t10, t01, t11);
}
// XCD-aware block -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one),
// and each XCD has its own 4 MiB L2.  With the natural order every XCD would walk tiles from all over the screen
// and pull the whole visible BVH / triangle set (> 4 MiB) through its L2; instead the blocks that share an XCD get
// one contiguous band of tile rows, so each L2 only holds the geometry its band's rays meet.  Bijective for any
// block count (the remainder rows go to the first bands); placement is a speed hint only, never correctness.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t id, uint32_t n) {
    const uint32_t xcd = id & 7u, slot = id >> 3;
    const uint32_t q = n >> 3, r = n & 7u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
}

}  // namespace vhr
