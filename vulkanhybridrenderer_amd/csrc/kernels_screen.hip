// K6 / K7 / K8: the screen-space alternatives of the hybrid render path for gfx950.
//
//   data/shaders/hybrid_render_path/ssao.comp:14-53      -> ssao_kernel        ("SSAO Pass", hybrid_render_path.cpp:143-168)
//   data/shaders/hybrid_render_path/ssao_blur.comp:11-26 -> ssao_blur_kernel   ("SSAO Blur Pass", :170-199)
//   data/shaders/hybrid_render_path/ssr.comp:16-137      -> ssr_kernel         ("SSR Pass", :210-242)
//
// Numerics: this unit is compiled with -ffp-contract=off and uses the operation order of the CPU restatement under oracle/ (which is
// the shaders'), correctly rounded divisions and square roots and the shared sin/cos polynomial, so the three kernels
// reproduce the oracle bit for bit.  That matters for ssr.comp: its march compares a difference of two distances with
// 0.3 and a thickness, and a last-bit difference moves the hit by a whole step.
// texture() = the default sampler (LINEAR, REPEAT) on linear images: four loads and fp32 weights (oracle decision x);
// min / max drop NaN operands (decision xii: sky texels unproject to inf / NaN); the r16f-declared outputs are RGBA16F
// images and receive all four components (decision xi).
#include "device_math.hpp"
#include "vhr_internal.hpp"

namespace vhr {

constexpr int kScreenBlockX = 64;   // one wave per row segment: 512-byte coalesced RGBA16F rows
constexpr int kScreenBlockY = 4;

__device__ __forceinline__ f4 unpack_half4(uint2 raw) {
    const float2 lo = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x));
    const float2 hi = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
    return f4{ lo.x, lo.y, hi.x, hi.y };
}
__device__ __forceinline__ uint2 pack_half4(float a, float b, float c, float d) {
    const __half2 lo = __floats2half2_rn(a, b), hi = __floats2half2_rn(c, d);
    uint2 r;
    r.x = *reinterpret_cast<const uint32_t *>(&lo);
    r.y = *reinterpret_cast<const uint32_t *>(&hi);
    return r;
}
__device__ __forceinline__ f4 sample_rgba16f(const uint2 *img, uint32_t W, uint32_t H, float u, float v) {
    const Taps t = bilinear_taps(W, H, u, v);
    const uint2 *r0 = img + size_t(t.y0) * W, *r1 = img + size_t(t.y1) * W;
    const f4 a = unpack_half4(r0[t.x0]), b = unpack_half4(r0[t.x1]), c = unpack_half4(r1[t.x0]), d = unpack_half4(r1[t.x1]);
    return f4{ blend(t, a.x, b.x, c.x, d.x), blend(t, a.y, b.y, c.y, d.y), blend(t, a.z, b.z, c.z, d.z), blend(t, a.w, b.w, c.w, d.w) };
}
// B8G8R8A8_UNORM sampled as (r, g, b, a)
__device__ __forceinline__ f4 unorm_bgra(uchar4 p) {
    return f4{ float(p.z) * (1.0f / 255.0f), float(p.y) * (1.0f / 255.0f), float(p.x) * (1.0f / 255.0f), float(p.w) * (1.0f / 255.0f) };
}
__device__ __forceinline__ f4 sample_bgra8(const uchar4 *img, uint32_t W, uint32_t H, float u, float v) {
    const Taps t = bilinear_taps(W, H, u, v);
    const uchar4 *r0 = img + size_t(t.y0) * W, *r1 = img + size_t(t.y1) * W;
    const f4 a = unorm_bgra(r0[t.x0]), b = unorm_bgra(r0[t.x1]), c = unorm_bgra(r1[t.x0]), d = unorm_bgra(r1[t.x1]);
    return f4{ blend(t, a.x, b.x, c.x, d.x), blend(t, a.y, b.y, c.y, d.y), blend(t, a.z, b.z, c.z, d.z), blend(t, a.w, b.w, c.w, d.w) };
}

// glsl_common.h:111-115 / :118-122
__device__ __forceinline__ f3 unproject(const float *inverse, float depth, float u, float v) {
    const f4 r = mat4_mul(inverse, f4{ u * 2.0f - 1.0f, v * 2.0f - 1.0f, depth, 1.0f });
    return f3{ r.x / r.w, r.y / r.w, r.z / r.w };
}

// ---------------------------------------------------------------------------------------------
// K6: ssao.comp
// ---------------------------------------------------------------------------------------------
struct SsaoArgs {
    const uint2 *normals;      // RGBA16F
    const float *depth;        // D32F
    uint2 *out;                // RGBA16F
    vhr_per_frame_data pfd;
    uint32_t width, height, row_begin, row_end, limit_x, limit_y;
    float radius;
};

__global__ __launch_bounds__(kScreenBlockX *kScreenBlockY) void ssao_kernel(const SsaoArgs a, const Stamps st) {
    vhr_stamp(st);
    const uint32_t x = blockIdx.x * kScreenBlockX + threadIdx.x;
    const uint32_t y = a.row_begin + blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= a.limit_x || y >= a.row_end || y >= a.limit_y) return;
    const uint32_t W = a.width, H = a.height;
    const float cu = float(x) * a.pfd.display_size_inverse[0], cv = float(y) * a.pfd.display_size_inverse[1];   // :15
    const float current_depth = sample_depth(a.depth, W, H, cu, cv);                                             // :16
    if (current_depth == 0.0f) { a.out[size_t(y) * W + x] = make_uint2(0u, 0u); return; }                       // :17-24
    const f3 P = unproject(a.pfd.camera_proj_inverse, current_depth, cu, cv);                                    // :25
    const f4 n4 = sample_rgba16f(a.normals, W, H, cu, cv);
    const float *m = a.pfd.camera_view;                                                                          // :26 mat3(view) * n
    const f3 N = f3{ (m[0] * n4.x + m[4] * n4.y) + m[8] * n4.z, (m[1] * n4.x + m[5] * n4.y) + m[9] * n4.z,
                     (m[2] * n4.x + m[6] * n4.y) + m[10] * n4.z };
    const float perspective_radius = a.radius / P.z;                                                             // :28-29
    const float beta = 1e-4f;
    uint32_t rng = seed_thread((y * uint32_t(a.pfd.display_size[1]) + x) * a.pfd.frame_index);                   // :32
    float sum = 0.0f;
    for (int i = 0; i < 16; ++i) {                                                                               // :33-45
        const float ang = (random01(rng) * 2.0f) * VHR_PI;
        const float dist = random01(rng) * perspective_radius;
        float sn, cs;
        exact_sincos(ang, sn, cs);
        const float su = cu + cs * dist, sv = cv + sn * dist;
        const f3 Q = unproject(a.pfd.camera_proj_inverse, sample_depth(a.depth, W, H, su, sv), su, sv);
        const f3 V = Q - P;
        sum += fmaxf(dot3(V, N) - beta, 0.0f) / (dot3(V, V) + 1e-4f);
    }
    const float ao = fmaxf(1.0f - ((2.0f * 1.0f) / 16.0f) * sum, 0.0f);                                          // :47
    a.out[size_t(y) * W + x] = pack_half4(ao, ao, ao, ao);                                                       // :49-53
}

// ---------------------------------------------------------------------------------------------
// K7: ssao_blur.comp -- 13x13 box in the shader's summation order (row by row, left to right), from an LDS tile
// ---------------------------------------------------------------------------------------------
struct BlurArgs {
    const uint2 *in;
    uint2 *out;
    uint32_t width, height, row_begin, row_end, limit_x, limit_y;
    float display_w, display_h;
};
constexpr int kBlurR = 6;
constexpr int kBlurTileX = 64, kBlurTileY = 16;
constexpr int kBlurLdsW = kBlurTileX + 2 * kBlurR, kBlurLdsH = kBlurTileY + 2 * kBlurR;
constexpr int kBlurLdsStride = 80;                  // floats per LDS row: a multiple of 4, so every thread's 16-float window is 16-byte aligned

__global__ __launch_bounds__(256) void ssao_blur_kernel(const BlurArgs a, const Stamps st) {
    vhr_stamp(st);
    // .x of the raw image as fp32; texels the shader skips (outside the display or the image) are stored as -0.0f, whose
    // addition leaves every partial sum unchanged (x + -0 == x, and the sum starts at +0)
    __shared__ __attribute__((aligned(16))) float tile[kBlurLdsH][kBlurLdsStride];
    const int tx0 = int(blockIdx.x) * kBlurTileX, ty0 = int(a.row_begin) + int(blockIdx.y) * kBlurTileY;
    // (constant trip count, unrolled: the nine loads of a thread are in flight together instead of one round trip each)
    constexpr int kFill = (kBlurLdsW * kBlurLdsH + 255) / 256;
    uint32_t raw[kFill];
#pragma unroll
    for (int it = 0; it < kFill; ++it) {
        const int i = int(threadIdx.x) + it * 256;
        const int ly = i / kBlurLdsW, lx = i - ly * kBlurLdsW;
        const int sx = tx0 + lx - kBlurR, sy = ty0 + ly - kBlurR;
        const bool inside = i < kBlurLdsW * kBlurLdsH && sx >= 0 && sy >= 0 && float(sx) < a.display_w && float(sy) < a.display_h &&
                            uint32_t(sx) < a.width && uint32_t(sy) < a.height;
        raw[it] = inside ? a.in[size_t(sy) * a.width + sx].x & 0xffffu : 0x8000u;        // fp16 -0.0
    }
#pragma unroll
    for (int it = 0; it < kFill; ++it) {
        const int i = int(threadIdx.x) + it * 256;
        if (i < kBlurLdsW * kBlurLdsH) {
            const int ly = i / kBlurLdsW, lx = i - ly * kBlurLdsW;
            tile[ly][lx] = __half2float(__ushort_as_half(uint16_t(raw[it])));
        }
    }
    __syncthreads();
    // A thread owns four pixels next to each other: per tile row it reads one 16-float window (four ds_read_b128) that holds
    // the 13 taps of each of them, and feeds four accumulators -- each in the shader's order (rows top to bottom, taps left to
    // right), so every sum has the shader's roundings while the LDS traffic per pixel drops from 169 to 52 words.
    const int gx = (int(threadIdx.x) & 15) * 4, ly = int(threadIdx.x) >> 4;
    float ao[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    float4 nxt[4];
    {
        const float4 *row = reinterpret_cast<const float4 *>(&tile[ly][gx]);
#pragma unroll
        for (int q = 0; q < 4; ++q) nxt[q] = row[q];
    }
    // one row per trip with the next row's window already on its way (fully unrolled, the compiler hoists all 52 reads:
    // 216 VGPRs, 2 waves per SIMD)
#pragma unroll 1
    for (int y = 0; y <= 2 * kBlurR; ++y) {
        float w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[4 * q] = nxt[q].x; w[4 * q + 1] = nxt[q].y; w[4 * q + 2] = nxt[q].z; w[4 * q + 3] = nxt[q].w; }
        if (y < 2 * kBlurR) {
            const float4 *row = reinterpret_cast<const float4 *>(&tile[ly + y + 1][gx]);
#pragma unroll
            for (int q = 0; q < 4; ++q) nxt[q] = row[q];
        }
#pragma unroll
        for (int x = 0; x <= 2 * kBlurR; ++x) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ao[k] += w[k + x];
        }
    }
    const uint32_t cy = uint32_t(ty0 + ly);
    if (cy >= a.row_end || cy >= a.limit_y) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cx = uint32_t(tx0 + gx + k);
        if (cx >= a.limit_x) continue;
        const float r = ao[k] / (13.0f * 13.0f);
        a.out[size_t(cy) * a.width + cx] = pack_half4(r, r, r, r);
    }
}

// ---------------------------------------------------------------------------------------------
// K8: ssr.comp
// ---------------------------------------------------------------------------------------------
struct SsrArgs {
    const uchar4 *albedo;      // B8G8R8A8_UNORM
    const uint2 *normals, *motion;
    const float *depth;
    uint2 *out;
    vhr_per_frame_data pfd;
    uint32_t width, height, row_begin, row_end, limit_x, limit_y;
    float ray_distance, step_size, thickness;
    int32_t bsearch_steps;
};

__device__ __forceinline__ float distance3(f3 p, f3 q) { const f3 d = p - q; return sqrtf(dot3(d, d)); }

__global__ __launch_bounds__(kScreenBlockX *kScreenBlockY) void ssr_kernel(const SsrArgs a, const Stamps st) {
    vhr_stamp(st);
    const uint32_t x = blockIdx.x * kScreenBlockX + threadIdx.x;
    const uint32_t y = a.row_begin + blockIdx.y * kScreenBlockY + threadIdx.y;
    if (x >= a.limit_x || y >= a.row_end || y >= a.limit_y) return;
    const uint32_t W = a.width, H = a.height;
    // ssr.comp:23 `pfd.camera_proj * pfd.camera_view * vec4(v, 1)`: the matrix product first (left to right)
    float pv[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            pv[c * 4 + r] = ((a.pfd.camera_proj[0 * 4 + r] * a.pfd.camera_view[c * 4 + 0] + a.pfd.camera_proj[1 * 4 + r] * a.pfd.camera_view[c * 4 + 1]) +
                             a.pfd.camera_proj[2 * 4 + r] * a.pfd.camera_view[c * 4 + 2]) + a.pfd.camera_proj[3 * 4 + r] * a.pfd.camera_view[c * 4 + 3];
    const f3 cam = f3{ a.pfd.camera_view_inverse[12], a.pfd.camera_view_inverse[13], a.pfd.camera_view_inverse[14] };
    const float cu = float(x) * a.pfd.display_size_inverse[0], cv = float(y) * a.pfd.display_size_inverse[1];   // :68
    const float fragment_depth = sample_depth(a.depth, W, H, cu, cv);
    const f3 P = unproject(a.pfd.camera_viewproj_inverse, fragment_depth, cu, cv);                               // :72
    const f4 n4 = sample_rgba16f(a.normals, W, H, cu, cv);
    const f3 N = f3{ n4.x, n4.y, n4.z };
    const f3 I = normalize3(P - cam);                                                                            // :74
    const float ni2 = 2.0f * dot3(N, I);
    const f3 rdir = normalize3(I - N * ni2);                                                                     // :75
    const int n_steps = int(a.ray_distance / a.step_size);                                                       // :83
    bool found = false;
    float prev_step = 0.0f, final_step = 0.0f;
    float fu = 0.0f, fv = 0.0f;
    auto probe = [&](float offset, float &su, float &sv) {          // delta_distance of :85-92 / :113-120
        const f3 rp = P + rdir * offset;
        const float d_ray = distance3(cam, rp);
        const f4 clip = mat4_mul(pv, f4{ rp.x, rp.y, rp.z, 1.0f });
        su = (clip.x / clip.w) * 0.5f + 0.5f;
        sv = (clip.y / clip.w) * 0.5f + 0.5f;
        const f3 sp = unproject(a.pfd.camera_viewproj_inverse, sample_depth(a.depth, W, H, su, sv), su, sv);
        return d_ray - distance3(cam, sp);
    };
    for (int i = 0; i < n_steps; ++i) {                                                                          // :83-101
        const float offset = a.step_size * float(i);
        float su, sv;
        const float delta = probe(offset, su, sv);
        if (delta > 0.3f && delta < a.thickness) { final_step = offset; found = true; break; }
        prev_step = offset;
    }
    if (!found) { a.out[size_t(y) * W + x] = make_uint2(0u, 0u); return; }                                       // :62-66, :103-105
    float mid_step = (prev_step + final_step) * 0.5f;                                                            // :108
    for (int i = 0; i < a.bsearch_steps; ++i) {                                                                  // :110-128
        const float delta = probe(mid_step, fu, fv);
        if (delta > 0.3f && delta < a.thickness) {
            mid_step = (prev_step + mid_step) * 0.5f;
        } else {
            const float tmp = mid_step;
            mid_step = mid_step + (mid_step - prev_step);
            prev_step = tmp;
        }
    }
    // compute_lighting(final_uv), :28-59
    const f4 a4 = sample_bgra8(a.albedo, W, H, fu, fv);
    const f3 albedo = f3{ a4.x, a4.y, a4.z };
    const f3 position = unproject(a.pfd.camera_viewproj_inverse, sample_depth(a.depth, W, H, fu, fv), fu, fv);
    const f4 mm = sample_rgba16f(a.motion, W, H, fu, fv);
    const f3 V = normalize3(cam - position);
    const f3 L = -f3{ a.pfd.directional_light.direction[0], a.pfd.directional_light.direction[1], a.pfd.directional_light.direction[2] };
    const f4 ln = sample_rgba16f(a.normals, W, H, fu, fv);
    const f3 Nl = f3{ ln.x, ln.y, ln.z };
    const f3 Hh = normalize3(L + V);
    const float metallic = fminf(fmaxf(mm.z, 0.0f), 1.0f);
    const float roughness = fminf(fmaxf(mm.w, 0.04f), 1.0f);
    const float ambient_factor = VHR_PI_INVERSE * 0.2f;
    const f3 li = f3{ a.pfd.directional_light.intensity[0], a.pfd.directional_light.intensity[1], a.pfd.directional_light.intensity[2] };
    const f3 lc = f3{ a.pfd.directional_light.color[0], a.pfd.directional_light.color[1], a.pfd.directional_light.color[2] };
    const f3 f0 = f3{ 0.04f * (1.0f - metallic) + albedo.x * metallic, 0.04f * (1.0f - metallic) + albedo.y * metallic,
                      0.04f * (1.0f - metallic) + albedo.z * metallic };
    const f3 F = fresnel_schlick(f0, Hh, V);
    const f3 ambient = albedo * ambient_factor;
    const f3 diff = diffuse_brdf(metallic, albedo, F);
    const f3 spec = specular_brdf(roughness, F, V, L, Nl, Hh);
    const float ndl = fmaxf(dot3(Nl, L), 0.0f);
    const f3 lit = ambient + mul3(mul3((diff + spec) * ndl, li), lc);
    a.out[size_t(y) * W + x] = pack_half4(lit.x, lit.y, lit.z, 1.0f);                                            // :131-135
}

// ---------------------------------------------------------------------------------------------
// launchers (row strips: see vhr_set_strip; `extend` rows beyond the owned ones are computed where a later pass reads them)
// ---------------------------------------------------------------------------------------------
static void owned_rows(const vhr_context *ctx, uint32_t height, uint32_t extend, uint32_t &r0, uint32_t &r1) {
    const uint32_t b = std::min(ctx->row_begin, height), e = std::min(ctx->row_end, height);
    r0 = b > extend ? b - extend : 0;
    r1 = uint32_t(std::min<uint64_t>(height, uint64_t(e) + extend));
    if (e <= b) { r0 = r1 = 0; }
}

static bool same_extent(const Image &a, const Image &b) { return a.width == b.width && a.height == b.height; }

int launch_ssao(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &depth, Image &out, float radius,
                uint32_t x_groups, uint32_t y_groups) {
    if (!same_extent(normals, depth) || !same_extent(normals, out)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssao.comp: image extents differ");
    if (normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || depth.format != VHR_FORMAT_D32_SFLOAT || out.format != VHR_FORMAT_R16G16B16A16_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssao.comp: unexpected image format (normals / output R16G16B16A16, depth D32)");
    SsaoArgs a;
    a.normals = static_cast<const uint2 *>(normals.ptr);
    a.depth = static_cast<const float *>(depth.ptr);
    a.out = static_cast<uint2 *>(out.ptr);
    a.pfd = pfd;
    a.width = out.width; a.height = out.height;
    a.limit_x = uint32_t(std::min<uint64_t>(out.width, uint64_t(x_groups) * 8));
    a.limit_y = uint32_t(std::min<uint64_t>(out.height, uint64_t(y_groups) * 8));
    owned_rows(ctx, out.height, kBlurR, a.row_begin, a.row_end);          // the blur reads 6 rows either side of the owned ones
    a.radius = radius;
    if (a.row_end <= a.row_begin || !a.limit_x || !a.limit_y) return VHR_OK;
    const dim3 grid((a.limit_x + kScreenBlockX - 1) / kScreenBlockX, (a.row_end - a.row_begin + kScreenBlockY - 1) / kScreenBlockY);
    ctx->time_begin(kKernelSsao);
    launch(ctx, ssao_kernel, grid, dim3(kScreenBlockX, kScreenBlockY), 0, a);
    ctx->time_end(kKernelSsao);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "ssao kernel launch failed");
    return VHR_OK;
}

int launch_ssao_blur(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &in, Image &out, uint32_t x_groups, uint32_t y_groups) {
    if (!same_extent(in, out)) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssao_blur.comp: image extents differ");
    if (in.format != VHR_FORMAT_R16G16B16A16_SFLOAT || out.format != VHR_FORMAT_R16G16B16A16_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssao_blur.comp: both images are R16G16B16A16 (hybrid_render_path.cpp:173-176)");
    BlurArgs a;
    a.in = static_cast<const uint2 *>(in.ptr);
    a.out = static_cast<uint2 *>(out.ptr);
    a.width = out.width; a.height = out.height;
    a.limit_x = uint32_t(std::min<uint64_t>(out.width, uint64_t(x_groups) * 8));
    a.limit_y = uint32_t(std::min<uint64_t>(out.height, uint64_t(y_groups) * 8));
    owned_rows(ctx, out.height, 0, a.row_begin, a.row_end);
    a.display_w = pfd.display_size[0];
    a.display_h = pfd.display_size[1];
    if (a.row_end <= a.row_begin || !a.limit_x || !a.limit_y) return VHR_OK;
    const dim3 grid((a.limit_x + kBlurTileX - 1) / kBlurTileX, (a.row_end - a.row_begin + kBlurTileY - 1) / kBlurTileY);
    ctx->time_begin(kKernelSsaoBlur);
    launch(ctx, ssao_blur_kernel, grid, dim3(256), 0, a);
    ctx->time_end(kKernelSsaoBlur);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "ssao blur kernel launch failed");
    return VHR_OK;
}

int launch_ssr(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &albedo, const Image &normals, const Image &motion,
               const Image &depth, Image &out, const vhr_ssr_push_constants &pc, uint32_t x_groups, uint32_t y_groups) {
    if (!same_extent(albedo, out) || !same_extent(normals, out) || !same_extent(motion, out) || !same_extent(depth, out))
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssr.comp: image extents differ");
    if (albedo.format != VHR_FORMAT_B8G8R8A8_UNORM || normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || motion.format != VHR_FORMAT_R16G16B16A16_SFLOAT ||
        depth.format != VHR_FORMAT_D32_SFLOAT || out.format != VHR_FORMAT_R16G16B16A16_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "ssr.comp: unexpected image format (hybrid_render_path.cpp:212-219)");
    SsrArgs a;
    a.albedo = static_cast<const uchar4 *>(albedo.ptr);
    a.normals = static_cast<const uint2 *>(normals.ptr);
    a.motion = static_cast<const uint2 *>(motion.ptr);
    a.depth = static_cast<const float *>(depth.ptr);
    a.out = static_cast<uint2 *>(out.ptr);
    a.pfd = pfd;
    a.width = out.width; a.height = out.height;
    a.limit_x = uint32_t(std::min<uint64_t>(out.width, uint64_t(x_groups) * 8));
    a.limit_y = uint32_t(std::min<uint64_t>(out.height, uint64_t(y_groups) * 8));
    owned_rows(ctx, out.height, 0, a.row_begin, a.row_end);
    a.ray_distance = pc.ray_distance; a.step_size = pc.step_size; a.thickness = pc.thickness; a.bsearch_steps = pc.bsearch_steps;
    if (a.row_end <= a.row_begin || !a.limit_x || !a.limit_y) return VHR_OK;
    const dim3 grid((a.limit_x + kScreenBlockX - 1) / kScreenBlockX, (a.row_end - a.row_begin + kScreenBlockY - 1) / kScreenBlockY);
    ctx->time_begin(kKernelSsr);
    launch(ctx, ssr_kernel, grid, dim3(kScreenBlockX, kScreenBlockY), 0, a);
    ctx->time_end(kKernelSsr);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "ssr kernel launch failed");
    return VHR_OK;
}

}  // namespace vhr
