// K3 / K4 / K5: the SVGF denoiser of the hybrid render path for gfx950.
//
//   data/shaders/hybrid_render_path/svgf.comp:16-145               -> svgf_temporal_kernel
//   data/shaders/hybrid_render_path/svgf_atrous_filter.comp:17-103 -> svgf_atrous_kernel
//   ComputeExecutionContext::BlitImage (compute_execution_context.cpp:178-211) -> copy_image_rows
//
// Numerics: fp32 compute, fp16 storage (stores round to nearest even), the same operation order as the
// shaders; exp / reciprocal use the hardware approximations, so agreement with the CPU oracle is within the
// float tolerance stated in tests/test_svgf_gpu.py, not bit-exact.  Images are linear row-major buffers:
// RGBA16F = one 8-byte uint2 per pixel, RG16F = one 4-byte word per pixel, so a 64-lane wave reading 64
// consecutive pixels of a row moves 512 contiguous bytes (dwordx2 per lane).
#include <type_traits>

#include "device_math.hpp"
#include "vhr_internal.hpp"
#include "svgf_temporal.hpp"

namespace vhr {

static int issue_cmd(vhr_context *ctx, const SvgfCmd &cmd);

#pragma clang fp contract(off)      // K3; K4 switches contraction back on below

constexpr int kSvgfBlockX = 64;    // (svgf_atrous_kernel, the direct A-B form) one wave per image row segment: fully coalesced 512-byte row reads
constexpr int kSvgfBlockY = 4;

// ---------------------------------------------------------------------------------------------
// K3: svgf.comp -- the per-pixel body lives in svgf_temporal.hpp (shared with the ray-tracing kernel's tile epilogue)
// ---------------------------------------------------------------------------------------------
template <int BX, int BY>
__global__ __launch_bounds__(BX *BY) void svgf_temporal_kernel(const TemporalArgs a, const Stamps st) {
    vhr_stamp(st);
    const uint32_t cx = a.col_begin + blockIdx.x * BX + threadIdx.x;
    const uint32_t cy = a.row_begin + blockIdx.y * BY + threadIdx.y;
    if (cx >= a.limit_x || cy >= a.row_end || cy >= a.limit_y) return;
    // 32-bit texel indices on the (uniform) image bases: an image is far below 2^29 texels, and a size_t index costs a v_mad_u64_u32 and
    // 64-bit shifts / adds per address (r3c: 27.0 -> 26.x us)
    const uint32_t idx = cy * a.width + cx;
    auto at8 = [](const uint2 *base, uint32_t i) { return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + i * 8u); };
    auto at4 = [](const uint32_t *base, uint32_t i) { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + i * 4u); };
    const f4 nid = unpack_rgba16f(at8(a.normals, idx));                                     // :43-45
    const float2 cur = unpack_rg16f(at4(a.raytraced, idx));                                 // :47-49
    svgf_temporal_pixel(a, cx, cy, nid, cur.x, cur.y);
}

static void strip_rows(const vhr_context *ctx, uint32_t height, uint32_t extend, uint32_t &r0, uint32_t &r1) {
    const uint32_t b = std::min(ctx->row_begin, height), e = std::min(ctx->row_end, height);
    r0 = b > extend ? b - extend : 0;
    r1 = uint32_t(std::min<uint64_t>(height, uint64_t(e) + extend));
    if (e <= b) { r0 = r1 = 0; }
}
// the same for the columns of a screen tile (vhr_set_tile; a strip owns every column)
static void strip_cols(const vhr_context *ctx, uint32_t width, uint32_t extend, uint32_t &c0, uint32_t &c1) {
    const uint32_t b = std::min(ctx->col_begin, width), e = std::min(ctx->col_end, width);
    c0 = b > extend ? b - extend : 0;
    c1 = uint32_t(std::min<uint64_t>(width, uint64_t(e) + extend));
    if (e <= b) { c0 = c1 = 0; }
}

int launch_svgf_temporal(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &motion,
                         const Image &raytraced, const Image &prev_normals, const Image &history, Image &moments,
                         Image &integrated_out, uint32_t x_groups, uint32_t y_groups) {
    const uint32_t W = normals.width, H = normals.height;
    const Image *all[] = { &motion, &raytraced, &prev_normals, &history, &moments, &integrated_out };
    for (const Image *im : all)
        if (im->width != W || im->height != H) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf.comp: image extents differ");
    if (normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || motion.format != VHR_FORMAT_R16G16B16A16_SFLOAT ||
        prev_normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || history.format != VHR_FORMAT_R16G16B16A16_SFLOAT ||
        integrated_out.format != VHR_FORMAT_R16G16B16A16_SFLOAT || raytraced.format != VHR_FORMAT_R16G16_SFLOAT ||
        moments.format != VHR_FORMAT_R16G16_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf.comp: unexpected image format (moments / raytraced must be R16G16, the rest R16G16B16A16)");
    if (!moments.alt) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf.comp: moments history must be a storage image");
    TemporalArgs a;
    a.normals = static_cast<const uint2 *>(normals.ptr);
    a.motion = static_cast<const uint2 *>(motion.ptr);
    a.prev_normals = static_cast<const uint2 *>(prev_normals.ptr);
    a.history = static_cast<const uint2 *>(history.ptr);
    a.raytraced = static_cast<const uint32_t *>(raytraced.ptr);
    a.moments_in = static_cast<const uint32_t *>(moments.ptr);
    a.integrated_out = static_cast<uint2 *>(integrated_out.ptr);
    a.moments_out = static_cast<uint32_t *>(moments.alt);
    a.width = W; a.height = H;
    a.limit_x = uint32_t(std::min<uint64_t>(W, uint64_t(x_groups) * 8));
    a.limit_y = uint32_t(std::min<uint64_t>(H, uint64_t(y_groups) * 8));
    strip_rows(ctx, H, ctx->overlap, a.row_begin, a.row_end);
    {
        uint32_t c1;
        strip_cols(ctx, W, ctx->overlap, a.col_begin, c1);
        a.limit_x = std::min(a.limit_x, c1);
    }
    a.display_w = pfd.display_size[0];
    a.display_h = pfd.display_size[1];
    // the dispatch reads a snapshot (ptr) and writes the new moments (alt): flip (oracle decision ii) -- host-side state, so
    // at record time: later commands of the pass see the flipped image
    std::swap(moments.ptr, moments.alt);
    SvgfCmd cmd{};
    cmd.kind = SvgfCmd::Temporal;
    cmd.t = a;
    if (ctx->recording) { ctx->recorded.push_back(cmd); return VHR_OK; }
    return issue_cmd(ctx, cmd);
}

// ---------------------------------------------------------------------------------------------
// K4: svgf_atrous_filter.comp
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(fast)

__device__ __forceinline__ float pow128(float x) {      // max(0, pow(x, 128)); pow of x <= 0 defined as 0
    x = fmaxf(x, 0.0f);
    x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
    return x;
}

__global__ __launch_bounds__(kSvgfBlockX *kSvgfBlockY) void svgf_atrous_kernel(const AtrousArgs a, const Stamps st) {
    vhr_stamp(st);
    const int cx = int(blockIdx.x * kSvgfBlockX + threadIdx.x);
    const int cy = int(a.row_begin + blockIdx.y * kSvgfBlockY + threadIdx.y);
    if (uint32_t(cx) >= a.limit_x || uint32_t(cy) >= a.row_end || uint32_t(cy) >= a.limit_y) return;
    const int W = int(a.width), H = int(a.height);
    // bounds as the shader writes them (float compare against display_size), plus the image extent
    const int max_x = min(W, int(ceilf(a.display_w))), max_y = min(H, int(ceilf(a.display_h)));
    const size_t idx = size_t(cy) * W + cx;
    const f4 np = unpack_rgba16f(a.normals[idx]);                                           // :55-57
    const f3 normal_p = f3{ np.x, np.y, np.z };
    const int id_p = f2i(np.w);
    const f4 p = unpack_rgba16f(a.in[idx]);                                                 // :59

    float var_s = 0.0f, var_a = 0.0f;                                                       // :17-38
#pragma unroll
    for (int y = -1; y <= 1; ++y)
#pragma unroll
        for (int x = -1; x <= 1; ++x) {
            const int sx = cx + x, sy = cy + y;
            if (sx < 0 || sx >= max_x || sy < 0 || sy >= max_y) continue;
            const float w = (x == 0 ? 0.5f : 0.25f) * (y == 0 ? 0.5f : 0.25f);
            const f4 q = unpack_rgba16f(a.in[size_t(sy) * W + sx]);
            var_s += w * q.z;
            var_a += w * q.w;
        }
    const float inv_s = __frcp_rn(4.0f * sqrtf(var_s) + 1e-6f);                             // :48-50 denominators
    const float inv_a = __frcp_rn(4.0f * sqrtf(var_a) + 1e-6f);

    float sw_s = 1.0f, sw_a = 1.0f;                                                         // :70-71
    float s0 = p.x, s1 = p.y, s2 = p.z, s3 = p.w;
    const int step = a.step;
#pragma unroll
    for (int y = -2; y <= 2; ++y)
#pragma unroll
        for (int x = -2; x <= 2; ++x) {                                                     // :72-94
            if (x == 0 && y == 0) continue;
            const int sx = cx + x * step, sy = cy + y * step;
            if (sx < 0 || sx >= max_x || sy < 0 || sy >= max_y) continue;
            const size_t sidx = size_t(sy) * W + sx;
            const f4 q = unpack_rgba16f(a.in[sidx]);
            const float kx = (x == 0) ? 0.375f : ((x == 1 || x == -1) ? 0.25f : 0.0625f);
            const float ky = (y == 0) ? 0.375f : ((y == 1 || y == -1) ? 0.25f : 0.0625f);
            const float kernel = kx * ky;                                                   // :62-68 (exact products)
            const f4 nq = unpack_rgba16f(a.normals[sidx]);
            const float wn = pow128(dot3(normal_p, f3{ nq.x, nq.y, nq.z }));                // :44-46
            const float w = (id_p == f2i(nq.w)) ? kernel * wn : 0.0f;                       // :40-42, :87
            const float wx = w * __expf(-(fabsf(p.x - q.x) * inv_s));                       // :88
            const float wy = w * __expf(-(fabsf(p.y - q.y) * inv_a));                       // :89
            sw_s += wx; sw_a += wy;                                                         // :91
            s0 += wx * q.x; s1 += wy * q.y; s2 += (wx * wx) * q.z; s3 += (wy * wy) * q.w;   // :92
        }
    const float rs = __frcp_rn(sw_s), ra = __frcp_rn(sw_a);
    const uint2 texel = pack_rgba16f(s0 * rs, s1 * ra, s2 * (rs * rs), s3 * (ra * ra));     // :97-101
    a.out[idx] = texel;
    if (a.out2) a.out2[idx] = texel;
    if (a.normals_out) a.normals_out[idx] = a.normals[idx];
}

// ---------------------------------------------------------------------------------------------
// K4, the default: comb tiles in LDS, weights in the exponent.
//
// The direct kernel above issues 58 cached loads per pixel and is bound by the CU's L1 line rate, not by HBM.  An a-trous pass with
// step s only ever combines pixels that are s apart, so a workgroup here owns a COMB of rows: output rows y0 + k*s (k = 0..R-1) over
// 64 consecutive columns (different workgroups take the s row phases).  It stages rows y0 + (k-2..R+1)*s over columns
// [x0 - 2s, x0 + 64 + 2s) of both inputs into LDS once -- (R+4)/R * (64+4s)/64 = 1.3x .. 4x the compulsory bytes instead of 25x -- and
// every one of the 24 taps of every output is then two LDS reads (ds_read_b128 + ds_read_b32, conflict free).
//   * All of a tile's global loads are issued up front into registers (32-bit byte offsets on the uniform image bases; a tile whose
//     halo lies inside the image -- nine in ten at 1080p, a workgroup-uniform test -- loads without bounds tests), converted ONCE per
//     texel and parked in LDS: shadow / AO as fp32, the two variances as the halves they are (they enter the sums through
//     v_fma_mix_f32), (nx, ny) as halves for v_dot2, and one word with nz (high half) and the truncated object id (low half: int(w)
//     of svgf_atrous_filter.comp:57,83 as a half; out-of-image texels get a NaN pattern no id equals, so their weight is exactly 0,
//     which is what the shader's `continue` amounts to).  20 bytes per staged texel: eight workgroups per CU at every step size.
//   * The 3x3 variance pre-filter (:17-38, separable) reads three rows per output pixel straight from memory (they are not comb
//     rows) and takes the horizontal neighbours from the adjacent lanes (DPP wave shifts; the tile's two edge lanes load theirs).
//   * The weight of a tap leaves the exponent only once.  The shader's
//       w_ch = k * max(0, n.n')^128 * [id == id'] * exp(-|l - l'| / (4 sqrt(var) + 1e-6))                      (:40-51, 86-89)
//     is evaluated as exp2(L - |l - l'| * inv_ch) with L = 128 * log2(max(0, n.n')) + log2(k), or -inf where the ids differ
//     (exp2(-inf) = 0 and log2(0) = -inf: the zeros the product form gives).  On this chip a packed fp32 instruction occupies the
//     SIMD for 4 cycles, a plain one for 2, a transcendental for 8 (profiles/r3_atrous_knockouts.txt: the kernel's time follows lane
//     operations): one v_log_f32 + one v_fma_f32 stand for the 7 squarings, the kernel constant and the product with the luminance
//     weight.  v_log_f32 / v_exp_f32 are 1-ulp instructions; 128 * log2(x) moves a weight by <= 1e-5 relative, far below the fp16
//     step of the output (tests/test_gpu_svgf.py: <= 2 fp16 steps per dispatch, RMSE <= 1e-4 per frame).
//   * One tile per workgroup, tiles dealt to the XCDs in contiguous bands (xcd_remap: neighbouring tiles share halo texels in one L2;
//     it halved the HBM-side traffic: profiles/traffic.json).
// Same operation order per channel as svgf_atrous_filter.comp:72-94, fp32 sums.  The fused stores: `out2` = a blit of the output
// (hybrid_render_path.cpp:310-315, 322-325), `normals_out` = the blit of the normals image (:321) for the pixels this launch computes.
// ---------------------------------------------------------------------------------------------
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half2_t as_half2(uint32_t u) { return *reinterpret_cast<const half2_t *>(&u); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

constexpr int kTileX = 64;
constexpr uint32_t kInvalidId = 0xffffu;     // NaN half: never equal to a truncated object id

// lane i receives lane i-1's (i+1's) value; lane 0 (63) keeps `edge`
__device__ __forceinline__ float wave_shr1(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

template <int STEP, int R>
__global__ __launch_bounds__(256) void svgf_atrous_tile_kernel(const AtrousArgs a, const uint32_t tiles_x, const uint32_t tiles_total, const Stamps st) {
    vhr_stamp(st);
    constexpr int TW = kTileX + 4 * STEP;            // staged columns
    constexpr int TH = R + 4;                        // staged comb rows
    constexpr int PASS = 256 / TW > 0 ? 256 / TW : 1;   // comb rows staged per pass of the workgroup
    constexpr int NP = (TH + PASS - 1) / PASS;       // passes = texels loaded per thread
    constexpr int NK = (R + 3) / 4;                  // output pixels per thread
    static_assert(TW <= 256, "one thread per staged column");
    // (a plain 16-byte vector type: the tap's texel is then ONE ds_read_b128; as HIP's uint4 struct the compiler fetched it as
    // ds_read2_b64 = two 8-byte reads at a 16-byte lane stride, each a two-way bank conflict)
    typedef uint32_t lds_u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t lds_u2 __attribute__((ext_vector_type(2)));
    __shared__ lds_u4 s_a[TH][TW];                   // shadow, ao, var_s, var_a: fp32 (all four enter the sums through packed fp32 instructions)
    __shared__ lds_u2 s_n[TH][TW];                   // (nx, ny) halves | nz (high half), truncated id as a half (low half)
    const int W = int(a.width), H = int(a.height);
    const int max_x = min(W, int(ceilf(a.display_w))), max_y = min(H, int(ceilf(a.display_h)));
    const int tid = int(threadIdx.x);
    const int tx = tid & 63, ty = tid >> 6;
    const int c = tid % TW, r0 = tid / TW;
    const bool stager = tid < PASS * TW;

    // ---- the tile ----
    const uint32_t t = xcd_remap(blockIdx.x, tiles_total);
    const uint32_t by = t / tiles_x, bx = t - by * tiles_x;
    const int group = int(by) / STEP, phase = int(by) - group * STEP;
    const int x0 = int(a.col_begin) + int(bx) * kTileX;
    const int y0 = int(a.row_begin) + group * (R * STEP) + phase;

    // ---- every global load of the tile, up front: the raw texels this thread stages and the raw variance columns of its pixels ----
    // Addresses are 32-bit byte offsets from the (uniform) image bases -- an image is far below 4 GiB, rows below 2^24 bytes: one
    // v_mad_u32_u24 per texel instead of the two v_mad_u64_u32 + 64-bit shifts and adds a size_t index costs.
    const char *const in_base = reinterpret_cast<const char *>(a.in), *const nm_base = reinterpret_cast<const char *>(a.normals);
    const uint32_t row_bytes = uint32_t(W) * 8u;
    auto texel_offset = [&](int sy, int sx) { return __umul24(uint32_t(sy), row_bytes) + uint32_t(sx) * 8u; };
    uint2 pf_in[NP], pf_nm[NP];
    uint32_t pf_ok = 0;
    uint32_t pv_own[NK][3], pv_edge[NK][3];
    const bool interior = x0 - 2 * STEP >= 0 && x0 + kTileX + 2 * STEP <= max_x && y0 - 2 * STEP >= 0 && y0 + (TH - 3) * STEP < max_y;
#if defined(VHR_ATROUS_KO) && (VHR_ATROUS_KO & 1)
    if (true) {
        pf_ok = (1u << NP) - 1u;
#pragma unroll
        for (int p = 0; p < NP; ++p) { pf_in[p] = make_uint2(0x38003800u + uint32_t(tid), 0x2c002c00u); pf_nm[p] = make_uint2(0x38003800u, 0x3c003800u + uint32_t(p)); }
#pragma unroll
        for (int kq = 0; kq < NK; ++kq)
#pragma unroll
            for (int j = 0; j < 3; ++j) { pv_own[kq][j] = 0x2c002c00u + uint32_t(tx); pv_edge[kq][j] = 0x2c002c00u; }
    } else
#endif
    if (interior) {
        pf_ok = (1u << NP) - 1u;
        if (stager) {
            const uint32_t off0 = texel_offset(y0 + (r0 - 2) * STEP, x0 - 2 * STEP + c);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (p * PASS + r0 < TH) {
                    const uint32_t off = off0 + uint32_t(p * PASS * STEP) * row_bytes;
                    pf_in[p] = *reinterpret_cast<const uint2 *>(in_base + off);
                    pf_nm[p] = *reinterpret_cast<const uint2 *>(nm_base + off);
                }
            }
        }
        const bool edge_lane = tx == 0 || tx == 63;
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) {
            const uint32_t off = texel_offset(y0 + (ty + 4 * kq) * STEP - 1, x0 + tx) + 4u;       // .zw = the two variances
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // (ONE 32-bit offset per load: `base + off + j * row_bytes` is two pointer additions, which the compiler carries out in 64 bits -- six
                // v_lshl_add_u64 per thread, r6)
                pv_own[kq][j] = *reinterpret_cast<const uint32_t *>(in_base + (off + uint32_t(j) * row_bytes));
                pv_edge[kq][j] = 0u;
                if (edge_lane) pv_edge[kq][j] = *reinterpret_cast<const uint32_t *>(in_base + ((tx == 0 ? off - 8u : off + 8u) + uint32_t(j) * row_bytes));
            }
        }
    } else {
        const int sx = x0 - 2 * STEP + c;
        const bool col_ok = stager && sx >= 0 && sx < max_x;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int kk = p * PASS + r0;
            const int sy = y0 + (kk - 2) * STEP;
            pf_in[p] = make_uint2(0u, 0u);
            pf_nm[p] = make_uint2(0u, 0u);
            if (col_ok && kk < TH && sy >= 0 && sy < max_y) {
                const uint32_t off = texel_offset(sy, sx);
                pf_in[p] = *reinterpret_cast<const uint2 *>(in_base + off);
                pf_nm[p] = *reinterpret_cast<const uint2 *>(nm_base + off);
                pf_ok |= 1u << p;
            }
        }
        const int cx = x0 + tx;
        const int ex = tx == 0 ? cx - 1 : cx + 1;
        const bool edge_lane = (tx == 0 || tx == 63) && ex >= 0 && ex < max_x;
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) {
            const int cy = y0 + (ty + 4 * kq) * STEP;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sy = cy + j - 1;
                const bool row_ok = sy >= 0 && sy < max_y;                                      // (the same for every lane of the wave)
                const uint32_t off = texel_offset(sy, cx) + 4u;                                  // .zw = the two variances
                pv_own[kq][j] = (row_ok && cx < max_x) ? *reinterpret_cast<const uint32_t *>(in_base + off) : 0u;
                pv_edge[kq][j] = (row_ok && edge_lane) ? *reinterpret_cast<const uint32_t *>(in_base + (tx == 0 ? off - 8u : off + 8u)) : 0u;
            }
        }
    }

    // ---- the 3x3 variance pre-filter (:17-38): vertical taps in-lane, horizontal via DPP ----
    // (the weights are powers of two, so every product is exact and each sum below rounds exactly where the shader's += rounds; written as
    // FMAs on the half operands -- v_fma_mix_f32 widens them inside the instruction -- 3 instructions per column and channel for the
    // 3 conversions + 3 products + 2 sums of the literal form)
    f2v var_p[NK];
#pragma unroll
    for (int kq = 0; kq < NK; ++kq) {
        auto column = [](const uint32_t (&v)[3]) {
            const half2_t h0 = as_half2(v[0]), h1 = as_half2(v[1]), h2 = as_half2(v[2]);
            float a0, a1;                          // 0.25 h0 (exact), the half widened by the instruction
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(a0) : "v"(v[0]), "v"(0.25f));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a1) : "v"(v[0]), "v"(0.25f));
            (void)h0;
            return f2v{ fmaf(0.25f, float(h2.x), fmaf(0.5f, float(h1.x), a0)), fmaf(0.25f, float(h2.y), fmaf(0.5f, float(h1.y), a1)) };
        };
        const f2v own = column(pv_own[kq]), edge = column(pv_edge[kq]);
        const f2v left = f2v{ wave_shr1(edge.x, own.x), wave_shr1(edge.y, own.y) };
        const f2v right = f2v{ wave_shl1(edge.x, own.x), wave_shl1(edge.y, own.y) };
        var_p[kq] = f2v{ fmaf(0.25f, right.x, fmaf(0.5f, own.x, 0.25f * left.x)), fmaf(0.25f, right.y, fmaf(0.5f, own.y, 0.25f * left.y)) };
    }
    // ---- registers -> LDS (converted once per texel) ----
    if (stager) {
        auto stage = [&](const bool all_loaded) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int kk = p * PASS + r0;
                if (kk < TH) {
                    uint4 va = make_uint4(0u, 0u, 0u, 0u);
                    uint32_t nxy = 0u, ni = kInvalidId;
                    if (all_loaded || (pf_ok & (1u << p))) {
                        const uint2 vin = pf_in[p], n = pf_nm[p];
                        const float2 xy = unpack_rg16f(vin.x), zw = unpack_rg16f(vin.y);
                        va = make_uint4(__float_as_uint(xy.x), __float_as_uint(xy.y), __float_as_uint(zw.x), __float_as_uint(zw.y));
                        nxy = n.x;
                        // int(w) as a half (:57, :83).  The taps compare ids as HALVES (v_cmp_eq_f16): -0 == +0 like int(-0.x) == 0, and a
                        // NaN id becomes 0 here, which is what int(NaN) is in the oracle (decision viii)
                        _Float16 idh = __builtin_truncf16(as_half2(n.y).y);
                        idh = idh == idh ? idh : _Float16(0.0f);
                        // nz (the low half of n.y) into the high half, the id's bits into the low half: one byte permute
                        ni = __builtin_amdgcn_perm(uint32_t(*reinterpret_cast<const uint16_t *>(&idh)), n.y, 0x01000504u);
                    }
                    s_a[kk][c] = lds_u4{ va.x, va.y, va.z, va.w };
                    s_n[kk][c] = lds_u2{ nxy, ni };
                }
            }
        };
        if (pf_ok == (1u << NP) - 1u) stage(true); else stage(false);       // (an interior tile: workgroup-uniform)
    }
    __syncthreads();

    const int cx = x0 + tx;
#pragma unroll
    for (int kq = 0; kq < NK; ++kq) {
        const int k = ty + 4 * kq;
        const int cy = y0 + k * STEP;
        if (k >= R || uint32_t(cx) >= a.limit_x || uint32_t(cy) >= a.row_end || uint32_t(cy) >= a.limit_y) continue;
        const lds_u4 pa = s_a[k + 2][tx + 2 * STEP];
        const f2v p_xy = f2v{ u2f(pa.x), u2f(pa.y) };
        const lds_u2 pn = s_n[k + 2][tx + 2 * STEP];
        // The centre's normal at HALF length: n.n' then arrives as dd / 2 <= 0.501, and the dot product's own output clamp to [0, 1] (v_fma_mix_f32's) IS
        // the shader's max(0, .) (:46) -- no v_max per tap; the factor comes back as +128 in the tap's constant below.  Preconditions, both met by the
        // G-buffer's unit normals (gbuf.frag:43 normalises) and outside the bit-exact contract anyway (this kernel's parity is a tolerance,
        // tests/test_gpu_svgf.py): the halving of a HALF is exact only for components >= 2^-13 (smaller ones round in the half multiply: an
        // error of <= 2^-25 in a dot product compared at 2^-11), and the clamp replaces max(0, .) only while n.n' <= 2, i.e. |n| |n'| <= 2.
        const half2_t np_xy = as_half2(pn.x) * half2_t{ _Float16(0.5f), _Float16(0.5f) };
        const float np_z = 0.5f * float(as_half2(pn.y).y);
        const _Float16 idp = as_half2(pn.y).x;
        // 1 / (4 sqrt(var) + 1e-6) (:48-50), times log2(e): the luminance weight is an exp2
        const f2v inv = f2v{ __builtin_amdgcn_rcpf(4.0f * __builtin_amdgcn_sqrtf(var_p[kq].x) + 1e-6f) * 1.44269504088896341f,
                             __builtin_amdgcn_rcpf(4.0f * __builtin_amdgcn_sqrtf(var_p[kq].y) + 1e-6f) * 1.44269504088896341f };
        f2v sw = f2v{ 1.0f, 1.0f };                                                     // :70-71
        f2v s01 = p_xy;
        f2v s23 = f2v{ u2f(pa.z), u2f(pa.w) };
        // (kept in a vector register on purpose: a packed instruction takes ONE scalar operand, and that slot is better spent on the per-pair
        // constants log2 k below, which would otherwise be moved into vector registers pair by pair)
        float k128 = 128.0f;
        asm volatile("" : "+v"(k128));
#pragma unroll
        for (int g = 0; g < 6; ++g) {                                                   // :72-94, four taps per trip
            lds_u4 qa[4];
            lds_u2 qn[4];
            float L[4];
            float lg[4];
            bool same[4];
            f2v kc[2];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int tp = 4 * g + h, idx = tp < 12 ? tp : tp + 1;                  // skip the centre (:77)
                const int y = idx / 5 - 2, x = idx % 5 - 2;
                const int row = k + 2 + y, col = tx + 2 * STEP + x * STEP;
#if defined(VHR_ATROUS_KO) && (VHR_ATROUS_KO & 4)
                // (knock-out: no LDS read in the taps -- values the compiler cannot fold, derived from the centre texel and the tap index)
                qa[h] = lds_u4{ pa.x + uint32_t(tp) * 8192u, pa.y ^ (uint32_t(tp) << 12), pa.z + uint32_t(tp), pa.w };
                qn[h] = lds_u2{ pn.x ^ (uint32_t(tp) << 3), pn.y ^ (uint32_t(tp) << 19) };
                (void)row; (void)col;
#else
                qa[h] = s_a[row][col];
                qn[h] = s_n[row][col];
#endif
                // log2 of the B3 spline factors (:62-68): 3/8, 1/4, 1/16
                const float lx = (x == 0) ? -1.41503749927884381f : ((x == 1 || x == -1) ? -2.0f : -4.0f);
                const float ly = (y == 0) ? -1.41503749927884381f : ((y == 1 || y == -1) ? -2.0f : -4.0f);
                kc[h >> 1][h & 1] = lx + ly + 128.0f;                                   // (+128 = -128 log2(1/2): the half-length normal)
                same[h] = as_half2(qn[h].y).x == idp;                                   // :40-42 (out of the image: a NaN id)
            }
            // :44-46 for the four taps: (nx nx' + ny ny') + nz nz' -- the shader's order --, the second instruction widening the half nz' itself and
            // clamping to [0, 1]: max(0, .) for free (v_dot2_f32_f16's own clamp bit does not act on gfx950: scratch/clamp_probe.hip).  ONE asm
            // statement: a dot product's result needs three wait states before a vector instruction may read it -- the compiler inserts them
            // (s_nop) for its own instructions and cannot for an asm statement's operands (the clamped FMA as a statement of its own read garbage)
            // -- and here the other three taps' instructions ARE the wait states.  The VOP3P dot takes its addend 0 inline (the compiler's own
            // choice, v_dot2c, needs a v_mov per tap to zero its accumulator).
            asm("v_dot2_f32_f16 %0, %4, %5, 0\n\tv_dot2_f32_f16 %1, %4, %6, 0\n\tv_dot2_f32_f16 %2, %4, %7, 0\n\tv_dot2_f32_f16 %3, %4, %8, 0\n\t"
                "v_fma_mix_f32 %0, %9, %13, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\tv_fma_mix_f32 %1, %10, %13, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\t"
                "v_fma_mix_f32 %2, %11, %13, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\tv_fma_mix_f32 %3, %12, %13, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                : "=&v"(lg[0]), "=&v"(lg[1]), "=&v"(lg[2]), "=&v"(lg[3])
                : "v"(np_xy), "v"(qn[0].x), "v"(qn[1].x), "v"(qn[2].x), "v"(qn[3].x), "v"(qn[0].y), "v"(qn[1].y), "v"(qn[2].y), "v"(qn[3].y), "v"(np_z));
#pragma unroll
            for (int h = 0; h < 4; ++h) lg[h] = __builtin_amdgcn_logf(lg[h]);              // log2; -inf at 0
            // 128 log2(n.n') + log2 k for two taps per instruction (a packed fp32 FMA occupies the SIMD like a plain one)
            const f2v L01 = __builtin_elementwise_fma(f2v{ lg[0], lg[1] }, f2v{ k128, k128 }, kc[0]);
            const f2v L23 = __builtin_elementwise_fma(f2v{ lg[2], lg[3] }, f2v{ k128, k128 }, kc[1]);
            // (r6, measured and not kept: skipping these four selects where a wave sees ONE object in all four taps -- the masks and-ed on the scalar
            // unit, one test per group -- makes the launch 3-6 % SLOWER: the branch costs the unrolled groups their interleaving; profiles/r6_atrous.txt)
            L[0] = same[0] ? L01.x : -__builtin_inff();                                 // :87 in the exponent
            L[1] = same[1] ? L01.y : -__builtin_inff();
            L[2] = same[2] ? L23.x : -__builtin_inff();
            L[3] = same[3] ? L23.y : -__builtin_inff();
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f2v q_xy = f2v{ u2f(qa[h].x), u2f(qa[h].y) };
                const f2v dl = p_xy - q_xy;
                const f2v w2 = f2v{ __builtin_amdgcn_exp2f(fmaf(-fabsf(dl.x), inv.x, L[h])),
                                    __builtin_amdgcn_exp2f(fmaf(-fabsf(dl.y), inv.y, L[h])) };          // :88-89 with :87 in the exponent
                sw += w2;                                                               // :91
                s01 = __builtin_elementwise_fma(w2, q_xy, s01);                         // :92
                s23 = __builtin_elementwise_fma(w2 * w2, f2v{ u2f(qa[h].z), u2f(qa[h].w) }, s23);
            }
        }
        const f2v r = f2v{ __builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y) };
        const f2v o01 = s01 * r, o23 = s23 * (r * r);                                     // :97-101 (three packed products)
        const uint2 texel = pack_rgba16f(o01.x, o01.y, o23.x, o23.y);
        const uint32_t out_off = texel_offset(cy, cx);
#if defined(VHR_ATROUS_KO) && (VHR_ATROUS_KO & 2)
        if (texel.x == 0x12345678u && texel.y == 0x9abcdef0u)       // (practically never: the arithmetic stays alive, nothing is stored)
#endif
        *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(a.out) + out_off) = texel;
        // The fused blits' stores (and the one load) with the image base in scalar registers and the 32-bit texel offset in a vector register, spelled
        // out: for these nullable pointers the compiler builds a 64-bit vector address per access (a v_lshl_add_u64 each) where the main store
        // above gets the scalar-base form.  Last instructions of the thread: nothing behind them depends on the counters the compiler tracks.
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        const u2v tv = u2v{ texel.x, texel.y };
        if (a.out2) asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(out_off), "v"(tv), "s"(a.out2) : "memory");
        if (a.normals_out) {
            u2v nv;
            asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(nv) : "v"(out_off), "s"(a.normals) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(out_off), "v"(nv), "s"(a.normals_out) : "memory");
        }
    }
}

// 4-row tiles (one pixel per thread, twice the workgroups, 2/3 of the LDS each) or 8-row tiles (two pixels per thread, a quarter less
// staging per pixel).  Thin launches -- the screen tile of one GPU out of 4 or 8 -- need the smaller tile to fill the chip, and so does
// the whole 1080p frame (214.5 -> 209.2 us for the five launches, r2), while at 4K (63 8-row tiles per CU) the 8-row tile stays ahead
// (154.7 vs 159.2 us per launch).  "atrous_small_tiles" -1 (auto): 4-row tiles below 32 8-row tiles per CU; 0 never, 1 always.
template <int STEP, int R>
static void launch_atrous_tiles(vhr_context *ctx, const AtrousArgs &a) {
    const uint32_t rows = a.row_end - a.row_begin;
    const uint32_t groups = (rows + R * STEP - 1) / (R * STEP);
    const uint32_t tiles_x = (a.limit_x - a.col_begin + kTileX - 1) / kTileX, tiles_total = tiles_x * groups * STEP;
    launch(ctx, (svgf_atrous_tile_kernel<STEP, R>), dim3(tiles_total), dim3(256), 0, a, tiles_x, tiles_total);
}
template <int STEP>
static void launch_atrous_tiles_auto(vhr_context *ctx, const AtrousArgs &a) {
    const uint32_t rows = a.row_end - a.row_begin;
    const uint32_t tiles8 = ((a.limit_x - a.col_begin + kTileX - 1) / kTileX) * ((rows + 8 * STEP - 1) / (8 * STEP)) * STEP;
    const int small = ctx->options[kOptAtrousSmallTiles];
    if (small == 1 || (small < 0 && tiles8 < 32u * uint32_t(ctx->cu_count))) launch_atrous_tiles<STEP, 4>(ctx, a);
    else launch_atrous_tiles<STEP, 8>(ctx, a);
}

int launch_svgf_atrous(vhr_context *ctx, const vhr_per_frame_data &pfd, const Image &normals, const Image &in, Image &out,
                       int32_t step, uint32_t x_groups, uint32_t y_groups) {
    const uint32_t W = normals.width, H = normals.height;
    if (in.width != W || in.height != H || out.width != W || out.height != H)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf_atrous_filter.comp: image extents differ");
    if (normals.format != VHR_FORMAT_R16G16B16A16_SFLOAT || in.format != VHR_FORMAT_R16G16B16A16_SFLOAT ||
        out.format != VHR_FORMAT_R16G16B16A16_SFLOAT)
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf_atrous_filter.comp: images must be R16G16B16A16_SFLOAT");
    if (in.ptr == out.ptr) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf_atrous_filter.comp: ping-pong images alias");
    if (step < 1 || step > 4096) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "svgf_atrous_filter.comp: atrous_step out of range");
    AtrousArgs a;
    a.normals = static_cast<const uint2 *>(normals.ptr);
    a.in = static_cast<const uint2 *>(in.ptr);
    a.out = static_cast<uint2 *>(out.ptr);
    a.width = W; a.height = H;
    a.limit_x = uint32_t(std::min<uint64_t>(W, uint64_t(x_groups) * 8));
    a.limit_y = uint32_t(std::min<uint64_t>(H, uint64_t(y_groups) * 8));
    // Strips: an iteration with step s reads +-2s rows, so after the doubling schedule 1, 2, ..., s (hybrid_render_path.cpp:
    // 299-319) only the rows within overlap - (4s - 2) of the strip can still be valid -- and only those are needed by the
    // iterations that follow.  With "strip_shrink_overlap" the launch computes just them (tiling.atrous_output_extent).
    uint32_t extend = ctx->overlap;
    if (ctx->options[kOptShrinkOverlap]) {
        const uint64_t consumed = 4ull * uint64_t(step) - 2ull;
        extend = consumed >= extend ? 0u : uint32_t(extend - consumed);
    }
    strip_rows(ctx, H, extend, a.row_begin, a.row_end);
    {
        uint32_t c1;
        strip_cols(ctx, W, extend, a.col_begin, c1);          // screen tiles: the same margin on the columns
        a.limit_x = std::min(a.limit_x, c1);
    }
    a.step = step;
    a.display_w = pfd.display_size[0];
    a.display_h = pfd.display_size[1];
    a.out2 = nullptr;
    a.normals_out = nullptr;
    SvgfCmd cmd{};
    cmd.kind = SvgfCmd::Atrous;
    cmd.a = a;
    if (ctx->recording) { ctx->recorded.push_back(cmd); return VHR_OK; }
    return issue_cmd(ctx, cmd);
}

// ---------------------------------------------------------------------------------------------
// Counter calibration: a streaming read of a known byte count with the access width the SVGF kernels use
// (8 B per lane for RGBA16F, 4 B for RG16F, 16 B for reference).  rocprofv3's FETCH_SIZE is known to under-report
// wide streaming reads on gfx950 (MI355X_MICROARCH.md, HBM section); tools/profile_traffic.sh runs this kernel
// under --pmc FETCH_SIZE to obtain the correction factor for OUR access pattern before pricing the a-trous traffic.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void calibration_read_kernel(const T *src, size_t count, uint32_t *sink, const Stamps st) {
    vhr_stamp(st);
    uint32_t acc = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
#pragma unroll
        for (int k = 0; k < int(sizeof(T) / 4); ++k) acc ^= w[k];
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;        // practically never true: keeps the loads alive
}

int launch_calibration_read(vhr_context *ctx, const Image &img, uint32_t bytes_per_lane, uint32_t *sink) {
    const size_t bytes = img.bytes();
    const dim3 grid(2048);
    if (bytes_per_lane == 4) launch(ctx, calibration_read_kernel<uint32_t>, grid, dim3(256), 0, static_cast<const uint32_t *>(img.ptr), bytes / 4, sink);
    else if (bytes_per_lane == 8) launch(ctx, calibration_read_kernel<uint2>, grid, dim3(256), 0, static_cast<const uint2 *>(img.ptr), bytes / 8, sink);
    else if (bytes_per_lane == 16) launch(ctx, calibration_read_kernel<uint4>, grid, dim3(256), 0, static_cast<const uint4 *>(img.ptr), bytes / 16, sink);
    else return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "calibration: bytes_per_lane must be 4, 8 or 16");
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "calibration kernel launch failed");
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// K5: same-extent, same-format VK_FILTER_NEAREST blit == copy (compute_execution_context.cpp:178-211)
// ---------------------------------------------------------------------------------------------
// 16 bytes per lane, 4 independent loads in flight per lane.  A plain kernel instead of hipMemcpyAsync: same bandwidth
// on a full 1080p image, but the runtime's copy path costs ~7 us however small the copy is, which is what the row
// strips of a multi-GPU run would pay three times per frame.
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t count, const Stamps st) {
    vhr_stamp(st);
    const size_t stride = size_t(gridDim.x) * 256;
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < count; i += stride) dst[i] = src[i];
}

int copy_image_rows(vhr_context *ctx, const Image &src, Image &dst) {
    if (src.width != dst.width || src.height != dst.height)          // asserts at compute_execution_context.cpp:179-180
        return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "BlitImage: extents differ");
    if (src.bpp != dst.bpp) return ctx->fail(VHR_ERROR_INVALID_ARGUMENT, "BlitImage: formats of different texel size are not supported");
    uint32_t r0, r1, c0, c1;
    strip_rows(ctx, src.height, ctx->halo, r0, r1);
    strip_cols(ctx, src.width, ctx->halo_cols, c0, c1);
    if (r1 <= r0 || c1 <= c0 || src.ptr == dst.ptr) return VHR_OK;
    const bool all_cols = c0 == 0 && c1 == src.width;
    const size_t row = size_t(src.width) * src.bpp, offset = r0 * row + size_t(c0) * src.bpp, bytes = (r1 - r0) * row;
    SvgfCmd cmd{};
    cmd.kind = SvgfCmd::Copy;
    cmd.copy_src = static_cast<const char *>(src.ptr) + offset;
    cmd.copy_dst = static_cast<char *>(dst.ptr) + offset;
    cmd.copy_bytes = bytes;
    cmd.copy_rows = all_cols ? 0u : r1 - r0;                        // a column range (screen tiles): row pieces
    cmd.copy_pitch = row;
    cmd.copy_row_bytes = size_t(c1 - c0) * src.bpp;
    cmd.src_base = src.ptr;
    cmd.dst_base = dst.ptr;
    if (!ctx->recording) return issue_cmd(ctx, cmd);
    // ---- fusion: the blit becomes a second store of the recorded a-trous dispatch that produced its source ----
    if (ctx->options[kOptFuseBlits] && src.bpp == 8) {
        const bool whole = ctx->row_begin == 0 && ctx->row_end >= src.height && all_cols;
        auto touches = [](const SvgfCmd &c, const void *p, bool writes_only) {
            switch (c.kind) {
                case SvgfCmd::Temporal:
                    return c.t.integrated_out == p || c.t.moments_out == p ||
                           (!writes_only && (c.t.normals == p || c.t.motion == p || c.t.prev_normals == p || c.t.history == p || c.t.raytraced == p || c.t.moments_in == p));
                case SvgfCmd::Atrous:
                    return c.a.out == p || c.a.out2 == p || c.a.normals_out == p || (!writes_only && (c.a.normals == p || c.a.in == p));
                default:
                    return c.dst_base == p || (!writes_only && c.src_base == p);
            }
        };
        for (size_t k = ctx->recorded.size(); k-- > 0;) {                 // newest first
            SvgfCmd &w = ctx->recorded[k];
            if (touches(w, src.ptr, true)) {
                // w wrote the source last.  Fusable iff it is an a-trous launch over whole rows that the blit covers (on a
                // strip the blit's extra rows hold nothing valid and are refilled by the neighbour exchange, see vhr_set_strip)
                if (w.kind == SvgfCmd::Atrous && w.a.out == src.ptr && !w.a.out2 && w.a.col_begin >= c0 && w.a.limit_x <= c1 && !touches(w, dst.ptr, false) &&
                    w.a.row_begin >= r0 && w.a.row_end <= r1 &&
                    (!whole || (w.a.row_begin == r0 && w.a.row_end == r1 && w.a.limit_y >= r1 && w.a.col_begin == 0 && w.a.limit_x == w.a.width))) {
                    w.a.out2 = static_cast<uint2 *>(dst.ptr);
                    return VHR_OK;
                }
                break;
            }
            if (touches(w, dst.ptr, false)) break;                         // a later command uses the destination: the copy cannot move before it
        }
    }
    // ---- ... or a store of the recorded a-trous dispatch that READS its source as its normals / ids image (r3c): the launch copies the
    // texel of every pixel it computes (hybrid_render_path.cpp:319: "World Space Normals and Object IDs" -> previous-frame normals; a
    // copy kernel of its own cost 6 us + a launch gap for 33 MB an a-trous launch fetches anyway).  Whole-image work only; the earliest
    // dispatch behind the last command that touches the destination takes it, so that the pass's unread dispatch stays unread.
    if (ctx->options[kOptFuseBlits] && src.bpp == 8 && ctx->row_begin == 0 && ctx->row_end >= src.height && all_cols) {
        auto uses = [](const SvgfCmd &c, const void *p, bool writes_only) {
            switch (c.kind) {
                case SvgfCmd::Temporal:
                    return c.t.integrated_out == p || c.t.moments_out == p ||
                           (!writes_only && (c.t.normals == p || c.t.motion == p || c.t.prev_normals == p || c.t.history == p || c.t.raytraced == p || c.t.moments_in == p));
                case SvgfCmd::Atrous:
                    return c.a.out == p || c.a.out2 == p || c.a.normals_out == p || (!writes_only && (c.a.normals == p || c.a.in == p));
                default:
                    return c.dst_base == p || (!writes_only && c.src_base == p);
            }
        };
        size_t first = ctx->recorded.size();           // commands [first, end) neither touch the destination nor write the source
        while (first > 0 && !uses(ctx->recorded[first - 1], dst.ptr, false) && !uses(ctx->recorded[first - 1], src.ptr, true)) --first;
        for (size_t k = first; k < ctx->recorded.size(); ++k) {
            SvgfCmd &w = ctx->recorded[k];
            if (w.kind == SvgfCmd::Atrous && w.a.normals == src.ptr && !w.a.normals_out && w.a.row_begin == 0 && w.a.row_end >= src.height && w.a.limit_y >= src.height &&
                w.a.col_begin == 0 && w.a.limit_x == w.a.width && w.a.width == src.width && w.a.height == src.height) {
                w.a.normals_out = static_cast<uint2 *>(dst.ptr);
                return VHR_OK;
            }
        }
    }
    ctx->recorded.push_back(cmd);
    return VHR_OK;
}

// a column range of rows (screen tiles): `rows` pieces of `words` 4-byte words, `pitch_words` apart in both images
__global__ __launch_bounds__(256) void copy_rect_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t words, uint32_t rows, size_t pitch_words, const Stamps st) {
    vhr_stamp(st);
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= words) return;
    for (uint32_t y = blockIdx.y; y < rows; y += gridDim.y) dst[size_t(y) * pitch_words + x] = src[size_t(y) * pitch_words + x];
}

static int issue_copy(vhr_context *ctx, const SvgfCmd &cmd) {
    const char *s8 = cmd.copy_src;
    char *d8 = cmd.copy_dst;
    const size_t bytes = cmd.copy_bytes;
    hipError_t copy_rc = hipSuccess;
    ctx->time_begin(kKernelCopy);
    if (cmd.copy_rows) {                                             // texels are 4 or 8 bytes: whole 4-byte words at 4-byte alignment
        const uint32_t words = uint32_t(cmd.copy_row_bytes / 4);
        launch(ctx, copy_rect_kernel, dim3((words + 255) / 256, std::min<uint32_t>(cmd.copy_rows, 1024u)), dim3(256), 0,
               reinterpret_cast<const uint32_t *>(s8), reinterpret_cast<uint32_t *>(d8), words, cmd.copy_rows, cmd.copy_pitch / 4);
        copy_rc = hipGetLastError();
        ctx->time_end(kKernelCopy);
        if (copy_rc != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "BlitImage: device copy failed");
        return VHR_OK;
    }
    if (((reinterpret_cast<uintptr_t>(s8) | reinterpret_cast<uintptr_t>(d8) | bytes) & 15u) == 0) {
        const size_t count = bytes / 16;
        const uint32_t blocks = uint32_t(std::min<size_t>((count + 1023) / 1024, size_t(ctx->cu_count) * 8));
        launch(ctx, copy_rows_kernel, dim3(std::max(1u, blocks)), dim3(256), 0, reinterpret_cast<const uint4 *>(s8),
                           reinterpret_cast<uint4 *>(d8), count);
        copy_rc = hipGetLastError();
    } else {
        copy_rc = hipMemcpyAsync(d8, s8, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    }
    ctx->time_end(kKernelCopy);
    if (copy_rc != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "BlitImage: device copy failed");
    return VHR_OK;
}

static int issue_temporal(vhr_context *ctx, const TemporalArgs &a) {
    if (a.row_end > a.row_begin && a.limit_x > a.col_begin && a.limit_y) {
        ctx->time_begin(kKernelTemporal);
        auto go = [&](auto kern, uint32_t bx, uint32_t by) {
            launch(ctx, kern, dim3((a.limit_x - a.col_begin + bx - 1) / bx, (a.row_end - a.row_begin + by - 1) / by), dim3(bx, by), 0, a);
        };
        // 32x8-pixel blocks (r3d): two rows of 32 pixels per wave: 26.6 -> 25.7 us at 1080p, 95.2 -> 90.1 at 4K (64x8, 32x16, 64x16, 16x16, 128x4,
        // 32x4, 32x6, 32x12, 16x8 measured too: scratch/ab_temporal.py; the counter traffic does not fall, the rate it moves at rises).
        go(svgf_temporal_kernel<32, 8>, 32, 8);
        ctx->time_end(kKernelTemporal);
        if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "svgf temporal kernel launch failed");
    }
    return VHR_OK;
}

// ---------------------------------------------------------------------------------------------
// Issue of recorded / immediate SVGF commands
// ---------------------------------------------------------------------------------------------
static int issue_atrous(vhr_context *ctx, const AtrousArgs &a) {
    const int32_t step = a.step;
    if (a.row_end <= a.row_begin || a.limit_x <= a.col_begin || !a.limit_y) return VHR_OK;
    ctx->time_begin(ctx->async_atrous ? kKernelAtrousAsync : kKernelAtrous);
    bool tiled = ctx->options[kOptAtrousVariant] != 0;          // 0: the literal form of the shader (svgf_atrous_kernel), every step size
    if (tiled) {
        switch (step) {                                          // the step sizes of the reference's schedule (hybrid_render_path.cpp:299-319)
            case 1: launch_atrous_tiles_auto<1>(ctx, a); break;
            case 2: launch_atrous_tiles_auto<2>(ctx, a); break;
            case 4: launch_atrous_tiles_auto<4>(ctx, a); break;
            case 8: launch_atrous_tiles_auto<8>(ctx, a); break;
            case 16: launch_atrous_tiles_auto<16>(ctx, a); break;
            default: tiled = false; break;                       // any other step: the literal kernel
        }
    }
    if (!tiled) {
        // (the literal kernel computes the rows from column 0: a superset of a screen tile's columns)
        const dim3 grid((a.limit_x + kSvgfBlockX - 1) / kSvgfBlockX, (a.row_end - a.row_begin + kSvgfBlockY - 1) / kSvgfBlockY);
        launch(ctx, svgf_atrous_kernel, grid, dim3(kSvgfBlockX, kSvgfBlockY), 0, a);
    }
    ctx->time_end(kKernelAtrous);
    if (hipGetLastError() != hipSuccess) return ctx->fail(VHR_ERROR_DEVICE, "svgf atrous kernel launch failed");
    return VHR_OK;
}

static int issue_cmd(vhr_context *ctx, const SvgfCmd &cmd) {
    switch (cmd.kind) {
        case SvgfCmd::Temporal: return issue_temporal(ctx, cmd.t);
        case SvgfCmd::Atrous: return issue_atrous(ctx, cmd.a);
        default: return issue_copy(ctx, cmd);
    }
}

// the end stamp of a pass that has no kernel of the library behind it in the frame (see vhr::Stamps)
__global__ void stamp_kernel(const Stamps st) {
    vhr_stamp(st);
}
void launch_stamp(vhr_context *ctx) {
    if (!ctx->pending_end) return;
    vhr::PassDescription *const pass = ctx->cur_pass;
    ctx->cur_pass = nullptr;
    launch(ctx, stamp_kernel, dim3(1), dim3(1), 0);
    ctx->cur_pass = pass;
}

// "svgf_elide_unread" (opt-in): an a-trous dispatch whose output image no later command of the same pass reads, and which is not
// published through a fused blit, is not launched.  That is exactly the reference's fifth a-trous iteration: its host loop
// (hybrid_render_path.cpp:299-328) publishes iteration 3's image and lets iteration 4's be overwritten by the next frame's first
// iteration before anything reads it -- 7 % of the frame for nothing.  Off by default because the library cannot
// know that the caller never looks at that storage image between frames; with the option on it is left holding older contents.
// Everything the pass publishes -- Denoised, history, moments, previous normals -- is bit-identical (tests/test_gpu_svgf.py).
static bool dead_atrous(const std::vector<SvgfCmd> &rec, size_t k) {
    const SvgfCmd &w = rec[k];
    if (w.kind != SvgfCmd::Atrous || w.a.out2 || w.a.normals_out) return false;
    const void *img = w.a.out;
    for (size_t j = k + 1; j < rec.size(); ++j) {
        const SvgfCmd &c = rec[j];
        switch (c.kind) {
            case SvgfCmd::Temporal:
                if (c.t.normals == img || c.t.motion == img || c.t.prev_normals == img || c.t.history == img || c.t.raytraced == img || c.t.moments_in == img) return false;
                break;
            case SvgfCmd::Atrous:
                if (c.a.normals == img || c.a.in == img) return false;
                break;
            default:
                if (c.src_base == img) return false;
                break;
        }
    }
    return true;
}

// "svgf_async_unread" (default 1): the same dead dispatch, when the option above leaves it in, does not have to sit on the frame's
// critical path either.  Nothing in its pass reads what it writes, so it is issued LAST, on the context's side stream, ordered
// behind the pass's other commands by an event; the caller's stream goes on with the next frame (the ray-tracing kernel, which is
// bound by memory latency and leaves vector issue slots free, while this kernel is bound by vector issue) and waits for the side
// stream's event before the next command of the library that could touch the images involved (join_side: the next compute pass,
// image uploads / downloads / queries, vhr_synchronize).  Conditions, all checked here, else the dispatch stays where it was
// recorded: one stream per frame ("frames_in_flight" 1), no later command of the pass writes the dispatch's input or output, and the
// G-buffer normals it reads are available as a copy, made by the same pass, of everything it reads of them (with screen tiles: the tile
// grown by the taps' reach; hybrid_render_path.cpp:319: "World Space Normals and Object IDs" -> previous-frame normals) -- the G-buffer itself is
// rewritten by the next frame's first pass, the copy only by the next SVGF pass, which joins first.  Same kernels on the same
// inputs: every image, the dead dispatch's own output included, is bit-identical (tests/test_gpu_svgf.py).
static bool async_candidate(vhr_context *ctx, const std::vector<SvgfCmd> &rec, size_t k, const void *&normals_copy, const void *&input_copy) {
    const SvgfCmd &w = rec[k];
    // Rectangles (half-open, pixels): what the dispatch READS of its two inputs -- its own rectangle grown by the taps' reach, 2 x step,
    // and for `in` by one more (the 3x3 variance pre-filter) -- against what a copy holds.  With screen tiles / row strips the pass's
    // blits cover the tile grown by the halos, its fused stores the launch's own (shrunk) rectangle: a copy counts only if it holds
    // everything the dispatch reads.
    struct Rect { int x0, x1, y0, y1; };
    const int W = int(w.a.width), H = int(w.a.height), reach = 2 * w.a.step;
    auto clip = [](int v, int hi) { return std::max(0, std::min(v, hi)); };
    auto rect_of = [&](const AtrousArgs &a) { return Rect{ int(a.col_begin), int(std::min(a.limit_x, a.width)), int(a.row_begin), int(std::min(std::min(a.row_end, a.limit_y), a.height)) }; };
    auto rect_of_copy = [&](const SvgfCmd &c) {
        const size_t off = size_t(c.copy_src - static_cast<const char *>(c.src_base)), pitch = c.copy_pitch ? c.copy_pitch : size_t(W) * sizeof(uint2);
        const int y0 = int(off / pitch), x0 = int((off % pitch) / sizeof(uint2));
        if (c.copy_rows == 0) return Rect{ x0 == 0 ? 0 : W, W, y0, y0 + int(c.copy_bytes / pitch) };      // whole rows (x0 != 0 cannot happen: an empty rectangle then)
        return Rect{ x0, x0 + int(c.copy_row_bytes / sizeof(uint2)), y0, y0 + int(c.copy_rows) };
    };
    auto covers = [](const Rect &have, const Rect &need) { return have.x0 <= need.x0 && have.x1 >= need.x1 && have.y0 <= need.y0 && have.y1 >= need.y1; };
    const Rect own = rect_of(w.a);
    if (own.x1 <= own.x0 || own.y1 <= own.y0) return false;
    // The event that orders the side stream and the wait for it cost the caller's stream 5-8 us; a dispatch has to be worth that.  Measured
    // on one GPU (scratch/strip_time.py, 1080p): the whole frame -12 us, half of it (N = 2) -5 us, a quarter 0, an eighth (8.5 us per
    // launch) +5 us.  900 k pixels = the half-frame tile of 1080p and the eighth of a 4K frame.
    if (ctx->options[kOptSvgfAsyncUnread] != 2 && size_t(own.x1 - own.x0) * size_t(own.y1 - own.y0) < 900000u) return false;      // (2: whatever the size)
    const Rect need_n{ clip(own.x0 - reach, W), clip(own.x1 + reach, W), clip(own.y0 - reach, H), clip(own.y1 + reach, H) };
    const Rect need_in{ clip(need_n.x0 - 1, W), clip(need_n.x1 + 1, W), clip(need_n.y0 - 1, H), clip(need_n.y1 + 1, H) };
    // The dispatch's input may exist twice: the a-trous dispatch that wrote it may have stored the same texels into a second image
    // (a fused blit: hybrid_render_path.cpp:320-323 publishes iteration 3's image as "Denoised ...").  Reading that copy instead moves
    // the point where the caller's stream has to wait from the next frame's svgf.comp (which overwrites the ping-pong image) to its
    // first a-trous dispatch (which overwrites this dispatch's output).  The same for the normals: an earlier dispatch's `normals_out`.
    input_copy = nullptr;
    const void *fused_normals = nullptr;
    for (size_t j = 0; j < k; ++j) {
        const SvgfCmd &c = rec[j];
        if (c.kind == SvgfCmd::Atrous && c.a.normals == w.a.normals && c.a.normals_out && covers(rect_of(c.a), need_n)) fused_normals = c.a.normals_out;
        else if ((c.kind == SvgfCmd::Atrous && (c.a.out == fused_normals || c.a.out2 == fused_normals)) || (c.kind == SvgfCmd::Copy && c.dst_base == fused_normals) ||
                 (c.kind == SvgfCmd::Temporal && c.t.integrated_out == fused_normals)) fused_normals = nullptr;
        if (c.kind == SvgfCmd::Atrous && c.a.out == w.a.in && c.a.out2 && covers(rect_of(c.a), need_in)) input_copy = c.a.out2;
        else if (c.kind == SvgfCmd::Atrous && (c.a.out == input_copy || c.a.out2 == input_copy)) input_copy = nullptr;
        else if (c.kind == SvgfCmd::Copy && c.dst_base == input_copy) input_copy = nullptr;
        else if (c.kind == SvgfCmd::Temporal && c.t.integrated_out == input_copy) input_copy = nullptr;
    }
    normals_copy = fused_normals;
    for (size_t j = k + 1; j < rec.size(); ++j) {
        const SvgfCmd &c = rec[j];
        switch (c.kind) {
            case SvgfCmd::Temporal:
                if (c.t.integrated_out == w.a.in || c.t.integrated_out == w.a.out || c.t.moments_out == static_cast<const void *>(w.a.in)) return false;
                break;
            case SvgfCmd::Atrous:
                if (c.a.out == w.a.in || c.a.out == w.a.out || c.a.out2 == w.a.in || c.a.out2 == w.a.out) return false;
                if (c.a.out == input_copy || c.a.out2 == input_copy) input_copy = nullptr;
                if (c.a.out == normals_copy || c.a.out2 == normals_copy || c.a.normals_out == normals_copy) normals_copy = nullptr;
                if (c.a.normals_out == w.a.in || c.a.normals_out == w.a.out || c.a.normals_out == input_copy) return false;
                break;
            default:
                if (c.dst_base == w.a.in || c.dst_base == w.a.out) return false;
                if (c.dst_base == normals_copy) normals_copy = nullptr;                    // overwritten again: not a copy any more
                if (c.dst_base == input_copy) input_copy = nullptr;
                if (c.src_base == w.a.normals && c.copy_src - static_cast<const char *>(c.src_base) == c.copy_dst - static_cast<const char *>(c.dst_base) &&
                    covers(rect_of_copy(c), need_n)) normals_copy = c.dst_base;
                if (c.src_base == w.a.in && c.copy_src - static_cast<const char *>(c.src_base) == c.copy_dst - static_cast<const char *>(c.dst_base) &&
                    covers(rect_of_copy(c), need_in)) input_copy = c.dst_base;            // (the same blit, not fused: "fuse_blits" 0)
                break;
        }
    }
    return normals_copy != nullptr;
}

int flush_recorded(vhr_context *ctx) {
    int rc = VHR_OK;
    const bool elide = ctx->options[kOptSvgfElideUnread] != 0;
    const bool async = ctx->options[kOptSvgfAsyncUnread] != 0 && ctx->frames_in_flight == 1;
    size_t deferred = size_t(-1);
    const void *normals_copy = nullptr, *input_copy = nullptr;
    // "reflection_async": a recorded command that reads or writes the image the mirror ray's pending launch writes, or writes one it reads,
    // waits for that launch (the SVGF pass's own commands do neither: they run beside it)
    if (ctx->refl_pending) {
        auto reads = [&](const void *p) { return p && p == ctx->refl_writes; };
        auto writes = [&](const void *p) { return p && (p == ctx->refl_writes || p == ctx->refl_reads[0] || p == ctx->refl_reads[1]); };
        bool touches = false;
        for (const SvgfCmd &c : ctx->recorded) {
            switch (c.kind) {
                case SvgfCmd::Temporal:
                    touches |= reads(c.t.normals) || reads(c.t.motion) || reads(c.t.prev_normals) || reads(c.t.history) || reads(c.t.raytraced) || reads(c.t.moments_in) ||
                               writes(c.t.integrated_out) || writes(c.t.moments_out);
                    break;
                case SvgfCmd::Atrous:
                    touches |= reads(c.a.normals) || reads(c.a.in) || writes(c.a.out) || writes(c.a.out2) || writes(c.a.normals_out);
                    break;
                default:
                    touches |= reads(c.src_base) || writes(c.dst_base);
            }
        }
        if (touches) { const int jrc = ctx->join_refl(); if (jrc != VHR_OK) { ctx->recorded.clear(); return jrc; } }
    }
    // does command c touch what the side stream's pending dispatch reads (`side_reads`) or writes (`side_writes`)?
    auto conflicts = [&](const SvgfCmd &c) {
        auto reads = [&](const void *p) { return p && p == ctx->side_writes; };
        auto writes = [&](const void *p) { return p && (p == ctx->side_writes || p == ctx->side_reads[0] || p == ctx->side_reads[1]); };
        switch (c.kind) {
            case SvgfCmd::Temporal:
                return reads(c.t.normals) || reads(c.t.motion) || reads(c.t.prev_normals) || reads(c.t.history) || reads(c.t.raytraced) || reads(c.t.moments_in) ||
                       writes(c.t.integrated_out) || writes(c.t.moments_out);
            case SvgfCmd::Atrous:
                return reads(c.a.normals) || reads(c.a.in) || writes(c.a.out) || writes(c.a.out2) || writes(c.a.normals_out);
            default:
                return reads(c.src_base) || writes(c.dst_base);
        }
    };
    size_t first = 0;
    if (ctx->deferred_raygen) {
        // "fuse_temporal": the pass in front held its TraceRays back.  If this pass starts with svgf.comp on that launch's images, the ray
        // tracing kernel runs it in its tiles' epilogues and the dispatch is done; else the launch is issued as it is, first.
        const bool fuse = !ctx->recorded.empty() && ctx->recorded[0].kind == SvgfCmd::Temporal && deferred_raygen_matches(ctx, ctx->recorded[0].t);
        if (fuse && ctx->side_pending && conflicts(ctx->recorded[0])) { rc = ctx->join_side(); if (rc != VHR_OK) { ctx->recorded.clear(); return rc; } }
        rc = flush_deferred_raygen(ctx, fuse ? &ctx->recorded[0].t : nullptr);
        if (rc != VHR_OK) { ctx->recorded.clear(); return rc; }
        if (fuse) first = 1;
    }
    for (size_t k = first; k < ctx->recorded.size(); ++k) {
        if (dead_atrous(ctx->recorded, k)) {
            if (elide) continue;
            if (async && deferred == size_t(-1) && async_candidate(ctx, ctx->recorded, k, normals_copy, input_copy)) { deferred = k; continue; }
        }
        if (ctx->side_pending && conflicts(ctx->recorded[k])) { rc = ctx->join_side(); if (rc != VHR_OK) break; }
        rc = issue_cmd(ctx, ctx->recorded[k]);
        if (rc != VHR_OK) break;
    }
    if (rc == VHR_OK && deferred != size_t(-1) && ctx->side_pending) rc = ctx->join_side();      // one dispatch at a time on the side stream
    if (rc == VHR_OK && deferred != size_t(-1)) {
        SvgfCmd cmd = ctx->recorded[deferred];
        cmd.a.normals = static_cast<const uint2 *>(normals_copy);
        if (input_copy) cmd.a.in = static_cast<const uint2 *>(input_copy);
        bool ok = true;
        if (!ctx->side_stream) {
            ok = hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->side_ready, VHR_JOIN_EVENT_FLAGS) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->side_done, VHR_JOIN_EVENT_FLAGS) == hipSuccess;
        }
        ok = ok && hipEventRecord(ctx->side_ready, ctx->stream) == hipSuccess && hipStreamWaitEvent(ctx->side_stream, ctx->side_ready, 0) == hipSuccess;
        if (!ok) {                                   // no side stream: the dispatch runs in order after all
            rc = issue_cmd(ctx, ctx->recorded[deferred]);
        } else {
            hipStream_t const main_stream = ctx->stream;
            PassDescription *const pass = ctx->cur_pass;
            ctx->stream = ctx->side_stream;
            ctx->cur_pass = nullptr;                 // the pass's time stamps stay on the caller's stream
            ctx->async_atrous = true;                // timed as its own kernel kind
            ctx->no_stamps = true;                   // (pass time stamps belong to the caller's stream)
            rc = issue_atrous(ctx, cmd.a);
            ctx->no_stamps = false;
            ctx->async_atrous = false;
            ctx->cur_pass = pass;
            if (hipEventRecord(ctx->side_done, ctx->side_stream) != hipSuccess && rc == VHR_OK) rc = ctx->fail(VHR_ERROR_DEVICE, "hipEventRecord(side stream) failed");
            ctx->stream = main_stream;
            ctx->side_pending = true;
            ctx->side_reads[0] = cmd.a.in; ctx->side_reads[1] = cmd.a.normals; ctx->side_writes = cmd.a.out;
        }
    }
    ctx->recorded.clear();
    return rc;
}

}  // namespace vhr
