// What one a-trous tap's arithmetic (svgf_atrous_stream_kernel, weights in the exponent: 16 vector instructions + 3 transcendentals) costs a SIMD
// with W waves resident when its operands are already in registers: the tap's source as in kernels_svgf.hip, the staged texel made opaque
// to the compiler per tap (an empty asm), four taps per loop trip like the kernel's groups.  No LDS, no memory: the issue rate of the mix itself.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scratch/tap_rate.hip -o scratch/tmp/tap_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2_t as_half2(uint32_t u) { return *reinterpret_cast<const half2_t *>(&u); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
template <int TAPS>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, float seed, uint32_t *sink) {
    uint32_t qx = __float_as_uint(seed + threadIdx.x * 1e-3f), qy = qx + 77u, qz = 0x3c003c00u, qw = 0x38003800u, niq = 0x3c000001u;
    const f2v p_xy = f2v{ seed, seed * 0.5f };
    const half2_t np_xy = as_half2(0x38003800u);
    const float np_z = 0.7f;
    const _Float16 idp = as_half2(niq).x;
    const f2v inv = f2v{ 1.5f, 2.5f };
    f2v sw = f2v{ 1.0f, 1.0f }, s01 = p_xy;
    float s2 = 0.1f, s3 = 0.2f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < TAPS / 4; ++g) {
            uint32_t ax[4], ay[4], az[4], aw[4], an[4];
            float L[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                ax[h] = qx; ay[h] = qy; az[h] = qz; aw[h] = qw; an[h] = niq;
                asm volatile("" : "+v"(ax[h]), "+v"(ay[h]), "+v"(az[h]), "+v"(aw[h]), "+v"(an[h]));      // "the staged texel": opaque per tap
                float dd;
                asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(dd) : "v"(an[h]), "v"(np_z));
                dd = __builtin_amdgcn_fdot2(np_xy, as_half2(aw[h]), dd, false);
                const float lg = __builtin_amdgcn_logf(fmaxf(dd, 0.0f));
                L[h] = as_half2(an[h]).x == idp ? fmaf(lg, 128.0f, -2.0f - 1.41503749927884381f) : -__builtin_inff();
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f2v q_xy = f2v{ u2f(ax[h]), u2f(ay[h]) };
                const f2v dl = p_xy - q_xy;
                const f2v w2 = f2v{ __builtin_amdgcn_exp2f(fmaf(-fabsf(dl.x), inv.x, L[h])), __builtin_amdgcn_exp2f(fmaf(-fabsf(dl.y), inv.y, L[h])) };
                sw += w2;
                s01 = __builtin_elementwise_fma(w2, q_xy, s01);
                const f2v wq = w2 * w2;
                const half2_t q_zw = as_half2(az[h]);
                s2 = fmaf(wq.x, float(q_zw.x), s2);
                s3 = fmaf(wq.y, float(q_zw.y), s3);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (sw.x + sw.y + s01.x + s01.y + s2 + s3 == 12345.678f) sink[0] = 0;
}
template <int TAPS> void run(int waves_per_simd) {
    unsigned long long *d; uint32_t *sink; (void)hipMalloc(&d, 1 << 20); (void)hipMalloc(&sink, 64);
    const int iters = 500, threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads, blocks = 256 * blocks_per_cu;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<TAPS>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += double(v);
    const double per_tap = sum / h.size() / (double(iters) * TAPS);
    printf("%2d taps per loop trip, waves/SIMD %d: %6.1f cycles per tap per wave = %5.1f cycles of the SIMD per tap (16 + 3 transcendental instructions: %.2f cycles per instruction)\n",
           TAPS, waves_per_simd, per_tap, per_tap / waves_per_simd, per_tap / waves_per_simd / 19.0);
    (void)hipFree(d); (void)hipFree(sink);
}
int main() {
    for (int w : { 1, 2, 4, 6, 8 }) { run<4>(w); run<24>(w); }
    return 0;
}
