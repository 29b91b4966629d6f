// Does the order of transcendental and plain vector instructions matter for a SIMD's throughput?  The a-trous tap costs a SIMD 8 cycles per
// v_exp_f32 / v_log_f32 by the knock-outs, a loop of nothing but v_exp_f32 4.4.  Same instruction totals (5 v_fma_f32 per v_exp_f32), grouped
// differently: pairs of transcendentals between plain instructions (what the compiler emits for the taps), batches of 8, batches of 16, and
// fully alternating.  8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 scratch/trans_batch.hip -o scratch/tmp/trans_batch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define F4 "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
#define F5 F4 "v_fma_f32 %0, %0, %8, %8\n"
#define E(n) "v_exp_f32 %" #n ", %" #n "\n"
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float e0 = -seed * 0.01f, e1 = e0 * 2, e2 = e0 * 3, e3 = e0 * 4;
    const float c = seed * 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        // every variant: 80 v_fma_f32 + 16 v_exp_f32 per trip
        if (KIND == 0) {       // alternating: 5 fma, 1 exp
            asm volatile(F5 E(4) F5 E(5) F5 E(6) F5 E(7) F5 E(4) F5 E(5) F5 E(6) F5 E(7) F5 E(4) F5 E(5) F5 E(6) F5 E(7) F5 E(4) F5 E(5) F5 E(6) F5 E(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
        if (KIND == 1) {       // pairs: 10 fma, 2 exp
            asm volatile(F5 F5 E(4) E(5) F5 F5 E(6) E(7) F5 F5 E(4) E(5) F5 F5 E(6) E(7) F5 F5 E(4) E(5) F5 F5 E(6) E(7) F5 F5 E(4) E(5) F5 F5 E(6) E(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
        if (KIND == 2) {       // batches of 8: 40 fma, 8 exp
            asm volatile(F5 F5 F5 F5 F5 F5 F5 F5 E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7) F5 F5 F5 F5 F5 F5 F5 F5 E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
        if (KIND == 3) {       // one batch of 16
            asm volatile(F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
        if (KIND == 4) {       // the fmas alone
            asm volatile(F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5 F5
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
        if (KIND == 5) {       // the exps alone
            asm volatile(E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7) E(4) E(5) E(6) E(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + e0 + e1 + e2 + e3 == 12345.678f) out[0] = 0;
}
template <int KIND> void run(const char *name, int waves_per_simd) {
    unsigned long long *d; (void)hipMalloc(&d, 1 << 20);
    const int iters = 200;
    const int threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;
    const int blocks = 256 * ((64 * 4 * waves_per_simd) / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += double(v);
    printf("%-34s waves/SIMD %d: %8.1f cycles of a SIMD per trip (80 v_fma_f32 + 16 v_exp_f32 unless noted), wall %7.1f us\n", name, waves_per_simd,
           sum / h.size() / iters / waves_per_simd * 1.0, ms * 1e3);
    (void)hipFree(d);
}
int main() {
    for (int w : { 8, 4, 1 }) {
        run<0>("alternating (5 fma, 1 exp)", w); run<1>("pairs (10 fma, 2 exp)", w); run<2>("batches of 8 (40 fma, 8 exp)", w); run<3>("one batch of 16 (80 fma, 16 exp)", w);
        run<4>("80 v_fma_f32 alone", w); run<5>("16 v_exp_f32 alone", w);
    }
    return 0;
}
