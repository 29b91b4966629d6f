// Issue cost of the vector instructions the traversal kernels are made of: cycles per wave-instruction on one SIMD with W waves resident,
// independent instructions, measured with s_memtime around an unrolled loop.  hipcc --offload-arch=gfx950 -O3 scratch/issue_rates.hip -o scratch/tmp/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 };
    const float c = seed * 0.5f + 1.0f;
    f2 pc = { c, c };
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
        if (KIND == 2) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 3) { REP16(asm volatile("v_max3_f32 %0, %0, %4, %1\n v_max3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_max3_f32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 4) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
        if (KIND == 5) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");) }
        if (KIND == 6) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");) }
        if (KIND == 7) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 8) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4 op_sel_hi:[0,1,0] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %1, %4, %4 op_sel_hi:[0,1,0] neg_lo:[1,0,0]\n v_pk_fma_f32 %2, %2, %4, %4 op_sel_hi:[0,1,0] neg_lo:[1,0,0]\n v_pk_fma_f32 %3, %3, %4, %4 op_sel_hi:[0,1,0] neg_lo:[1,0,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
        if (KIND == 9) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 10) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 11) { REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" ::);) }
        if (KIND == 12) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s10", "s11");) }
        if (KIND == 13) { REP16(asm volatile("v_fma_f32 %0, %0, s10, %4\n v_fma_f32 %1, %1, s10, %4\n v_fma_f32 %2, %2, s10, %4\n v_fma_f32 %3, %3, s10, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s10");) }
        if (KIND == 14) { REP16(asm volatile("v_pk_fma_f32 %0, s[10:11], %4, %4\n v_pk_fma_f32 %1, s[10:11], %4, %4\n v_pk_fma_f32 %2, s[10:11], %4, %4\n v_pk_fma_f32 %3, s[10:11], %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc) : "s10", "s11");) }
        if (KIND == 15) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_le_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");) }
        if (KIND == 16) { REP16(asm volatile("v_cmp_le_f32 s[10:11], %0, %4\n v_cmp_le_f32 s[12:13], %1, %4\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[12:13]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s10", "s11", "s12", "s13");) }
        if (KIND == 17) { REP16(asm volatile("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 18) { REP16(asm volatile("v_bfi_b32 %0, %0, %4, %1\n v_bfi_b32 %1, %1, %4, %2\n v_bfi_b32 %2, %2, %4, %3\n v_bfi_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 19) { REP16(asm volatile("v_readfirstlane_b32 s10, %0\n v_readfirstlane_b32 s11, %1\n v_readfirstlane_b32 s12, %2\n v_readfirstlane_b32 s13, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s10", "s11", "s12", "s13");) }
        if (KIND == 20) { REP16(asm volatile("s_and_b64 s[10:11], s[12:13], exec\n s_bcnt1_i32_b64 s14, s[10:11]\n s_and_b64 s[10:11], s[12:13], exec\n s_bcnt1_i32_b64 s14, s[10:11]" ::: "s10", "s11", "s12", "s13", "s14", "scc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + p0.x + p1.x + p2.x + p3.x + p0.y == 12345.678f) out[0] = 0;
}
template <int KIND> void run(const char *name, int waves_per_simd) {
    unsigned long long *d; hipMalloc(&d, 1 << 20);
    const int iters = 200, threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;   // one block per CU-ish; waves spread over 4 SIMDs
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += double(v);
    const double per_wave = sum / h.size();                 // s_memtime ticks (100 MHz-based? reported raw) for iters * 64 instructions of ONE wave
    printf("%-28s waves/SIMD %d: %.2f ticks per wave-instruction (own), x%d waves sharing the SIMD -> %.2f ticks per instruction issued on the SIMD\n", name, waves_per_simd,
           per_wave / (iters * 64.0), waves_per_simd, per_wave / (iters * 64.0) / waves_per_simd);
    hipFree(d);
}
int main() {
    for (int w : { 1, 2, 8 }) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<8>("v_pk_fma_f32 op_sel/neg", w); run<4>("v_pk_mul_f32", w); run<2>("v_min_f32", w); run<3>("v_max3_f32", w);
        run<5>("v_cndmask_b32 vcc", w); run<12>("v_cndmask_b32_e64 s[10:11]", w); run<15>("cmp vcc + cndmask vcc pairs", w); run<16>("cmp sgpr + cndmask sgpr (x2)", w); run<13>("v_fma_f32 with SGPR src", w); run<14>("v_pk_fma_f32 with SGPR pair", w); run<17>("v_max/v_min mix", w); run<18>("v_bfi_b32", w); run<19>("v_readfirstlane", w); run<20>("s_and_b64+s_bcnt1", w); run<6>("v_cmp_le_f32", w); run<9>("v_mov_b32", w); run<10>("v_add_u32", w); run<7>("v_exp_f32", w); run<11>("s_nop 0", w);
    }
    return 0;
}
