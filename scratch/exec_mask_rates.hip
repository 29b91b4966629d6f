// Does a vector instruction / a vector load cost less when only part of the wave is enabled?  The ray-tracing queue kernel makes 47 % of
// its wave-level trips after its queue has run dry, with <= 8 of 64 lanes enabled; whether compacting those lanes into one half / one
// row of the wave would make such a trip cheaper depends on this.  Cycles per wave-instruction on one SIMD with 8 waves resident, for
// several EXEC masks.   hipcc --offload-arch=gfx950 -O3 scratch/exec_mask_rates.hip -o scratch/tmp/exec_mask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long *out, const uint4 *__restrict__ buf, int iters, float seed, unsigned long long mask, unsigned long long mask2) {
    if ((threadIdx.x >> 6) & 4) mask = mask2;
    mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(int(mask)) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(int(mask >> 32)) << 32);          // waves 4..7 of a 512-thread block (the second wave of each SIMD's pair) take the other mask
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float c = seed * 0.5f;      // 0.5: a = a * 0.5 + 0.5 stays bounded
    const uint4 *p = buf + (threadIdx.x & 63) * 2 + (threadIdx.x >> 6) * 128;      // 32 bytes per lane, 2 KB per wave: L1 resident
    uint4 l0 = {}, l1 = {};
    __shared__ uint4 lds[512 * 2];
    lds[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
    lds[threadIdx.x + 512] = make_uint4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    const unsigned lds_addr = threadIdx.x * 16;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_mix_f32 %0, %0, %4, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %1, %4, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %2, %4, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %3, %4, %4 op_sel_hi:[0,1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 6) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 7) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s10", "s11");) }
        if (KIND == 8) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4" : "+v"(a0) : "v"(c), "v"(a1), "v"(a2), "v"(a3));) }   // one dependent chain
        if (KIND == 3) {
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(p) : "memory");
        }
        if (KIND == 9) {      // scattered: every lane its own 128-byte line (like the walkers' node loads), the same 8 KB for every wave: L1 resident
            const uint4 *q = buf + (threadIdx.x & 63) * 8;
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(q) : "memory");
        }
        if (KIND == 10) {     // scattered, one 16-byte load per trip pair replaced by a single load: what a 16-byte node would cost
            const uint4 *q = buf + (threadIdx.x & 63) * 8;
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
        }
        if (KIND == 11) {     // scattered 8-byte loads
            const uint4 *q = buf + (threadIdx.x & 63) * 8;
            asm volatile("global_load_dwordx2 %0, %2, off\n global_load_dwordx2 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(*(uint2 *)&l0), "=v"(*(uint2 *)&l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx2 %0, %2, off\n global_load_dwordx2 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(*(uint2 *)&l0), "=v"(*(uint2 *)&l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx2 %0, %2, off\n global_load_dwordx2 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(*(uint2 *)&l0), "=v"(*(uint2 *)&l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx2 %0, %2, off\n global_load_dwordx2 %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(*(uint2 *)&l0), "=v"(*(uint2 *)&l1) : "v"(q) : "memory");
        }
        if (KIND == 12 || KIND == 13 || KIND == 14) {     // ONE 16-byte load per lane; lane pairs (12) / quads (13) share a 128-byte line, (14) every lane its own line
            const uint4 *q = KIND == 12 ? buf + ((threadIdx.x & 63) >> 1) * 8 + (threadIdx.x & 1) : KIND == 13 ? buf + ((threadIdx.x & 63) >> 2) * 8 + (threadIdx.x & 3) : buf + (threadIdx.x & 63) * 8;
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(l0) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:64\n s_waitcnt vmcnt(0)" : "=v"(l1) : "v"(q) : "memory");
        }
        if (KIND == 4) {
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:8192\n s_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(lds_addr) : "memory");
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:8192\n s_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(lds_addr) : "memory");
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:8192\n s_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(lds_addr) : "memory");
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:8192\n s_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(lds_addr) : "memory");
        }
        if (KIND == 5) {      // 4-byte loads, two per trip like the stack / link words
            asm volatile("global_load_dword %0, %2, off\n global_load_dword %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0.x), "=v"(l1.x) : "v"(p) : "memory");
            asm volatile("global_load_dword %0, %2, off\n global_load_dword %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0.x), "=v"(l1.x) : "v"(p) : "memory");
            asm volatile("global_load_dword %0, %2, off\n global_load_dword %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0.x), "=v"(l1.x) : "v"(p) : "memory");
            asm volatile("global_load_dword %0, %2, off\n global_load_dword %1, %2, off offset:16\n s_waitcnt vmcnt(0)" : "=v"(l0.x), "=v"(l1.x) : "v"(p) : "memory");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + float(l0.x + l1.y) == 12345.678f) out[0] = 0;
}
template <int KIND> void run(const char *name, int waves_per_simd, unsigned long long mask, const char *mname, const uint4 *buf, unsigned long long mask2 = 0) {
    if (!mask2) mask2 = mask;
    unsigned long long *d; hipMalloc(&d, 1 << 20);
    const bool mem = (KIND >= 3 && KIND <= 5) || KIND >= 9;
    const int iters = mem ? 400 : 200, per_iter = mem ? 8 : 64;
    const int threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, buf, iters, 1.0f, mask, mask2);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, buf, iters, 1.0f, mask, mask2);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, sum_a = 0, sum_b = 0; size_t na = 0, nb = 0;
    for (size_t i = 0; i < h.size(); ++i) { sum += double(h[i]); if ((i % (threads / 64)) & 4) { sum_b += double(h[i]); ++nb; } else { sum_a += double(h[i]); ++na; } }
    const double per_wave = sum / h.size();
    if (mask2 != mask) printf("   mixed: waves with the first mask %.2f, waves with the second mask (all 64) %.2f cycles per own instruction\n", sum_a / na / (iters * double(per_iter)), sum_b / nb / (iters * double(per_iter)));
    printf("%-22s exec %-28s waves/SIMD %d: %7.2f cycles per wave-instruction (own wave), %6.2f per instruction on the SIMD; wall %7.1f us = %5.2f ns per instruction on the SIMD\n", name, mname, waves_per_simd,
           per_wave / (iters * double(per_iter)), per_wave / (iters * double(per_iter)) / waves_per_simd, ms * 1e3, ms * 1e6 / (iters * double(per_iter) * waves_per_simd));
    hipFree(d);
}
int main() {
    uint4 *buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
    struct M { unsigned long long m; const char *n; };
    std::vector<M> masks = { { ~0ull, "all 64" }, { 0xffffffffull, "low 32" }, { 0xffffffull, "low 24" }, { 0xfffffull, "low 20" }, { 0x1ffffull, "low 17" }, { 0xffffull, "low 16" },
        { 0xfffull, "low 12" }, { 0x1ffull, "low 9" }, { 0xffull, "low 8" }, { 0xfull, "low 4" }, { 0x1ull, "lane 0" }, { 0x0101010101010101ull, "8 lanes, one per 8" },
        { 0x1111111111111111ull, "16 lanes, one per 4" }, { 0x5555555555555555ull, "32 lanes, every other" }, { 0x00ff00ff00ff00ffull, "32 lanes, 8 per 16" }, { 0xffff0000ffffull, "32 lanes: rows 0 and 2" },
        { 0xffffffff00000000ull, "high 32" }, { 0xff00000000000000ull, "high 8" } };
    for (int w : { 8, 1 })
        for (auto &mk : masks) {
            run<0>("v_fma_f32", w, mk.m, mk.n, buf); run<8>("v_fma_f32 dependent", w, mk.m, mk.n, buf); run<1>("v_min/max_f32", w, mk.m, mk.n, buf); run<2>("v_fma_mix_f32", w, mk.m, mk.n, buf);
            run<6>("v_add_u32", w, mk.m, mk.n, buf); run<7>("v_cndmask_b32_e64", w, mk.m, mk.n, buf);
            if (mk.m == ~0ull || mk.m == 0xffull || mk.m == 0xffffull) { run<3>("global_load_dwordx4", w, mk.m, mk.n, buf); run<5>("global_load_dword", w, mk.m, mk.n, buf); run<4>("ds_read_b128", w, mk.m, mk.n, buf); }
        }
    for (auto &mk : masks)
        if (mk.m == ~0ull || mk.m == 0xffffffffull || mk.m == 0xffffull || mk.m == 0xffull || mk.m == 1ull || mk.m == 0x0101010101010101ull || mk.m == 0x1111111111111111ull) {
            run<14>("x4, own line", 8, mk.m, mk.n, buf); run<12>("x4, pairs share a line", 8, mk.m, mk.n, buf); run<13>("x4, quads share a line", 8, mk.m, mk.n, buf);
            run<9>("x4 pair, scattered", 8, mk.m, mk.n, buf); run<10>("x4 single, scattered", 8, mk.m, mk.n, buf); run<11>("x2 pair, scattered", 8, mk.m, mk.n, buf);
        }
    // mixed company: half of a SIMD's waves sparse, half dense
    run<0>("v_fma_f32 MIXED", 8, 0xffull, "low 8 | all 64", buf, ~0ull); run<1>("v_min/max_f32 MIXED", 8, 0xffull, "low 8 | all 64", buf, ~0ull);
    run<2>("v_fma_mix_f32 MIXED", 8, 0xffffull, "low 16 | all 64", buf, ~0ull); run<7>("v_cndmask MIXED", 8, 0xffull, "low 8 | all 64", buf, ~0ull);
    run<0>("v_fma_f32 MIXED", 8, 0x1ull, "lane 0 | all 64", buf, ~0ull);
    run<0>("v_fma_f32 MIXED", 8, 0xffull, "low 8 | low 32", buf, 0xffffffffull);
    return 0;
}
