// What do the clamp bits of v_fma_mix_f32 and v_dot2_f32_f16 do on gfx950?  hipcc --offload-arch=gfx950 scratch/clamp_probe.hip -o /tmp/clamp_probe && /tmp/clamp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__global__ void probe(const uint32_t *a, const float *b, const float *c, float *out) {
    const int i = threadIdx.x;
    float r0, r1;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r1) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    const half2_t h = *reinterpret_cast<const half2_t *>(&a[i]);
    const float d0 = __builtin_amdgcn_fdot2(h, h, c[i], false), d1 = __builtin_amdgcn_fdot2(h, h, c[i], true);
    float m;
    asm volatile("v_max_f32 %0, %1, %1 clamp" : "=v"(m) : "v"(c[i]));
    out[i * 5 + 0] = r0; out[i * 5 + 1] = r1; out[i * 5 + 2] = d0; out[i * 5 + 3] = d1; out[i * 5 + 4] = m;
}
int main() {
    const int n = 6;
    uint32_t ha[n]; float hb[n], hc[n];
    const uint16_t halves[n] = { 0x3800 /*0.5*/, 0xb800 /*-0.5*/, 0x3c00 /*1*/, 0x4000 /*2*/, 0x3400 /*.25*/, 0xbc00 /*-1*/ };
    for (int i = 0; i < n; ++i) { ha[i] = (uint32_t(halves[i]) << 16) | 0x3800u; hb[i] = 0.5f; hc[i] = i == 3 ? 0.75f : (i == 5 ? -0.2f : 0.1f); }
    uint32_t *a; float *b, *c, *o; float ho[n * 5];
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&c, sizeof hc); hipMalloc(&o, sizeof ho);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(c, hc, sizeof hc, hipMemcpyHostToDevice);
    probe<<<1, n>>>(a, b, c, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("hi-half %g * 0.5 + %g: mix %g, mix clamp %g | dot2(h,h)+c %g, clamp %g | max clamp(c) %g\n", (float)(_Float16&)halves[i], hc[i], ho[i*5], ho[i*5+1], ho[i*5+2], ho[i*5+3], ho[i*5+4]);
    return 0;
}
