// Micro-benchmark: issue rate of scalar vs packed f32 VALU instructions on gfx950 (cycles per wave64 instruction).
// hipcc --offload-arch=gfx950 -O3 pkrate.hip -o pkrate && ./pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float s0, float s1, int iters) {
    f2 a[8];
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < 8; ++i) a[i] = f2{x + i, x - i};
    f2 b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    f2 sg = {s0, s1};
    unsigned long long mask = __builtin_amdgcn_read_exec() ^ (unsigned long long) iters, dummy = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
                if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(sg), "v"(c)); }
                if (MODE == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "s"(sg)); }
                if (MODE == 4) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
                if (MODE == 5) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
                if (MODE == 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "s"(s0), "v"(c.x)); }
                if (MODE == 7) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
                if (MODE == 8) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x)); }
                if (MODE == 9) { asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(b.x) : "vcc"); }
                if (MODE == 10) { asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i].x) : "v"(b.x) : "vcc"); }
                if (MODE == 11) { asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
                if (MODE == 13) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "s"(mask)); }
                if (MODE == 14) { asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x), "v"(c.x) : "vcc"); }
                if (MODE == 15) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(a[(i + 1) & 7].y), "v"(a[(i + 2) & 7].y)); }
                if (MODE == 16) { asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
                if (MODE == 17) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
                if (MODE == 18) { asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
                if (MODE == 19) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
                if (MODE == 20) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x)); }
                if (MODE == 21) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "s"(s0), "v"(c.x)); }
                if (MODE == 22) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s0)); }
                if (MODE == 23) { asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(a[i].x), "=s"(dummy) : "s"(mask)); }
                if (MODE == 24) { asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a[i].x) : "v"(b.x), "s"(s0)); }
                if (MODE == 25) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
                if (MODE == 26) { asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(dummy) : "v"(a[i].x), "v"(b.x)); }
                if (MODE == 27) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
                if (MODE == 28) { asm volatile("v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %3, %1, %4" : "+v"(a[i].x), "+v"(a[i].y) : "s"(s0), "v"(c.x), "v"(c.y)); }
                if (MODE == 29) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(sg), "v"(c)); }
                if (MODE == 12) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x)); }
            }
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    float *out; hipMalloc(&out, 4 << 20);
    const int iters = 20000, blocks = 256 * waves_per_simd;    // 256-thread block = 1 wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.9999f, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double) iters * 64;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-44s waves/SIMD=%d  %.3f ms  cycles per instr per SIMD = %.2f\n", name, waves_per_simd, ms, cycles / (insts_per_wave * waves_per_simd));
    hipFree(out);
}

int main() {
    for (int w : {4}) {
        run<0>("v_fma_f32 vgpr", w);
        run<1>("v_pk_fma_f32 vgpr", w);
        run<2>("v_pk_fma_f32 sgpr-pair op_sel_hi", w);
        run<3>("v_pk_mul_f32 sgpr-pair neg", w);
        run<4>("v_pk_mul_f32 vgpr", w);
        run<5>("v_pk_add_f32 vgpr", w);
        run<6>("v_fma_f32 sgpr", w);
        run<7>("v_mul_f32 vgpr", w);
        run<8>("v_rcp_f32", w);
        run<9>("v_cmp_le_f32", w);
        run<10>("v_div_scale_f32", w);
        run<11>("v_div_fixup_f32", w);
        run<12>("v_cndmask_b32 vcc", w);
        run<13>("v_cndmask_b32_e64 sgpr mask", w);
        run<14>("v_cmp + v_cndmask pair (2 instrs)", w);
        run<15>("v_fma_f32 distinct vgprs", w);
        run<16>("v_min3_f32", w);
        run<17>("v_xor_b32", w);
        run<18>("v_mov_b32", w);
        run<19>("v_fmac_f32", w);
        run<20>("v_sqrt_f32", w);
        run<21>("v_fmac_f32 sgpr src0 (VOP2)", w);
        run<22>("v_mul_f32 sgpr src0 (VOP2)", w);
        run<23>("v_addc_co_u32_e64 sgpr carry", w);
        run<24>("v_bitop3_b32 sgpr", w);
        run<25>("v_and_or_b32", w);
        run<26>("v_cmp_le_f32_e64 -> sgpr pair", w);
        run<27>("v_min_f32", w);
        run<28>("2 x v_fmac_f32 sgpr (per pair)", w);
        run<29>("v_pk_fma_f32 sgpr acc form", w);
    }
    return 0;
}
