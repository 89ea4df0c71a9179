// Issue-rate probe for the float64 / conversion VALU instructions the k-means kernels lean on (gfx950).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
#define LOOPS 4096
template <int OP>
__global__ void __launch_bounds__(256) probe(float* out, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    float f0 = seed, f1 = seed * 2, f2 = seed * 3, f3 = seed * 4;
    float g0 = seed * 5, g1 = seed * 6, g2 = seed * 7, g3 = seed * 8;
    for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 4; ++r) {
            if (OP == 0) { asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            if (OP == 1) { asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            if (OP == 2) { asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3)); }
            if (OP == 3) { asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 4) { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 5) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 6) { asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            if (OP == 8) { asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 9) { asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 10) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 11) { asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1" : "+v"(a0), "+v"(a1)); asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1" : "+v"(a2), "+v"(a3)); }
            if (OP == 12) { asm volatile("v_pk_mul_f16 %0, %0, %0\n v_pk_mul_f16 %1, %1, %1\n v_pk_mul_f16 %2, %2, %2\n v_pk_mul_f16 %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
            if (OP == 13) { asm volatile("v_max_f64 %0, %0, %0\n v_max_f64 %1, %1, %1\n v_max_f64 %2, %2, %2\n v_max_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            // OP 14 / 15: a transcendental and one / two independent full-rate instructions alternating (counted as REP transcendentals):
            // does the transcendental unit run beside the main VALU?
            if (OP == 14) { asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3)); }
            if (OP == 15) { asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %7, %7, %7, %7\n v_exp_f32 %2, %2\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %7, %7, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3)); }
            if (OP == 7) { asm volatile("v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1\n v_cvt_f32_f64 %6, %2\n v_cvt_f32_f64 %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3) + f0 + f1 + f2 + f3 + g0 + g1 + g2 + g3;
}
template <int OP>
static void run(const char* name, int waves_per_simd) {
    float* out;
    hipMalloc(&out, 4 << 20);
    const int blocks = 256 * waves_per_simd;       // 4 waves per block = 1 per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP><<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)LOOPS * REP * waves_per_simd;
    printf("%-16s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<4>("v_fma_f32", w);
        run<0>("v_fma_f64", w);
        run<1>("v_add_f64", w);
        run<6>("v_mul_f64", w);
        run<2>("v_cvt_f64_f32", w);
        run<7>("v_cvt_f32_f64", w);
        run<3>("v_cvt_f32_f16", w);
        run<5>("v_exp_f32", w);
        run<10>("v_rcp_f32", w);
        run<8>("v_exp_f16", w);
        run<9>("v_rcp_f16", w);
        run<11>("v_pk_mul_f32", w);
        run<12>("v_pk_mul_f16", w);
        run<13>("v_max_f64", w);
        run<14>("v_exp_f32+1fma", w);
        run<15>("v_exp_f32+2fma", w);
    }
    return 0;
}
