// Micro-benchmark: what the chip sustains on random fp16 operands for the two MFMA shapes the similarity kernel can use,
// with the wave geometry of sim_topk_rb8_kernel (A fragments re-read from LDS by ds_read_b128, B fragments in registers).
// Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime, median over blocks).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// SHAPE 32: 32 steps of v_mfma_f32_32x32x16_f16 on one accumulator (32 names x 32 images x 512);
// SHAPE 16: 16 k32-steps x 2 x 2 v_mfma_f32_16x16x32_f16 on four accumulators (the same tile).  LDS: A fragments come from LDS
// (one ds_read_b128 per 32x32x16, two per k32-step of 16x16x32), three steps ahead.
template <int SHAPE, bool LDS>
__global__ void __launch_bounds__(512) rate_kernel(const half8* __restrict__ src, int iters, float* sink, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    half8 bf[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) bf[s] = src[(blockIdx.x * 64 + lane) * 32 + s];
    for (int i = threadIdx.x; i < 32768 / 16; i += blockDim.x) ((half8*)smem)[i] = src[i + 977];
    __syncthreads();
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int r = lane & 31, hh = lane >> 5;
    unsigned fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = sbase + (unsigned)(r * 1024 + ((32 * j) ^ (16 * (hh ^ (r & 15)))));
    half8 fr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fr[j] = src[lane + 64 * j + 5];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 32) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (LDS) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(s + 3) & 3]) : "v"(fa[(s + 3) & 7]), "n"((((s + 3) & 31) >> 3) * 256));
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr[(s + 1) & 3]));
                }
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fr[s & 3]), "v"(bf[s]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
        if (acc[0] == 123.456f) sink[0] = acc[1];
    } else {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[q][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 32; s += 2) {                    // k32-step s / 2: fragments fr[s & 3], fr[(s + 1) & 3] = two name tiles
                if (LDS) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(s + 2) & 3]) : "v"(fa[(s + 2) & 7]), "n"((((s + 2) & 31) >> 3) * 256));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(s + 3) & 3]) : "v"(fa[(s + 3) & 7]), "n"((((s + 3) & 31) >> 3) * 256));
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr[s & 3]), "+v"(fr[(s + 1) & 3]));
                }
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(fr[s & 3]), "v"(bf[s]));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(fr[s & 3]), "v"(bf[s + 1]));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(fr[(s + 1) & 3]), "v"(bf[s]));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(fr[(s + 1) & 3]), "v"(bf[s + 1]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        if (acc[0][0] == 123.456f) sink[0] = acc[1][1] + acc[2][0] + acc[3][0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, bool LDS>
static void run(const char* name, int threads, const half8* src, float* sink, unsigned long long* stamps, int iters) {
    CK(hipFuncSetAttribute((const void*)rate_kernel<SHAPE, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        rate_kernel<SHAPE, LDS><<<256, threads, 32768>>>(src, iters, sink, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
    }
    std::vector<unsigned long long> st(512);
    CK(hipMemcpy(st.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < 256; ++b) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);   // GHz (realtime = 100 MHz)
    std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 32 * 32 * 512 * (double)iters * (threads / 64) * 256;
    printf("%-44s %2d waves/CU: %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz  cycles/(32x32x512 tile) %.0f\n", name, threads / 64, best,
           flop / best / 1e9, clk[128], clk[128] * 1e9 * best * 1e-3 / iters);
}

int main() {
    const size_t n = 256 * 64 * 32 + 4096;
    std::vector<_Float16> h(n * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)(((rand() % 2001) - 1000) / 4000.0f);
    half8* src;
    float* sink;
    unsigned long long* stamps;
    CK(hipMalloc(&src, n * 16));
    CK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&stamps, 512 * 8));
    const int iters = 6000;
    for (int rep = 0; rep < 2; ++rep) {
        run<32, false>("32x32x16 registers only", 256, src, sink, stamps, iters);
        run<16, false>("16x16x32 registers only", 256, src, sink, stamps, iters);
        run<32, false>("32x32x16 registers only", 512, src, sink, stamps, iters);
        run<16, false>("16x16x32 registers only", 512, src, sink, stamps, iters);
        run<32, true>("32x32x16 A from LDS (ds_read_b128 / MFMA)", 512, src, sink, stamps, iters);
        run<16, true>("16x16x32 A from LDS (ds_read_b128 / 2 MFMA)", 512, src, sink, stamps, iters);
    }
    return 0;
}
