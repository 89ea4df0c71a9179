// Micro-benchmark: rate of global_load_lds_dwordx4 ring fills from an L2-resident panel, 64-byte vs 128-byte row segments.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_fill_bench dma_fill_bench.hip ; run: ./dma_fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// panel: rows x K halves (row stride K*2 bytes). Each block sweeps `steps` sub-steps; a sub-step fills 32 KB:
// SEG=64: 512 rows x 64 B (k32 step), SEG=128: 256 rows x 128 B. 4 waves, 8 instrs per wave per sub-step, ring of 4 slots.
template <int SEG>
__global__ void __launch_bounds__(256) fill_kernel(const char* __restrict__ P, int rows, int rowbytes, int steps, int nblk_rows, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per instruction
    const int lrow = lane / LPR, lc = lane % LPR;
    const int segs_per_row = rowbytes / SEG;
    int rb = (blockIdx.x * 37) % nblk_rows;  // row-block this block reads from (512 rows each)
    int kseg = 0;
    auto issue = [&](int slot) {
        const char* base = P + (size_t)rb * 512 * rowbytes + (size_t)kseg * SEG + lc * 16;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = (wave * 8 + p) * RPI + lrow;       // SEG=64: 512 rows; SEG=128: 256 rows
            __builtin_amdgcn_global_load_lds((const void*)(base + (size_t)row * rowbytes), (lds_ptr_t)(smem + slot * 32768 + (wave * 8 + p) * 1024), 16, 0, 0);
        }
        if (++kseg == segs_per_row) { kseg = 0; rb = (rb + 1) % nblk_rows; }
    };
    issue(0); issue(1); issue(2);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue((s + 3) & 3);
        acc += *(const float*)(smem + (s & 3) * 32768 + threadIdx.x * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int rowbytes = 1536;               // K = 768 halves
    const int nblk_rows = 4;                 // 4 x 512 rows x 1536 B = 3 MB panel per... (shared by all blocks: L2 resident)
    const size_t bytes = (size_t)nblk_rows * 512 * rowbytes;
    char* P; float* sink;
    CK(hipMalloc(&P, bytes)); CK(hipMemset(P, 1, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)fill_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)fill_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 4000;
    for (int rep = 0; rep < 2; ++rep)
        for (int seg : {64, 128}) {
            for (int it = 0; it < 2; ++it) {
                CK(hipEventRecord(e0));
                if (seg == 64) fill_kernel<64><<<256, 256, 131072>>>(P, 512 * nblk_rows, rowbytes, steps, nblk_rows, sink);
                else fill_kernel<128><<<256, 256, 131072>>>(P, 512 * nblk_rows, rowbytes, steps, nblk_rows, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it) printf("seg=%3d B: %.3f ms, %.1f GB/s per CU, %.2f TB/s chip\n", seg, ms, 32768.0 * (steps + 3) / ms / 1e6, 256 * 32768.0 * (steps + 3) / ms / 1e9);
            }
        }
    return 0;
}
