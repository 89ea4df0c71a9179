// What a pure WRITE achieves on this chip in the pattern of the GEMM epilogue (gemm_w4_kernel): one persistent 256-thread block
// per CU, a block writes 256 x 256 fp16 output tiles of a row-major [M][N] matrix, a wave its 128 x 128 quarter as 32 store
// instructions of 4 rows x 256 B.  Question behind it: the epilogue's stores cost 17-65 us per GEMM of a block of the encoder
// (epilogue ablations) - is that the chip's write rate when all CUs store at once, and is the limit global (HBM /
// Infinity Cache) or per XCD (the XCD's fabric port)?  Variants: plain / non-temporal stores; all 8 XCDs or only 1 / 2 / 4 of them
// writing (blockIdx % 8 < active: the round-robin block -> XCD mapping of the guide).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_write tools/micro/hbm_write_rate.hip && /tmp/hbm_write
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// PAT = 1 (round 5): the pattern a register-only epilogue would store in (no LDS transposition: after one v_permlane16_swap per register
// pair a lane holds 16 contiguous bytes) - an instruction covers 16 rows x 64 B (four 16-byte pieces per row) instead of 4 rows x 256 B
template <bool NT, int PAT = 0>
__global__ void __launch_bounds__(256) write_kernel(char* __restrict__ c, int tiles_m, int tiles_n, int n_bytes, int active) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (xcd >= active) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, q16 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int total = tiles_m * tiles_n, nblk = (gridDim.x >> 3) * active;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = slot * active + xcd; t < total; t += nblk) {
        const int bm = t / tiles_n, bn = t - bm * tiles_n;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const size_t row = PAT ? (size_t)bm * 256 + wm * 128 + (i >> 2) * 16 + c16 : (size_t)bm * 256 + wm * 128 + i * 4 + q16;
            char* p = PAT ? c + row * n_bytes + bn * 512 + wn * 256 + (i & 3) * 64 + ((q16 & 1) * 32 + (q16 >> 1) * 16)
                          : c + row * n_bytes + bn * 512 + wn * 256 + c16 * 16;
            if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
        }
    }
}
template <bool NT, int PAT = 0>
static float run(char* c, int tiles_m, int tiles_n, int active, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    write_kernel<NT, PAT><<<256, 256>>>(c, tiles_m, tiles_n, tiles_n * 512, active);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) write_kernel<NT, PAT><<<256, 256>>>(c, tiles_m, tiles_n, tiles_n * 512, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const int tiles_m = 512;                                   // 131,072 rows = one 665-image batch of the encoder
    char* c;
    hipMalloc(&c, (size_t)tiles_m * 256 * 3072 * 2);
    for (int tiles_n : {3, 9, 12})
        for (int active : {8, 4, 2, 1}) {
            const double bytes = (double)tiles_m * 256 * tiles_n * 512 * active / 8.0;
            const float t0 = run<false>(c, tiles_m * active / 8, tiles_n, active, 10), t1 = run<true>(c, tiles_m * active / 8, tiles_n, active, 10);
            printf("N = %4d, %d XCDs writing, %6.0f MB: plain stores %7.1f us = %5.2f TB/s | non-temporal %7.1f us = %5.2f TB/s\n", tiles_n * 256,
                   active, bytes / 1e6, t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9);
        }
    // the two store patterns at the bench's launch size (3,990 images = 3,072 row tiles), all XCDs
    hipFree(c);
    const int tm2 = 3072;
    hipMalloc(&c, (size_t)tm2 * 256 * 3072 * 2);
    for (int tiles_n : {9, 12}) {
        const double bytes = (double)tm2 * 256 * tiles_n * 512;
        const float a0 = run<false, 0>(c, tm2, tiles_n, 8, 5), a1 = run<true, 0>(c, tm2, tiles_n, 8, 5);
        const float b0 = run<false, 1>(c, tm2, tiles_n, 8, 5), b1 = run<true, 1>(c, tm2, tiles_n, 8, 5);
        printf("N = %4d, %6.0f MB: 4 rows x 256 B per instruction: plain %7.1f us (%5.2f TB/s), nt %7.1f us (%5.2f TB/s) | 16 rows x 64 B: plain %7.1f us (%5.2f TB/s), nt %7.1f us (%5.2f TB/s)\n",
               tiles_n * 256, bytes / 1e6, a0 * 1e3, bytes / a0 / 1e9, a1 * 1e3, bytes / a1 / 1e9, b0 * 1e3, bytes / b0 / 1e9, b1 * 1e3, bytes / b1 / 1e9);
    }
    return 0;
}