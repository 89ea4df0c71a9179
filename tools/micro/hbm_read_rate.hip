// What a pure streaming READ achieves on this chip at the E-step's size: a grid-stride kernel that only loads (16 B per lane,
// non-temporal or plain) and folds the values into one register, over buffers of 146 MB / 292 MB / 1.2 GB, back to back (the 256-MB
// Infinity Cache can serve a 146-MB buffer) and alternating with a second buffer (evicts it).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_read tools/micro/hbm_read_rate.hip && /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) read_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u32x4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT>
static float run(const u32x4* a, const u32x4* b, size_t bytes, int blocks, int reps, unsigned* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) { read_kernel<NT><<<blocks, 256>>>(a, bytes / 16, out); if (b) read_kernel<NT><<<blocks, 256>>>(b, bytes / 16, out); }
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) { read_kernel<NT><<<blocks, 256>>>(a, bytes / 16, out); if (b) read_kernel<NT><<<blocks, 256>>>(b, bytes / 16, out); }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / (reps * (b ? 2 : 1));
}
int main() {
    const size_t sizes[3] = {146ull << 20, 292ull << 20, 1200ull << 20};
    u32x4 *a, *b;
    unsigned* out;
    hipMalloc(&a, sizes[2]); hipMalloc(&b, sizes[2]); hipMalloc(&out, 64);
    hipMemset(a, 1, sizes[2]); hipMemset(b, 2, sizes[2]);
    for (int s = 0; s < 3; ++s)
        for (int blocks : {1024, 2048, 8192}) {
            const float t0 = run<false>(a, nullptr, sizes[s], blocks, 20, out), t1 = run<false>(a, b, sizes[s], blocks, 20, out);
            const float t2 = run<true>(a, b, sizes[s], blocks, 20, out);
            printf("%5zu MB, %4d blocks: same buffer back to back %6.1f us = %5.2f TB/s | alternating with a second buffer %6.1f us = %5.2f TB/s | the same, non-temporal loads %6.1f us = %5.2f TB/s\n",
                   sizes[s] >> 20, blocks, t0 * 1e3, sizes[s] / t0 / 1e9, t1 * 1e3, sizes[s] / t1 / 1e9, t2 * 1e3, sizes[s] / t2 / 1e9);
        }
    return 0;
}
