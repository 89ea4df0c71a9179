// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns this library uses: every kernel below reads each byte of
// a 1.5-GB buffer (beyond the 256-MB Infinity Cache) exactly once, so FETCH_SIZE / bytes is the tally factor of that pattern.
//   read_wave1k   global_load_dwordx4, a wave reads 1024 contiguous bytes per instruction (assemble_visual_kernel, row kernels)
//   dma_seg128    global_load_lds_dwordx4, 8 lanes per 128-byte row segment, rows 1536 bytes apart - the ring fill of gemm_w4_kernel
//   dma_seg64     the same with 64-byte segments (4 lanes per segment)
//   read_32B      global_load_dwordx2 pairs covering 32-byte pieces at a 64-byte stride and then the other halves (im2col-like)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib     (prints the byte count per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ROWB = 1536;                       // bytes per row (K = 768 halves)

__global__ void __launch_bounds__(256) read_wave1k(const char* __restrict__ P, size_t bytes, float* sink) {
    const size_t per_block = bytes / gridDim.x;  // multiple of 4096
    const char* p = P + blockIdx.x * per_block + threadIdx.x * 16;
    float acc = 0.f;
    for (size_t o = 0; o < per_block; o += 4096) {
        const float4 v = *(const float4*)(p + o);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

// rows_per_block rows of ROWB bytes per block; a wave instruction covers 64 / (SEG / 16) rows x SEG bytes
template <int SEG>
__global__ void __launch_bounds__(256) dma_seg(const char* __restrict__ P, int rows_per_block, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;
    const int lrow = lane / LPR, lc = lane % LPR;
    const char* base = P + (size_t)blockIdx.x * rows_per_block * ROWB;
    float acc = 0.f;
    for (int r0 = 0; r0 < rows_per_block; r0 += 4 * RPI) {             // 4 waves x RPI rows per step
        const char* rp = base + (size_t)(r0 + wave * RPI + lrow) * ROWB + lc * 16;
        for (int s = 0; s < ROWB / SEG; ++s)
            __builtin_amdgcn_global_load_lds((const void*)(rp + s * SEG), (lds_ptr_t)(smem + wave * 16384 + (s & 15) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *(const float*)(smem + wave * 16384 + lane * 4);
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void __launch_bounds__(256) read_32B(const char* __restrict__ P, size_t bytes, float* sink) {
    const size_t per_block = bytes / gridDim.x;  // multiple of 16384
    const char* p = P + blockIdx.x * per_block;
    float acc = 0.f;
    for (size_t o = 0; o < per_block; o += 16384)
        for (int half = 0; half < 2; ++half) {   // lane t reads 32 bytes at 64 t + 32 half: first all the even 32-byte pieces, then the odd ones
            const float4 a = *(const float4*)(p + o + threadIdx.x * 64 + half * 32);
            const float4 b = *(const float4*)(p + o + threadIdx.x * 64 + half * 32 + 16);
            acc += a.x + b.y;
        }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int blocks = 2048, rows_per_block = 512;
    const size_t bytes = (size_t)blocks * rows_per_block * ROWB;        // 1.61 GB
    char* P; float* sink;
    CK(hipMalloc(&P, bytes)); CK(hipMemset(P, 1, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)dma_seg<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)dma_seg<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int rep = 0; rep < 2; ++rep) {
        read_wave1k<<<blocks, 256>>>(P, bytes, sink);
        dma_seg<128><<<blocks, 256, 65536>>>(P, rows_per_block, sink);
        dma_seg<64><<<blocks, 256, 65536>>>(P, rows_per_block, sink);
        read_32B<<<blocks, 256>>>(P, bytes, sink);
    }
    CK(hipDeviceSynchronize());
    printf("bytes read once per kernel launch: %zu (%.1f MB)\n", bytes, bytes / 1e6);
    return 0;
}
