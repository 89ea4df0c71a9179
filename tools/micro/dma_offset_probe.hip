// Probe: does the instruction offset of global_load_lds_dwordx4 move the LDS destination, the global source, or both?
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_offset_probe dma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void probe(const unsigned* src, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // 2048 dwords = 8 KB
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ::"s"(sbase), "v"(voff), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *src, *out, h[2048], hs[4096];
    for (int i = 0; i < 4096; ++i) hs[i] = i;      // dword i holds i: the value tells which global dword arrived
    CK(hipMalloc(&src, sizeof(hs))); CK(hipMalloc(&out, sizeof(h)));
    CK(hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice));
    probe<<<1, 64, 8192>>>(src, out);
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    int first = -1;
    for (int i = 0; i < 2048; ++i) if (h[i] != 0xdeadbeef) { first = i; break; }
    printf("first written LDS dword: %d (byte %d), holds global dword %u (byte %u)\n", first, first * 4, first >= 0 ? h[first] : 0, first >= 0 ? h[first] * 4 : 0);
    return 0;
}
