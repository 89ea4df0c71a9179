// fp16 GEMM with fused epilogue for the encoder blocks (gfx950, v_mfma_f32_32x32x16_f16).
//   C[m,n] = act(A[m,k] @ W[n,k]^T + bias[n]) + residual[m,n]        (fp16 in/out, fp32 accumulate)
// W is stored [n][k] exactly like torch Linear / MultiheadAttention.in_proj_weight, so both operands are
// k-contiguous.  Replaces the cuBLAS calls torch issues for the CLIP / DINO ViT blocks (a1, a4, a19 in
// SURVEY.md 8a; call sites main_unsup.py:127, clip_lang_util.py:102).
//
// Tiling: block = 4 waves (2x2), block tile 128(m) x 128(n) x 64(k), wave tile 64x64 = 2x2 MFMA tiles.
// The MFMA "A" operand is the W tile (rows = n) and the "B" operand the activation tile (cols = m), so a lane
// ends up with 4 consecutive n for one m per register quad: the fp16 output is written as 8-byte pieces.
// LDS tiles are [128 rows][64 k] fp16 with 128-B rows, 16-B chunks XOR-swizzled by (row>>1)&7 so the
// ds_read_b128 fragment reads are bank-conflict-free; staging is register double-buffering (global loads for
// tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after them).
#include "common.h"
#include "gemm.h"
#include <stdlib.h>

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == SCD_ACT_QUICKGELU) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    if (act == SCD_ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    return x;
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ void __launch_bounds__(256) gemm_f16_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                       const float* __restrict__ bias, const half_t* __restrict__ R,
                                                       half_t* __restrict__ C, int M, int N, int K, int tiles_n,
                                                       int total_tiles) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][128 * 128];   // [buf][A=0/W=1][16 KB]
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous
    // run of tiles (n fastest) so neighbouring tiles re-use the same activation rows from that XCD's L2.
    const int nwg = total_tiles;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    const int bm = tile / tiles_n, bn = tile % tiles_n;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const half_t* Ab = A + (size_t)bm * 128 * K;
    const half_t* Wb = W + (size_t)bn * 128 * K;
    // staging: thread -> (row = p*32 + tid/8, chunk = tid%8), 4 passes per operand
    const int srow = tid >> 3, schunk = tid & 7;

    const half_t* ga = Ab + (size_t)srow * K + 8 * schunk;     // + p*32*K + k0
    const half_t* gw = Wb + (size_t)srow * K + 8 * schunk;
    const size_t pstride = (size_t)32 * K;
    uint4 ra0 = *(const uint4*)(ga), ra1 = *(const uint4*)(ga + pstride), ra2 = *(const uint4*)(ga + 2 * pstride),
          ra3 = *(const uint4*)(ga + 3 * pstride);
    uint4 rw0 = *(const uint4*)(gw), rw1 = *(const uint4*)(gw + pstride), rw2 = *(const uint4*)(gw + 2 * pstride),
          rw3 = *(const uint4*)(gw + 3 * pstride);
    const int so0 = lds_off(srow, schunk), so1 = lds_off(32 + srow, schunk), so2 = lds_off(64 + srow, schunk),
              so3 = lds_off(96 + srow, schunk);
    *(uint4*)(lds[0][0] + so0) = ra0; *(uint4*)(lds[0][0] + so1) = ra1;
    *(uint4*)(lds[0][0] + so2) = ra2; *(uint4*)(lds[0][0] + so3) = ra3;
    *(uint4*)(lds[0][1] + so0) = rw0; *(uint4*)(lds[0][1] + so1) = rw1;
    *(uint4*)(lds[0][1] + so2) = rw2; *(uint4*)(lds[0][1] + so3) = rw3;
    __syncthreads();

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }

    const int nk = K >> 6;
    const int ow0 = wn * 64 + r, ow1 = wn * 64 + 32 + r, oa0 = wm * 64 + r, oa1 = wm * 64 + 32 + r;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // prefetch the next k-tile into registers (the last iteration re-reads its own tile: harmless)
        const int k0 = (kt + 1 < nk ? kt + 1 : kt) << 6;
        ra0 = *(const uint4*)(ga + k0); ra1 = *(const uint4*)(ga + pstride + k0);
        ra2 = *(const uint4*)(ga + 2 * pstride + k0); ra3 = *(const uint4*)(ga + 3 * pstride + k0);
        rw0 = *(const uint4*)(gw + k0); rw1 = *(const uint4*)(gw + pstride + k0);
        rw2 = *(const uint4*)(gw + 2 * pstride + k0); rw3 = *(const uint4*)(gw + 3 * pstride + k0);
        const char* la = lds[cur][0];
        const char* lw = lds[cur][1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 fw0 = *(const half8*)(lw + lds_off(ow0, 2 * s + hh));
            const half8 fw1 = *(const half8*)(lw + lds_off(ow1, 2 * s + hh));
            const half8 fa0 = *(const half8*)(la + lds_off(oa0, 2 * s + hh));
            const half8 fa1 = *(const half8*)(la + lds_off(oa1, 2 * s + hh));
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw0, fa0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw0, fa1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fa0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fa1, acc11, 0, 0, 0);
        }
        char* na = lds[cur ^ 1][0];
        char* nw = lds[cur ^ 1][1];
        *(uint4*)(na + so0) = ra0; *(uint4*)(na + so1) = ra1; *(uint4*)(na + so2) = ra2; *(uint4*)(na + so3) = ra3;
        *(uint4*)(nw + so0) = rw0; *(uint4*)(nw + so1) = rw1; *(uint4*)(nw + so2) = rw2; *(uint4*)(nw + so3) = rw3;
        __syncthreads();
    }
    f32x16 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};

    // epilogue: lane holds, for m = col r, the n-quads 8g+4h .. +3 of each 32x32 tile
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const size_t m = (size_t)bm * 128 + wm * 64 + tm * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = bn * 128 + wn * 64 + tn * 32 + 8 * g + 4 * hh;
                f32x4 v;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = acc[tn][tm][4 * g + q4];
                if (HAS_BIAS) {
                    const float4 b4 = *(const float4*)(bias + n0);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = act_apply(v[q4], ACT);
                if (HAS_RES) {
                    const half4 r4 = *(const half4*)(R + m * N + n0);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) v[q4] += (float)r4[q4];
                }
                half4 o;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) o[q4] = (half_t)v[q4];
                *(half4*)(C + m * N + n0) = o;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// v2: persistent 256x256x64 kernel.  8 waves (2 m x 4 n), wave tile 128(m) x 64(n) = 4x2 MFMA tiles (128 accumulator
// registers), LDS-DMA staging (global_load_lds, 16 B/lane) into a 2-stage ring of [A 256x64 | W 256x64] = 64 KB per stage.
// A block owns a contiguous run of output tiles (n fastest) and runs ONE flattened software pipeline over
// (tile, k-step): the loads of step s+1 are issued right after the barrier that opens step s, so they fly during the
// 32 MFMAs/wave of step s, also across tile boundaries - a tile's epilogue overlaps the next tile's first loads.
// LDS-DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
// address and again on the ds_read side (same involution on both).
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ void __launch_bounds__(512) gemm256_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                      const float* __restrict__ bias, const half_t* __restrict__ R,
                                                      half_t* __restrict__ C, int M, int N, int K, int tiles_n, int total_tiles, int xmode, int ng) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 x (32 KB A + 32 KB W) + 8 x 4 KB epilogue patches
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K >> 5;
    // tile order: n-tiles are grouped (ng per group) and a group is swept over ALL m-tiles before the next one, so the
    // group's W panels stay resident in the XCD's L2 while the activation panels stream through exactly once per group
    const int tiles_m = total_tiles / tiles_n;
    const int per_group = tiles_m * ng;
    auto tile_mn = [&](int t, int& bm, int& bn) {
        const int g = t / per_group;
        const int local = t - g * per_group;
        const int n0 = g * ng;
        const int w = tiles_n - n0 < ng ? tiles_n - n0 : ng;       // width of this (possibly last, narrower) group
        bm = local / w;
        bn = n0 + local - bm * w;
    };
    // Tile schedule for L2 locality: blocks b and b+8 share an XCD (round-robin dispatch).  XCD x owns the contiguous
    // tile range [x*T/8, (x+1)*T/8) (n fastest) and its resident blocks take tiles round-robin, so that at any time the
    // blocks of one XCD hold ~32 CONSECUTIVE tiles: a handful of A row-panels x all W column-panels, swept through k in
    // lockstep -> each k-slab is fetched into that XCD's L2 once and then hit by the other blocks that share it.
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;
    const int c0 = (int)((long long)xcd * total_tiles / nxcd), c1 = (int)((long long)(xcd + 1) * total_tiles / nxcd);
    const int tb = c0 + slot;
    const int my_tiles = tb < c1 ? (c1 - tb + per_xcd - 1) / per_xcd : 0;
    const int steps = my_tiles * nk;
    if (steps <= 0) return;
    const int tstride = per_xcd;

    // 64-byte LDS rows (32 k): 16-B chunk XOR ((row>>2)&3) keeps the ds_read_b128 fragment reads conflict-free.
    // per-lane source offsets of the 2 row groups this wave stages per operand (rows wave*32 + p*16 + lane/4)
    const int lrow = lane >> 2, pc = lane & 3;
    int src_off[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rowl = wave * 32 + p * 16 + lrow;
        src_off[p] = rowl * K + ((pc ^ ((rowl >> 2) & 3)) << 3);
    }
    auto issue = [&](int tile, int kt, int slot) {
        int bm, bn;
        tile_mn(tile, bm, bn);
        const half_t* ga = A + (size_t)bm * 256 * K + kt * 32;
        const half_t* gw = W + (size_t)bn * 256 * K + kt * 32;
        char* sa = smem + slot * 32768 + wave * 2048;
        char* sw = sa + 16384;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const void*)(ga + src_off[p]), (lds_ptr_t)(sa + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(gw + src_off[p]), (lds_ptr_t)(sw + p * 1024), 16, 0, 0);
        }
    };
    auto off32 = [](int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };

    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;

    // flattened pipeline over (tile, 32-deep k sub-step): 4-slot ring, 3 sub-tiles (12 LDS-DMA per wave) in flight
    int tile = (int)tb, kt = 0;          // sub-step being computed
    int ntile = tile, nkt = 0;           // sub-step being loaded
#pragma unroll
    for (int pre = 0; pre < 3; ++pre) {
        if (pre < steps) issue(ntile, nkt, pre);
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
    }
    int store_age = 4;                   // iterations since the last epilogue issued its 16 stores
    half8 rpre[4];                       // residual rows of the next epilogue block, prefetched under the MFMAs
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) rpre[p][q] = (half_t)0.f;

    // Fragment registers: F0 feeds the first 8 MFMAs of a sub-step (k16 group 0), F1 the second 8 (group 1).  The LDS
    // reads of a group are issued one MFMA group ahead, and the wait+barrier that publishes slot s+1 sits in the MIDDLE
    // of sub-step s, so a staged slot is first read half a sub-step after the barrier that retired its DMAs.
    half8 f0w[2], f0a[4], f1w[2], f1a[4];
    const int rw0 = wn * 64 + r, ra0 = wm * 128 + r;
    auto rd = [&](const char* slot, int k16, half8 (&fw)[2], half8 (&fa)[4]) {
        const char* la = slot;
        const char* lw = slot + 16384;
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[j] = *(const half8*)(lw + off32(rw0 + j * 32, 2 * k16 + hh));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const half8*)(la + off32(ra0 + i * 32, 2 * k16 + hh));
    };
    // slot 0 must be visible before the first fragment reads
    if (steps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd(smem, 0, f0w, f0a);
    for (int s = 0; s < steps; ++s) {
        const char* cur = smem + (s & 3) * 32768;
        if (HAS_RES && kt == nk - 2) {     // residual rows of epilogue block i=0: two sub-steps of MFMAs hide the latency
            int bm, bn;
            tile_mn(tile, bm, bn);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                rpre[p] = *(const half8*)(R + ((size_t)bm * 256 + wm * 128 + p * 8 + (lane >> 3)) * N + bn * 256 + wn * 64 + (lane & 7) * 8);
        }
        rd(cur, 1, f1w, f1a);                                  // group 1 of this sub-step, overlaps the MFMAs below
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0w[j], f0a[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // publish slot s+1: everything younger than its DMAs (1 sub-tile = 4 DMAs, + 16 epilogue stores if recent) may fly
        if (s + 1 < steps) {
            if (s + 2 >= steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (store_age < 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        ++store_age;
        __builtin_amdgcn_s_barrier();    // slot s+1 visible to all waves; slot (s+3)&3 == (s-1)&3 is free again
        asm volatile("" ::: "memory");
        if (s + 3 < steps && !((xmode & 1) && s >= 2)) issue(ntile, nkt, (s + 3) & 3);
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
        if (s + 1 < steps) rd(smem + ((s + 1) & 3) * 32768, 0, f0w, f0a);     // group 0 of the next sub-step
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1w[j], f1a[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (++kt == nk) {
            // epilogue of this tile (the next tile's first loads are already in flight).  The accumulators are
            // transposed through a per-wave LDS patch [32 m][64 n] fp16 (128-B rows, 16-B chunks XOR (row&7)) so that
            // every global store instruction writes 8 full 128-byte row segments instead of 32 16-byte pieces.
            int bm, bn;
            tile_mn(tile, bm, bn);
            char* ep = smem + 131072 + wave * 4096;
            const int nb0 = bn * 256 + wn * 64;
            f32x4 bq[2][4];
            if (HAS_BIAS) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *(const float4*)(bias + nb0 + j * 32 + 8 * g + 4 * hh);
                        bq[j][g][0] = b4.x; bq[j][g][1] = b4.y; bq[j][g][2] = b4.z; bq[j][g][3] = b4.w;
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        half4 o;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            float v = acc[j][i][4 * g + q4];
                            if (HAS_BIAS) v += bq[j][g][q4];
                            o[q4] = (half_t)act_apply(v, ACT);
                            acc[j][i][4 * g + q4] = 0.f;
                        }
                        *(half4*)(ep + r * 128 + (((j * 4 + g) ^ (r & 7)) << 4) + hh * 8) = o;
                    }
                }
                half8 rcur[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) rcur[p] = rpre[p];
                if (HAS_RES && i < 3) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        rpre[p] = *(const half8*)(R + ((size_t)bm * 256 + wm * 128 + (i + 1) * 32 + p * 8 + (lane >> 3)) * N + nb0 + (lane & 7) * 8);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int rr = p * 8 + (lane >> 3), cc = lane & 7;
                    half8 h = *(const half8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
                    const size_t off = ((size_t)bm * 256 + wm * 128 + i * 32 + rr) * N + nb0 + cc * 8;
                    if (HAS_RES) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) h[q] = (half_t)((float)h[q] + (float)rcur[p][q]);
                    }
                    if (!(xmode & 2)) *(half8*)(C + off) = h;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            kt = 0;
            tile += tstride;
            store_age = 0;
        }
    }
}

template <int ACT, bool B, bool RR>
static int launch256(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                     hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        SCD_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<ACT, B, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        attr = true;
    }
    const int tiles_n = N / 256, total = (M / 256) * tiles_n;
    const int grid = total < 256 ? (total >= 8 ? total / 8 * 8 : total) : 256;
    static const int xmode = getenv("SCD_GEMM_X") ? atoi(getenv("SCD_GEMM_X")) : 0;
    static const int ng_env = getenv("SCD_GEMM_NG") ? atoi(getenv("SCD_GEMM_NG")) : 0;
    // n-tiles per group: W panels of a group (ng*256*K*2 bytes) should stay inside one XCD's 4 MB L2; every extra group
    // re-reads the activations once.  Estimate the beyond-L2 traffic of each candidate and keep the cheapest.
    int ng = tiles_n;
    {
        const double a_bytes = 2.0 * M * (double)K, panel = 512.0 * K;
        double best = 1e300;
        for (int groups = 1; groups <= tiles_n; ++groups) {
            const int cand = (tiles_n + groups - 1) / groups;
            const double wg = cand * panel;
            const double rounds = (double)total / 256.0;                      // tile rounds per resident block
            const double w_traffic = wg <= 2.6e6 ? 8.0 * tiles_n * panel : 8.0 * rounds * wg;
            const double cost = groups * a_bytes + w_traffic;
            if (cost < best) { best = cost; ng = cand; }
        }
        if (ng_env > 0) ng = ng_env < tiles_n ? ng_env : tiles_n;
    }
    gemm256_kernel<ACT, B, RR><<<grid, 512, 163840, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng);
    return SCD_OK;
}
template <int ACT>
static int launch256_act(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                         hipStream_t st) {
    if (bias && R) return launch256<ACT, true, true>(A, W, bias, R, C, M, N, K, st);
    if (bias) return launch256<ACT, true, false>(A, W, bias, R, C, M, N, K, st);
    if (R) return launch256<ACT, false, true>(A, W, bias, R, C, M, N, K, st);
    return launch256<ACT, false, false>(A, W, bias, R, C, M, N, K, st);
}

template <int ACT>
static void launch_act(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                       hipStream_t st) {
    const int tiles_n = N / 128, total = (M / 128) * tiles_n;
    if (bias && R) gemm_f16_kernel<ACT, true, true><<<total, 256, 0, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total);
    else if (bias) gemm_f16_kernel<ACT, true, false><<<total, 256, 0, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total);
    else if (R) gemm_f16_kernel<ACT, false, true><<<total, 256, 0, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total);
    else gemm_f16_kernel<ACT, false, false><<<total, 256, 0, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total);
}

int scd_gemm_launch(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                    int act, hipStream_t st) {
    SCD_REQUIRE(A && W && C, "gemm: null operand");
    SCD_REQUIRE(M > 0 && M % 128 == 0 && N > 0 && N % 128 == 0 && K > 0 && K % 64 == 0 && M < (1ll << 31),
                "gemm: shape m=%lld n=%d k=%d must be multiples of 128/128/64", (long long)M, N, K);
    SCD_REQUIRE(C != (half_t*)A, "gemm: C must not alias A");
    static const bool force128 = getenv("SCD_GEMM128") != nullptr;
    if (M % 256 == 0 && N % 256 == 0 && !force128) {
        int rc;
        if (act == SCD_ACT_NONE) rc = launch256_act<SCD_ACT_NONE>(A, W, bias, R, C, (int)M, N, K, st);
        else if (act == SCD_ACT_QUICKGELU) rc = launch256_act<SCD_ACT_QUICKGELU>(A, W, bias, R, C, (int)M, N, K, st);
        else if (act == SCD_ACT_GELU) rc = launch256_act<SCD_ACT_GELU>(A, W, bias, R, C, (int)M, N, K, st);
        else SCD_REQUIRE(false, "gemm: bad activation %d", act);
        if (rc) return rc;
        SCD_LAUNCH_CHECK();
        return SCD_OK;
    }
    if (act == SCD_ACT_NONE) launch_act<SCD_ACT_NONE>(A, W, bias, R, C, (int)M, N, K, st);
    else if (act == SCD_ACT_QUICKGELU) launch_act<SCD_ACT_QUICKGELU>(A, W, bias, R, C, (int)M, N, K, st);
    else if (act == SCD_ACT_GELU) launch_act<SCD_ACT_GELU>(A, W, bias, R, C, (int)M, N, K, st);
    else SCD_REQUIRE(false, "gemm: bad activation %d", act);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_gemm_f16(scd_handle h, const void* A, const void* W, const float* bias, const void* residual, void* C,
                            int64_t m, int n, int k, int act, void* stream) {
    SCD_REQUIRE(h, "scd_gemm_f16: null handle");
    return scd_gemm_launch((const half_t*)A, (const half_t*)W, bias, (const half_t*)residual, (half_t*)C, m, n, k, act,
                           (hipStream_t)stream);
}
