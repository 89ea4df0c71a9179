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

// Exact (erf) GELU, nn.GELU of the DINO / GCD tower (gcd/models/vision_transformer.py:48-64), branch-free and in QuickGELU's shape:
//   x Phi(x) = x / (1 + exp(-L(x))),   L = logit Phi = log(Phi(x) / Phi(-x)) ~ x (c0 + c1 x^2 + ... + c6 x^12)
// L is odd, smooth and grows like x^2 / 2; the coefficients are the weighted minimax fit (linear program on |x| <= 9, weight
// d gelu / d L = x Phi (1 - Phi)) whose polynomial is monotone, so nothing is clamped: beyond the fitted range the sigmoid saturates
// faster than Phi does and the error stays below |x| Phi(-|x|) < 1e-8.  One v_exp_f32 and one v_rcp_f32 per value like QuickGELU plus
// seven packed fmas per PAIR; in fp32 arithmetic max |error| 6.0e-7 over [-12, 12], within 2 fp16 ulps of the correctly rounded value
// everywhere (0.5 x (1 + erff(x / sqrt 2)), torch's own form, is also 2: it cancels in the negative tail).  libdevice's erff is two
// divergent branches (|z| < 1, >= 1), ~36 VALU instructions per value for a wave that holds both sides: round 6 measured the fc1
// launch of the DINO tower with it (docs/design/round6.md).  SCD_GELU_K* = -c * log2(e) (the exponent goes to v_exp_f32 = 2^x).
#define SCD_GELU_K0 (-2.30220745f)
#define SCD_GELU_K1 (-0.104839488f)
#define SCD_GELU_K2 (9.69095658e-05f)
#define SCD_GELU_K3 (0.000158966471f)
#define SCD_GELU_K4 (-1.14071321e-05f)
#define SCD_GELU_K5 (3.83554556e-07f)
#define SCD_GELU_K6 (-5.212611e-09f)
__device__ __forceinline__ float gelu_erf(float x) {
    const float s = x * x;
    float p = fmaf(s, SCD_GELU_K6, SCD_GELU_K5);
    p = fmaf(p, s, SCD_GELU_K4);
    p = fmaf(p, s, SCD_GELU_K3);
    p = fmaf(p, s, SCD_GELU_K2);
    p = fmaf(p, s, SCD_GELU_K1);
    p = fmaf(p, s, SCD_GELU_K0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
typedef float float2p __attribute__((ext_vector_type(2)));
// the same arithmetic on a register pair (packed-fp32 multiplies / fmas; the fmas contract exactly as fmaf does above)
__device__ __forceinline__ float2p gelu_erf_pair(float2p x) {
#define SCD_P2(c) ((float2p){c, c})
    const float2p s = x * x;
    float2p p = __builtin_elementwise_fma(s, SCD_P2(SCD_GELU_K6), SCD_P2(SCD_GELU_K5));
    p = __builtin_elementwise_fma(p, s, SCD_P2(SCD_GELU_K4));
    p = __builtin_elementwise_fma(p, s, SCD_P2(SCD_GELU_K3));
    p = __builtin_elementwise_fma(p, s, SCD_P2(SCD_GELU_K2));
    p = __builtin_elementwise_fma(p, s, SCD_P2(SCD_GELU_K1));
    p = __builtin_elementwise_fma(p, s, SCD_P2(SCD_GELU_K0));
    float2p e = x * p;
    e.x = __builtin_amdgcn_exp2f(e.x); e.y = __builtin_amdgcn_exp2f(e.y);
    e += SCD_P2(1.0f);
    e.x = __builtin_amdgcn_rcpf(e.x); e.y = __builtin_amdgcn_rcpf(e.y);
    return x * e;
#undef SCD_P2
}

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == SCD_ACT_QUICKGELU) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    if (act == SCD_ACT_GELU) return gelu_erf(x);
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
// Persistent LDS-DMA kernel, block tile BM(m) x 256(n), k consumed in 32-deep sub-steps.
//   BM = 256: 8 waves (2 m x 4 n), 4-slot ring (128 KB) + 8 x 4 KB epilogue patches: one block per CU.
//   BM = 128: 4 waves (1 m x 4 n), 3-slot ring (72 KB)  + 4 x 2 KB epilogue patches: TWO independent blocks per CU, so one
//             block's barrier waits / epilogue VALU overlap the other block's MFMAs (the two waves sharing a SIMD belong to
//             different blocks and are not in lockstep).
// Wave tile 128(m) x 64(n) = 4x2 v_mfma_f32_32x32x16_f16 tiles (128 accumulator registers).  Operands are staged with
// global_load_lds (16 B/lane, LDS-DMA): LDS rows are 64 B (32 k) and the 16-B chunk index is XORed with (row>>2)&3 - on the
// per-lane SOURCE address for the DMA (its LDS destination is linear) and again on the ds_read_b128 side - which keeps the
// fragment reads bank-conflict-free.  A block walks its tiles as ONE flattened software pipeline over (tile, sub-step):
// the DMAs of sub-step s+NSLOT-1 are issued after the barrier in the middle of sub-step s (which also publishes slot
// s+1), waits are counted (s_waitcnt vmcnt(N)) so DMAs and the epilogue's stores stay in flight across barriers, and the
// fragment reads of an MFMA group are issued one group ahead.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ void __launch_bounds__(BM * 2, 2) gemm_dma_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                          const float* __restrict__ bias, const half_t* __restrict__ R,
                                                          half_t* __restrict__ C, int M, int N, int K, int tiles_n, int total_tiles,
                                                          int xmode, int ng) {
    constexpr int NW = BM / 32;                       // waves: 8 or 4
    constexpr int NSLOT = BM == 256 ? 4 : 3;
    constexpr int SLOT = BM * 64 + 16384;             // A sub-tile [BM][32] + W sub-tile [256][32], fp16
    constexpr int WG = 256 / NW / 16;                 // W DMAs per wave per sub-step (2 or 4); A: always 2
    constexpr int G = 2 + WG;                         // DMAs per wave per sub-step
    constexpr int PATCH = BM == 256 ? 4096 : 2048;    // per-wave epilogue patch
    constexpr int NST = BM == 256 ? 16 : 16;          // epilogue stores per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = BM == 256 ? wave >> 2 : 0, wn = wave & 3;
    const int nk = K >> 5;
    // tile order: n-tiles are grouped (ng per group) and a group is swept over ALL m-tiles before the next one, so the
    // group's W panels stay resident in the XCD's L2 while the activation panels stream through once per group
    const int tiles_m = total_tiles / tiles_n;
    const int per_group = tiles_m * ng;
    auto tile_mn = [&](int t, int& bm, int& bn) {
        const int g = t / per_group;
        const int local = t - g * per_group;
        const int n0 = g * ng;
        const int w = tiles_n - n0 < ng ? tiles_n - n0 : ng;
        bm = local / w;
        bn = n0 + local - bm * w;
    };
    // blocks b and b+8 share an XCD (round-robin dispatch): XCD x owns the contiguous tile range [x*T/8, (x+1)*T/8) and
    // its resident blocks take tiles round-robin, so at any time the blocks of one XCD hold consecutive tiles and sweep k
    // together: a k-slab is fetched into that XCD's L2 once and then hit by the blocks that share it.
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int xcd = blockIdx.x % nxcd, slot_id = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;
    const int c0 = (int)((long long)xcd * total_tiles / nxcd), c1 = (int)((long long)(xcd + 1) * total_tiles / nxcd);
    const int tb = c0 + slot_id;
    const int my_tiles = tb < c1 ? (c1 - tb + per_xcd - 1) / per_xcd : 0;
    const int steps = my_tiles * nk;
    if (steps <= 0) return;
    const int tstride = per_xcd;

    // per-lane source offsets: a DMA instruction covers 16 rows x 64 B; lane -> (row lane/4, physical chunk lane%4)
    const int lrow = lane >> 2, pc = lane & 3;
    int a_off[2], w_off[WG];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rowl = wave * 32 + p * 16 + lrow;
        a_off[p] = rowl * K + ((pc ^ ((rowl >> 2) & 3)) << 3);
    }
#pragma unroll
    for (int p = 0; p < WG; ++p) {
        const int rowl = wave * (16 * WG) + p * 16 + lrow;
        w_off[p] = rowl * K + ((pc ^ ((rowl >> 2) & 3)) << 3);
    }
    auto issue = [&](int tile, int kt, int slot) {
        int bm, bn;
        tile_mn(tile, bm, bn);
        if (xmode & 4) { bm = 0; bn = 0; }          // timing ablation: every block streams the same (L2-resident) panels
        const half_t* ga = A + (size_t)bm * BM * K + kt * 32;
        const half_t* gw = W + (size_t)bn * 256 * K + kt * 32;
        char* sa = smem + slot * SLOT + wave * 2048;
        char* sw = smem + slot * SLOT + BM * 64 + wave * (1024 * WG);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((const void*)(ga + a_off[p]), (lds_ptr_t)(sa + p * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < WG; ++p)
            __builtin_amdgcn_global_load_lds((const void*)(gw + w_off[p]), (lds_ptr_t)(sw + p * 1024), 16, 0, 0);
    };
    auto off32 = [](int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };

    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;

    int tile = tb, kt = 0;               // sub-step being computed
    int ntile = tile, nkt = 0;           // sub-step being loaded
#pragma unroll
    for (int pre = 0; pre < NSLOT - 1; ++pre) {
        if (pre < steps) issue(ntile, nkt, pre);
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
    }
    int store_age = 8;                   // sub-steps since the last epilogue issued its NST stores
    half8 rpre[4];                       // residual rows of the next epilogue block, prefetched under the MFMAs
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) rpre[p][q] = (half_t)0.f;

    // F0 feeds the first 8 MFMAs of a sub-step (k16 group 0), F1 the second 8 (group 1)
    half8 f0w[2], f0a[4], f1w[2], f1a[4];
    const int rw0 = wn * 64 + r, ra0 = wm * 128 + r;
    auto rd = [&](const char* slot, int k16, half8 (&fw)[2], half8 (&fa)[4]) {
        const char* la = slot;
        const char* lw = slot + BM * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[j] = *(const half8*)(lw + off32(rw0 + j * 32, 2 * k16 + hh));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const half8*)(la + off32(ra0 + i * 32, 2 * k16 + hh));
    };
    // slot 0 must be visible before the first fragment reads: at most NSLOT-2 younger sub-tiles may still fly
    if (steps >= NSLOT - 1) {
        if (NSLOT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (WG == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd(smem, 0, f0w, f0a);
    int cslot = 0;                       // ring slot of sub-step s
    for (int s = 0; s < steps; ++s) {
        const char* cur = smem + cslot * SLOT;
        const int nslot = cslot + 1 == NSLOT ? 0 : cslot + 1;
        if (HAS_RES && kt == nk - 2) {     // residual rows of epilogue block i=0: two sub-steps of MFMAs hide the latency
            int bm, bn;
            tile_mn(tile, bm, bn);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                rpre[p] = *(const half8*)(R + ((size_t)bm * BM + wm * 128 + p * 8 + (lane >> 3)) * N + bn * 256 + wn * 64 + (lane & 7) * 8);
        }
        rd(cur, 1, f1w, f1a);                                  // group 1 of this sub-step, overlaps the MFMAs below
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0w[j], f0a[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // publish slot s+1.  In flight at this point (oldest first): DMAs(s+1) .. DMAs(s+NSLOT-2), and the epilogue's
        // NST stores if they were issued after DMAs(s+1): everything younger than DMAs(s+1) may keep flying.
        if (s + 1 < steps) {
            if (s + NSLOT - 2 >= steps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (NSLOT == 4) {
                if (store_age < 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // 4 + 16
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                if (store_age < 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // 0 + 16
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        ++store_age;
        __builtin_amdgcn_s_barrier();    // slot s+1 visible to all waves; the slot of sub-step s-1 is free again
        asm volatile("" ::: "memory");
        if (s + NSLOT - 1 < steps && !((xmode & 1) && s >= 2)) {
            int ls = cslot + NSLOT - 1;
            if (ls >= NSLOT) ls -= NSLOT;
            issue(ntile, nkt, ls);
        }
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
        if (s + 1 < steps) rd(smem + nslot * SLOT, 0, f0w, f0a);     // group 0 of the next sub-step
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1w[j], f1a[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        cslot = nslot;
        if (++kt == nk) {
            // epilogue of this tile (the next tile's first DMAs are already in flight).  The accumulators are transposed
            // through a per-wave LDS patch so that global stores write whole row segments (128 B for BM=256, 64 B for
            // BM=128) instead of 32 16-byte pieces; the residual is added after the transpose from prefetched rows.
            int bm, bn;
            tile_mn(tile, bm, bn);
            char* ep = smem + NSLOT * SLOT + wave * PATCH;
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
                half8 rcur[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) rcur[p] = rpre[p];
                if (HAS_RES && i < 3) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        rpre[p] = *(const half8*)(R + ((size_t)bm * BM + wm * 128 + (i + 1) * 32 + p * 8 + (lane >> 3)) * N + nb0 + (lane & 7) * 8);
                }
                if (BM == 256) {
                    // patch [32 m][64 n] fp16, 128-B rows, 16-B chunk XOR (row&7)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
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
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int rr = p * 8 + (lane >> 3), cc = lane & 7;
                        half8 h = *(const half8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
                        const size_t off = ((size_t)bm * BM + wm * 128 + i * 32 + rr) * N + nb0 + cc * 8;
                        if (HAS_RES) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) h[q] = (half_t)((float)h[q] + (float)rcur[p][q]);
                        }
                        if (!(xmode & 2)) *(half8*)(C + off) = h;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    // 2 KB patch: one [32 m][32 n] MFMA tile at a time, 64-B rows, 16-B chunk XOR ((row>>1)&3)
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
                            *(half4*)(ep + r * 64 + ((g ^ ((r >> 1) & 3)) << 4) + hh * 8) = o;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const int rr = p * 16 + (lane >> 2), cc = lane & 3;
                            half8 h = *(const half8*)(ep + rr * 64 + ((cc ^ ((rr >> 1) & 3)) << 4));
                            const size_t off = ((size_t)bm * BM + i * 32 + rr) * N + nb0 + j * 32 + cc * 8;
                            if (HAS_RES) {
                                // rcur rows are laid out for the 128-B pattern (row p*8 + lane/8, chunk lane%8): re-read directly
                                const half8 rres = *(const half8*)(R + off);
#pragma unroll
                                for (int q = 0; q < 8; ++q) h[q] = (half_t)((float)h[q] + (float)rres[q]);
                            }
                            if (!(xmode & 2)) *(half8*)(C + off) = h;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
            }
            kt = 0;
            tile += tstride;
            store_age = 0;
        }
    }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#ifdef SCD_ABLATE   // A/B kernel (SCD_GEMM_MFMA=16): not in the default build
#include "ablate/gemm_dma16_kernel.h"
#endif

// ------------------------------------------------------------------------------------------------
// Four-wave kernel: 256x256 block tile, one wave per SIMD, each wave a 128(m) x 128(n) sub-tile whose 256 accumulator
// registers live in AGPRs; K advances in 64-deep chunks through two 64 KB LDS slots.
//
// Why 64-deep: a ring fill of 64-byte row segments moves half the bytes per L1 request - measured with
// tools/micro/dma_fill_bench.hip: 66 GB/s per CU against 129 GB/s per CU for 128-byte (whole-line) segments out of L2 -
// and at 128 FLOP per L2 byte that fill rate, not the MFMA pipe, bounded the 32-deep kernels above.
//
// LDS slot (c & 1): A part 256 rows x 128 B, W part 256 rows x 128 B; the 16-byte chunk lc of row r sits at physical chunk
// lc ^ ((r >> 1) & 7), which keeps both the DMA image (8 rows x 128 B per instruction, the swizzle applied to the per-lane
// source address) and the fragment ds_read_b128 (lane = row l&15, chunk (l>>4) + 4*khalf) conflict-free.
//
// Schedule of chunk c (two 32-deep sub-steps, 64 MFMAs per wave each, in 4 groups of 2 m-tiles x 8 n-tiles):
//   even: A fragments stream one group ahead; under the last group the W fragments and the first two A pairs of the odd
//         sub-step are read.
//   odd:  the remaining A pairs are read up front, so by the middle of the sub-step every read of chunk c has completed:
//         there - one barrier per chunk - the wave waits for chunk c+1 (issued two sub-steps earlier), and refills this
//         slot with chunk c+2; under the last group it reads the first fragments of chunk c+1.
// The hot loop is hand-scheduled in inline asm: MFMAs with AGPR accumulators ("a" constraints; left alone hipcc shuffles
// the accumulators between AGPRs and VGPRs around every group), C = 0 on a tile's first sub-step instead of zeroing,
// ds_read_b128 with counted lgkmcnt (for asm operands hipcc only emits lgkmcnt(0)); s_nop covers the MFMA -> v_accvgpr_read
// hazard the compiler cannot see.
#include "ablate/w4_knobs.h"      // the schedule constants of this kernel (shipped values) and the experiment hooks of rounds 3-5
#include "ablate/w4_probes.h"     // cycle-counter probes: real code with -DSCD_ABLATE, empty otherwise

// LN = 1: a LayerNorm over A's rows is folded into this GEMM.  W already carries gamma (W' = W * gamma[k]), bias carries
//         beta (b' = b + W beta), colsum[n] = sum_k W'[n][k], and ln_rs[m] = {rstd, -mean * rstd} of the raw input rows, which
//         ln_finish_kernel forms once per row from {sum_k x, sum_k x^2} (64-bit fixed point, units 2^-24 and 2^-20):
//         out = rstd * (acc - mean * colsum[n]) + b'[n], then the activation.  The raw x goes through the MFMAs unchanged.
// LN = 2: this GEMM produces the rows the NEXT LayerNorm normalises: the epilogue adds each row's {sum, sum of squares} of
//         the fp16 values it stores into ln_out[m] (64-bit integer atomics - order-independent, so the encoder stays
//         bit-reproducible - one 64-row instruction per 64 rows).
// IMG: A is not a row-major [M, K] matrix but the fp16 image batch [B, 3, img_w, img_w] itself, and row m = patch (b, py, px) of the
//      16 x 16 / stride-16 patch convolution (K = 768, k = c * 256 + i * 16 + j as in conv1.weight.reshape(width, -1)): the ring fills
//      gather the patch rows from the image - a 64-deep chunk is four image rows of one channel, a lane's 16-byte piece half a patch row,
//      and neighbouring patches continue each other's image rows, so the fills still move whole cache lines.  Per-lane byte offsets of
//      the tile's eight fill instructions (the patch's place in the image; recomputed per tile) + a scalar offset per chunk (channel and
//      image-row group).  Replaces im2col_kernel and its 1.2-GB round trip per 3,990-image launch (round 6; main_unsup.py:114-147).
template <int NT, int ACT, bool HAS_BIAS, bool HAS_RES, int LN, bool NTS, bool IMG = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_w4_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W, const float* __restrict__ bias,
               const half_t* __restrict__ R, half_t* __restrict__ C, int M, int N, int K, int tiles_n, int total_tiles,
               int xmode_in, int ng, const float2* __restrict__ ln_rs, const float* __restrict__ ln_colsum,
               long long* __restrict__ ln_out, int stagger_in, int img_w = 0, int img_gp = 0, int img_rows = 0) {
    static_assert(NT == 8, "wave tile is 128 x 128");
#ifdef SCD_ABLATE
    const int xmode = xmode_in, stagger = stagger_in;
#else
    // the shipped build keeps the one switch it uses (512: non-temporal C stores): the timing ablations, the cycle counters and the
    // start-up stagger exist in the -DSCD_ABLATE build only.  With them a runtime possibility, their state (seven 64-bit counters, the
    // probe registers) stayed live through the kernel and the residual variant spilled 11 VGPRs / 66 SGPRs (round 4)
    // Non-temporal C stores: for the residual variants (proj, fc2) a template parameter (NTS) - as a runtime test in front of each of
    // the tile's 32 store instructions they cost 61 branches with their exec-mask bookkeeping per tile, and with the test gone the
    // variant needs 196 VGPRs and spills nothing; the LayerNorm-folded variants (QKV, fc1) keep the runtime test: without it the
    // register allocator ends at 256 VGPRs with spills reloaded once per tile (hipcc 7.2, -Rpass-analysis=kernel-resource-usage)
    const int xmode = NTS ? 512 : 0, stagger = 0; (void)xmode_in;
    (void)stagger_in;
#endif
    constexpr bool DMA_SPLIT = W4_DMA_SPLIT;
    constexpr bool DEFER_ST = HAS_RES && W4_DEFER_STORES;   // residual variants: all stores after the last residual load
    constexpr bool LATE_BAR = W4_LATE_BAR;     // the chunk's barrier after the odd sub-step's first L0 MFMAs instead of before them
    constexpr int L0 = W4_LATE_TM * 8, LS = (64 - L0) / 8;   // first hooked MFMA; MFMAs per ring fill   // true: A part of a refill in the odd sub-step, W part in the next even one
    constexpr int BM = 256, BN = 256, SLOT = 65536, WPART = 32768, EPI = 2 * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, q16 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lane_el = (unsigned)(q16 * N + c16 * 8);   // element offset of this lane's row / 16-byte piece inside a 4-row group
    const int nkc = K >> 6;                       // 64-deep chunks per tile
    const int tiles_m = total_tiles / tiles_n;
    const int per_group = tiles_m * ng;
    // Tile order: n-groups of ng tile columns, row-major inside a group (see choose_ng).  A block walks its tiles with a
    // stride, so (bm, bn) is carried incrementally - the integer divisions of the closed form, executed by a lone wave per
    // SIMD between two MFMA blocks, cost several hundred idle matrix-pipe cycles per chunk.
    struct TileIt { int t, bm, bnl, n0, w, q, r; };
    auto it_init = [&](TileIt& it, int t) {
        it.t = t;
        const int g = t / per_group;
        const int local = t - g * per_group;
        it.n0 = g * ng;
        it.w = tiles_n - it.n0 < ng ? tiles_n - it.n0 : ng;
        it.bm = local / it.w;
        it.bnl = local - it.bm * it.w;
    };
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int xcd = blockIdx.x % nxcd, slot_id = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;
    const int c0 = (int)((long long)xcd * total_tiles / nxcd), c1 = (int)((long long)(xcd + 1) * total_tiles / nxcd);
    const int tb = c0 + slot_id;
    const int my_tiles = tb < c1 ? (c1 - tb + per_xcd - 1) / per_xcd : 0;
    const int chunks = my_tiles * nkc;
    if (chunks <= 0) return;
    const int tstride = per_xcd;
    if (stagger > 0) {   // phase-shift the blocks of an XCD (see launch_w4): wall_clock64 ticks at 100 MHz
        // stagger >= 2^20: whole XCDs are shifted against each other (their 32 blocks stay in lock-step and keep sharing panels in
        // the XCD's L2) instead of the blocks inside an XCD
        const long long target = wall_clock64() + (stagger >= (1 << 20) ? (long long)xcd * (stagger - (1 << 20)) : (long long)(slot_id & 7) * stagger);
        while (wall_clock64() < target) __builtin_amdgcn_s_sleep(32);
    }
    auto it_step = [&](TileIt& it) {       // t += tstride
        it.t += tstride;
        it.bnl += it.r;
        it.bm += it.q;
        if (it.bnl >= it.w) { it.bnl -= it.w; ++it.bm; }
        if (it.bm >= tiles_m) {            // crossed into the next n-group (rare): closed form
            it_init(it, it.t);
            it.q = tstride / it.w;
            it.r = tstride - it.q * it.w;
        }
    };

    // DMA: instruction p of a wave covers rows wave*64 + p*8 + (lane>>3), lane&7 = physical chunk
    const int drow = lane >> 3, dpc = lane & 7;
    int g_off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int rowl = wave * 64 + par * 8 + drow;
        g_off[par] = rowl * K + ((dpc ^ ((rowl >> 1) & 7)) << 3);
    }
    const int k16 = 16 * K;
    const half_t *ga = A, *gw = W;      // panel pointers of the chunk being issued
    // IMG: element offset of chunk kc inside a patch's image window = channel kc / 4, image rows 4 (kc % 4) .. + 3
    auto img_chunk = [&](int kc) { return (kc >> 2) * img_w * img_w + (kc & 3) * 4 * img_w; };
    unsigned vimg[8];                   // IMG: per-lane byte offsets of the tile's eight A fills (img_rows = valid patch rows)
    auto img_offsets = [&](int bm) {
        const int np = img_gp * img_gp;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rowl = wave * 64 + (p & 1) * 8 + drow;
            int row = bm * BM + rowl + (p >> 1) * 16;
            row = row < img_rows ? row : img_rows - 1;      // padding rows of the last tile: any valid patch (their output is not read)
            const int b = row / np, pp = row - b * np, py = pp / img_gp, px = pp - py * img_gp;
            const int lp = dpc ^ ((rowl >> 1) & 7);         // logical 16-byte piece of the 128-byte LDS row: image row lp / 2, half lp % 2
            vimg[p] = 2u * (unsigned)(b * 3 * img_w * img_w + (py * 16 + (lp >> 1)) * img_w + px * 16 + (lp & 1) * 8);
        }
    };
    auto chunk_ptrs = [&](const TileIt& it, int kc) {
        int bm = it.bm, bn = it.n0 + it.bnl;
        if (xmode & 4) { bm = 0; bn = 0; }
        ga = IMG ? A + img_chunk(kc) : A + (size_t)bm * BM * K + kc * 64;
        gw = W + (size_t)bn * BN * K + kc * 64;
        // the per-lane offsets change where the scalar pointer does - with the first chunk of a tile, never earlier: the A fills of the
        // previous tile's last chunk are still being issued when the tile iterator has already moved on
        if constexpr (IMG) { if (kc == 0) img_offsets(bm); }
    };
    // one ring-fill instruction (p = 0..7 of the A part or of the W part): scalar panel base + per-lane byte offset, so it
    // costs the wave no VALU work between the MFMAs it is interleaved with.  M0 carries the LDS destination.
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_lds = sbase + wave * 8192;
    const unsigned voff0 = (unsigned)g_off[0] * 2, voff1 = (unsigned)g_off[1] * 2;
    // M0 = the slot's LDS base (one SGPR per slot, formed once) + an immediate: one SALU instruction per fill.  (As "s"(base + slot
    // offset + 1024 p) hipcc rebuilt the address with xor / add / mov + its own s_nop in front of every fill: five scalar
    // instructions in the lone wave's stream per fill, sixteen fills per chunk.)
#define W4_DMA(BASE, P, SLOTLDS, PART, MOD)                                                                      \
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD                        \
                 ::"s"(SLOTLDS), "v"(((P) & 1) ? voff1 : voff0), "s"((BASE) + ((P) >> 1) * k16), "n"((PART) + (P) * 1024) : "memory", "scc")
#if W4_RES_NT
#define W4_RLOAD(P) __builtin_nontemporal_load((const half8*)(P))
#else
#define W4_RLOAD(P) (*(const half8*)(P))
#endif
    auto issue_a = [&](int p, int slot) {
        if constexpr (IMG) {
            asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         ::"s"(dma_lds + slot * SLOT), "v"(vimg[p]), "s"(ga), "n"(p * 1024) : "memory", "scc");
        } else if constexpr (W4_A_NT_LN1 && LN == 1) { W4_DMA(ga, p, dma_lds + slot * SLOT, 0, " nt"); }
        else { W4_DMA(ga, p, dma_lds + slot * SLOT, 0, W4_A_MOD); }
    };
    auto issue_w = [&](int p, int slot) { W4_DMA(gw, p, dma_lds + slot * SLOT, WPART, W4_W_MOD); };
    auto issue = [&](const TileIt& it, int kc, int slot) {
        chunk_ptrs(it, kc);
#pragma unroll
        for (int p = 0; p < 8; ++p) issue_a(p, slot);
#pragma unroll
        for (int p = 0; p < W4_TNW; ++p) issue_w(p, slot);
    };
    // fragment addresses: tile t adds t * 2048; k-half j uses chunk (q16 + 4j) ^ sw
    const int fsw = (c16 >> 1) & 7;
    const unsigned fa0 = sbase + (wm * 128 + c16) * 128 + ((q16 ^ fsw) << 4);
    const unsigned fa1 = sbase + (wm * 128 + c16) * 128 + (((q16 + 4) ^ fsw) << 4);
    const unsigned fw0 = sbase + WPART + (wn * 128 + c16) * 128 + ((q16 ^ fsw) << 4);
    const unsigned fw1 = sbase + WPART + (wn * 128 + c16) * 128 + (((q16 + 4) ^ fsw) << 4);

    // prologue: chunk 0 whole, chunk 1's A part (its W part is issued under the first even sub-step, see W4_H_EVEN)
    TileIt cit, nit;                      // tile being computed / tile whose chunks are being issued
    it_init(cit, tb);
    cit.q = tstride / cit.w;
    cit.r = tstride - cit.q * cit.w;
    nit = cit;
    int nkt = 0, ntiles = 0;              // ntiles: tiles completely issued
    // panel bases of the tile being issued (nit): recomputed when that tile changes, so that a chunk's pointers are base + 128 kc bytes
    // (the closed form per chunk - two 64-bit multiply-adds - sat in the lone wave's instruction stream between two MFMA groups)
    const half_t *ga_t = A, *gw_t = W;
    auto tile_ptrs = [&]() {
        int bm = nit.bm, bn = nit.n0 + nit.bnl;
        if (xmode & 4) { bm = 0; bn = 0; }
        ga_t = IMG ? A : A + (size_t)bm * BM * K;
        gw_t = W + (size_t)bn * BN * K;
    };
    tile_ptrs();
    auto issue_advance = [&]() {
        if (++nkt == nkc) {
            nkt = 0;
            if (++ntiles < my_tiles) { it_step(nit); tile_ptrs(); }
        }
    };
    issue(nit, nkt, 0);
    issue_advance();
    if (chunks > 1) chunk_ptrs(nit, nkt); else chunk_ptrs(cit, 0);
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_a(p, 1);
    if (!DMA_SPLIT) {
#pragma unroll
        for (int p = 0; p < W4_TNW; ++p) issue_w(p, 1);
    } else if (W4_WSPLIT) {
#pragma unroll
        for (int p = 0; p < W4_WSPLIT; ++p) issue_w(p, 1);     // (what an odd sub-step would have issued of chunk 1's W part)
    }
    issue_advance();
    constexpr int RD = LN == 2 ? W4_RD_LN2 : 3;   // residual rows in flight (m-tiles); the stats epilogue needs the registers
    half8 rq[RD][4];
#pragma unroll
    for (int e = 0; e < RD; ++e)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 8; ++q) rq[e][p][q] = (half_t)0.f;

    // timing probe (build with -DW4_PROBE_VALU=n): n independent VALU instructions behind every MFMA - what an epilogue dealt over
    // the next tile's MFMA stream would add to the wave's instruction stream
    float pv0 = 1.f + lane, pv1 = 2.f + lane, pv2 = 3.f + lane, pv3 = 0.5f;
#if W4_PROBE_VALU == 0
#define W4_PROBE()
#elif W4_PROBE_VALU == 1
#define W4_PROBE() asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pv0));
#elif W4_PROBE_VALU == 2
#define W4_PROBE() asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(pv0), "+v"(pv1));
#elif W4_PROBE_VALU == 3
#define W4_PROBE() asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2" : "+v"(pv0), "+v"(pv1), "+v"(pv2));
#else
#define W4_PROBE() asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_exp_f32 %1, %1\n\tv_fma_f32 %2, %2, %2, %2" : "+v"(pv0), "+v"(pv3), "+v"(pv2));
#endif
    half8 fwA[8], fwB[8], faA[8], faB[8];   // W / A fragments of the even (A) and odd (B) sub-step
#define W4_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define W4_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N))
    if (DMA_SPLIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + W4_WSPLIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        W4_RD(fwA[t], fw0, t * 2048);
        W4_RD(faA[t], fa0, t * 2048);
    }
    int cslot = 0, g = 0;   // g: chunk counter over all of this block's tiles
    // one 32-deep sub-step: 64 MFMAs (8 m-tiles x 8 n-tiles); HOOK(i) runs after MFMA i = 0..63.  Everything else the wave
    // has to issue - the fragment reads of the NEXT sub-step and the ring refill - is spread between the MFMAs through the
    // hook: issued back to back they hold up the wave's in-order instruction stream, and with it the matrix pipe, for as
    // long as the LDS / texture queues take to accept them.
#define W4_SUB(FW, FA, Z, HOOK) W4_SUB_RANGE(FW, FA, Z, HOOK, 0, 8)
#define W4_SUB_RANGE(FW, FA, Z, HOOK, TM0, TM1)                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int o_ = (TM0); o_ < (TM1); ++o_) _Pragma("unroll") for (int i_ = 0; i_ < W4_TNW; ++i_) {     \
        const int tm = W4_TN_MAJOR ? i_ : o_, tn = W4_TN_MAJOR ? o_ : i_;                                        \
        if (Z) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[tn][tm]) : "v"(FW[tn]), "v"(FA[tm]));         \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[tn][tm]) : "v"(FW[tn]), "v"(FA[tm]));          \
        HOOK(o_ * W4_TNW + i_)                                                                                   \
        W4_PROBE()                                                                                               \
    }                                                                                                            \
    __builtin_amdgcn_s_setprio(0);
#define W4_H_NONE(i)
    // even sub-step: reads the odd sub-step's fragments (k-half 1 of the same slot) under MFMAs 0..47 and issues the W part
    // of the refill that the previous odd sub-step began (the other slot), one fill per four MFMAs of the first half
#if W4_TNW == 8
#define W4_H_EVEN(i)                                                                                             \
    if (DMA_SPLIT && !W4_WSPLIT && (i) < 32 && ((i) & 3) == 2) issue_w(((i) >> 2) & 7, cslot ^ 1);               \
    if (DMA_SPLIT && W4_WSPLIT && (i) < 4 * (8 - W4_WSPLIT) && ((i) & 3) == 2) issue_w((W4_WSPLIT + ((i) >> 2)) & 7, cslot ^ 1); \
    if ((i) < 48 && (i) % 3 == 0) W4_RD(fwB[((i) / 3) & 7], fw1 + so, (((i) / 3) & 7) * 2048);                   \
    if ((i) < 48 && (i) % 3 == 1) W4_RD(faB[((i) / 3) & 7], fa1 + so, (((i) / 3) & 7) * 2048);
    // odd sub-step: reads the next chunk's first fragments (other slot) under MFMAs 0..47 and starts refilling this slot with
    // chunk c+2: the A part (the one that can miss L2), one fill per eight MFMAs.  Measured: a wave is held ~64 cycles per
    // fill while the CU's texture path (64 B/clk) takes the four waves' 1 KB instructions, so 16 fills inside one sub-step
    // doubled its length; spread over two sub-steps they fit under the MFMAs.
#define W4_H_ODD(i)                                                                                              \
    if (!LATE_BAR && (i) < 48 && (i) % 3 == 0) W4_RD(fwA[((i) / 3) & 7], fw0 + no, (((i) / 3) & 7) * 2048);      \
    if (!LATE_BAR && (i) < 48 && (i) % 3 == 1) W4_RD(faA[((i) / 3) & 7], fa0 + no, (((i) / 3) & 7) * 2048);      \
    if (!LATE_BAR && DMA_SPLIT && ((i) & 7) == 4) issue_a(((i) >> 3) & 7, cslot);                                \
    if (!DMA_SPLIT && ((i) & 3) == 2) { if ((i) < 32) issue_a(((i) >> 2) & 7, cslot); else issue_w(((i) >> 2) & 7, cslot); } \
    /* LATE_BAR: the barrier sits after MFMA L0-1, so everything that needs it is packed under MFMAs L0..63 */    \
    if (LATE_BAR && (i) >= L0 && (((i) - L0) & 1) == 0 && (((i) - L0) >> 1) < 8) W4_RD(fwA[(((i) - L0) >> 1) & 7], fw0 + no, ((((i) - L0) >> 1) & 7) * 2048); \
    if (LATE_BAR && (i) >= L0 && (((i) - L0) & 1) == 0 && (((i) - L0) >> 1) >= 8 && (((i) - L0) >> 1) < 16) W4_RD(faA[(((i) - L0) >> 1) & 7], fa0 + no, ((((i) - L0) >> 1) & 7) * 2048); \
    if (LATE_BAR && DMA_SPLIT && !W4_WSPLIT && (i) >= L0 && ((i) - L0) % LS == LS / 2 && ((i) - L0) / LS < 8) issue_a((((i) - L0) / LS) & 7, cslot); \
    if (LATE_BAR && DMA_SPLIT && W4_WSPLIT && (i) >= L0 && (((i) - L0) & 3) == 1 && ((i) - L0) / 4 < 8) issue_a((((i) - L0) / 4) & 7, cslot); \
    if (LATE_BAR && DMA_SPLIT && W4_WSPLIT && (i) >= L0 + 32 && (((i) - L0) & 3) == 1 && ((i) - L0 - 32) / 4 < W4_WSPLIT) issue_w((((i) - L0 - 32) / 4) & 7, cslot);
#else
    // 32 MFMAs per sub-step: 12 fragment reads under the first 24, the W part of the refill (4 instructions per wave) every 8th
#define W4_H_EVEN(i)                                                                                             \
    if (DMA_SPLIT && ((i) & 7) == 2) issue_w(((i) >> 3) & 3, cslot ^ 1);                                         \
    if ((i) < 24 && ((i) & 1) == 0 && ((i) >> 1) < 4) W4_RD(fwB[((i) >> 1) & 3], fw1 + so, (((i) >> 1) & 3) * 2048); \
    if ((i) < 24 && ((i) & 1) == 0 && ((i) >> 1) >= 4) W4_RD(faB[(((i) >> 1) - 4) & 7], fa1 + so, ((((i) >> 1) - 4) & 7) * 2048);
#define W4_H_ODD(i)                                                                                              \
    if ((i) >= 8 && (((i) - 8) & 1) == 0 && (((i) - 8) >> 1) < 4) W4_RD(fwA[(((i) - 8) >> 1) & 3], fw0 + no, ((((i) - 8) >> 1) & 3) * 2048); \
    if ((i) >= 8 && (((i) - 8) & 1) == 0 && (((i) - 8) >> 1) >= 4) W4_RD(faA[((((i) - 8) >> 1) - 4) & 7], fa0 + no, (((((i) - 8) >> 1) - 4) & 7) * 2048); \
    if (DMA_SPLIT && (i) >= 8 && ((i) - 8) % 3 == 1 && ((i) - 8) / 3 < 8) issue_a((((i) - 8) / 3) & 7, cslot);
#endif
#define W4_EVEN(Z)                                                                                               \
    {                                                                                                            \
        const unsigned so = cslot * SLOT;                                                                        \
        W4_LGKM(0);                                                                                              \
        W4_PROBE_SUB(t_odd)                                                                                      \
        W4_SUB(fwA, faA, Z, W4_H_EVEN)                                                                           \
    }
#define W4_ODD()                                                                                                 \
    {                                                                                                            \
        const unsigned no = (cslot ^ 1) * SLOT; /* past the last chunk: a stale slot, values unused */           \
        /* all of this wave's reads of the chunk have completed: after the barrier the slot can be refilled */   \
        W4_LGKM(0);                                                                                              \
        W4_PROBE_SUB(t_even)                                                                                     \
        /* chunk g+1: A part issued under the previous odd sub-step, W part under the even one just finished; an epilogue's  \
           stores, if any, sit between the two in the in-order queue, so this waits for them as well */                 \
        if (LATE_BAR) { W4_SUB_RANGE(fwB, faB, 0, W4_H_NONE, 0, W4_LATE_TM) }   /* MFMAs that need nothing new */    \
        if (!(xmode & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
        asm volatile("" ::: "memory");                                                                           \
        W4_PROBE_BAR()                                                                                           \
        tile_first = false;                                                                                      \
        /* past this block's last chunk the refill re-reads the current tile's first chunk into the free slot: no branch  \
           around the asm groups (a diamond makes hipcc copy accumulators between paths), and nothing reads the slot */   \
        if (g + 2 < chunks) {                                                                                    \
            ga = ga_t + (IMG ? img_chunk(nkt) : nkt * 64);                                                       \
            gw = gw_t + nkt * 64;                                                                                \
            if constexpr (IMG) { if (nkt == 0) img_offsets(nit.bm); }                                            \
        } else chunk_ptrs(cit, 0);                                                                               \
        issue_advance();                                                                                         \
        if (LATE_BAR) { W4_SUB_RANGE(fwB, faB, 0, W4_H_ODD, W4_LATE_TM, 8) } else { W4_SUB(fwB, faB, 0, W4_H_ODD) }  \
        cslot ^= 1;                                                                                              \
        ++g;                                                                                                     \
    }
    W4_PROBE_DECL()
    bool tile_first = true;      // (-DSCD_ABLATE, SCD_GEMM_X & 2048: the wait + barrier counter takes a tile's FIRST chunk only - the one behind the previous tile's stores)
    for (int ti = 0; ti < my_tiles; ++ti) {
        tile_first = true;
        const int bm = cit.bm, bn = cit.n0 + cit.bnl;
        const int nb0 = bn * BN + wn * 128;
        // C / R addresses of the epilogue = a wave-uniform tile base (SGPRs) + ONE 32-bit per-lane element offset (row q16 of the
        // wave's rows, column piece c16): as 64-bit per-lane pointers the loop-invariant parts were hoisted into VGPR pairs that
        // lived through the kernel, and the residual variant spilled ten of them (round 4)
        const size_t tile_el = ((size_t)((xmode & 1024) ? 0 : bm) * BM + wm * 128) * N + ((xmode & 1024) ? wn * 128 : nb0);
        const half_t* const Rt = HAS_RES ? R + tile_el : nullptr;
        half_t* const Ct = C + tile_el;
        W4_PROBE_MARK(t0)
        f32x4v acc[8][8];   // [tn][tm]; first written by the C = 0 MFMAs of the first sub-step
        f32x4v bq[8], sq[8];
        float2 lrs[8];
        float keep1[2] = {0.f, 0.f}, keep2[2] = {0.f, 0.f};   // LN = 2: this lane's rows 4*c16 + 64*j + q16
        // bias and the first residual rows are fetched one chunk before the tile ends: a plain load issued in the epilogue
        // would sit behind the ring refills in the (in-order) vmcnt queue and stall on them.
#define W4_PRE()                                                                                                 \
    {                                                                                                            \
        if (HAS_RES) {                                                                                           \
            _Pragma("unroll") for (int e = 0; e < RD - 1; ++e) _Pragma("unroll") for (int p = 0; p < 4; ++p) rq[e][p] = \
                W4_RLOAD(Rt + (size_t)((e * 16 + p * 4) * N) + lane_el);                                       \
        }                                                                                                        \
        if (HAS_BIAS && W4_ABL_PRE) {                                                                            \
            _Pragma("unroll") for (int tn = 0; tn < 8; ++tn) { bq[tn][0] = 0.f; bq[tn][1] = 0.f; bq[tn][2] = 0.f; bq[tn][3] = 0.f; sq[tn] = bq[tn]; } \
        }                                                                                                        \
        if (HAS_BIAS && !W4_ABL_PRE) {                                                                           \
            _Pragma("unroll") for (int tn = 0; tn < 8; ++tn) {                                                   \
                const float4 b4 = *(const float4*)(bias + nb0 + tn * 16 + q16 * 4);                              \
                bq[tn][0] = b4.x; bq[tn][1] = b4.y; bq[tn][2] = b4.z; bq[tn][3] = b4.w;                          \
            }                                                                                                    \
        }                                                                                                        \
        if (LN == 1 && W4_LN_ABL < 2 && !W4_ABL_PRE) {                                                                          \
            _Pragma("unroll") for (int tn = 0; tn < 8; ++tn) {                                                   \
                const float4 c4 = *(const float4*)(ln_colsum + nb0 + tn * 16 + q16 * 4);                         \
                sq[tn][0] = c4.x; sq[tn][1] = c4.y; sq[tn][2] = c4.z; sq[tn][3] = c4.w;                          \
            }                                                                                                    \
            if (W4_LN_ABL == 0) {                                                                                \
                _Pragma("unroll") for (int tm = 0; tm < 8; ++tm)                                                 \
                    lrs[tm] = ln_rs[(size_t)bm * BM + wm * 128 + tm * 16 + c16];                                 \
            }                                                                                                    \
        }                                                                                                        \
    }
        if (nkc == 1) W4_PRE()
        W4_EVEN(1)
        W4_ODD()
        for (int kc = 1; kc < nkc; ++kc) {
            if (kc == nkc - 1) W4_PRE()
            W4_EVEN(0)
            W4_ODD()
        }
#undef W4_PRE
        W4_LGKM(0);                                            // next tile's first fragments (read under the last group)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA -> accumulator read
        W4_PROBE_MARK(t1)
        if (!(xmode & 16)) {
            // epilogue, one 16-row m-tile at a time through a per-wave LDS patch [16 m][128 n] fp16 (256-B rows, chunk XOR row):
            // lane (c16 = m, q16) holds n = tn*16 + q16*4 + 0..3; rows leave as whole 256-byte segments.  No lgkmcnt waits
            // between the patch writes and reads: one wave's LDS operations execute in order.  Residual rows are fetched two
            // m-tiles ahead (ring of three).
            char* ep = smem + EPI + wave * 4096;
            // software pipeline over the 8 m-tiles: the patch read of m-tile tm-1 is in flight while m-tile tm is converted,
            // and its rows are stored after that (one wave per SIMD: nothing else would cover the LDS round trip)
            half8 hvb[4];
            unsigned stash[8][16];
            float rstd_a[8], nmr_a[8];
            if (LN == 1 && W4_LN_ABL) {
#pragma unroll
                for (int tm = 0; tm < 8; ++tm) { rstd_a[tm] = 1.f; nmr_a[tm] = 0.f; }
            } else if (LN == 1) {
                // {rstd, -mean * rstd} of the row, formed once per row by ln_finish_kernel from the fixed-point row sums (as eight
                // 64-bit loads + conversions per lane and tile, in front of every epilogue, they cost 2 % of an encode: round 5)
#pragma unroll
                for (int tm = 0; tm < 8; ++tm) { rstd_a[tm] = lrs[tm].x; nmr_a[tm] = lrs[tm].y; }
            }
#pragma unroll
            for (int tm = 0; tm <= 8; ++tm) {
                if (tm < 8) {
                    float rstd = 1.f, nmr = 0.f;
                    if (LN == 1) { rstd = rstd_a[tm]; nmr = nmr_a[tm]; }
#pragma unroll
                    for (int tn = 0; tn < 8; ++tn) {
                        // four values as two register pairs, so that bias / LayerNorm terms are packed-fp32 operations and the
                        // fp16 conversion is v_cvt_pk_f16_f32 (8-10 VALU instructions per 4 values instead of 14)
                        float2v v01, v23;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v01.x) : "a"(acc[tn][tm][0]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v01.y) : "a"(acc[tn][tm][1]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v23.x) : "a"(acc[tn][tm][2]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v23.y) : "a"(acc[tn][tm][3]));
                        if (LN == 1 && W4_LN_ABL < 2) {
                            const float2v r2 = {rstd, rstd}, m2 = {nmr, nmr};
                            float2v t01 = m2 * sq[tn].lo, t23 = m2 * sq[tn].hi;
                            if (HAS_BIAS) { t01 += bq[tn].lo; t23 += bq[tn].hi; }
                            v01 = v01 * r2 + t01;
                            v23 = v23 * r2 + t23;
                        } else if (HAS_BIAS) {
                            v01 += bq[tn].lo;
                            v23 += bq[tn].hi;
                        }
                        if (ACT == SCD_ACT_QUICKGELU) {
                            // x * sigmoid(1.702 x) = x * rcp(1 + 2^(-1.702 log2(e) x)): the three non-transcendental steps packed
                            const float2v c2 = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};
                            const float2v one2 = {1.f, 1.f};
                            float2v e01 = v01 * c2, e23 = v23 * c2;
                            e01.x = __builtin_amdgcn_exp2f(e01.x); e01.y = __builtin_amdgcn_exp2f(e01.y);
                            e23.x = __builtin_amdgcn_exp2f(e23.x); e23.y = __builtin_amdgcn_exp2f(e23.y);
                            e01 += one2; e23 += one2;
                            e01.x = __builtin_amdgcn_rcpf(e01.x); e01.y = __builtin_amdgcn_rcpf(e01.y);
                            e23.x = __builtin_amdgcn_rcpf(e23.x); e23.y = __builtin_amdgcn_rcpf(e23.y);
                            v01 *= e01; v23 *= e23;
                        } else if (ACT == SCD_ACT_GELU) {
                            v01 = gelu_erf_pair(v01); v23 = gelu_erf_pair(v23);
                        }
                        const half2v h01 = __builtin_convertvector(v01, half2v), h23 = __builtin_convertvector(v23, half2v);
                        const half4 o = {h01.x, h01.y, h23.x, h23.y};
                        *(half4*)(ep + c16 * 256 + (((tn * 2 + (q16 >> 1)) ^ c16) << 4) + (q16 & 1) * 8) = o;
                    }
                }
                if (tm > 0) {
                    const int ts = tm - 1;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int rr = p * 4 + q16;
                        half8 hv = hvb[p];
                        half_t* const crow = Ct + (size_t)((ts * 16 + p * 4) * N);          // wave-uniform (residual variants)
                        (void)rr;
                        if (HAS_RES) hv = hv + rq[ts % RD][p];   // fp16 add of two fp16 values: the same rounding as via fp32
                        if (LN == 2 && !W4_ABL_STATS) {
                            // row sums of the stored fp16 values over this wave's 128 columns: 8 values in-lane, then the 16 lanes
                            // (c16) that share row rr by DPP (rotations by 8 and 4 inside the 16-lane row, then inside the quad)
                            float s1 = 0.f, s2 = 0.f;
                            const half2v ones = {(half_t)1.f, (half_t)1.f};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {   // v_dot2_f32_f16: exact fp16 products, fp32 accumulation
                                const half2v pr = {hv[2 * q], hv[2 * q + 1]};
                                s1 = __builtin_amdgcn_fdot2(pr, ones, s1, false);
                                s2 = __builtin_amdgcn_fdot2(pr, pr, s2, false);
                            }
#define W4_DPP_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xF, 0xF, true))
                            W4_DPP_ADD(s1, 0x128); W4_DPP_ADD(s2, 0x128);   // row_ror:8
                            W4_DPP_ADD(s1, 0x124); W4_DPP_ADD(s2, 0x124);   // row_ror:4
                            W4_DPP_ADD(s1, 0x4E); W4_DPP_ADD(s2, 0x4E);     // quad_perm [2,3,0,1]
                            W4_DPP_ADD(s1, 0xB1); W4_DPP_ADD(s2, 0xB1);     // quad_perm [1,0,3,2]
#undef W4_DPP_ADD
                            // every lane of the row group now holds the totals; lane c16 keeps those of (tm*4+p) == c16 (mod 16)
                            const bool mine = ((ts * 4 + p) & 15) == c16;
                            keep1[ts >> 2] = mine ? s1 : keep1[ts >> 2];
                            keep2[ts >> 2] = mine ? s2 : keep2[ts >> 2];
                        }
                        // a large C streams past L2 ("nt"): written normally, each round of tiles pushes 32 MB of dirty lines
                        // through the 32 MB of L2 and evicts the W panels every CU is about to re-read (measured +12 % on the
                        // n = 2304 / 3072 shapes, nothing on n = 768)
                        if (DEFER_ST) {
                            // residual variants: a row's residual load issued behind earlier rows' stores waits in the in-order
                            // queue until those have drained (measured: 17 k cycles of epilogue against 6 k without the stores).
                            // The finished rows are parked in the accumulator registers their m-tile has just vacated and all
                            // 32 stores are issued after the last residual load.
                            const uint4 w4 = __builtin_bit_cast(uint4, hv);
                            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(stash[ts][p * 4 + 0]) : "v"(w4.x));
                            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(stash[ts][p * 4 + 1]) : "v"(w4.y));
                            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(stash[ts][p * 4 + 2]) : "v"(w4.z));
                            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(stash[ts][p * 4 + 3]) : "v"(w4.w));
                        } else if (xmode & 2) { /* ablation: no stores */
                        } else if (xmode & 512) {
                            // (s_nop 1 behind every asm store: a VMEM store of more than 64 bits must not be followed within two wait
                            // states by a write of its data registers, and hipcc's hazard recognizer cannot see a store inside inline asm -
                            // the deferred-store loop below refills the same four registers for the next store right away.  Round 4: with
                            // the per-store branches gone the big launches wrote corrupted rows until the nops went in)
                            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_el * 2u), "v"(hv), "s"(crow) : "memory");
                        } else *(half8*)(crow + lane_el) = hv;
                    }
                }
                if (HAS_RES && tm < 8 && tm + RD - 1 < 8) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        rq[(tm + RD - 1) % RD][p] = W4_RLOAD(Rt + (size_t)(((tm + RD - 1) * 16 + p * 4) * N) + lane_el);
                }
                if (tm < 8) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int rr = p * 4 + q16;
                        hvb[p] = *(const half8*)(ep + rr * 256 + ((c16 ^ rr) << 4));
                    }
                }
            }
            if (DEFER_ST) {
#pragma unroll
                for (int ts = 0; ts < 8; ++ts)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        uint4 w4;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(w4.x) : "a"(stash[ts][p * 4 + 0]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(w4.y) : "a"(stash[ts][p * 4 + 1]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(w4.z) : "a"(stash[ts][p * 4 + 2]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(w4.w) : "a"(stash[ts][p * 4 + 3]));
                        const half8 hv = __builtin_bit_cast(half8, w4);
                        half_t* const crow = Ct + (size_t)((ts * 16 + p * 4) * N);
                        if (xmode & 2) { /* ablation: no stores */
                        } else if (xmode & 512) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_el * 2u), "v"(hv), "s"(crow) : "memory");
                        else *(half8*)(crow + lane_el) = hv;
                    }
            }
        }
        if (LN == 2 && !(xmode & 16)) {
            if (W4_ABL_STATS) { keep1[0] = keep1[1] = 0.f; keep2[0] = keep2[1] = 128.f; }   // probe: mean 0 / variance 1 rows, the atomics stay
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // fixed point so that the six partial sums of a row (3 tile columns x 2 waves) add up deterministically
                unsigned long long* dst = (unsigned long long*)(ln_out + 2 * ((size_t)bm * BM + wm * 128)) + (unsigned)(2 * (4 * c16 + 64 * j + q16));
                atomicAdd(dst, (unsigned long long)__float2ll_rn(keep1[j] * 16777216.f));
                atomicAdd(dst + 1, (unsigned long long)__float2ll_rn(keep2[j] * 1048576.f));
            }
        }
        if (ti + 1 < my_tiles) it_step(cit);
        W4_PROBE_TILE_END()
    }
    W4_PROBE_FINISH()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail refills must land before the LDS is handed to another block
    if (W4_PROBE_VALU && pv0 + pv1 + pv2 + pv3 == 12345.678f) C[0] = (half_t)pv0;
#undef W4_PROBE
#undef W4_EVEN
#undef W4_ODD
#undef W4_SUB
#undef W4_SUB_RANGE
#undef W4_H_NONE
#undef W4_H_EVEN
#undef W4_H_ODD
#undef W4_DMA
#undef W4_RD
#undef W4_LGKM
}

#ifdef SCD_ABLATE   // A/B kernel (SCD_GEMM_MFMA=8): not in the default build
#include "ablate/gemm_w8_kernel.h"
#endif

// n-tiles per group: the W panels of a group (ng*256*K*2 bytes) should stay inside one XCD's 4 MB L2; every extra group
// re-reads the activations once.  Estimate the beyond-L2 traffic of each candidate and keep the cheapest.
static int choose_ng(int M, int K, int tiles_n, int total, int resident) {
    static const int ng_env = getenv("SCD_GEMM_NG") ? atoi(getenv("SCD_GEMM_NG")) : 0;
    if (ng_env > 0) return ng_env < tiles_n ? ng_env : tiles_n;
    int ng = tiles_n;
    const double a_bytes = 2.0 * M * (double)K, panel = 512.0 * K;
    double best = 1e300;
    for (int groups = 1; groups <= tiles_n; ++groups) {
        const int cand = (tiles_n + groups - 1) / groups;
        const double wg = cand * panel;
        const double rounds = (double)total / resident;
        // W panels of a group stay in an XCD's 4-MB L2 up to ~3.6 MB: round 4's sweep at the default launch size (786,432 rows,
        // profiles/r04_gemm_ng_sweep.txt) - QKV's nine tile columns (3.54 MB) as ONE group: -5.9 % (one pass over A instead of two),
        // -3.2 % at 393,216 rows, neutral at 131,072; fc1's twelve (4.7 MB) as one group: +3 %, two groups of six stay best
        const double w_traffic = wg <= 3.6e6 ? 8.0 * tiles_n * panel : 8.0 * rounds * wg;
        const double cost = groups * a_bytes + w_traffic;
        if (cost < best) { best = cost; ng = cand; }
    }
    return ng;
}

#ifdef SCD_ABLATE
#include "ablate/gemm_w8_launch.h"
#endif

template <int NT, int ACT, bool B, bool RR, int LN>
static int launch_w4(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                     const scd_gemm_ln* ln, hipStream_t st) {
    constexpr int LDS = 2 * 65536 + 16384;
    if (M % 256 || N % 256 || K % 64) return SCD_EINVAL;
    { const int rc_ = scd_set_max_lds((const void*)gemm_w4_kernel<NT, ACT, B, RR, LN, false>, LDS); if (rc_) return rc_; }
    { const int rc_ = scd_set_max_lds((const void*)gemm_w4_kernel<NT, ACT, B, RR, LN, true>, LDS); if (rc_) return rc_; }
    const int tiles_m = M / 256, tiles_n = N / 256, total = tiles_m * tiles_n;
    static const int xenv = SCD_ABLATE_ENV("SCD_GEMM_X", 0);
    static const int nt_env = getenv("SCD_GEMM_NT") ? atoi(getenv("SCD_GEMM_NT")) : -1;   // -1: by size
    const bool nt = nt_env >= 0 ? nt_env != 0 : 2.0 * M * (double)N > 64e6;   // C beyond what L2 (32 MB) could keep anyway
    const int xmode = xenv | (nt ? 512 : 0);
    const int ng = choose_ng(M, K, tiles_n, total, 256);
    const int grid = total < 256 ? (total >= 8 ? total / 8 * 8 : total) : 256;
    static const int stagger_env = SCD_ABLATE_ENV("SCD_GEMM_STAGGER", 0);   // ticks per phase, experiment
    const int stagger = stagger_env;
#define W4_GO(NTSV)                                                                                                            \
    gemm_w4_kernel<NT, ACT, B, RR, LN, NTSV><<<grid, 256, LDS, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng,         \
                                                                     LN == 1 ? (const float2*)ln->rs_in : nullptr, LN == 1 ? ln->colsum : nullptr, \
                                                                     LN == 2 ? ln->stats_out : nullptr, stagger)
    if (nt) W4_GO(true); else W4_GO(false);
#undef W4_GO
    W4_PROBE_REPORT()
    return SCD_OK;
}

// The patch-embedding GEMM fed from the image (gemm_w4_kernel<..., IMG>): C[b * np + p][n] = sum_k patch(b, p)[k] W[n][k], no bias, no
// activation.  pixels fp16 [batch, 3, image, image], patch 16, M = rows of C (a multiple of 256 >= batch * np; rows beyond are scratch).
int scd_gemm_launch_img(const half_t* pixels, const half_t* W, half_t* C, int64_t M, int N, int batch, int image, hipStream_t st) {
    const int gp = image / 16, K = 768;
    SCD_REQUIRE(pixels && W && C && batch > 0 && image % 16 == 0 && M % 256 == 0 && N % 256 == 0 && M >= (int64_t)batch * gp * gp && M < (1ll << 31),
                "gemm_img: bad shape (m=%lld n=%d batch=%d image=%d)", (long long)M, N, batch, image);
    SCD_REQUIRE((double)batch * 3.0 * image * image * 2.0 < 4294967296.0, "gemm_img: the image batch exceeds 32-bit byte offsets (batch %d)", batch);
    constexpr int LDS = 2 * 65536 + 16384;
    { const int rc_ = scd_set_max_lds((const void*)gemm_w4_kernel<8, SCD_ACT_NONE, false, false, 0, false, true>, LDS); if (rc_) return rc_; }
    { const int rc_ = scd_set_max_lds((const void*)gemm_w4_kernel<8, SCD_ACT_NONE, false, false, 0, true, true>, LDS); if (rc_) return rc_; }
    const int tiles_m = (int)(M / 256), tiles_n = N / 256, total = tiles_m * tiles_n;
    const bool nt = 2.0 * M * (double)N > 64e6;
    const int ng = choose_ng((int)M, K, tiles_n, total, 256);
    const int grid = total < 256 ? (total >= 8 ? total / 8 * 8 : total) : 256;
    if (nt) gemm_w4_kernel<8, SCD_ACT_NONE, false, false, 0, true, true><<<grid, 256, LDS, st>>>(pixels, W, nullptr, nullptr, C, (int)M, N, K, tiles_n, total, 512, ng, nullptr, nullptr, nullptr, 0, image, gp, batch * gp * gp);
    else gemm_w4_kernel<8, SCD_ACT_NONE, false, false, 0, false, true><<<grid, 256, LDS, st>>>(pixels, W, nullptr, nullptr, C, (int)M, N, K, tiles_n, total, 0, ng, nullptr, nullptr, nullptr, 0, image, gp, batch * gp * gp);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// the LayerNorm-folded variants exist for the shapes the encoders use: bias, no residual (LN = 1) and bias + residual (LN = 2)
static int launch_w4_ln(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K, int act,
                        const scd_gemm_ln* ln, hipStream_t st) {
#ifdef SCD_ABLATE
    static const int w8 = SCD_ABLATE_ENV("SCD_GEMM_MFMA", 4) == 8;
    if (w8) {
        if (ln->stats_in) {
            if (!bias || R || ln->stats_out) return SCD_EINVAL;
            if (act == SCD_ACT_NONE) return launch_w8<SCD_ACT_NONE, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
            if (act == SCD_ACT_QUICKGELU) return launch_w8<SCD_ACT_QUICKGELU, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
            return launch_w8<SCD_ACT_GELU, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
        }
        if (!bias || !R || act != SCD_ACT_NONE) return SCD_EINVAL;
        return launch_w8<SCD_ACT_NONE, true, true, 2>(A, W, bias, R, C, M, N, K, ln, st);
    }
#endif
    if (ln->stats_in) {
        if (!bias || R || ln->stats_out || !ln->rs_in) return SCD_EINVAL;
        if (act == SCD_ACT_NONE) return launch_w4<8, SCD_ACT_NONE, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
        if (act == SCD_ACT_QUICKGELU) return launch_w4<8, SCD_ACT_QUICKGELU, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
        return launch_w4<8, SCD_ACT_GELU, true, false, 1>(A, W, bias, R, C, M, N, K, ln, st);
    }
    if (!bias || !R || act != SCD_ACT_NONE) return SCD_EINVAL;
    return launch_w4<8, SCD_ACT_NONE, true, true, 2>(A, W, bias, R, C, M, N, K, ln, st);
}

template <int BM, int ACT, bool B, bool RR>
static int launch_dma(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                      hipStream_t st) {
    constexpr int LDS = BM == 256 ? 4 * (256 * 64 + 16384) + 8 * 4096 : 3 * (128 * 64 + 16384) + 4 * 2048;   // 160 KB / 80 KB
#ifdef SCD_ABLATE
    { const int rc_ = scd_set_max_lds((const void*)gemm_dma_kernel<BM, ACT, B, RR>, LDS); if (rc_) return rc_; }
#else
    if constexpr (BM != 256) { const int rc_ = scd_set_max_lds((const void*)gemm_dma_kernel<BM, ACT, B, RR>, LDS); if (rc_) return rc_; }
#endif
    const int tiles_n = N / 256, total = (M / BM) * tiles_n;
    const int resident = BM == 256 ? 256 : 512;
    const int grid = total < resident ? (total >= 8 ? total / 8 * 8 : total) : resident;
    static const int xmode = SCD_ABLATE_ENV("SCD_GEMM_X", 0);
    const int ng = choose_ng(M, K, tiles_n, total, resident);
#ifdef SCD_ABLATE
    static const int mfma_sel = SCD_ABLATE_ENV("SCD_GEMM_MFMA", 4);   // 4: four-wave kernel (default); 8 / 16 / 32: eight-wave kernels
    if (BM == 256 && mfma_sel == 4) return launch_w4<8, ACT, B, RR, 0>(A, W, bias, R, C, M, N, K, nullptr, st);
    if (BM == 256 && mfma_sel == 8) return launch_w8<ACT, B, RR, 0>(A, W, bias, R, C, M, N, K, nullptr, st);
    if (BM == 256 && mfma_sel == 16) {
        { const int rc_ = scd_set_max_lds((const void*)gemm_dma16_kernel<ACT, B, RR>, 163840); if (rc_) return rc_; }
        gemm_dma16_kernel<ACT, B, RR><<<grid, 512, 163840, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng);
        return SCD_OK;
    }
    gemm_dma_kernel<BM, ACT, B, RR><<<grid, BM * 2, LDS, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng);
    return SCD_OK;
#else
    // default build: 256-row tiles go to the four-wave kernel; the eight-wave DMA kernel serves M % 256 == 128 only
    if constexpr (BM == 256) {
        return launch_w4<8, ACT, B, RR, 0>(A, W, bias, R, C, M, N, K, nullptr, st);
    } else {
        gemm_dma_kernel<BM, ACT, B, RR><<<grid, BM * 2, LDS, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng);
        return SCD_OK;
    }
#endif
}
template <int BM, int ACT>
static int launch_dma_act(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                          hipStream_t st) {
    if (bias && R) return launch_dma<BM, ACT, true, true>(A, W, bias, R, C, M, N, K, st);
    if (bias) return launch_dma<BM, ACT, true, false>(A, W, bias, R, C, M, N, K, st);
    if (R) return launch_dma<BM, ACT, false, true>(A, W, bias, R, C, M, N, K, st);
    return launch_dma<BM, ACT, false, false>(A, W, bias, R, C, M, N, K, st);
}
template <int BM>
static int launch_dma_bm(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                         int act, hipStream_t st) {
    if (act == SCD_ACT_NONE) return launch_dma_act<BM, SCD_ACT_NONE>(A, W, bias, R, C, M, N, K, st);
    if (act == SCD_ACT_QUICKGELU) return launch_dma_act<BM, SCD_ACT_QUICKGELU>(A, W, bias, R, C, M, N, K, st);
    return launch_dma_act<BM, SCD_ACT_GELU>(A, W, bias, R, C, M, N, K, st);
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

// {rstd, -mean * rstd} per row from the fixed-point row sums, exactly the arithmetic the GEMM epilogue used to repeat per lane and tile
// (float(sum) * 2^-24 / k, max(E[x^2] - mean^2, 0), v_rsq_f32): the features keep their bits.  Also clears the statistics buffer the
// NEXT residual GEMM accumulates into (was a loop in the GEMM's prologue).
__global__ void __launch_bounds__(256) ln_finish_kernel(const long long* __restrict__ stats, long long m, float invk, float eps,
                                                        float2* __restrict__ rs, float4* __restrict__ zero) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const longlong2 q = *(const longlong2*)(stats + 2 * i);
    const float mu = __ll2float_rn(q.x) * (5.9604644775390625e-8f * invk);
    const float var = fmaxf(fmaf(-mu, mu, __ll2float_rn(q.y) * (9.5367431640625e-7f * invk)), 0.f);
    const float rstd = __builtin_amdgcn_rsqf(var + eps);
    rs[i] = make_float2(rstd, -mu * rstd);
    if (zero) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
int scd_gemm_ln_finish(const long long* stats, int64_t M, float inv_k, float eps, float* rs_out, long long* zero_out, hipStream_t st) {
    SCD_REQUIRE(stats && rs_out && M > 0, "scd_gemm_ln_finish: bad arguments");
    ln_finish_kernel<<<(unsigned)scd_cdiv(M, 256), 256, 0, st>>>(stats, M, inv_k, eps, (float2*)rs_out, (float4*)zero_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

int scd_gemm_launch_ln(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                       int act, const scd_gemm_ln* ln, hipStream_t st) {
    SCD_REQUIRE(ln && (ln->stats_in || ln->stats_out), "gemm_ln: no LayerNorm term");
    SCD_REQUIRE(A && W && C && M > 0 && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && M < (1ll << 31),
                "gemm_ln: shape m=%lld n=%d k=%d must be multiples of 256/256/64", (long long)M, N, K);
    SCD_REQUIRE(C != (half_t*)A, "gemm_ln: C must not alias A");
    const int rc = launch_w4_ln(A, W, bias, R, C, (int)M, N, K, act, ln, st);
    SCD_REQUIRE(rc == SCD_OK, "gemm_ln: unsupported combination (bias %d residual %d act %d)", bias != nullptr, R != nullptr, act);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

int scd_gemm_launch(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                    int act, hipStream_t st) {
    SCD_REQUIRE(A && W && C, "gemm: null operand");
    SCD_REQUIRE(M > 0 && M % 128 == 0 && N > 0 && N % 128 == 0 && K > 0 && K % 64 == 0 && M < (1ll << 31),
                "gemm: shape m=%lld n=%d k=%d must be multiples of 128/128/64", (long long)M, N, K);
    SCD_REQUIRE(C != (half_t*)A, "gemm: C must not alias A");
    SCD_REQUIRE(act == SCD_ACT_NONE || act == SCD_ACT_QUICKGELU || act == SCD_ACT_GELU, "gemm: bad activation %d", act);
    static const int force = SCD_ABLATE_ENV("SCD_GEMM_TILE", 0);    // 64 -> legacy 128x128 kernel, 128 -> 128-row DMA kernel
    if (N % 256 == 0 && force != 64) {
        int rc = -1;
        // BM=256 (one 8-wave block per CU) measures a few % ahead of BM=128 (two 4-wave blocks per CU) on the ViT shapes
        if (M % 256 == 0 && force != 128) rc = launch_dma_bm<256>(A, W, bias, R, C, (int)M, N, K, act, st);
        else rc = launch_dma_bm<128>(A, W, bias, R, C, (int)M, N, K, act, st);
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
    SCD_DEVICE_ENTRY(h, "scd_gemm_f16");
    SCD_REQUIRE(h, "scd_gemm_f16: null handle");
    return scd_gemm_launch((const half_t*)A, (const half_t*)W, bias, (const half_t*)residual, (half_t*)C, m, n, k, act,
                           (hipStream_t)stream);
}
