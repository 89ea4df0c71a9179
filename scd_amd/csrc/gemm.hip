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

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == SCD_ACT_QUICKGELU) return x / (1.0f + __expf(-1.702f * x));
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
