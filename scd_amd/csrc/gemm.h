// Internal: fp16 GEMM launcher shared by gemm.hip and encoder.hip.
#pragma once
#include "common.h"
enum { SCD_ACT_NONE = 0, SCD_ACT_QUICKGELU = 1, SCD_ACT_GELU = 2 };
int scd_gemm_launch(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                    int act, hipStream_t st);
// The patch-embedding GEMM of a ViT fed from the fp16 image batch itself (patch 16; gemm_w4_kernel IMG): no im2col matrix.  M = rows
// of C, a multiple of 256 >= batch * (image / 16)^2; N % 256 == 0.
int scd_gemm_launch_img(const half_t* pixels, const half_t* W, half_t* C, int64_t M, int N, int batch, int image, hipStream_t st);

// LayerNorm folded into the four-wave GEMM (gemm.hip, gemm_w4_kernel LN = 1 / 2).  Requires M % 256 = N % 256 = K % 64 = 0.
struct scd_gemm_ln {
    const long long* stats_in;   // [M][2] {sum * 2^24, sum of squares * 2^20} of A's rows (fixed point): normalise A on the fly - or null
    const float* colsum;     // [N] sum_k W'[n][k]                                       (with stats_in)
    float inv_k, eps;        // 1 / (row length), LayerNorm epsilon                       (with stats_in)
    long long* stats_out;    // [M][2] += the same fixed-point sums of the rows of C (integer atomics) - or null; needs bias + residual
    long long* zero_out;     // with stats_in: a second [M][2] buffer cleared for the next residual GEMM's stats_out - or null (done by scd_gemm_ln_finish)
    const float* rs_in;      // [M][2] {rstd, -mean * rstd} of A's rows: what scd_gemm_ln_finish made of stats_in (the four-wave kernel reads these)
};
// Once per folded LayerNorm, in front of the GEMM that applies it: rs_out[m] = {rstd, -mean * rstd} from the fixed-point row sums
// stats[m], and zero_out (the other statistics buffer, or null) cleared.  One thread per row.
int scd_gemm_ln_finish(const long long* stats, int64_t M, float inv_k, float eps, float* rs_out, long long* zero_out, hipStream_t st);
int scd_gemm_launch_ln(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                       int act, const scd_gemm_ln* ln, hipStream_t st);
