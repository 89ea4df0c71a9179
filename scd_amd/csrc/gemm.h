// Internal: fp16 GEMM launcher shared by gemm.hip and encoder.hip.
#pragma once
#include "common.h"
enum { SCD_ACT_NONE = 0, SCD_ACT_QUICKGELU = 1, SCD_ACT_GELU = 2 };
int scd_gemm_launch(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int64_t M, int N, int K,
                    int act, hipStream_t st);
