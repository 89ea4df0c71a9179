// Launcher of the ablation-only eight-wave kernel: textually included by gemm.hip under -DSCD_ABLATE.
template <int ACT, bool B, bool RR, int LN>
static int launch_w8(const half_t* A, const half_t* W, const float* bias, const half_t* R, half_t* C, int M, int N, int K,
                     const scd_gemm_ln* ln, hipStream_t st) {
    constexpr int LDS = 2 * 65536 + 16384;
    if (M % 256 || N % 256 || K % 64) return SCD_EINVAL;
    { const int rc_ = scd_set_max_lds((const void*)gemm_w8_kernel<ACT, B, RR, LN>, LDS); if (rc_) return rc_; }
    const int tiles_m = M / 256, tiles_n = N / 256, total = tiles_m * tiles_n;
    static const int xenv = SCD_ABLATE_ENV("SCD_GEMM_X", 0);
    static const int nt_env = getenv("SCD_GEMM_NT") ? atoi(getenv("SCD_GEMM_NT")) : -1;
    const bool nt = nt_env >= 0 ? nt_env != 0 : 2.0 * M * (double)N > 64e6;
    const int xmode = xenv | (nt ? 512 : 0);
    const int ng = choose_ng(M, K, tiles_n, total, 256);
    const int grid = total < 256 ? (total >= 8 ? total / 8 * 8 : total) : 256;
    gemm_w8_kernel<ACT, B, RR, LN><<<grid, 512, LDS, st>>>(A, W, bias, R, C, M, N, K, tiles_n, total, xmode, ng,
                                                           LN == 1 ? ln->stats_in : nullptr, LN == 1 ? ln->colsum : nullptr,
                                                           LN == 1 ? ln->inv_k : 0.f, LN == 1 ? ln->eps : 0.f,
                                                           LN == 2 ? ln->stats_out : nullptr, LN == 1 ? ln->zero_out : nullptr);
    return SCD_OK;
}

