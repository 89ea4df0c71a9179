// Cycle-counter probes of gemm_w4_kernel (SCD_GEMM_X bits 64 / 128 / 2048 of the -DSCD_ABLATE build; textually included by gemm.hip).
// With SCD_ABLATE they are the code that sat in the kernel through round 5; without it every macro is empty, so the shipped kernel
// carries neither the counters, nor the g_w4_dbg device symbol, nor the host read-back.
#pragma once
#ifdef SCD_ABLATE
__device__ unsigned long long g_w4_dbg[256 * 4];   // SCD_GEMM_X & 64: per block {main-loop cycles, epilogue cycles, tiles, total}
#define W4_PROBE_DECL()                                                                       \
    unsigned long long t_main = 0, t_epi = 0, t_begin = __builtin_readcyclecounter();        \
    unsigned long long t_even = 0, t_odd = 0, t_bar = 0, t_sub = t_begin;                     \
    unsigned long long t0 = 0, t1 = 0;
#define W4_PROBE_SUB(ACC) if (xmode & 128) { const unsigned long long t = __builtin_readcyclecounter(); ACC += t - t_sub; t_sub = t; }
#define W4_PROBE_BAR() if (xmode & 128) { const unsigned long long t = __builtin_readcyclecounter(); t_bar += ((xmode & 2048) && !tile_first) ? 0ull : t - t_sub; t_sub = t; }
#define W4_PROBE_MARK(T) T = (xmode & 64) ? __builtin_readcyclecounter() : 0;
#define W4_PROBE_TILE_END()                                          \
    if (xmode & 64) {                                                \
        const unsigned long long t2 = __builtin_readcyclecounter();  \
        t_main += t1 - t0;                                           \
        t_epi += t2 - t1;                                            \
    }
#define W4_PROBE_FINISH()                                                                                                              \
    if ((xmode & 64) && tid == 0) {                                                                                                    \
        g_w4_dbg[blockIdx.x * 4 + 0] = t_main;                                                                                         \
        g_w4_dbg[blockIdx.x * 4 + 1] = t_epi;                                                                                          \
        g_w4_dbg[blockIdx.x * 4 + 2] = my_tiles;                                                                                       \
        g_w4_dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter() - t_begin;                                                         \
        if (xmode & 128) { /* per sub-step: even, wait+barrier, odd (the odd figure of a tile's last chunk includes the epilogue) */  \
            g_w4_dbg[blockIdx.x * 4 + 0] = t_even;                                                                                     \
            g_w4_dbg[blockIdx.x * 4 + 1] = t_bar;                                                                                      \
            g_w4_dbg[blockIdx.x * 4 + 2] = t_odd;                                                                                      \
            g_w4_dbg[blockIdx.x * 4 + 3] = chunks;                                                                                     \
        }                                                                                                                              \
    }
// host side, behind the launch (launch_w4): read the counters back and print the averages
#define W4_PROBE_REPORT()                                                                                                              \
    if (xmode & 64) {                                                                                                                  \
        static unsigned long long h[256 * 4];                                                                                          \
        SCD_HIP(hipDeviceSynchronize());                                                                                               \
        SCD_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w4_dbg), sizeof(h)));                                                              \
        double tm = 0, te = 0, tt = 0, nt2 = 0;                                                                                        \
        for (int b = 0; b < grid; ++b) { tm += h[b * 4]; te += h[b * 4 + 1]; nt2 += h[b * 4 + 2]; tt += h[b * 4 + 3]; }                \
        if (xmode & 128)                                                                                                               \
            fprintf(stderr, "[w4 m=%d n=%d k=%d] per chunk: even %.0f, wait+barrier %.0f, odd(+epilogue share) %.0f cyc\n", M, N, K, tm / tt, te / tt, nt2 / tt); \
        else                                                                                                                           \
            fprintf(stderr, "[w4 m=%d n=%d k=%d] per tile: main %.0f cyc, epilogue %.0f cyc; per block total %.0f cyc, tiles %.1f\n", M, N, K, \
                    tm / nt2, te / nt2, tt / grid, nt2 / grid);                                                                        \
    }
#else
#define W4_PROBE_DECL()
#define W4_PROBE_SUB(ACC)
#define W4_PROBE_BAR()
#define W4_PROBE_MARK(T)
#define W4_PROBE_TILE_END()
#define W4_PROBE_FINISH()
#define W4_PROBE_REPORT()
#endif
