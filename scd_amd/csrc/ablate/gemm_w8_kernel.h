// Ablation-only kernel (builds with -DSCD_ABLATE, selected by SCD_GEMM_MFMA=8): textually included by gemm.hip, not a standalone header.
// ------------------------------------------------------------------------------------------------
// Eight-wave sibling of gemm_w4_kernel (SCD_GEMM_MFMA=8, A/B candidate): the same 256x256 block tile, 64-deep chunks, LDS
// image, tile order, non-temporal stores and LayerNorm folding, but two waves per SIMD with 128(m) x 64(n) wave tiles
// (128 accumulator VGPRs), MFMAs left to the compiler's scheduler.  The idea: the ~25 % of a chunk that the four-wave kernel
// loses to instruction-issue stalls (ring fills, ds_read issue, the barrier) is covered by the other wave of the SIMD.
template <int ACT, bool HAS_BIAS, bool HAS_RES, int LN>
__global__ void __launch_bounds__(512, 2)
gemm_w8_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W, const float* __restrict__ bias,
               const half_t* __restrict__ R, half_t* __restrict__ C, int M, int N, int K, int tiles_n, int total_tiles,
               int xmode, int ng, const long long* __restrict__ ln_stats, const float* __restrict__ ln_colsum, float ln_invk,
               float ln_eps, long long* __restrict__ ln_out, long long* __restrict__ ln_zero) {
    constexpr int BM = 256, BN = 256, SLOT = 65536, WPART = 32768, EPI = 2 * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, q16 = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int nkc = K >> 6;
    const int tiles_m = total_tiles / tiles_n;
    const int per_group = tiles_m * ng;
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
    if (LN == 1 && ln_zero) {
        float4* z = (float4*)ln_zero;
        for (int i = blockIdx.x * 512 + tid; i < M; i += gridDim.x * 512) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int chunks = my_tiles * nkc;
    if (chunks <= 0) return;
    const int tstride = per_xcd;
    auto it_step = [&](TileIt& it) {
        it.t += tstride;
        it.bnl += it.r;
        it.bm += it.q;
        if (it.bnl >= it.w) { it.bnl -= it.w; ++it.bm; }
        if (it.bm >= tiles_m) {
            it_init(it, it.t);
            it.q = tstride / it.w;
            it.r = tstride - it.q * it.w;
        }
    };
    // DMA: instruction p (0..3) of a wave covers rows wave*32 + p*8 + (lane>>3); lane&7 = physical chunk
    const int drow = lane >> 3, dpc = lane & 7;
    unsigned voff[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int rowl = wave * 32 + par * 8 + drow;
        voff[par] = (unsigned)(rowl * K + ((dpc ^ ((rowl >> 1) & 7)) << 3)) * 2;
    }
    const int k16 = 16 * K;
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_lds = sbase + wave * 4096;
    auto issue = [&](const TileIt& it, int kc, int slot) {
        int bm = it.bm, bn = it.n0 + it.bnl;
        if (xmode & 4) { bm = 0; bn = 0; }
        const half_t* ga = A + (size_t)bm * BM * K + kc * 64;
        const half_t* gw = W + (size_t)bn * BN * K + kc * 64;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         ::"s"(dma_lds + slot * SLOT + p * 1024), "v"(voff[p & 1]), "s"(ga + (p >> 1) * k16) : "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         ::"s"(dma_lds + slot * SLOT + WPART + p * 1024), "v"(voff[p & 1]), "s"(gw + (p >> 1) * k16) : "memory");
    };
    // fragment byte offsets inside a slot (tile t adds t * 2048): k-half j uses chunk (q16 + 4j) ^ sw
    const int fsw = (c16 >> 1) & 7;
    const int fa_off[2] = {(wm * 128 + c16) * 128 + ((q16 ^ fsw) << 4), (wm * 128 + c16) * 128 + (((q16 + 4) ^ fsw) << 4)};
    const int fw_off[2] = {WPART + (wn * 64 + c16) * 128 + ((q16 ^ fsw) << 4), WPART + (wn * 64 + c16) * 128 + (((q16 + 4) ^ fsw) << 4)};

    TileIt cit, nit;
    it_init(cit, tb);
    cit.q = tstride / cit.w;
    cit.r = tstride - cit.q * cit.w;
    nit = cit;
    int nkt = 0, ntiles = 0;
    auto issue_advance = [&]() {
        if (++nkt == nkc) {
            nkt = 0;
            if (++ntiles < my_tiles) it_step(nit);
        }
    };
    issue(nit, nkt, 0);
    issue_advance();
    if (chunks > 1) issue(nit, nkt, 1); else issue(cit, 0, 1);
    issue_advance();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    constexpr int RD = 2;
    int g = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int bm = cit.bm, bn = cit.n0 + cit.bnl;
        const int nb0 = bn * BN + wn * 64;
        f32x4v acc[4][8];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 8; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[tn][tm][q] = 0.f;
        for (int kc = 0; kc < nkc; ++kc) {
            const char* sl = smem + (g & 1) * SLOT;
            {   // k-half 0
                half8 fw[4], fa[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) fw[t] = *(const half8*)(sl + fw_off[0] + t * 2048);
#pragma unroll
                for (int t = 0; t < 8; ++t) fa[t] = *(const half8*)(sl + fa_off[0] + t * 2048);
#pragma unroll
                for (int tm = 0; tm < 8; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
            }
            {   // k-half 1: once its fragments are in registers this wave is done with the slot; the chunk's barrier (chunk g+1
                // has landed, everybody is done with slot g&1 -> refill it with chunk g+2) sits behind the first 8 MFMAs
                half8 fw[4], fa[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) fw[t] = *(const half8*)(sl + fw_off[1] + t * 2048);
#pragma unroll
                for (int t = 0; t < 8; ++t) fa[t] = *(const half8*)(sl + fa_off[1] + t * 2048);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (g + 2 < chunks) issue(nit, nkt, g & 1); else issue(cit, 0, g & 1);
                issue_advance();
#pragma unroll
                for (int tm = 2; tm < 8; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
            }
            ++g;
        }
        // epilogue: per 16-row m-tile through a per-wave LDS patch [16][64] fp16 (128-B rows, chunk XOR (row & 7))
        {
            char* ep = smem + EPI + wave * 2048;
            f32x4v bq[4], sq[4];
            if (HAS_BIAS) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float4 b4 = *(const float4*)(bias + nb0 + tn * 16 + q16 * 4);
                    bq[tn][0] = b4.x; bq[tn][1] = b4.y; bq[tn][2] = b4.z; bq[tn][3] = b4.w;
                }
            }
            if (LN == 1) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float4 c4 = *(const float4*)(ln_colsum + nb0 + tn * 16 + q16 * 4);
                    sq[tn][0] = c4.x; sq[tn][1] = c4.y; sq[tn][2] = c4.z; sq[tn][3] = c4.w;
                }
            }
            float keep1[2] = {0.f, 0.f}, keep2[2] = {0.f, 0.f};
            half8 rq[RD][2];
            const int rrow = lane >> 3, rch = lane & 7;
            if (HAS_RES) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    rq[0][p] = *(const half8*)(R + ((size_t)bm * BM + wm * 128 + p * 8 + rrow) * N + nb0 + rch * 8);
            }
#pragma unroll
            for (int tm = 0; tm < 8; ++tm) {
                if (HAS_RES && tm + 1 < 8) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        rq[(tm + 1) % RD][p] = *(const half8*)(R + ((size_t)bm * BM + wm * 128 + (tm + 1) * 16 + p * 8 + rrow) * N + nb0 + rch * 8);
                }
                float rstd = 1.f, nmr = 0.f;
                if (LN == 1) {
                    const longlong2 qs = *(const longlong2*)(ln_stats + 2 * ((size_t)bm * BM + wm * 128 + tm * 16 + c16));
                    const float mu = __ll2float_rn(qs.x) * (5.9604644775390625e-8f * ln_invk);
                    const float var = fmaxf(fmaf(-mu, mu, __ll2float_rn(qs.y) * (9.5367431640625e-7f * ln_invk)), 0.f);
                    rstd = __builtin_amdgcn_rsqf(var + ln_eps);
                    nmr = -mu * rstd;
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    float2v v01 = {acc[tn][tm][0], acc[tn][tm][1]}, v23 = {acc[tn][tm][2], acc[tn][tm][3]};
                    if (LN == 1) {
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
                        const float2v c2 = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f};
                        const float2v one2 = {1.f, 1.f};
                        float2v e01 = v01 * c2, e23 = v23 * c2;
                        e01.x = __builtin_amdgcn_exp2f(e01.x); e01.y = __builtin_amdgcn_exp2f(e01.y);
                        e23.x = __builtin_amdgcn_exp2f(e23.x); e23.y = __builtin_amdgcn_exp2f(e23.y);
                        e01 += one2; e23 += one2;
                        e01.x = __builtin_amdgcn_rcpf(e01.x); e01.y = __builtin_amdgcn_rcpf(e01.y);
                        e23.x = __builtin_amdgcn_rcpf(e23.x); e23.y = __builtin_amdgcn_rcpf(e23.y);
                        v01 *= e01; v23 *= e23;
                    } else if (ACT != SCD_ACT_NONE) {
                        v01.x = act_apply(v01.x, ACT); v01.y = act_apply(v01.y, ACT);
                        v23.x = act_apply(v23.x, ACT); v23.y = act_apply(v23.y, ACT);
                    }
                    const half2v h01 = __builtin_convertvector(v01, half2v), h23 = __builtin_convertvector(v23, half2v);
                    const half4 o = {h01.x, h01.y, h23.x, h23.y};
                    *(half4*)(ep + c16 * 128 + (((tn * 2 + (q16 >> 1)) ^ (c16 & 7)) << 4) + (q16 & 1) * 8) = o;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int rr = p * 8 + rrow;
                    half8 hv = *(const half8*)(ep + rr * 128 + ((rch ^ (rr & 7)) << 4));
                    const size_t off = ((size_t)bm * BM + wm * 128 + tm * 16 + rr) * N + nb0 + rch * 8;
                    if (HAS_RES) hv = hv + rq[tm % RD][p];
                    if (LN == 2) {
                        float s1 = 0.f, s2 = 0.f;
                        const half2v ones = {(half_t)1.f, (half_t)1.f};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const half2v pr = {hv[2 * q], hv[2 * q + 1]};
                            s1 = __builtin_amdgcn_fdot2(pr, ones, s1, false);
                            s2 = __builtin_amdgcn_fdot2(pr, pr, s2, false);
                        }
#define W8_DPP_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xF, 0xF, true))
                        W8_DPP_ADD(s1, 0x141); W8_DPP_ADD(s2, 0x141);   // row_half_mirror: lane i <-> 7-i of its group of 8
                        W8_DPP_ADD(s1, 0x4E); W8_DPP_ADD(s2, 0x4E);     // quad_perm [2,3,0,1]
                        W8_DPP_ADD(s1, 0xB1); W8_DPP_ADD(s2, 0xB1);     // quad_perm [1,0,3,2]
#undef W8_DPP_ADD
                        const bool mine = ((tm * 2 + p) & 7) == rch;
                        keep1[tm >> 2] = mine ? s1 : keep1[tm >> 2];
                        keep2[tm >> 2] = mine ? s2 : keep2[tm >> 2];
                    }
                    if (xmode & 2) {
                    } else if (xmode & 512) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(C + off), "v"(hv) : "memory");
                    else *(half8*)(C + off) = hv;
                }
            }
            if (LN == 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned long long* dst = (unsigned long long*)(ln_out + 2 * ((size_t)bm * BM + wm * 128 + 8 * (rch + 8 * j) + rrow));
                    atomicAdd(dst, (unsigned long long)__float2ll_rn(keep1[j] * 16777216.f));
                    atomicAdd(dst + 1, (unsigned long long)__float2ll_rn(keep2[j] * 1048576.f));
                }
            }
        }
        if (ti + 1 < my_tiles) it_step(cit);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

