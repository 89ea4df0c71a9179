// Ablation-only kernel (builds with -DSCD_ABLATE, selected by SCD_GEMM_MFMA=16): textually included by gemm.hip, not a standalone header.
// ------------------------------------------------------------------------------------------------
// Same block tile / ring / schedule as gemm_dma_kernel<256>, but on v_mfma_f32_16x16x32_f16 (the shape on which gfx950
// sustains the higher clock under load): a 32-deep sub-step is ONE k-step of 8(m) x 4(n) 16x16 tiles = 32 MFMAs per wave.
// Fragment (A or B operand): lane l reads row (l&15), 16-B chunk (l>>4) of a 64-B LDS row.  Chunk swizzle
// pc = chunk ^ ((-(row>>2)) & 3): every ds_read_b128 lane group then touches 16 distinct 16-B slots.
// MFMA group 0 = m-tiles 0-3, group 1 = m-tiles 4-7; the A fragments of group 1 are read while group 0 computes, and
// the W + A(0-3) fragments of the next sub-step while group 1 computes.

template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ void __launch_bounds__(512, 2) gemm_dma16_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                            const float* __restrict__ bias, const half_t* __restrict__ R,
                                                            half_t* __restrict__ C, int M, int N, int K, int tiles_n, int total_tiles,
                                                            int xmode, int ng) {
    constexpr int BM = 256, NSLOT = 4, SLOT = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, q16 = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K >> 5;
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
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int xcd = blockIdx.x % nxcd, slot_id = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;
    const int c0 = (int)((long long)xcd * total_tiles / nxcd), c1 = (int)((long long)(xcd + 1) * total_tiles / nxcd);
    const int tb = c0 + slot_id;
    const int my_tiles = tb < c1 ? (c1 - tb + per_xcd - 1) / per_xcd : 0;
    const int steps = my_tiles * nk;
    if (steps <= 0) return;
    const int tstride = per_xcd;

    auto swz = [](int row) { return (0 - (row >> 2)) & 3; };
    const int lrow = lane >> 2, pc = lane & 3;
    int a_off[2], w_off[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rowl = wave * 32 + p * 16 + lrow;
        a_off[p] = rowl * K + ((pc ^ swz(rowl)) << 3);
        w_off[p] = a_off[p];
    }
    auto issue = [&](int tile, int kt, int slot) {
        int bm, bn;
        tile_mn(tile, bm, bn);
        if (xmode & 4) { bm = 0; bn = 0; }
        const half_t* ga = A + (size_t)bm * BM * K + kt * 32;
        const half_t* gw = W + (size_t)bn * 256 * K + kt * 32;
        char* sa = smem + slot * SLOT + wave * 2048;
        char* sw = sa + 16384;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            __builtin_amdgcn_global_load_lds((const void*)(ga + a_off[p]), (lds_ptr_t)(sa + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(gw + w_off[p]), (lds_ptr_t)(sw + p * 1024), 16, 0, 0);
        }
    };
    // fragment byte offsets inside a sub-tile: rows (base + 16*t + c16), chunk q16
    int offw[4], offa[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = wn * 64 + t * 16 + c16;
        offw[t] = 16384 + row * 64 + ((q16 ^ swz(row)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int row = wm * 128 + t * 16 + c16;
        offa[t] = row * 64 + ((q16 ^ swz(row)) << 4);
    }

    f32x4v acc[4][8];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 8; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[tn][tm][q] = 0.f;

    int tile = tb, kt = 0, ntile = tb, nkt = 0;
#pragma unroll
    for (int pre = 0; pre < NSLOT - 1; ++pre) {
        if (pre < steps) issue(ntile, nkt, pre);
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
    }
    int store_age = 8;
    half8 rpre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) rpre[p][q] = (half_t)0.f;

    half8 fw[4], fa_lo[4], fa_hi[4], fwn[4];
    if (steps >= NSLOT - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fw[t] = *(const half8*)(smem + offw[t]);
        fa_lo[t] = *(const half8*)(smem + offa[t]);
    }
    int cslot = 0;
    for (int s = 0; s < steps; ++s) {
        const char* cur = smem + cslot * SLOT;
        const int nslot = cslot + 1 == NSLOT ? 0 : cslot + 1;
        if (HAS_RES && kt == nk - 2) {
            int bm, bn;
            tile_mn(tile, bm, bn);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                rpre[p] = *(const half8*)(R + ((size_t)bm * BM + wm * 128 + p * 8 + (lane >> 3)) * N + bn * 256 + wn * 64 + (lane & 7) * 8);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) fa_hi[t] = *(const half8*)(cur + offa[4 + t]);          // m-tiles 4-7 of this sub-step
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[tn], fa_lo[tm], acc[tn][tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (s + 1 < steps) {
            if (s + NSLOT - 2 >= steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (store_age < 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        ++store_age;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NSLOT - 1 < steps && !((xmode & 1) && s >= 2)) {
            int ls = cslot + NSLOT - 1;
            if (ls >= NSLOT) ls -= NSLOT;
            issue(ntile, nkt, ls);
        }
        if (++nkt == nk) { nkt = 0; ntile += tstride; }
        if (s + 1 < steps) {
            const char* nx = smem + nslot * SLOT;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fwn[t] = *(const half8*)(nx + offw[t]);
                fa_lo[t] = *(const half8*)(nx + offa[t]);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tn][4 + tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[tn], fa_hi[tm], acc[tn][4 + tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) fw[t] = fwn[t];
        cslot = nslot;
        if (++kt == nk) {
            // epilogue: D tile (tn, tm): lane (c16 = m column, q16) holds n = tn*16 + q16*4 + 0..3.  Per 32-row m block
            // (two m-tiles) the values go through a per-wave LDS patch [32 m][64 n] fp16 (128-B rows, chunk XOR row&7)
            // and leave as whole 128-byte row segments.
            int bm, bn;
            tile_mn(tile, bm, bn);
            char* ep = smem + NSLOT * SLOT + wave * 4096;
            const int nb0 = bn * 256 + wn * 64;
            f32x4v bq[4];
            if (HAS_BIAS) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float4 b4 = *(const float4*)(bias + nb0 + tn * 16 + q16 * 4);
                    bq[tn][0] = b4.x; bq[tn][1] = b4.y; bq[tn][2] = b4.z; bq[tn][3] = b4.w;
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
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = h * 16 + c16;
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        half4 o;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            float v = acc[tn][2 * i + h][q4];
                            if (HAS_BIAS) v += bq[tn][q4];
                            o[q4] = (half_t)act_apply(v, ACT);
                            acc[tn][2 * i + h][q4] = 0.f;
                        }
                        *(half4*)(ep + row * 128 + (((tn * 2 + (q16 >> 1)) ^ (row & 7)) << 4) + (q16 & 1) * 8) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int rr = p * 8 + (lane >> 3), cc = lane & 7;
                    half8 hv = *(const half8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
                    const size_t off = ((size_t)bm * BM + wm * 128 + i * 32 + rr) * N + nb0 + cc * 8;
                    if (HAS_RES) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) hv[q] = (half_t)((float)hv[q] + (float)rcur[p][q]);
                    }
                    if (!(xmode & 2)) *(half8*)(C + off) = hv;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            kt = 0;
            tile += tstride;
            store_age = 0;
        }
    }
}

