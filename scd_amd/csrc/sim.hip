// Image x text similarity with fused top-k for gfx950.  The N x V logit matrix is never written.
//
// Replaces (paths under /root/reference):
//   100*(F @ W) -> softmax -> topk x2      main_unsup.py:504-531   (unsup, softmax values)
//   100*(F @ W) -> topk x2                 main_ptsup.py:526-545   (ptsup, raw values)
//   argmax(100*F_u @ W_sel)                main_unsup.py:601-614, main_ptsup.py:668-676
//
// Decision semantics (shared with oracle/naming_oracle.py): order = (float64 dot product desc, index asc).
// Pass 1 (MFMA, fp32 accumulate) keeps 2 x 8 approximate candidates per image in registers; pass 2
// recomputes the candidates' dot products in float64, sorts them, and certifies that no non-candidate
// could reach rank k given the accumulation error bound; uncertified rows take an exact full-row pass.
#include "common.h"

#define TOPM 8

__device__ __forceinline__ void topm_insert(float (&lv)[TOPM], int (&li)[TOPM], float v, int idx) {
#pragma unroll
    for (int j = TOPM - 1; j >= 1; --j) {
        const bool up = v > lv[j - 1];
        const bool here = v > lv[j];
        lv[j] = up ? lv[j - 1] : (here ? v : lv[j]);
        li[j] = up ? li[j - 1] : (here ? idx : li[j]);
    }
    if (v > lv[0]) {
        lv[0] = v;
        li[0] = idx;
    }
}

// bf[dc*4 + k16] for a runtime d-chunk dc (0..7) and a compile-time k16: a switch over constants keeps the fragment
// array in registers (a runtime-indexed array would be demoted to scratch)
__device__ __forceinline__ half8 bf_sel(const half8 (&bf)[32], int dc, int k16) {
    switch (dc) {
        case 0: return bf[0 + k16];
        case 1: return bf[4 + k16];
        case 2: return bf[8 + k16];
        case 3: return bf[12 + k16];
        case 4: return bf[16 + k16];
        case 5: return bf[20 + k16];
        case 6: return bf[24 + k16];
        default: return bf[28 + k16];
    }
}

// Block = 8 waves = 256 images; wave w keeps the B fragments (F rows, K <= 512) of its 32 images in registers for the
// whole kernel and sweeps the vocabulary in tiles of 128 names.  W^T tiles are streamed by LDS-DMA (global_load_lds,
// 16 B/lane) in sub-tiles of [128 names][64 d] (128-B rows, 16-B chunk XOR (row>>1)&7 applied to the DMA source address
// and to the ds_read side) through a 4-slot ring; one flattened software pipeline runs over (tile, d-chunk): the DMAs of
// sub-step s+3 are issued after the mid-sub-step barrier that publishes slot s+1, waits are counted, fragment reads run
// one MFMA group ahead (same schedule as gemm_dma_kernel).
// v_mfma_f32_32x32x16_f16: A[row = name][k] from LDS, B[k][col = image] from registers,
// D[row = (reg&3)+8(reg>>2)+4h][col = image]: every lane sees 16 logits of ONE image per 32-name block, so the
// running top-8 list, its threshold and the online-softmax statistics are lane-private registers.
typedef __attribute__((address_space(3))) void* sim_lds_ptr_t;

template <bool SOFTMAX>
__global__ void __launch_bounds__(512, 2) sim_topk_kernel(const half_t* __restrict__ F, const half_t* __restrict__ Wt,
                                                          long long n, int d, long long v, float scale,
                                                          float* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                          float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 x 16 KB
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const long long img = (long long)blockIdx.x * 256 + wave * 32 + r;
    const long long irow = img < n ? img : n - 1;
    const half_t* frow = F + irow * d + 8 * hh;

    half8 bf[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        if (16 * s < d) {
            bf[s] = *(const half8*)(frow + 16 * s);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) bf[s][q] = (half_t)0.f;
        }
    }
    float lv[TOPM];
    int li[TOPM];
#pragma unroll
    for (int j = 0; j < TOPM; ++j) {
        lv[j] = -INFINITY;
        li[j] = -1;
    }
    float sm_m = -INFINITY, sm_z = 0.f;

    const int nd = d >> 6;                                  // 64-deep sub-steps per tile
    const int ntiles = (int)((v + 127) / 128);
    const int steps = ntiles * nd;
    // DMA source of this lane: wave w stages rows w*16 .. w*16+15 (2 instructions x 8 rows); lane -> (row lane/8, chunk lane%8)
    int src_row[2], src_col[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rowl = wave * 16 + p * 8 + (lane >> 3);
        src_row[p] = rowl;
        src_col[p] = ((lane & 7) ^ ((rowl >> 1) & 7)) << 3;
    }
    auto issue = [&](int tile, int dc, int slot) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            long long vr = (long long)tile * 128 + src_row[p];
            vr = vr < v ? vr : v - 1;                        // padded names re-read the last row; masked in the epilogue
            __builtin_amdgcn_global_load_lds((const void*)(Wt + vr * d + dc * 64 + src_col[p]),
                                             (sim_lds_ptr_t)(smem + slot * 16384 + wave * 2048 + p * 1024), 16, 0, 0);
        }
    };
    auto off128 = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };

    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;

    int ntile = 0, ndc = 0;
#pragma unroll
    for (int pre = 0; pre < 3; ++pre) {
        if (pre < steps) issue(ntile, ndc, pre);
        if (++ndc == nd) { ndc = 0; ++ntile; }
    }
    if (steps >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    half8 fa[4], fb[4];                                     // two fragment sets, alternating between the four k16 groups
    auto rd = [&](const char* slot, int k16, half8 (&f)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) f[cb] = *(const half8*)(slot + off128(cb * 32 + r, 2 * k16 + hh));
    };
    auto mm = [&](const half8 (&f)[4], const half8 b) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[cb], b, acc[cb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    rd(smem, 0, fa);
    int s = 0;                                              // flattened sub-step counter (ring slot = s & 3)
    for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll
        for (int dcc = 0; dcc < 8; ++dcc) {                 // d-chunk index is a compile-time constant: bf[] stays in registers
            if (dcc < nd) {
                const char* cur = smem + (s & 3) * 16384;
                rd(cur, 1, fb);
                mm(fa, bf[dcc * 4 + 0]);
                rd(cur, 2, fa);
                mm(fb, bf[dcc * 4 + 1]);
                if (s + 1 < steps) {
                    if (s + 2 >= steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();               // slot s+1 visible; slot s-1 free
                asm volatile("" ::: "memory");
                if (s + 3 < steps) issue(ntile, ndc, (s + 3) & 3);
                if (++ndc == nd) { ndc = 0; ++ntile; }
                rd(cur, 3, fb);
                mm(fa, bf[dcc * 4 + 2]);
                if (s + 1 < steps) rd(smem + ((s + 1) & 3) * 16384, 0, fa);
                mm(fb, bf[dcc * 4 + 3]);
                ++s;
            }
        }
        {
            // tile epilogue on UNSCALED dot products (scale > 0 is applied when the lists are written): one max3 tree per
            // 32-name block decides whether any of its 16 values can enter the list; only the last tile has padded names.
            const long long vbase = (long long)tile * 128 + 4 * hh;
            const bool last = tile == ntiles - 1;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (last) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (vbase + cb * 32 + (i & 3) + 8 * (i >> 2) >= v) acc[cb][i] = -INFINITY;
                }
                float bm = fmaxf(fmaxf(acc[cb][0], acc[cb][1]), acc[cb][2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) bm = fmaxf(fmaxf(bm, acc[cb][i]), acc[cb][i + 1]);
                bm = fmaxf(bm, acc[cb][15]);
                if (bm > lv[TOPM - 1]) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float val = acc[cb][i];
                        if (val > lv[TOPM - 1]) topm_insert(lv, li, val, (int)(vbase + cb * 32 + (i & 3) + 8 * (i >> 2)));
                    }
                }
                if (SOFTMAX) {
                    if (bm > -INFINITY) {
                        const float mn = fmaxf(sm_m, bm);
                        float z = sm_z * __expf((sm_m - mn) * scale);
#pragma unroll
                        for (int i = 0; i < 16; ++i) z += __expf((acc[cb][i] - mn) * scale);
                        sm_z = z;
                        sm_m = mn;
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
            }
        }
    }
    if (img < n) {
        float* cv = cand_val + (img * 2 + hh) * TOPM;
        int* ci = cand_idx + (img * 2 + hh) * TOPM;
#pragma unroll
        for (int j = 0; j < TOPM; ++j) {
            cv[j] = lv[j] * scale;
            ci[j] = li[j];
        }
        if (SOFTMAX) {
            stats[(img * 2 + hh) * 2] = sm_m * scale;
            stats[(img * 2 + hh) * 2 + 1] = sm_z;
        }
    }
}

// max ||w_v||^2 over the vocabulary (error-bound scale), one wave per row
__global__ void __launch_bounds__(256) wmax_kernel(const half_t* __restrict__ Wt, long long v, int d, unsigned* out_bits) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float best = 0.f;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < v; row += (long long)gridDim.x * 4) {
        float s = 0.f;
        for (int j = lane * 8; j < d; j += 512) {
            const half8 w8 = *(const half8*)(Wt + row * d + j);
#pragma unroll
            for (int q = 0; q < 8; ++q) s = fmaf((float)w8[q], (float)w8[q], s);
        }
        best = fmaxf(best, wave_sum_f32(s));
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 1.0001f));
}

struct SimHdr {
    unsigned wmax2_bits;
    int fb_cnt;
    int pad[14];
};

// exact float64 dot of one image row with one vocabulary row, all 64 lanes
__device__ __forceinline__ double dot64(const half_t* f, const half_t* w, int d, int lane) {
    double s = 0.0;
    for (int j = lane * 8; j < d; j += 512) {
        const half8 a = *(const half8*)(f + j);
        const half8 b = *(const half8*)(w + j);
#pragma unroll
        for (int q = 0; q < 8; ++q) s = fma((double)(float)a[q], (double)(float)b[q], s);
    }
    return wave_sum_f64(s);
}

// pass 2: one wave per image
template <bool SOFTMAX>
__global__ void __launch_bounds__(256) sim_refine_kernel(const half_t* __restrict__ F, const half_t* __restrict__ Wt,
                                                         long long n, int d, long long v, float scale, int k,
                                                         const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                         const float* __restrict__ stats, SimHdr* hdr, int* fb_list,
                                                         long long* idx_out, float* val_out) {
    const int lane = threadIdx.x & 63;
    const long long img = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= n) return;
    const half_t* f = F + img * d;
    // lanes 0..15 own one candidate each
    int myi = -1;
    float mya = -INFINITY;
    if (lane < 2 * TOPM) {
        myi = cand_idx[img * 2 * TOPM + lane];
        mya = cand_val[img * 2 * TOPM + lane];
    }
    double mye = -INFINITY;
    for (int c = 0; c < 2 * TOPM; ++c) {
        const int ci = __shfl(myi, c, 64);
        if (ci < 0) continue;                                   // wave-uniform
        const double e = (double)scale * dot64(f, Wt + (long long)ci * d, d, lane);
        if (lane == c) mye = e;
    }
    // rank of each candidate among the 16: (value desc, index asc)
    int rank = 0;
    for (int c = 0; c < 2 * TOPM; ++c) {
        const double oe = __shfl(mye, c, 64);
        const int oi = __shfl(myi, c, 64);
        if (oi >= 0 && (oe > mye || (oe == mye && oi < myi))) ++rank;
    }
    // certification: a non-candidate of half h has approx <= list_h[TOPM-1]
    const float a0 = __shfl(mya, TOPM - 1, 64), a1 = __shfl(mya, 2 * TOPM - 1, 64);
    const int i0 = __shfl(myi, TOPM - 1, 64), i1 = __shfl(myi, 2 * TOPM - 1, 64);
    // if a list is not full, that half has no non-candidates at all
    float astar = -INFINITY;
    if (i0 >= 0) astar = fmaxf(astar, a0);
    if (i1 >= 0) astar = fmaxf(astar, a1);
    double f2 = 0.0;
    for (int j = lane; j < d; j += 64) {
        const double x = (double)(float)f[j];
        f2 = fma(x, x, f2);
    }
    f2 = wave_sum_f64(f2);
    const float wmax = sqrtf(__uint_as_float(hdr->wmax2_bits));
    const float E = 1.5f * fabsf(scale) * ((float)d * 5.9604645e-8f + 2.4e-7f) * (float)sqrt(f2) * wmax + 1e-30f;
    // exact value of the k-th ranked candidate
    double kth = -INFINITY;
    {
        const unsigned long long m = __ballot(rank == k - 1 && myi >= 0);
        if (m) kth = __shfl(mye, __ffsll((long long)m) - 1, 64);
    }
    const bool certified = (astar == -INFINITY) || (kth > (double)astar + (double)E);
    if (!certified) {
        if (lane == 0) {
            const int pos = atomicAdd(&hdr->fb_cnt, 1);
            fb_list[pos] = (int)img;
        }
        return;
    }
    if (myi >= 0 && rank < k) {
        idx_out[img * k + rank] = myi;
        float o = (float)mye;
        if (SOFTMAX) {
            const float m0 = stats[img * 4], z0 = stats[img * 4 + 1], m1 = stats[img * 4 + 2], z1 = stats[img * 4 + 3];
            const float mm = fmaxf(m0, m1);
            const float z = z0 * __expf(m0 - mm) + z1 * __expf(m1 - mm);
            o = __expf((float)mye - mm) / z;
        }
        val_out[img * k + rank] = o;
    }
}

// pass 3 (rare): exact full-row top-k in float64 for uncertified rows; one block (256 threads) per row
template <bool SOFTMAX>
__global__ void __launch_bounds__(256) sim_exact_kernel(const half_t* __restrict__ F, const half_t* __restrict__ Wt, int d,
                                                        long long v, float scale, int k, const SimHdr* hdr,
                                                        const int* fb_list, long long* idx_out, float* val_out) {
    __shared__ float fs[1024];
    __shared__ double cval[256 * TOPM];
    __shared__ int cidx[256 * TOPM];
    __shared__ double red_m[256];
    __shared__ double red_z[256];
    const int cnt = hdr->fb_cnt;
    for (int fb = blockIdx.x; fb < cnt; fb += gridDim.x) {
        const long long img = fb_list[fb];
        __syncthreads();
        for (int j = threadIdx.x; j < d; j += 256) fs[j] = (float)F[img * d + j];
        __syncthreads();
        double lv[TOPM];
        int li[TOPM];
#pragma unroll
        for (int j = 0; j < TOPM; ++j) { lv[j] = -INFINITY; li[j] = -1; }
        double m = -INFINITY, z = 0.0;
        for (long long vi = threadIdx.x; vi < v; vi += 256) {
            const half_t* w = Wt + vi * d;
            double s = 0.0;
            for (int j = 0; j < d; j += 8) {
                const half8 b = *(const half8*)(w + j);
#pragma unroll
                for (int q = 0; q < 8; ++q) s = fma((double)fs[j + q], (double)(float)b[q], s);
            }
            s *= (double)scale;
            if (SOFTMAX) {
                const double mn = s > m ? s : m;
                z = z * exp(m - mn) + exp(s - mn);
                m = mn;
            }
            if (s > lv[TOPM - 1]) {        // ascending vi per thread: strict > keeps the lower index on ties
#pragma unroll
                for (int j = TOPM - 1; j >= 1; --j) {
                    const bool up = s > lv[j - 1];
                    const bool here = s > lv[j];
                    lv[j] = up ? lv[j - 1] : (here ? s : lv[j]);
                    li[j] = up ? li[j - 1] : (here ? (int)vi : li[j]);
                }
                if (s > lv[0]) { lv[0] = s; li[0] = (int)vi; }
            }
        }
#pragma unroll
        for (int j = 0; j < TOPM; ++j) {
            cval[threadIdx.x * TOPM + j] = lv[j];
            cidx[threadIdx.x * TOPM + j] = li[j];
        }
        red_m[threadIdx.x] = m;
        red_z[threadIdx.x] = z;
        __syncthreads();
        if (threadIdx.x == 0) {
            double mm = -INFINITY, zz = 0.0;
            if (SOFTMAX) {
                for (int t = 0; t < 256; ++t) mm = red_m[t] > mm ? red_m[t] : mm;
                for (int t = 0; t < 256; ++t)
                    if (red_z[t] > 0.0) zz += red_z[t] * exp(red_m[t] - mm);
            }
            for (int out = 0; out < k; ++out) {
                int bt = -1;
                for (int t = 0; t < 256 * TOPM; ++t) {
                    if (cidx[t] < 0) continue;
                    if (bt < 0 || cval[t] > cval[bt] || (cval[t] == cval[bt] && cidx[t] < cidx[bt])) bt = t;
                }
                if (bt < 0) break;
                idx_out[img * k + out] = cidx[bt];
                val_out[img * k + out] = SOFTMAX ? (float)(exp(cval[bt] - mm) / zz) : (float)cval[bt];
                cidx[bt] = -1;
            }
        }
    }
}

extern "C" size_t scd_sim_topk_ws_bytes(int64_t n, int d, int64_t v, int k) {
    (void)d; (void)v; (void)k;
    return 64 + scd_align((size_t)n * 2 * TOPM * 4) * 2 + scd_align((size_t)n * 16) + scd_align((size_t)n * 4) + 256;
}

extern "C" int scd_sim_topk(scd_handle h, const void* F, const void* Wt, int64_t n, int d, int64_t v, float scale, int k,
                            int mode, int64_t* idx_out, float* val_out, int32_t* fallback_rows_out, void* ws,
                            size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && F && Wt && idx_out && val_out && ws, "scd_sim_topk: null argument");
    SCD_REQUIRE(n > 0 && v > 0 && n < (1ll << 31) && v < (1ll << 31), "scd_sim_topk: bad shape n=%lld v=%lld", (long long)n, (long long)v);
    SCD_REQUIRE(d > 0 && d <= 512 && d % 64 == 0, "scd_sim_topk: d=%d must be a multiple of 64, <= 512", d);
    SCD_REQUIRE(k >= 1 && k <= TOPM && k <= v, "scd_sim_topk: k=%d must be in [1,%d] and <= v", k, TOPM);
    SCD_REQUIRE(mode == SCD_SIM_RAW || mode == SCD_SIM_SOFTMAX, "scd_sim_topk: bad mode %d", mode);
    SCD_REQUIRE(scale > 0.f, "scd_sim_topk: scale must be positive");
    SCD_REQUIRE(ws_bytes >= scd_sim_topk_ws_bytes(n, d, v, k), "scd_sim_topk: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    char* w = (char*)ws;
    SimHdr* hdr = (SimHdr*)w;
    const size_t csz = scd_align((size_t)n * 2 * TOPM * 4);
    float* cval = (float*)(w + 64);
    int* cidx = (int*)(w + 64 + csz);
    float* stats = (float*)(w + 64 + 2 * csz);
    int* fb = (int*)(w + 64 + 2 * csz + scd_align((size_t)n * 16));
    const half_t* f = (const half_t*)F;
    const half_t* wt = (const half_t*)Wt;
    SCD_HIP(hipMemsetAsync(hdr, 0, 64, st));
    wmax_kernel<<<256, 256, 0, st>>>(wt, v, d, &hdr->wmax2_bits);
    const unsigned g1 = (unsigned)scd_cdiv(n, 256), g2 = (unsigned)scd_cdiv(n, 4);
    static bool attr = false;
    if (!attr) {
        SCD_HIP(hipFuncSetAttribute((const void*)sim_topk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        SCD_HIP(hipFuncSetAttribute((const void*)sim_topk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr = true;
    }
    if (mode == SCD_SIM_SOFTMAX) {
        sim_topk_kernel<true><<<g1, 512, 65536, st>>>(f, wt, n, d, v, scale, cval, cidx, stats);
        sim_refine_kernel<true><<<g2, 256, 0, st>>>(f, wt, n, d, v, scale, k, cval, cidx, stats, hdr, fb, (long long*)idx_out, val_out);
        sim_exact_kernel<true><<<256, 256, 0, st>>>(f, wt, d, v, scale, k, hdr, fb, (long long*)idx_out, val_out);
    } else {
        sim_topk_kernel<false><<<g1, 512, 65536, st>>>(f, wt, n, d, v, scale, cval, cidx, stats);
        sim_refine_kernel<false><<<g2, 256, 0, st>>>(f, wt, n, d, v, scale, k, cval, cidx, stats, hdr, fb, (long long*)idx_out, val_out);
        sim_exact_kernel<false><<<256, 256, 0, st>>>(f, wt, d, v, scale, k, hdr, fb, (long long*)idx_out, val_out);
    }
    if (fallback_rows_out) SCD_HIP(hipMemcpyAsync(fallback_rows_out, &hdr->fb_cnt, 4, hipMemcpyDeviceToDevice, st));
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_f16_kernel(const half_t* __restrict__ in, long long r, long long c,
                                                            half_t* __restrict__ out) {
    __shared__ half_t tile[64][66];
    const long long c0 = (long long)blockIdx.x * 64, r0 = (long long)blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int y = i >> 6, x = i & 63;
        tile[y][x] = (r0 + y < r && c0 + x < c) ? in[(r0 + y) * c + c0 + x] : (half_t)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int y = i >> 6, x = i & 63;       // out row = c0+y, col = r0+x
        if (c0 + y < c && r0 + x < r) out[(c0 + y) * r + r0 + x] = tile[x][y];
    }
}
extern "C" int scd_transpose_f16(scd_handle h, const void* in, int64_t r, int64_t c, void* out, void* stream_) {
    SCD_REQUIRE(h && in && out && r > 0 && c > 0, "scd_transpose_f16: bad arguments");
    transpose_f16_kernel<<<dim3((unsigned)scd_cdiv(c, 64), (unsigned)scd_cdiv(r, 64)), 256, 0, (hipStream_t)stream_>>>(
        (const half_t*)in, r, c, (half_t*)out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const half_t* __restrict__ Wt, const long long* __restrict__ idx,
                                                          long long m, int d, half_t* __restrict__ out) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const long long src = idx[row];
    for (int j = threadIdx.x & 63; j < d; j += 64) out[row * d + j] = Wt[src * d + j];
}
extern "C" int scd_gather_rows_f16(scd_handle h, const void* Wt, const int64_t* idx, int64_t m, int d, void* out,
                                   void* stream_) {
    SCD_REQUIRE(h && Wt && idx && out && m > 0 && d > 0, "scd_gather_rows_f16: bad arguments");
    gather_rows_kernel<<<(unsigned)scd_cdiv(m, 4), 256, 0, (hipStream_t)stream_>>>((const half_t*)Wt, (const long long*)idx, m,
                                                                                 d, (half_t*)out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// F.normalize(x, dim=-1): x / max(||x||, 1e-12); one wave per row, float32 math on the stored values
template <typename T>
__global__ void __launch_bounds__(256) l2norm_kernel(const T* __restrict__ x, long long n, int d, T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float v = (float)x[row * d + j];
        s = fmaf(v, v, s);
    }
    s = wave_sum_f32(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int j = lane; j < d; j += 64) out[row * d + j] = (T)((float)x[row * d + j] * inv);
}
extern "C" int scd_l2norm_rows(scd_handle h, const void* x, int dtype, int64_t n, int d, void* out, void* stream_) {
    SCD_REQUIRE(h && x && out && n > 0 && d > 0, "scd_l2norm_rows: bad arguments");
    const unsigned g = (unsigned)scd_cdiv(n, 4);
    if (dtype == SCD_F32) l2norm_kernel<float><<<g, 256, 0, (hipStream_t)stream_>>>((const float*)x, n, d, (float*)out);
    else if (dtype == SCD_F16) l2norm_kernel<half_t><<<g, 256, 0, (hipStream_t)stream_>>>((const half_t*)x, n, d, (half_t*)out);
    else SCD_REQUIRE(false, "scd_l2norm_rows: bad dtype %d", dtype);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// zeroshot_classifier pooling (local_utils/clip_lang_util.py:103-107): per name, L2-normalise its T prompt
// embeddings, average, L2-normalise, and store as COLUMN `name` of out[d, n_names] (torch.stack(dim=1)).
// One block (4 waves) per name; float32 accumulation; d <= 1024.
__global__ void __launch_bounds__(256) prompt_pool_kernel(const half_t* __restrict__ emb, int n_names, int t_per, int d,
                                                          long long col0, long long ld_out, half_t* __restrict__ out) {
    __shared__ float part[4][1024];
    __shared__ float red[4];
    const int name = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int t = wave; t < t_per; t += 4) {
        const half_t* row = emb + ((size_t)name * t_per + t) * d;
        float v[16];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = i * 64 + lane;
            v[i] = j < d ? (float)row[j] : 0.f;
            s = fmaf(v[i], v[i], s);
        }
        const float inv = 1.0f / sqrtf(wave_sum_f32(s));
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(v[i], inv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave][i * 64 + lane] = acc[i];
    __syncthreads();
    float s = 0.f;
    float m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = q * 256 + threadIdx.x;
        m[q] = (part[0][j] + part[1][j] + part[2][j] + part[3][j]) / (float)t_per;
        if (j >= d) m[q] = 0.f;
        s = fmaf(m[q], m[q], s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / sqrtf(red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = q * 256 + threadIdx.x;
        if (j < d) out[(size_t)j * ld_out + col0 + name] = (half_t)(m[q] * inv);
    }
}
extern "C" int scd_prompt_pool(scd_handle h, const void* emb, int n_names, int t_per, int d, int64_t col0, int64_t ld_out,
                               void* out, void* stream_) {
    SCD_REQUIRE(h && emb && out && n_names > 0 && t_per > 0 && d > 0 && d <= 1024, "scd_prompt_pool: bad arguments");
    prompt_pool_kernel<<<n_names, 256, 0, (hipStream_t)stream_>>>((const half_t*)emb, n_names, t_per, d, col0, ld_out, (half_t*)out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
