// K-Means hot path for gfx950: E-step (fp16 MFMA filter + float64 refine), exact row distances,
// M-step partial sums, centre finalisation, incremental k-means++.
//
// Replaces (paths under /root/reference):
//   pairwise_distance            local_utils/sskm_constrained.py:189-224
//   torch.min(dist, 1)           gcd/methods/clustering/faster_mix_k_means_pytorch.py:140,192
//   per-cluster mean loop        local_utils/sskm_constrained.py:125-128
//   kpp                          local_utils/sskm_constrained.py:28-44
//
// Decision semantics (shared with oracle/kmeans_oracle.py): every argmin is taken on the float64
// difference-form distance, ties to the lowest index.  The MFMA pass only FILTERS: it evaluates
// ||c'||^2 - 2 x'.c' on a centred, power-of-two scaled fp16 copy of the data with an a-priori error
// bound; rows whose best/second margin is inside the bound are re-evaluated exactly in float64.
#include "common.h"
#include <stdlib.h>
#include <map>
#include <mutex>

// ------------------------------------------------------------------------------------------------
// prepared data set layout (scd_kmeans_prepare):
//   [0,64)            PrepHdr
//   [64, 64+8*Dp)     mu (double[Dp], zero padded)
//   xnorm_off         float[n]    ||x'_i||  (x' = (x-mu)*scale, float64 norm)
//   xh_off            half[n*Dp]  fp16(x'), zero padded columns
struct PrepHdr {
    float scale;        // 2^e
    unsigned maxabs_bits;
    int d, dp;
    long long n;
    unsigned long long xnorm_off, xh_off;
    int pad[6];
};
static_assert(sizeof(PrepHdr) == 64, "PrepHdr must be 64 bytes");

static inline int dpad(int d) { return (d + 127) / 128 * 128; }

extern "C" size_t scd_kmeans_prep_bytes(int64_t n, int d) {
    size_t dp = dpad(d);
    return scd_align(64 + 8 * dp) + scd_align(4 * (size_t)n) + scd_align(2 * ((size_t)n + 32) * dp) + 256;   // + one unit of padding (estep_stream_kernel)
}

// column sums in float64: block (0..gridDim.x) strides over rows, thread t owns columns t, t+256, ...
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ X, long long n, int d, double* mu) {
    for (int c = threadIdx.x; c < d; c += 256) {
        double s = 0.0;
        for (long long r = blockIdx.x; r < n; r += gridDim.x) s += (double)X[r * d + c];
        atomicAdd(&mu[c], s);
    }
}
__global__ void mu_finish_kernel(double* mu, int d, long long n) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < d) mu[c] = mu[c] / (double)n;
}
__global__ void __launch_bounds__(256) maxabs_kernel(const float* __restrict__ X, long long n, int d, const double* mu,
                                                     unsigned* maxabs_bits) {
    float m = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
        double muc = mu[c];
        for (long long r = blockIdx.x; r < n; r += gridDim.x) m = fmaxf(m, fabsf((float)((double)X[r * d + c] - muc)));
    }
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) atomicMax(maxabs_bits, __float_as_uint(m));
}
__global__ void scale_kernel(PrepHdr* hdr) {
    float m = __uint_as_float(hdr->maxabs_bits);
    int e = 0;
    if (m > 0.f && isfinite(m)) {
        int ex;
        frexpf(m, &ex);          // m = f * 2^ex, f in [0.5,1)
        e = 4 - ex;              // max |x'| in [8,16)
    }
    hdr->scale = ldexpf(1.0f, e);
}
// one wave per row: x' = (x-mu)*scale -> fp16, ||x'||
__global__ void __launch_bounds__(256) center_kernel(const float* __restrict__ X, long long n, int d, int dp,
                                                     const PrepHdr* hdr, const double* mu, float* xnorm, half_t* xh) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double sc = (double)hdr->scale;
    double ss = 0.0;
    for (int c = lane; c < dp; c += 64) {
        double v = 0.0;
        if (c < d) v = ((double)X[row * d + c] - mu[c]) * sc;
        ss += v * v;
        xh[row * dp + c] = (half_t)(float)v;
    }
    ss = wave_sum_f64(ss);
    if (lane == 0) xnorm[row] = (float)sqrt(ss);
}

extern "C" int scd_kmeans_prepare(scd_handle h, const float* X, int64_t n, int d, void* prep, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_prepare");
    SCD_REQUIRE(h && X && prep && n > 0 && d > 0, "scd_kmeans_prepare: bad arguments (n=%lld d=%d)", (long long)n, d);
    hipStream_t st = (hipStream_t)stream_;
    const int dp = dpad(d);
    PrepHdr hh = {};
    hh.d = d; hh.dp = dp; hh.n = n;
    hh.xnorm_off = scd_align(64 + 8 * (size_t)dp);
    hh.xh_off = hh.xnorm_off + scd_align(4 * (size_t)n);
    char* p = (char*)prep;
    SCD_HIP(hipMemsetAsync(p, 0, 64 + 8 * (size_t)dp, st));
    SCD_HIP(hipMemcpyAsync(p, &hh, sizeof(hh), hipMemcpyHostToDevice, st));
    double* mu = (double*)(p + 64);
    int blocks = (int)((n < 1024) ? n : 1024);
    colsum_kernel<<<blocks, 256, 0, st>>>(X, n, d, mu);
    mu_finish_kernel<<<(d + 255) / 256, 256, 0, st>>>(mu, d, n);
    maxabs_kernel<<<blocks, 256, 0, st>>>(X, n, d, mu, &((PrepHdr*)p)->maxabs_bits);
    scale_kernel<<<1, 1, 0, st>>>((PrepHdr*)p);
    center_kernel<<<(unsigned)scd_cdiv(n, 4), 256, 0, st>>>(X, n, d, dp, (const PrepHdr*)p, mu, (float*)(p + hh.xnorm_off),
                                                             (half_t*)(p + hh.xh_off));
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// E-step workspace:  [0,64) EHdr | cn float[Kp] | ch half[Kp*Dp] | ct float[Dp*Kp] (centres transposed, exact)
//                       | pair list int32[n] | pair candidates int32[n] | full list int32[n] | chf half[Kp*Dp] (fragment order)
//                       | tkey float[3n] | tidx int32[3n] (multi-pass streaming filter, K > 128)
struct EHdr {
    unsigned cmax_bits;   // max ||c'||
    int flag_cnt;         // rows whose exact argmin is among two known candidates
    int full_cnt;         // rows that need the exact distance to every centre
    int pad[13];
};
static inline int kpad(int k) { return (k + 127) / 128 * 128; }

extern "C" size_t scd_kmeans_estep_ws_bytes(int64_t n, int d, int k) {
    size_t kp = kpad(k), dp = dpad(d);
    return 64 + scd_align(4 * kp) + scd_align(2 * kp * dp) + scd_align(4 * kp * dp) + 3 * scd_align(4 * (size_t)n) + scd_align(2 * kp * dp) + 2 * scd_align(12 * (size_t)n) + 256;
}

// one block per (padded) centre: c' = (c-mu)*scale -> fp16; cn = ||c'||^2 (float64 -> float32)
// E-step operands of ONE centre (block-wide, 256 threads): norm, fp16 centred / scaled row (row-major and MFMA-fragment order),
// fp32 transposed copy.  Called by prep_centers_kernel and, fused, by finalize_kernel (the block that has just produced the centre).
__device__ __forceinline__ void prep_center_row(const float* C, int c, int k, int d, int dp, const PrepHdr* hdr,
                                                const double* mu, EHdr* eh, float* cn, half_t* ch, float* ct, int kp,
                                                int zero_counts, half_t* chf) {
    __shared__ double red[4];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    if (zero_counts && c == 0 && threadIdx.x == 0) { eh->flag_cnt = 0; eh->full_cnt = 0; }   // streaming path: no memset launch
    __syncthreads();
    const double sc = (double)hdr->scale;
    double ss = 0.0;
    for (int j = threadIdx.x; j < dp; j += 256) {
        double v = 0.0;
        if (c < k && j < d) {
            float cv = C[(size_t)c * d + j];
            if (!isfinite(cv)) bad = 1;
            v = ((double)cv - mu[j]) * sc;
        }
        ss += v * v;
        ch[(size_t)c * dp + j] = (half_t)(float)v;
        // fragment order of estep_stream_kernel: [wave c/32][segment j/128][k-step][lane (c%32) + 32 hh][8]
        if (chf) chf[(((((size_t)(c >> 5) * (dp >> 7) + (j >> 7)) * 8 + ((j & 127) >> 4)) * 64 + (c & 31) + 32 * ((j >> 3) & 1)) << 3) + (j & 7)] = (half_t)(float)v;
        // untouched float32 values, transposed.  ct == null: the caller transposes all centres at once (ct_transpose_kernel) - written
        // from here, a centre's column is dp separate 4-byte writes at a stride of 4 kp bytes, and with a thousand centres doing that
        // at once finalize_kernel took 113 us at K = 1000 (round 6)
        if (ct) ct[(size_t)j * kp + c] = (c < k && j < d) ? C[(size_t)c * d + j] : NAN;
    }
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const bool dead = (c >= k) || bad;
    if (dead) {   // padded or NaN centre (empty cluster): can never win
        for (int j = threadIdx.x; j < dp; j += 256) {
            ch[(size_t)c * dp + j] = (half_t)0.f;
            if (chf) chf[(((((size_t)(c >> 5) * (dp >> 7) + (j >> 7)) * 8 + ((j & 127) >> 4)) * 64 + (c & 31) + 32 * ((j >> 3) & 1)) << 3) + (j & 7)] = (half_t)0.f;
        }
    }
    if (threadIdx.x == 0) {
        double t = red[0] + red[1] + red[2] + red[3];
        if (dead) {
            cn[c] = INFINITY;
        } else {
            cn[c] = (float)t;
            atomicMax(&eh->cmax_bits, __float_as_uint((float)sqrt(t) * 1.0000002f));
        }
    }
}

// ct [dp][kp] = C^T (float32 values untouched; NaN in the padding) through a 32 x 32 LDS tile: whole 128-byte pieces both ways
__global__ void __launch_bounds__(256) ct_transpose_kernel(const float* __restrict__ C, int k, int d, int dp, int kp, float* __restrict__ ct) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, j0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, j = j0 + tx;
        t[ty + 8 * i][tx] = (c < k && j < d) ? C[(size_t)c * d + j] : NAN;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = j0 + ty + 8 * i, c = c0 + tx;
        if (j < dp && c < kp) ct[(size_t)j * kp + c] = t[tx][ty + 8 * i];
    }
}
// many centres: the transposed copy by ct_transpose_kernel behind the per-centre blocks instead of from inside them
static inline bool ct_separate(int kp) { return kp >= 512; }
// Only estep_refine_full_kernel reads the transposed copy: the legacy path (Kp > 2048 or Dp > 768) and, in -DSCD_ABLATE builds, the
// split-refine switch of the streaming path.  The single-pass and the streaming path of the default build re-evaluate rows against the
// row-major centres (refine_full_row), so there the copy is not written at all (2 MB of scattered writes per E-step at K = 1000).
static inline bool ct_dead(int dp, int kp) {
#ifdef SCD_ABLATE
    (void)dp; (void)kp;
    return false;
#else
    static const int use_stream = getenv("SCD_ESTEP_STREAM") ? atoi(getenv("SCD_ESTEP_STREAM")) : 1;
    return use_stream && kp <= 2048 && dp <= 768;
#endif
}
// what the per-centre blocks are handed / whether the tiled transpose follows them
static inline float* ct_inline(float* ct, int dp, int kp) { return (ct_dead(dp, kp) || ct_separate(kp)) ? nullptr : ct; }
static inline bool ct_after(int dp, int kp) { return !ct_dead(dp, kp) && ct_separate(kp); }
// the fragment-order copy `chf` serves the streaming filter (K <= 128) only; on the single-pass path of 128 < K <= 2048 at Dp = 512 its
// place holds 16 bytes per centre written by estep_ext_kernel, so the per-centre blocks do not write it there
static inline bool chf_unused(int dp, int kp) {
    static const int use_stream = getenv("SCD_ESTEP_STREAM") ? atoi(getenv("SCD_ESTEP_STREAM")) : 1;
    static const int use_rb = getenv("SCD_ESTEP_RB") ? atoi(getenv("SCD_ESTEP_RB")) : 1;
    return use_stream && use_rb && dp == 512 && kp > 128 && kp <= 2048;
}

__global__ void __launch_bounds__(256) prep_centers_kernel(const float* __restrict__ C, int k, int d, int dp,
                                                           const PrepHdr* hdr, const double* mu, EHdr* eh, float* cn,
                                                           half_t* ch, float* ct, int kp, int zero_counts, half_t* chf) {
    prep_center_row(C, blockIdx.x, k, d, dp, hdr, mu, eh, cn, ch, ct, kp, zero_counts, chf);
}

// MFMA filter.  Block = 4 waves = 128 points; each wave owns 32 points (MFMA columns) against a chunk
// of 128 centres (4 x 32 MFMA rows) staged through LDS; v_mfma_f32_32x32x16_f16:
//   A[row = centre r][k = 8h+j]  from LDS (XOR-swizzled 256-B rows, ds_read_b128)
//   B[k = 8h+j][col = point r]   straight from global (16 B per lane, each row streamed once)
//   D[row = (reg&3)+8(reg>>2)+4h][col = point r]
// so every lane ends with 16 scores per 32-centre block for ONE point: the running best/second is
// in-lane; lanes r and r+32 are merged once at the end.
__global__ void __launch_bounds__(256) estep_mfma_kernel(const half_t* __restrict__ xh, const float* __restrict__ xnorm,
                                                         const half_t* __restrict__ ch, const float* __restrict__ cn,
                                                         EHdr* eh, int* flag_list, int* flag_cand, int* full_list,
                                                         long long n, int dp, int kp, int32_t* __restrict__ labels) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 256];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const long long point = (long long)blockIdx.x * 128 + wave * 32 + r;
    const long long prow = point < n ? point : n - 1;
    const half_t* xrow = xh + prow * dp + 8 * hh;

    // running three smallest scores of this lane (b0 <= b1 <= b2) and the centres of the first two
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    int i0 = 0, i1 = 0;

    for (int kc = 0; kc < kp; kc += 128) {
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;

        for (int dc = 0; dc < dp; dc += 128) {
            half8 bf[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) bf[s] = *(const half8*)(xrow + dc + 16 * s);
            __syncthreads();   // previous tile fully consumed
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int row = p * 16 + (tid >> 4);
                const int c16 = tid & 15;
                const uint4 v = *(const uint4*)(ch + (size_t)(kc + row) * dp + dc + 8 * c16);
                *(uint4*)(lds + row * 256 + ((c16 ^ (row & 15)) << 4)) = v;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int row = cb * 32 + r;
                    const half8 a = *(const half8*)(lds + row * 256 + (((2 * s + hh) ^ (row & 15)) << 4));
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[s], acc[cb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int centre = kc + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const float s = cn[centre] - 2.0f * acc[cb][i];
                if (s < b0) {
                    b2 = b1; b1 = b0; i1 = i0; b0 = s; i0 = centre;
                } else if (s < b1) {
                    b2 = b1; b1 = s; i1 = centre;
                } else if (s < b2) {
                    b2 = s;
                }
            }
        }
    }
    // merge the two half-wave lanes that share a point: three smallest of the six, order (value, centre)
    const float o0 = __shfl_xor(b0, 32, 64), o1 = __shfl_xor(b1, 32, 64), o2 = __shfl_xor(b2, 32, 64);
    const int p0 = __shfl_xor(i0, 32, 64), p1 = __shfl_xor(i1, 32, 64);
    float m0, m1, m2;
    int j0, j1;
    {
        // candidates in ascending order within each list; pick three by repeated front comparison
        float av[3] = {b0, b1, b2}, ov[3] = {o0, o1, o2};
        int ai[3] = {i0, i1, 0x7fffffff}, oi[3] = {p0, p1, 0x7fffffff};
        int a = 0, o = 0;
        float mv[3];
        int mi[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const bool take_a = av[a > 2 ? 2 : a] < ov[o > 2 ? 2 : o] ||
                                (av[a > 2 ? 2 : a] == ov[o > 2 ? 2 : o] && ai[a > 2 ? 2 : a] <= oi[o > 2 ? 2 : o]);
            if (take_a) { mv[t] = av[a > 2 ? 2 : a]; mi[t] = ai[a > 2 ? 2 : a]; ++a; }
            else { mv[t] = ov[o > 2 ? 2 : o]; mi[t] = oi[o > 2 ? 2 : o]; ++o; }
        }
        m0 = mv[0]; m1 = mv[1]; m2 = mv[2]; j0 = mi[0]; j1 = mi[1];
    }
    if (hh == 0 && point < n) {
        labels[point] = j0;
        const float cmax = __uint_as_float(eh->cmax_bits);
        const float sq = sqrtf((float)dp);
        // |s~ - s| <= A*||x'|| + B : fp16 rounding of both operands (2^-10), fp32 accumulation (dp*2^-24),
        // fp16 subnormal flush-free absolute term, final fp32 ops; x1.5 safety.
        const float A = 1.5f * (2.02f * (9.765625e-4f + dp * 5.9604645e-8f) * cmax + 4.8e-7f * cmax + 6.0e-8f * sq);
        const float B = 1.5f * (6.0e-8f * sq * cmax + 2.4e-7f * cmax * cmax);
        const float E = A * xnorm[point] + B;
        if (!(m1 - m0 > 2.0f * E)) {       // also catches NaN
            // the true argmin is among the centres whose filtered score is within 2E of the best: if the third
            // smallest is already outside, only {j0, j1} need the exact distance; otherwise all K do.
            const bool pair_only = (m2 - m0 > 2.0f * E) && j1 != 0x7fffffff;
            if (pair_only) {
                const int pos = atomicAdd(&eh->flag_cnt, 1);
                flag_list[pos] = (int)point;
                flag_cand[pos] = (j0 & 0xffff) | (j1 << 16);
            } else {
                full_list[atomicAdd(&eh->full_cnt, 1)] = (int)point;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming MFMA filter for K <= 128, D <= 768 (the SSKM shapes of BASELINE C1-C3): the fp16 data set is read from HBM
// exactly once, in whole cache lines, by LDS-DMA, and nothing else touches global memory while it streams.
//   * unit of work = 32 consecutive rows = one CONTIGUOUS 64*Dp-byte span (48 KB at D = 768).  Block b takes units b, b+G,
//     b+2G, ...: at any moment the grid reads one contiguous front of G units, every CU a sequential 48 KB of it;
//   * the 128 (padded) centres live in REGISTERS for the whole kernel: wave w holds centres 32w..32w+31 as the A
//     operands of v_mfma_f32_32x32x16_f16 (NCH*8 fragments of 4 registers = 192 at D = 768), one wave per SIMD;
//   * a unit lands in one slot of a 3-slot (4 below D = 640) LDS ring by global_load_lds_dwordx4, 1 KB of consecutive
//     bytes per instruction; the LDS image is the unit itself with the 16-B pieces of every 256-B segment XOR-swizzled by
//     row&15 (applied on the SOURCE address), so the B-fragment ds_read_b128 (lane = point) is conflict-free; two units
//     (96 KB per CU) are always in flight;
//   * one barrier per unit; every wave reads the whole unit (B operand) against its own 32 centres: 8*NCH MFMAs on two
//     accumulators; fragment reads run four k-steps ahead;
//   * per unit each lane owns 16 scores of one point: the three smallest are kept with a 4-instruction min/med3/max network
//     on KEYS = score bits with the centre index in the low mantissa bits (5 bits in-lane, 7 bits after the merge: a
//     relative perturbation < 2^-16 that is added to the a-priori bound E); lanes r / r+32 merge by shuffle, the four
//     waves through a double-buffered 1.5 KB LDS patch read by 32 threads after the NEXT unit's barrier; the merged
//     triples of all the block's rows stay in LDS;
//   * after the stream has drained: E is evaluated (||x'|| was fetched at kernel start), labels are stored, flagged rows
//     are appended to the refine lists with ONE global atomic per block and list.
// The prepared data set carries 32 rows of padding behind row n-1 (scd_kmeans_prep_bytes), so the last unit needs no clamp.
#define ES_RMAX 768
#define ES_RING 147456
#define ES_LDS (ES_RING + 512 + 2 * 4 * 3 * 32 * 4 + 3 * ES_RMAX * 4 + 64)

__device__ __forceinline__ float es_min(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float es_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float es_med3(float a, float b, float c) { float d; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// the value held by lane l ^ 32 (v_permlane32_swap exchanges the upper half of its first operand with the lower half of the second)
__device__ __forceinline__ float es_swap32(float v) {
    const unsigned u = __float_as_uint(v);
    const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // builtin: hipcc pads the permlane hazards
    return __uint_as_float((threadIdx.x & 32) ? p[0] : p[1]);
}
// insert key k into the ascending triple (b0, b1, b2)
__device__ __forceinline__ void es_insert(float& b0, float& b1, float& b2, float k) {
    const float t = es_max(b1, k);
    b2 = es_min(b2, t);
    b1 = es_med3(b0, b1, k);
    b0 = es_min(b0, k);
}

// out[u] = wave_sum_f64(a[u]) for eight values at once.  Step o of the butterfly adds lanes l and l ^ o; here, in the first three steps,
// each lane keeps only half of its values and sends the partner the other half (lane bit 5 / 4 / 3 <-> bit 2 / 1 / 0 of u), so 4 + 2 + 1
// exchanges replace 3 x 8, and three more finish all eight sums in one register.  The same pairs are added in the same order (IEEE
// addition is commutative), hence bit-identical totals; v_readlane hands them out.
__device__ __forceinline__ void wave_sum8_f64(const double (&a)[8], double (&out)[8]) {
    const int lane = threadIdx.x & 63;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
    double b[4], c[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double keep = h5 ? a[u + 4] : a[u], send = h5 ? a[u] : a[u + 4];
        b[u] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double keep = h4 ? b[u + 2] : b[u], send = h4 ? b[u] : b[u + 2];
        c[u] = keep + __shfl_xor(send, 16, 64);
    }
    const double keep = h3 ? c[1] : c[0], send = h3 ? c[0] : c[1];
    double e = keep + __shfl_xor(send, 8, 64);
    e += __shfl_xor(e, 4, 64);
    e += __shfl_xor(e, 2, 64);
    e += __shfl_xor(e, 1, 64);
    const long long bits = __builtin_bit_cast(long long, e);
    const int lo = (int)bits, hi = (int)(bits >> 32);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int src = 32 * ((u >> 2) & 1) + 16 * ((u >> 1) & 1) + 8 * (u & 1);
        const unsigned rl = (unsigned)__builtin_amdgcn_readlane(lo, src), rh = (unsigned)__builtin_amdgcn_readlane(hi, src);
        out[u] = __builtin_bit_cast(double, ((unsigned long long)rh << 32) | rl);
    }
}

// exact re-evaluation of ONE flagged row (float64 difference form), shared by estep_refine_both_kernel and the tail of
// estep_stream_kernel.  Pair form: one wave, the two candidate centres.  Full form: one 256-thread block, every centre.
__device__ __forceinline__ void refine_pair_row(const float* __restrict__ X, const float* __restrict__ C, long long row, int cand,
                                                int d, int lane, int32_t* labels) {
    const float* x = X + row * d;
    const int ca = cand & 0xffff, cb = cand >> 16;
    const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
    const float *c0 = C + (size_t)lo * d, *c1 = C + (size_t)hi * d;
    double s0 = 0.0, s1 = 0.0;
    if ((d & 3) == 0) {
        // all loads of a 1024-column slab are issued before the first is consumed (a scalar loop paid one memory
        // latency per 64 columns); the per-lane partial sums differ from the scalar loop's, the float64 total does
        // not beyond 1e-16 relative (decision semantics)
        for (int j0 = 0; j0 < d; j0 += 1024) {
            float4 xv[4], av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 256 * u + 4 * lane;
                if (j < d) {
                    xv[u] = *(const float4*)(x + j);
                    av[u] = *(const float4*)(c0 + j);
                    bv[u] = *(const float4*)(c1 + j);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 256 * u + 4 * lane;
                if (j < d) {
                    double t;
                    t = (double)xv[u].x - (double)av[u].x; s0 = fma(t, t, s0);
                    t = (double)xv[u].y - (double)av[u].y; s0 = fma(t, t, s0);
                    t = (double)xv[u].z - (double)av[u].z; s0 = fma(t, t, s0);
                    t = (double)xv[u].w - (double)av[u].w; s0 = fma(t, t, s0);
                    t = (double)xv[u].x - (double)bv[u].x; s1 = fma(t, t, s1);
                    t = (double)xv[u].y - (double)bv[u].y; s1 = fma(t, t, s1);
                    t = (double)xv[u].z - (double)bv[u].z; s1 = fma(t, t, s1);
                    t = (double)xv[u].w - (double)bv[u].w; s1 = fma(t, t, s1);
                }
            }
        }
    } else
    for (int j = lane; j < d; j += 64) {
        const double xv = (double)x[j];
        const double d0 = xv - (double)c0[j], d1 = xv - (double)c1[j];
        s0 = fma(d0, d0, s0);
        s1 = fma(d1, d1, s1);
    }
    s0 = wave_sum_f64(s0);
    s1 = wave_sum_f64(s1);
    // same decision sequence as estep_refine_kernel: NaN never wins, ties keep the lower index
    double best = INFINITY;
    int bi = 0;
    if (s0 < best) { best = s0; bi = lo; }
    if (s1 < best) { best = s1; bi = hi; }
    if (lane == 0) labels[row] = bi;
}

// NW = waves of the block: wave w sweeps the centres w, w + NW, ... (eight per pass).  A row is a chain of dependent latencies (the
// row into LDS, then per pass the centre loads, the float64 sweep and eight wave reductions), so the refine LAUNCH uses eight waves per
// row (two passes at K <= 128 instead of four); the streaming kernel's tail runs with that kernel's four.
template <int NW>
__device__ __forceinline__ void refine_full_row(const float* __restrict__ X, const float* __restrict__ C, long long row, int d, int k,
                                                double* xs, double* rv, int* ri, int32_t* labels) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 64 * NW) xs[j] = (double)X[row * d + j];
    __syncthreads();
    double best = INFINITY;
    int bi = 0x7fffffff;
    for (int c0 = wv; c0 < k; c0 += 8 * NW) {
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if ((d & 3) == 0) {
            for (int j0 = 4 * lane; j0 < d; j0 += 768) {
                float4 cv[8][3];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        const int c = c0 + NW * u, j = j0 + 256 * v;
                        if (c < k && j < d) cv[u][v] = *(const float4*)(C + (size_t)c * d + j);
                    }
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int j = j0 + 256 * v;
                    if (j < d) {
                        const double x0 = xs[j], x1 = xs[j + 1], x2 = xs[j + 2], x3 = xs[j + 3];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (c0 + NW * u < k) {
                                double t;
                                t = x0 - (double)cv[u][v].x; a[u] = fma(t, t, a[u]);
                                t = x1 - (double)cv[u][v].y; a[u] = fma(t, t, a[u]);
                                t = x2 - (double)cv[u][v].z; a[u] = fma(t, t, a[u]);
                                t = x3 - (double)cv[u][v].w; a[u] = fma(t, t, a[u]);
                            }
                        }
                    }
                }
            }
        } else {
            for (int j = lane; j < d; j += 64) {
                const double x0 = xs[j];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + NW * u;
                    if (c < k) {
                        const double t = x0 - (double)C[(size_t)c * d + j];
                        a[u] = fma(t, t, a[u]);
                    }
                }
            }
        }
        // the eight wave sums in ONE butterfly (wave_sum8_f64: bit-identical to eight wave_sum_f64, a fifth of the cross-lane traffic:
        // eight separate reductions, serialised behind their guards, were most of a pass)
        double tot[8];
        wave_sum8_f64(a, tot);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + NW * u;
            if (c < k) {
                const double sum = tot[u];
                if (sum < best || (sum == best && c < bi)) { best = sum; bi = c; }      // NaN never wins
            }
        }
    }
    if (lane == 0) { rv[wv] = best; ri[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int w = 0;
        for (int q = 1; q < NW; ++q)
            if (rv[q] < rv[w] || (rv[q] == rv[w] && ri[q] < ri[w])) w = q;
        labels[row] = ri[w] == 0x7fffffff ? 0 : ri[w];
    }
}

// The same re-evaluation for FOUR listed rows at once (the refine launch): a centre value loaded once serves four rows.  At K = 1000 a
// row's sweep reads 2 MB of centres from L2 and the launch was bound by exactly that (279 rows: 72 us; 22,000 rows behind a data-point
// seeding: 1 ms); per (row, centre) the additions and their order are those of refine_full_row, so the sums are the same bits.
template <int NW>
__device__ __forceinline__ void refine_full_rows4(const float* __restrict__ X, const float* __restrict__ C, const int* __restrict__ rows, int nr,
                                                  int d, int k, double* xs, double* rv, int* ri, int32_t* labels) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        const long long row = rows[r < nr ? r : 0];
        for (int j = threadIdx.x; j < d; j += 64 * NW) xs[r * d + j] = (double)X[row * d + j];
    }
    __syncthreads();
    double best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int c0 = wv; c0 < k; c0 += 8 * NW) {
        double a[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[r][u] = 0.0;
        if ((d & 3) == 0) {
            // one 256-column slice at a time: eight centres' float4 in flight per lane (four rows of work per loaded value hide them)
            for (int j = 4 * lane; j < d; j += 256) {
                float4 cv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + NW * u;
                    if (c < k) cv[u] = *(const float4*)(C + (size_t)c * d + j);
                }
                double x[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { x[r][0] = xs[r * d + j]; x[r][1] = xs[r * d + j + 1]; x[r][2] = xs[r * d + j + 2]; x[r][3] = xs[r * d + j + 3]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c0 + NW * u < k) {
                        const double cx = (double)cv[u].x, cy = (double)cv[u].y, cz = (double)cv[u].z, cw = (double)cv[u].w;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double t;
                            t = x[r][0] - cx; a[r][u] = fma(t, t, a[r][u]);
                            t = x[r][1] - cy; a[r][u] = fma(t, t, a[r][u]);
                            t = x[r][2] - cz; a[r][u] = fma(t, t, a[r][u]);
                            t = x[r][3] - cw; a[r][u] = fma(t, t, a[r][u]);
                        }
                    }
                }
            }
        } else {
            for (int j = lane; j < d; j += 64) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + NW * u;
                    if (c < k) {
                        const double cvv = (double)C[(size_t)c * d + j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double t = xs[r * d + j] - cvv;
                            a[r][u] = fma(t, t, a[r][u]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double tot[8];
            wave_sum8_f64(a[r], tot);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + NW * u;
                if (c < k) {
                    const double sum = tot[u];
                    if (sum < best[r] || (sum == best[r] && c < bi[r])) { best[r] = sum; bi[r] = c; }      // NaN never wins
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { rv[wv * 4 + r] = best[r]; ri[wv * 4 + r] = bi[r]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) {
        const int r = threadIdx.x;
        int w = 0;
        for (int q = 1; q < NW; ++q)
            if (rv[q * 4 + r] < rv[w * 4 + r] || (rv[q * 4 + r] == rv[w * 4 + r] && ri[q * 4 + r] < ri[w * 4 + r])) w = q;
        labels[rows[r]] = ri[w * 4 + r] == 0x7fffffff ? 0 : ri[w * 4 + r];
    }
}

template <int NCH>
__global__ void __launch_bounds__(256) estep_stream_kernel(const half_t* __restrict__ xh, const float* __restrict__ xnorm,
                                                           const half_t* __restrict__ ch, const float* __restrict__ cn,
                                                           EHdr* eh, int* flag_list, int* flag_cand, int* full_list,
                                                           long long n, int32_t* __restrict__ labels, int dbg, int cbase,
                                                           int pass, float* __restrict__ tkey, int* __restrict__ tidx,
                                                           const float* __restrict__ cn_all, int kp_all,
                                                           const float* __restrict__ Xf, const float* __restrict__ Cf, int d_f,
                                                           int k_f) {
    // K > 128 runs one launch per 128-centre chunk (`ch` / `cn` point at the chunk, cbase = its first centre): every pass but
    // the last leaves each row's three smallest (key, centre) pairs in tkey / tidx [3][n], every pass but the first merges
    // them in; the last pass (pass & 2) takes the decisions.  pass = 1 first | 2 last.
    constexpr int DP = NCH * 128;
    constexpr int NSLOT = NCH >= 5 ? 3 : 4;
    constexpr int SLOTB = 8192 * NCH;                 // one unit: 32 rows x DP fp16
    constexpr int IPW = 2 * NCH;                      // ring-fill instructions per wave and unit
    constexpr int NG = 2 * NCH;                       // fragment groups (4 k-steps each) per unit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cnl = (float*)(smem + ES_RING);                            // [128] ||c'||^2, dead centres = 3e38
    float* scr = (float*)(smem + ES_RING + 512);                      // [2][4 waves][3][32 points]
    float* res = (float*)(smem + ES_RING + 512 + 3072);               // [3][ES_RMAX]
    int* cnts = (int*)(smem + ES_RING + 512 + 3072 + 3 * ES_RMAX * 4);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;

    const long long U = (n + 31) >> 5;
    const int G = gridDim.x, bid = blockIdx.x;
    if (bid >= U) return;
    const int nu = (int)((U - 1 - bid) / G) + 1;      // units bid, bid + G, ...

    // ring fill: instruction i of wave w covers the 1 KB [1024 (w IPW + i), +1024) of the slot; lane l owns 16-B piece
    // P = 64 (w IPW + i) + l = (row, segment, stored position q) and fetches logical piece q ^ (row & 15) of that segment
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned soff[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int P = 64 * (wave * IPW + i) + lane;
        const int row = P / (16 * NCH), wq = P - row * (16 * NCH);
        soff[i] = (unsigned)(row * (DP * 2) + (wq >> 4) * 256 + (((wq & 15) ^ (row & 15)) << 4));
    }
    // one ring-fill instruction of unit j (a unit past the end issues nothing: the waits below count exactly)
    auto issue_one = [&](int j, int slot, int i) {
        if (j >= nu) return;
        const half_t* ubase = xh + (size_t)(bid + (long long)j * G) * (32 * DP);
        const unsigned lds = sbase + slot * SLOTB + wave * (IPW * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds + i * 1024), "v"(soff[i]), "s"(ubase) : "memory");
    };
    auto issue = [&](int j, int slot) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) issue_one(j, slot, i);
    };
#pragma unroll
    for (int j = 0; j < NSLOT - 1; ++j) issue(j, j);   // the ring starts first: the loads below overlap with its flight

    // resident centre fragments: k-step s of segment c covers columns 128c + 16s + 8hh .. +7 of centre 32w + r
    half8 cf[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int s = 0; s < 8; ++s) cf[c][s] = *(const half8*)(ch + ((((size_t)wave * NCH + c) * 8 + s) * 64 + lane) * 8);   // fragment order: 1 KB per load
    if (tid < 128) cnl[tid] = fminf(cn[tid], 3.0e38f);
    if (tid < 4) cnts[tid] = 0;
    float cv[16];                                   // ||c'||^2 of this lane's 16 centres (MFMA output rows)
#pragma unroll
    for (int i = 0; i < 16; ++i) cv[i] = fminf(cn[32 * wave + (i & 3) + 8 * (i >> 2) + 4 * hh], 3.0e38f);
    float xn_r[ES_RMAX / 256];                      // ||x'|| of the rows this thread decides at the end
#pragma unroll
    for (int j = 0; j < ES_RMAX / 256; ++j) {
        const int p = tid + 256 * j;
        const long long point = (bid + (long long)(p >> 5) * G) * 32 + (p & 31);
        xn_r[j] = (p < nu * 32 && point < n) ? xnorm[point] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(cf[c][s]));     // no compiler-visible load is left in flight
    // values touched once per unit (or once per kernel) are parked in AGPRs: the 192 centre-fragment registers, the fragment
    // ring and the selection state fill the 256 architectural VGPRs, and left alone hipcc parks centre fragments instead
    // (four v_accvgpr_read in front of every MFMA)
#pragma unroll
    for (int j = 0; j < ES_RMAX / 256; ++j) asm volatile("" : "+a"(xn_r[j]));
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(cv[i]));
#pragma unroll
    for (int i = 0; i < IPW; ++i) asm volatile("" : "+a"(soff[i]));

    const int xsw = (hh ^ (r & 15)) << 4;
    const int base_idx = 32 * wave + 4 * hh;
    // cross-wave merge of unit j (threads 0..31, one point each)
    auto merge_unit = [&](int j) {
        const float* sc = scr + (j & 1) * 384;
        float b0 = sc[tid], b1 = sc[32 + tid], b2 = sc[64 + tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            es_insert(b0, b1, b2, sc[(w * 3 + 0) * 32 + tid]);
            es_insert(b0, b1, b2, sc[(w * 3 + 1) * 32 + tid]);
            es_insert(b0, b1, b2, sc[(w * 3 + 2) * 32 + tid]);
        }
        res[j * 32 + tid] = b0;
        res[ES_RMAX + j * 32 + tid] = b1;
        res[2 * ES_RMAX + j * 32 + tid] = b2;
    };

    // The selection ("epilogue") of unit j-1 runs INSIDE unit j's MFMA stream: its 19 steps (16 key inserts, index rewrite +
    // sort, half-wave merge, patch write) are dealt over the NG fragment groups, and the running triple is threaded through
    // each group's wait asm as an operand, so step group g sits between wait g and wait g+1, i.e. in the shadow of group g's
    // MFMAs (a lone wave per SIMD overlaps nothing by itself).  accp holds unit j-1's accumulators.
    f32x16 accp;
#pragma unroll
    for (int i = 0; i < 16; ++i) accp[i] = 0.f;
    float e0 = 3.0e38f, e1 = 3.0e38f, e2 = 3.0e38f;
    auto epi_step = [&](int st, int unit) {
        if (st < 16) {
            if (st == 0) { e0 = 3.0e38f; e1 = 3.0e38f; e2 = 3.0e38f; }
            const float sc = fmaf(-2.0f, accp[st], cv[st]);
            const unsigned key = (__float_as_uint(sc) & 0xffffffe0u) | (unsigned)((st & 3) + 8 * (st >> 2));
            es_insert(e0, e1, e2, __uint_as_float(key));
        } else if (st == 16) {
            // full centre index (7 bits): bits 2, 5, 6 come from the wave / half-wave.  Rewriting low bits can reorder keys
            // that agree above bit 6: the insert network needs an ascending triple
            e0 = __uint_as_float((__float_as_uint(e0) & 0xffffff9bu) | (unsigned)base_idx);
            e1 = __uint_as_float((__float_as_uint(e1) & 0xffffff9bu) | (unsigned)base_idx);
            e2 = __uint_as_float((__float_as_uint(e2) & 0xffffff9bu) | (unsigned)base_idx);
            float n0, n1, n2;
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(n0) : "v"(e0), "v"(e1), "v"(e2));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(n2) : "v"(e0), "v"(e1), "v"(e2));
            n1 = es_med3(e0, e1, e2);
            e0 = n0; e1 = n1; e2 = n2;
        } else if (st == 17) {
            const float o0 = es_swap32(e0), o1 = es_swap32(e1), o2 = es_swap32(e2);
            es_insert(e0, e1, e2, o0);
            es_insert(e0, e1, e2, o1);
            es_insert(e0, e1, e2, o2);
        } else {
            if (hh == 0) {
                float* sc = scr + (unit & 1) * 384 + wave * 96;
                sc[r] = e0;
                sc[32 + r] = e1;
                sc[64 + r] = e2;
            }
        }
    };

    int slot = 0, islot = NSLOT - 1;
    for (int j = 0; j < nu; ++j) {
        {   // my part of unit j has landed: at most min(NSLOT - 2, nu - 1 - j) younger units may still fly
            const int younger = nu - 1 - j;
            if (younger >= NSLOT - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSLOT - 2) * IPW) : "memory");
            else if (NSLOT == 4 && younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                                        // everyone's has; slot islot is free
        asm volatile("" ::: "memory");
        if (dbg & 17) issue(j + NSLOT - 1, islot);     // 16: ring-only ablation; 1: all fills right behind the barrier
        if (j > 1 && tid < 32) merge_unit(j - 2);      // unit j-2's patch was written during unit j-1, before this barrier
        if (!(dbg & 16)) {
            const unsigned sl = sbase + slot * SLOTB + r * (DP * 2);
            // one accumulation chain: back-to-back dependent MFMAs of this shape run at full rate (round 5 measured it: odd k-steps on a
            // second chain + one add per unit changed nothing, 31.3 -> 31.9 us at 131,072 x 512; profiles/r05_estep_chains_ab.txt)
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            half8 fb[3][4];
            // k-step kk = 8c + s reads the 16-B piece (2s + hh) ^ (r & 15) of segment c of row r
#define ES_RD(DST, KK) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(sl + (((((KK) & 7) << 5) ^ xsw))), "n"(((KK) >> 3) * 256))
#define ES_WAIT(N, F) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(e0), "+v"(e1), "+v"(e2))
#pragma unroll
            for (int i = 0; i < 4; ++i) ES_RD(fb[0][i], i);
            if (NG > 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ES_RD(fb[1][i], 4 + i);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // fragment reads run two groups (8 MFMAs) ahead.  The waits carry the fragments as operands, so the MFMAs
                // (builtins: hipcc sees them and handles the MFMA hazards and the accumulator allocation) cannot be
                // scheduled above them
                if (g + 2 < NG) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ES_RD(fb[(g + 2) % 3][i], 4 * (g + 2) + i);
                    ES_WAIT(8, fb[g % 3]);
                } else if (g + 1 < NG) {
                    ES_WAIT(4, fb[g % 3]);
                } else {
                    ES_WAIT(0, fb[g % 3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = 4 * g + i;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf[kk >> 3][kk & 7], fb[g % 3][i], acc, 0, 0, 0);
                }
#pragma unroll
                for (int st = 19 * g / NG; st < 19 * (g + 1) / NG; ++st) epi_step(st, j - 1);     // unit j-1 (a dummy at j = 0)
                if (!(dbg & 1)) issue_one(j + NSLOT - 1, islot, g);      // one ring-fill instruction per group, in the shadow of its MFMAs
            }
            accp = acc;
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        islot = islot + 1 == NSLOT ? 0 : islot + 1;
    }
    if (!(dbg & 16)) {
#pragma unroll
        for (int st = 0; st < 19; ++st) epi_step(st, nu - 1);       // the last unit's selection has no stream to hide in
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the ring is dead, every patch is written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid < 32) {
        if (nu > 1) merge_unit(nu - 2);
        merge_unit(nu - 1);
    }
    __syncthreads();

    // decisions for all rows of the block
    float cm2 = 0.f;
    for (int c = lane; c < kp_all; c += 64) {
        const float v = cn_all[c];
        if (v < 3.0e38f) cm2 = fmaxf(cm2, v);
    }
    cm2 = wave_max_f32(cm2);
    const float cmax = sqrtf(cm2) * 1.0000002f;
    const float sq = sqrtf((float)DP);
    // |s~ - s| <= A*||x'|| + B (see estep_mfma_kernel) + the key's low 7 bits: 2^-16 * (||c'||^2 + 2 ||x'|| ||c'||)
    const float A = 1.5f * (2.02f * (9.765625e-4f + DP * 5.9604645e-8f) * cmax + 4.8e-7f * cmax + 6.0e-8f * sq + 3.06e-5f * cmax);
    const float B = 1.5f * (6.0e-8f * sq * cmax + 2.4e-7f * cmax * cmax + 1.53e-5f * cmax * cmax);
    int* l_flag = (int*)smem;
    int* l_cand = l_flag + ES_RMAX;
    int* l_full = l_cand + ES_RMAX;
#pragma unroll
    for (int j = 0; j < ES_RMAX / 256; ++j) {
        const int p = tid + 256 * j;
        const long long point = (bid + (long long)(p >> 5) * G) * 32 + (p & 31);
        if (p >= nu * 32 || point >= n) continue;
        float m0 = res[p], m1 = res[ES_RMAX + p], m2 = res[2 * ES_RMAX + p];
        int j0 = cbase + (int)(__float_as_uint(m0) & 127u), j1 = cbase + (int)(__float_as_uint(m1) & 127u),
            j2 = cbase + (int)(__float_as_uint(m2) & 127u);
        if (!(pass & 1)) {                 // merge the triple of the earlier chunks (ascending; ties keep the earlier chunk)
            float pv[3];
            int pi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { pv[q] = tkey[(size_t)q * n + point]; pi[q] = tidx[(size_t)q * n + point]; }
            float a0 = pv[0], a1 = pv[1], a2 = pv[2];
            int i0 = pi[0], i1 = pi[1], i2 = pi[2];
            const float nv[3] = {m0, m1, m2};
            const int ni[3] = {j0, j1, j2};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (nv[q] < a2) {
                    a2 = nv[q]; i2 = ni[q];
                    if (a2 < a1) { const float tv = a1; a1 = a2; a2 = tv; const int ti = i1; i1 = i2; i2 = ti; }
                    if (a1 < a0) { const float tv = a0; a0 = a1; a1 = tv; const int ti = i0; i0 = i1; i1 = ti; }
                }
            }
            m0 = a0; m1 = a1; m2 = a2; j0 = i0; j1 = i1; j2 = i2;
        }
        if (!(pass & 2)) {
            tkey[point] = m0; tkey[(size_t)n + point] = m1; tkey[2 * (size_t)n + point] = m2;
            tidx[point] = j0; tidx[(size_t)n + point] = j1; tidx[2 * (size_t)n + point] = j2;
            continue;
        }
        labels[point] = j0;
        const float E = A * xn_r[j] + B;
        if (!(m1 - m0 > 2.0f * E)) {       // also catches NaN
            if (m2 - m0 > 2.0f * E) {
                const int pos = atomicAdd(&cnts[0], 1);
                l_flag[pos] = (int)point;
                l_cand[pos] = j0 | (j1 << 16);
            } else {
                l_full[atomicAdd(&cnts[1], 1)] = (int)point;
            }
        }
    }
    __syncthreads();
    if (Xf) {
        // refine in the tail: every block re-evaluates the rows IT flagged (exact float64 difference form) - no refine launch, no
        // global lists; converged centres flag nothing and the kernel simply ends.  The lists sit at the start of the (drained)
        // ring, the row buffer of the all-centres form behind them.
        double* xs = (double*)(smem + 3 * ES_RMAX * 4);
        double* rv = xs + 1024;
        int* ri = (int*)(rv + 4);
        const int n0 = cnts[0], n1 = cnts[1];
        for (int i = tid >> 6; i < n0; i += 4) refine_pair_row(Xf, Cf, l_flag[i], l_cand[i], d_f, tid & 63, labels);
        for (int i = 0; i < n1; ++i) refine_full_row<4>(Xf, Cf, l_full[i], d_f, k_f, xs, rv, ri, labels);
        if (tid == 0) {
            if (n0) atomicAdd(&eh->flag_cnt, n0);
            if (n1) atomicAdd(&eh->full_cnt, n1);
        }
        return;
    }
    if (tid == 0) {
        cnts[2] = cnts[0] ? atomicAdd(&eh->flag_cnt, cnts[0]) : 0;
        cnts[3] = cnts[1] ? atomicAdd(&eh->full_cnt, cnts[1]) : 0;
    }
    __syncthreads();
    for (int i = tid; i < cnts[0]; i += 256) {
        flag_list[cnts[2] + i] = l_flag[i];
        flag_cand[cnts[2] + i] = l_cand[i];
    }
    for (int i = tid; i < cnts[1]; i += 256) full_list[cnts[3] + i] = l_full[i];
}

// ------------------------------------------------------------------------------------------------
// Single-pass filter for 128 < K <= 2048 at Dp = 512 (BASELINE C4: K = 1000, CLIP features; round 3).  The streaming kernel above
// keeps 128 centres in registers and re-reads X once per 128 centres (8 passes over X at K = 1000: 350 us for 164 MB of
// algorithmic bytes, 0.19 of the MFMA peak - and at K = 1000 the E-step is matrix-bound).  Here the roles are swapped,
// in the structure of sim_topk_rb8_kernel: a block keeps 256 ROWS of x' in registers (eight waves, two per SIMD, 32 rows each as B
// fragments in 128 AGPRs) and the centres stream past them ONCE, in units of 32 centres x 512 columns through the 4-slot LDS-DMA ring
// (one 1-KB row per instruction, 16-B chunk c of row r at chunk c ^ (r & 15)); the whole centre matrix is 1 MB and stays in L2.
//   * score: the 33rd MFMA of a unit multiplies the centres' extension column (||c'||^2 / 2 as an fp16 pair hi + lo, 16 bytes per
//     centre in LDS behind the ring) with a constant -1 fragment, so an accumulator ends as x'.c' - ||c'||^2 / 2 (maximal where the
//     distance is minimal) and no per-unit norm vector has to reach the lanes;
//   * selection: every value becomes a key (low 11 mantissa bits = centre index) and goes through a max / med3 / min network that
//     keeps the lane's three LARGEST keys over all units - 6 instructions per value, dealt over the 32 MFMA steps of the next
//     unit, no branches, no lists; lanes r / r + 32 merge at the end;
//   * decisions as in estep_stream_kernel: label = best key's centre; a row whose first two scores are within 2 E goes to the pair
//     list (third outside 2 E) or to the all-centres list; estep_refine_both_kernel re-evaluates them in float64.
// Error bound: that of the streaming kernel with the key term for 11 index bits (2^-12 of |x'.c' - ||c'||^2 / 2|) and 2^-22 ||c'||^2
// for the hi + lo split.
#define ERB_LDS (131072 + 32768)
__device__ __forceinline__ void erb_insert(float& b0, float& b1, float& b2, float k) {       // descending triple (b0 >= b1 >= b2)
    const float t = es_min(b1, k);
    b2 = es_max(b2, t);
    b1 = es_med3(b0, b1, k);
    b0 = es_max(b0, k);
}
__global__ void __launch_bounds__(256) estep_ext_kernel(const float* __restrict__ cn, int kp, half_t* __restrict__ ext) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= kp) return;
    const float v = cn[c];
    // ||c'||^2 / 8, multiplied by -4 in the MFMA: |x'| < 16 gives ||c'||^2 / 2 <= 65,536, just past the fp16 range
    float hv = 0.125f * v;
    if (!(v < 3.0e38f)) hv = 60000.f;                            // dead centre (padding, NaN): never the best of a row
    const half_t hi = (half_t)hv;
    const half_t lo = (half_t)(hv - (float)hi);
    half8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (half_t)0.f;
    o[0] = hi;
    o[1] = lo;
    *(half8*)(ext + (size_t)c * 8) = o;
}
// decision for one row from its three largest keys (shared by the kernel and by the merge of split row blocks)
__device__ __forceinline__ void erb_decide(long long row, float b0, float b1, float b2, float cm2, const float* __restrict__ xnorm, EHdr* eh,
                                           int* flag_list, int* flag_cand, int* full_list, int32_t* __restrict__ labels) {
    constexpr int D = 512;
    // scores s = -2 * key value, ascending: m0 <= m1 <= m2
    const float m0 = -2.f * b0, m1 = -2.f * b1, m2 = -2.f * b2;
    const int j0 = (int)(__float_as_uint(b0) & 2047u), j1 = (int)(__float_as_uint(b1) & 2047u);
    const float cmax = sqrtf(cm2) * 1.0000002f;
    const float sq = 22.627417f;                                 // sqrt(512)
    const float A = 1.5f * (2.02f * (9.765625e-4f + D * 5.9604645e-8f) * cmax + 4.8e-7f * cmax + 6.0e-8f * sq + 4.9e-4f * cmax);
    const float B = 1.5f * (6.0e-8f * sq * cmax + 4.8e-7f * cmax * cmax + 2.45e-4f * cmax * cmax);
    labels[row] = j0;
    const float E = A * xnorm[row] + B;
    if (!(m1 - m0 > 2.0f * E)) {                                 // also catches NaN
        if (m2 - m0 > 2.0f * E) {
            const int pos = atomicAdd(&eh->flag_cnt, 1);
            flag_list[pos] = (int)row;
            flag_cand[pos] = j0 | (j1 << 16);
        } else {
            full_list[atomicAdd(&eh->full_cnt, 1)] = (int)row;
        }
    }
}
// Grid: the first `nfull` blocks take one 256-row block each against all centres.  The row blocks of the last, partial round of
// the chip (626 row blocks on 256 CUs at C4: 114 left for a third round) are SPLIT: `nsplit` blocks each sweep 1 / nsplit of the
// centres, leave their rows' three best keys in `tkeys` [nsplit][3][tail rows], and estep_rb_merge_kernel merges and decides - the
// last round then takes 1 / nsplit of a full one.
__global__ void __launch_bounds__(512) estep_rb_kernel(const half_t* __restrict__ xh, const float* __restrict__ xnorm,
                                                       const half_t* __restrict__ ch_all, const half_t* __restrict__ ext_all,
                                                       const float* __restrict__ cn, EHdr* eh, int* flag_list, int* flag_cand,
                                                       int* full_list, long long n, int kp_all, int32_t* __restrict__ labels,
                                                       int nfull, int nsplit, float* __restrict__ tkeys) {
    constexpr int D = 512, UB = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // this block's rows and its share of the centres
    const bool split = (int)blockIdx.x >= nfull;
    const int part = split ? ((int)blockIdx.x - nfull) % nsplit : 0;
    const long long rblock = split ? nfull + ((int)blockIdx.x - nfull) / nsplit : blockIdx.x;
    const int nunits = split ? (kp_all >> 5) / nsplit : kp_all >> 5, u0 = part * nunits, kp = nunits * 32;
    const half_t* ch = ch_all + (size_t)u0 * 32 * D;
    const half_t* ext = ext_all + (size_t)u0 * 32 * 8;

    // the extension column of every centre: 16 B per centre behind the ring
    for (int c = tid; c < kp; c += 512) *(half8*)(smem + 4 * UB + c * 16) = *(const half8*)(ext + (size_t)c * 8);

    half8 bf[32];
    const long long row = rblock * 256 + wave * 32 + r;
    {
        const half_t* xr = xh + (row < n ? row : n - 1) * D + 8 * hh;
#pragma unroll
        for (int b = 0; b < 4; ++b) {                              // eight fragments at a time (see sim_topk_rb8_kernel)
#pragma unroll
            for (int s = 8 * b; s < 8 * b + 8; ++s) bf[s] = *(const half8*)(xr + 16 * s);
#pragma unroll
            for (int s = 8 * b; s < 8 * b + 8; ++s) asm volatile("" : "+a"(bf[s]) : : "memory");
        }
    }
    half8 bx;                                                    // B fragment of the extension step: -4 against (hi, lo) = ||c'||^2 / 8, lanes r only
#pragma unroll
    for (int q = 0; q < 8; ++q) bx[q] = (half_t)((hh == 0 && q < 2) ? -4.f : 0.f);

    const unsigned bsw = (unsigned)((lane ^ ((4 * wave) & 12)) << 4);
    const half_t* fbase = ch;
    unsigned fm0 = 0;
    auto fill_unit = [&](int unit) {
        fbase = ch + (size_t)unit * 32 * D;
        fm0 = sbase + (unit & 3) * UB + 4 * wave * 1024;
    };
    auto fill = [&](int p) {
        const unsigned off = (bsw ^ (unsigned)(p << 4)) + (unsigned)(4 * wave + p) * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     ::"s"(fm0 + p * 1024), "v"(off), "s"(fbase) : "memory");
    };
    unsigned fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = sbase + (unsigned)(r * 1024 + ((32 * j) ^ (16 * (hh ^ (r & 15)))));
    unsigned fxa = sbase + 4 * UB + (unsigned)(r * 16);          // extension fragment of unit 0, row r (both half-waves read it)

#define ERB_RD(DST, J, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(fa[J]), "n"(IMM))
#define ERB_RDX(DST) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(fxa))
#define ERB_WAIT(N, FR) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(FR))
#define ERB_MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(B))
#define ERB_MFMAV(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define ERB_MFMA0(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "a"(B))

    float b0 = -INFINITY, b1 = -INFINITY, b2 = -INFINITY;        // the lane's three largest keys
    f32x16 acc[2];
    half8 fr[4], fx;
    using yes = std::true_type;
    using no = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto body = [&](auto has_prev, auto parity, int u) {
        constexpr int P = decltype(parity)::value;
        constexpr bool EPI = decltype(has_prev)::value;
        if (u + 2 < nunits) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = u + 1 < nunits;
        const bool fills = u + 3 < nunits;
        if (fills) fill_unit(u + 3);
        const unsigned ub = (unsigned)((u0 + u - 1) * 32 + 4 * hh);   // centre index of value i of unit u - 1: ub + (i & 3) + 8 (i >> 2)
        static_for<0, 32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s == 29) {
                const unsigned delta = ((u + 1) & 3) ? (unsigned)UB : (unsigned)(-3 * UB);      // wave-uniform
#pragma unroll
                for (int j = 0; j < 8; ++j) fa[j] += delta;
            }
            if constexpr (s < 29) ERB_RD(fr[(s + 3) & 3], (s + 3) & 7, ((s + 3) >> 3) * 256);
            else if (more) ERB_RD(fr[(s + 3) & 3], (s + 3 - 32) & 7, 0);
            // the extension fragment of THIS unit is fetched at step 8 (one more read in flight for three waits) and used at step 20
            if constexpr (s == 8) ERB_RDX(fx);
            if constexpr (s >= 8 && s <= 10) ERB_WAIT(3, fr[(s + 1) & 3]);
            else if (s < 29 || more) ERB_WAIT(2, fr[(s + 1) & 3]);
            else if (s == 29) ERB_WAIT(1, fr[(s + 1) & 3]);
            else if (s == 30) ERB_WAIT(0, fr[(s + 1) & 3]);
            if (s == 0) ERB_MFMA0(acc[P], fr[s & 3], bf[s]);
            else ERB_MFMA(acc[P], fr[s & 3], bf[s]);
            if constexpr (s == 20) {
                asm volatile("" : "+v"(fx));
                ERB_MFMAV(acc[P], fx, bx);
                fxa += 512;                                      // next unit's 32 extension entries
            }
            if ((s & 7) == 7 && fills) fill(s >> 3);
            if constexpr (EPI && s >= 2 && s < 18) {
                constexpr int i = s - 2;
                const unsigned idx = ub + (unsigned)((i & 3) + 8 * (i >> 2));
                const float k = __uint_as_float((__float_as_uint(acc[1 - P][i]) & 0xfffff800u) | idx);
                erb_insert(b0, b1, b2, k);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
#pragma unroll 1
    for (int pre = 0; pre < 3; ++pre)
        if (pre < nunits) {
            fill_unit(pre);
#pragma unroll
            for (int p = 0; p < 4; ++p) fill(p);
        }
    if (nunits > 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (nunits > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // unit 0 and the extension table are in LDS
    asm volatile("" ::: "memory");
    ERB_RD(fr[0], 0, 0);
    ERB_RD(fr[1], 1, 0);
    ERB_RD(fr[2], 2, 0);
    ERB_WAIT(2, fr[0]);

    body(no{}, P0{}, 0);
    int u = 1;
    for (; u + 1 < nunits; u += 2) {
        body(yes{}, P1{}, u);
        body(yes{}, P0{}, u + 1);
    }
    const bool odd_tail = u < nunits;
    if (odd_tail) body(yes{}, P1{}, u);
    {   // keys of the last unit
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        const unsigned ub = (unsigned)((u0 + nunits - 1) * 32 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = odd_tail ? acc[1][i] : acc[0][i];
            erb_insert(b0, b1, b2, __uint_as_float((__float_as_uint(a) & 0xfffff800u) | (ub + (unsigned)((i & 3) + 8 * (i >> 2)))));
        }
    }
    {   // the other half-wave's three
        const float o0 = es_swap32(b0), o1 = es_swap32(b1), o2 = es_swap32(b2);
        erb_insert(b0, b1, b2, o0);
        erb_insert(b0, b1, b2, o1);
        erb_insert(b0, b1, b2, o2);
    }
    float cm2 = 0.f;
    for (int c = lane; c < kp_all; c += 64) {
        const float v = cn[c];
        if (v < 3.0e38f) cm2 = fmaxf(cm2, v);
    }
    cm2 = wave_max_f32(cm2);
    if (hh != 0 || row >= n) return;
    if (split) {                                                 // this part's three best of the row, for estep_rb_merge_kernel
        const long long trows = n - (long long)nfull * 256, tr = row - (long long)nfull * 256;
        tkeys[((size_t)part * 3 + 0) * trows + tr] = b0;
        tkeys[((size_t)part * 3 + 1) * trows + tr] = b1;
        tkeys[((size_t)part * 3 + 2) * trows + tr] = b2;
        return;
    }
    erb_decide(row, b0, b1, b2, cm2, xnorm, eh, flag_list, flag_cand, full_list, labels);
#undef ERB_RD
#undef ERB_RDX
#undef ERB_WAIT
#undef ERB_MFMA
#undef ERB_MFMAV
#undef ERB_MFMA0
}

// The same sweep for the centres of SEVERAL restarts of one fit at once (scd_kmeans_lloyd_run_multi: the restarts' Lloyd loops in
// lock-step; faster_mix_k_means_pytorch.py:244-275 runs them one after the other): the rows of a block stay in registers while the centre
// matrices of the restarts stream past them back to back, `upseg` units (kp / 32) per restart.  A restart is a SEGMENT of the unit
// stream: its keys carry the centre's index inside the segment, at the segment's end the lane's three best keys are decided exactly as
// estep_rb_kernel decides (erb_decide's arithmetic) - into one packed register per segment, the selection state is reset - and the label
// stores / list atomics of all segments happen after the stream has drained (a store inside the loop would sit among the ring fills in
// the vmcnt queue, and loads and stores do not retire in order with each other: the counted waits would no longer mean "unit u + 1 has
// landed").  The restarts' E-step workspaces (EHdr | norms | fp16 centres | ... | lists, the layout of scd_kmeans_estep) are equally
// strided (`ws_stride` bytes from `ws0`), so are their label slots; `segmap` packs the workspace slot of the a-th RUNNING restart in
// 4-bit fields.  Grid: row blocks x nparts; part p takes the running restarts [p A / nparts, (p + 1) A / nparts) - whole segments, so
// the parts need no merge.  At most 8 segments and 1,920 centres per part.
struct RbmArgs {
    char* ws0; size_t ws_stride;                   // restart slot s: ws0 + s * ws_stride
    size_t cn_off, ch_off, flags_off, fcand_off, fulls_off;
    int32_t* lab0; size_t lab_stride;              // labels of slot s: lab0 + s * lab_stride (elements)
    unsigned long long segmap;
    int n_active, nparts, kp;
};
__global__ void __launch_bounds__(512) estep_rbm_kernel(const half_t* __restrict__ xh, const float* __restrict__ xnorm, const RbmArgs a,
                                                        long long n) {
    constexpr int D = 512, UB = 32768, MAXSEG = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int part = (int)blockIdx.x % a.nparts;
    const long long rblock = blockIdx.x / a.nparts;
    const int a0 = part * a.n_active / a.nparts, a1 = (part + 1) * a.n_active / a.nparts;
    const int nseg = a1 - a0, upseg = a.kp >> 5, nunits = nseg * upseg, kp = a.kp;
    if (nseg <= 0) return;
    auto slot_of = [&](int sg) { return (int)((a.segmap >> (4 * (a0 + sg))) & 15ull); };
    float* cm2s = (float*)(smem + 4 * UB + 1920 * 16);            // [MAXSEG] max ||c'||^2 of the part's segments

    // the extension column of every centre of the part (||c'||^2 / 8 as an fp16 pair hi + lo, see estep_ext_kernel), 16 B per centre
    // behind the ring - formed here from the restarts' norms (no launch per restart) - and the segments' largest norms
    for (int c = tid; c < nseg * kp; c += 512) {
        const int sg = c / kp, cc = c - sg * kp;
        const float v = ((const float*)(a.ws0 + (size_t)slot_of(sg) * a.ws_stride + a.cn_off))[cc];
        float hv = v * 0.125f;
        if (!(v < 3.0e38f)) hv = 60000.f;                         // dead centre (padding, NaN): never the best of a row
        const half_t hi = (half_t)hv;
        const half_t lo = (half_t)(hv - (float)hi);
        half8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (half_t)0.f;
        o[0] = hi;
        o[1] = lo;
        *(half8*)(smem + 4 * UB + c * 16) = o;
    }
    for (int sg = wave; sg < nseg; sg += 8) {
        const float* cn = (const float*)(a.ws0 + (size_t)slot_of(sg) * a.ws_stride + a.cn_off);
        float cm2 = 0.f;
        for (int c = lane; c < kp; c += 64) {
            const float v = cn[c];
            if (v < 3.0e38f) cm2 = fmaxf(cm2, v);
        }
        cm2 = wave_max_f32(cm2);
        if (lane == 0) cm2s[sg] = cm2;
    }

    half8 bf[32];
    const long long row = rblock * 256 + wave * 32 + r;
    const float xn = xnorm[row < n ? row : n - 1];
    {
        const half_t* xr = xh + (row < n ? row : n - 1) * D + 8 * hh;
#pragma unroll
        for (int b = 0; b < 4; ++b) {                              // eight fragments at a time (see sim_topk_rb8_kernel)
#pragma unroll
            for (int s = 8 * b; s < 8 * b + 8; ++s) bf[s] = *(const half8*)(xr + 16 * s);
#pragma unroll
            for (int s = 8 * b; s < 8 * b + 8; ++s) asm volatile("" : "+a"(bf[s]) : : "memory");
        }
    }
    half8 bx;                                                    // B fragment of the extension step: -4 against (hi, lo) = ||c'||^2 / 8, lanes r only
#pragma unroll
    for (int q = 0; q < 8; ++q) bx[q] = (half_t)((hh == 0 && q < 2) ? -4.f : 0.f);

    const unsigned bsw = (unsigned)((lane ^ ((4 * wave) & 12)) << 4);
    const half_t* fbase = nullptr;
    unsigned fm0 = 0;
    int f_unit = 0, f_seg = 0, f_in = 0;                         // the unit whose fills are being issued: number, segment, unit inside it
    auto fill_next = [&]() {                                     // units are filled in order: 0, 1, 2, ...
        fbase = (const half_t*)(a.ws0 + (size_t)slot_of(f_seg) * a.ws_stride + a.ch_off) + (size_t)f_in * 32 * D;
        fm0 = sbase + (f_unit & 3) * UB + 4 * wave * 1024;
        ++f_unit;
        if (++f_in == upseg) { f_in = 0; ++f_seg; }
    };
    auto fill = [&](int p) {
        const unsigned off = (bsw ^ (unsigned)(p << 4)) + (unsigned)(4 * wave + p) * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     ::"s"(fm0 + p * 1024), "v"(off), "s"(fbase) : "memory");
    };
    unsigned fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = sbase + (unsigned)(r * 1024 + ((32 * j) ^ (16 * (hh ^ (r & 15)))));
    unsigned fxa = sbase + 4 * UB + (unsigned)(r * 16);          // extension fragment of unit 0, row r (both half-waves read it)

#define ERB_RD(DST, J, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(fa[J]), "n"(IMM))
#define ERB_RDX(DST) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(fxa))
#define ERB_WAIT(N, FR) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(FR))
#define ERB_MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(B))
#define ERB_MFMAV(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define ERB_MFMA0(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "a"(B))

    float b0 = -INFINITY, b1 = -INFINITY, b2 = -INFINITY;        // the lane's three largest keys of the current segment
    unsigned resv[MAXSEG];                                       // per segment: label | second centre << 8 | state << 16 (1 pair list, 2 all-centres list)
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) resv[i] = 0u;
    // decision of one finished segment from the half-wave's three best keys: erb_decide's arithmetic, the outcome packed
    auto decide = [&](int sg) {
        const float o0 = es_swap32(b0), o1 = es_swap32(b1), o2 = es_swap32(b2);
        erb_insert(b0, b1, b2, o0);
        erb_insert(b0, b1, b2, o1);
        erb_insert(b0, b1, b2, o2);
        const float m0 = -2.f * b0, m1 = -2.f * b1, m2 = -2.f * b2;
        // (a segment holds at most 256 centres: EIGHT index bits in a key, i.e. a relative perturbation below 2^-15 of the score where
        // estep_rb_kernel's eleven bits cost 2^-12 - the key terms of the bound shrink by 8 and with them the rows sent to the refine)
        const unsigned j0 = __float_as_uint(b0) & 255u, j1 = __float_as_uint(b1) & 255u;
        const float cmax = sqrtf(cm2s[sg]) * 1.0000002f;
        const float sq = 22.627417f;                             // sqrt(512)
        const float A = 1.5f * (2.02f * (9.765625e-4f + D * 5.9604645e-8f) * cmax + 4.8e-7f * cmax + 6.0e-8f * sq + 6.2e-5f * cmax);
        const float B = 1.5f * (6.0e-8f * sq * cmax + 4.8e-7f * cmax * cmax + 3.1e-5f * cmax * cmax);
        const float E = A * xn + B;
        unsigned state = 0u;
        if (!(m1 - m0 > 2.0f * E)) state = (m2 - m0 > 2.0f * E) ? 1u : 2u;     // also catches NaN
        const unsigned packed = j0 | (j1 << 8) | (state << 16);
#pragma unroll
        for (int i = 0; i < MAXSEG; ++i) resv[i] = (i == sg) ? packed : resv[i];
        b0 = -INFINITY; b1 = -INFINITY; b2 = -INFINITY;
    };
    f32x16 acc[2];
    half8 fr[4], fx;
    using yes = std::true_type;
    using no = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int e_in = 0, e_seg = 0;                                      // unit u - 1 (the one whose keys are inserted): unit inside its segment, segment
    auto body = [&](auto has_prev, auto parity, int u) {
        constexpr int P = decltype(parity)::value;
        constexpr bool EPI = decltype(has_prev)::value;
        if (u + 2 < nunits) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = u + 1 < nunits;
        const bool fills = u + 3 < nunits;
        if (fills) fill_next();
        const unsigned ub = (unsigned)(e_in * 32 + 4 * hh);      // centre index (inside its segment) of value i of unit u - 1: ub + (i & 3) + 8 (i >> 2)
        static_for<0, 32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s == 29) {
                const unsigned delta = ((u + 1) & 3) ? (unsigned)UB : (unsigned)(-3 * UB);      // wave-uniform
#pragma unroll
                for (int j = 0; j < 8; ++j) fa[j] += delta;
            }
            if constexpr (s < 29) ERB_RD(fr[(s + 3) & 3], (s + 3) & 7, ((s + 3) >> 3) * 256);
            else if (more) ERB_RD(fr[(s + 3) & 3], (s + 3 - 32) & 7, 0);
            if constexpr (s == 8) ERB_RDX(fx);
            if constexpr (s >= 8 && s <= 10) ERB_WAIT(3, fr[(s + 1) & 3]);
            else if (s < 29 || more) ERB_WAIT(2, fr[(s + 1) & 3]);
            else if (s == 29) ERB_WAIT(1, fr[(s + 1) & 3]);
            else if (s == 30) ERB_WAIT(0, fr[(s + 1) & 3]);
            if (s == 0) ERB_MFMA0(acc[P], fr[s & 3], bf[s]);
            else ERB_MFMA(acc[P], fr[s & 3], bf[s]);
            if constexpr (s == 20) {
                asm volatile("" : "+v"(fx));
                ERB_MFMAV(acc[P], fx, bx);
                fxa += 512;                                      // next unit's 32 extension entries
            }
            if ((s & 7) == 7 && fills) fill(s >> 3);
            if constexpr (EPI && s >= 2 && s < 18) {
                constexpr int i = s - 2;
                const unsigned idx = ub + (unsigned)((i & 3) + 8 * (i >> 2));
                const float k = __uint_as_float((__float_as_uint(acc[1 - P][i]) & 0xffffff00u) | idx);
                erb_insert(b0, b1, b2, k);
            }
            if constexpr (EPI && s == 22) {                      // unit u - 1 closed its segment: decide it (wave-uniform branch)
                if (e_in + 1 == upseg) decide(e_seg);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (EPI) { if (++e_in == upseg) { e_in = 0; ++e_seg; } }
    };
#pragma unroll 1
    for (int pre = 0; pre < 3; ++pre)
        if (pre < nunits) {
            fill_next();
#pragma unroll
            for (int p = 0; p < 4; ++p) fill(p);
        }
    if (nunits > 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (nunits > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // unit 0, the extension table and the segments' norms are in LDS
    asm volatile("" ::: "memory");
    ERB_RD(fr[0], 0, 0);
    ERB_RD(fr[1], 1, 0);
    ERB_RD(fr[2], 2, 0);
    ERB_WAIT(2, fr[0]);

    body(no{}, P0{}, 0);
    int u = 1;
    for (; u + 1 < nunits; u += 2) {
        body(yes{}, P1{}, u);
        body(yes{}, P0{}, u + 1);
    }
    const bool odd_tail = u < nunits;
    if (odd_tail) body(yes{}, P1{}, u);
    {   // keys of the last unit (the last of the last segment), then that segment's decision
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        const unsigned ub = (unsigned)((upseg - 1) * 32 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float av = odd_tail ? acc[1][i] : acc[0][i];
            erb_insert(b0, b1, b2, __uint_as_float((__float_as_uint(av) & 0xffffff00u) | (ub + (unsigned)((i & 3) + 8 * (i >> 2)))));
        }
        decide(nseg - 1);
    }
    if (hh != 0 || row >= n) return;
    // the stream has drained: labels and list entries of every segment
#pragma unroll
    for (int sg = 0; sg < MAXSEG; ++sg) {
        if (sg >= nseg) break;
        char* wsb = a.ws0 + (size_t)slot_of(sg) * a.ws_stride;
        const unsigned pk = resv[sg];
        const int j0 = (int)(pk & 255u), j1 = (int)((pk >> 8) & 255u), state = (int)(pk >> 16);
        (a.lab0 + (size_t)slot_of(sg) * a.lab_stride)[row] = j0;
        EHdr* eh = (EHdr*)wsb;
        if (state == 1) {
            const int pos = atomicAdd(&eh->flag_cnt, 1);
            ((int*)(wsb + a.flags_off))[pos] = (int)row;
            ((int*)(wsb + a.fcand_off))[pos] = j0 | (j1 << 16);
        } else if (state == 2) {
            ((int*)(wsb + a.fulls_off))[atomicAdd(&eh->full_cnt, 1)] = (int)row;
        }
    }
#undef ERB_RD
#undef ERB_RDX
#undef ERB_WAIT
#undef ERB_MFMA
#undef ERB_MFMAV
#undef ERB_MFMA0
}

__global__ void __launch_bounds__(256) estep_rb_merge_kernel(const float* __restrict__ tkeys, int nsplit, long long row0, long long n,
                                                             const float* __restrict__ xnorm, const float* __restrict__ cn, int kp,
                                                             EHdr* eh, int* flag_list, int* flag_cand, int* full_list,
                                                             int32_t* __restrict__ labels) {
    float cm2 = 0.f;
    for (int c = threadIdx.x & 63; c < kp; c += 64) {
        const float v = cn[c];
        if (v < 3.0e38f) cm2 = fmaxf(cm2, v);
    }
    cm2 = wave_max_f32(cm2);
    const long long trows = n - row0, tr = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tr >= trows) return;
    float b0 = -INFINITY, b1 = -INFINITY, b2 = -INFINITY;
    for (int p = 0; p < nsplit; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) erb_insert(b0, b1, b2, tkeys[((size_t)p * 3 + q) * trows + tr]);
    erb_decide(row0 + tr, b0, b1, b2, cm2, xnorm, eh, flag_list, flag_cand, full_list, labels);
}

// exact re-evaluation of flagged rows: one wave per row, float64 difference form over all K centres
__global__ void __launch_bounds__(64) estep_refine_kernel(const float* __restrict__ X, const float* __restrict__ C,
                                                          const EHdr* eh, const int* flag_list, const int* flag_cand, int d,
                                                          int k, int32_t* labels) {
    const int lane = threadIdx.x;
    const int cnt = eh->flag_cnt;
    for (int f = blockIdx.x; f < cnt; f += gridDim.x) {
        const long long row = flag_list[f];
        const int cand = flag_cand[f];
        const float* x = X + row * d;
        double best = INFINITY;
        int bi = 0;
        const int nc = cand < 0 ? k : 2;
        for (int t = 0; t < nc; ++t) {
            int c = t;
            if (cand >= 0) {                       // two candidates, visited in ascending centre order
                const int ca = cand & 0xffff, cb = cand >> 16;
                const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
                c = t == 0 ? lo : hi;
            }
            const float* cc = C + (size_t)c * d;
            double s = 0.0;
            for (int j = lane; j < d; j += 64) {
                const double df = (double)x[j] - (double)cc[j];
                s = fma(df, df, s);
            }
            s = wave_sum_f64(s);
            if (s < best) {     // NaN never wins; ties keep the lowest index
                best = s;
                bi = c;
            }
        }
        if (lane == 0) labels[row] = bi;
    }
}

// full-mode refine: one block (2 waves) per row; lane t owns centre t of a 128-centre chunk and walks the TRANSPOSED centre
// matrix ct[j][t] (coalesced across lanes) against the row held as doubles in LDS.  Exact float64 difference form.
__global__ void __launch_bounds__(128) estep_refine_full_kernel(const float* __restrict__ X, const float* __restrict__ ct,
                                                                const EHdr* eh, const int* full_list, int d, int k, int kp,
                                                                int32_t* labels) {
    extern __shared__ double xs[];            // d doubles, then 4 doubles of reduction scratch
    __shared__ double rv[2];
    __shared__ int ri[2];
    const int cnt = eh->full_cnt;
    for (int f = blockIdx.x; f < cnt; f += gridDim.x) {
        const long long row = full_list[f];
        __syncthreads();
        for (int j = threadIdx.x; j < d; j += 128) xs[j] = (double)X[row * d + j];
        __syncthreads();
        double best = INFINITY;
        int bi = 0x7fffffff;
        for (int c0 = 0; c0 < k; c0 += 128) {
            const int c = c0 + threadIdx.x;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const float* col = ct + c;
            int j = 0;
            for (; j + 4 <= d; j += 4) {
                const double d0 = xs[j] - (double)col[(size_t)j * kp];
                const double d1 = xs[j + 1] - (double)col[(size_t)(j + 1) * kp];
                const double d2 = xs[j + 2] - (double)col[(size_t)(j + 2) * kp];
                const double d3 = xs[j + 3] - (double)col[(size_t)(j + 3) * kp];
                a0 = fma(d0, d0, a0); a1 = fma(d1, d1, a1); a2 = fma(d2, d2, a2); a3 = fma(d3, d3, a3);
            }
            for (; j < d; ++j) {
                const double d0 = xs[j] - (double)col[(size_t)j * kp];
                a0 = fma(d0, d0, a0);
            }
            const double s = (a0 + a1) + (a2 + a3);
            if (c < k && s < best) { best = s; bi = c; }      // NaN never wins
        }
        // (value, index) minimum over the 128 lanes
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int w = (rv[1] < rv[0] || (rv[1] == rv[0] && ri[1] < ri[0])) ? 1 : 0;
            labels[row] = ri[w] == 0x7fffffff ? 0 : ri[w];
        }
    }
}

// both refine passes in one launch (streaming path): all-centres rows one 8-wave block per row, pair rows one wave per row.
#define REFINE_GRID (unsigned)SCD_ABLATE_ENV("SCD_REFINE_GRID", 1024)
#define REFINE_PAIR SCD_ABLATE_ENV("SCD_REFINE_PAIR", 256)
__device__ __forceinline__ void refine_both_body(const float* __restrict__ X, const float* __restrict__ C, const float* __restrict__ ct,
                                                 const EHdr* eh, const int* flag_list, const int* flag_cand, const int* full_list, int d,
                                                 int k, int kp, int32_t* labels, int32_t* refine_rows_out, int pair_blocks, int bx, int gx) {
    extern __shared__ double xs[];
    __shared__ double rv[32];
    __shared__ int ri[32];
    // the first gx - pair_blocks blocks: the all-centres list, one block per row (the long chain of latencies: scheduled first);
    // the others: the pair list, one wave per row.  A block without work ends at once and makes room.
    const int full_grid = gx - pair_blocks;
    if (bx == 0 && threadIdx.x == 0 && refine_rows_out) *refine_rows_out = eh->flag_cnt + eh->full_cnt;
    if (bx >= full_grid) {
        const int lane = threadIdx.x & 63;
        const int cnt = eh->flag_cnt;
        for (int f = (bx - full_grid) * 8 + (threadIdx.x >> 6); f < cnt; f += 8 * pair_blocks) {
            refine_pair_row(X, C, flag_list[f], flag_cand[f], d, lane, labels);
        }
        return;
    }
    // all-centres list: four rows as doubles in LDS; the eight waves sweep every eighth centre, eight centres per pass, whole
    // centre rows in coalesced float4 loads that are all issued before the first is consumed (a walk down the transposed
    // matrix paid one memory latency per 4 columns)
    // up to a round of rows: one row per block (every CU busy, latency-bound); many rows (behind a data-point seeding): four rows per block
    // against the same centre loads (the sweep's L2 traffic is the bound then)
    const int cnt = eh->full_cnt;
    if (cnt <= 2 * full_grid) {
        for (int f = bx; f < cnt; f += full_grid) refine_full_row<8>(X, C, full_list[f], d, k, xs, rv, ri, labels);
    } else {
        for (int f = bx * 4; f < cnt; f += full_grid * 4)
            refine_full_rows4<8>(X, C, full_list + f, cnt - f < 4 ? cnt - f : 4, d, k, xs, rv, ri, labels);
    }
}
__global__ void __launch_bounds__(512) estep_refine_both_kernel(const float* __restrict__ X, const float* __restrict__ C,
                                                                const float* __restrict__ ct, const EHdr* eh, const int* flag_list,
                                                                const int* flag_cand, const int* full_list, int d, int k, int kp,
                                                                int32_t* labels, int32_t* refine_rows_out, int pair_blocks) {
    refine_both_body(X, C, ct, eh, flag_list, flag_cand, full_list, d, k, kp, labels, refine_rows_out, pair_blocks, (int)blockIdx.x, (int)gridDim.x);
}
// the same refine for SEVERAL restarts in one launch (behind estep_rbm_kernel): blockIdx.y = the a-th running restart, whose centres,
// workspace and label slot are strided like the filter's (RbmArgs); the latency-bound chain of an all-centres row is then paid once per
// iteration, not once per restart
struct RefmArgs {
    const float* C0; size_t c_stride;              // centres of slot s: C0 + s * c_stride (floats)
    char* ws0; size_t ws_stride; size_t ct_off, flags_off, fcand_off, fulls_off;
    int32_t* lab0; size_t lab_stride;
    unsigned long long segmap;
};
__global__ void __launch_bounds__(512) estep_refine_multi_kernel(const float* __restrict__ X, const RefmArgs a, int d, int k, int kp, int pair_blocks) {
    const int slot = (int)((a.segmap >> (4 * blockIdx.y)) & 15ull);
    char* w = a.ws0 + (size_t)slot * a.ws_stride;
    refine_both_body(X, a.C0 + (size_t)slot * a.c_stride, (const float*)(w + a.ct_off), (const EHdr*)w, (const int*)(w + a.flags_off),
                     (const int*)(w + a.fcand_off), (const int*)(w + a.fulls_off), d, k, kp, a.lab0 + (size_t)slot * a.lab_stride, nullptr,
                     pair_blocks, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void refine_count_kernel(const EHdr* eh, int32_t* out) { *out = eh->flag_cnt + eh->full_cnt; }

extern "C" int scd_kmeans_estep(scd_handle h, const float* X, const void* prep, const float* C, int64_t n, int d, int k,
                                int32_t* labels_out, int32_t* refine_rows_out, void* ws, size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && X && prep && C && labels_out && ws, "scd_kmeans_estep: null argument");
    { const int rc_ = scd_check_device(h, "scd_kmeans_estep"); if (rc_) return rc_; }
    SCD_REQUIRE(n > 0 && d > 0 && k > 0 && k < 32768 && n < (1ll << 31), "scd_kmeans_estep: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    // (the exact re-evaluation stages four rows as float64 in LDS: 32 d + 64 bytes of the 64 KB a launch gets without opting in)
    SCD_REQUIRE(d <= 2032, "scd_kmeans_estep: d=%d > 2032 is not supported", d);
    SCD_REQUIRE(ws_bytes >= scd_kmeans_estep_ws_bytes(n, d, k), "scd_kmeans_estep: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    // one-shot hints and the finalize hand-over are consumed by THIS call whichever path it takes (left set by a call that took the
    // legacy path, they would apply to an unrelated later call on the handle)
    const bool few_hint = h->estep_few != 0;
    const bool handover = h->prep_ok && h->prep_C == C && h->prep_ws == ws && h->prep_k == k && h->prep_d == d;
    h->estep_few = 0;
    h->prep_ok = 0;
    h->prep_C = nullptr;
    const int dp = dpad(d), kp = kpad(k);
    char* w = (char*)ws;
    EHdr* eh = (EHdr*)w;
    float* cn = (float*)(w + 64);
    half_t* ch = (half_t*)(w + 64 + scd_align(4 * (size_t)kp));
    float* ct = (float*)((char*)ch + scd_align(2 * (size_t)kp * dp));
    int* flags = (int*)((char*)ct + scd_align(4 * (size_t)kp * dp));
    int* fcand = (int*)((char*)flags + scd_align(4 * (size_t)n));
    int* fulls = (int*)((char*)fcand + scd_align(4 * (size_t)n));
    half_t* chf = (half_t*)((char*)fulls + scd_align(4 * (size_t)n));       // centres in MFMA-fragment order (streaming path)
    float* tkey = (float*)((char*)chf + scd_align(2 * (size_t)kp * dp));    // [3][n] keys / centres between the passes of K > 128
    int* tidx = (int*)((char*)tkey + scd_align(12 * (size_t)n));
    const char* p = (const char*)prep;
    const PrepHdr* ph = (const PrepHdr*)p;
    const size_t xnorm_off = scd_align(64 + 8 * (size_t)dp);
    const size_t xh_off = xnorm_off + scd_align(4 * (size_t)n);
    static const int use_stream = getenv("SCD_ESTEP_STREAM") ? atoi(getenv("SCD_ESTEP_STREAM")) : 1;
    static const int use_rb = getenv("SCD_ESTEP_RB") ? atoi(getenv("SCD_ESTEP_RB")) : 1;
    if (use_stream && use_rb && dp == 512 && kp > 128 && kp <= 2048) {
        // single-pass filter (128 < K <= 2048 at Dp = 512): the rows of a block stay in registers, the centres stream past them once
        if (!handover) {
            prep_centers_kernel<<<kp, 256, 0, st>>>(C, k, d, dp, ph, (const double*)(p + 64), eh, cn, ch, ct_inline(ct, dp, kp), kp, 1, nullptr);
            if (ct_after(dp, kp)) ct_transpose_kernel<<<dim3(kp / 32, dp / 32), 256, 0, st>>>(C, k, d, dp, kp, ct);
        }
        half_t* ext = chf;                                       // the fragment-order copy is not used on this path: 16 B per centre
        estep_ext_kernel<<<(unsigned)scd_cdiv(kp, 256), 256, 0, st>>>(cn, kp, ext);
        { const int rc_ = scd_set_max_lds((const void*)estep_rb_kernel, ERB_LDS); if (rc_) return rc_; }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->km_timing) {
            SCD_HIP(hipEventCreate(&ev0));
            SCD_HIP(hipEventCreate(&ev1));
            SCD_HIP(hipEventRecord(ev0, st));
        }
        // the partial last round of the chip is split over the centres (see estep_rb_kernel): 2 or 4 parts while a part keeps >= 8 units
        const long long nblk = scd_cdiv(n, 256);
        const int ncu = h->n_cu > 0 ? h->n_cu : 256;
        const int rem = (int)(nblk % ncu);
        int nsplit = 1;
        if (nblk > ncu && rem > 0) {
            while (nsplit < 4 && rem * (nsplit * 2) <= ncu && (kp / 32) % (nsplit * 2) == 0 && (kp / 32) / (nsplit * 2) >= 8) nsplit *= 2;
        }
        const int nfull = nsplit > 1 ? (int)(nblk - rem) : (int)nblk;
        float* tk = tkey;                                        // [nsplit][3][tail rows] <= 12 n floats
        estep_rb_kernel<<<(unsigned)(nfull + (nblk - nfull) * nsplit), 512, ERB_LDS, st>>>((const half_t*)(p + xh_off), (const float*)(p + xnorm_off), ch,
                                                                                             ext, cn, eh, flags, fcand, fulls, n, kp, labels_out,
                                                                                             nfull, nsplit, tk);
        if (nsplit > 1)
            estep_rb_merge_kernel<<<(unsigned)scd_cdiv(n - (long long)nfull * 256, 256), 256, 0, st>>>(tk, nsplit, (long long)nfull * 256, n,
                                                                                                        (const float*)(p + xnorm_off), cn, kp, eh,
                                                                                                        flags, fcand, fulls, labels_out);
        if (h->km_timing) {
            SCD_HIP(hipEventRecord(ev1, st));
            h->km_ev.emplace_back(ev0, ev1);
        }
        estep_refine_both_kernel<<<REFINE_GRID, 512, (size_t)d * 32 + 64, st>>>(X, C, ct, eh, flags, fcand, fulls, d, k, kp, labels_out,
                                                                                refine_rows_out, REFINE_PAIR);
        SCD_LAUNCH_CHECK();
        return SCD_OK;
    }
    if (use_stream && kp <= 2048 && dp <= 768) {
        // streaming filter (D <= 768): centre prep (unless scd_kmeans_finalize has just produced these very centres and their
        // operands into this workspace), one filter launch per 128 centres, refine; no memset
        const bool prepared = handover;
        if (!prepared) {
            prep_centers_kernel<<<kp, 256, 0, st>>>(C, k, d, dp, ph, (const double*)(p + 64), eh, cn, ch, ct_inline(ct, dp, kp), kp, 1, chf);
            if (ct_after(dp, kp)) ct_transpose_kernel<<<dim3(kp / 32, dp / 32), 256, 0, st>>>(C, k, d, dp, kp, ct);
        }
        const long long g32 = (n + 31) / 32;               // units of 32 rows
        long long grid = g32 < h->n_cu ? g32 : h->n_cu;
        if (grid < scd_cdiv(g32, ES_RMAX / 32)) grid = scd_cdiv(g32, ES_RMAX / 32);
        const half_t* xh = (const half_t*)(p + xh_off);
        const float* xn = (const float*)(p + xnorm_off);
        static const int es_dbg = SCD_ABLATE_ENV("SCD_ESTEP_DBG", 0);
        static const int split = SCD_ABLATE_ENV("SCD_ESTEP_REFINE_SPLIT", 0);   // 0: refine in the stream kernel's tail
        // refine in the stream kernel's tail only when the caller expects few flagged rows (scd_kmeans_estep_hint: Lloyd iterations
        // after the first two): with ~10 % of the rows flagged the one-block-per-CU tail takes 190 us where the refine kernel at
        // full occupancy takes 60-100; with none flagged the tail saves the launch (37 -> 31 us per call)
        const bool tail = split == 0 && d <= 1024 && few_hint;
#define ES_LAUNCH(NCH)                                                                                                       \
    case NCH: {                                                                                                              \
        { const int rc_ = scd_set_max_lds((const void*)estep_stream_kernel<NCH>, ES_LDS); if (rc_) return rc_; }                                                                                                                    \
        for (int cb = 0; cb < kp / 128; ++cb)                                                                                \
            estep_stream_kernel<NCH><<<(unsigned)grid, 256, ES_LDS, st>>>(xh, xn, chf + (size_t)cb * 128 * dp, cn + cb * 128, eh, flags, fcand, \
                                                                          fulls, n, labels_out, es_dbg, cb * 128,           \
                                                                          (cb == 0 ? 1 : 0) | (cb == kp / 128 - 1 ? 2 : 0), tkey, tidx, cn, kp,  \
                                                                          tail ? X : nullptr, C, d, k);                       \
    } break;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->km_timing) {
            SCD_HIP(hipEventCreate(&ev0));
            SCD_HIP(hipEventCreate(&ev1));
            SCD_HIP(hipEventRecord(ev0, st));
        }
        switch (dp / 128) {
            ES_LAUNCH(1) ES_LAUNCH(2) ES_LAUNCH(3) ES_LAUNCH(4) ES_LAUNCH(5) ES_LAUNCH(6)
        }
#undef ES_LAUNCH
        if (h->km_timing) {
            SCD_HIP(hipEventRecord(ev1, st));
            h->km_ev.emplace_back(ev0, ev1);
        }
        if (tail) {
            if (refine_rows_out) refine_count_kernel<<<1, 1, 0, st>>>(eh, refine_rows_out);
        } else if (split == 2) {
            // debugging: filter labels only
        } else if (split == 1) {
            estep_refine_kernel<<<2048, 64, 0, st>>>(X, C, eh, flags, fcand, d, k, labels_out);
            estep_refine_full_kernel<<<2048, 128, (size_t)d * 8 + 64, st>>>(X, ct, eh, fulls, d, k, kp, labels_out);
            if (refine_rows_out) refine_count_kernel<<<1, 1, 0, st>>>(eh, refine_rows_out);
        } else
        estep_refine_both_kernel<<<REFINE_GRID, 512, (size_t)d * 32 + 64, st>>>(X, C, ct, eh, flags, fcand, fulls, d, k, kp, labels_out,
                                                                                refine_rows_out, REFINE_PAIR);
        SCD_LAUNCH_CHECK();
        return SCD_OK;
    }
    SCD_HIP(hipMemsetAsync(eh, 0, 64, st));
    // (legacy path: Kp > 2048 or Dp > 768 - estep_refine_full_kernel below reads the transposed copy)
    prep_centers_kernel<<<kp, 256, 0, st>>>(C, k, d, dp, ph, (const double*)(p + 64), eh, cn, ch, ct_separate(kp) ? nullptr : ct, kp, 0, nullptr);
    if (ct_separate(kp)) ct_transpose_kernel<<<dim3(kp / 32, dp / 32), 256, 0, st>>>(C, k, d, dp, kp, ct);
    estep_mfma_kernel<<<(unsigned)scd_cdiv(n, 128), 256, 0, st>>>((const half_t*)(p + xh_off), (const float*)(p + xnorm_off),
                                                                    ch, cn, eh, flags, fcand, fulls, n, dp, kp, labels_out);
    estep_refine_kernel<<<2048, 64, 0, st>>>(X, C, eh, flags, fcand, d, k, labels_out);
    estep_refine_full_kernel<<<2048, 128, (size_t)d * 8 + 64, st>>>(X, ct, eh, fulls, d, k, kp, labels_out);
    if (refine_rows_out) refine_count_kernel<<<1, 1, 0, st>>>(eh, refine_rows_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// exact row distance to the assigned centre / to one new centre (k-means++), one wave per row
template <bool MINUPD>
__global__ void __launch_bounds__(256) rowdist_kernel(const float* __restrict__ X, const float* __restrict__ C,
                                                      const int32_t* __restrict__ labels, long long n, int d, int k,
                                                      float* d2) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + row * d;
    const float* c = C;
    if (!MINUPD) {
        int l = labels[row];
        if (l < 0 || l >= k) {
            if (lane == 0) d2[row] = NAN;
            return;
        }
        c = C + (size_t)l * d;
    }
    double s = 0.0;
    if ((d & 3) == 0) {
        for (int j = lane * 4; j < d; j += 256) {
            const float4 xv = *(const float4*)(x + j);
            const float4 cv = *(const float4*)(c + j);
            double a = (double)xv.x - (double)cv.x; s = fma(a, a, s);
            a = (double)xv.y - (double)cv.y; s = fma(a, a, s);
            a = (double)xv.z - (double)cv.z; s = fma(a, a, s);
            a = (double)xv.w - (double)cv.w; s = fma(a, a, s);
        }
    } else {
        for (int j = lane; j < d; j += 64) {
            const double a = (double)x[j] - (double)c[j];
            s = fma(a, a, s);
        }
    }
    s = wave_sum_f64(s);
    if (lane == 0) {
        const float v = (float)s;
        d2[row] = MINUPD ? fminf(d2[row], v) : v;
    }
}

extern "C" int scd_kmeans_rowdist(scd_handle h, const float* X, const float* C, const int32_t* labels, int64_t n, int d,
                                  int k, float* d2_out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_rowdist");
    SCD_REQUIRE(h && X && C && labels && d2_out && n > 0 && d > 0 && k > 0, "scd_kmeans_rowdist: bad arguments");
    rowdist_kernel<false><<<(unsigned)scd_cdiv(n, 4), 256, 0, (hipStream_t)stream_>>>(X, C, labels, n, d, k, d2_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_kmeans_min_update_multi(scd_handle h, const float* X, const float* c_new, int64_t n, int d, int R,
                                           float* d2_inout, int64_t ld, void* stream_);
extern "C" int scd_kmeans_min_update(scd_handle h, const float* X, const float* c_new, int64_t n, int d, float* d2_inout,
                                     void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_min_update");
    SCD_REQUIRE(h && X && c_new && d2_inout && n > 0 && d > 0, "scd_kmeans_min_update: bad arguments");
    return scd_kmeans_min_update_multi(h, X, c_new, n, d, 1, d2_inout, n, stream_);
}

// full [n,k] exact distances (pairwise_distance, sskm_constrained.py:189-224; the ConSSKM cost matrix, :277-287): float64 difference
// form, rounded once to float32.  A block takes 64 rows x 32 centres: the rows' values go through LDS transposed ([j][row], so a wave
// reads one j of its 64 rows conflict-free), a lane owns one row and CG = 8 centres of its wave, the centres' values arrive through
// scalar loads (wave-uniform addresses) - no cross-lane reduction anywhere.  Round 6: the one-wave-per-4-rows kernel it replaces spent
// 661 us per 9,000 x 120 x 768 launch, most of it in the 6-step float64 butterflies behind every (row, centre) pair; 92 launches per
// ConSSKM fit were 61 ms of the 149-ms fit (profiles/r06_bench_c3_kernel_stats.csv).
template <int CG>
__global__ void __launch_bounds__(256) dist_tile_kernel(const float* __restrict__ X, const float* __restrict__ C, long long n,
                                                        int d, int k, int mode, float* __restrict__ out, int32_t* __restrict__ cost) {
    __shared__ float xs[64][65];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long row0 = (long long)blockIdx.x * 64;
    const int c0 = (int)blockIdx.y * (4 * CG) + wave * CG;
    double acc[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) acc[q] = 0.0;
    const float* cq[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) cq[q] = C + (size_t)(c0 + q < k ? c0 + q : k - 1) * d;
    // staging: thread -> (j = tid % 64, rows tid / 64 + 4 i): coalesced along j in HBM, stride-65 (conflict-free) in LDS
    const int sj = threadIdx.x & 63, sr = threadIdx.x >> 6;
    const float* xrow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xrow[i] = X + (row0 + sr + 4 * i < n ? row0 + sr + 4 * i : n - 1) * d + sj;
    for (int j0 = 0; j0 < d; j0 += 64) {
        const int jn = d - j0 < 64 ? d - j0 : 64;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[sj][sr + 4 * i] = sj < jn ? xrow[i][j0] : 0.f;
        __syncthreads();
        if (c0 >= k) continue;
        if (jn == 64) {
#pragma unroll 8
            for (int jj = 0; jj < 64; ++jj) {
                const double xv = (double)xs[jj][lane];
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const double a = xv - (double)cq[q][j0 + jj];
                    acc[q] = fma(a, a, acc[q]);
                }
            }
        } else {
            for (int jj = 0; jj < jn; ++jj) {
                const double xv = (double)xs[jj][lane];
#pragma unroll
                for (int q = 0; q < CG; ++q) {
                    const double a = xv - (double)cq[q][j0 + jj];
                    acc[q] = fma(a, a, acc[q]);
                }
            }
        }
    }
    const long long row = row0 + lane;
    if (row < n) {
#pragma unroll
        for (int q = 0; q < CG; ++q) {
            if (c0 + q < k) {
                const float d2 = (float)acc[q];
                const float rt = sqrtf(d2);           // correctly rounded
                out[row * k + c0 + q] = mode ? rt : d2;
                if (cost) cost[row * k + c0 + q] = (int32_t)rintf(rt * 1000.0f);
            }
        }
    }
}

extern "C" int scd_kmeans_dist(scd_handle h, const float* X, const float* C, int64_t n, int d, int k, int mode, float* out,
                               int32_t* cost_out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_dist");
    SCD_REQUIRE(h && X && C && out && n > 0 && d > 0 && k > 0, "scd_kmeans_dist: bad arguments");
    SCD_REQUIRE(scd_cdiv(n, 64) < (1ll << 31) && scd_cdiv(k, 16) < 65536, "scd_kmeans_dist: n = %lld, k = %d exceed the launch grid", (long long)n, k);
    // centres per lane: 8, or 4 while the 8-centre grid is below four blocks per CU (the ConSSKM shape, 9,000 x 120: 564 blocks -> 1,128,
    // 168 -> 143 us; at 95,000 x 100 and 30,000 x 200 the two are within 3 %, 8 ahead).  16 is register- and grid-starved (358 us), 2 re-reads
    // the rows too often (156 us).  Same sums either way (a lane's sum does not depend on how many centres it carries).
    static const int cg_env = getenv("SCD_DIST_CG") ? atoi(getenv("SCD_DIST_CG")) : 0;
    const int cg = cg_env == 4 || cg_env == 8 ? cg_env : (scd_cdiv(n, 64) * scd_cdiv(k, 32) < 1024 ? 4 : 8);
    if (cg == 4) dist_tile_kernel<4><<<dim3((unsigned)scd_cdiv(n, 64), (unsigned)scd_cdiv(k, 16)), 256, 0, (hipStream_t)stream_>>>(X, C, n, d, k, mode, out, cost_out);
    else dist_tile_kernel<8><<<dim3((unsigned)scd_cdiv(n, 64), (unsigned)scd_cdiv(k, 32)), 256, 0, (hipStream_t)stream_>>>(X, C, n, d, k, mode, out, cost_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// (M-step partial sums: see mstep.hip)

// double-double helpers (used by finalize_kernel's fused inertia and the incremental M-step further down)
struct dd_t { double hi, lo; };
__device__ __forceinline__ dd_t dd_add_d(dd_t a, double b) {      // a + b, b a plain double
    const double s = a.hi + b;
    const double bb = s - a.hi;
    const double e = (a.hi - (s - bb)) + (b - bb);
    const double lo = a.lo + e;
    const double hi = s + lo;
    return {hi, lo - (hi - s)};
}
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b) { return dd_add_d(dd_add_d(a, b.hi), b.lo); }
__device__ __forceinline__ dd_t dd_add_prod(dd_t a, double x, double y) {    // a + x * y, the product error-free
    const double p = x * y;
    const double e = fma(x, y, -p);
    return dd_add_d(dd_add_d(a, p), e);
}
__device__ __forceinline__ dd_t dd_wave_sum(dd_t v) {            // fixed butterfly order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dd_t ov;
        ov.hi = __shfl_xor(v.hi, o, 64);
        ov.lo = __shfl_xor(v.lo, o, 64);
        v = dd_add(v, ov);
    }
    return v;
}
// centres = sums / counts; shift = (sum_k ||c_k - c_old_k||)^2.  One block per centre; the per-centre norms go to a scratch
// array and the LAST block to arrive (ticket) adds them in index order, so the float64 result does not depend on the
// arrival order.  (A single 1024-thread block took 31 us for K x D = 77k values.)  The scratch belongs to the handle: one
// finalize in flight per handle.
__global__ void __launch_bounds__(256) finalize_kernel(const double* sums, const long long* counts, int k, int d,
                                                       const float* Cold, float* Cout, double* shift, double* part,
                                                       unsigned* ticket, int shift_mode, const PrepHdr* ph, const double* mu,
                                                       EHdr* eh, float* cn, half_t* ch, float* ct, int kp, half_t* chf,
                                                       double* refined_out, double* changed_out, double* changed_acc,
                                                       const double* stats5, double* mirror, double seq,
                                                       const double* sums_lab, const long long* counts_lab, const double* sumsq4,
                                                       double* inertia_out) {
    __shared__ double wred[4];
    __shared__ double red[4][6];
    __shared__ bool last;
    // inertia_out != NULL (incremental M-step): this launch also evaluates the iteration's inertia from the sums (the arithmetic of
    // inertia_dd_kernel, partials behind the k shift partials) - both walk Cold / sums / counts per centre, one launch less
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // rows the E-step of this iteration re-evaluated exactly (read before this block's prep_center_row zeroes the counters)
    if (refined_out && eh && c == 0 && threadIdx.x == 0) *refined_out = (double)(eh->flag_cnt + eh->full_cnt);
    if (c >= k) {                                   // padded centres of the fused E-step prep (grid = kp blocks)
        prep_center_row(Cout, c, k, d, ph->dp, ph, mu, eh, cn, ch, ct, kp, 1, chf);
        return;
    }
    const double cnt = (double)counts[c];
    double ss = 0.0;
    const long long nl = (inertia_out && counts_lab) ? counts_lab[c] : 0, nu = counts[c] - nl;
    dd_t cc = {0.0, 0.0}, dl = {0.0, 0.0}, du = {0.0, 0.0};
    for (int j = threadIdx.x; j < d; j += 256) {
        const double sj = sums[(size_t)c * d + j];
        const float v = (float)(sj / cnt);     // 0/0 -> NaN for an empty cluster
        Cout[(size_t)c * d + j] = v;
        if (Cold) {
            const double co = (double)Cold[(size_t)c * d + j];
            const double df = (double)v - co;
            ss = fma(df, df, ss);
            if (inertia_out) {
                const double sl = sums_lab ? sums_lab[(size_t)c * d + j] : 0.0;
                const double su = sj - sl;          // exact: both are exact sums of multiples of 2^-24
                cc = dd_add_prod(cc, co, co);
                if (nl) dl = dd_add_prod(dl, co, sl);
                if (nu) du = dd_add_prod(du, co, su);
            }
        }
    }
    if (inertia_out) {
        cc = dd_wave_sum(cc); dl = dd_wave_sum(dl); du = dd_wave_sum(du);
        if (lane == 0) { red[wave][0] = cc.hi; red[wave][1] = cc.lo; red[wave][2] = dl.hi; red[wave][3] = dl.lo; red[wave][4] = du.hi; red[wave][5] = du.lo; }
    }
    if (ph) {                                       // the next E-step's operands of this centre, while its row is still in cache
        __syncthreads();
        prep_center_row(Cout, c, k, d, ph->dp, ph, mu, eh, cn, ch, ct, kp, 1, chf);
        __syncthreads();
    }
    if (!shift) return;
    ss = wave_sum_f64(ss);
    if (lane == 0) wred[wave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sq = (wred[0] + wred[1]) + (wred[2] + wred[3]);
        part[c] = shift_mode ? sq : sqrt(sq);
        if (inertia_out) {
            dd_t n2 = {0.0, 0.0}, pl = {0.0, 0.0}, pu = {0.0, 0.0};
            for (int w = 0; w < 4; ++w) { n2 = dd_add(n2, {red[w][0], red[w][1]}); pl = dd_add(pl, {red[w][2], red[w][3]}); pu = dd_add(pu, {red[w][4], red[w][5]}); }
            // t = n ||c||^2 - 2 <c, S>   (n < 2^31: the products with hi / lo are formed error-free)
            dd_t tl = {0.0, 0.0}, tu = {0.0, 0.0};
            if (nl) { tl = dd_add_prod(dd_add_prod(tl, (double)nl, n2.hi), (double)nl, n2.lo); tl = dd_add_d(dd_add_d(tl, -2.0 * pl.hi), -2.0 * pl.lo); }
            if (nu) { tu = dd_add_prod(dd_add_prod(tu, (double)nu, n2.hi), (double)nu, n2.lo); tu = dd_add_d(dd_add_d(tu, -2.0 * pu.hi), -2.0 * pu.lo); }
            double* ip = part + k;
            ip[c * 4 + 0] = tl.hi; ip[c * 4 + 1] = tl.lo; ip[c * 4 + 2] = tu.hi; ip[c * 4 + 3] = tu.lo;
        }
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned)k - 1;
    }
    __syncthreads();
    if (last) {                                   // fixed order: thread t adds part[t], part[t+256], ...; then a fixed tree
        __shared__ double tred[256];
        __threadfence();
        double t = 0.0;
        for (int i = threadIdx.x; i < k; i += 256) t += ((volatile double*)part)[i];
        tred[threadIdx.x] = t;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) tred[threadIdx.x] += tred[threadIdx.x + o];
            __syncthreads();
        }
        // clusters that received no row (scd_kmeans_lloyd_run_sk: sklearn relocates them, the caller has to step in)
        __shared__ int n_empty;
        if (threadIdx.x == 0) n_empty = 0;
        __syncthreads();
        if (mirror) {
            int ne = 0;
            for (int i = threadIdx.x; i < k; i += 256) ne += counts[i] == 0;
            if (ne) atomicAdd(&n_empty, ne);
        }
        __syncthreads();
        double in_l = 0.0, in_u = 0.0;
        if (inertia_out) {                        // the K partials in the fixed tree order of inertia_dd_kernel
            __shared__ double tr[256][4];
            const double* ip = part + k;
            dd_t il = {0.0, 0.0}, iu = {0.0, 0.0};
            for (int q = threadIdx.x; q < k; q += 256) {
                il = dd_add(il, {((volatile const double*)ip)[q * 4], ((volatile const double*)ip)[q * 4 + 1]});
                iu = dd_add(iu, {((volatile const double*)ip)[q * 4 + 2], ((volatile const double*)ip)[q * 4 + 3]});
            }
            tr[threadIdx.x][0] = il.hi; tr[threadIdx.x][1] = il.lo; tr[threadIdx.x][2] = iu.hi; tr[threadIdx.x][3] = iu.lo;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) {
                    const dd_t a = dd_add({tr[threadIdx.x][0], tr[threadIdx.x][1]}, {tr[threadIdx.x + o][0], tr[threadIdx.x + o][1]});
                    const dd_t b = dd_add({tr[threadIdx.x][2], tr[threadIdx.x][3]}, {tr[threadIdx.x + o][2], tr[threadIdx.x + o][3]});
                    tr[threadIdx.x][0] = a.hi; tr[threadIdx.x][1] = a.lo; tr[threadIdx.x][2] = b.hi; tr[threadIdx.x][3] = b.lo;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                const dd_t fl = dd_add({sumsq4[0], sumsq4[1]}, {tr[0][0], tr[0][1]}), fu = dd_add({sumsq4[2], sumsq4[3]}, {tr[0][2], tr[0][3]});
                in_l = fl.hi + fl.lo;
                in_u = fu.hi + fu.lo;
                inertia_out[0] = in_l;
                inertia_out[1] = in_u;
            }
        }
        if (threadIdx.x == 0) {
            const double sh = shift_mode ? tred[0] : tred[0] * tred[0];
            *shift = sh;
            *ticket = 0;
            // rows whose label changed this iteration (accumulated by mstep_delta_kernel / labels_sync_kernel in the handle's scratch;
            // nothing adds to it until the next iteration's kernels, which run after this one): hand over and reset
            double chg = 0.0;
            if (changed_out) {
                chg = *(volatile double*)changed_acc;
                *changed_out = chg;
                *changed_acc = 0.0;
            }
            if (mirror) {
                // scd_kmeans_lloyd_run: the iteration's five statistics straight into pinned host memory, then the sequence number the
                // host spins on (system-scope fence in between) - no copy operation, no event in the stream
                mirror[0] = inertia_out ? in_l : ((volatile const double*)stats5)[0];
                mirror[1] = inertia_out ? in_u : ((volatile const double*)stats5)[1];
                mirror[2] = sh;
                mirror[3] = ((volatile const double*)stats5)[3];
                mirror[4] = chg;
                mirror[5] = (double)n_empty;
                __threadfence_system();
                ((volatile double*)mirror)[7] = seq;
            }
        }
    }
}

static int finalize_impl(scd_handle h, const double* sums, const int64_t* counts, int k, int d, const float* C_old, float* C_out,
                         double* shift_out, int shift_mode, const void* prep, void* estep_ws, size_t estep_ws_bytes, int64_t n,
                         void* stream_, double* refined_out, double* changed_out = nullptr, const double* stats5 = nullptr,
                         double* mirror = nullptr, double seq = 0.0, const double* sums_lab = nullptr, const int64_t* counts_lab = nullptr,
                         const double* sumsq4 = nullptr, double* inertia_out = nullptr) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_finalize");
    double* changed_acc = (double*)((char*)h->scratch + 262144 + 40);
    SCD_REQUIRE(h && sums && counts && C_out && k > 0 && d > 0, "scd_kmeans_finalize: bad arguments");
    SCD_REQUIRE(C_old != C_out, "scd_kmeans_finalize: C_out must not alias C_old");
    SCD_REQUIRE(k <= 32768, "scd_kmeans_finalize: k=%d > 32768", k);
    const int dp = dpad(d), kp = kpad(k);
    static const int use_stream = getenv("SCD_ESTEP_STREAM") ? atoi(getenv("SCD_ESTEP_STREAM")) : 1;
    static const int fuse_env = getenv("SCD_FINALIZE_PREP") ? atoi(getenv("SCD_FINALIZE_PREP")) : 1;
    const bool fuse = prep && estep_ws && fuse_env && use_stream && kp <= 2048 && dp <= 768;
    h->prep_C = nullptr;
    if (fuse) {
        // the NEXT E-step's centre operands, produced by the blocks that produce the centres (saves the prep launch and its gap)
        SCD_REQUIRE(n > 0 && estep_ws_bytes >= scd_kmeans_estep_ws_bytes(n, d, k), "scd_kmeans_finalize: E-step workspace too small");
        char* w = (char*)estep_ws;
        EHdr* eh = (EHdr*)w;
        float* cn = (float*)(w + 64);
        half_t* ch = (half_t*)(w + 64 + scd_align(4 * (size_t)kp));
        float* ct = (float*)((char*)ch + scd_align(2 * (size_t)kp * dp));
        half_t* chf = (half_t*)((char*)ct + scd_align(4 * (size_t)kp * dp) + 3 * scd_align(4 * (size_t)n));
        const char* p = (const char*)prep;
        finalize_kernel<<<kp, 256, 0, (hipStream_t)stream_>>>(sums, (const long long*)counts, k, d, C_old, C_out, shift_out,
                                                              (double*)h->scratch, (unsigned*)((char*)h->scratch + 262144), shift_mode,
                                                              (const PrepHdr*)p, (const double*)(p + 64), eh, cn, ch, ct_inline(ct, dp, kp), kp,
                                                              chf_unused(dp, kp) ? nullptr : chf, refined_out,
                                                              changed_out, changed_acc, stats5, mirror, seq, sums_lab, (const long long*)counts_lab,
                                                              sumsq4, inertia_out);
        if (ct_after(dp, kp)) ct_transpose_kernel<<<dim3(kp / 32, dp / 32), 256, 0, (hipStream_t)stream_>>>(C_out, k, d, dp, kp, ct);
        h->prep_C = C_out;
        h->prep_ws = estep_ws;
        h->prep_k = k;
        h->prep_d = d;
    } else {
        finalize_kernel<<<k, 256, 0, (hipStream_t)stream_>>>(sums, (const long long*)counts, k, d, C_old, C_out, shift_out,
                                                             (double*)h->scratch, (unsigned*)((char*)h->scratch + 262144), shift_mode,
                                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kp, nullptr, nullptr,
                                                             changed_out, changed_acc, stats5, mirror, seq, sums_lab, (const long long*)counts_lab,
                                                             sumsq4, inertia_out);
    }
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_kmeans_finalize(scd_handle h, const double* sums, const int64_t* counts, int k, int d,
                                   const float* C_old, float* C_out, double* shift_out, int shift_mode, const void* prep,
                                   void* estep_ws, size_t estep_ws_bytes, int64_t n, void* stream_) {
    return finalize_impl(h, sums, counts, k, d, C_old, C_out, shift_out, shift_mode, prep, estep_ws, estep_ws_bytes, n, stream_, nullptr);
}

extern "C" int scd_kmeans_timing(scd_handle h, int enable, double* samples_ms_out, int cap, int* launches_out) {
    SCD_REQUIRE(h, "scd_kmeans_timing: null handle");
    int i = 0;
    for (auto& pr : h->km_ev) {
        SCD_HIP(hipEventSynchronize(pr.second));
        float t = 0.f;
        SCD_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
        if (samples_ms_out && i < cap) samples_ms_out[i] = t;
        ++i;
        hipEventDestroy(pr.first);
        hipEventDestroy(pr.second);
    }
    if (launches_out) *launches_out = i;
    h->km_ev.clear();
    h->km_timing = enable != 0;
    return SCD_OK;
}

extern "C" int scd_kmeans_estep_hint(scd_handle h, int flags) {
    SCD_REQUIRE(h, "scd_kmeans_estep_hint: null handle");
    h->estep_few = (flags & SCD_ESTEP_FEW) != 0;
    h->prep_ok = (flags & SCD_ESTEP_CENTRES_FROM_FINALIZE) != 0;
    return SCD_OK;
}

// number of positions where two label vectors differ (sklearn's strict-convergence test `np.array_equal(labels, labels_old)`,
// sklearn/cluster/_kmeans.py `_kmeans_single_lloyd`): one atomic per block into *out, which the caller zeroed.
__global__ void __launch_bounds__(256) labels_changed_kernel(const int* __restrict__ a, const int* __restrict__ b, long long n,
                                                             unsigned long long* out) {
    long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    unsigned c = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q, i += 256)
        if (i < n) c += a[i] != b[i];
    c = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_sum_f32((float)c));     // <= 256 per wave: exact in float
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

extern "C" int scd_labels_changed(scd_handle h, const int32_t* a, const int32_t* b, int64_t n, int64_t* out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_labels_changed");
    SCD_REQUIRE(h && a && b && out && n > 0, "scd_labels_changed: bad arguments");
    SCD_HIP(hipMemsetAsync(out, 0, 8, (hipStream_t)stream_));
    labels_changed_kernel<<<(unsigned)scd_cdiv(n, 1024), 256, 0, (hipStream_t)stream_>>>(a, b, n, (unsigned long long*)out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// deterministic float64 sum and the k-means++ draw: single block of 1024 threads, thread t owns a
// contiguous segment so that prefix sums follow the sequential (torch CPU cumsum) order.
__device__ __forceinline__ double block_scan_excl_1024(double v, double* sh, double* total) {
    // sh: 1024 doubles.  Simple two-level scan: per-wave inclusive scan + serial over 16 wave totals.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int i = 0; i < 16; ++i) {
            const double t = sh[i];
            sh[i] = run;
            run += t;
        }
        sh[16] = run;
    }
    __syncthreads();
    const double excl = sh[wave] + (inc - v);
    if (total) *total = sh[16];
    __syncthreads();
    return excl;
}

// Thread t sums its contiguous segment [t seg, (t + 1) seg) in index order, the 1,024 segment sums go through the block scan: that
// ORDER defines the result's bits (float64 sum of float32 values).  The values reach the threads through LDS - a slab of whole segments
// loaded with coalesced reads, then every owner walks its segment there: read straight from global memory each thread strode seg
// floats apart from its neighbours (75-86 us for 95,000 values; one call per start of a --cluster KM fit).
#define SUM_SLAB 32768                      // floats per slab (128 KB of dynamic LDS)
__global__ void __launch_bounds__(1024) sum_kernel(const float* __restrict__ x, long long n, double* out) {
    extern __shared__ float slab[];
    __shared__ double sh[32];
    const long long seg = scd_cdiv_dev(n, 1024);
    double s = 0.0;
    if (seg <= SUM_SLAB) {
        const long long per = (SUM_SLAB / seg) * seg;              // whole segments per slab
        for (long long base = 0; base < n; base += per) {
            const long long cnt = n - base < per ? n - base : per;
            __syncthreads();
            for (long long i = threadIdx.x; i < cnt; i += 1024) slab[i] = x[base + i];
            __syncthreads();
            const long long a = threadIdx.x * seg;                  // my segment, if it lies in this slab
            if (a >= base && a < base + cnt) {
                const long long b = (a + seg < n) ? a + seg : n;
                for (long long i = a; i < b; ++i) s += (double)slab[i - base];
            }
        }
    } else {
        const long long a = threadIdx.x * seg, b = (a + seg < n) ? a + seg : n;
        for (long long i = a; i < b; ++i) s += (double)x[i];
    }
    double tot;
    block_scan_excl_1024(s, sh, &tot);
    if (threadIdx.x == 0) *out = tot;
}

extern "C" int scd_sum_f32(scd_handle h, const float* x, int64_t n, double* out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_sum_f32");
    SCD_REQUIRE(h && x && out && n > 0, "scd_sum_f32: bad arguments");
    { const int rc_ = scd_set_max_lds((const void*)sum_kernel, SUM_SLAB * 4); if (rc_) return rc_; }
    sum_kernel<<<1, 1024, SUM_SLAB * 4, (hipStream_t)stream_>>>(x, n, out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// k-means++ draw, three short multi-block kernels over tiles of KPP_TILE elements (thread t owns 4 consecutive elements,
// so within-tile prefix sums follow index order):
//   kpp_tile_sum   bsum[b] = sum of d2 over tile b (float64)
//   kpp_tile_prob  psum[b] = sum of prob = d2/float32(total) over tile b (float64); total = sum(bsum) or *total_in
//   kpp_pick       prefix over psum (+ *prefix_in) locates the first tile whose running sum reaches r; that tile is
//                  scanned in index order for the first i with float32(running) >= r.
#define KPP_TILE 4096
#define KPP_STAGE 1024            /* per-tile sums a pick / search block stages in LDS (n <= 4.2 M rows; beyond: read from L2) */
extern "C" size_t scd_kpp_draw_ws_bytes(int64_t n) { return scd_align(16 * (size_t)scd_cdiv(n, KPP_TILE) + 64) + 256; }

__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    return t;
}

__global__ void __launch_bounds__(1024) kpp_tile_sum_kernel(const float* __restrict__ d2, long long n, double* bsum) {
    __shared__ double sh[16];
    const long long i0 = (long long)blockIdx.x * KPP_TILE + threadIdx.x * 4;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < n) s += (double)d2[i0 + q];
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

__device__ __forceinline__ float kpp_total(const double* bsum, int nb, const double* total_in) {
    if (total_in) return (float)*total_in;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += bsum[b];      // fixed order
    return (float)t;
}

__global__ void __launch_bounds__(1024) kpp_tile_prob_kernel(const float* __restrict__ d2, long long n, const double* bsum, int nb,
                                                             const double* total_in, double* psum) {
    __shared__ double sh[16];
    const float totf = kpp_total(bsum, nb, total_in);
    const long long i0 = (long long)blockIdx.x * KPP_TILE + threadIdx.x * 4;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < n) s += (double)__fdiv_rn(d2[i0 + q], totf);
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) psum[blockIdx.x] = s;
}

__global__ void __launch_bounds__(1024) kpp_pick_kernel(const float* __restrict__ d2, long long n, float r, const double* bsum,
                                                        const double* psum, int nb, const double* total_in, const double* prefix_in,
                                                        long long* idx_out, double* probsum_out) {
    __shared__ double sh[32];
    __shared__ long long best;
    __shared__ int owner;
    __shared__ double owner_pre;
    const float totf = kpp_total(bsum, nb, total_in);
    if (threadIdx.x == 0) {
        double run = prefix_in ? *prefix_in : 0.0;
        int ow = -1;
        double opre = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double nxt = run + psum[b];
            if (ow < 0 && (float)nxt >= r) { ow = b; opre = run; }
            run = nxt;
        }
        owner = ow;
        owner_pre = opre;
        best = 0x7fffffffffffffffll;
        if (probsum_out) *probsum_out = run - (prefix_in ? *prefix_in : 0.0);
    }
    __syncthreads();
    if (!idx_out) return;
    if (owner < 0) {
        if (threadIdx.x == 0) *idx_out = -1;
        return;
    }
    // scan the owner tile (and, for the last-ulp case where association differs, the following tiles) in index order
    double pre = owner_pre;
    for (int b = owner; b < nb; ++b) {
        const long long i0 = (long long)b * KPP_TILE + threadIdx.x * 4;
        float pv[4];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pv[q] = (i0 + q < n) ? __fdiv_rn(d2[i0 + q], totf) : 0.f;
            s += (double)pv[q];
        }
        double tot;
        double run = pre + block_scan_excl_1024(s, sh, &tot);
        long long found = 0x7fffffffffffffffll;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            run += (double)pv[q];
            if (found == 0x7fffffffffffffffll && i0 + q < n && (float)run >= r) found = i0 + q;
        }
        if (found != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&best, (unsigned long long)found);
        __syncthreads();
        if (best != 0x7fffffffffffffffll) break;
        pre += tot;
    }
    if (threadIdx.x == 0) *idx_out = (best == 0x7fffffffffffffffll) ? -1 : best;
}

extern "C" int scd_kpp_draw(scd_handle h, const float* d2, int64_t n, float r, const double* total, const double* prefix,
                            int64_t* idx_out, double* probsum_out, void* ws, size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_draw");
    SCD_REQUIRE(h && d2 && (idx_out || probsum_out) && n > 0 && ws, "scd_kpp_draw: bad arguments");
    SCD_REQUIRE(ws_bytes >= scd_kpp_draw_ws_bytes(n), "scd_kpp_draw: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const int nb = (int)scd_cdiv(n, KPP_TILE);
    double* bsum = (double*)ws;
    double* psum = bsum + nb;
    if (!total) kpp_tile_sum_kernel<<<nb, 1024, 0, st>>>(d2, n, bsum);
    kpp_tile_prob_kernel<<<nb, 1024, 0, st>>>(d2, n, bsum, nb, total, psum);
    kpp_pick_kernel<<<1, 1024, 0, st>>>(d2, n, r, bsum, psum, nb, total, prefix, (long long*)idx_out, probsum_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// sklearn's k-means++ candidate draw (sklearn/cluster/_kmeans.py `_kmeans_plusplus`):
//     rand_vals = random_state.uniform(size=L) * current_pot;  ids = searchsorted(stable_cumsum(closest_dist_sq), rand_vals)
// for L uniforms in one launch: block l finds the first i with cumsum_f64(d2)[i] >= u[l] * pot, clipped to n - 1, where
// pot = float32(sum d2) as sklearn's float32 `closest_dist_sq @ sample_weight`.  pot_out (may be NULL) receives the float64 sum.
__global__ void __launch_bounds__(1024) kpp_search_kernel(const float* __restrict__ d2, long long n, const double* __restrict__ u,
                                                          const double* __restrict__ bsum, int nb, long long* idx_out,
                                                          double* pot_out) {
    __shared__ double sh[32];
    __shared__ long long best;
    __shared__ int owner;
    __shared__ double owner_pre;
    if (threadIdx.x == 0) {
        double pot = 0.0;
        for (int b = 0; b < nb; ++b) pot += bsum[b];
        if (pot_out && blockIdx.x == 0) *pot_out = pot;
        const double rv = u[blockIdx.x] * (double)(float)pot;
        double run = 0.0;
        int ow = -1;
        double opre = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double nxt = run + bsum[b];
            if (ow < 0 && nxt >= rv) { ow = b; opre = run; }
            run = nxt;
        }
        owner = ow;
        owner_pre = opre;
        best = 0x7fffffffffffffffll;
        sh[31] = rv;
    }
    __syncthreads();
    const double rv = sh[31];
    if (owner < 0) {
        if (threadIdx.x == 0) idx_out[blockIdx.x] = n - 1;
        return;
    }
    double pre = owner_pre;
    for (int b = owner; b < nb; ++b) {
        const long long i0 = (long long)b * KPP_TILE + threadIdx.x * 4;
        float pv[4];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pv[q] = (i0 + q < n) ? d2[i0 + q] : 0.f;
            s += (double)pv[q];
        }
        double tot;
        double run = pre + block_scan_excl_1024(s, sh, &tot);
        long long found = 0x7fffffffffffffffll;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            run += (double)pv[q];
            if (found == 0x7fffffffffffffffll && i0 + q < n && run >= rv) found = i0 + q;
        }
        if (found != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&best, (unsigned long long)found);
        __syncthreads();
        if (best != 0x7fffffffffffffffll) break;
        pre += tot;
    }
    if (threadIdx.x == 0) idx_out[blockIdx.x] = (best == 0x7fffffffffffffffll) ? n - 1 : best;
}

extern "C" int scd_kpp_searchsorted(scd_handle h, const float* d2, int64_t n, const double* u, int n_draws, int64_t* idx_out,
                                    double* pot_out, void* ws, size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_searchsorted");
    SCD_REQUIRE(h && d2 && u && idx_out && n > 0 && n_draws > 0 && ws, "scd_kpp_searchsorted: bad arguments");
    SCD_REQUIRE(ws_bytes >= scd_kpp_draw_ws_bytes(n), "scd_kpp_searchsorted: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const int nb = (int)scd_cdiv(n, KPP_TILE);
    double* bsum = (double*)ws;
    kpp_tile_sum_kernel<<<nb, 1024, 0, st>>>(d2, n, bsum);
    kpp_search_kernel<<<n_draws, 1024, 0, st>>>(d2, n, u, bsum, nb, (long long*)idx_out, pot_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// k-means++ for R restarts in lock-step (the restarts of one fit share X and draw from a fixed random stream, so their t-th
// centres can be added together): X is read ONCE per round instead of R times, and a round is 4 launches instead of 4 R.
//   scd_kmeans_min_update_multi   d2[r][i] = min(d2[r][i], ||x_i - c_r||^2), r < R (scd_kmeans_min_update is its R = 1 case, so the
//                                 lock-step and the one-restart-at-a-time seedings see the same bits)
//   scd_kpp_draw_multi            the draw of scd_kpp_draw for R vectors, blockIdx.y = restart
// 256 rows per block; X goes through LDS in 32-column chunks (coalesced 128-byte row pieces in, row stride 36 dwords =
// conflict-free b128 on both sides), the RB centres of the pass as float64 in LDS (uniform address: broadcast reads).
// Thread-private float64 accumulators: no cross-lane reduction, and the sum of a row does not depend on R or RB.
// Work split inside a 256-row block: a thread owns TWO rows (p and p + 128: a centre value read from LDS serves both - at
// RB = 10 the broadcast centre reads, 8 cycles of LDS issue per ds_read_b128 whatever the address pattern, outweigh the float64
// VALU work with one row per thread) and HALF of every chunk's columns (waves 0-1: columns 0-15, waves 2-3: 16-31: N / 64 waves
// as with one row per thread, so the SIMDs stay evenly loaded).  A row's sum is (its low-half columns in ascending order) +
// (its high-half columns in ascending order), whatever R and RB.
// J = rows per thread: 2 (256-row blocks) or 1 (128-row blocks, for inputs whose 256-row blocks would not fill the chip: the same
// sums in the same order, twice the blocks).
constexpr int MU_COLS = 32, MU_LD = 36, MU_ROWS = 256;
template <int RB, bool VEC, int J>
// blockIdx.y = trial l of a greedy k-means++ round (gridDim.y = 1 otherwise): centres Cn + l * lc, output d2 + l * lo.  d2in: the current
// distances when the result goes elsewhere (out = min(d2in, dist): the trials of a round all start from the same d2), null = in place.
__global__ void __launch_bounds__(256) minupd_tile_kernel(const float* __restrict__ X, const float* __restrict__ Cn, long long n, int d,
                                                          int r0_in, int R, float* d2, long long ld, long long ldc, const float* d2in,
                                                          long long lc, long long lo) {
    const int r0 = r0_in + (int)blockIdx.z * RB;       // gridDim.z > 1: the restarts of a group split over blocks (small inputs)
    Cn += (size_t)blockIdx.y * lc;
    d2 += (size_t)blockIdx.y * lo;
    const float* din = d2in ? d2in : d2;
    constexpr int NROW = 128 * J;
    __shared__ __attribute__((aligned(16))) float xs[NROW * MU_LD];          // reused for the high-half partials at the end
    __shared__ __attribute__((aligned(16))) double cs[RB * MU_COLS];
    static_assert(128 * J * RB * 8 <= NROW * MU_LD * 4, "partials must fit the x tile");
    const int t = threadIdx.x, p = t & 127, half = t >> 7;
    const long long row0 = (long long)blockIdx.x * NROW;
    const int nch = (d + MU_COLS - 1) / MU_COLS;
    constexpr int NC = (RB * MU_COLS + 255) / 256;
    float4 pre[4 * J];
    float prs[VEC ? 1 : 16 * J];
    float prc[NC];
    auto fetch = [&](int ch) {
        const int c0 = ch * MU_COLS;
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4 * J; ++i) {
                const long long r = row0 + (t >> 3) + 32 * i;
                const int c = c0 + (t & 7) * 4;
                pre[i] = (r < n && c < d) ? *(const float4*)(X + r * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16 * J; ++i) {
                const long long r = row0 + (t >> 5) + 8 * i;
                const int c = c0 + (t & 31);
                prs[VEC ? 0 : i] = (r < n && c < d) ? X[r * d + c] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int e = t + 256 * i, r = r0 + e / MU_COLS, c = c0 + e % MU_COLS;
            prc[i] = (e < RB * MU_COLS && r < R && c < d) ? Cn[(size_t)r * ldc + c] : 0.f;
        }
    };
    double acc[J][RB];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[j][r] = 0.0;
    fetch(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4 * J; ++i) *(float4*)(xs + ((t >> 3) + 32 * i) * MU_LD + (t & 7) * 4) = pre[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16 * J; ++i) xs[((t >> 5) + 8 * i) * MU_LD + (t & 31)] = prs[VEC ? 0 : i];
        }
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (t + 256 * i < RB * MU_COLS) cs[t + 256 * i] = (double)prc[i];
        __syncthreads();
        if (ch + 1 < nch) fetch(ch + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 16 * half + 4 * q;
            double x[J][4];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float4 xv = *(const float4*)(xs + (p + 128 * j) * MU_LD + col);
                x[j][0] = (double)xv.x; x[j][1] = (double)xv.y; x[j][2] = (double)xv.z; x[j][3] = (double)xv.w;
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const double2 c01 = *(const double2*)(cs + r * MU_COLS + col);
                const double2 c23 = *(const double2*)(cs + r * MU_COLS + col + 2);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    double a = x[j][0] - c01.x; acc[j][r] = fma(a, a, acc[j][r]);
                    a = x[j][1] - c01.y; acc[j][r] = fma(a, a, acc[j][r]);
                    a = x[j][2] - c23.x; acc[j][r] = fma(a, a, acc[j][r]);
                    a = x[j][3] - c23.y; acc[j][r] = fma(a, a, acc[j][r]);
                }
            }
        }
    }
    __syncthreads();
    double* part = (double*)xs;                       // [128][J][RB]
    if (half) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < RB; ++r) part[(p * J + j) * RB + r] = acc[j][r];
    }
    __syncthreads();
    if (!half) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const long long row = row0 + p + 128 * j;
            if (row < n) {
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if (r0 + r < R) {
                        const size_t at = (size_t)(r0 + r) * ld + row;
                        d2[at] = fminf(din[at], (float)(acc[j][r] + part[(p * J + j) * RB + r]));
                    }
            }
        }
    }
}

template <int RB>
static int minupd_launch(const float* X, const float* Cn, long long n, int d, int r0, int R, float* d2, long long ld, hipStream_t st,
                         long long ldc, const float* d2in, int L, long long lc, long long lo, int zsplit = 1) {
    if (scd_cdiv(n, MU_ROWS) * L < 256) {             // fewer 256-row blocks than CUs: 128-row blocks (same bits)
        const dim3 g((unsigned)scd_cdiv(n, 128), (unsigned)L, (unsigned)zsplit);
        if ((d & 3) == 0) minupd_tile_kernel<RB, true, 1><<<g, 256, 0, st>>>(X, Cn, n, d, r0, R, d2, ld, ldc, d2in, lc, lo);
        else minupd_tile_kernel<RB, false, 1><<<g, 256, 0, st>>>(X, Cn, n, d, r0, R, d2, ld, ldc, d2in, lc, lo);
        return SCD_OK;
    }
    const dim3 g((unsigned)scd_cdiv(n, MU_ROWS), (unsigned)L, (unsigned)zsplit);
    if ((d & 3) == 0) minupd_tile_kernel<RB, true, 2><<<g, 256, 0, st>>>(X, Cn, n, d, r0, R, d2, ld, ldc, d2in, lc, lo);
    else minupd_tile_kernel<RB, false, 2><<<g, 256, 0, st>>>(X, Cn, n, d, r0, R, d2, ld, ldc, d2in, lc, lo);
    return SCD_OK;
}
// c_new row r at Cn + r * ldc.  L > 1: the L trials of a greedy round in one launch per restart group - trial l's centres at
// c_new + l * lc, its result d2 + l * lo = min(d2in, distances), d2in shared by the trials (round 6: at 4,500 rows a launch is 18 blocks,
// and seven of them one after the other were 73 % of the CUB-shaped step).
static int minupd_all(const float* X, const float* c_new, long long n, int d, int R, float* d2, long long ld, long long ldc, hipStream_t st,
                      const float* d2in = nullptr, int L = 1, long long lc = 0, long long lo = 0) {
    int r0 = 0;
    // small inputs (one chip round of 128-row blocks at most): ten restarts as two groups of five in the SAME launch (gridDim.z = 2) -
    // twice the blocks, two per CU, so that one block's loads hide behind the other's float64 arithmetic; a (row, restart) sum is the
    // same arithmetic in the same order whatever the group size
    while (R - r0 >= 10 && scd_cdiv(n, 128) * L <= 256) { minupd_launch<5>(X, c_new, n, d, r0, R, d2, ld, st, ldc, d2in, L, lc, lo, 2); r0 += 10; }
    while (R - r0 >= 10) { minupd_launch<10>(X, c_new, n, d, r0, R, d2, ld, st, ldc, d2in, L, lc, lo); r0 += 10; }
    while (R - r0 >= 3) { minupd_launch<4>(X, c_new, n, d, r0, R, d2, ld, st, ldc, d2in, L, lc, lo); r0 += 4; }
    while (R - r0 >= 1) { minupd_launch<1>(X, c_new, n, d, r0, R, d2, ld, st, ldc, d2in, L, lc, lo); r0 += 1; }
    return SCD_OK;
}

extern "C" int scd_kmeans_min_update_multi(scd_handle h, const float* X, const float* c_new, int64_t n, int d, int R,
                                           float* d2_inout, int64_t ld, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_min_update_multi");
    SCD_REQUIRE(h && X && c_new && d2_inout && n > 0 && d > 0 && R > 0 && ld >= n, "scd_kmeans_min_update_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    minupd_all(X, c_new, n, d, R, d2_inout, ld, d, st);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

__global__ void __launch_bounds__(1024) kpp_tile_sum_multi_kernel(const float* __restrict__ d2, long long n, long long ld, double* bsum, int nb) {
    __shared__ double sh[16];
    const float* v = d2 + (size_t)blockIdx.y * ld;
    const long long i0 = (long long)blockIdx.x * KPP_TILE + threadIdx.x * 4;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < n) s += (double)v[i0 + q];
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) bsum[(size_t)blockIdx.y * nb + blockIdx.x] = s;
}
__global__ void __launch_bounds__(1024) kpp_tile_prob_multi_kernel(const float* __restrict__ d2, long long n, long long ld, const double* bsum,
                                                                   int nb, const double* total_in, double* psum) {
    __shared__ double sh[16];
    const int y = blockIdx.y;
    const float totf = kpp_total(bsum + (size_t)y * nb, nb, total_in ? total_in + y : nullptr);
    const float* v = d2 + (size_t)y * ld;
    const long long i0 = (long long)blockIdx.x * KPP_TILE + threadIdx.x * 4;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < n) s += (double)__fdiv_rn(v[i0 + q], totf);
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) psum[(size_t)y * nb + blockIdx.x] = s;
}
// (the pick of kpp_pick_kernel, one block per restart)
__global__ void __launch_bounds__(1024) kpp_pick_multi_kernel(const float* __restrict__ d2, long long n, long long ld, const float* __restrict__ rarr,
                                                              const double* bsum_all, const double* psum_all, int nb, const double* total_in,
                                                              const double* prefix_in, long long* idx_out, double* probsum_out) {
    __shared__ double sh[32];
    __shared__ long long best;
    __shared__ int owner;
    __shared__ double owner_pre;
    const int y = blockIdx.x;
    const float* v = d2 + (size_t)y * ld;
    const double* bsum = bsum_all + (size_t)y * nb;
    const double* psum = psum_all + (size_t)y * nb;
    // the restart's per-tile sums into LDS first: thread 0's serial walks over them below were a chain of ~30 dependent L2 reads
    // (11.5 us per launch at 95,000 rows, the longest of the draw's three kernels)
    __shared__ double s_bs[KPP_STAGE], s_ps[KPP_STAGE];
    if (nb <= KPP_STAGE) {
        for (int b = threadIdx.x; b < nb; b += 1024) { s_bs[b] = bsum[b]; s_ps[b] = psum[b]; }
        __syncthreads();
        bsum = s_bs;
        psum = s_ps;
    }
    const float r = rarr[y];
    const float totf = kpp_total(bsum, nb, total_in ? total_in + y : nullptr);
    if (threadIdx.x == 0) {
        double run = prefix_in ? prefix_in[y] : 0.0;
        int ow = -1;
        double opre = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double nxt = run + psum[b];
            if (ow < 0 && (float)nxt >= r) { ow = b; opre = run; }
            run = nxt;
        }
        owner = ow;
        owner_pre = opre;
        best = 0x7fffffffffffffffll;
        if (probsum_out) probsum_out[y] = run - (prefix_in ? prefix_in[y] : 0.0);
    }
    __syncthreads();
    if (!idx_out) return;
    if (owner < 0) {
        if (threadIdx.x == 0) idx_out[y] = -1;
        return;
    }
    double pre = owner_pre;
    for (int b = owner; b < nb; ++b) {
        const long long i0 = (long long)b * KPP_TILE + threadIdx.x * 4;
        float pv[4];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pv[q] = (i0 + q < n) ? __fdiv_rn(v[i0 + q], totf) : 0.f;
            s += (double)pv[q];
        }
        double tot;
        double run = pre + block_scan_excl_1024(s, sh, &tot);
        long long found = 0x7fffffffffffffffll;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            run += (double)pv[q];
            if (found == 0x7fffffffffffffffll && i0 + q < n && (float)run >= r) found = i0 + q;
        }
        if (found != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&best, (unsigned long long)found);
        __syncthreads();
        if (best != 0x7fffffffffffffffll) break;
        pre += tot;
    }
    if (threadIdx.x == 0) idx_out[y] = (best == 0x7fffffffffffffffll) ? -1 : best;
}

extern "C" int scd_kpp_draw_multi(scd_handle h, const float* d2, int64_t n, int64_t ld, int R, const float* r_dev, const double* total,
                                  const double* prefix, int64_t* idx_out, double* probsum_out, void* ws, size_t ws_bytes,
                                  void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_draw_multi");
    SCD_REQUIRE(h && d2 && r_dev && (idx_out || probsum_out) && n > 0 && ld >= n && R > 0 && ws, "scd_kpp_draw_multi: bad arguments");
    SCD_REQUIRE(ws_bytes >= (size_t)R * scd_kpp_draw_ws_bytes(n), "scd_kpp_draw_multi: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const int nb = (int)scd_cdiv(n, KPP_TILE);
    double* bsum = (double*)ws;
    double* psum = bsum + (size_t)R * nb;
    if (!total) kpp_tile_sum_multi_kernel<<<dim3(nb, R), 1024, 0, st>>>(d2, n, ld, bsum, nb);
    kpp_tile_prob_multi_kernel<<<dim3(nb, R), 1024, 0, st>>>(d2, n, ld, bsum, nb, total, psum);
    kpp_pick_multi_kernel<<<R, 1024, 0, st>>>(d2, n, ld, r_dev, bsum, psum, nb, total, prefix, (long long*)idx_out, probsum_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// per-row float64 sums of R float32 vectors (the shard totals of the lock-step k-means++ under a process group)
__global__ void __launch_bounds__(1024) sum_multi_kernel(const float* __restrict__ x, long long n, long long ld, double* out) {
    __shared__ double sh[32];
    const float* v = x + (size_t)blockIdx.x * ld;
    const long long seg = scd_cdiv_dev(n, 1024);
    const long long a = threadIdx.x * seg, b = (a + seg < n) ? a + seg : n;
    double s = 0.0;
    for (long long i = a; i < b; ++i) s += (double)v[i];
    double tot;
    block_scan_excl_1024(s, sh, &tot);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}
extern "C" int scd_sum_f32_multi(scd_handle h, const float* x, int64_t n, int64_t ld, int R, double* out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_sum_f32_multi");
    SCD_REQUIRE(h && x && out && n > 0 && ld >= n && R > 0, "scd_sum_f32_multi: bad arguments");
    sum_multi_kernel<<<R, 1024, 0, (hipStream_t)stream_>>>(x, n, ld, out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------ the lock-step seeding loop in C
// scd_kpp_seed_lockstep: the rounds of KMeansEngine.kpp_lockstep behind one call - draw, fetch the drawn rows, update the distances -
// so that nothing but these kernels enters the stream (the Python loop added an index_select, a clamp and a strided copy per round).
//
// With the exact fp16 copy of X (scd_f16_exact) the distance update is a FILTER: d2 changes only where the new centre is closer than
// every earlier one - about n / k rows per restart and round - and for the other rows a lower bound of the distance is enough to prove
// it.  muf_filter_kernel evaluates ||x||^2 + ||c||^2 - 2 x.c for 16 rows x up to 16 new centres per 16x16x32 MFMA chain (x and the new
// centres - rows of X themselves - are exact in fp16, their products exact in fp32: only the fp32 accumulation rounds), reads X once
// at 2 bytes per value, and lists the (row, restart) pairs it cannot rule out; muf_exact_kernel gives those the float64 value of
// minupd_tile_kernel's definition (float32(sum_j (x_j - c_j)^2), float64 accumulation) and takes the minimum.  Every float32 that
// is written is such an exactly rounded value, so the d2 arrays - and the draws - are those of the tile kernel.
// The pairs a block cannot rule out go to the block's OWN region of the list (LDS counter; one global counter took ~10 ns per wave
// atomic, 120 us per round at 12,000 pairs), and block b of muf_exact_kernel works region b off.
// Grid = the blocks that are resident at once (the kernel's registers allow 4 / 3 / 2 waves per SIMD at Dp <= 128 / <= 256 / <= 768):
// a larger grid would run its last blocks in a half-empty second round.
static inline int muf_grid(int dp) { return dp <= 128 ? 1024 : dp <= 256 ? 768 : 512; }
__global__ void __launch_bounds__(256) muf_rown2_kernel(const half_t* __restrict__ X16, long long n, int d, float* __restrict__ rn2) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double s = 0.0;
    for (int j = lane; j < d; j += 64) {
        const double v = (double)(float)X16[row * d + j];
        s = fma(v, v, s);
    }
    s = wave_sum_f64(s);
    if (lane == 0) rn2[row] = (float)s;
}
// block r < 16: fetches the drawn row (C_r = X[max(pick[r], 0)]: a negative pick - no row drawn - is reported by the caller after the
// loop) and writes c16[r][0..dp) = fp16(c_r) (zero beyond d, zero rows beyond R); info[r] = {||c16_r||^2, ||c_r - c16_r||}
__global__ void __launch_bounds__(256) muf_prep_kernel(const float* __restrict__ X, const long long* __restrict__ pick, float* __restrict__ Cn,
                                                       long long ldc, int R, int d, int dp, half_t* __restrict__ c16, double* __restrict__ info) {
    __shared__ double red[4][2];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s2 = 0.0, e2 = 0.0;
    const long long src = r < R ? (pick[r] < 0 ? 0 : pick[r]) : 0;
    for (int j = threadIdx.x; j < dp; j += 256) {
        const float c = (r < R && j < d) ? X[src * d + j] : 0.f;
        if (r < R && j < d) Cn[(size_t)r * ldc + j] = c;
        const half_t hc = (half_t)c;
        c16[(size_t)r * dp + j] = hc;
        const double hv = (double)(float)hc, e = (double)c - hv;
        s2 = fma(hv, hv, s2);
        e2 = fma(e, e, e2);
    }
    s2 = wave_sum_f64(s2);
    e2 = wave_sum_f64(e2);
    if (lane == 0) { red[wave][0] = s2; red[wave][1] = e2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[r * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        info[r * 2 + 1] = sqrt((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}
// NKS = dp / 32 k-steps.  A = the centres (m = restart), B = 16 rows of X (n = row): lane l holds row l & 15 and restarts 4 (l >> 4) + j.
// GROUPS = 1: the launch's 16 centres.  GROUPS = 4 (the greedy seeding of kmeans_sk_impl.h, up to 64 candidates per launch): grid
// (g / 4, 4), blockIdx.y = group of 16 centres; the four blocks with one blockIdx.x walk the same row tiles at the same time on the
// same XCD (linear block id mod 8 = blockIdx.x mod 8 when g / 4 is a multiple of 8), so X crosses HBM once for 64 candidates and
// three of four tile loads hit that XCD's L2.  (Four waves of a block sharing one tile - a quarter of the tiles in flight - took
// 93 us per launch against 4 x 33.5 for four launches.)
template <int NKS, int GROUPS = 1>
__global__ void __launch_bounds__(256) muf_filter_kernel(const half_t* __restrict__ X16, const float* __restrict__ rn2, const half_t* __restrict__ c16,
                                                         const double* __restrict__ info, long long n, int d, int R,
                                                         const float* __restrict__ d2, long long ld, unsigned* __restrict__ counts,
                                                         unsigned long long* __restrict__ list, long long cap, int mbase = 0, int L = 1) {
    // mbase, L (the greedy seeding of kmeans_sk_impl.h): the 16 centres of this launch are candidates mbase .. mbase + 15 of R in all,
    // candidate m is measured against row m / L of d2 (L candidates per start); the lock-step seeding has one per restart (0, 1)
    constexpr int DP = NKS * 32;
    __shared__ unsigned lcount;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    unsigned long long* mine = list + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * cap;
    const int lane = threadIdx.x & 63, c16i = lane & 15, q = lane >> 4;
    if (GROUPS == 4) {
        const int grp = blockIdx.y;
        c16 += (size_t)grp * 16 * DP;
        info += grp * 32;
        mbase += grp * 16;
    }
    half8 ca[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) ca[ks] = *(const half8*)(c16 + (size_t)c16i * DP + ks * 32 + 8 * q);
    double cn2[4], cdl[4], cnr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { cn2[j] = info[(4 * q + j) * 2]; cdl[j] = info[(4 * q + j) * 2 + 1]; cnr[j] = sqrt(cn2[j]); }
    int drow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) drow[j] = (mbase + 4 * q + j) / L;
    // fp32 accumulation of DP exact products: |acc - x.c| <= DP * 2^-24 * sum |x_j c_j| <= DP * 2^-24 ||x|| ||c||; x1.5 safety
    const double gam = 1.5 * DP * 5.9604644775390625e-8;
    const long long ntile = (n + 15) >> 4;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
    for (long long tile = wave_id; tile < ntile; tile += nwave) {
        const long long row = tile * 16 + c16i;
        const long long rowc = row < n ? row : n - 1;
        const half_t* xr = X16 + rowc * d + 8 * q;
        half8 xb[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks * 32 + 8 * q < d) xb[ks] = *(const half8*)(xr + ks * 32);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[ks][e] = (half_t)0.f;
            }
        }
        const double xn2 = (double)rn2[rowc];
        float old[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) old[j] = (mbase + 4 * q + j < R) ? d2[(size_t)drow[j] * ld + rowc] : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca[ks], xb[ks], acc, 0, 0, 0);
        const double xnr = sqrt(xn2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mbase + 4 * q + j;
            // lower bound of ||x - c||^2: the fp32 dot product's error, the float32 rounding of ||x||^2 (2^-24 relative) and, for a
            // centre that is not exact in fp16, |x.(c - c16)| and the change of ||c||^2
            const double a = xn2 + cn2[j] - 2.0 * (double)acc[j];
            const double E = 2.0 * gam * xnr * cnr[j] + 1.2e-7 * xn2 + 2.0 * (xnr + cnr[j]) * cdl[j] + cdl[j] * cdl[j];
            const bool need = m < R && row < n && !(a - E > (double)old[j]);      // NaN anywhere: not ruled out
            const unsigned long long mask = __ballot(need);
            if (mask) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&lcount, (unsigned)__popcll(mask));
                base = __shfl(base, 0, 64);
                if (need) mine[base + __popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned long long)row << 8) | (unsigned)m;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.y * gridDim.x + blockIdx.x] = lcount;
}
// the listed (row, restart) pairs: d2 = min(d2, float32(sum_j (x_j - c_j)^2)), float64 accumulation, x from the exact fp16 copy.
// Sixteen lanes per pair, four pairs per wave at a time, and every load of a pair - its list entry's row, the centre, the old d2 -
// is issued before the first is used: a pair is one chain of memory latencies (one pair per wave took 6.6 us per pair).
__global__ void __launch_bounds__(256) muf_exact_kernel(const half_t* __restrict__ X16, const float* __restrict__ Cn, long long ldc, int d,
                                                        const unsigned* __restrict__ counts, const unsigned long long* __restrict__ list,
                                                        long long cap, float* __restrict__ d2, long long ld) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    const unsigned cnt = counts[blockIdx.x];
    const unsigned long long* mine = list + (size_t)blockIdx.x * cap;
    for (unsigned p0 = (threadIdx.x >> 6) * 4; p0 < cnt; p0 += 16) {
        const unsigned p = p0 + sub;
        const bool on = p < cnt;
        const unsigned long long e = mine[on ? p : p0];
        const long long row = (long long)(e >> 8);
        const int m = (int)(e & 255);
        const half_t* x = X16 + row * d;
        const float* c = Cn + (size_t)m * ldc;
        float* qd = d2 + (size_t)m * ld + row;
        const float old = *qd;
        double s = 0.0;
        for (int j0 = l16 * 8; j0 < d; j0 += 512) {
            half8 xv[4];
            float4 cv[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 128 * u;
                if (j < d) {
                    xv[u] = *(const half8*)(x + j);
                    cv[u][0] = *(const float4*)(c + j);
                    cv[u][1] = *(const float4*)(c + j + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 128 * u;
                if (j < d) {
                    const float cc[8] = {cv[u][0].x, cv[u][0].y, cv[u][0].z, cv[u][0].w, cv[u][1].x, cv[u][1].y, cv[u][1].z, cv[u][1].w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const double t = (double)(float)xv[u][q] - (double)cc[q];
                        s = fma(t, t, s);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (on && l16 == 0) *qd = fminf(old, (float)s);
    }
}
// C[r][slot] = X[max(pick[r], 0)]  (a negative pick - no row drawn - is reported by the caller after the loop)
__global__ void __launch_bounds__(256) kpp_fetch_rows_kernel(const float* __restrict__ X, const long long* __restrict__ pick, int d,
                                                             float* __restrict__ Cslot, long long ldc) {
    const long long i = pick[blockIdx.x] < 0 ? 0 : pick[blockIdx.x];
    for (int j = threadIdx.x; j < d; j += 256) Cslot[(size_t)blockIdx.x * ldc + j] = X[i * d + j];
}

static inline int muf_dp(int d) { return (d + 31) / 32 * 32; }
// list capacity per block: the rows of the tiles its four waves can be dealt (16 rows x 16 restarts per tile)
static inline long long muf_cap(int64_t n, int g) { return 4 * scd_cdiv(scd_cdiv(n, 16), 4 * g) * 256; }
extern "C" size_t scd_kpp_seed_ws_bytes(int64_t n, int d, int R) {
    const int g = muf_grid(muf_dp(d));
    return (size_t)R * scd_kpp_draw_ws_bytes(n) + scd_align(4 * (size_t)n) + scd_align(2 * 16 * (size_t)muf_dp(d)) + 256 + scd_align(4 * 1024) +
           scd_align(8 * (size_t)muf_cap(n, g) * g) + 256;
}
extern "C" int scd_kpp_seed_lockstep(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, float* d2, int64_t ld,
                                     const float* r_dev, int T, float* C_buf, int k, int m0, int64_t* picks_out, void* ws,
                                     size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_seed_lockstep");
    SCD_REQUIRE(X && d2 && r_dev && C_buf && picks_out && ws && n > 0 && d > 0 && R > 0 && ld >= n && T >= 0 && m0 >= 1 && m0 + T <= k,
                "scd_kpp_seed_lockstep: bad arguments");
    SCD_REQUIRE(ws_bytes >= scd_kpp_seed_ws_bytes(n, d, R), "scd_kpp_seed_lockstep: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const size_t draw_nb = (size_t)R * scd_kpp_draw_ws_bytes(n);
    const int dp = muf_dp(d);
    char* w = (char*)ws;
    void* draw_ws = w;
    float* rn2 = (float*)(w + draw_nb);
    half_t* c16 = (half_t*)((char*)rn2 + scd_align(4 * (size_t)n));
    double* info = (double*)((char*)c16 + scd_align(2 * 16 * (size_t)dp));
    unsigned* counts = (unsigned*)((char*)info + 256);
    unsigned long long* list = (unsigned long long*)((char*)counts + scd_align(4 * 1024));
    const int g = muf_grid(dp);
    const long long cap = muf_cap(n, g);
    const long long ldc = (long long)k * d;
    const int filt_env = getenv("SCD_KPP_FILTER") ? atoi(getenv("SCD_KPP_FILTER")) : 1;     // 0: the tile kernel reads the float32 rows (A/B)
    const bool filt = X16 && filt_env && R <= 16 && d % 32 == 0 && (dp == 128 || dp == 256 || dp == 384 || dp == 512 || dp == 768) &&
                      n < (1ll << 40) && T > 1;
    if (filt) muf_rown2_kernel<<<(unsigned)scd_cdiv(n, 4), 256, 0, st>>>((const half_t*)X16, n, d, rn2);
    for (int t = 0; t < T; ++t) {
        int64_t* pick = picks_out + (size_t)t * R;
        const int rc = scd_kpp_draw_multi(h, d2, n, ld, R, r_dev + (size_t)t * R, nullptr, nullptr, pick, nullptr, draw_ws, draw_nb, stream_);
        if (rc) return rc;
        float* slot = C_buf + (size_t)(m0 + t) * d;
        const bool use_filter = filt && m0 + t >= 8 && t + 1 < T;
        if (!use_filter) kpp_fetch_rows_kernel<<<R, 256, 0, st>>>(X, (const long long*)pick, d, slot, ldc);
        if (t + 1 == T) break;
        // with m centres a new one is the nearest for ~1 / (m + 1) of the rows: the filter pays once that is a small share (the float64
        // pass costs ~1 us per pair and wave); the first rounds update most rows and stay on the tile kernel
        if (!filt || m0 + t < 8) {
            minupd_all(X, slot, n, d, R, d2, ld, ldc, st);
            continue;
        }
        muf_prep_kernel<<<16, 256, 0, st>>>(X, (const long long*)pick, slot, ldc, R, d, dp, c16, info);     // fetch + operands in one launch
#define MUF_GO(NKS) muf_filter_kernel<NKS><<<g, 256, 0, st>>>((const half_t*)X16, rn2, c16, info, n, d, R, d2, ld, counts, list, cap)
        switch (dp / 32) {
            case 4: MUF_GO(4); break;
            case 8: MUF_GO(8); break;
            case 12: MUF_GO(12); break;
            case 16: MUF_GO(16); break;
            default: MUF_GO(24); break;
        }
#undef MUF_GO
        muf_exact_kernel<<<g, 256, 0, st>>>((const half_t*)X16, slot, ldc, d, counts, list, cap, d2, ld);
    }
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// One Lloyd iteration of the (unconstrained) semi-supervised K-Means behind ONE call (faster_mix_k_means_pytorch.py:187-214):
// E-step of the unlabelled rows against C_in (labels written behind the labelled rows' fixed ones), M-step partial sums + inertia
// over [labelled ; unlabelled], new centres + shift, and the next E-step's centre operands.  Nothing here is new arithmetic - it is
// scd_kmeans_estep + scd_kmeans_mstep[_f16] + scd_kmeans_finalize with the buffers wired together - but a Lloyd iteration is
// ~105 us of device work and the caller's per-call overhead (a Python caller: 15 tensor / FFI calls, ~185 us) otherwise bounds it.
// stats[3] = {inertia of the labelled rows, inertia of the unlabelled rows, centre shift}, float64 on the device.
// ------------------------------------------------------------------------------------------------
// Incremental Lloyd step (round 3).  When every value of X is exactly representable in fp16 (features that left an fp16 encoder:
// scd_f16_exact), the float64 per-cluster sums are EXACT - every value is a multiple of 2^-24 below 2^16 and a cluster of N <= 2^13
// ... 2^29 such rows sums below 2^53 units for unit-scale features - so they do not depend on the order of additions, and the sums of
// iteration i are the sums of iteration i - 1 plus the rows whose label changed (added to the new cluster, subtracted from the old):
// bit-identical to a fresh M-step at a cost proportional to the changes (a few hundred rows of 95,000 after the third iteration
// on clustered features) instead of a pass over X.  The inertia of the reference's bookkeeping (sum over rows of ||x - c_label||^2
// with the centres the labels were computed from, sskm.py:118-132) follows without X as well:
//     sum_i ||x_i - c_l(i)||^2 = sum_i ||x_i||^2 + sum_k ( n_k ||c_k||^2 - 2 <c_k, S_k> )
// evaluated in double-double arithmetic (error-free products by fma, two-sum accumulation) from the exact S_k, n_k, the float32
// centres and sum ||x||^2 (once per fit, also double-double), separately for the labelled and the unlabelled rows - the value agrees
// with the float64 row-by-row sum to ~1e-15 relative, like two summation orders of that sum.
// sum of squares of the rows [0, split) and [split, n), double-double, two launches: per-block partials, then one block.
// The matrix is walked as a flat array, eight consecutive values per lane and load (a first version walked it a row per wave with
// 2-byte loads and one row in flight: 327 us for 146 MB, latency-bound); elements below e0 = split * d go to the first sum.  The
// partition is fixed by the launch shape, so the result is reproducible; any two partitions agree to ~1e-30 relative.
#define SUMSQ_BLOCKS 1024
__global__ void __launch_bounds__(256) sumsq_dd_kernel(const half_t* __restrict__ X16, const float* __restrict__ X, long long e0, long long e1,
                                                       double* part) {
    __shared__ double red[4][4];
    dd_t a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < e1; i += (long long)gridDim.x * 256 * 8) {
        float v[8];
        if (i + 8 <= e1) {
            if (X16) {
                const half8 h = *(const half8*)(X16 + i);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
            } else {
                const float4 p = *(const float4*)(X + i), q = *(const float4*)(X + i + 4);
                v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = i + e < e1 ? (X16 ? (float)X16[i + e] : X[i + e]) : 0.f;
        }
        if (i + 8 <= e0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a0 = dd_add_prod(a0, (double)v[e], (double)v[e]);
        } else if (i >= e0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a1 = dd_add_prod(a1, (double)v[e], (double)v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (i + e < e0) a0 = dd_add_prod(a0, (double)v[e], (double)v[e]);
                else a1 = dd_add_prod(a1, (double)v[e], (double)v[e]);
            }
        }
    }
    a0 = dd_wave_sum(a0);
    a1 = dd_wave_sum(a1);
    if (lane == 0) { red[wave][0] = a0.hi; red[wave][1] = a0.lo; red[wave][2] = a1.hi; red[wave][3] = a1.lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dd_t b0 = {0.0, 0.0}, b1 = {0.0, 0.0};
        for (int w = 0; w < 4; ++w) { b0 = dd_add(b0, {red[w][0], red[w][1]}); b1 = dd_add(b1, {red[w][2], red[w][3]}); }
        part[blockIdx.x * 4 + 0] = b0.hi; part[blockIdx.x * 4 + 1] = b0.lo;
        part[blockIdx.x * 4 + 2] = b1.hi; part[blockIdx.x * 4 + 3] = b1.lo;
    }
}
// one block: thread t adds the partials t, t + 256, ..., then a fixed tree over the 256 threads
__global__ void __launch_bounds__(256) sumsq_dd_final_kernel(const double* part, int nblk, double* out) {
    __shared__ double tr[256][4];
    dd_t b0 = {0.0, 0.0}, b1 = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256) { b0 = dd_add(b0, {part[b * 4], part[b * 4 + 1]}); b1 = dd_add(b1, {part[b * 4 + 2], part[b * 4 + 3]}); }
    tr[threadIdx.x][0] = b0.hi; tr[threadIdx.x][1] = b0.lo; tr[threadIdx.x][2] = b1.hi; tr[threadIdx.x][3] = b1.lo;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const dd_t x = dd_add({tr[threadIdx.x][0], tr[threadIdx.x][1]}, {tr[threadIdx.x + o][0], tr[threadIdx.x + o][1]});
            const dd_t y = dd_add({tr[threadIdx.x][2], tr[threadIdx.x][3]}, {tr[threadIdx.x + o][2], tr[threadIdx.x + o][3]});
            tr[threadIdx.x][0] = x.hi; tr[threadIdx.x][1] = x.lo; tr[threadIdx.x][2] = y.hi; tr[threadIdx.x][3] = y.lo;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = tr[0][0]; out[1] = tr[0][1]; out[2] = tr[0][2]; out[3] = tr[0][3]; }
}
extern "C" int scd_kmeans_sumsq(scd_handle h, const void* X16, const float* X, int64_t n, int d, int64_t split, double* out4,
                                void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_sumsq");
    SCD_REQUIRE((X16 || X) && out4 && n > 0 && d > 0 && split >= 0 && split <= n, "scd_kmeans_sumsq: bad arguments");
    // the rows are walked as ONE flat array in 8-value steps with 16-byte loads (half8 / float4 pairs; the split boundary and the tail are
    // handled value by value): the base pointer must be 16-byte aligned - a whole row set from an allocator is, a row-offset pointer with
    // an odd row length need not be
    SCD_REQUIRE(((uintptr_t)(X16 ? X16 : (const void*)X) & 15) == 0, "scd_kmeans_sumsq: the rows' base pointer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream_;
    double* part = (double*)h->scratch;                          // 1,024 blocks x 4 doubles of the handle's 256-KB scratch
    sumsq_dd_kernel<<<SUMSQ_BLOCKS, 256, 0, st>>>((const half_t*)X16, X, (long long)split * d, (long long)n * d, part);
    sumsq_dd_final_kernel<<<1, 256, 0, st>>>(part, SUMSQ_BLOCKS, out4);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// rows [row0, n) whose label differs from labels_prev: +x to the new cluster's sums, -x from the old one's (float64 atomics of exact
// values: order-free), counts, labels_prev updated, the number of changes added to *changed (double).  One wave per 64 rows.
__global__ void __launch_bounds__(256) mstep_delta_kernel(const half_t* __restrict__ X16, const int32_t* __restrict__ labels,
                                                          int32_t* __restrict__ labels_prev, long long row0, long long n, int d, int k,
                                                          double* __restrict__ sums, unsigned long long* __restrict__ counts,
                                                          double* __restrict__ changed) {
    const int lane = threadIdx.x & 63;
    const long long base = row0 + ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (base >= n) return;
    const long long row = base + lane;
    int ln = -1, lo = -1;
    if (row < n) { ln = labels[row]; lo = labels_prev[row]; }
    const bool ch = row < n && ln != lo;
    unsigned long long m = __ballot(ch);
    if (!m) return;
    if (ch) labels_prev[row] = ln;
    if (lane == 0) atomicAdd(changed, (double)__popcll(m));
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int a = __shfl(ln, src, 64), b = __shfl(lo, src, 64);
        const half_t* xr = X16 + (base + src) * d;
        for (int c = 2 * lane; c < d; c += 128) {
            const double x0 = (double)(float)xr[c], x1 = (double)(float)xr[c + 1];
            if ((unsigned)a < (unsigned)k) { atomicAdd(&sums[(size_t)a * d + c], x0); atomicAdd(&sums[(size_t)a * d + c + 1], x1); }
            if ((unsigned)b < (unsigned)k) { atomicAdd(&sums[(size_t)b * d + c], -x0); atomicAdd(&sums[(size_t)b * d + c + 1], -x1); }
        }
        if (lane == 0) {
            if ((unsigned)a < (unsigned)k) atomicAdd(&counts[a], 1ull);
            if ((unsigned)b < (unsigned)k) atomicAdd(&counts[b], ~0ull);      // - 1
        }
    }
}
// labels_prev = labels for rows [row0, n) and the number of differences (the full step's bookkeeping for the next incremental one)
__global__ void __launch_bounds__(256) labels_sync_kernel(const int32_t* __restrict__ labels, int32_t* __restrict__ labels_prev, long long row0,
                                                          long long n, double* __restrict__ changed) {
    const long long i = row0 + (long long)blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (i < n) {
        const int a = labels[i];
        ch = a != labels_prev[i];
        if (ch) labels_prev[i] = a;
    }
    // one atomic per block: with every row changed (iteration 0) 1,500 wave atomics on the one address took 15 us
    const int cnt = __syncthreads_count(ch);
    if (threadIdx.x == 0 && cnt) atomicAdd(changed, (double)cnt);
}
// inertia of both row groups from the exact sums: block k -> n_k ||c_k||^2 - 2 <c_k, S_k> for the labelled rows (S_lab, n_lab) and for
// the others (S - S_lab, n - n_lab); the last block adds the K partials in index order and the sums of squares
__global__ void __launch_bounds__(256) inertia_dd_kernel(const float* __restrict__ C, const double* __restrict__ sums,
                                                         const long long* __restrict__ counts, const double* __restrict__ sums_lab,
                                                         const long long* __restrict__ counts_lab, const double* __restrict__ sumsq4, int k,
                                                         int d, double* part, unsigned* ticket, double* out2) {
    __shared__ double red[4][6];
    __shared__ bool last;
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nl = counts_lab ? counts_lab[c] : 0, nu = counts[c] - nl;
    dd_t cc = {0.0, 0.0}, dl = {0.0, 0.0}, du = {0.0, 0.0};
    for (int j = threadIdx.x; j < d; j += 256) {
        const double cv = (double)C[(size_t)c * d + j];
        const double sl = sums_lab ? sums_lab[(size_t)c * d + j] : 0.0;
        const double su = sums[(size_t)c * d + j] - sl;          // exact: both are exact sums of multiples of 2^-24
        cc = dd_add_prod(cc, cv, cv);
        if (nl) dl = dd_add_prod(dl, cv, sl);
        if (nu) du = dd_add_prod(du, cv, su);
    }
    cc = dd_wave_sum(cc); dl = dd_wave_sum(dl); du = dd_wave_sum(du);
    if (lane == 0) { red[wave][0] = cc.hi; red[wave][1] = cc.lo; red[wave][2] = dl.hi; red[wave][3] = dl.lo; red[wave][4] = du.hi; red[wave][5] = du.lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dd_t n2 = {0.0, 0.0}, pl = {0.0, 0.0}, pu = {0.0, 0.0};
        for (int w = 0; w < 4; ++w) { n2 = dd_add(n2, {red[w][0], red[w][1]}); pl = dd_add(pl, {red[w][2], red[w][3]}); pu = dd_add(pu, {red[w][4], red[w][5]}); }
        // t = n ||c||^2 - 2 <c, S>   (n < 2^31: the products with hi / lo are formed error-free)
        dd_t tl = {0.0, 0.0}, tu = {0.0, 0.0};
        if (nl) { tl = dd_add_prod(dd_add_prod(tl, (double)nl, n2.hi), (double)nl, n2.lo); tl = dd_add_d(dd_add_d(tl, -2.0 * pl.hi), -2.0 * pl.lo); }
        if (nu) { tu = dd_add_prod(dd_add_prod(tu, (double)nu, n2.hi), (double)nu, n2.lo); tu = dd_add_d(dd_add_d(tu, -2.0 * pu.hi), -2.0 * pu.lo); }
        part[c * 4 + 0] = tl.hi; part[c * 4 + 1] = tl.lo; part[c * 4 + 2] = tu.hi; part[c * 4 + 3] = tu.lo;
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned)k - 1;
    }
    __syncthreads();
    if (last) {                                     // the K partials in a fixed tree order (a serial loop on one thread took 50 us)
        __shared__ double tr[256][4];
        __threadfence();
        dd_t il = {0.0, 0.0}, iu = {0.0, 0.0};
        for (int q = threadIdx.x; q < k; q += 256) {
            il = dd_add(il, {((volatile double*)part)[q * 4], ((volatile double*)part)[q * 4 + 1]});
            iu = dd_add(iu, {((volatile double*)part)[q * 4 + 2], ((volatile double*)part)[q * 4 + 3]});
        }
        tr[threadIdx.x][0] = il.hi; tr[threadIdx.x][1] = il.lo; tr[threadIdx.x][2] = iu.hi; tr[threadIdx.x][3] = iu.lo;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const dd_t a = dd_add({tr[threadIdx.x][0], tr[threadIdx.x][1]}, {tr[threadIdx.x + o][0], tr[threadIdx.x + o][1]});
                const dd_t b = dd_add({tr[threadIdx.x][2], tr[threadIdx.x][3]}, {tr[threadIdx.x + o][2], tr[threadIdx.x + o][3]});
                tr[threadIdx.x][0] = a.hi; tr[threadIdx.x][1] = a.lo; tr[threadIdx.x][2] = b.hi; tr[threadIdx.x][3] = b.lo;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const dd_t fl = dd_add({sumsq4[0], sumsq4[1]}, {tr[0][0], tr[0][1]}), fu = dd_add({sumsq4[2], sumsq4[3]}, {tr[0][2], tr[0][3]});
            out2[0] = fl.hi + fl.lo;
            out2[1] = fu.hi + fu.lo;
            *ticket = 0;
        }
    }
}

// Exchange hook of the sharded Lloyd loop (scd_kmeans_lloyd_run_sharded): every rank's [k*d sums | k counts as float64] are summed in
// place by `fn` (an all-reduce the caller owns); the k int64 counts the finalize launch reads are rebuilt behind them.  The rank's
// own sums / counts stay untouched - the incremental M-step keeps updating them.
struct LloydXch { double* buf; scd_exchange_fn fn; void* ctx; };
__global__ void __launch_bounds__(256) xch_pack_kernel(const double* sums, const long long* counts, size_t kd, int k, double* buf) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < kd) buf[i] = sums[i];
    else if (i < kd + (size_t)k) buf[i] = (double)counts[i - kd];
}
__global__ void __launch_bounds__(256) xch_unpack_kernel(const double* counts_f64, int k, long long* counts_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k) counts_out[i] = llrint(counts_f64[i]);
}

// One Lloyd iteration in two halves, so that the restarts of a fit can advance in lock-step (scd_kmeans_lloyd_run_multi): A = E-step +
// M-step (+ the rank's [sums | counts] packed at `pack_dst` under a process group), B = centres / shift / inertia / the next E-step's
// operands from `fsums` / `fcounts` (the rank's own, or the exchanged ones).
static int lloyd_step_a(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat, int64_t n_cat, int d, int k,
                        int32_t* labels_cat, int32_t* labels_prev, const float* C_in, double* sums, int64_t* counts, double* stats,
                        int flags, void* ws_e, size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream, double* pack_dst,
                        bool estep_done = false) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_lloyd_step_delta");
    const int64_t l_num = n_cat - n_u;
    hipStream_t st = (hipStream_t)stream;
    int rc = SCD_OK;
    if (!estep_done) {          // (else: the filter of all restarts ran as one launch, estep_multi_launch, and this restart's refine follows it)
        rc = scd_kmeans_estep_hint(h, flags & (SCD_ESTEP_FEW | SCD_ESTEP_CENTRES_FROM_FINALIZE));
        if (!rc) rc = scd_kmeans_estep(h, X_u, prep_u, C_in, n_u, d, k, labels_cat + l_num, nullptr, ws_e, ws_e_bytes, stream);
        if (rc) return rc;
    }
    // rows whose label changed: accumulated in the handle's scratch (zero between iterations), handed to stats[4] by finalize_kernel
    double* changed_acc = (double*)((char*)h->scratch + 262144 + 40);
    if (flags & SCD_LLOYD_FULL) {
        // a fresh M-step (sums, counts, inertia from the rows), then labels_prev = labels for the incremental steps that follow
        rc = scd_kmeans_mstep_f16(h, X16_cat, labels_cat, C_in, n_cat, d, k, l_num, sums, counts, stats, ws_m, ws_m_bytes, stream);
        if (rc) return rc;
        labels_sync_kernel<<<(unsigned)scd_cdiv(n_u, 256), 256, 0, st>>>(labels_cat, labels_prev, l_num, n_cat, changed_acc);
    } else {
        mstep_delta_kernel<<<(unsigned)scd_cdiv(n_u, 256), 256, 0, st>>>((const half_t*)X16_cat, labels_cat, labels_prev, l_num, n_cat, d, k, sums,
                                                                          (unsigned long long*)counts, changed_acc);
    }
    if (pack_dst) {
        const size_t kd = (size_t)k * d;
        xch_pack_kernel<<<(unsigned)scd_cdiv((int64_t)(kd + k), 256), 256, 0, st>>>(sums, (const long long*)counts, kd, k, pack_dst);
    }
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
static int lloyd_step_b(scd_handle h, const void* prep_u, int64_t n_u, int d, int k, const float* C_in, float* C_out, const double* fsums,
                        const int64_t* fcounts, const double* sums_lab, const int64_t* counts_lab, const double* sumsq4, double* stats,
                        int flags, void* ws_e, size_t ws_e_bytes, void* stream, double* mirror, double seq, int shift_mode, bool exchanged) {
    hipStream_t st = (hipStream_t)stream;
    bool fused_inertia = false;
    // sharded: the inertia always comes from the (global) sums - a rank's row-wise partial of a fresh M-step could not be mixed with
    // the sums-based form another rank's incremental step needs, and each rank picks fresh / incremental from its own change count
    const bool sums_inertia = exchanged || !(flags & SCD_LLOYD_FULL);
    if (sums_inertia) {
        // the inertia from the sums: inside finalize_kernel when its 4 k partials fit behind the k shift partials in the handle's scratch
        // (32,768 doubles), else in a launch of its own (partials at the start of the scratch, free between two finalize launches)
        fused_inertia = 5 * (long long)k <= 32768;
        if (!fused_inertia)
            inertia_dd_kernel<<<k, 256, 0, st>>>(C_in, fsums, (const long long*)fcounts, sums_lab, (const long long*)counts_lab, sumsq4, k, d,
                                                 (double*)h->scratch, (unsigned*)((char*)h->scratch + 262144 + 32), stats);
    }
    SCD_LAUNCH_CHECK();
    return finalize_impl(h, fsums, fcounts, k, d, C_in, C_out, stats + 2, shift_mode, prep_u, ws_e, ws_e_bytes, n_u, stream, stats + 3, stats + 4,
                         stats, mirror, seq, sums_lab, counts_lab, sumsq4, fused_inertia ? stats : nullptr);
}

static int lloyd_step_delta_impl(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat,
                                 int64_t n_cat, int d, int k, int32_t* labels_cat, int32_t* labels_prev, const float* C_in,
                                 float* C_out, double* sums, int64_t* counts, const double* sums_lab,
                                 const int64_t* counts_lab, const double* sumsq4, double* stats, int flags, void* ws_e,
                                 size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream, double* mirror, double seq,
                                 int shift_mode = 0, const LloydXch* xch = nullptr) {
    SCD_REQUIRE(X_u && prep_u && X16_cat && labels_cat && labels_prev && C_in && C_out && sums && counts && sumsq4 && stats && ws_e && ws_m,
                "scd_kmeans_lloyd_step_delta: null argument");
    SCD_REQUIRE(n_u > 0 && n_cat >= n_u && C_in != C_out && k <= 8192, "scd_kmeans_lloyd_step_delta: bad arguments");
    int rc = lloyd_step_a(h, X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_cat, labels_prev, C_in, sums, counts, stats, flags, ws_e, ws_e_bytes,
                          ws_m, ws_m_bytes, stream, xch ? xch->buf : nullptr);
    if (rc) return rc;
    const double* fsums = sums;
    const int64_t* fcounts = counts;
    if (xch) {
        const size_t kd = (size_t)k * d;
        rc = xch->fn(xch->ctx, xch->buf, (int64_t)(kd + k), stream);
        if (rc) {
            scd_set_error("scd_kmeans_lloyd_run_sharded: the exchange callback failed (%d)", rc);
            return SCD_ERCCL;
        }
        xch_unpack_kernel<<<(unsigned)scd_cdiv(k, 256), 256, 0, (hipStream_t)stream>>>(xch->buf + kd, k, (long long*)(xch->buf + kd + k));
        fsums = xch->buf;
        fcounts = (const int64_t*)(xch->buf + kd + k);
    }
    return lloyd_step_b(h, prep_u, n_u, d, k, C_in, C_out, fsums, fcounts, sums_lab, counts_lab, sumsq4, stats, flags, ws_e, ws_e_bytes, stream,
                        mirror, seq, shift_mode, xch != nullptr);
}

extern "C" int scd_kmeans_lloyd_step_delta(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat,
                                           int64_t n_cat, int d, int k, int32_t* labels_cat, int32_t* labels_prev, const float* C_in,
                                           float* C_out, double* sums, int64_t* counts, const double* sums_lab,
                                           const int64_t* counts_lab, const double* sumsq4, double* stats, int flags, void* ws_e,
                                           size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream) {
    return lloyd_step_delta_impl(h, X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_cat, labels_prev, C_in, C_out, sums, counts, sums_lab,
                                 counts_lab, sumsq4, stats, flags, ws_e, ws_e_bytes, ws_m, ws_m_bytes, stream, nullptr, 0.0);
}

// The filter launch of scd_kmeans_estep for SEVERAL restarts at once (estep_rbm_kernel; Dp = 512, Kp <= 256 per restart): the centre
// operands of restart j live in its own E-step workspace ws[j] (written by its finalize launch, or by prep_centers_kernel here when the
// hand-over does not hold - the same test as scd_kmeans_estep's), the workspaces and the label slots are equally strided.  Per restart the
// refine launch follows on the restart's own stream.  Returns SCD_OK, or a status; *served = false when the shape is not served.
struct EstepMultiItem { scd_handle h; const float* C; void* ws; int32_t* labels; int slot; };
static bool estep_multi_serves(int d, int k, int n_items) {
    const char* ev = getenv("SCD_ESTEP_MERGED");          // read per fit (a test switches it inside one process)
    const int en = ev ? atoi(ev) : 0;          // opt-in: at C2 the per-restart filters over four streams are faster (9.7-9.9 against 10.2-10.3 ms per stage, round 5)
    return en && dpad(d) == 512 && kpad(k) <= 256 && n_items >= 2 && n_items <= 16;
}
static int estep_multi_launch(const EstepMultiItem* it, int n_items, const float* X, const void* prep, int64_t n, int d, int k, char* ws0,
                              size_t ws_stride, int32_t* lab0, size_t lab_stride, const float* C0, size_t c_stride, bool vouch, int n_cu,
                              hipStream_t st) {
    const int dp = dpad(d), kp = kpad(k);
    const char* p = (const char*)prep;
    const PrepHdr* ph = (const PrepHdr*)p;
    const size_t xnorm_off = scd_align(64 + 8 * (size_t)dp);
    const size_t xh_off = xnorm_off + scd_align(4 * (size_t)n);
    RbmArgs a;
    a.ws0 = ws0; a.ws_stride = ws_stride;
    a.cn_off = 64;
    a.ch_off = 64 + scd_align(4 * (size_t)kp);
    const size_t ct_off = a.ch_off + scd_align(2 * (size_t)kp * dp);
    a.flags_off = ct_off + scd_align(4 * (size_t)kp * dp);
    a.fcand_off = a.flags_off + scd_align(4 * (size_t)n);
    a.fulls_off = a.fcand_off + scd_align(4 * (size_t)n);
    const size_t chf_off = a.fulls_off + scd_align(4 * (size_t)n);
    a.lab0 = lab0; a.lab_stride = lab_stride;
    a.segmap = 0ull;
    a.n_active = n_items; a.kp = kp;
    for (int j = 0; j < n_items; ++j) {
        scd_handle h = it[j].h;
        char* w = (char*)it[j].ws;
        SCD_REQUIRE(w == ws0 + (size_t)it[j].slot * ws_stride && it[j].labels == lab0 + (size_t)it[j].slot * lab_stride && it[j].slot < 16,
                    "estep_multi_launch: workspaces / label slots are not equally strided");
        a.segmap |= (unsigned long long)it[j].slot << (4 * j);
        SCD_REQUIRE(!C0 || it[j].C == C0 + (size_t)it[j].slot * c_stride, "estep_multi_launch: centres are not equally strided");
        // the finalize hand-over of this restart's handle, consumed exactly as scd_kmeans_estep consumes it (`vouch` = the caller's
        // SCD_ESTEP_CENTRES_FROM_FINALIZE: these centres ARE the previous step's output)
        const bool handover = vouch && h->prep_C == it[j].C && h->prep_ws == it[j].ws && h->prep_k == k && h->prep_d == d;
        h->estep_few = 0;
        h->prep_ok = 0;
        h->prep_C = nullptr;
        if (!handover)
            prep_centers_kernel<<<kp, 256, 0, st>>>(it[j].C, k, d, dp, ph, (const double*)(p + 64), (EHdr*)w, (float*)(w + a.cn_off), (half_t*)(w + a.ch_off),
                                                    (float*)(w + ct_off), kp, 1, (half_t*)(w + chf_off));
    }
    // parts: whole restarts per part (no merge), at most 8 restarts and 1,920 centres each; the count that minimises rounds x (units + ramp)
    const long long nblk = scd_cdiv(n, 256);
    const int upseg = kp / 32, ncu = n_cu > 0 ? n_cu : 256;
    int best_parts = 0;
    double best_cost = 0.;
    for (int np = 1; np <= n_items; ++np) {
        const int segs = (n_items + np - 1) / np;
        if (segs > 8 || segs * kp > 1920) continue;
        const double cost = (double)scd_cdiv(nblk * np, ncu) * (segs * upseg + 3);
        if (!best_parts || cost < best_cost) { best_parts = np; best_cost = cost; }
    }
    SCD_REQUIRE(best_parts > 0, "estep_multi_launch: no part count fits");
    a.nparts = best_parts;
    { const int rc_ = scd_set_max_lds((const void*)estep_rbm_kernel, ERB_LDS); if (rc_) return rc_; }
    estep_rbm_kernel<<<(unsigned)(nblk * best_parts), 512, ERB_LDS, st>>>((const half_t*)(p + xh_off), (const float*)(p + xnorm_off), a, n);
    if (C0) {
        // the refine of all running restarts in one launch as well (blockIdx.y = restart), each with the single-restart launch's shape
        // (768 all-centres + 256 pair blocks: with 192 + 64 the all-centres rows of a late iteration took several rounds of their
        // ~16-us chain, 137 us per launch where the single-restart launches take 10-20)
        RefmArgs r;
        r.C0 = C0; r.c_stride = c_stride;
        r.ws0 = ws0; r.ws_stride = ws_stride; r.ct_off = ct_off; r.flags_off = a.flags_off; r.fcand_off = a.fcand_off; r.fulls_off = a.fulls_off;
        r.lab0 = lab0; r.lab_stride = lab_stride; r.segmap = a.segmap;
        estep_refine_multi_kernel<<<dim3(REFINE_GRID, (unsigned)n_items), 512, (size_t)d * 32 + 64, st>>>(X, r, d, k, kp, REFINE_PAIR);
    }
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
// the refine launch of scd_kmeans_estep for one restart, behind the merged filter
static int estep_multi_refine(const float* X, const float* C, void* ws, int64_t n, int d, int k, int32_t* labels, hipStream_t st) {
    const int dp = dpad(d), kp = kpad(k);
    char* w = (char*)ws;
    const size_t ch_off = 64 + scd_align(4 * (size_t)kp);
    const size_t ct_off = ch_off + scd_align(2 * (size_t)kp * dp);
    const size_t flags_off = ct_off + scd_align(4 * (size_t)kp * dp);
    const size_t fcand_off = flags_off + scd_align(4 * (size_t)n);
    const size_t fulls_off = fcand_off + scd_align(4 * (size_t)n);
    estep_refine_both_kernel<<<REFINE_GRID, 512, (size_t)d * 32 + 64, st>>>(X, C, (const float*)(w + ct_off), (EHdr*)w, (int*)(w + flags_off),
                                                                            (int*)(w + fcand_off), (int*)(w + fulls_off), d, k, kp, labels, nullptr,
                                                                            REFINE_PAIR);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// The Lloyd loops of a fit's restarts (the loop of scd_amd/kmeans.py:_lloyd_pipelined, faster_mix_k_means_pytorch.py:187-214, :244-275),
// ONE restart (scd_kmeans_lloyd_run[_sharded]) or ALL of them in lock-step (scd_kmeans_lloyd_run_multi: the restarts are independent
// once seeded).  Per restart the host runs one iteration behind the device - iteration i + 1 is enqueued (from iteration i's centres,
// which is what the sequential loop uses unless i has converged) before iteration i's statistics are looked at; if i turns out to have
// converged, i + 1 is dropped unseen.  Nothing but the iteration's kernels enters the stream:
//  * the statistics reach the host through pinned memory written by finalize_kernel's last block (the host spins on a sequence
//    number), not through a copy + event;
//  * labels and centres of iteration i live in slot i % 3 of caller-owned rings, so the least-inertia iteration's (the reference's
//    bookkeeping; nearly always one of the last two) are still there when the loop ends - a slot is copied out only when an OLDER best
//    iteration's slot is about to be re-used.
// In lock-step, iteration i of every restart still running is enqueued (restart by restart: each has its own handle - scratch, centre
// hand-over, statistics ring - and its own buffers, and may have its own stream), under a process group the [sums | counts] of all of
// them travel in ONE exchange, and then iteration i - 1 of every restart is settled.  Every restart executes exactly the launches of the
// one-restart loop with the same arguments, so its labels, centres, inertia and iteration count are the same bits.
struct LloydShared {
    const float* X_u; const void* prep_u; int64_t n_u; const void* X16_cat; int64_t n_cat; int d, k;
    const int32_t* labels_lab; const double* sums_lab; const int64_t* counts_lab; const double* sumsq4;
    int max_iter; double tol; size_t ws_e_bytes, ws_m_bytes;
};
struct LloydRestart {
    scd_handle h; int32_t* lab_ring; int32_t* labels_prev; const float* C_start; float* C_ring; double* sums; int64_t* counts;
    double* stats_ring; int32_t* best_labels; float* best_C; double* result_host; void* ws_e; void* ws_m; hipStream_t st;
    bool have_best = false, best_in_ring = false, active = true, died = false;
    int best_it = -1;
    float best = 0.f;
    double refined_seen = -1., changed_seen = -1., changed_prev = -1.;      // counts of the iterations the host has seen last (-1: none yet)
    int pending = -1, n_done = 0, delta_steps = 0, launched = 0, flags = 0;
    double seq_of[2] = {0., 0.};
};
static int lr_save_best(LloydRestart& r, const LloydShared& S) {          // ring slot of the best iteration -> the output buffers
    const size_t kd = (size_t)S.k * S.d;
    SCD_HIP(hipMemcpyAsync(r.best_labels, r.lab_ring + (size_t)(r.best_it % 3) * S.n_cat, (size_t)S.n_cat * 4, hipMemcpyDeviceToDevice, r.st));
    SCD_HIP(hipMemcpyAsync(r.best_C, r.C_ring + (size_t)(r.best_it % 3) * kd, kd * 4, hipMemcpyDeviceToDevice, r.st));
    r.best_in_ring = false;
    return SCD_OK;
}
static int lr_begin(LloydRestart& r, const LloydShared& S) {
    SCD_DEVICE_ENTRY(r.h, "scd_kmeans_lloyd_run");
    SCD_REQUIRE(r.C_start && r.C_ring && r.stats_ring && r.lab_ring && r.best_labels && r.best_C && r.result_host && r.labels_prev && r.sums &&
                r.counts && r.ws_e && r.ws_m, "scd_kmeans_lloyd_run: null argument");
    scd_handle h = r.h;
    if (!h->run_host) {
        SCD_HIP(hipHostMalloc((void**)&h->run_host, 2 * 8 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        for (int i = 0; i < 16; ++i) h->run_host[i] = 0.0;
        SCD_HIP(hipHostGetDevicePointer((void**)&h->run_dev, h->run_host, 0));
    }
    h->prep_C = nullptr;                            // no hand-over survives from an earlier fit on this handle
    h->prep_ok = 0;
    const int64_t l_num = S.n_cat - S.n_u;
    for (int sl = 0; sl < 3 && l_num > 0; ++sl)     // the labelled rows' labels never change: every slot carries them
        SCD_HIP(hipMemcpyAsync(r.lab_ring + (size_t)sl * S.n_cat, S.labels_lab, (size_t)l_num * 4, hipMemcpyDeviceToDevice, r.st));
    return SCD_OK;
}
// first half of iteration `it`: E-step + M-step (+ pack)
static int lr_launch_a(LloydRestart& r, const LloydShared& S, int it, double* pack_dst, bool estep_done = false, bool refine_done = false) {
    if (r.best_in_ring && r.best_it % 3 == it % 3) { const int rc0 = lr_save_best(r, S); if (rc0) return rc0; }   // that slot is about to be re-used
    const size_t kd = (size_t)S.k * S.d;
    const float* c_in = it == 0 ? r.C_start : r.C_ring + (size_t)((it - 1) % 3) * kd;
    double* stats = r.stats_ring + (it & 1) * 5;
    // a changed row costs the incremental M-step ~24 ns (two 768-column float64 flushes), a fresh M-step ~70 us: break-even ~ 2,900 rows at C2
    const double many = (double)(S.n_u / 32 > 256 ? S.n_u / 32 : 256);
    // SCD_ESTEP_FEW pays only when a handful of rows are flagged, the incremental M-step while few labels move: the cue is the
    // count the host has seen last (iteration it - 2)
    const bool few = it >= 2 && r.refined_seen >= 0. && r.refined_seen <= 64.;
    // the changes seen last are two iterations old, and they decay fast (95,000 / 28,600 / 3,700 / 40 / 27 at C2): extrapolate with
    // the last ratio squared.  A wrong guess costs time only - both M-steps give the same bits
    double pred = r.changed_seen;
    if (it >= 3 && r.changed_prev > 0. && r.changed_seen < r.changed_prev)
        pred = r.changed_seen * (r.changed_seen / r.changed_prev) * (r.changed_seen / r.changed_prev);
    const bool full = it < 2 || r.changed_seen < 0. || pred > many;
    // the hand-over is vouched for only when c_in IS the previous step's C_out: C_start was not produced by a finalize of this
    // run, and a recycled address from an earlier fit must not be mistaken for one
    r.flags = (few ? SCD_ESTEP_FEW : 0) | (it > 0 ? SCD_ESTEP_CENTRES_FROM_FINALIZE : 0) | (full ? SCD_LLOYD_FULL : 0);
    r.h->run_seq += 1.0;
    r.seq_of[it & 1] = r.h->run_seq;
    SCD_REQUIRE(S.k <= 8192, "scd_kmeans_lloyd_run: k > 8192");
    if (estep_done && !refine_done) {          // the merged filter has run: this restart's refine, then its M-step
        const int rc = estep_multi_refine(S.X_u, c_in, r.ws_e, S.n_u, S.d, S.k, r.lab_ring + (size_t)(it % 3) * S.n_cat + (S.n_cat - S.n_u), r.st);
        if (rc) return rc;
    }
    return lloyd_step_a(r.h, S.X_u, S.prep_u, S.n_u, S.X16_cat, S.n_cat, S.d, S.k, r.lab_ring + (size_t)(it % 3) * S.n_cat, r.labels_prev, c_in,
                        r.sums, r.counts, stats, r.flags, r.ws_e, S.ws_e_bytes, r.ws_m, S.ws_m_bytes, (void*)r.st, pack_dst, estep_done);
}
// second half: centres, shift, inertia, the next E-step's operands, the statistics to the host
static int lr_launch_b(LloydRestart& r, const LloydShared& S, int it, const double* fsums, const int64_t* fcounts, bool exchanged) {
    const size_t kd = (size_t)S.k * S.d;
    const float* c_in = it == 0 ? r.C_start : r.C_ring + (size_t)((it - 1) % 3) * kd;
    float* c_out = r.C_ring + (size_t)(it % 3) * kd;
    double* stats = r.stats_ring + (it & 1) * 5;
    const int rc = lloyd_step_b(r.h, S.prep_u, S.n_u, S.d, S.k, c_in, c_out, fsums ? fsums : r.sums, fcounts ? fcounts : r.counts, S.sums_lab,
                                S.counts_lab, S.sumsq4, stats, r.flags, r.ws_e, S.ws_e_bytes, (void*)r.st, r.h->run_dev + (it & 1) * 8,
                                r.seq_of[it & 1], 0, exchanged);
    if (rc) return rc;
    ++r.launched;
    r.delta_steps += (r.flags & SCD_LLOYD_FULL) ? 0 : 1;
    return SCD_OK;
}
// settle(i): wait for iteration i's statistics; book-keep; *converged when its centre shift is below tol
static int lr_settle(LloydRestart& r, const LloydShared& S, int i, bool* converged) {
    volatile double* host = r.h->run_host + (i & 1) * 8;
    const double want = r.seq_of[i & 1];
    long long spins = 0;
    while (host[7] != want) {
        if ((++spins & 0xFFFFF) == 0) {           // every ~1M polls: has the stream failed, or is this taking absurdly long?
            const hipError_t e = hipStreamQuery(r.st);
            if (e != hipSuccess && e != hipErrorNotReady) {
                scd_set_error("scd_kmeans_lloyd_run: stream error while waiting for iteration %d: %s", i, hipGetErrorString(e));
                return SCD_EHIP;
            }
            if (e == hipSuccess && host[7] != want) {
                scd_set_error("scd_kmeans_lloyd_run: iteration %d finished without publishing its statistics", i);
                return SCD_EHIP;
            }
        }
    }
    r.refined_seen = host[3];
    r.changed_prev = r.changed_seen;
    r.changed_seen = host[4];
    const float inertia = (float)host[1] + (float)host[0];          // float32 sum of the two float32 parts, as the reference's
    if (!r.have_best || inertia < r.best) {
        r.have_best = true;
        r.best = inertia;
        r.best_it = i;
        r.best_in_ring = true;
    }
    // An M-step that left a cluster empty ends the restart (faster_mix_k_means_pytorch.py:140-160, 192-214): the reference's centre of that
    // cluster is NaN, `torch.min` then returns NaN at the first NaN column for EVERY row, so every later iteration has a NaN inertia (never
    // the best) and a NaN shift (never below the tolerance) - the restart runs to max_iterations and keeps the best of the iterations up
    // to this one.  (Callers route the one configuration in which such a loop can recover - exactly one cluster without labelled rows -
    // to the Python-driven loop.)  The iteration launched on speculation is dropped, as after convergence.
    if (host[5] > 0.0) r.died = true;
    *converged = host[2] < S.tol || r.died;
    return SCD_OK;
}
static int lr_end(LloydRestart& r, const LloydShared& S) {
    if (r.best_in_ring) { const int rc0 = lr_save_best(r, S); if (rc0) return rc0; }
    r.result_host[0] = (double)r.best;
    r.result_host[1] = (double)(r.died ? S.max_iter : r.n_done);      // (the reference's dead loop runs on to max_iterations)
    r.result_host[2] = (double)r.delta_steps;
    r.result_host[3] = (double)r.launched;
    return SCD_OK;
}

// streams / events of the lock-step driver, created once per device and kept (a stream costs ~100 us to create)
struct LloydStreams {
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> ev;      // one per restart slot + 2
};
static LloydStreams* lloyd_streams(int device, int n_streams, int n_events) {
    static std::mutex mu;
    static std::map<int, LloydStreams> pool;
    std::lock_guard<std::mutex> lock(mu);
    LloydStreams& p = pool[device];
    while ((int)p.st.size() < n_streams) {
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        p.st.push_back(s);
    }
    while ((int)p.ev.size() < n_events) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        p.ev.push_back(e);
    }
    return &p;
}

static int lloyd_run_multi_impl(std::vector<LloydRestart>& rs, const LloydShared& S, hipStream_t st, const LloydXch* xch, int n_streams) {
    const int R = (int)rs.size();
    SCD_REQUIRE(R >= 1 && S.max_iter >= 1 && S.n_u > 0 && S.n_cat >= S.n_u && (S.n_cat == S.n_u || S.labels_lab), "scd_kmeans_lloyd_run: bad arguments");
    const size_t kd = (size_t)S.k * S.d, per = kd + (size_t)S.k;
    LloydStreams* ls = nullptr;
    const int ns = n_streams > R ? R : n_streams;
    // Every error return below leaves through this guard: the library-owned streams may still be using the caller's buffers (label and
    // centre rings, the E-step workspaces, best_labels), and the caller frees them as soon as it sees the status - so a failed call
    // drains its side streams first (the success path joins them into the caller's stream with events instead, at the end).
    // The stream / event pool is per device: one lock-step fit per device at a time (scd_kmeans_lloyd_run_multi is not re-entrant).
    struct Drain {
        LloydStreams* ls = nullptr;
        int ns = 0;
        bool joined = false;
        ~Drain() {
            if (ls && !joined)
                for (int j = 0; j < ns; ++j) (void)hipStreamSynchronize(ls->st[j]);
        }
    } drain;
    if (ns > 1) {
        ls = lloyd_streams(rs[0].h->device, ns, R + 2);
        SCD_REQUIRE(ls, "scd_kmeans_lloyd_run_multi: could not create streams / events");
        drain.ls = ls;
        drain.ns = ns;
        SCD_HIP(hipEventRecord(ls->ev[R], st));                       // whatever the caller enqueued before the call
        for (int j = 0; j < ns; ++j) SCD_HIP(hipStreamWaitEvent(ls->st[j], ls->ev[R], 0));
    }
    for (int j = 0; j < R; ++j) {
        rs[j].st = ls ? ls->st[j % ns] : st;
        const int rc = lr_begin(rs[j], S);
        if (rc) return rc;
    }
    // one merged filter launch per iteration when the shape is served and the restarts' E-step workspaces and label rings are equally
    // strided (the Python mirror allocates them as one tensor each)
    bool merge_ok = R >= 2 && estep_multi_serves(S.d, S.k, R);
    size_t ws_stride = 0, lab_stride = 0;
    if (merge_ok) {
        ws_stride = (size_t)((char*)rs[1].ws_e - (char*)rs[0].ws_e);
        lab_stride = (size_t)(rs[1].lab_ring - rs[0].lab_ring);
        merge_ok = (char*)rs[1].ws_e > (char*)rs[0].ws_e && rs[1].lab_ring > rs[0].lab_ring && ws_stride >= S.ws_e_bytes;
        for (int j = 0; j < R && merge_ok; ++j)
            merge_ok = (char*)rs[j].ws_e == (char*)rs[0].ws_e + (size_t)j * ws_stride && rs[j].lab_ring == rs[0].lab_ring + (size_t)j * lab_stride;
    }
    size_t cs_stride = 0, cr_stride = 0;
    bool cs_ok = false, cr_ok = false;
    if (merge_ok) {
        cs_ok = rs[1].C_start > rs[0].C_start;
        cr_ok = rs[1].C_ring > rs[0].C_ring;
        cs_stride = cs_ok ? (size_t)(rs[1].C_start - rs[0].C_start) : 0;
        cr_stride = cr_ok ? (size_t)(rs[1].C_ring - rs[0].C_ring) : 0;
        for (int j = 0; j < R; ++j) {
            cs_ok = cs_ok && rs[j].C_start == rs[0].C_start + (size_t)j * cs_stride;
            cr_ok = cr_ok && rs[j].C_ring == rs[0].C_ring + (size_t)j * cr_stride;
        }
    }
    std::vector<int> act;
    size_t n_act_prev = (size_t)R;
    int n_active = R;
    for (int it = 0; it < S.max_iter && n_active > 0; ++it) {
        act.clear();
        for (int j = 0; j < R; ++j) if (rs[j].active) act.push_back(j);
        if (ls && xch && it > 0 && act.size() != n_act_prev) {
            // a restart has dropped out: the running ones move up in the densely packed exchange buffer, i.e. a restart is about to pack
            // into the region another restart's finalize (on ANOTHER stream, possibly the dropped restart's speculative iteration) may
            // still be reading.  Once per drop-out: every stream waits for every stream
            for (int j = 0; j < ns; ++j) {
                SCD_HIP(hipEventRecord(ls->ev[j], ls->st[j]));
                SCD_HIP(hipStreamWaitEvent(st, ls->ev[j], 0));
            }
            SCD_HIP(hipEventRecord(ls->ev[R + 1], st));
            for (int j = 0; j < ns; ++j) SCD_HIP(hipStreamWaitEvent(ls->st[j], ls->ev[R + 1], 0));
        }
        n_act_prev = act.size();
        bool merged = false, refined = false;
        if (merge_ok && act.size() >= 2) {
            // ONE filter launch for every running restart (estep_rbm_kernel): the slots the labels go to are freed first, every
            // restart's stream is joined (its previous finalize wrote the centre operands), and the restarts' streams continue behind it
            std::vector<EstepMultiItem> items(act.size());
            for (size_t a = 0; a < act.size(); ++a) {
                LloydRestart& r = rs[act[a]];
                if (r.best_in_ring && r.best_it % 3 == it % 3) { const int rc0 = lr_save_best(r, S); if (rc0) return rc0; }
                const float* c_in = it == 0 ? r.C_start : r.C_ring + (size_t)((it - 1) % 3) * kd;
                items[a] = {r.h, c_in, r.ws_e, r.lab_ring + (size_t)(it % 3) * S.n_cat + (S.n_cat - S.n_u), act[a]};
                if (ls) {
                    SCD_HIP(hipEventRecord(ls->ev[act[a]], r.st));
                    SCD_HIP(hipStreamWaitEvent(st, ls->ev[act[a]], 0));
                }
            }
            // the restarts' centres of this iteration: the seedings (it = 0) or slot (it - 1) % 3 of the centre rings - one refine launch for
            // all of them when they are equally strided too
            const float* c00 = it == 0 ? rs[0].C_start : rs[0].C_ring + (size_t)((it - 1) % 3) * kd;
            const size_t c_stride = it == 0 ? cs_stride : cr_stride;
            const bool c_ok = it == 0 ? cs_ok : cr_ok;
            const int rc = estep_multi_launch(items.data(), (int)items.size(), S.X_u, S.prep_u, S.n_u, S.d, S.k, (char*)rs[0].ws_e, ws_stride,
                                              rs[0].lab_ring + (size_t)(it % 3) * S.n_cat + (S.n_cat - S.n_u), lab_stride, c_ok ? c00 : nullptr,
                                              c_stride, it > 0, rs[0].h->n_cu, st);
            refined = c_ok;
            if (rc) return rc;
            if (ls) {
                SCD_HIP(hipEventRecord(ls->ev[R + 1], st));
                for (int j = 0; j < ns; ++j) SCD_HIP(hipStreamWaitEvent(ls->st[j], ls->ev[R + 1], 0));
            }
            merged = true;
        }
        for (size_t a = 0; a < act.size(); ++a) {
            const int rc = lr_launch_a(rs[act[a]], S, it, xch ? xch->buf + a * per : nullptr, merged, refined);
            if (rc) return rc;
        }
        if (xch) {
            // ONE exchange for the iteration: [sums | counts] of every restart still running, densely packed in restart order (every rank
            // holds the same set: the stop decisions come from exchanged statistics)
            if (ls) {
                for (size_t a = 0; a < act.size(); ++a) {
                    SCD_HIP(hipEventRecord(ls->ev[act[a]], rs[act[a]].st));
                    SCD_HIP(hipStreamWaitEvent(st, ls->ev[act[a]], 0));
                }
            }
            const int rc = xch->fn(xch->ctx, xch->buf, (int64_t)(per * act.size()), (void*)st);
            if (rc) {
                scd_set_error("scd_kmeans_lloyd_run_sharded: the exchange callback failed (%d)", rc);
                return SCD_ERCCL;
            }
            if (ls) {
                SCD_HIP(hipEventRecord(ls->ev[R + 1], st));
                for (int j = 0; j < ns; ++j) SCD_HIP(hipStreamWaitEvent(ls->st[j], ls->ev[R + 1], 0));
            }
        }
        long long* cnt64 = xch ? (long long*)(xch->buf + per * (size_t)R) : nullptr;       // the int64 counts finalize reads, behind the exchange region
        for (size_t a = 0; a < act.size(); ++a) {
            LloydRestart& r = rs[act[a]];
            const double* fsums = nullptr;
            const int64_t* fcounts = nullptr;
            if (xch) {
                double* b = xch->buf + a * per;
                xch_unpack_kernel<<<(unsigned)scd_cdiv(S.k, 256), 256, 0, r.st>>>(b + kd, S.k, cnt64 + a * (size_t)S.k);
                fsums = b;
                fcounts = (const int64_t*)(cnt64 + a * (size_t)S.k);
            }
            const int rc = lr_launch_b(r, S, it, fsums, fcounts, xch != nullptr);
            if (rc) return rc;
        }
        for (size_t a = 0; a < act.size(); ++a) {
            LloydRestart& r = rs[act[a]];
            if (r.pending >= 0) {
                bool conv = false;
                r.n_done = r.pending + 1;
                { const int rc2 = lr_settle(r, S, r.pending, &conv); if (rc2) return rc2; }
                r.pending = -1;
                if (conv) {                       // iteration `it` was launched on speculation: dropped unseen (whatever follows on
                    r.active = false;             // this stream is ordered behind it)
                    --n_active;
                    continue;
                }
            }
            r.pending = it;
        }
    }
    for (int j = 0; j < R; ++j) {
        LloydRestart& r = rs[j];
        if (r.active && r.pending >= 0) {
            bool conv = false;
            r.n_done = r.pending + 1;
            { const int rc2 = lr_settle(r, S, r.pending, &conv); if (rc2) return rc2; }
        }
        { const int rc = lr_end(r, S); if (rc) return rc; }
    }
    if (ls) {                                                       // the caller's stream continues behind every restart's stream
        for (int j = 0; j < ns; ++j) {
            SCD_HIP(hipEventRecord(ls->ev[j], ls->st[j]));
            SCD_HIP(hipStreamWaitEvent(st, ls->ev[j], 0));
        }
    }
    drain.joined = true;
    return SCD_OK;
}

static int lloyd_run_impl(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat, int64_t n_cat,
                          int d, int k, const int32_t* labels_lab, int32_t* lab_ring, int32_t* labels_prev,
                          const float* C_start, float* C_ring, double* sums, int64_t* counts, const double* sums_lab,
                          const int64_t* counts_lab, const double* sumsq4, double* stats_ring, int max_iter, double tol,
                          int32_t* best_labels, float* best_C, double* result_host, void* ws_e, size_t ws_e_bytes, void* ws_m,
                          size_t ws_m_bytes, void* stream, const LloydXch* xch) {
    const LloydShared S = {X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_lab, sums_lab, counts_lab, sumsq4, max_iter, tol, ws_e_bytes, ws_m_bytes};
    std::vector<LloydRestart> rs(1);
    LloydRestart& r = rs[0];
    r.h = h; r.lab_ring = lab_ring; r.labels_prev = labels_prev; r.C_start = C_start; r.C_ring = C_ring; r.sums = sums; r.counts = counts;
    r.stats_ring = stats_ring; r.best_labels = best_labels; r.best_C = best_C; r.result_host = result_host; r.ws_e = ws_e; r.ws_m = ws_m;
    return lloyd_run_multi_impl(rs, S, (hipStream_t)stream, xch, 0);
}

extern "C" int scd_kmeans_lloyd_run(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat, int64_t n_cat,
                                    int d, int k, const int32_t* labels_lab, int32_t* lab_ring, int32_t* labels_prev,
                                    const float* C_start, float* C_ring, double* sums, int64_t* counts, const double* sums_lab,
                                    const int64_t* counts_lab, const double* sumsq4, double* stats_ring, int max_iter, double tol,
                                    int32_t* best_labels, float* best_C, double* result_host, void* ws_e, size_t ws_e_bytes, void* ws_m,
                                    size_t ws_m_bytes, void* stream) {
    return lloyd_run_impl(h, X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_lab, lab_ring, labels_prev, C_start, C_ring, sums, counts, sums_lab,
                          counts_lab, sumsq4, stats_ring, max_iter, tol, best_labels, best_C, result_host, ws_e, ws_e_bytes, ws_m, ws_m_bytes,
                          stream, nullptr);
}

// The same loop over a row shard (one process per GPU): sums / counts / sums_lab / counts_lab / sumsq4 semantics as above, but
// sums_lab / counts_lab / sumsq4 are the GLOBAL ones (the caller reduces them once per fit) and every iteration's [sums | counts] go
// through `exchange` before the centres are formed, so every rank forms the same centres, shift and inertia and takes the same
// stop / keep decisions; fresh-or-incremental M-step and the E-step hints stay per-rank choices (same bits either way).
extern "C" int scd_kmeans_lloyd_run_sharded(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const void* X16_cat,
                                            int64_t n_cat, int d, int k, const int32_t* labels_lab, int32_t* lab_ring, int32_t* labels_prev,
                                            const float* C_start, float* C_ring, double* sums, int64_t* counts, const double* sums_lab,
                                            const int64_t* counts_lab, const double* sumsq4, double* stats_ring, int max_iter, double tol,
                                            int32_t* best_labels, float* best_C, double* result_host, void* ws_e, size_t ws_e_bytes,
                                            void* ws_m, size_t ws_m_bytes, void* stream, double* xbuf, scd_exchange_fn exchange,
                                            void* exchange_ctx) {
    SCD_REQUIRE(xbuf && exchange, "scd_kmeans_lloyd_run_sharded: null exchange buffer / callback");
    const LloydXch x = {xbuf, exchange, exchange_ctx};
    return lloyd_run_impl(h, X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_lab, lab_ring, labels_prev, C_start, C_ring, sums, counts, sums_lab,
                          counts_lab, sumsq4, stats_ring, max_iter, tol, best_labels, best_C, result_host, ws_e, ws_e_bytes, ws_m, ws_m_bytes,
                          stream, &x);
}

// All restarts of a fit in lock-step (see above).  rs[j] carries restart j's handle and buffers (the per-restart arguments of
// scd_kmeans_lloyd_run); everything else is shared.  xbuf / exchange non-NULL: a row shard - xbuf holds R * (k*d + 2k) doubles and ONE
// exchange per iteration carries the [k*d sums | k counts] of the restarts still running, (running restarts) * (k*d + k) doubles, in
// restart order.  n_streams > 1: restart j's launches go to one of that many library-owned streams (ordered behind everything already
// in `stream`, and `stream` continues behind them): the restarts' small latency-bound launches overlap each other's E-steps.
extern "C" int scd_kmeans_lloyd_run_multi(const scd_lloyd_restart* restarts, int R, const float* X_u, const void* prep_u, int64_t n_u,
                                          const void* X16_cat, int64_t n_cat, int d, int k, const int32_t* labels_lab,
                                          const double* sums_lab, const int64_t* counts_lab, const double* sumsq4, int max_iter, double tol,
                                          size_t ws_e_bytes, size_t ws_m_bytes, void* stream, double* xbuf, scd_exchange_fn exchange,
                                          void* exchange_ctx, int n_streams) {
    SCD_REQUIRE(restarts && R >= 1 && R <= 1024, "scd_kmeans_lloyd_run_multi: bad restart list");
    SCD_REQUIRE((xbuf != nullptr) == (exchange != nullptr), "scd_kmeans_lloyd_run_multi: exchange buffer and callback come together");
    const LloydShared S = {X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_lab, sums_lab, counts_lab, sumsq4, max_iter, tol, ws_e_bytes, ws_m_bytes};
    std::vector<LloydRestart> rs((size_t)R);
    for (int j = 0; j < R; ++j) {
        const scd_lloyd_restart& q = restarts[j];
        LloydRestart& r = rs[j];
        SCD_REQUIRE(q.h, "scd_kmeans_lloyd_run_multi: restart %d has no handle", j);
        for (int i = 0; i < j; ++i) SCD_REQUIRE(restarts[i].h != q.h, "scd_kmeans_lloyd_run_multi: restarts %d and %d share a handle", i, j);
        r.h = q.h; r.lab_ring = q.lab_ring; r.labels_prev = q.labels_prev; r.C_start = q.C_start; r.C_ring = q.C_ring; r.sums = q.sums;
        r.counts = q.counts; r.stats_ring = q.stats_ring; r.best_labels = q.best_labels; r.best_C = q.best_C; r.result_host = q.result_host;
        r.ws_e = q.ws_e; r.ws_m = q.ws_m;
    }
    const LloydXch x = {xbuf, exchange, exchange_ctx};
    return lloyd_run_multi_impl(rs, S, (hipStream_t)stream, xbuf ? &x : nullptr, n_streams);
}

extern "C" int scd_kmeans_lloyd_step(scd_handle h, const float* X_u, const void* prep_u, int64_t n_u, const float* X_cat,
                                     const void* X16_cat, int64_t n_cat, int d, int k, int32_t* labels_cat, const float* C_in,
                                     float* C_out, double* sums, int64_t* counts, double* stats, int expect_few, void* ws_e,
                                     size_t ws_e_bytes, void* ws_m, size_t ws_m_bytes, void* stream) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_lloyd_step");
    SCD_REQUIRE(h && X_u && prep_u && (X_cat || X16_cat) && labels_cat && C_in && C_out && sums && counts && stats && ws_e && ws_m,
                "scd_kmeans_lloyd_step: null argument");
    SCD_REQUIRE(n_u > 0 && n_cat >= n_u && C_in != C_out, "scd_kmeans_lloyd_step: bad arguments (n_u=%lld n_cat=%lld)", (long long)n_u,
                (long long)n_cat);
    const int64_t l_num = n_cat - n_u;
    int rc = SCD_OK;
    rc = scd_kmeans_estep_hint(h, expect_few);
    if (!rc) rc = scd_kmeans_estep(h, X_u, prep_u, C_in, n_u, d, k, labels_cat + l_num, nullptr, ws_e, ws_e_bytes, stream);
    if (!rc)
        rc = X16_cat ? scd_kmeans_mstep_f16(h, X16_cat, labels_cat, C_in, n_cat, d, k, l_num, sums, counts, stats, ws_m, ws_m_bytes, stream)
                     : scd_kmeans_mstep(h, X_cat, labels_cat, C_in, n_cat, d, k, l_num, sums, counts, stats, ws_m, ws_m_bytes, stream);
    // stats[3]: rows this iteration's E-step re-evaluated exactly (the caller's cue for SCD_ESTEP_FEW two iterations later)
    if (!rc) rc = finalize_impl(h, sums, counts, k, d, C_in, C_out, stats + 2, 0, prep_u, ws_e, ws_e_bytes, n_u, stream, stats + 3);
    return rc;
}

#include "kmeans_sk_impl.h"
