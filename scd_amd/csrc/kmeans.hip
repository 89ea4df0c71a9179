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
    return scd_align(64 + 8 * dp) + scd_align(4 * (size_t)n) + scd_align(2 * (size_t)n * dp) + 256;
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
// E-step workspace:  [0,64) EHdr | cn float[Kp] | ch half[Kp*Dp] | flag list int32[n]
struct EHdr {
    unsigned cmax_bits;   // max ||c'||
    int flag_cnt;
    int pad[14];
};
static inline int kpad(int k) { return (k + 127) / 128 * 128; }

extern "C" size_t scd_kmeans_estep_ws_bytes(int64_t n, int d, int k) {
    size_t kp = kpad(k), dp = dpad(d);
    return 64 + scd_align(4 * kp) + scd_align(2 * kp * dp) + scd_align(4 * (size_t)n) + 256;
}

// one block per (padded) centre: c' = (c-mu)*scale -> fp16; cn = ||c'||^2 (float64 -> float32)
__global__ void __launch_bounds__(256) prep_centers_kernel(const float* __restrict__ C, int k, int d, int dp,
                                                           const PrepHdr* hdr, const double* mu, EHdr* eh, float* cn,
                                                           half_t* ch) {
    __shared__ double red[4];
    __shared__ int bad;
    const int c = blockIdx.x;
    if (threadIdx.x == 0) bad = 0;
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
    }
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const bool dead = (c >= k) || bad;
    if (dead) {   // padded or NaN centre (empty cluster): can never win
        for (int j = threadIdx.x; j < dp; j += 256) ch[(size_t)c * dp + j] = (half_t)0.f;
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

// MFMA filter.  Block = 4 waves = 128 points; each wave owns 32 points (MFMA columns) against a chunk
// of 128 centres (4 x 32 MFMA rows) staged through LDS; v_mfma_f32_32x32x16_f16:
//   A[row = centre r][k = 8h+j]  from LDS (XOR-swizzled 256-B rows, ds_read_b128)
//   B[k = 8h+j][col = point r]   straight from global (16 B per lane, each row streamed once)
//   D[row = (reg&3)+8(reg>>2)+4h][col = point r]
// so every lane ends with 16 scores per 32-centre block for ONE point: the running best/second is
// in-lane; lanes r and r+32 are merged once at the end.
__global__ void __launch_bounds__(256) estep_mfma_kernel(const half_t* __restrict__ xh, const float* __restrict__ xnorm,
                                                         const half_t* __restrict__ ch, const float* __restrict__ cn,
                                                         EHdr* eh, int* flag_list, long long n, int dp, int kp,
                                                         int32_t* __restrict__ labels) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 256];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const long long point = (long long)blockIdx.x * 128 + wave * 32 + r;
    const long long prow = point < n ? point : n - 1;
    const half_t* xrow = xh + prow * dp + 8 * hh;

    float best = INFINITY, second = INFINITY;
    int bidx = 0;

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
                if (s < best) {
                    second = best;
                    best = s;
                    bidx = centre;
                } else if (s < second) {
                    second = s;
                }
            }
        }
    }
    // merge the two half-wave lanes that share a point
    const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64);
    float mb, ms;
    int mi;
    if (ob < best || (ob == best && oi < bidx)) {
        mb = ob; mi = oi; ms = fminf(best, os);
    } else {
        mb = best; mi = bidx; ms = fminf(second, ob);
    }
    if (hh == 0 && point < n) {
        labels[point] = mi;
        const float cmax = __uint_as_float(eh->cmax_bits);
        const float sq = sqrtf((float)dp);
        // |s~ - s| <= A*||x'|| + B : fp16 rounding of both operands (2^-10), fp32 accumulation (dp*2^-24),
        // fp16 subnormal flush-free absolute term, final fp32 ops; x1.5 safety.
        const float A = 1.5f * (2.02f * (9.765625e-4f + dp * 5.9604645e-8f) * cmax + 4.8e-7f * cmax + 6.0e-8f * sq);
        const float B = 1.5f * (6.0e-8f * sq * cmax + 2.4e-7f * cmax * cmax);
        const float E = A * xnorm[point] + B;
        if (!(ms - mb > 2.0f * E)) {       // also catches NaN
            const int pos = atomicAdd(&eh->flag_cnt, 1);
            flag_list[pos] = (int)point;
        }
    }
}

// exact re-evaluation of flagged rows: one wave per row, float64 difference form over all K centres
__global__ void __launch_bounds__(64) estep_refine_kernel(const float* __restrict__ X, const float* __restrict__ C,
                                                          const EHdr* eh, const int* flag_list, int d, int k,
                                                          int32_t* labels) {
    const int lane = threadIdx.x;
    const int cnt = eh->flag_cnt;
    for (int f = blockIdx.x; f < cnt; f += gridDim.x) {
        const long long row = flag_list[f];
        const float* x = X + row * d;
        double best = INFINITY;
        int bi = 0;
        for (int c = 0; c < k; ++c) {
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

extern "C" int scd_kmeans_estep(scd_handle h, const float* X, const void* prep, const float* C, int64_t n, int d, int k,
                                int32_t* labels_out, int32_t* refine_rows_out, void* ws, size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && X && prep && C && labels_out && ws, "scd_kmeans_estep: null argument");
    SCD_REQUIRE(n > 0 && d > 0 && k > 0 && n < (1ll << 31), "scd_kmeans_estep: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    SCD_REQUIRE(ws_bytes >= scd_kmeans_estep_ws_bytes(n, d, k), "scd_kmeans_estep: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const int dp = dpad(d), kp = kpad(k);
    char* w = (char*)ws;
    EHdr* eh = (EHdr*)w;
    float* cn = (float*)(w + 64);
    half_t* ch = (half_t*)(w + 64 + scd_align(4 * (size_t)kp));
    int* flags = (int*)((char*)ch + scd_align(2 * (size_t)kp * dp));
    const char* p = (const char*)prep;
    const PrepHdr* ph = (const PrepHdr*)p;
    const size_t xnorm_off = scd_align(64 + 8 * (size_t)dp);
    const size_t xh_off = xnorm_off + scd_align(4 * (size_t)n);
    SCD_HIP(hipMemsetAsync(eh, 0, 64, st));
    prep_centers_kernel<<<kp, 256, 0, st>>>(C, k, d, dp, ph, (const double*)(p + 64), eh, cn, ch);
    estep_mfma_kernel<<<(unsigned)scd_cdiv(n, 128), 256, 0, st>>>((const half_t*)(p + xh_off), (const float*)(p + xnorm_off),
                                                                    ch, cn, eh, flags, n, dp, kp, labels_out);
    estep_refine_kernel<<<2048, 64, 0, st>>>(X, C, eh, flags, d, k, labels_out);
    if (refine_rows_out) SCD_HIP(hipMemcpyAsync(refine_rows_out, &eh->flag_cnt, 4, hipMemcpyDeviceToDevice, st));
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
    SCD_REQUIRE(h && X && C && labels && d2_out && n > 0 && d > 0 && k > 0, "scd_kmeans_rowdist: bad arguments");
    rowdist_kernel<false><<<(unsigned)scd_cdiv(n, 4), 256, 0, (hipStream_t)stream_>>>(X, C, labels, n, d, k, d2_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_kmeans_min_update(scd_handle h, const float* X, const float* c_new, int64_t n, int d, float* d2_inout,
                                     void* stream_) {
    SCD_REQUIRE(h && X && c_new && d2_inout && n > 0 && d > 0, "scd_kmeans_min_update: bad arguments");
    rowdist_kernel<true><<<(unsigned)scd_cdiv(n, 4), 256, 0, (hipStream_t)stream_>>>(X, c_new, nullptr, n, d, 1, d2_inout);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// full [n,k] exact distances: each wave keeps 4 rows in registers and sweeps the centres
__global__ void __launch_bounds__(256) dist_kernel(const float* __restrict__ X, const float* __restrict__ C, long long n,
                                                   int d, int k, int mode, float* out, int32_t* cost) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= n) return;
    for (int c = 0; c < k; ++c) {
        const float* cc = C + (size_t)c * d;
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = lane; j < d; j += 64) {
            const double cv = (double)cc[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long row = row0 + q < n ? row0 + q : n - 1;
                const double a = (double)X[row * d + j] - cv;
                s[q] = fma(a, a, s[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double t = wave_sum_f64(s[q]);
            if (lane == 0 && row0 + q < n) {
                const float d2 = (float)t;
                const float rt = sqrtf(d2);           // correctly rounded
                out[(row0 + q) * k + c] = mode ? rt : d2;
                if (cost) cost[(row0 + q) * k + c] = (int32_t)rintf(rt * 1000.0f);
            }
        }
    }
}

extern "C" int scd_kmeans_dist(scd_handle h, const float* X, const float* C, int64_t n, int d, int k, int mode, float* out,
                               int32_t* cost_out, void* stream_) {
    SCD_REQUIRE(h && X && C && out && n > 0 && d > 0 && k > 0, "scd_kmeans_dist: bad arguments");
    dist_kernel<<<(unsigned)scd_cdiv(n, 16), 256, 0, (hipStream_t)stream_>>>(X, C, n, d, k, mode, out, cost_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------
// M-step partial sums.  grid = (column slices, row chunks); a block owns SW columns of its row chunk and
// accumulates per-cluster float64 sums in LDS (ds_add_f64), plus the inertia of (x - c_old[label]).
struct MstepPlan { int sw, slices, chunks; size_t lds; };
static MstepPlan mstep_plan(int64_t n, int d, int k) {
    MstepPlan p;
    p.sw = 64;
    while (p.sw > 4 && (size_t)k * p.sw * 12 + 4 * (size_t)k + 64 > 150 * 1024) p.sw >>= 1;
    p.slices = (d + p.sw - 1) / p.sw;
    int want = (768 + p.slices - 1) / p.slices;
    long long maxc = scd_cdiv(n, 64);
    p.chunks = (int)(want < maxc ? want : maxc);
    if (p.chunks < 1) p.chunks = 1;
    p.lds = (size_t)k * p.sw * 12 + 4 * (size_t)k + 64;
    return p;
}
extern "C" size_t scd_kmeans_mstep_ws_bytes(int64_t n, int d, int k) {
    MstepPlan p = mstep_plan(n, d, k);
    return scd_align((size_t)p.chunks * k * d * 8) + scd_align((size_t)p.chunks * p.slices * 16) + 256;
}

__global__ void __launch_bounds__(256) mstep_kernel(const float* __restrict__ X, const int32_t* __restrict__ labels,
                                                    const float* __restrict__ Cold, long long n, int d, int k, int sw,
                                                    long long split, double* part, double* ipart,
                                                    unsigned long long* counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sums = (double*)smem;                       // [k][sw]
    float* cold = (float*)(smem + (size_t)k * sw * 8);  // [k][sw]
    int* cnt = (int*)(smem + (size_t)k * sw * 12);      // [k]
    __shared__ double ired[8];
    const int slice = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
    const int col0 = slice * sw;
    for (int i = threadIdx.x; i < k * sw; i += 256) {
        sums[i] = 0.0;
        const int c = i / sw, j = col0 + i % sw;
        cold[i] = (Cold && j < d) ? Cold[(size_t)c * d + j] : 0.f;
    }
    for (int i = threadIdx.x; i < k; i += 256) cnt[i] = 0;
    __syncthreads();
    const long long rows_per = scd_cdiv_dev(n, nchunks);
    const long long r0 = (long long)chunk * rows_per;
    const long long r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    const int lpr = sw / 4;                  // lanes per row (float4 each)
    const int rpb = 256 / lpr;               // rows per block step
    const int sub = threadIdx.x / lpr, q4 = (threadIdx.x % lpr) * 4;
    double in0 = 0.0, in1 = 0.0;
    for (long long r = r0 + sub; r < r1; r += rpb) {
        const int l = labels[r];
        if (l < 0 || l >= k) continue;
        if (slice == 0 && q4 == 0) atomicAdd(&cnt[l], 1);
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = col0 + q4 + q;
            if (j < d) {
                const float xv = X[r * d + j];
                const double df = (double)xv - (double)cold[l * sw + q4 + q];
                acc = fma(df, df, acc);
                atomicAdd(&sums[l * sw + q4 + q], (double)xv);
            }
        }
        if (r < split) in0 += acc; else in1 += acc;
    }
    in0 = wave_sum_f64(in0);
    in1 = wave_sum_f64(in1);
    if ((threadIdx.x & 63) == 0) {
        ired[(threadIdx.x >> 6) * 2] = in0;
        ired[(threadIdx.x >> 6) * 2 + 1] = in1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * sw; i += 256) {
        const int c = i / sw, j = col0 + i % sw;
        if (j < d) part[((size_t)chunk * k + c) * d + j] = sums[i];
    }
    if (threadIdx.x == 0) {
        ipart[((size_t)chunk * gridDim.x + slice) * 2] = ired[0] + ired[2] + ired[4] + ired[6];
        ipart[((size_t)chunk * gridDim.x + slice) * 2 + 1] = ired[1] + ired[3] + ired[5] + ired[7];
    }
    if (slice == 0)
        for (int i = threadIdx.x; i < k; i += 256)
            if (cnt[i]) atomicAdd(&counts[i], (unsigned long long)cnt[i]);
}

__global__ void __launch_bounds__(256) mstep_reduce_kernel(const double* part, const double* ipart, int chunks, int nip,
                                                           long long kd, double* sums, double* inertia) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < kd) {
        double s = 0.0;
        for (int c = 0; c < chunks; ++c) s += part[(size_t)c * kd + i];
        sums[i] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 && inertia) {
        double s = 0.0;
        for (int c = 0; c < nip; ++c) s += ipart[(size_t)c * 2 + threadIdx.x];
        inertia[threadIdx.x] = s;
    }
}

extern "C" int scd_kmeans_mstep(scd_handle h, const float* X, const int32_t* labels, const float* C_old, int64_t n, int d,
                                int k, int64_t split, double* sums, int64_t* counts, double* inertia, void* ws,
                                size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && X && labels && sums && counts && ws, "scd_kmeans_mstep: null argument");
    SCD_REQUIRE(n > 0 && d > 0 && k > 0, "scd_kmeans_mstep: bad shape");
    SCD_REQUIRE(ws_bytes >= scd_kmeans_mstep_ws_bytes(n, d, k), "scd_kmeans_mstep: workspace too small");
    MstepPlan p = mstep_plan(n, d, k);
    SCD_REQUIRE(p.lds <= 160 * 1024 - 256, "scd_kmeans_mstep: k=%d too large for the LDS-privatised M-step", k);
    hipStream_t st = (hipStream_t)stream_;
    double* part = (double*)ws;
    double* ipart = (double*)((char*)ws + scd_align((size_t)p.chunks * k * d * 8));
    SCD_HIP(hipMemsetAsync(counts, 0, 8 * (size_t)k, st));
    static bool attr_set = false;
    if (!attr_set) {
        SCD_HIP(hipFuncSetAttribute((const void*)mstep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        attr_set = true;
    }
    mstep_kernel<<<dim3(p.slices, p.chunks), 256, p.lds, st>>>(X, labels, C_old, n, d, k, p.sw, split, part, ipart,
                                                               (unsigned long long*)counts);
    const long long kd = (long long)k * d;
    mstep_reduce_kernel<<<(unsigned)scd_cdiv(kd, 256), 256, 0, st>>>(part, ipart, p.chunks, p.chunks * p.slices, kd, sums,
                                                                    inertia);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// centres = sums / counts; shift = (sum_k ||c_k - c_old_k||)^2.  Single block, fixed reduction order.
__global__ void __launch_bounds__(1024) finalize_kernel(const double* sums, const long long* counts, int k, int d,
                                                        const float* Cold, float* Cout, double* shift) {
    __shared__ double wred[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double tot = 0.0;
    for (int c = wave; c < k; c += 16) {
        const double cnt = (double)counts[c];
        double ss = 0.0;
        for (int j = lane; j < d; j += 64) {
            const float v = (float)(sums[(size_t)c * d + j] / cnt);     // 0/0 -> NaN for an empty cluster
            Cout[(size_t)c * d + j] = v;
            if (Cold) {
                const double df = (double)v - (double)Cold[(size_t)c * d + j];
                ss = fma(df, df, ss);
            }
        }
        ss = wave_sum_f64(ss);
        tot += sqrt(ss);
    }
    if (lane == 0) wred[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0 && shift) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += wred[i];
        *shift = t * t;
    }
}

extern "C" int scd_kmeans_finalize(scd_handle h, const double* sums, const int64_t* counts, int k, int d,
                                   const float* C_old, float* C_out, double* shift_out, void* stream_) {
    SCD_REQUIRE(h && sums && counts && C_out && k > 0 && d > 0, "scd_kmeans_finalize: bad arguments");
    SCD_REQUIRE(C_old != C_out, "scd_kmeans_finalize: C_out must not alias C_old");
    finalize_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(sums, (const long long*)counts, k, d, C_old, C_out, shift_out);
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

__global__ void __launch_bounds__(1024) sum_kernel(const float* __restrict__ x, long long n, double* out) {
    __shared__ double sh[32];
    const long long seg = scd_cdiv_dev(n, 1024);
    const long long a = threadIdx.x * seg, b = (a + seg < n) ? a + seg : n;
    double s = 0.0;
    for (long long i = a; i < b; ++i) s += (double)x[i];
    double tot;
    block_scan_excl_1024(s, sh, &tot);
    if (threadIdx.x == 0) *out = tot;
}

extern "C" int scd_sum_f32(scd_handle h, const float* x, int64_t n, double* out, void* stream_) {
    SCD_REQUIRE(h && x && out && n > 0, "scd_sum_f32: bad arguments");
    sum_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(x, n, out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" size_t scd_kpp_draw_ws_bytes(int64_t n) { (void)n; return 256; }

// total_in / prefix_in (device doubles, nullable) make the draw shard-aware: prob = d2 / float(total), the running
// cumulative starts at *prefix_in (sum of prob over the lower-ranked shards).  probsum_out (nullable) receives this
// shard's float64 sum of prob.  idx_out may be NULL when only probsum_out is wanted.
__global__ void __launch_bounds__(1024) kpp_draw_kernel(const float* __restrict__ d2, long long n, float r,
                                                        const double* total_in, const double* prefix_in,
                                                        long long* idx_out, double* probsum_out) {
    __shared__ double sh[32];
    __shared__ long long best;
    const long long seg = scd_cdiv_dev(n, 1024);
    const long long a = threadIdx.x * seg, b = (a + seg < n) ? a + seg : n;
    if (threadIdx.x == 0) best = 0x7fffffffffffffffll;
    double tot;
    if (total_in) {
        tot = *total_in;
        __syncthreads();
    } else {
        double s = 0.0;
        for (long long i = a; i < b; ++i) s += (double)d2[i];
        block_scan_excl_1024(s, sh, &tot);
    }
    const float totf = (float)tot;
    double ps = 0.0;
    for (long long i = a; i < b; ++i) ps += (double)__fdiv_rn(d2[i], totf);
    double ptot;
    const double pre = block_scan_excl_1024(ps, sh, &ptot) + (prefix_in ? *prefix_in : 0.0);
    if (threadIdx.x == 0 && probsum_out) *probsum_out = ptot;
    if (!idx_out) return;
    double run = pre;
    long long found = 0x7fffffffffffffffll;
    for (long long i = a; i < b; ++i) {
        run += (double)__fdiv_rn(d2[i], totf);
        if ((float)run >= r) {
            found = i;
            break;
        }
    }
    if (found != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&best, (unsigned long long)found);
    __syncthreads();
    if (threadIdx.x == 0) *idx_out = (best == 0x7fffffffffffffffll) ? -1 : best;
}

extern "C" int scd_kpp_draw(scd_handle h, const float* d2, int64_t n, float r, const double* total, const double* prefix,
                            int64_t* idx_out, double* probsum_out, void* ws, size_t ws_bytes, void* stream_) {
    (void)ws; (void)ws_bytes;
    SCD_REQUIRE(h && d2 && (idx_out || probsum_out) && n > 0, "scd_kpp_draw: bad arguments");
    kpp_draw_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(d2, n, r, total, prefix, (long long*)idx_out, probsum_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
