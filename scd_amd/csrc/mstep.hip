// K-Means M-step partial sums + inertia for gfx950 (sskm_constrained.py:118-128 under /root/reference):
//   for idx in range(k): centers[idx] = cat_feats[labels == idx].mean(0);   inertia = sum ||x - c_old[label]||^2
// Sort-then-segment instead of scatter: a counting sort of (label, row) keys (rocPRIM radix sort for k > 8191) turns the
// scatter into contiguous runs; each wave then streams 64 sorted rows (whole 2-3 KB rows, coalesced, eight in flight), keeps its 4*D/256 columns per
// lane in float64 REGISTERS while the label is unchanged, and flushes a run with float64 global atomics (about
// N/64 + K flushes of D values).  Float64 accumulation makes the result independent of the flush order after the
// single rounding to float32 in scd_kmeans_finalize.  No LDS, no partial slabs.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

extern "C" size_t scd_kmeans_mstep_ws_bytes(int64_t n, int d, int k) {
    (void)d; (void)k;
    return 2 * scd_align(8 * (size_t)n) + scd_align(24 * (size_t)n + (8u << 20)) + 256;
}

__global__ void __launch_bounds__(256) mstep_keys_kernel(const int32_t* __restrict__ labels, long long n, int k,
                                                         unsigned long long* keys) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = labels[i];
    const unsigned long long bucket = (l < 0 || l >= k) ? (unsigned long long)k : (unsigned long long)l;
    keys[i] = (bucket << 32) | (unsigned long long)i;
}

// Counting sort by label (k + 1 buckets, bucket k = invalid labels) in three short kernels; rows of one label end up in
// one run, in no particular order inside it (float64 sums do not depend on it after the single rounding to float32).
// rocPRIM's radix_sort_keys took 9 launches / ~55 us for 95k keys (merge-sort path); this takes ~15 us.  The first kernel
// also clears sums / counts / inertia (three memset launches fewer).
#define MS_BLK 1024
__global__ void __launch_bounds__(256) mstep_hist_kernel(const int32_t* __restrict__ labels, long long n, int k, int* __restrict__ hist,
                                                         double* sums, long long nsum, unsigned long long* counts, double* inertia) {
    extern __shared__ int lh[];
    for (int i = threadIdx.x; i <= k; i += 256) lh[i] = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nsum; i += (long long)gridDim.x * 256) sums[i] = 0.0;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < k; i += 256) counts[i] = 0;
        if (inertia && threadIdx.x < 2) inertia[threadIdx.x] = 0.0;
    }
    __syncthreads();
    const long long base = (long long)blockIdx.x * MS_BLK;
#pragma unroll
    for (int u = 0; u < MS_BLK / 256; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const int l = labels[i];
            atomicAdd(&lh[(l < 0 || l >= k) ? k : l], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= k; i += 256) hist[(size_t)blockIdx.x * (k + 1) + i] = lh[i];
}
// hist[b][l] -> (rows with label l in blocks before b), tot[l] = rows with label l.  One wave per label, lane = block
// (wave-level exclusive scans over the blocks, 64 at a time).
__global__ void __launch_bounds__(256) mstep_scan_kernel(int* __restrict__ hist, int nblk, int k, int* __restrict__ tot) {
    const int kb = k + 1;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= kb) return;
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < nblk ? hist[(size_t)b * kb + l] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (b < nblk) hist[(size_t)b * kb + l] = carry + inc - c;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) tot[l] = carry;
}
// keys[offset of (block, label) + arrival order] = (label, row); every block first scans the label totals itself
// (<= 8192 values: thread t owns a contiguous segment, thread 0 chains the 256 segment sums)
__global__ void __launch_bounds__(256) mstep_scatter_kernel(const int32_t* __restrict__ labels, long long n, int k,
                                                            const int* __restrict__ offs, const int* __restrict__ tot,
                                                            unsigned long long* __restrict__ keys) {
    extern __shared__ int cur[];
    __shared__ int seg_sum[256];
    const int kb = k + 1;
    const int seg = (kb + 255) / 256;
    const int l0 = threadIdx.x * seg, l1 = l0 + seg < kb ? l0 + seg : kb;
    int sum = 0;
    for (int l = l0; l < l1; ++l) sum += tot[l];
    seg_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = 0;
        for (int t = 0; t < 256; ++t) {
            const int c = seg_sum[t];
            seg_sum[t] = r;
            r += c;
        }
    }
    __syncthreads();
    int run = seg_sum[threadIdx.x];
    for (int l = l0; l < l1; ++l) {
        cur[l] = run + offs[(size_t)blockIdx.x * kb + l];
        run += tot[l];
    }
    __syncthreads();
    const long long base = (long long)blockIdx.x * MS_BLK;
#pragma unroll
    for (int u = 0; u < MS_BLK / 256; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const int l = labels[i];
            const int b = (l < 0 || l >= k) ? k : l;
            const int pos = atomicAdd(&cur[b], 1);
            keys[pos] = ((unsigned long long)b << 32) | (unsigned long long)i;
        }
    }
}

// Lane owns columns lane, lane+64, ... (MAXG = ceil(d/64) <= 16): every load and every float64 atomic of a wave touches
// 64 consecutive elements (256 B / 512 B contiguous).  ROWS sorted rows per wave.
template <int MAXG, int MSTEP_ROWS, int MU>
__global__ void __launch_bounds__(256) mstep_segment_kernel(const float* __restrict__ X, const unsigned long long* __restrict__ keys,
                                                            const float* __restrict__ Cold, long long n, int d, int k,
                                                            long long split, double* __restrict__ sums,
                                                            unsigned long long* __restrict__ counts, double* __restrict__ inertia) {
    const int lane = threadIdx.x & 63;
    const long long s0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * MSTEP_ROWS;
    // (a wave past the end of the keys stays alive - empty row range, no run - so that every wave reaches the block barrier below)
    const long long s1 = s0 + MSTEP_ROWS < n ? s0 + MSTEP_ROWS : n;
    double acc[MAXG] = {};
    float co[MAXG] = {};
    double in0 = 0.0, in1 = 0.0;
    int cur = -1;
    long long run = 0;
    auto flush = [&]() {
        if (cur >= 0 && cur < k) {
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                const int c = g * 64 + lane;
                if (c < d) atomicAdd(&sums[(size_t)cur * d + c], acc[g]);
            }
            if (lane == 0) atomicAdd(&counts[cur], (unsigned long long)run);
        }
    };
    // 4 rows per iteration: their keys and row slices are all loaded before the first one is consumed
    for (long long sb = s0; sb < s1; sb += MU) {
        unsigned long long key[MU];
        float xv[MU][MAXG];
#pragma unroll
        for (int u = 0; u < MU; ++u) key[u] = sb + u < s1 ? keys[sb + u] : ~0ull;
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            // unconditional loads from clamped (always valid) addresses: a per-element "load or zero" branch would make
            // the compiler wait for every load separately; out-of-range lanes are masked when the value is used
            const bool live = (unsigned)(key[u] >> 32) < (unsigned)k;
            const long long row = live ? (long long)(key[u] & 0xffffffffull) : 0;
            const float* xr = X + row * d;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                const int c = g * 64 + lane;
                xv[u][g] = __builtin_nontemporal_load(xr + (c < d ? c : d - 1));      // non-temporal: see mstep_segment16_kernel
            }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int l = (int)(key[u] >> 32);
            const long long row = (long long)(key[u] & 0xffffffffull);
            if ((unsigned)l >= (unsigned)k) continue;        // sentinel bucket (invalid labels) / past the end
            if (l != cur) {
                flush();
                cur = l;
                run = 0;
#pragma unroll
                for (int g = 0; g < MAXG; ++g) {
                    acc[g] = 0.0;
                    const int c = g * 64 + lane;
                    co[g] = (Cold && c < d) ? Cold[(size_t)l * d + c] : 0.f;
                }
            }
            ++run;
            double a = 0.0;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                if (g * 64 + lane < d) {
                    acc[g] += (double)xv[u][g];
                    const double df = (double)xv[u][g] - (double)co[g];
                    a = fma(df, df, a);
                }
            }
            if (row < split) in0 += a; else in1 += a;
        }
    }
    {   // merged last flush of the block's waves (see mstep_segment16_kernel)
        __shared__ double fsum[4][MAXG * 64];
        __shared__ int flab[4];
        __shared__ long long frun[4];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) fsum[wv][g * 64 + lane] = acc[g];
        if (lane == 0) {
            flab[wv] = (cur >= 0 && cur < k) ? cur : -1;
            frun[wv] = run;
        }
        __syncthreads();
        const int nw = (int)(((n - (long long)blockIdx.x * 4 * MSTEP_ROWS) + MSTEP_ROWS - 1) / MSTEP_ROWS);
        const int lw = nw < 4 ? nw : 4;
        bool leader = flab[wv] >= 0;
        for (int w2 = 0; w2 < wv; ++w2) leader = leader && flab[w2] != flab[wv];
        if (leader) {
            long long tot_run = 0;
#pragma unroll
            for (int g = 0; g < MAXG; ++g) acc[g] = 0.0;
            for (int w2 = wv; w2 < lw; ++w2)
                if (flab[w2] == flab[wv]) {
                    tot_run += frun[w2];
#pragma unroll
                    for (int g = 0; g < MAXG; ++g) acc[g] += fsum[w2][g * 64 + lane];
                }
            run = tot_run;
            flush();
        }
    }
    if (inertia) {
        in0 = wave_sum_f64(in0);
        in1 = wave_sum_f64(in1);
        if (lane == 0) {
            if (in0 != 0.0) atomicAdd(&inertia[0], in0);
            if (in1 != 0.0) atomicAdd(&inertia[1], in1);
        }
    }
}

// The same segment reduction over an fp16 copy of X (scd_f16_exact: every value of X is exactly representable in fp16, as
// features that left an fp16 encoder are): half the row bytes, the float64 sums are bit-identical (fp16 -> float64 is exact
// like float32 -> float64).  Lane owns the column PAIRS 2 lane, 2 lane + 1 (+128, ...): one 4-byte load per 128 columns.
template <int MAXG2, int MSTEP_ROWS, int MU>
__global__ void __launch_bounds__(256) mstep_segment16_kernel(const half_t* __restrict__ X, const unsigned long long* __restrict__ keys,
                                                              const float* __restrict__ Cold, long long n, int d, int k,
                                                              long long split, double* __restrict__ sums,
                                                              unsigned long long* __restrict__ counts, double* __restrict__ inertia) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const long long s0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * MSTEP_ROWS;
    // (a wave past the end of the keys stays alive - empty row range, no run - so that every wave reaches the block barrier below)
    const long long s1 = s0 + MSTEP_ROWS < n ? s0 + MSTEP_ROWS : n;
    double acc[MAXG2][2] = {};
    float co[MAXG2][2] = {};
    double in0 = 0.0, in1 = 0.0;
    int cur = -1;
    long long run = 0;
    auto flush = [&]() {
        if (cur >= 0 && cur < k) {
#pragma unroll
            for (int g = 0; g < MAXG2; ++g) {
                const int c = g * 128 + 2 * lane;
                if (c < d) {
                    atomicAdd(&sums[(size_t)cur * d + c], acc[g][0]);
                    atomicAdd(&sums[(size_t)cur * d + c + 1], acc[g][1]);
                }
            }
            if (lane == 0) atomicAdd(&counts[cur], (unsigned long long)run);
        }
    };
    for (long long sb = s0; sb < s1; sb += MU) {
        unsigned long long key[MU];
        h2 xv[MU][MAXG2];
#pragma unroll
        for (int u = 0; u < MU; ++u) key[u] = sb + u < s1 ? keys[sb + u] : ~0ull;
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const bool live = (unsigned)(key[u] >> 32) < (unsigned)k;
            const long long row = live ? (long long)(key[u] & 0xffffffffull) : 0;
            const half_t* xr = X + row * d;
#pragma unroll
            for (int g = 0; g < MAXG2; ++g) {
                const int c = g * 128 + 2 * lane;
                // non-temporal: this pass must not push the E-step's operand (the prepared fp16 copy, read next) out of the
                // Infinity Cache - the two copies together exceed its 256 MB at C2
                xv[u][g] = __builtin_nontemporal_load((const h2*)(xr + (c < d ? c : d - 2)));   // clamped (always valid) address, masked at use
            }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int l = (int)(key[u] >> 32);
            const long long row = (long long)(key[u] & 0xffffffffull);
            if ((unsigned)l >= (unsigned)k) continue;
            if (l != cur) {
                flush();
                cur = l;
                run = 0;
#pragma unroll
                for (int g = 0; g < MAXG2; ++g) {
                    acc[g][0] = acc[g][1] = 0.0;
                    const int c = g * 128 + 2 * lane;
                    co[g][0] = (Cold && c < d) ? Cold[(size_t)l * d + c] : 0.f;
                    co[g][1] = (Cold && c < d) ? Cold[(size_t)l * d + c + 1] : 0.f;
                }
            }
            ++run;
            double a = 0.0;
#pragma unroll
            for (int g = 0; g < MAXG2; ++g) {
                if (g * 128 + 2 * lane < d) {
                    const double x0 = (double)(float)xv[u][g][0], x1 = (double)(float)xv[u][g][1];
                    acc[g][0] += x0;
                    acc[g][1] += x1;
                    const double d0 = x0 - (double)co[g][0], d1 = x1 - (double)co[g][1];
                    a = fma(d0, d0, a);
                    a = fma(d1, d1, a);
                }
            }
            if (row < split) in0 += a; else in1 += a;
        }
    }
    // The last runs of the block's four waves usually belong to one cluster (256 consecutive sorted rows against ~1,000 per
    // cluster): they are added up in LDS and leave as ONE set of float64 atomics (the kernel's time follows the number of
    // flushes, ~12 ns per 768-column flush).  Wave w flushes for its label unless an earlier wave of the block ends in the same one.
    {
        __shared__ double fsum[4][MAXG2 * 128];
        __shared__ int flab[4];
        __shared__ long long frun[4];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int g = 0; g < MAXG2; ++g) {
            fsum[wv][g * 128 + 2 * lane] = acc[g][0];
            fsum[wv][g * 128 + 2 * lane + 1] = acc[g][1];
        }
        if (lane == 0) {
            flab[wv] = (cur >= 0 && cur < k) ? cur : -1;
            frun[wv] = run;
        }
        __syncthreads();
        const int nw = (int)(((n - (long long)blockIdx.x * 4 * MSTEP_ROWS) + MSTEP_ROWS - 1) / MSTEP_ROWS);
        const int lw = nw < 4 ? nw : 4;
        bool leader = flab[wv] >= 0;
        for (int w2 = 0; w2 < wv; ++w2) leader = leader && flab[w2] != flab[wv];
        if (leader) {
            long long tot_run = 0;
#pragma unroll
            for (int g = 0; g < MAXG2; ++g) acc[g][0] = acc[g][1] = 0.0;
            for (int w2 = wv; w2 < lw; ++w2)
                if (flab[w2] == flab[wv]) {
                    tot_run += frun[w2];
#pragma unroll
                    for (int g = 0; g < MAXG2; ++g) {
                        acc[g][0] += fsum[w2][g * 128 + 2 * lane];
                        acc[g][1] += fsum[w2][g * 128 + 2 * lane + 1];
                    }
                }
            run = tot_run;
            flush();
        }
    }
    if (inertia) {
        in0 = wave_sum_f64(in0);
        in1 = wave_sum_f64(in1);
        if (lane == 0) {
            if (in0 != 0.0) atomicAdd(&inertia[0], in0);
            if (in1 != 0.0) atomicAdd(&inertia[1], in1);
        }
    }
}

// out16 = fp16(X); *inexact_out (device int32) != 0 iff some value does not survive the round trip (0: the copy is exact)
// absmax_bits (may be NULL): max |x| over the values as float bits (non-negative floats order like their bit patterns; +-inf -> 0x7f800000)
// Grid-stride, ONE atomic per block and output: an atomicMax per wave (round 4's first version) was 285,000 same-address atomics at
// 95,000 x 768 - 3.2 ms for a 0.1-ms copy.
__global__ void __launch_bounds__(256) f16_exact_kernel(const float* __restrict__ X, long long n4, half_t* __restrict__ out,
                                                        int* inexact, unsigned* absmax_bits) {
    __shared__ float wmax[4];
    __shared__ int wbad[4];
    bool bad = false;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = *(const float4*)(X + i * 4);
        half4 h;
        h[0] = (half_t)v.x; h[1] = (half_t)v.y; h[2] = (half_t)v.z; h[3] = (half_t)v.w;
        *(half4*)(out + i * 4) = h;
        bad |= (float)h[0] != v.x || (float)h[1] != v.y || (float)h[2] != v.z || (float)h[3] != v.w;     // NaN counts as inexact
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));                // fmaxf drops NaN (counted above)
    }
    const int any_bad = __any(bad) ? 1 : 0;
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = m; wbad[threadIdx.x >> 6] = any_bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (wbad[0] | wbad[1] | wbad[2] | wbad[3]) atomicAdd(inexact, 1);
        if (absmax_bits) atomicMax(absmax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}
static unsigned f16_exact_grid(int64_t n_elems) {
    const long long need = scd_cdiv(n_elems / 4, 256);
    return (unsigned)(need < 2048 ? need : 2048);
}
extern "C" int scd_f16_exact(scd_handle h, const float* X, int64_t n_elems, void* out16, int32_t* inexact_out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_f16_exact");
    SCD_REQUIRE(h && X && out16 && inexact_out && n_elems > 0 && n_elems % 4 == 0, "scd_f16_exact: bad arguments (n_elems % 4 != 0?)");
    SCD_HIP(hipMemsetAsync(inexact_out, 0, 4, (hipStream_t)stream_));
    f16_exact_kernel<<<f16_exact_grid(n_elems), 256, 0, (hipStream_t)stream_>>>(X, n_elems / 4, (half_t*)out16, inexact_out, nullptr);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
extern "C" int scd_f16_exact_max(scd_handle h, const float* X, int64_t n_elems, void* out16, int32_t* inexact_out, float* absmax_out,
                                 void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_f16_exact_max");
    SCD_REQUIRE(h && X && out16 && inexact_out && absmax_out && n_elems > 0 && n_elems % 4 == 0, "scd_f16_exact_max: bad arguments (n_elems % 4 != 0?)");
    SCD_HIP(hipMemsetAsync(inexact_out, 0, 4, (hipStream_t)stream_));
    SCD_HIP(hipMemsetAsync(absmax_out, 0, 4, (hipStream_t)stream_));
    f16_exact_kernel<<<f16_exact_grid(n_elems), 256, 0, (hipStream_t)stream_>>>(X, n_elems / 4, (half_t*)out16, inexact_out,
                                                                                             (unsigned*)absmax_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

static int mstep_impl(scd_handle h, const float* X, const half_t* X16, const int32_t* labels, const float* C_old, int64_t n, int d,
                      int k, int64_t split, double* sums, int64_t* counts, double* inertia, void* ws, size_t ws_bytes, void* stream_);

extern "C" int scd_kmeans_mstep(scd_handle h, const float* X, const int32_t* labels, const float* C_old, int64_t n, int d,
                                int k, int64_t split, double* sums, int64_t* counts, double* inertia, void* ws,
                                size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_mstep");
    SCD_REQUIRE(X, "scd_kmeans_mstep: null X");
    return mstep_impl(h, X, nullptr, labels, C_old, n, d, k, split, sums, counts, inertia, ws, ws_bytes, stream_);
}
extern "C" int scd_kmeans_mstep_f16(scd_handle h, const void* X16, const int32_t* labels, const float* C_old, int64_t n, int d,
                                    int k, int64_t split, double* sums, int64_t* counts, double* inertia, void* ws,
                                    size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_mstep_f16");
    SCD_REQUIRE(X16 && d % 2 == 0, "scd_kmeans_mstep_f16: null X16 or odd d");
    return mstep_impl(h, nullptr, (const half_t*)X16, labels, C_old, n, d, k, split, sums, counts, inertia, ws, ws_bytes, stream_);
}

static int mstep_impl(scd_handle h, const float* X, const half_t* X16, const int32_t* labels, const float* C_old, int64_t n, int d,
                      int k, int64_t split, double* sums, int64_t* counts, double* inertia, void* ws, size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && labels && sums && counts && ws, "scd_kmeans_mstep: null argument");
    SCD_REQUIRE(n > 0 && d > 0 && d <= 1024 && k > 0 && n < (1ll << 32), "scd_kmeans_mstep: bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    SCD_REQUIRE(ws_bytes >= scd_kmeans_mstep_ws_bytes(n, d, k), "scd_kmeans_mstep: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    char* w = (char*)ws;
    unsigned long long* k0 = (unsigned long long*)w; w += scd_align(8 * (size_t)n);
    unsigned long long* k1 = (unsigned long long*)w; w += scd_align(8 * (size_t)n);
    void* temp = w;
    const size_t temp_avail = scd_align(24 * (size_t)n + (8u << 20));
    const int nblk = (int)scd_cdiv(n, MS_BLK);
    if (k <= 8191 && ((size_t)nblk + 1) * (k + 1) * 4 <= temp_avail) {
        int* hist = (int*)temp;
        int* tot = hist + (size_t)nblk * (k + 1);
        const size_t lds = (size_t)(k + 1) * 4;
        mstep_hist_kernel<<<nblk, 256, lds, st>>>(labels, n, k, hist, sums, (long long)k * d, (unsigned long long*)counts, inertia);
        mstep_scan_kernel<<<(unsigned)scd_cdiv(k + 1, 4), 256, 0, st>>>(hist, nblk, k, tot);
        mstep_scatter_kernel<<<nblk, 256, lds, st>>>(labels, n, k, hist, tot, k1);
    } else {
        SCD_HIP(hipMemsetAsync(sums, 0, 8 * (size_t)k * d, st));
        SCD_HIP(hipMemsetAsync(counts, 0, 8 * (size_t)k, st));
        if (inertia) SCD_HIP(hipMemsetAsync(inertia, 0, 16, st));
        mstep_keys_kernel<<<(unsigned)scd_cdiv(n, 256), 256, 0, st>>>(labels, n, k, k0);
        int bits = 1;
        while ((1ll << bits) <= k) ++bits;                       // buckets 0..k
        size_t need = 0;
        SCD_HIP(rocprim::radix_sort_keys(nullptr, need, k0, k1, (size_t)n, 32, 32 + bits, st));
        SCD_REQUIRE(need <= temp_avail, "scd_kmeans_mstep: rocprim temp storage %zu > %zu", need, temp_avail);
        SCD_HIP(rocprim::radix_sort_keys(temp, need, k0, k1, (size_t)n, 32, 32 + bits, st));
    }
    static const int ms_var = getenv("SCD_MSTEP_VAR") ? atoi(getenv("SCD_MSTEP_VAR")) : 3;      // rows per wave / rows in flight: 3 = 64/8 (default), 1 = 64/4, 2 = 32/8, 0 = 32/4
    // few rows (CUB-sized inputs: 4,500 x 768): 64 rows per wave are 18 blocks on a 256-CU chip and one long dependent chain per wave
    // (55 us per launch); 8 rows per wave fill the chip.  The sums are float64 atomics either way (order-free after the one rounding).
    const bool few_rows = scd_cdiv(n, 4 * 64) < 128;
#define MSTEP_LAUNCH(G)                                                                                                      \
    do {                                                                                                                     \
        if (few_rows) mstep_segment_kernel<G, 8, 8><<<(unsigned)scd_cdiv(n, 4 * 8), 256, 0, st>>>(X, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia); \
        else if (ms_var == 1) mstep_segment_kernel<G, 64, 4><<<(unsigned)scd_cdiv(n, 4 * 64), 256, 0, st>>>(X, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia); \
        else if (ms_var == 2) mstep_segment_kernel<G, 32, 8><<<(unsigned)scd_cdiv(n, 4 * 32), 256, 0, st>>>(X, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia); \
        else if (ms_var == 3) mstep_segment_kernel<G, 64, 8><<<(unsigned)scd_cdiv(n, 4 * 64), 256, 0, st>>>(X, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia); \
        else mstep_segment_kernel<G, 32, 4><<<(unsigned)scd_cdiv(n, 4 * 32), 256, 0, st>>>(X, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia); \
    } while (0)
    if (X16) {
#define MSTEP16_LAUNCH(G2) mstep_segment16_kernel<G2, 64, 8><<<(unsigned)scd_cdiv(n, 4 * 64), 256, 0, st>>>(X16, k1, C_old, n, d, k, split, sums, (unsigned long long*)counts, inertia)
        if (d <= 128) MSTEP16_LAUNCH(1);
        else if (d <= 256) MSTEP16_LAUNCH(2);
        else if (d <= 512) MSTEP16_LAUNCH(4);
        else if (d <= 768) MSTEP16_LAUNCH(6);
        else MSTEP16_LAUNCH(8);
#undef MSTEP16_LAUNCH
        SCD_LAUNCH_CHECK();
        return SCD_OK;
    }
    if (d <= 64) MSTEP_LAUNCH(1);
    else if (d <= 128) MSTEP_LAUNCH(2);
    else if (d <= 256) MSTEP_LAUNCH(4);
    else if (d <= 512) MSTEP_LAUNCH(8);
    else if (d <= 768) MSTEP_LAUNCH(12);
    else MSTEP_LAUNCH(16);
#undef MSTEP_LAUNCH
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
