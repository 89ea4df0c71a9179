// Per-cluster name vote histogram for gfx950.
//
// Replaces the Python of main_unsup.py:573-586 / main_ptsup.py:636-648 (paths under /root/reference):
//   cluster_to_counter[i] = Counter(x for x in name_idx_top5[u_preds==i, :top_k].view(-1) [if x not in known])
//   cluster_to_counter[i].most_common(m)
// Counter.most_common(m) = sort by count descending, ties in first-seen (row-major) order.  Here: one 64-bit radix
// sort groups (cluster, name) runs with their first-seen position, a second sort orders the runs of every cluster by
// (count desc, first-seen asc); rocPRIM supplies the device radix sort.
#include "common.h"
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#define VOTE_MAX_CLUSTER_ID 65536
static const unsigned long long SENT = ~0ull;

extern "C" size_t scd_vote_hist_ws_bytes(int64_t n, int top_k) {
    const size_t e = (size_t)n * top_k;
    return scd_align(VOTE_MAX_CLUSTER_ID * 4) + 4 * scd_align(e * 8) + 2 * scd_align(e * 4) + scd_align(e * 24 + (8u << 20)) + 256;
}

__global__ void slot_kernel(const long long* clusters, int nc, int* slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nc) {
        const long long c = clusters[i];
        if (c >= 0 && c < VOTE_MAX_CLUSTER_ID) slot_of[c] = i;
    }
}

__global__ void __launch_bounds__(256) vote_keys_kernel(const long long* __restrict__ name_idx, long long n, int ld, int top_k,
                                                        const long long* __restrict__ preds, const int* __restrict__ slot_of,
                                                        const long long* __restrict__ known, int n_known,
                                                        unsigned long long* keys) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * top_k) return;
    const long long i = e / top_k;
    const int j = (int)(e % top_k);
    const long long p = preds[i];
    const long long name = name_idx[i * ld + j];
    int slot = -1;
    if (p >= 0 && p < VOTE_MAX_CLUSTER_ID) slot = slot_of[p];
    bool drop = slot < 0 || name < 0 || name >= (1ll << 24);
    for (int q = 0; q < n_known && !drop; ++q) drop = known[q] == name;
    keys[e] = drop ? SENT : ((unsigned long long)slot << 48) | ((unsigned long long)name << 24) | (unsigned long long)e;
}

// run heads -> (cluster, 2^24-1-count, first-seen) records; everything else -> sentinel
__global__ void __launch_bounds__(256) vote_runs_kernel(const unsigned long long* __restrict__ keys, long long e_total,
                                                        unsigned long long* rec, unsigned* rec_name) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= e_total) return;
    const unsigned long long k = keys[e];
    unsigned long long out = SENT;
    unsigned nm = 0;
    if (k != SENT && (e == 0 || (keys[e - 1] >> 24) != (k >> 24))) {
        // upper bound of the run: first index whose (cluster,name) prefix differs
        const unsigned long long pre = k >> 24;
        long long lo = e + 1, hi = e_total;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if ((keys[mid] >> 24) == pre) lo = mid + 1; else hi = mid;
        }
        const unsigned long long count = (unsigned long long)(lo - e);
        out = (k & 0xFFFF000000000000ull) | ((0xFFFFFFull - count) << 24) | (k & 0xFFFFFFull);
        nm = (unsigned)((k >> 24) & 0xFFFFFFull);
    }
    rec[e] = out;
    rec_name[e] = nm;
}

__global__ void __launch_bounds__(64) vote_out_kernel(const unsigned long long* __restrict__ rec, const unsigned* __restrict__ rec_name,
                                                      long long e_total, int nc, int m, long long* keys_out, int* counts_out) {
    const int slot = blockIdx.x;
    if (slot >= nc) return;
    // lower bound of this cluster's records
    const unsigned long long target = (unsigned long long)slot << 48;
    long long lo = 0, hi = e_total;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (rec[mid] < target) lo = mid + 1; else hi = mid;
    }
    for (int j = threadIdx.x; j < m; j += 64) {
        long long key = -1;
        int cnt = 0;
        const long long p = lo + j;
        if (p < e_total) {
            const unsigned long long r = rec[p];
            if (r != SENT && (r >> 48) == (unsigned long long)slot) {
                key = rec_name[p];
                cnt = (int)(0xFFFFFFull - ((r >> 24) & 0xFFFFFFull));
            }
        }
        keys_out[(long long)slot * m + j] = key;
        counts_out[(long long)slot * m + j] = cnt;
    }
}

extern "C" int scd_vote_hist(scd_handle h, const int64_t* name_idx, int64_t n, int ld, int top_k, const int64_t* preds,
                             const int64_t* clusters, int n_clusters, const int64_t* known, int n_known, int m,
                             int64_t* keys_out, int32_t* counts_out, void* ws, size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_vote_hist");
    SCD_REQUIRE(h && name_idx && preds && clusters && keys_out && counts_out && ws, "scd_vote_hist: null argument");
    SCD_REQUIRE(n > 0 && top_k > 0 && top_k <= ld && m > 0, "scd_vote_hist: bad shape");
    SCD_REQUIRE(n_clusters > 0 && n_clusters <= VOTE_MAX_CLUSTER_ID, "scd_vote_hist: bad n_clusters");
    SCD_REQUIRE(n * top_k < (1ll << 24), "scd_vote_hist: n*top_k must be < 2^24");
    SCD_REQUIRE(n_known == 0 || known, "scd_vote_hist: known is null");
    SCD_REQUIRE(ws_bytes >= scd_vote_hist_ws_bytes(n, top_k), "scd_vote_hist: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const long long e = n * top_k;
    char* w = (char*)ws;
    int* slot_of = (int*)w; w += scd_align(VOTE_MAX_CLUSTER_ID * 4);
    unsigned long long* k0 = (unsigned long long*)w; w += scd_align(e * 8);
    unsigned long long* k1 = (unsigned long long*)w; w += scd_align(e * 8);
    unsigned long long* r0 = (unsigned long long*)w; w += scd_align(e * 8);
    unsigned long long* r1 = (unsigned long long*)w; w += scd_align(e * 8);
    unsigned* n0 = (unsigned*)w; w += scd_align(e * 4);
    unsigned* n1 = (unsigned*)w; w += scd_align(e * 4);
    void* temp = w;
    const size_t temp_avail = scd_align(e * 24 + (8u << 20));
    SCD_HIP(hipMemsetAsync(slot_of, 0xFF, VOTE_MAX_CLUSTER_ID * 4, st));
    slot_kernel<<<(n_clusters + 255) / 256, 256, 0, st>>>((const long long*)clusters, n_clusters, slot_of);
    const unsigned g = (unsigned)scd_cdiv(e, 256);
    vote_keys_kernel<<<g, 256, 0, st>>>((const long long*)name_idx, n, ld, top_k, (const long long*)preds, slot_of,
                                        (const long long*)known, n_known, k0);
    size_t need = 0;
    SCD_HIP(rocprim::radix_sort_keys(nullptr, need, k0, k1, (size_t)e, 0, 64, st));
    SCD_REQUIRE(need <= temp_avail, "scd_vote_hist: rocprim temp storage %zu > %zu", need, temp_avail);
    SCD_HIP(rocprim::radix_sort_keys(temp, need, k0, k1, (size_t)e, 0, 64, st));
    vote_runs_kernel<<<g, 256, 0, st>>>(k1, e, r0, n0);
    size_t need2 = 0;
    SCD_HIP(rocprim::radix_sort_pairs(nullptr, need2, r0, r1, n0, n1, (size_t)e, 0, 64, st));
    SCD_REQUIRE(need2 <= temp_avail, "scd_vote_hist: rocprim temp storage %zu > %zu", need2, temp_avail);
    SCD_HIP(rocprim::radix_sort_pairs(temp, need2, r0, r1, n0, n1, (size_t)e, 0, 64, st));
    vote_out_kernel<<<n_clusters, 64, 0, st>>>(r1, n1, e, n_clusters, m, (long long*)keys_out, counts_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}


// ------------------------------------------------------------------------------------------------
// The same histogram for row SHARDS: each rank builds the dense table of ITS rows - counts[c][name] and the first
// position (global row * top_k + column) at which cluster c saw the name - the ranks all-reduce the two tables (sum / min), and
// most_common(m) of every cluster is read off the reduced table: (count desc, first-seen asc), first-seen in (rank, row) = global
// row order, exactly what Counter.most_common gives on the concatenated rows (main_unsup.py:573-586).
__global__ void __launch_bounds__(256) vote_table_kernel(const long long* __restrict__ name_idx, long long n, int ld, int top_k,
                                                         const long long* __restrict__ preds, const int* __restrict__ slot_of, int n_slots,
                                                         long long row_offset, long long v, int* __restrict__ counts,
                                                         unsigned long long* __restrict__ first) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * top_k) return;
    const long long i = e / top_k;
    const int j = (int)(e % top_k);
    const long long p = preds[i];
    const long long name = name_idx[i * ld + j];
    if (p < 0 || p >= n_slots || name < 0 || name >= v) return;
    const int slot = slot_of[p];
    if (slot < 0) return;
    atomicAdd(&counts[(size_t)slot * v + name], 1);
    atomicMin(&first[(size_t)slot * v + name], (unsigned long long)((row_offset + i) * top_k + j));
}

// one block per cluster: m rounds of a block-wide arg-max over key = count << 40 | (2^40 - 1 - first); a taken name's count is zeroed
__global__ void __launch_bounds__(256) vote_table_topm_kernel(int* __restrict__ counts, const unsigned long long* __restrict__ first, long long v,
                                                              int m, long long* __restrict__ keys_out, int* __restrict__ counts_out) {
    __shared__ unsigned long long rk[4];
    __shared__ long long rn[4];
    const int slot = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* cnt = counts + (size_t)slot * v;
    const unsigned long long* fs = first + (size_t)slot * v;
    for (int out = 0; out < m; ++out) {
        unsigned long long bk = 0;
        long long bn = -1;
        for (long long nm = threadIdx.x; nm < v; nm += 256) {
            const int c = cnt[nm];
            if (c > 0) {
                const unsigned long long key = ((unsigned long long)c << 40) | ((1ull << 40) - 1 - (fs[nm] & ((1ull << 40) - 1)));
                if (key > bk) { bk = key; bn = nm; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ok = __shfl_xor(bk, o, 64);
            const long long on = __shfl_xor(bn, o, 64);
            if (ok > bk) { bk = ok; bn = on; }
        }
        if (lane == 0) { rk[wave] = bk; rn[wave] = bn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int w = 0;
            for (int q = 1; q < 4; ++q)
                if (rk[q] > rk[w]) w = q;
            keys_out[(long long)slot * m + out] = rn[w];
            counts_out[(long long)slot * m + out] = rn[w] >= 0 ? (int)(rk[w] >> 40) : 0;
            if (rn[w] >= 0) cnt[rn[w]] = 0;
        }
        __syncthreads();
    }
}

extern "C" int scd_vote_table(scd_handle h, const int64_t* name_idx, int64_t n, int ld, int top_k, const int64_t* preds,
                              const int32_t* slot_of, int n_slots, int64_t row_offset, int64_t v, int n_clusters, int32_t* counts,
                              int64_t* first, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_vote_table");
    // n == 0 (a rank whose shard holds no unlabelled row): empty tables - the caller's all-reduce must still find this rank
    SCD_REQUIRE(counts && first && slot_of && (n == 0 || (name_idx && preds)), "scd_vote_table: null argument");
    SCD_REQUIRE(n >= 0 && top_k > 0 && top_k <= ld && v > 0 && n_clusters > 0 && n_slots > 0, "scd_vote_table: bad shape");
    SCD_REQUIRE((row_offset + n) * top_k < (1ll << 40), "scd_vote_table: global rows x top_k must be < 2^40");
    hipStream_t st = (hipStream_t)stream_;
    SCD_HIP(hipMemsetAsync(counts, 0, (size_t)n_clusters * v * 4, st));
    SCD_HIP(hipMemsetAsync(first, 0x7F, (size_t)n_clusters * v * 8, st));          // "never seen" = 0x7f7f...: above every position, also as a SIGNED int64 (MIN all-reduce)
    if (n > 0)
        vote_table_kernel<<<(unsigned)scd_cdiv(n * top_k, 256), 256, 0, st>>>((const long long*)name_idx, n, ld, top_k, (const long long*)preds, slot_of,
                                                                           n_slots, row_offset, v, counts, (unsigned long long*)first);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_vote_table_topm(scd_handle h, int32_t* counts, const int64_t* first, int n_clusters, int64_t v, int m,
                                   int64_t* keys_out, int32_t* counts_out, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_vote_table_topm");
    SCD_REQUIRE(counts && first && keys_out && counts_out && n_clusters > 0 && v > 0 && m > 0, "scd_vote_table_topm: bad arguments");
    vote_table_topm_kernel<<<n_clusters, 256, 0, (hipStream_t)stream_>>>(counts, (const unsigned long long*)first, v, m, (long long*)keys_out,
                                                                         counts_out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}
