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
