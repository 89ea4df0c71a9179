// A host program on the C ABI alone (no torch, no Python): what a C++ caller of the hot path links against.
//   hipcc --offload-arch=gfx950 -O2 -I include -o /tmp/c_abi_host examples/c_abi_host.cpp -L scd_amd/lib -lscd_hip -lpthread -Wl,-rpath,$PWD/scd_amd/lib
// It runs the full-vocabulary similarity + top-k (main_unsup.py:504-531) and one k-means E-step (faster_mix_k_means_pytorch.py:
// 139-141) on random data and checks both against plain float64 loops on the host; then a restart's Lloyd loop
// (faster_mix_k_means_pytorch.py:187-214) twice - once over all rows (scd_kmeans_lloyd_run) and once over two ROW SHARDS driven by two
// host threads, two handles and two streams (scd_kmeans_lloyd_run_sharded), whose exchange callback adds the two shards' packed sums on
// the host - and checks that the sharded run reproduces the single one bit for bit.  Exit code 0 = equal.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "scd_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        const int rc_ = (call);                                                  \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s failed: %d %s\n", #call, rc_, scd_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)
#define HIP(call)                                                                \
    do {                                                                         \
        const hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));    \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static uint32_t rng_state = 12345u;
static float rnd() {      // uniform in (-1, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}


// ---- the exchange of the sharded Lloyd loop for two shards in one process: each thread hands in its packed [sums | counts] (device),
// both leave with the element-wise sum.  (One process per GPU would pass an all-reduce here: scd_allreduce_centroids has this signature.)
struct TwoWay {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<double> acc;
    int arrived = 0;
    long generation = 0;
};
static int two_way_exchange(void* ctx, double* buf, int64_t n, void* stream) {
    TwoWay* t = (TwoWay*)ctx;
    std::vector<double> mine((size_t)n);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;           // the pack kernel has to be done
    if (hipMemcpy(mine.data(), buf, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::vector<double> total;
    {
        std::unique_lock<std::mutex> lock(t->mu);
        if (t->arrived == 0) {
            t->acc = mine;
            t->arrived = 1;
            const long g = t->generation;
            if (!t->cv.wait_for(lock, std::chrono::seconds(60), [&] { return t->generation != g; })) return 1;   // the other shard never came
        } else {
            for (int64_t i = 0; i < n; ++i) t->acc[i] += mine[i];                    // two terms: a + b == b + a, whoever arrives first
            t->arrived = 0;
            ++t->generation;
            t->cv.notify_all();
        }
        total = t->acc;
    }
    return hipMemcpyAsync(buf, total.data(), (size_t)n * 8, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess &&
                   hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
}

struct Shard {      // one rank's buffers of scd_kmeans_lloyd_run[_sharded]
    scd_handle h = nullptr;
    hipStream_t st = nullptr;
    int64_t n = 0;
    float* X = nullptr;
    void *X16 = nullptr, *prep = nullptr, *wse = nullptr, *wsm = nullptr;
    size_t nwse = 0, nwsm = 0;
    int32_t *lab_ring = nullptr, *lab_prev = nullptr, *best_lab = nullptr;
    float *C_ring = nullptr, *best_C = nullptr;
    double *sums = nullptr, *stats_ring = nullptr, *xbuf = nullptr;
    int64_t* counts = nullptr;
    double result[4] = {0, 0, 0, 0};
    int rc = 0;
};
static int shard_setup(Shard& s, const float* hostX, int64_t n, int d, int k, bool own_handle) {
    s.n = n;
    if (own_handle) { CHECK(scd_create(0, &s.h)); }
    HIP(hipStreamCreate(&s.st));
    const size_t kd = (size_t)k * d;
    HIP(hipMalloc((void**)&s.X, (size_t)n * d * 4));
    HIP(hipMalloc(&s.X16, (size_t)n * d * 2));
    HIP(hipMemcpy(s.X, hostX, (size_t)n * d * 4, hipMemcpyHostToDevice));
    int32_t* inexact;
    float* absmax;
    HIP(hipMalloc((void**)&inexact, 4));
    HIP(hipMalloc((void**)&absmax, 4));
    CHECK(scd_f16_exact_max(s.h, s.X, n * d, s.X16, inexact, absmax, s.st));
    int32_t bad = 1;
    HIP(hipStreamSynchronize(s.st));
    HIP(hipMemcpy(&bad, inexact, 4, hipMemcpyDeviceToHost));
    if (bad) { fprintf(stderr, "rows are not fp16-exact\n"); return 1; }
    const size_t nprep = scd_kmeans_prep_bytes(n, d);
    s.nwse = scd_kmeans_estep_ws_bytes(n, d, k);
    s.nwsm = scd_kmeans_mstep_ws_bytes(n, d, k);
    HIP(hipMalloc(&s.prep, nprep));
    HIP(hipMalloc(&s.wse, s.nwse));
    HIP(hipMalloc(&s.wsm, s.nwsm));
    CHECK(scd_kmeans_prepare(s.h, s.X, n, d, s.prep, s.st));
    HIP(hipMalloc((void**)&s.lab_ring, 3 * (size_t)n * 4));
    HIP(hipMalloc((void**)&s.lab_prev, (size_t)n * 4));
    HIP(hipMemset(s.lab_prev, 0xFF, (size_t)n * 4));                                  // -1: no previous labels
    HIP(hipMalloc((void**)&s.best_lab, (size_t)n * 4));
    HIP(hipMalloc((void**)&s.C_ring, 3 * kd * 4));
    HIP(hipMalloc((void**)&s.best_C, kd * 4));
    HIP(hipMalloc((void**)&s.sums, kd * 8));
    HIP(hipMalloc((void**)&s.counts, (size_t)k * 8));
    HIP(hipMalloc((void**)&s.stats_ring, 10 * 8));
    HIP(hipMemset(s.stats_ring, 0, 10 * 8));
    HIP(hipMalloc((void**)&s.xbuf, (kd + 2 * (size_t)k) * 8));
    HIP(hipStreamSynchronize(s.st));
    return 0;
}

int main() {
    scd_handle h = nullptr;
    CHECK(scd_create(0, &h));
    // ---- similarity + top-k: 300 images x 2,000 names x 512, softmax top-3
    const int64_t n = 300, v = 2000;
    const int d = 512, k = 3;
    std::vector<_Float16> F(n * d), Wt(v * d);
    auto unit_rows = [&](std::vector<_Float16>& m, int64_t rows) {
        for (int64_t r = 0; r < rows; ++r) {
            std::vector<float> t(d);
            double s = 0;
            for (int j = 0; j < d; ++j) { t[j] = rnd(); s += (double)t[j] * t[j]; }
            const float inv = (float)(1.0 / sqrt(s));
            for (int j = 0; j < d; ++j) m[r * d + j] = (_Float16)(t[j] * inv);
        }
    };
    unit_rows(F, n);
    unit_rows(Wt, v);
    void *dF, *dW, *ws;
    int64_t* dIdx;
    float* dVal;
    const size_t nb = scd_sim_topk_ws_bytes(n, d, v, k);
    HIP(hipMalloc(&dF, F.size() * 2));
    HIP(hipMalloc(&dW, Wt.size() * 2));
    HIP(hipMalloc(&ws, nb));
    HIP(hipMalloc((void**)&dIdx, n * k * 8));
    HIP(hipMalloc((void**)&dVal, n * k * 4));
    HIP(hipMemcpy(dF, F.data(), F.size() * 2, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dW, Wt.data(), Wt.size() * 2, hipMemcpyHostToDevice));
    CHECK(scd_sim_topk(h, dF, dW, n, d, v, 100.0f, k, SCD_SIM_SOFTMAX, dIdx, dVal, nullptr, ws, nb, nullptr));
    std::vector<int64_t> idx(n * k);
    std::vector<float> val(n * k);
    HIP(hipMemcpy(idx.data(), dIdx, n * k * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(val.data(), dVal, n * k * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        std::vector<double> logit(v);
        for (int64_t c = 0; c < v; ++c) {
            double s = 0;
            for (int j = 0; j < d; ++j) s = fma((double)(float)F[i * d + j], (double)(float)Wt[c * d + j], s);
            logit[c] = 100.0 * s;
        }
        std::vector<int64_t> order(v);
        for (int64_t c = 0; c < v; ++c) order[c] = c;
        std::partial_sort(order.begin(), order.begin() + k, order.end(),
                          [&](int64_t a, int64_t b) { return logit[a] > logit[b] || (logit[a] == logit[b] && a < b); });
        double mx = logit[order[0]], z = 0;
        for (int64_t c = 0; c < v; ++c) z += exp(logit[c] - mx);
        for (int j = 0; j < k; ++j) {
            if (idx[i * k + j] != order[j]) ++bad;
            const double p = exp(logit[order[j]] - mx) / z;
            if (fabs((double)val[i * k + j] - p) > 1e-4 * p + 1e-7) ++bad;
        }
    }
    printf("scd_sim_topk: %lld images x %lld names, top-%d names + softmax probabilities: %d mismatches\n", (long long)n, (long long)v, k, bad);
    // ---- one k-means E-step: 5,000 rows x 128 dims against 20 centres
    const int64_t m = 5000;
    const int dk = 128, kk = 20;
    std::vector<float> X(m * dk), C(kk * dk);
    for (auto& x : X) x = rnd();
    for (auto& c : C) c = rnd();
    float *dX, *dC;
    void *prep, *wse;
    int32_t* dLab;
    const size_t nprep = scd_kmeans_prep_bytes(m, dk), nwse = scd_kmeans_estep_ws_bytes(m, dk, kk);
    HIP(hipMalloc((void**)&dX, X.size() * 4));
    HIP(hipMalloc((void**)&dC, C.size() * 4));
    HIP(hipMalloc(&prep, nprep));
    HIP(hipMalloc(&wse, nwse));
    HIP(hipMalloc((void**)&dLab, m * 4));
    HIP(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    CHECK(scd_kmeans_prepare(h, dX, m, dk, prep, nullptr));
    CHECK(scd_kmeans_estep(h, dX, prep, dC, m, dk, kk, dLab, nullptr, wse, nwse, nullptr));
    std::vector<int32_t> lab(m);
    HIP(hipMemcpy(lab.data(), dLab, m * 4, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int64_t i = 0; i < m; ++i) {
        double best = INFINITY;
        int bi = 0;
        for (int c = 0; c < kk; ++c) {
            double s = 0;
            for (int j = 0; j < dk; ++j) { const double t = (double)X[i * dk + j] - (double)C[c * dk + j]; s = fma(t, t, s); }
            if (s < best) { best = s; bi = c; }
        }
        if (lab[i] != bi) ++bad2;
    }
    printf("scd_kmeans_estep: %lld rows x %d centres: %d label mismatches\n", (long long)m, kk, bad2);
    // ---- a restart's Lloyd loop over all rows, and over two row shards with a host-side exchange: 6,000 fp16-exact rows x 64 dims, 8 centres
    const int64_t nl = 6000, n0 = 2600;              // shard 0: rows [0, 2600), shard 1: the rest (uneven on purpose)
    const int dl = 64, kl = 8, max_iter = 12;
    std::vector<float> XL(nl * dl), C0(kl * dl);
    for (int64_t i = 0; i < nl; ++i) {                // eight blobs on a 1/256 grid: every value is exactly representable in fp16
        const int c = (int)(i % kl);
        for (int j = 0; j < dl; ++j) XL[i * dl + j] = floorf((0.6f * (((c * 37 + j * 11) % 17) / 8.0f - 1.0f) + 0.25f * rnd()) * 256.0f) / 256.0f;
    }
    for (int c = 0; c < kl; ++c)
        for (int j = 0; j < dl; ++j) C0[c * dl + j] = XL[(int64_t)(c * 701 + 13) * dl + j];       // data rows as initial centres
    float* dC0;
    double* dSq;
    HIP(hipMalloc((void**)&dC0, C0.size() * 4));
    HIP(hipMemcpy(dC0, C0.data(), C0.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMalloc((void**)&dSq, 4 * 8));
    Shard all, sh[2];
    all.h = h;
    if (shard_setup(all, XL.data(), nl, dl, kl, false)) return 1;
    if (shard_setup(sh[0], XL.data(), n0, dl, kl, true) || shard_setup(sh[1], XL.data() + n0 * dl, nl - n0, dl, kl, true)) return 1;
    // the rows' sum of squares (double-double) is a per-FIT global: here simply taken over all rows (ranks would add their shards' pairs)
    CHECK(scd_kmeans_sumsq(h, all.X16, nullptr, nl, dl, 0, dSq, all.st));
    HIP(hipStreamSynchronize(all.st));
    CHECK(scd_kmeans_lloyd_run(h, all.X, all.prep, nl, all.X16, nl, dl, kl, nullptr, all.lab_ring, all.lab_prev, dC0, all.C_ring, all.sums,
                               all.counts, nullptr, nullptr, dSq, all.stats_ring, max_iter, 1e-4, all.best_lab, all.best_C, all.result,
                               all.wse, all.nwse, all.wsm, all.nwsm, all.st));
    HIP(hipStreamSynchronize(all.st));
    TwoWay tw;
    auto run_shard = [&](Shard* s) {
        s->rc = scd_kmeans_lloyd_run_sharded(s->h, s->X, s->prep, s->n, s->X16, s->n, dl, kl, nullptr, s->lab_ring, s->lab_prev, dC0, s->C_ring,
                                             s->sums, s->counts, nullptr, nullptr, dSq, s->stats_ring, max_iter, 1e-4, s->best_lab, s->best_C,
                                             s->result, s->wse, s->nwse, s->wsm, s->nwsm, s->st, s->xbuf, two_way_exchange, &tw);
        hipStreamSynchronize(s->st);
    };
    std::thread t0(run_shard, &sh[0]), t1(run_shard, &sh[1]);
    t0.join();
    t1.join();
    if (sh[0].rc || sh[1].rc) { fprintf(stderr, "scd_kmeans_lloyd_run_sharded failed: %d %d %s\n", sh[0].rc, sh[1].rc, scd_last_error()); return 1; }
    std::vector<int32_t> la(nl), ls(nl);
    std::vector<float> ca(kl * dl), cs0(kl * dl), cs1(kl * dl);
    HIP(hipMemcpy(la.data(), all.best_lab, nl * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ls.data(), sh[0].best_lab, n0 * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ls.data() + n0, sh[1].best_lab, (nl - n0) * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ca.data(), all.best_C, ca.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(cs0.data(), sh[0].best_C, ca.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(cs1.data(), sh[1].best_C, ca.size() * 4, hipMemcpyDeviceToHost));
    int bad3 = 0;
    for (int64_t i = 0; i < nl; ++i) bad3 += la[i] != ls[i];
    for (size_t i = 0; i < ca.size(); ++i) bad3 += memcmp(&ca[i], &cs0[i], 4) != 0 || memcmp(&ca[i], &cs1[i], 4) != 0;
    bad3 += all.result[0] != sh[0].result[0] || all.result[0] != sh[1].result[0] || all.result[1] != sh[0].result[1] || all.result[1] != sh[1].result[1];
    printf("scd_kmeans_lloyd_run_sharded: 2 shards (%lld + %lld rows) vs one run of %lld rows, %d iterations, inertia %.6f: %d differences\n",
           (long long)n0, (long long)(nl - n0), (long long)nl, (int)all.result[1], all.result[0], bad3);
    CHECK(scd_destroy(sh[0].h));
    CHECK(scd_destroy(sh[1].h));
    CHECK(scd_destroy(h));
    return (bad || bad2 || bad3) ? 2 : 0;
}
