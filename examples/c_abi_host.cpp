// A host program on the C ABI alone (no torch, no Python): what a C++ caller of the hot path links against.
//   hipcc --offload-arch=gfx950 -O2 -I include -o /tmp/c_abi_host examples/c_abi_host.cpp -L scd_amd/lib -lscd_hip -lpthread -Wl,-rpath,$PWD/scd_amd/lib
// It runs the full-vocabulary similarity + top-k (main_unsup.py:504-531) and one k-means E-step (faster_mix_k_means_pytorch.py:
// 139-141) on random data and checks both against plain float64 loops on the host; then a restart's Lloyd loop
// (faster_mix_k_means_pytorch.py:187-214) twice - once over all rows (scd_kmeans_lloyd_run) and once over two ROW SHARDS driven by two
// host threads, two handles and two streams (scd_kmeans_lloyd_run_sharded), whose exchange callback adds the two shards' packed sums on
// the host - and checks that the sharded run reproduces the single one bit for bit; the k-means++ rounds of two restarts get the same treatment (scd_kpp_seed_lockstep against
// scd_kpp_seed_lockstep_sharded with a host-side all-gather).  Exit code 0 = equal.
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

// ---- the all-gather of the sharded seeding rounds for two shards in one process: shard w's bytes land at recv + w * bytes on both
struct TwoWayGather {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> stage;
    int arrived = 0;
    long generation = 0;
    int rank_of_thread[2] = {0, 1};
};
struct GatherCtx { TwoWayGather* g; int rank; };
static int two_way_gather(void* ctx, const void* send, void* recv, int64_t bytes, void* stream) {
    GatherCtx* c = (GatherCtx*)ctx;
    TwoWayGather* t = c->g;
    std::vector<char> mine((size_t)bytes), all;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    if (hipMemcpy(mine.data(), send, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    {
        std::unique_lock<std::mutex> lock(t->mu);
        if (t->arrived == 0) t->stage.assign(2 * (size_t)bytes, 0);
        memcpy(t->stage.data() + (size_t)c->rank * bytes, mine.data(), (size_t)bytes);
        if (t->arrived == 0) {
            t->arrived = 1;
            const long g = t->generation;
            if (!t->cv.wait_for(lock, std::chrono::seconds(60), [&] { return t->generation != g; })) return 1;
        } else {
            t->arrived = 0;
            ++t->generation;
            t->cv.notify_all();
        }
        all = t->stage;          // (the first arriver copies after the second has filled its half; nobody clears it before both left:
    }                            //  the next gather's first arriver re-assigns the buffer only after taking the lock again)
    return hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess &&
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
    // ---- the k-means++ rounds of two restarts: all rows behind scd_kpp_seed_lockstep, and the same two shards behind
    // scd_kpp_seed_lockstep_sharded with a host-side all-gather as the callback (sskm_constrained.py:28-44 for every restart in lock-step)
    const int RS = 2, KS = 6, TS = KS - 1;
    const int64_t first[2] = {17, 4211};                       // the restarts' first centres (global rows; the second lies in shard 1)
    std::vector<float> uni(TS * RS), cfirst(RS * dl);
    for (auto& u : uni) u = 0.5f * (rnd() + 1.0f);              // the uniforms a RandomState would supply, [round][restart]
    for (int r = 0; r < RS; ++r)
        for (int j = 0; j < dl; ++j) cfirst[r * dl + j] = XL[first[r] * dl + j];
    float *dU, *dCf;
    HIP(hipMalloc((void**)&dU, uni.size() * 4));
    HIP(hipMemcpy(dU, uni.data(), uni.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMalloc((void**)&dCf, cfirst.size() * 4));
    HIP(hipMemcpy(dCf, cfirst.data(), cfirst.size() * 4, hipMemcpyHostToDevice));
    struct Seed { float *d2 = nullptr, *cbuf = nullptr; int64_t* picks = nullptr; void *ws = nullptr, *xbuf = nullptr; size_t nws = 0, nx = 0; int rc = 0; };
    auto seed_setup = [&](Seed& q, Shard& s, bool sharded) -> int {
        HIP(hipMalloc((void**)&q.d2, (size_t)RS * s.n * 4));
        std::vector<float> inf((size_t)RS * s.n, INFINITY);
        HIP(hipMemcpy(q.d2, inf.data(), inf.size() * 4, hipMemcpyHostToDevice));
        CHECK(scd_kmeans_min_update_multi(s.h, s.X, dCf, s.n, dl, RS, q.d2, s.n, s.st));          // distances to the first centres
        HIP(hipMalloc((void**)&q.cbuf, (size_t)RS * KS * dl * 4));
        for (int r = 0; r < RS; ++r) HIP(hipMemcpy(q.cbuf + (size_t)r * KS * dl, dCf + (size_t)r * dl, dl * 4, hipMemcpyDeviceToDevice));
        HIP(hipMalloc((void**)&q.picks, (size_t)TS * RS * 8));
        q.nws = sharded ? scd_kpp_seed_sharded_ws_bytes(s.n, dl, RS) : scd_kpp_seed_ws_bytes(s.n, dl, RS);
        HIP(hipMalloc(&q.ws, q.nws));
        if (sharded) { q.nx = scd_kpp_seed_sharded_xbuf_bytes(dl, RS, 2); HIP(hipMalloc(&q.xbuf, q.nx)); }
        HIP(hipStreamSynchronize(s.st));
        return 0;
    };
    Seed sa, ss[2];
    if (seed_setup(sa, all, false) || seed_setup(ss[0], sh[0], true) || seed_setup(ss[1], sh[1], true)) return 1;
    CHECK(scd_kpp_seed_lockstep(h, all.X, all.X16, nl, dl, RS, sa.d2, nl, dU, TS, sa.cbuf, KS, 1, sa.picks, sa.ws, sa.nws, all.st));
    HIP(hipStreamSynchronize(all.st));
    TwoWayGather tg;
    GatherCtx gc[2] = {{&tg, 0}, {&tg, 1}};
    auto run_seed = [&](int w) {
        ss[w].rc = scd_kpp_seed_lockstep_sharded(sh[w].h, sh[w].X, sh[w].X16, sh[w].n, dl, RS, ss[w].d2, sh[w].n, dU, TS, ss[w].cbuf, KS, 1,
                                                 ss[w].picks, ss[w].ws, ss[w].nws, sh[w].st, ss[w].xbuf, ss[w].nx, two_way_gather, &gc[w], w, 2);
        hipStreamSynchronize(sh[w].st);
    };
    std::thread u0(run_seed, 0), u1(run_seed, 1);
    u0.join();
    u1.join();
    if (ss[0].rc || ss[1].rc) { fprintf(stderr, "scd_kpp_seed_lockstep_sharded failed: %d %d %s\n", ss[0].rc, ss[1].rc, scd_last_error()); return 1; }
    std::vector<float> ca2((size_t)RS * KS * dl), c0s(ca2.size()), c1s(ca2.size());
    HIP(hipMemcpy(ca2.data(), sa.cbuf, ca2.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(c0s.data(), ss[0].cbuf, ca2.size() * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(c1s.data(), ss[1].cbuf, ca2.size() * 4, hipMemcpyDeviceToHost));
    int bad4 = 0;
    for (size_t i = 0; i < ca2.size(); ++i) bad4 += memcmp(&ca2[i], &c0s[i], 4) != 0 || memcmp(&ca2[i], &c1s[i], 4) != 0;
    printf("scd_kpp_seed_lockstep_sharded: 2 shards vs one call, %d restarts x %d centres: %d seed differences\n", RS, KS, bad4);
    CHECK(scd_destroy(sh[0].h));
    CHECK(scd_destroy(sh[1].h));
    CHECK(scd_destroy(h));
    return (bad || bad2 || bad3 || bad4) ? 2 : 0;
}
