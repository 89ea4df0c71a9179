// A host program on the C ABI alone (no torch, no Python): what a C++ caller of the hot path links against.
//   hipcc --offload-arch=gfx950 -O2 -I include -o /tmp/c_abi_host examples/c_abi_host.cpp -L scd_amd/lib -lscd_hip -Wl,-rpath,$PWD/scd_amd/lib
// It runs the full-vocabulary similarity + top-k (main_unsup.py:504-531) and one k-means E-step (faster_mix_k_means_pytorch.py:
// 139-141) on random data and checks both against plain float64 loops on the host.  Exit code 0 = equal.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
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
    CHECK(scd_destroy(h));
    return (bad || bad2) ? 2 : 0;
}
