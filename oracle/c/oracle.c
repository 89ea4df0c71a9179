/* oracle.c - plain-C restatement of the two all-pairs sweeps of the SCD hot path, used ONLY as a checker and as the
 * timed CPU baseline ("port") of bench.py.  TEST INFRASTRUCTURE: nothing under scd_amd/ links or calls this.
 *
 *  oracle_estep     pairwise_distance + torch.min(dist,1)
 *                   /root/reference/local_utils/sskm_constrained.py:189-224,
 *                   /root/reference/gcd/methods/clustering/faster_mix_k_means_pytorch.py:192
 *                   (difference form, float64 accumulate, ties -> lowest index; same semantics as
 *                   oracle/kmeans_oracle.py:estep, against which tests/test_oracle_c.py checks it)
 *  oracle_sim_topk  logits = scale * F @ W ; top-k   /root/reference/main_unsup.py:504-531
 *                   (float64 accumulate, order = value desc, index asc; checked against naming_oracle.sim_topk)
 *  oracle_mstep     per-cluster mean  /root/reference/local_utils/sskm_constrained.py:125-128
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void oracle_estep(const float* x, const float* c, int64_t n, int d, int k, int64_t* labels, float* mind) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float* xi = x + i * d;
        double best = INFINITY;
        int bi = 0;
        for (int j = 0; j < k; ++j) {
            const float* cj = c + (int64_t)j * d;
            double s = 0.0;
            for (int t = 0; t < d; ++t) {
                const double df = (double)xi[t] - (double)cj[t];
                s += df * df;
            }
            if (s < best) { best = s; bi = j; }
        }
        labels[i] = bi;
        mind[i] = (float)best;
    }
}

void oracle_mstep(const float* x, const int64_t* labels, int64_t n, int d, int k, float* centers) {
    double* sums = (double*)calloc((size_t)k * d, sizeof(double));
    int64_t* cnt = (int64_t*)calloc((size_t)k, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t l = labels[i];
        if (l < 0 || l >= k) continue;
        cnt[l]++;
        for (int t = 0; t < d; ++t) sums[l * d + t] += (double)x[i * d + t];
    }
    for (int j = 0; j < k; ++j)
        for (int t = 0; t < d; ++t) centers[(int64_t)j * d + t] = cnt[j] ? (float)(sums[(int64_t)j * d + t] / (double)cnt[j]) : NAN;
    free(sums);
    free(cnt);
}

/* f [n,d], wt [v,d] (name-major), both float32 holding the stored (fp16-representable) values */
void oracle_sim_topk(const float* f, const float* wt, int64_t n, int d, int64_t v, double scale, int k, int64_t* idx,
                     float* val) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double bv[16];
        int64_t bi[16];
        for (int q = 0; q < k; ++q) { bv[q] = -INFINITY; bi[q] = -1; }
        const float* fi = f + i * d;
        for (int64_t j = 0; j < v; ++j) {
            const float* wj = wt + j * d;
            double s = 0.0;
            for (int t = 0; t < d; ++t) s += (double)fi[t] * (double)wj[t];
            s *= scale;
            if (s > bv[k - 1]) {            /* ascending j: strict > keeps the lower index on ties */
                int q = k - 1;
                while (q > 0 && s > bv[q - 1]) { bv[q] = bv[q - 1]; bi[q] = bi[q - 1]; --q; }
                bv[q] = s;
                bi[q] = j;
            }
        }
        for (int q = 0; q < k; ++q) { idx[i * k + q] = bi[q]; val[i * k + q] = (float)bv[q]; }
    }
}
