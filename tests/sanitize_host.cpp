// Sanitizer driver for the host solvers of libscd_hip.so (munkres.cpp, munkres_sparse.cpp, transport.cpp), built by
// tests/test_cpu_abi_and_host.py::test_host_solvers_under_sanitizers with clang's AddressSanitizer + UndefinedBehaviorSanitizer
// (host compilation only: the GPU sanitizers are not available on this pool).  Random instances, checked against brute force
// (assignment) and against the constraints (transport); any sanitizer report aborts the run.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include "scd_hip.h"

static thread_local char g_err[512];
extern "C" const char* scd_last_error(void) { return g_err; }
void scd_set_error(const char* fmt, ...) {      // api.cpp's definition lives beside the HIP entry points; the solvers only report through it
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }

int main() {
    int fails = 0;
    // dense assignment: optimal total against brute force on small matrices (ties included: values from a small range)
    for (int trial = 0; trial < 300; ++trial) {
        const int n = 1 + rnd() % 6, m = 1 + rnd() % 6;
        std::vector<int64_t> c((size_t)n * m);
        for (auto& x : c) x = rnd() % 7;
        std::vector<int64_t> pairs(2 * (size_t)std::min(n, m));
        int np = 0;
        if (scd_munkres(c.data(), n, m, pairs.data(), &np) != 0 || np != std::min(n, m)) { ++fails; continue; }
        int64_t tot = 0;
        std::vector<int> usedr(n, 0), usedc(m, 0);
        for (int p = 0; p < np; ++p) {
            const int64_t r = pairs[2 * p], col = pairs[2 * p + 1];
            if (r < 0 || r >= n || col < 0 || col >= m || usedr[r]++ || usedc[col]++) { ++fails; break; }
            tot += c[r * m + col];
        }
        // brute force over injections of the smaller side
        const bool rows_small = n <= m;
        const int a = rows_small ? n : m, b = rows_small ? m : n;
        std::vector<int> perm(b);
        std::iota(perm.begin(), perm.end(), 0);
        int64_t best = INT64_MAX;
        do {
            int64_t t = 0;
            for (int i = 0; i < a; ++i) t += rows_small ? c[i * m + perm[i]] : c[perm[i] * m + i];
            best = std::min(best, t);
        } while (std::next_permutation(perm.begin(), perm.end()));
        if (tot != best) ++fails;
    }
    // sparse = dense on the vote-shaped problem linear_assignment(w.max() - w)
    for (int trial = 0; trial < 200; ++trial) {
        const int d = 1 + rnd() % 40;
        const int64_t nnz = rnd() % (3 * d + 1);
        std::vector<int32_t> rows(nnz), cols(nnz);
        std::vector<int64_t> vals(nnz), w((size_t)d * d, 0);
        for (int64_t i = 0; i < nnz; ++i) {
            rows[i] = rnd() % d; cols[i] = rnd() % d; vals[i] = 1 + rnd() % 5;
            w[(size_t)rows[i] * d + cols[i]] += vals[i];
        }
        const int64_t wmax = *std::max_element(w.begin(), w.end());
        std::vector<int64_t> cost((size_t)d * d);
        for (size_t i = 0; i < cost.size(); ++i) cost[i] = wmax - w[i];
        std::vector<int64_t> pd(2 * (size_t)d), ps(2 * (size_t)d);
        int nd = 0, ns = 0;
        if (scd_munkres(cost.data(), d, d, pd.data(), &nd) != 0) { ++fails; continue; }
        if (scd_munkres_sparse(d, nnz, rows.data(), cols.data(), vals.data(), ps.data(), &ns) != 0) { ++fails; continue; }
        if (nd != ns || !std::equal(pd.begin(), pd.begin() + 2 * nd, ps.begin())) ++fails;
    }
    // transport: every point labelled, cluster sizes inside the bounds, infeasible bounds rejected
    for (int trial = 0; trial < 100; ++trial) {
        const int k = 1 + rnd() % 6;
        const int64_t n = k + rnd() % 60;
        const int smin = rnd() % (int)(n / k + 1), smax = (int)((n + k - 1) / k) + rnd() % 5;
        std::vector<int32_t> cost((size_t)n * k), lab(n, -1);
        for (auto& x : cost) x = rnd() % 1000;
        int64_t total = 0;
        const int rc = scd_transport_solve(cost.data(), n, k, smin, smax, lab.data(), &total);
        const bool feasible = (int64_t)smin * k <= n && (int64_t)smax * k >= n;
        if (!feasible) { if (rc == 0) ++fails; continue; }
        if (rc != 0) { ++fails; continue; }
        std::vector<int> cnt(k, 0);
        int64_t t = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (lab[i] < 0 || lab[i] >= k) { ++fails; break; }
            ++cnt[lab[i]];
            t += cost[i * k + lab[i]];
        }
        for (int c = 0; c < k; ++c) if (cnt[c] < smin || cnt[c] > smax) ++fails;
        if (t != total) ++fails;
        // the same problem with every cost moved up by 2^30 takes the 64-bit tables (and the unpacked node selection): the differences
        // are the same, so the labels must be, and the total moves by n * 2^30
        std::vector<int32_t> big(cost), lab2(n, -1);
        for (auto& x : big) x += 1 << 30;
        int64_t total2 = 0;
        if (scd_transport_solve(big.data(), n, k, smin, smax, lab2.data(), &total2) != 0) { ++fails; continue; }
        if (lab2 != lab || total2 != total + n * ((int64_t)1 << 30)) ++fails;
    }
    // wider problems (k past one vector of the row loops, many augmentations): bounds, total, and the batch = single solves
    for (int trial = 0; trial < 6; ++trial) {
        const int k = 17 + rnd() % 120;
        const int64_t n = 6 * k + rnd() % 500;
        const int smin = (int)(n / k) - 1, smax = (int)(n / k) + 2;
        std::vector<int32_t> cost((size_t)n * k), lab(n, -1);
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < k; ++c) cost[i * k + c] = 1000 + (c < k / 4 ? 0 : 3000) + rnd() % 2000;     // a few popular centres
        int64_t total = 0;
        if (scd_transport_solve(cost.data(), n, k, smin, smax, lab.data(), &total) != 0) { ++fails; continue; }
        std::vector<int> cnt(k, 0);
        int64_t t = 0;
        for (int64_t i = 0; i < n; ++i) { ++cnt[lab[i]]; t += cost[i * k + lab[i]]; }
        for (int c = 0; c < k; ++c) if (cnt[c] < smin || cnt[c] > smax) ++fails;
        if (t != total) ++fails;
    }
    // the batch form (the restarts' problems of one iteration on host threads): every problem's labels and total are the single solve's
    for (int trial = 0; trial < 12; ++trial) {
        const int k = 2 + rnd() % 5, batch = 1 + rnd() % 7, threads = 1 + rnd() % 4;
        const int64_t n = 2 * k + rnd() % 80;
        const int smin = (int)(n / k) / 2, smax = (int)((n + k - 1) / k) + 3;
        std::vector<int32_t> cost((size_t)batch * n * k), lab((size_t)batch * n, -1), one(n);
        for (auto& x : cost) x = rnd() % 1000;
        std::vector<int64_t> totals(batch, -1);
        if (scd_transport_solve_batch(cost.data(), n, k, batch, smin, smax, lab.data(), totals.data(), threads) != 0) { ++fails; continue; }
        for (int b = 0; b < batch; ++b) {
            int64_t t = 0;
            if (scd_transport_solve(cost.data() + (size_t)b * n * k, n, k, smin, smax, one.data(), &t) != 0) { ++fails; continue; }
            if (t != totals[b] || !std::equal(one.begin(), one.end(), lab.begin() + (size_t)b * n)) ++fails;
        }
    }
    printf("sanitize_host: %d failures\n", fails);
    return fails ? 1 : 0;
}
