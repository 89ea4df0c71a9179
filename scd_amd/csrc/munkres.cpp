// scd_munkres: Kuhn-Munkres assignment with the tie-breaking of the state machine the reference vendors at
// gcd/project_utils/cluster_utils.py:316-493 (zeros are starred / primed in ROW-MAJOR order, which is what makes
// its answers differ from scipy's linear_sum_assignment on tie-heavy vote matrices).  Callers in the reference:
// assign_name (local_utils/clip_lang_util.py:178) and split_cluster_acc_v2 (cluster_and_log_utils.py:53).
//
// Same decisions, different machinery: zero positions live in per-row bitsets so "first uncovered zero in
// row-major order" is a word scan, and the scan start is moved backwards only when a column is uncovered.
#include "common.h"
#include <vector>
#include <algorithm>

namespace {
struct Munkres {
    int n, m, W;
    std::vector<int64_t> C;
    std::vector<uint64_t> Z;        // n x W zero bitsets
    std::vector<uint64_t> cu;       // uncovered columns bitset
    std::vector<char> ru;           // uncovered rows
    std::vector<int> star_of_row, star_of_col, prime_of_row;

    Munkres(const int64_t* cost, int rows, int cols, bool transpose) {
        n = transpose ? cols : rows;
        m = transpose ? rows : cols;
        W = (m + 63) / 64;
        C.resize((size_t)n * m);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) C[(size_t)i * m + j] = transpose ? cost[(size_t)j * cols + i] : cost[(size_t)i * cols + j];
        Z.assign((size_t)n * W, 0);
        cu.assign(W, 0);
        ru.assign(n, 1);
        star_of_row.assign(n, -1);
        star_of_col.assign(m, -1);
        prime_of_row.assign(n, -1);
    }
    void rebuild_zeros() {
        std::fill(Z.begin(), Z.end(), 0);
        for (int i = 0; i < n; ++i) {
            const int64_t* row = &C[(size_t)i * m];
            uint64_t* z = &Z[(size_t)i * W];
            for (int j = 0; j < m; ++j)
                if (row[j] == 0) z[j >> 6] |= 1ull << (j & 63);
        }
    }
    void uncover_all() {
        std::fill(ru.begin(), ru.end(), 1);
        for (int w = 0; w < W; ++w) cu[w] = ~0ull;
        if (m & 63) cu[W - 1] = (1ull << (m & 63)) - 1;
    }
    // first uncovered zero of row i, or -1
    int first_zero(int i) const {
        const uint64_t* z = &Z[(size_t)i * W];
        for (int w = 0; w < W; ++w) {
            uint64_t v = z[w] & cu[w];
            if (v) return (w << 6) + __builtin_ctzll(v);
        }
        return -1;
    }
    void solve() {
        // step 1: row reduction, then star zeros greedily in row-major order (:362-380)
        for (int i = 0; i < n; ++i) {
            int64_t* row = &C[(size_t)i * m];
            int64_t mn = *std::min_element(row, row + m);
            for (int j = 0; j < m; ++j) row[j] -= mn;
        }
        rebuild_zeros();
        {
            std::vector<char> col_used(m, 0);
            for (int i = 0; i < n; ++i) {
                const int64_t* row = &C[(size_t)i * m];
                for (int j = 0; j < m; ++j)
                    if (row[j] == 0 && !col_used[j]) {
                        star_of_row[i] = j;
                        star_of_col[j] = i;
                        col_used[j] = 1;
                        break;
                    }
            }
        }
        uncover_all();
        for (;;) {
            // step 3: cover starred columns; done when n stars (:383-394)
            int stars = 0;
            for (int j = 0; j < m; ++j)
                if (star_of_col[j] >= 0) {
                    cu[j >> 6] &= ~(1ull << (j & 63));
                    ++stars;
                }
            if (stars >= n) return;
            // step 4 / step 6 until an augmenting zero is found (:397-434, :481-493)
            int z0r = -1, z0c = -1;
            while (z0r < 0) {
                int cur = 0;
                for (;;) {
                    int r = -1, q = -1;
                    for (int i = cur; i < n; ++i)
                        if (ru[i]) {
                            int j = first_zero(i);
                            if (j >= 0) { r = i; q = j; break; }
                        }
                    if (r < 0) break;                        // no uncovered zero left -> step 6
                    prime_of_row[r] = q;
                    const int sc = star_of_row[r];
                    if (sc < 0) { z0r = r; z0c = q; break; }  // -> step 5
                    ru[r] = 0;
                    cu[sc >> 6] |= 1ull << (sc & 63);
                    // rows before r had no uncovered zero; uncovering column sc may give them one
                    cur = r + 1;
                    for (int i = 0; i < r; ++i)
                        if (ru[i] && (Z[(size_t)i * W + (sc >> 6)] >> (sc & 63) & 1)) { cur = i; break; }
                }
                if (z0r >= 0) break;
                // step 6: smallest uncovered value; add to covered rows, subtract from uncovered columns
                bool any_r = false, any_c = false;
                for (int i = 0; i < n; ++i) any_r |= ru[i] != 0;
                for (int w = 0; w < W; ++w) any_c |= cu[w] != 0;
                if (any_r && any_c) {
                    int64_t mv = INT64_MAX;
                    for (int i = 0; i < n; ++i)
                        if (ru[i]) {
                            const int64_t* row = &C[(size_t)i * m];
                            for (int j = 0; j < m; ++j)
                                if ((cu[j >> 6] >> (j & 63) & 1) && row[j] < mv) mv = row[j];
                        }
                    for (int i = 0; i < n; ++i) {
                        int64_t* row = &C[(size_t)i * m];
                        const bool cov = !ru[i];
                        for (int j = 0; j < m; ++j) {
                            const bool unc_c = cu[j >> 6] >> (j & 63) & 1;
                            if (cov) row[j] += mv;
                            if (unc_c) row[j] -= mv;
                        }
                    }
                    rebuild_zeros();
                } else {
                    return;   // degenerate: nothing can change (matches the reference looping guard)
                }
            }
            // step 5: alternate primed / starred zeros from Z0, flip them (:437-478)
            std::vector<std::pair<int, int>> path;
            path.emplace_back(z0r, z0c);
            for (;;) {
                const int col = path.back().second;
                const int r = star_of_col[col];
                if (r < 0) break;
                path.emplace_back(r, col);
                path.emplace_back(r, prime_of_row[r]);
            }
            for (size_t i = 0; i < path.size(); ++i) {
                const int r = path[i].first, c = path[i].second;
                if (i & 1) {                      // starred zero -> unstar
                    if (star_of_col[c] == r) star_of_col[c] = -1;
                    if (star_of_row[r] == c) star_of_row[r] = -1;
                }
            }
            for (size_t i = 0; i < path.size(); i += 2) {   // primed zeros -> star
                const int r = path[i].first, c = path[i].second;
                star_of_row[r] = c;
                star_of_col[c] = r;
            }
            uncover_all();
            std::fill(prime_of_row.begin(), prime_of_row.end(), -1);
        }
    }
};
}  // namespace

extern "C" int scd_munkres(const int64_t* cost, int n, int m, int64_t* pairs_out, int* n_pairs_out) {
    SCD_REQUIRE(n_pairs_out && n >= 0 && m >= 0, "scd_munkres: bad arguments");
    if (n == 0 || m == 0) {
        *n_pairs_out = 0;
        return SCD_OK;
    }
    SCD_REQUIRE(cost && pairs_out, "scd_munkres: null matrix");
    const bool transposed = m < n;
    Munkres s(cost, n, m, transposed);
    s.solve();
    std::vector<std::pair<int64_t, int64_t>> res;
    for (int i = 0; i < s.n; ++i)
        if (s.star_of_row[i] >= 0) {
            if (transposed) res.emplace_back(s.star_of_row[i], i);
            else res.emplace_back(i, s.star_of_row[i]);
        }
    std::sort(res.begin(), res.end());
    for (size_t i = 0; i < res.size(); ++i) {
        pairs_out[2 * i] = res[i].first;
        pairs_out[2 * i + 1] = res[i].second;
    }
    *n_pairs_out = (int)res.size();
    return SCD_OK;
}
