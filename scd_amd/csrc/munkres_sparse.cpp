// scd_munkres_sparse: the assignment step of assign_name (local_utils/clip_lang_util.py:167-178),
//     ind = linear_assignment(w.max() - w),
// for the D x D vote matrix w given by its non-zero entries, without ever building the D x D matrix.
//
// Why: D = max(#voted names, #clusters) reaches 10,000-20,000 at BASELINE configs[3] (K = 1000 clusters x num_common_vote
// 10-20), while w has at most num_common_linear non-zeros in each of the K cluster rows.  The dense state machine
// (munkres.cpp, the reference's cluster_utils.py:316-493) is O(D^3) there, and the matrix alone is 3.2 GB.
//
// The DECISIONS are those of the reference state machine, unchanged: zeros are starred in row-major order (step 2), step 4
// always primes the row-major-first uncovered zero, step 6 adds the smallest uncovered value to covered rows and subtracts
// it from uncovered columns.  Only the bookkeeping differs:
//   * the matrix is implicit: C[i][j] = a[i] + dev(i,j) - v[j] with dev(i,j) = -w[i][j] (0 on the "background"), a[i] the row
//     potential (row reduction, then +minval whenever the row is covered in a step 6) and v[j] the column potential
//     (+minval whenever the column is uncovered in a step 6): a step 6 costs O(D), not O(D^2);
//   * a background zero of row i is a column with v[j] == a[i]: uncovered columns are kept in ordered sets keyed by their
//     potential, rows without entries in classes keyed by theirs, so "the first uncovered zero in row-major order" is the
//     minimum over the active classes' first uncovered rows and the (few) rows with entries, which are tracked one by one.
// tests/test_cpu_abi_and_host.py compares it with the dense solver and the oracle on random matrices with heavy ties,
// tests/golden/munkres.npz holds vote-shaped cases solved by the reference's own linear_assignment.
#include "common.h"
#include <algorithm>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

namespace {
struct SpMunkres {
    int n;                                       // square: n rows, n columns
    std::vector<int64_t> a, v;
    std::vector<int64_t> rptr, cptr;             // CSR / CSC of the entries (dev != 0)
    std::vector<int> rcol, crow;
    std::vector<int64_t> rdev, cdev;
    std::vector<char> ru, cu, has_entries, s_is_active;
    std::vector<int> star_of_row, star_of_col, prime_of_row, sparse_rows;
    int n_unc_rows = 0, n_unc_cols = 0;

    struct RowClass {
        int64_t val;
        std::vector<int> rows;
        size_t ptr;
        bool active;
    };
    std::vector<RowClass> rcs;
    std::unordered_map<int64_t, int> rc_of_val;
    std::set<std::pair<int, int>> act;                           // (first uncovered row, class) of the active classes
    std::unordered_map<int64_t, std::set<int>> colsets;          // v[j] - off -> uncovered columns
    int64_t off = 0;
    std::set<int> s_active;                                      // uncovered rows with entries that have an uncovered zero
    std::unordered_map<int64_t, std::vector<int>> s_inactive;    // a[r] -> such rows without one (background part only)

    bool has_bg(int i) const { return rptr[i + 1] - rptr[i] < n; }
    bool in_row(int i, int j) const {
        const int* b = rcol.data() + rptr[i];
        const int* e = rcol.data() + rptr[i + 1];
        return std::binary_search(b, e, j);
    }
    // first uncovered zero column of a row with entries, -1 if none
    int first_zero(int i) const {
        int best = -1;
        for (int64_t e = rptr[i]; e < rptr[i + 1]; ++e) {
            const int j = rcol[e];
            if (cu[j] && a[i] + rdev[e] - v[j] == 0) { best = j; break; }
        }
        if (has_bg(i)) {
            auto it = colsets.find(a[i] - off);
            if (it != colsets.end())
                for (int j : it->second) {
                    if (best >= 0 && j >= best) break;
                    if (!in_row(i, j)) { best = j; break; }
                }
        }
        return best;
    }
    void advance(RowClass& c) {
        while (c.ptr < c.rows.size() && !ru[c.rows[c.ptr]]) ++c.ptr;
    }
    void refresh_sparse() {
        s_active.clear();
        s_inactive.clear();
        for (int r : sparse_rows) {
            s_is_active[r] = 0;
            if (!ru[r]) continue;
            if (first_zero(r) >= 0) {
                s_is_active[r] = 1;
                s_active.insert(r);
            } else if (has_bg(r)) {
                s_inactive[a[r]].push_back(r);
            }
        }
    }
    void refresh_classes() {
        act.clear();
        for (size_t c = 0; c < rcs.size(); ++c) {
            advance(rcs[c]);
            auto it = colsets.find(rcs[c].val - off);
            rcs[c].active = rcs[c].ptr < rcs[c].rows.size() && it != colsets.end() && !it->second.empty();
            if (rcs[c].active) act.insert({rcs[c].rows[rcs[c].ptr], (int)c});
        }
    }
    void uncover_col(int sc) {
        cu[sc] = 1;
        ++n_unc_cols;
        std::set<int>& S = colsets[v[sc] - off];
        const bool was_empty = S.empty();
        S.insert(sc);
        if (was_empty) {
            auto it = rc_of_val.find(v[sc]);
            if (it != rc_of_val.end()) {
                RowClass& c = rcs[it->second];
                advance(c);
                if (!c.active && c.ptr < c.rows.size()) {
                    c.active = true;
                    act.insert({c.rows[c.ptr], it->second});
                }
            }
        }
        auto si = s_inactive.find(v[sc]);
        if (si != s_inactive.end()) {
            std::vector<int>& lst = si->second;
            size_t k = 0;
            for (size_t t = 0; t < lst.size(); ++t) {
                const int r = lst[t];
                if (!ru[r] || s_is_active[r]) continue;
                if (!in_row(r, sc)) {
                    s_is_active[r] = 1;
                    s_active.insert(r);
                } else {
                    lst[k++] = r;
                }
            }
            lst.resize(k);
        }
        for (int64_t e = cptr[sc]; e < cptr[sc + 1]; ++e) {
            const int r = crow[e];
            if (ru[r] && !s_is_active[r] && a[r] + cdev[e] - v[sc] == 0) {
                s_is_active[r] = 1;
                s_active.insert(r);
            }
        }
    }
    void cover_row(int r, int cls) {
        ru[r] = 0;
        --n_unc_rows;
        if (cls < 0) {
            s_active.erase(r);
            s_is_active[r] = 0;
        } else {
            RowClass& c = rcs[cls];
            act.erase({r, cls});
            advance(c);
            if (c.ptr < c.rows.size()) act.insert({c.rows[c.ptr], cls});
            else c.active = false;
        }
    }

    SpMunkres(int d, int64_t nnz, const int32_t* rows, const int32_t* cols, const int64_t* vals) : n(d) {
        // merge duplicates (w[i, col] += v), drop zeros, sort by (row, col)
        std::vector<std::pair<std::pair<int, int>, int64_t>> ent;
        ent.reserve((size_t)nnz);
        for (int64_t e = 0; e < nnz; ++e) ent.push_back({{rows[e], cols[e]}, vals[e]});
        std::sort(ent.begin(), ent.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        std::vector<std::pair<std::pair<int, int>, int64_t>> m2;
        for (auto& x : ent) {
            if (!m2.empty() && m2.back().first == x.first) m2.back().second += x.second;
            else m2.push_back(x);
        }
        rptr.assign(n + 1, 0);
        cptr.assign(n + 1, 0);
        for (auto& x : m2)
            if (x.second != 0) {
                ++rptr[x.first.first + 1];
                ++cptr[x.first.second + 1];
            }
        for (int i = 0; i < n; ++i) {
            rptr[i + 1] += rptr[i];
            cptr[i + 1] += cptr[i];
        }
        rcol.resize(rptr[n]);
        rdev.resize(rptr[n]);
        crow.resize(rptr[n]);
        cdev.resize(rptr[n]);
        std::vector<int64_t> rp(rptr.begin(), rptr.end() - 1), cp(cptr.begin(), cptr.end() - 1);
        for (auto& x : m2)
            if (x.second != 0) {
                const int i = x.first.first, j = x.first.second;
                rcol[rp[i]] = j;
                rdev[rp[i]++] = -x.second;               // C0 = w.max() - w: the constant cancels in the row reduction
                crow[cp[j]] = i;
                cdev[cp[j]++] = -x.second;
            }
        a.assign(n, 0);
        v.assign(n, 0);
        ru.assign(n, 1);
        cu.assign(n, 1);
        has_entries.assign(n, 0);
        s_is_active.assign(n, 0);
        star_of_row.assign(n, -1);
        star_of_col.assign(n, -1);
        prime_of_row.assign(n, -1);
        for (int i = 0; i < n; ++i)
            if (rptr[i + 1] > rptr[i]) {
                has_entries[i] = 1;
                sparse_rows.push_back(i);
            }
    }

    void solve() {
        // step 1 (row reduction, :362-366) and step 2 (star zeros greedily in row-major order, :367-377)
        for (int i = 0; i < n; ++i) {
            int64_t mn = has_bg(i) ? 0 : INT64_MAX;
            for (int64_t e = rptr[i]; e < rptr[i + 1]; ++e) mn = std::min(mn, rdev[e]);
            a[i] = -mn;
        }
        {
            std::vector<char> used(n, 0);
            int p = 0;
            for (int i = 0; i < n; ++i) {
                int best = -1;
                for (int64_t e = rptr[i]; e < rptr[i + 1]; ++e)
                    if (a[i] + rdev[e] == 0 && !used[rcol[e]]) { best = rcol[e]; break; }
                if (has_bg(i) && a[i] == 0) {
                    while (p < n && used[p]) ++p;
                    int j = p;
                    while (j < n && (used[j] || (has_entries[i] && in_row(i, j)))) ++j;
                    if (j < n && (best < 0 || j < best)) best = j;
                }
                if (best >= 0) {
                    star_of_row[i] = best;
                    star_of_col[best] = i;
                    used[best] = 1;
                }
            }
        }
        for (;;) {
            // _clear_covers + step 3 (:383-394)
            std::fill(ru.begin(), ru.end(), 1);
            std::fill(prime_of_row.begin(), prime_of_row.end(), -1);
            n_unc_rows = n;
            n_unc_cols = 0;
            int stars = 0;
            off = 0;
            colsets.clear();
            for (int j = 0; j < n; ++j) {
                cu[j] = star_of_col[j] < 0;
                if (cu[j]) {
                    ++n_unc_cols;
                    colsets[v[j]].insert(j);
                } else {
                    ++stars;
                }
            }
            if (stars >= n) return;
            rcs.clear();
            rc_of_val.clear();
            for (int i = 0; i < n; ++i)
                if (!has_entries[i]) {
                    auto it = rc_of_val.find(a[i]);
                    if (it == rc_of_val.end()) {
                        it = rc_of_val.emplace(a[i], (int)rcs.size()).first;
                        rcs.push_back(RowClass{a[i], {}, 0, false});
                    }
                    rcs[it->second].rows.push_back(i);
                }
            refresh_classes();
            refresh_sparse();
            // step 4 / step 6 until a primed zero without a star in its row turns up (:397-434, :481-493)
            int z0r = -1, z0c = -1;
            while (z0r < 0) {
                int r = -1, cls = -1;
                if (!act.empty()) {
                    r = act.begin()->first;
                    cls = act.begin()->second;
                }
                if (!s_active.empty() && (r < 0 || *s_active.begin() < r)) {
                    r = *s_active.begin();
                    cls = -1;
                }
                if (r >= 0) {
                    const int q = cls < 0 ? first_zero(r) : *colsets[rcs[cls].val - off].begin();
                    prime_of_row[r] = q;
                    const int sc = star_of_row[r];
                    if (sc < 0) {
                        z0r = r;
                        z0c = q;
                        break;
                    }
                    cover_row(r, cls);
                    uncover_col(sc);
                    continue;
                }
                // step 6
                if (n_unc_rows == 0 || n_unc_cols == 0) return;      // nothing can change (the reference would spin here)
                std::vector<int64_t> keys;
                for (auto& kv : colsets)
                    if (!kv.second.empty()) keys.push_back(kv.first);
                std::sort(keys.begin(), keys.end(), [](int64_t x, int64_t y) { return x > y; });
                const int64_t vmax = keys[0] + off;
                int64_t mv = INT64_MAX;
                for (auto& c : rcs) {
                    advance(c);
                    if (c.ptr < c.rows.size()) mv = std::min(mv, c.val - vmax);
                }
                for (int r2 : sparse_rows) {
                    if (!ru[r2]) continue;
                    for (int64_t e = rptr[r2]; e < rptr[r2 + 1]; ++e)
                        if (cu[rcol[e]]) mv = std::min(mv, a[r2] + rdev[e] - v[rcol[e]]);
                    if (has_bg(r2))
                        for (int64_t key : keys) {
                            size_t mine = 0;
                            for (int64_t e = rptr[r2]; e < rptr[r2 + 1]; ++e)
                                if (cu[rcol[e]] && v[rcol[e]] - off == key) ++mine;
                            if (colsets[key].size() > mine) {
                                mv = std::min(mv, a[r2] - (key + off));
                                break;
                            }
                        }
                }
                for (int i = 0; i < n; ++i)
                    if (!ru[i]) a[i] += mv;
                for (int j = 0; j < n; ++j)
                    if (cu[j]) v[j] += mv;
                off += mv;
                refresh_classes();
                refresh_sparse();
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
            for (size_t i = 1; i < path.size(); i += 2) {          // starred zeros of the series -> unstarred
                const int r = path[i].first, c = path[i].second;
                if (star_of_col[c] == r) star_of_col[c] = -1;
                if (star_of_row[r] == c) star_of_row[r] = -1;
            }
            for (size_t i = 0; i < path.size(); i += 2) {          // primed zeros -> starred
                const int r = path[i].first, c = path[i].second;
                star_of_row[r] = c;
                star_of_col[c] = r;
            }
        }
    }
};
}  // namespace

extern "C" int scd_munkres_sparse(int d, int64_t nnz, const int32_t* rows, const int32_t* cols, const int64_t* vals,
                                  int64_t* pairs_out, int* n_pairs_out) {
    SCD_REQUIRE(n_pairs_out && d >= 0 && nnz >= 0, "scd_munkres_sparse: bad arguments");
    if (d == 0) {
        *n_pairs_out = 0;
        return SCD_OK;
    }
    SCD_REQUIRE(pairs_out && (nnz == 0 || (rows && cols && vals)), "scd_munkres_sparse: null array");
    for (int64_t e = 0; e < nnz; ++e)
        SCD_REQUIRE(rows[e] >= 0 && rows[e] < d && cols[e] >= 0 && cols[e] < d, "scd_munkres_sparse: entry %lld out of range",
                    (long long)e);
    SpMunkres s(d, nnz, rows, cols, vals);
    s.solve();
    int k = 0;
    for (int i = 0; i < d; ++i)
        if (s.star_of_row[i] >= 0) {
            pairs_out[2 * k] = i;
            pairs_out[2 * k + 1] = s.star_of_row[i];
            ++k;
        }
    *n_pairs_out = k;
    return SCD_OK;
}
