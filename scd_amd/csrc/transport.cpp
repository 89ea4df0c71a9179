// scd_transport_solve: the size-constrained assignment of the ConSSKM E-step, i.e. the min-cost-flow problem the
// reference builds at local_utils/sskm_constrained.py:277-328 and hands to OR-Tools SimpleMinCostFlow (:331-356,
// third-party, not in the reference tree).  Input is the dense int32 cost matrix cost[i][j] = round(1000*dist(i,j))
// (the X -> C' arcs); the dummy->centre (cap size_max) and centre->sink (demand size_min) arcs are the bounds.
//
// Algorithm (round 5; rounds 1-4 inserted the points one at a time, 0.46 s at the Stanford-Dogs shape): RELAX, THEN REPAIR.
//   1. Every point goes to its nearest centre.  That is the optimum of the problem without the size bounds, and as a pseudo-flow
//      of the bounded problem it satisfies the reduced-cost optimality conditions with all node potentials zero (every residual arc
//      "move a member of a to b" costs c[p][b] - c[p][a] >= 0).
//   2. What it violates are node balances: a centre with fewer than size_min members has a deficit, one with more than size_max an
//      excess, and the free sink (node G below: it takes the n - k size_min units no centre is obliged to take) has the difference.
//      Successive shortest paths repair them one unit at a time: from an excess node to the nearest deficit node in the residual
//      graph COLLAPSED TO THE k CENTRES + G - arc a -> b costs the cheapest re-assignment of one member of a to b (a dense k x k
//      table; when a centre loses a member, the columns that member was cheapest for are searched again over its member list), arc a -> G exists while a has fewer than
//      size_max members, arc G -> a while a has more than size_min.  Each search is a dense Dijkstra on reduced costs (Johnson
//      potentials, O(k^2)); augmenting along a shortest path keeps the reduced-cost conditions, so when no imbalance is left the flow
//      is a minimum-cost flow (Ahuja / Magnanti / Orlin, "successive shortest path algorithm" started from a pseudo-flow).
// The number of searches is the total imbalance of the nearest-centre assignment (tens at the reference's 50 / 1000 bounds on 9,000
// points) instead of one per point.  The total cost is THE optimum (unique); the labels are one optimum - they need not be OR-Tools'
// on integer cost ties.  Deterministic: lowest index wins every tie (nearest centre, cheapest member, next node of the search).
// Round 6: the same sequence of decisions, laid out for the host's vector units (a solve at 9,000 x 120: 12-14 ms -> ~2 ms on one core).
// The k x k table holds int32 differences when every cost is below 2^30 (always, for round(1000 * distance)); a point's row update, the
// rebuild of a row from its member list, the nearest-centre scan and the relax / select loops of the dense Dijkstra are branch-free loops
// over k that the compiler vectorises (AVX2 / AVX-512 clones picked at load time, the plain build of the same source otherwise: integer
// arithmetic, so every clone computes the same numbers).  A centre that loses a member rebuilds its whole row instead of searching the
// affected columns one at a time: the table is a pure function of the member sets (minimum difference, lowest point index among equals),
// so the values are the ones the column searches produced.
#include "common.h"
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>
#include <exception>
#include <string>

// hipcc parses host code in the device pass as well; function multiversioning exists for the host target only.  Under ThreadSanitizer
// the clones are off: their load-time resolvers run before the sanitizer's runtime is up (tests/test_cpu_abi_and_host.py builds that way)
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define SCD_NO_CLONES 1
#endif
#endif
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__) || defined(SCD_NO_CLONES)
#define SCD_HOST_SIMD
#else
#define SCD_HOST_SIMD __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

// the 64-bit loops over k run in vectors of 8 with no interleaving (the default, 4 x 8 lanes, leaves k = 120 a scalar remainder of 24);
// the 32-bit row updates measured faster as the compiler lays them out by itself
#define SCD_LOOP8 _Pragma("clang loop vectorize_width(8) interleave_count(1)")

namespace {

constexpr int64_t INF64 = INT64_MAX / 4;

template <class T> struct Lim;
template <> struct Lim<int32_t> { static constexpr int32_t INF = INT32_MAX; };
template <> struct Lim<int64_t> { static constexpr int64_t INF = INF64; };

// Selection key of a reached, unsettled node: (distance << 20) | node, so that ONE min-reduction finds the nearest node and the lowest
// index among equals; KEY_NONE for settled / unreached nodes.  Used when no distance can reach 2^42 (transport_solve_one checks).
constexpr int KEY_BITS = 20;
constexpr int64_t KEY_NONE = INT64_MAX;
static inline int64_t pack_key(int64_t d, int v) { return (int64_t)(((uint64_t)d << KEY_BITS) | (uint64_t)v); }
SCD_HOST_SIMD int64_t min_key(const int64_t* __restrict__ key, int nn) {
    int64_t best = KEY_NONE;
    SCD_LOOP8
    for (int v = 0; v < nn; ++v) best = key[v] < best ? key[v] : best;
    return best;
}

// (multiversioned functions cannot be templates: the two table types are overloads stamped out by a macro)
// row_add: row `r` / `rp` of centre a takes in member p (cost row c): r[b] = min over members of c[b] - c[a], rp[b] the lowest member
// attaining it.  relax_row: Dijkstra, leaving a settled centre u: dist[v] = min(dist[v], du + r[v] - pi[v]) for the centres not yet
// settled (reduced cost r[v] + pi[u] - pi[v] >= 0); returns the smallest selection key of the centres after the update (the next
// node to settle, up to G's own key: one pass per settled node).
#define SCD_ROW_FUNCS(T, U)                                                                                                                \
    SCD_HOST_SIMD void row_add(T* __restrict__ r, int32_t* __restrict__ rp, const int32_t* __restrict__ c, int a, int32_t p, int k) {      \
        const T ca = c[a];                                                                                                                 \
        for (int b = 0; b < k; ++b) {                                                                                                      \
            const T dlt = (T)((U)(T)c[b] - (U)ca); /* (wraps instead of overflowing: the optimistic 32-bit pass may meet huge costs) */ \
            const bool m = (dlt < r[b]) | ((dlt == r[b]) & (p < rp[b]));                                                                   \
            r[b] = m ? dlt : r[b];                                                                                                         \
            rp[b] = m ? p : rp[b];                                                                                                         \
        }                                                                                                                                  \
    }                                                                                                                                      \
    SCD_HOST_SIMD int64_t relax_row(int64_t* __restrict__ dist, int64_t* __restrict__ key, int32_t* __restrict__ pred,                     \
                                    const T* __restrict__ r, const int64_t* __restrict__ pi, const int8_t* __restrict__ done, int64_t du,  \
                                    int u, int k) {                                                                                        \
        int64_t best = KEY_NONE;                                                                                                           \
        SCD_LOOP8                                                                                                                          \
        for (int v = 0; v < k; ++v) {                                                                                                      \
            const int64_t nd = du + (int64_t)r[v] - pi[v];                                                                                 \
            const bool m = (nd < dist[v]) & (done[v] == 0) & (v != u);                                                                     \
            const int64_t kv = m ? pack_key(nd, v) : key[v];                                                                               \
            dist[v] = m ? nd : dist[v];                                                                                                    \
            key[v] = kv;                                                                                                                   \
            pred[v] = m ? u : pred[v];                                                                                                     \
            best = kv < best ? kv : best;                                                                                                  \
        }                                                                                                                                  \
        return best;                                                                                                                       \
    }
SCD_ROW_FUNCS(int32_t, uint32_t)
SCD_ROW_FUNCS(int64_t, uint64_t)
#undef SCD_ROW_FUNCS
// nearest centre of one point, lowest index among ties; also the row's largest cost
SCD_HOST_SIMD int row_argmin(const int32_t* __restrict__ c, int k, int32_t* mx_io) {
    int32_t mn = c[0], mx = c[0];
    for (int b = 1; b < k; ++b) { mn = c[b] < mn ? c[b] : mn; mx = c[b] > mx ? c[b] : mx; }
    if (mx > *mx_io) *mx_io = mx;
    int a = 0;
    while (c[a] != mn) ++a;
    return a;
}
// the unsettled node with the smallest finite distance, lowest index among equals (-1: none)
SCD_HOST_SIMD int select_min(const int64_t* __restrict__ dist, const int8_t* __restrict__ done, int nn) {
    int64_t best = INF64;
    for (int v = 0; v < nn; ++v) {
        const int64_t dv = done[v] ? INF64 : dist[v];
        best = dv < best ? dv : best;
    }
    if (best >= INF64) return -1;
    int u = 0;
    while (done[u] || dist[u] != best) ++u;
    return u;
}

template <class T>
struct Solver {
    const int32_t* cost;
    int64_t n;
    int k;
    std::vector<int32_t> assign;            // point -> centre
    std::vector<int32_t> pos;               // point -> position in its centre's member list
    std::vector<std::vector<int32_t>> mem;  // centre -> members
    std::vector<T> best;                    // [k][k]: min over members p of a of c[p][b] - c[p][a]   (INF: a is empty)
    std::vector<int32_t> bestp;             // the member attaining it (lowest index among ties)

    void add(int a, int32_t p) { row_add(best.data() + (size_t)a * k, bestp.data() + (size_t)a * k, cost + (int64_t)p * k, a, p, k); }
    // a member has left a (mem[a] no longer holds it): the row from the members that remain
    void rebuild(int a) {
        T* r = best.data() + (size_t)a * k;
        int32_t* rp = bestp.data() + (size_t)a * k;
        for (int b = 0; b < k; ++b) { r[b] = Lim<T>::INF; rp[b] = -1; }
        const auto& m = mem[a];
        const size_t nm = m.size();
        constexpr size_t AHEAD = 6;                                   // member rows sit at random places of the cost matrix
        for (size_t i = 0; i < nm; ++i) {
            if (i + AHEAD < nm) {
                const char* nx = (const char*)(cost + (int64_t)m[i + AHEAD] * k);
                for (int o = 0; o < k * 4; o += 64) __builtin_prefetch(nx + o);
            }
            row_add(r, rp, cost + (int64_t)m[i] * k, a, m[i], k);
        }
    }
    void attach(int32_t p, int a) {
        assign[p] = a;
        pos[p] = (int32_t)mem[a].size();
        mem[a].push_back(p);
    }
    void detach(int32_t p) {
        const int a = assign[p];
        auto& m = mem[a];
        const int32_t last = m.back();
        m[pos[p]] = last;
        pos[last] = pos[p];
        m.pop_back();
    }
};

// steps 1 and 2 on tables of type T; `a0` = every point's nearest centre (lowest index among ties)
template <class T>
void solver_init(Solver<T>& s, const int32_t* cost, int64_t n, int k) {
    s.cost = cost; s.n = n; s.k = k;
    s.assign.assign(n, -1);
    s.pos.assign(n, 0);
    s.mem.clear();
    s.mem.resize(k);
    s.best.assign((size_t)k * k, Lim<T>::INF);
    s.bestp.assign((size_t)k * k, -1);
}

// step 1 in ONE pass over the cost matrix: a point's nearest centre (lowest index among ties) and its row of the table while the cost
// row is in cache; also the largest and the smallest cost seen (they decide whether 32-bit tables were admissible)
template <class T>
void first_pass(Solver<T>& s, int32_t* mx_out, int32_t* mn_out) {
    int32_t mx = 0, mn = 0;
    for (int64_t p = 0; p < s.n; ++p) {
        const int32_t* c = s.cost + p * s.k;
        const int a = row_argmin(c, s.k, &mx);
        if (c[a] < mn) mn = c[a];
        s.attach((int32_t)p, a);
        s.add(a, (int32_t)p);
    }
    *mx_out = mx;
    *mn_out = mn;
}

// step 2 on a solver whose step 1 is done
template <class T>
int repair(Solver<T>& s, int size_min, int size_max, bool packed, int32_t* labels_out, int64_t* total_cost_out) {
    const int32_t* cost = s.cost;
    const int64_t n = s.n;
    const int k = s.k;
    // 2. repair the balances.  Node k is G, the free sink.  cnt(a) = members of a.
    const int G = k;
    const int nn = k + 1;
    auto cnt = [&](int a) { return (int64_t)s.mem[a].size(); };
    int64_t under = 0, over = 0;
    for (int a = 0; a < k; ++a) {
        if (cnt(a) < size_min) under += size_min - cnt(a);
        if (cnt(a) > size_max) over += cnt(a) - size_max;
    }
    std::vector<int64_t> pi(nn, 0), dist(nn), key(nn);
    std::vector<int32_t> pred(nn);
    std::vector<int8_t> done(nn);
    std::vector<int> path;
    std::vector<int32_t> movers;
    const int64_t max_aug = under + over + 8;
    int64_t n_aug = 0;
    while (under > 0 || over > 0) {
        if (++n_aug > max_aug) {
            scd_set_error("scd_transport_solve: internal error, the repair did not terminate");
            return SCD_EINFEASIBLE;
        }
        // source: an excess node - G while the centres lack more units than others hold too many (G then holds units that belong to
        // a centre below size_min), else the lowest over-full centre
        int src = -1;
        if (under > over) src = G;
        else
            for (int a = 0; a < k && src < 0; ++a)
                if (cnt(a) > size_max) src = a;
        // deficit nodes: centres below size_min; G when over > under
        auto is_deficit = [&](int v) { return v == G ? over > under : cnt(v) < size_min; };
        for (int v = 0; v < nn; ++v) { dist[v] = INF64; key[v] = KEY_NONE; pred[v] = -1; done[v] = 0; }
        dist[src] = 0;
        key[src] = pack_key(0, src);
        int t = -1;
        int64_t knext = src == G ? KEY_NONE : key[src];   // smallest key of the centres 0..k-1 when `fresh` (the last relax pass left it), else unknown
        bool fresh = true;
        for (;;) {
            int u;
            if (packed) {
                int64_t kmin = fresh ? knext : min_key(key.data(), k);
                kmin = key[G] < kmin ? key[G] : kmin;
                u = kmin == KEY_NONE ? -1 : (int)(kmin & ((1 << KEY_BITS) - 1));
            } else {
                u = select_min(dist.data(), done.data(), nn);
            }
            fresh = false;
            if (u < 0) break;
            done[u] = 1;
            key[u] = KEY_NONE;
            if (u != src && is_deficit(u)) { t = u; break; }
            const int64_t du = dist[u] + pi[u];
            if (u == G) {
                for (int v = 0; v < k; ++v)                       // G -> v: v gives up a unit it was free to hold
                    if (!done[v] && cnt(v) > size_min) {
                        const int64_t nd = du - pi[v];
                        if (nd < dist[v]) { dist[v] = nd; key[v] = pack_key(nd, v); pred[v] = G; }
                    }
                continue;
            }
            if (!done[G] && cnt(u) < size_max) {                  // u -> G: u keeps the unit it has just received
                const int64_t nd = du - pi[G];
                if (nd < dist[G]) { dist[G] = nd; key[G] = pack_key(nd, G); pred[G] = u; }
            }
            if (cnt(u) == 0) continue;
            knext = relax_row(dist.data(), key.data(), pred.data(), s.best.data() + (size_t)u * k, pi.data(), done.data(), du, u, k);
            fresh = true;
        }
        if (t < 0) {
            scd_set_error("There was an issue with the min cost flow input.");
            return SCD_EINFEASIBLE;
        }
        // potentials: pi += min(dist, dist[t]) (nodes the search did not settle are at least dist[t] away)
        const int64_t dt = dist[t];
        for (int v = 0; v < nn; ++v) pi[v] += (done[v] && dist[v] < dt) ? dist[v] : dt;
        path.clear();
        for (int v = t; v >= 0; v = pred[v]) {
            path.push_back(v);
            if (path.size() > (size_t)nn + 1) {
                scd_set_error("scd_transport_solve: predecessor cycle");
                return SCD_EINFEASIBLE;
            }
        }
        std::reverse(path.begin(), path.end());                  // src, ..., t
        // the members that move, resolved before any assignment changes
        movers.assign(path.size(), -1);
        for (size_t i = 0; i + 1 < path.size(); ++i)
            if (path[i] != G && path[i + 1] != G) movers[i] = s.bestp[(size_t)path[i] * k + path[i + 1]];
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == G || b == G) continue;
            const int32_t p = movers[i];
            s.detach(p);
            s.attach(p, b);
        }
        // rows: a centre that lost a member is rebuilt from its member list (final here: a member that arrived in the same
        // augmentation is in it); a centre that only gained a member takes it in
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == G || b == G) continue;
            s.rebuild(a);
        }
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == G || b == G) continue;
            s.add(b, movers[i]);
        }
        // the balances follow from the member counts (a centre's flow to G is min(cnt - size_min, size_max - size_min) by construction)
        under = over = 0;
        for (int a = 0; a < k; ++a) {
            if (cnt(a) < size_min) under += size_min - cnt(a);
            if (cnt(a) > size_max) over += cnt(a) - size_max;
        }
    }
    int64_t total = 0;
    for (int64_t p = 0; p < n; ++p) {
        labels_out[p] = s.assign[p];
        total += cost[p * k + s.assign[p]];
    }
    for (int b = 0; b < k; ++b)
        if (cnt(b) < size_min || cnt(b) > size_max) {
            scd_set_error("scd_transport_solve: internal error, cluster %d has %lld members", b, (long long)cnt(b));
            return SCD_EINFEASIBLE;
        }
    if (total_cost_out) *total_cost_out = total;
    return SCD_OK;
}

}  // namespace

static int transport_solve_one(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                               int64_t* total_cost_out) {
    SCD_REQUIRE(cost && labels_out && n > 0 && k > 0 && size_min >= 0 && size_max >= size_min,
                "scd_transport_solve: bad arguments");
    if ((int64_t)k * size_min > n || (int64_t)k * size_max < n) {
        scd_set_error("There was an issue with the min cost flow input.");   // message of sskm_constrained.py:350
        return SCD_EINFEASIBLE;
    }
    SCD_REQUIRE(n < INT32_MAX, "scd_transport_solve: more than 2^31 points");
    // optimistic: 32-bit tables (every cost in [0, 2^30): always, for round(1000 * distance)); the pass itself finds out whether that held
    int32_t mx = 0, mn = 0;
    {
        Solver<int32_t> s32;
        solver_init(s32, cost, n, k);
        first_pass(s32, &mx, &mn);
        if (mn >= 0 && mx < (1 << 30)) {
            // a search's distances are sums of at most k + 1 reduced costs, each below 2 (mx + 1) in magnitude: packed selection keys
            // while that stays below 2^42
            const bool packed = k + 1 < (1 << KEY_BITS) && (int64_t)(k + 2) * 2 * ((int64_t)mx + 1) < ((int64_t)1 << 42);
            return repair(s32, size_min, size_max, packed, labels_out, total_cost_out);
        }
    }
    Solver<int64_t> s64;
    solver_init(s64, cost, n, k);
    first_pass(s64, &mx, &mn);
    return repair(s64, size_min, size_max, false, labels_out, total_cost_out);
}

// no exception crosses the C boundary or ends a worker thread (std::terminate): an allocation failure becomes a status
static int transport_solve_guarded(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                                   int64_t* total_cost_out) {
    try {
        return transport_solve_one(cost, n, k, size_min, size_max, labels_out, total_cost_out);
    } catch (const std::bad_alloc&) {
        scd_set_error("scd_transport_solve: out of host memory (n = %lld, k = %d)", (long long)n, k);
        return SCD_EINVAL;
    } catch (const std::exception& e) {
        scd_set_error("scd_transport_solve: %s", e.what());
        return SCD_EINVAL;
    }
}

extern "C" int scd_transport_solve(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                                   int64_t* total_cost_out) {
    return transport_solve_guarded(cost, n, k, size_min, size_max, labels_out, total_cost_out);
}

// `batch` independent problems of one shape (the ConSSKM E-steps of the restarts of one fit, sskm_constrained.py:165-176: the restarts
// share nothing but X), solved on up to `threads` host threads.  Problem b reads cost + b * n * k and writes labels_out + b * n,
// totals_out[b].  Each problem's result is what scd_transport_solve gives for it (one problem never spans threads), so the batch is
// deterministic whatever the thread count.  Returns the status of the lowest-numbered problem that failed, with that problem's own
// message (a worker's message lives in its thread: it is copied into a per-problem slot and restated on the calling thread).
extern "C" int scd_transport_solve_batch(const int32_t* cost, int64_t n, int k, int batch, int size_min, int size_max,
                                         int32_t* labels_out, int64_t* totals_out, int threads) {
    SCD_REQUIRE(cost && labels_out && batch > 0 && n > 0 && k > 0, "scd_transport_solve_batch: bad arguments");
    try {
        std::vector<int> rc((size_t)batch, SCD_OK);
        std::vector<std::string> msg((size_t)batch);
        auto one = [&](int b) {
            int64_t tot = 0;
            rc[b] = transport_solve_guarded(cost + (size_t)b * n * k, n, k, size_min, size_max, labels_out + (size_t)b * n, &tot);
            if (rc[b] != SCD_OK) {
                try { msg[b] = scd_last_error(); } catch (...) {}
            }
            if (totals_out) totals_out[b] = tot;
        };
        const int nt = std::max(1, std::min(threads, batch));
        if (nt == 1) {
            for (int b = 0; b < batch; ++b) one(b);
        } else {
            std::atomic<int> next(0);
            auto work = [&]() {
                for (int b = next.fetch_add(1); b < batch; b = next.fetch_add(1)) one(b);
            };
            std::vector<std::thread> pool;
            pool.reserve(nt - 1);
            try {
                for (int t = 1; t < nt; ++t) pool.emplace_back(work);
            } catch (...) {}                       // fewer threads than asked for: the ones that started (and this one) do the work
            work();
            for (auto& th : pool) th.join();
        }
        for (int b = 0; b < batch; ++b)
            if (rc[b] != SCD_OK) {
                if (batch > 1 && rc[b] != SCD_EINFEASIBLE) scd_set_error("scd_transport_solve_batch: problem %d: %s", b, msg[b].c_str());
                else scd_set_error("%s", msg[b].empty() ? "There was an issue with the min cost flow input." : msg[b].c_str());
                return rc[b];
            }
        return SCD_OK;
    } catch (const std::exception& e) {
        scd_set_error("scd_transport_solve_batch: %s", e.what());
        return SCD_EINVAL;
    }
}
