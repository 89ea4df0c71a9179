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
#include "common.h"
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>

namespace {

struct Solver {
    const int32_t* cost;
    int64_t n;
    int k, lo, hi;
    std::vector<int32_t> assign;            // point -> centre
    std::vector<int32_t> pos;               // point -> position in its centre's member list
    std::vector<std::vector<int32_t>> mem;  // centre -> members
    std::vector<int64_t> best;              // [k][k]: min over members p of a of c[p][b] - c[p][a]   (INF: a is empty)
    std::vector<int32_t> bestp;             // the member attaining it (lowest index among ties)
    static constexpr int64_t INF = INT64_MAX / 4;

    void row_add(int a, int32_t p) {
        const int32_t* c = cost + (int64_t)p * k;
        int64_t* r = best.data() + (size_t)a * k;
        int32_t* rp = bestp.data() + (size_t)a * k;
        const int64_t ca = c[a];
        for (int b = 0; b < k; ++b) {
            const int64_t dlt = (int64_t)c[b] - ca;
            if (dlt < r[b] || (dlt == r[b] && p < rp[b])) { r[b] = dlt; rp[b] = p; }
        }
    }
    // member p has left a (mem[a] no longer holds it): only the columns p was the cheapest member for are searched again
    void row_remove(int a, int32_t p) {
        int64_t* r = best.data() + (size_t)a * k;
        int32_t* rp = bestp.data() + (size_t)a * k;
        for (int b = 0; b < k; ++b) {
            if (rp[b] != p) continue;
            int64_t bd = INF;
            int32_t bp = -1;
            for (int32_t q : mem[a]) {
                const int32_t* c = cost + (int64_t)q * k;
                const int64_t dlt = (int64_t)c[b] - c[a];
                if (dlt < bd || (dlt == bd && q < bp)) { bd = dlt; bp = q; }
            }
            r[b] = bd;
            rp[b] = bp;
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
    Solver s;
    s.cost = cost; s.n = n; s.k = k; s.lo = size_min; s.hi = size_max;
    s.assign.assign(n, -1);
    s.pos.assign(n, 0);
    s.mem.resize(k);
    s.best.assign((size_t)k * k, Solver::INF);
    s.bestp.assign((size_t)k * k, -1);
    // 1. nearest centre (lowest index among ties)
    for (int64_t p = 0; p < n; ++p) {
        const int32_t* c = cost + p * k;
        int a = 0;
        for (int b = 1; b < k; ++b)
            if (c[b] < c[a]) a = b;
        s.attach((int32_t)p, a);
        s.row_add(a, (int32_t)p);
    }
    // 2. repair the balances.  Node k is G, the free sink.  cnt(a) = members of a.
    const int G = k;
    const int nn = k + 1;
    auto cnt = [&](int a) { return (int64_t)s.mem[a].size(); };
    int64_t under = 0, over = 0;
    for (int a = 0; a < k; ++a) {
        if (cnt(a) < size_min) under += size_min - cnt(a);
        if (cnt(a) > size_max) over += cnt(a) - size_max;
    }
    std::vector<int64_t> pi(nn, 0), dist(nn);
    std::vector<int> pred(nn);
    std::vector<char> done(nn);
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
        for (int v = 0; v < nn; ++v) { dist[v] = Solver::INF; pred[v] = -1; done[v] = 0; }
        dist[src] = 0;
        int t = -1;
        for (;;) {
            int u = -1;
            for (int v = 0; v < nn; ++v)
                if (!done[v] && dist[v] < Solver::INF && (u < 0 || dist[v] < dist[u])) u = v;
            if (u < 0) break;
            done[u] = 1;
            if (u != src && is_deficit(u)) { t = u; break; }
            const int64_t du = dist[u] + pi[u];
            if (u == G) {
                for (int v = 0; v < k; ++v)                       // G -> v: v gives up a unit it was free to hold
                    if (!done[v] && cnt(v) > size_min) {
                        const int64_t nd = du - pi[v];
                        if (nd < dist[v]) { dist[v] = nd; pred[v] = G; }
                    }
                continue;
            }
            if (!done[G] && cnt(u) < size_max) {                  // u -> G: u keeps the unit it has just received
                const int64_t nd = du - pi[G];
                if (nd < dist[G]) { dist[G] = nd; pred[G] = u; }
            }
            if (cnt(u) == 0) continue;
            const int64_t* r = s.best.data() + (size_t)u * k;
            for (int v = 0; v < k; ++v) {
                if (v == u || done[v]) continue;
                const int64_t nd = du + r[v] - pi[v];             // reduced cost r[v] + pi[u] - pi[v] >= 0
                if (nd < dist[v]) { dist[v] = nd; pred[v] = u; }
            }
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
        // rows: a centre that lost a member searches the columns that member was cheapest for (its member list is final here: a
        // member that arrived in the same augmentation is in it); every centre that gained a member takes it in
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == G || b == G) continue;
            s.row_remove(a, movers[i]);
        }
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == G || b == G) continue;
            s.row_add(b, movers[i]);
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

extern "C" int scd_transport_solve(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                                   int64_t* total_cost_out) {
    return transport_solve_one(cost, n, k, size_min, size_max, labels_out, total_cost_out);
}

// `batch` independent problems of one shape (the ConSSKM E-steps of the restarts of one fit, sskm_constrained.py:165-176: the restarts
// share nothing but X), solved on up to `threads` host threads.  Problem b reads cost + b * n * k and writes labels_out + b * n,
// totals_out[b].  Each problem's result is what scd_transport_solve gives for it (one problem never spans threads), so the batch is
// deterministic whatever the thread count.  Returns the status of the lowest-numbered problem that failed.
extern "C" int scd_transport_solve_batch(const int32_t* cost, int64_t n, int k, int batch, int size_min, int size_max,
                                         int32_t* labels_out, int64_t* totals_out, int threads) {
    SCD_REQUIRE(cost && labels_out && batch > 0 && n > 0 && k > 0, "scd_transport_solve_batch: bad arguments");
    std::vector<int> rc((size_t)batch, SCD_OK);
    auto one = [&](int b) {
        int64_t tot = 0;
        rc[b] = transport_solve_one(cost + (size_t)b * n * k, n, k, size_min, size_max, labels_out + (size_t)b * n, &tot);
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
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    for (int b = 0; b < batch; ++b)
        if (rc[b] != SCD_OK) {
            // the worker's message lives in ITS thread: restate it here
            if (rc[b] == SCD_EINFEASIBLE) scd_set_error("There was an issue with the min cost flow input.");
            else scd_set_error("scd_transport_solve_batch: problem %d failed (status %d)", b, rc[b]);
            return rc[b];
        }
    return SCD_OK;
}
