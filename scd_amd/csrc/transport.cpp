// scd_transport_solve: the size-constrained assignment of the ConSSKM E-step, i.e. the min-cost-flow problem the
// reference builds at local_utils/sskm_constrained.py:277-328 and hands to OR-Tools SimpleMinCostFlow (:331-356,
// third-party, not in the reference tree).  Input is the dense int32 cost matrix cost[i][j] = round(1000*dist(i,j))
// (the X -> C' arcs); the dummy->centre (cap size_max) and centre->sink (demand size_min) arcs are the bounds.
//
// Algorithm: successive shortest paths specialised to "n unit supplies, k sinks".  Points enter one at a time; the
// residual graph is collapsed to the k cluster nodes, where arc a->b costs the cheapest re-assignment of one member of
// a to b (kept in lazy binary heaps).  Each augmentation is a label-correcting shortest path over k nodes; while every
// point still sits in its nearest cluster all collapsed arcs are non-negative and the direct arc is the shortest path,
// so those points are placed without a search.  The flow after every augmentation is min-cost for the points seen so
// far, hence the final total cost is optimal (labels are one optimum; they need not be OR-Tools' optimum on ties).
#include "common.h"
#include <vector>
#include <queue>
#include <algorithm>

namespace {
struct Move {
    int32_t delta;
    int32_t point;
    bool operator<(const Move& o) const { return delta > o.delta || (delta == o.delta && point > o.point); }  // min-heap
};
}  // namespace

extern "C" int scd_transport_solve(const int32_t* cost, int64_t n, int k, int size_min, int size_max, int32_t* labels_out,
                                   int64_t* total_cost_out) {
    SCD_REQUIRE(cost && labels_out && n > 0 && k > 0 && size_min >= 0 && size_max >= size_min,
                "scd_transport_solve: bad arguments");
    if ((int64_t)k * size_min > n || (int64_t)k * size_max < n) {
        scd_set_error("There was an issue with the min cost flow input.");   // message of sskm_constrained.py:350
        return SCD_EINFEASIBLE;
    }
    std::vector<int32_t> assign(n, -1);
    std::vector<int64_t> cnt(k, 0);
    std::vector<std::priority_queue<Move>> heap;       // k*k, built lazily
    bool heaps_built = false;
    bool all_nearest = true;                            // every placed point is in its nearest cluster
    int64_t deficit = (int64_t)k * size_min;            // sum_j max(0, size_min - cnt_j)
    std::vector<int64_t> dist(k + 1);
    std::vector<int> pred(k + 1);
    std::vector<char> inq(k + 1);
    std::vector<int> queue;

    auto push_point = [&](int64_t p, int a) {
        const int32_t* c = cost + p * k;
        for (int b = 0; b < k; ++b)
            if (b != a) heap[(size_t)a * k + b].push(Move{c[b] - c[a], (int32_t)p});
    };
    auto top = [&](int a, int b, Move& out) -> bool {
        auto& h = heap[(size_t)a * k + b];
        while (!h.empty() && assign[h.top().point] != a) h.pop();
        if (h.empty()) return false;
        out = h.top();
        return true;
    };

    for (int64_t p = 0; p < n; ++p) {
        const int32_t* c = cost + p * k;
        const int64_t remaining_after = n - p - 1;
        auto valid_terminal = [&](int t) {
            if (cnt[t] < size_min) return true;
            return cnt[t] < size_max && deficit <= remaining_after;
        };
        int nearest = 0;
        for (int b = 1; b < k; ++b)
            if (c[b] < c[nearest]) nearest = b;
        if (all_nearest && valid_terminal(nearest)) {
            assign[p] = nearest;
            if (cnt[nearest] < size_min) --deficit;
            ++cnt[nearest];
            if (heaps_built) push_point(p, nearest);
            continue;
        }
        if (!heaps_built) {
            heap.resize((size_t)k * k);
            heaps_built = true;
            for (int64_t q = 0; q < p; ++q) push_point(q, assign[q]);
        }
        // label-correcting shortest paths from p over the cluster graph + the sink hub (node k):
        //   a -> b   cheapest re-assignment of one member of a to b
        //   t -> S   cost 0 when size_min <= cnt_t < size_max (one more unit flows centre_t -> sink)
        //   S -> u   cost 0 when cnt_u > size_min (cancel one unit of centre_u -> sink; u must then shed a member)
        const int S = k;
        queue.clear();
        for (int b = 0; b < k; ++b) {
            dist[b] = c[b];
            pred[b] = -1;
            inq[b] = 1;
            queue.push_back(b);
        }
        dist[S] = INT64_MAX / 4;
        pred[S] = -1;
        inq[S] = 0;
        size_t head = 0;
        while (head < queue.size()) {
            const int a = queue[head++];
            inq[a] = 0;
            if (a == S) {
                for (int u = 0; u < k; ++u)
                    if (cnt[u] > size_min && dist[S] < dist[u]) {
                        dist[u] = dist[S];
                        pred[u] = S;
                        if (!inq[u]) { inq[u] = 1; queue.push_back(u); }
                    }
                continue;
            }
            if (cnt[a] >= size_min && cnt[a] < size_max && dist[a] < dist[S]) {
                dist[S] = dist[a];
                pred[S] = a;
                if (!inq[S]) { inq[S] = 1; queue.push_back(S); }
            }
            if (cnt[a] == 0) continue;
            for (int b = 0; b < k; ++b) {
                if (b == a) continue;
                Move mv;
                if (!top(a, b, mv)) continue;
                const int64_t nd = dist[a] + mv.delta;
                if (nd < dist[b]) {
                    dist[b] = nd;
                    pred[b] = a;
                    if (!inq[b]) { inq[b] = 1; queue.push_back(b); }
                }
            }
            if (queue.size() > (size_t)(k + 1) * (k + 1) * 64) {
                scd_set_error("scd_transport_solve: shortest-path search did not converge");
                return SCD_EINFEASIBLE;
            }
        }
        // terminals: a cluster still below size_min absorbs the unit; the sink absorbs it while it has demand left
        int t = -1;
        for (int b = 0; b < k; ++b)
            if (cnt[b] < size_min && (t < 0 || dist[b] < dist[t])) t = b;
        if (deficit <= remaining_after && pred[S] >= 0 && (t < 0 || dist[S] < dist[t])) t = S;
        if (t < 0) {
            scd_set_error("There was an issue with the min cost flow input.");
            return SCD_EINFEASIBLE;
        }
        std::vector<int> path;
        for (int b = t; b >= 0; b = pred[b]) {
            path.push_back(b);
            if (path.size() > (size_t)4 * (k + 2)) {
                scd_set_error("scd_transport_solve: predecessor cycle");
                return SCD_EINFEASIBLE;
            }
        }
        std::reverse(path.begin(), path.end());              // first cluster, ..., terminal
        // resolve movers before any assignment changes (heap tops refer to the current state)
        std::vector<int32_t> movers(path.size(), -1);
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            if (path[i] == S || path[i + 1] == S) continue;
            Move mv;
            top(path[i], path[i + 1], mv);
            movers[i] = mv.point;
        }
        for (size_t i = 0; i + 1 < path.size(); ++i) {
            const int a = path[i], b = path[i + 1];
            if (a == S) {                                     // S -> u: u gives one unit back (it sheds a member next)
                --cnt[b];
            } else if (b == S) {                              // a -> S: a keeps the unit it just received
                ++cnt[a];
            } else {
                assign[movers[i]] = b;
                push_point(movers[i], b);
            }
        }
        assign[p] = path[0];
        push_point(p, path[0]);
        if (t != S) {
            if (cnt[t] < size_min) --deficit;
            ++cnt[t];
        }
        if (path.size() > 1 || path[0] != nearest) all_nearest = false;
    }
    int64_t total = 0;
    for (int64_t p = 0; p < n; ++p) {
        labels_out[p] = assign[p];
        total += cost[p * k + assign[p]];
    }
    for (int b = 0; b < k; ++b)
        if (cnt[b] < size_min || cnt[b] > size_max) {
            scd_set_error("scd_transport_solve: internal error, cluster %d has %lld members", b, (long long)cnt[b]);
            return SCD_EINFEASIBLE;
        }
    if (total_cost_out) *total_cost_out = total;
    return SCD_OK;
}
