"""CPU oracle for the size-constrained E-step.

TEST INFRASTRUCTURE ONLY - never imported from scd_amd/.

Restates /root/reference/local_utils/sskm_constrained.py:
  _labels_constrained            :226-274
  minimum_cost_flow_problem_graph :277-328   (pure numpy in the reference -> bit-exact goldens)
  solve_min_cost_flow_graph       :331-356   (calls OR-Tools SimpleMinCostFlow)

OR-Tools (ortools==9.3.10497, requirements.txt:103) is a third-party dependency
that is NOT under /root/reference and is not installed here, so the solver
itself cannot be run.  Parity for the solver is therefore "unpinned" on labels:
the minimum-cost assignment is not unique when integer costs tie.  What IS
pinned: the graph arrays (against the reference's own numpy code, goldens
tests/golden/mcf_graph.npz), the optimal total cost (unique; checked here with
scipy's HiGHS LP, which is integral on a transportation polytope), feasibility
of the size bounds, and label equality on instances with a unique optimum.
"""
import numpy as np
from . import kmeans_oracle as ko

F32 = np.float32


def int_costs(d2_f32):
    """round(1000*sqrt(d2)) as int32: torch.sqrt(dist) float32 (:116) then
    np.around(costs*1000,0).astype('int32') (:324)."""
    d = np.sqrt(np.asarray(d2_f32, dtype=F32)).astype(F32)
    return np.around(d * F32(1000.0), 0).astype(np.int32)


def mcf_graph(n_x, n_c, d_sqrt, size_min, size_max):
    """Arrays of minimum_cost_flow_problem_graph (:277-328).

    Node ids: points [0,N), dummies [N,N+K), centres [N+K,N+2K), sink N+2K.
    Arc order: point-major (i*K+j), then dummy->centre, then centre->sink.
    """
    x_ix = np.arange(n_x)
    dummy = n_x + np.arange(n_c)
    cent = n_x + n_c + np.arange(n_c)
    sink = n_x + 2 * n_c
    e0 = np.stack([np.repeat(x_ix, n_c), np.tile(dummy, n_x)], axis=1)
    e1 = np.stack([dummy, cent], axis=1)
    e2 = np.stack([cent, np.full(n_c, sink)], axis=1)
    edges = np.concatenate([e0, e1, e2]).astype(np.int32)
    costs = np.concatenate([np.asarray(d_sqrt).reshape(-1), np.zeros(2 * n_c)])
    costs = np.around(costs * 1000, 0).astype(np.int32)
    caps = np.concatenate([np.ones(n_x * n_c), size_max * np.ones(n_c), n_x * np.ones(n_c)]).astype(np.int32)
    supplies = np.concatenate([np.ones(n_x), np.zeros(n_c), -size_min * np.ones(n_c),
                               [-(n_x - n_c * size_min)]]).astype(np.int32)
    return edges, costs, caps, supplies


def solve_lp(cost_i32, size_min, size_max):
    """Optimal (labels, total_cost) of the transportation problem via HiGHS.

    Raises Exception('There was an issue with the min cost flow input.') when
    infeasible, as solve_min_cost_flow_graph does (:349-350).
    """
    from scipy.optimize import linprog
    from scipy.sparse import coo_matrix
    c = np.asarray(cost_i32, dtype=np.float64)
    n, k = c.shape
    if k * size_min > n or k * size_max < n:
        raise Exception("There was an issue with the min cost flow input.")
    nv = n * k
    rows = np.repeat(np.arange(n), k)
    cols = np.arange(nv)
    a_eq = coo_matrix((np.ones(nv), (rows, cols)), shape=(n, nv)).tocsr()
    crow = np.tile(np.arange(k), n)
    a_col = coo_matrix((np.ones(nv), (crow, cols)), shape=(k, nv)).tocsr()
    from scipy.sparse import vstack
    a_ub = vstack([a_col, -a_col]).tocsr()
    b_ub = np.concatenate([np.full(k, size_max, dtype=np.float64), np.full(k, -size_min, dtype=np.float64)])
    res = linprog(c.reshape(-1), A_ub=a_ub, b_ub=b_ub, A_eq=a_eq, b_eq=np.ones(n), bounds=(0, 1), method="highs-ds")
    if res.status != 0:
        raise Exception("There was an issue with the min cost flow input.")
    flow = np.rint(res.x).astype(np.int64).reshape(n, k)
    labels = flow.argmax(axis=1)
    total = int(np.sum(cost_i32[np.arange(n), labels].astype(np.int64)))
    return labels.astype(np.int32), total


def labels_constrained(d2_f32, size_min, size_max):
    """_labels_constrained (:226-274): labels, float32 inertia = sum(D[i,label]^2)
    where D = sqrt(dist) float32 and the square is float32 (:271)."""
    d_sqrt = np.sqrt(np.asarray(d2_f32, dtype=F32)).astype(F32)
    labels, total = solve_lp(int_costs(d2_f32), size_min, size_max)
    dist = (d_sqrt[np.arange(len(labels)), labels] ** 2).astype(F32)
    inertia = F32(np.sum(dist.astype(np.float64)))
    return labels, inertia, total


def check_assignment(cost_i32, labels, size_min, size_max):
    """Feasibility + total cost of a candidate assignment."""
    n, k = cost_i32.shape
    cnt = np.bincount(labels, minlength=k)
    ok = bool(np.all(cnt >= size_min) and np.all(cnt <= size_max) and len(labels) == n)
    total = int(np.sum(cost_i32[np.arange(n), labels].astype(np.int64)))
    return ok, total


class K_Means(ko.K_Means):
    """ConSSKM oracle: sskm_constrained.K_Means (:15-187)."""

    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, size_min=100, size_max=1000,
                 init="k-means++", n_init=10, random_state=None, n_jobs=None, pairwise_batch_size=None):
        super().__init__(k, tolerance, max_iterations, init, n_init, random_state, n_jobs, pairwise_batch_size)
        self.size_min = size_min
        self.size_max = size_max

    def assign(self, x, centers):
        d2 = ko.pairwise_distance(x, centers)
        labels, inertia, _ = labels_constrained(d2, self.size_min, self.size_max)
        return labels.astype(np.int64), inertia


def check_optimal(cost_i32, labels, size_min, size_max):
    """Independent optimality certificate for a FEASIBLE assignment, without an LP: the assignment is optimal iff the residual
    graph has no negative cycle.  Cluster-level form: arc a -> b with weight min_{i in a} (c[i,b] - c[i,a]) (move the cheapest
    point of a to b), plus a node Z with 0-weight arcs Z -> a for clusters above size_min (may lose a point) and b -> Z for
    clusters below size_max (may gain one).  Bellman-Ford over K + 1 nodes.  Returns True when no improving move exists."""
    c = np.asarray(cost_i32, dtype=np.int64)
    n, k = c.shape
    labels = np.asarray(labels)
    cnt = np.bincount(labels, minlength=k)
    big = np.int64(1) << 60
    w = np.full((k + 1, k + 1), big, dtype=np.int64)
    for a in range(k):
        rows = np.nonzero(labels == a)[0]
        if rows.size:
            w[a, :k] = (c[rows] - c[rows, a][:, None]).min(axis=0)
        w[a, a] = big
        if cnt[a] > size_min:
            w[k, a] = 0
        if cnt[a] < size_max:
            w[a, k] = 0
    dist = np.zeros(k + 1, dtype=np.int64)               # virtual source to every node
    for _ in range(k + 2):
        cand = (dist[:, None] + np.where(w >= big, big, w)).min(axis=0)
        new = np.minimum(dist, cand)
        if np.array_equal(new, dist):
            return True
        dist = new
    return False
