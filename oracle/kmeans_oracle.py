"""CPU oracle for the K-Means part of the SCD hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under scd_amd/ may import this module; it is
the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

It restates, in numpy, the algorithm of
  * /root/reference/local_utils/sskm_constrained.py            (ConSSKM, K_Means :15-187)
  * /root/reference/gcd/methods/clustering/faster_mix_k_means_pytorch.py (SSKM, :47-275)
with ONE deliberate sharpening that the HIP path shares: every *decision*
(argmin of the E-step, the k-means++ draw, best-of-restarts) is taken on values
computed in float64 from the given float inputs and then rounded once to
float32, with ties broken towards the lowest index.  The reference takes the
same decisions on float32 values whose summation order is whatever the torch
build uses (sskm_constrained.py:212-213), so the two agree except on rows whose
top-2 margin is inside float32 round-off; the golden fixtures
(tests/golden/kmeans_*.npz, produced by oracle/gen_golden.py from the reference
itself) pin that agreement on clustered data.

Parity status: pinned against reference outputs generated in the build
container (labels, centres, inertia, k-means++ picks).  The OR-Tools solver the
constrained E-step calls is a third-party dependency absent from
/root/reference (ortools==9.3.10497, requirements.txt:103): for that sub-step
only the optimal total cost and feasibility are pinned (see transport_oracle).
"""
import numpy as np

F32 = np.float32
F64 = np.float64


def check_random_state(seed):
    """sklearn.utils.check_random_state semantics used at sskm_constrained.py:29,142,166."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("bad seed %r" % (seed,))


def pairwise_distance64(a, b, block=2048):
    """Squared Euclidean distances, difference form, float64.

    Follows pairwise_distance (sskm_constrained.py:189-224): ((A[:,None]-B[None])**2).sum(-1).
    """
    a = np.asarray(a, dtype=F64)
    b = np.asarray(b, dtype=F64)
    out = np.empty((a.shape[0], b.shape[0]), dtype=F64)
    if a.shape[0] < b.shape[0] and b.shape[0] > block:          # few rows against many (k-means++ candidates): block over the many
        return pairwise_distance64(b, a, block).T
    block = max(1, min(block, int(2 ** 21 // max(1, b.shape[0] * a.shape[1]))))       # the [block, k, d] temporary stays in cache
    for s in range(0, a.shape[0], block):
        blk = a[s:s + block]
        # (x-c)^2 summed; keep the difference form (no ||x||^2 - 2xc + ||c||^2 cancellation)
        d = blk[:, None, :] - b[None, :, :]
        out[s:s + block] = np.einsum("nkd,nkd->nk", d, d)
    return out


def dist_f32(a, b):
    """float32(difference-form float64 squared distance) of every pair - pairwise_distance64(a, b).astype(float32) - at BLAS speed for
    the large test shapes (CUB-sized k-means++ rounds): the float64 GEMM form g = |a|^2 + |b|^2 - 2 a.b is within ~D 2^-52 (|a|^2 + |b|^2)
    of the difference form; where float32(g - e) == float32(g + e) for e = 1e-11 (|a|^2 + |b|^2) the rounding of the true value is
    decided, every other pair (and every NaN) is recomputed in the difference form."""
    a64, b64 = np.asarray(a, dtype=F64), np.asarray(b, dtype=F64)
    if a64.shape[0] * b64.shape[0] * a64.shape[1] <= 2 ** 22:
        return pairwise_distance64(a64, b64).astype(F32)
    an, bn = np.einsum("nd,nd->n", a64, a64), np.einsum("kd,kd->k", b64, b64)
    g = an[:, None] + bn[None, :] - 2.0 * (a64 @ b64.T)
    e = 1e-11 * (an[:, None] + bn[None, :])
    lo, hi = (g - e).astype(F32), (g + e).astype(F32)
    out = g.astype(F32)
    ia, ib = np.nonzero(lo != hi)
    for s in range(0, ia.size, 65536):
        df = a64[ia[s:s + 65536]] - b64[ib[s:s + 65536]]
        out[ia[s:s + 65536], ib[s:s + 65536]] = np.einsum("nd,nd->n", df, df).astype(F32)
    return out


def pairwise_distance(a, b, batch_size=None):
    """float32 result of the reference call (sskm_constrained.py:189)."""
    return pairwise_distance64(a, b).astype(F32)


def estep(x, centers):
    """argmin_k ||x-c_k||^2 (ties -> lowest k) and float32 min distance.

    Reference: torch.min(dist, dim=1) at faster_mix_k_means_pytorch.py:140,192.
    NaN centres (empty clusters, :203) never win: NaN distances are treated as +inf,
    which is what torch.min does not guarantee - documented divergence, the
    reference propagates NaN.
    """
    x64 = np.asarray(x, dtype=F64)
    c64 = np.asarray(centers, dtype=F64)
    n, k = x64.shape[0], c64.shape[0]
    if n * k * x64.shape[1] <= 2 ** 24:              # small cases: the difference form for every pair
        d = pairwise_distance64(x64, c64)
        d = np.where(np.isnan(d), np.inf, d)
        lab = np.argmin(d, axis=1)
        return lab.astype(np.int64), d[np.arange(n), lab].astype(F32), d
    # large cases (C1 / C3-shaped tests): the same decisions at BLAS speed.  The float64 GEMM form |x|^2 + |c|^2 - 2 x.c is within
    # ~D 2^-52 (|x|^2 + |c|^2) of the difference form; a row whose two smallest values are further apart than 1e-9 (|x|^2 + max |c|^2)
    # - four orders above that bound - has the difference form's argmin, every other row is re-evaluated in the difference form.  The
    # returned distances of the chosen centres are difference-form values in either case.
    xn, cn = np.einsum("nd,nd->n", x64, x64), np.einsum("kd,kd->k", c64, c64)
    d = xn[:, None] + cn[None, :] - 2.0 * (x64 @ c64.T)
    d = np.where(np.isnan(d), np.inf, d)
    lab = np.argmin(d, axis=1)
    if k > 1:
        two = np.partition(d, 1, axis=1)[:, :2]
        near = ~((two[:, 1] - two[:, 0]) > 1e-9 * (xn + np.nanmax(cn)))
    else:
        near = np.zeros(n, dtype=bool)
    rows = np.nonzero(near)[0]
    if rows.size:
        dr = pairwise_distance64(x64[rows], c64)
        dr = np.where(np.isnan(dr), np.inf, dr)
        d[rows] = dr
        lab[rows] = np.argmin(dr, axis=1)
    diff = x64 - c64[lab]
    mind = np.einsum("nd,nd->n", diff, diff)
    mind = np.where(np.isnan(mind), np.inf, mind)
    d[np.arange(n), lab] = mind
    return lab.astype(np.int64), mind.astype(F32), d


def mstep(x, labels, k):
    """centres[idx] = mean of members (sskm_constrained.py:125-128); empty -> NaN."""
    x64 = np.asarray(x, dtype=F64)
    d = x64.shape[1]
    sums = np.zeros((k, d), dtype=F64)
    np.add.at(sums, labels, x64)
    cnt = np.bincount(labels, minlength=k).astype(F64)
    with np.errstate(invalid="ignore", divide="ignore"):
        c = sums / cnt[:, None]
    return c.astype(F32), cnt.astype(np.int64)


def kpp_draw(d2_f32, r):
    """One k-means++ draw (sskm_constrained.py:38-42).

    prob = d2/d2.sum() in float32; torch's CPU cumsum accumulates in double and
    rounds each prefix to float32; `cum_prob >= r` compares in float32.
    Returns the first index, or -1 when no prefix reaches r (the reference then
    raises IndexError at :42).
    """
    d2 = np.asarray(d2_f32, dtype=F32)
    tot = F32(np.sum(d2.astype(F64)))
    prob = (d2 / tot).astype(F32)
    cum = np.cumsum(prob.astype(F64)).astype(F32)
    hit = np.nonzero(cum >= F32(r))[0]
    return int(hit[0]) if hit.size else -1


def kpp(x, pre_centers, k, random_state, trace=None):
    """K_Means.kpp (sskm_constrained.py:28-44 / faster_mix...:82-110)."""
    rs = check_random_state(random_state)
    x = np.asarray(x, dtype=F32)
    if pre_centers is not None:
        c = np.asarray(pre_centers, dtype=F32).reshape(-1, x.shape[1])
    else:
        c = x[rs.randint(0, len(x))].reshape(1, -1)
    d2 = pairwise_distance64(x, c).min(axis=1)
    while c.shape[0] < k:
        r = rs.rand()
        ind = kpp_draw(d2.astype(F32), r)
        if ind < 0:
            raise IndexError("k-means++ draw fell off the end of cum_prob")
        if trace is not None:
            trace.append(ind)
        c = np.concatenate([c, x[ind:ind + 1]], axis=0)
        d2 = np.minimum(d2, pairwise_distance64(x, x[ind:ind + 1])[:, 0])
    return c


def _center_shift_sq(c_new, c_old):
    """(sum_k ||c_k - c_k_old||_2)^2  (sskm_constrained.py:135-136)."""
    d = np.asarray(c_new, dtype=F64) - np.asarray(c_old, dtype=F64)
    return float(np.sum(np.sqrt(np.sum(d * d, axis=1))) ** 2)


class K_Means:
    """SSKM oracle: faster_mix_k_means_pytorch.K_Means (:47-275).

    `assign` is the E-step hook: the constrained variant swaps it for the
    min-cost-flow assignment (sskm_constrained.py:116).
    """

    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, init="k-means++", n_init=10,
                 random_state=None, n_jobs=None, pairwise_batch_size=None, mode=None):
        self.k = k
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.init = init
        self.n_init = n_init
        self.random_state = random_state
        self.n_jobs = n_jobs
        self.pairwise_batch_size = pairwise_batch_size
        self.mode = mode
        self.trace = []

    # E-step on the unlabelled rows -> (labels int64, float32 inertia contribution)
    def assign(self, x, centers):
        # A NaN centre (a cluster the previous M-step left empty: the mean of no rows, faster_mix...:147-150, :199-203): the
        # reference's `torch.min(dist, dim=1)` (:140, :192) propagates NaN - every row's minimum is NaN, found at the FIRST NaN
        # column - so every row goes to the lowest-numbered empty cluster and the iteration's inertia is NaN: it is never the
        # best one, its NaN centre shift never ends the loop, and (unless exactly one cluster can be empty at all) every later
        # iteration is the same.  The restart keeps the best of the iterations up to the one that emptied a cluster and reports
        # max_iterations.  (The E-step OPERATION below keeps NaN centres out of the argmin; this is the LOOP's semantics.)
        nan_rows = np.isnan(np.asarray(centers, dtype=F64)).any(axis=1)
        if nan_rows.any():
            return np.full(len(x), int(np.argmax(nan_rows)), dtype=np.int64), F32(np.nan)
        # inertia contribution = float32(sum of the float64 row minima); the reference sums the float32
        # minima in float32 (faster_mix...:193) - both are within 1e-7 of each other
        lab, _, d = estep(x, centers)
        return lab, F32(np.sum(d[np.arange(d.shape[0]), lab]))

    def fit_once(self, x, random_state):
        x = np.asarray(x, dtype=F32)
        if self.init == "k-means++":
            centers = kpp(x, None, self.k, random_state, self.trace)
        elif self.init == "random":
            rs = check_random_state(self.random_state)
            centers = x[rs.choice(len(x), self.k, replace=False)].copy()
        else:
            centers = x[: self.k].copy()
        best = (None, None, None)
        it = 0
        for it in range(self.max_iterations):
            old = centers.copy()
            labels, inertia = self.assign(x, centers)
            centers, _ = mstep(x, labels, self.k)
            if best[1] is None or inertia < best[1]:
                best = (labels.copy(), inertia, centers.copy())
            if _center_shift_sq(centers, old) < self.tolerance:
                break
        return best[0], best[1], best[2], it + 1

    def fit_mix_once(self, u, l, l_targets, random_state):
        u = np.asarray(u, dtype=F32)
        l = np.asarray(l, dtype=F32)
        l_targets = np.asarray(l_targets)
        classes = np.unique(l_targets)                       # torch.unique sorts (:165)
        l_centers = np.stack([l[l_targets == c].astype(F64).mean(0) for c in classes]).astype(F32)
        cat = np.concatenate([l, u])
        l_num = len(l_targets)
        labels = -np.ones(len(cat), dtype=np.int64)
        lut = {c: i for i, c in enumerate(classes.tolist())}
        labels[:l_num] = [lut[t] for t in l_targets.tolist()]
        centers = kpp(u, l_centers, self.k, random_state, self.trace)
        best = (None, None, None)
        for it in range(self.max_iterations):
            old = centers.copy()
            u_lab, u_inertia = self.assign(u, centers)
            ld = l.astype(F64) - centers[labels[:l_num]].astype(F64)
            l_inertia = F32(np.sum(ld * ld))
            inertia = F32(F32(u_inertia) + l_inertia)
            labels[l_num:] = u_lab
            centers, _ = mstep(cat, labels, self.k)
            if best[1] is None or inertia < best[1]:
                best = (labels.copy(), inertia, centers.copy())
            if _center_shift_sq(centers, old) < self.tolerance:
                break
        # reference returns `i + 1` with i the stale labelled-sample loop index
        # (faster_mix...:181,216 / sskm_constrained.py:104,139) => n_iter == l_num
        return best[0], best[1], best[2], l_num

    def _run(self, once, *args):
        rs = check_random_state(self.random_state)
        best_inertia = None
        for _ in range(self.n_init):
            labels, inertia, centers, n_iters = once(*args, rs)
            if best_inertia is None or inertia < best_inertia:
                self.labels_ = labels.copy()
                self.cluster_centers_ = centers.copy()
                best_inertia = inertia
                self.inertia_ = inertia
                self.n_iter_ = n_iters

    def fit(self, x):
        self._run(self.fit_once, x)

    def fit_mix(self, u, l, l_targets):
        self._run(self.fit_mix_once, u, l, l_targets)


# ----------------------------------------------------------------------------- sklearn.cluster.KMeans (--cluster KM)
def sklearn_tolerance(x, tol):
    """sklearn/cluster/_kmeans.py `_tolerance`: mean(var(X, axis=0)) * tol."""
    return float(np.mean(np.var(np.asarray(x, dtype=F64), axis=0)) * tol)


def sklearn_lloyd(x, init, max_iter=300, tol=1e-4):
    """`_kmeans_single_lloyd` of scikit-learn 1.7.2 (third-party; call site /root/reference/main_unsup.py:362) from an explicit
    init, with the decision semantics of this oracle (float64 distances, ties to the lowest index): E-step, centre update with
    `_relocate_empty_clusters_dense`, strict / tol convergence, final E-step when the stop was not strict.
    Returns (labels int32, inertia, centres float32, n_iter).  Pinned by tests/golden/kmeans_sklearn.npz (sklearn's own output)."""
    x = np.asarray(x, dtype=F32)
    n = x.shape[0]
    centers = np.asarray(init, dtype=F32).copy()
    k = centers.shape[0]
    tol_abs = sklearn_tolerance(x, tol)
    labels_old = np.full(n, -1, dtype=np.int64)
    strict = False
    it = 0
    for it in range(max_iter):
        labels, _, dmat = estep(x, centers)
        x64 = x.astype(F64)
        sums = np.zeros((k, x.shape[1]), dtype=F64)
        np.add.at(sums, labels, x64)
        cnt = np.bincount(labels, minlength=k).astype(np.int64)
        empty = np.nonzero(cnt == 0)[0]
        if empty.size:
            dist = dmat[np.arange(n), labels].astype(F32)
            far = np.argpartition(dist, -empty.size)[:-empty.size - 1:-1]
            for j, e in enumerate(empty):
                old = labels[far[j]]
                sums[old] -= x64[far[j]]
                sums[e] = x64[far[j]]
                cnt[e] = 1
                cnt[old] -= 1
        new = (sums / cnt[:, None]).astype(F32)
        shift = float(((new.astype(F64) - centers.astype(F64)) ** 2).sum())
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift <= tol_abs:
            break
        labels_old = labels
    if not strict:
        labels, _, dmat = estep(x, centers)
    inertia = float(pairwise_distance64(x, centers)[np.arange(n), labels].astype(F32).astype(F64).sum())
    return labels.astype(np.int32), inertia, centers, it + 1


def sklearn_kpp(x, k, random_state, compat="1.7.2"):
    """`_kmeans_plusplus` of scikit-learn with this oracle's arithmetic: closest distances are float32(float64 exact),
    the potential is float32(float64 sum) (sklearn: a float32 BLAS dot), candidates = searchsorted(cumsum_f64(d2), u * pot),
    the candidate with the smallest new potential (float64 sum) wins.  RandomState consumption as in sklearn: the first centre,
    then 2 + int(log k) uniforms per added centre.  The first centre is `choice(n, p=uniform)` in scikit-learn 1.7.2 (compat
    "1.7.2": pinned by tests/golden/kmeans_sklearn.npz `*_kpp_picks`, the output of sklearn.cluster.kmeans_plusplus) and
    `randint(n)` in the 1.0.2 that /root/reference/requirements.txt pins (compat "1.0.2": restated from the public source,
    that version is not installed here => unpinned).  Returns the chosen row indices."""
    rs = check_random_state(random_state)
    x = np.asarray(x, dtype=F32)
    n = x.shape[0]
    trials = 2 + int(np.log(k))
    if compat == "1.0.2":
        picks = [int(rs.randint(n))]
    else:
        p = np.ones(n, dtype=F32)
        picks = [int(rs.choice(n, p=p / p.sum()))]
    d2 = dist_f32(x, x[picks[0]][None])[:, 0]
    for _ in range(1, k):
        pot = F32(d2.astype(F64).sum())
        rv = rs.uniform(size=trials) * F64(pot)
        cand = np.searchsorted(np.cumsum(d2.astype(F64)), rv)
        np.clip(cand, None, n - 1, out=cand)
        dc = np.minimum(d2[None, :], dist_f32(x[cand], x))
        best = int(np.argmin(dc.astype(F64).sum(axis=1)))
        picks.append(int(cand[best]))
        d2 = dc[best]
    return np.array(picks)


def _same_clustering(a, b, k):
    """sklearn/cluster/_k_means_common.pyx `_is_same_clustering`: equal up to a permutation of the labels."""
    mapping = np.full(k, -1, dtype=np.int64)
    for la, lb in zip(a.tolist(), b.tolist()):
        if mapping[la] == -1:
            mapping[la] = lb
        elif mapping[la] != lb:
            return False
    return True


def sklearn_kmeans(x, k, random_state=0, n_init="auto", compat="1.7.2", max_iter=300, tol=1e-4):
    """`KMeans(n_clusters=k, random_state=0[, n_init]).fit(x)` as /root/reference/main_unsup.py:362 / main_ptsup.py:381 call
    it: n_init k-means++ starts on ONE RandomState, each followed by Lloyd; the kept start is the first with a strictly smaller
    inertia (1.7.2: and a different clustering; 1.0.2: smaller than best * (1 - 1e-6)).  n_init 'auto' -> 1 (1.7.2) / 10 (the
    1.0.2 default).  Pinned for compat "1.7.2" by the golden's `*_default_labels` and `*_n10_labels`.
    Returns (labels int32, inertia, centres float32, n_iter)."""
    rs = check_random_state(random_state)
    x = np.asarray(x, dtype=F32)
    if n_init == "auto":
        n_init = 10 if compat == "1.0.2" else 1
    best = None
    for _ in range(n_init):
        picks = sklearn_kpp(x, k, rs, compat)
        lab, inertia, cent, n_iter = sklearn_lloyd(x, x[picks], max_iter, tol)
        if best is None:
            better = True
        elif compat == "1.0.2":
            better = inertia < best[1] * (1 - 1e-6)
        else:
            better = inertia < best[1] and not _same_clustering(lab, best[0], k)
        if better:
            best = (lab, inertia, cent, n_iter)
    return best
