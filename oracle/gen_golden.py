#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run from the repo root:   PYTHONHASHSEED=0 python -m oracle.gen_golden
Needs /root/reference (read-only); it does not exist on the GPU box, which is
why the outputs are committed as small fixtures.  Nothing here is copied into
the repo: reference modules are imported (with stub modules for the packages
that are not installed: clip, nltk, ortools/k_means_constrained, tensorboard,
sklearn.utils._joblib) and, for the hot loops that the reference inlines under
`if __name__ == "__main__"` (main_unsup.py:504-531,568-614; main_ptsup.py:629-676),
the corresponding LINE RANGE of the reference file is read at generation time
and exec'd in a prepared namespace.
"""
import os
import sys
import types
import warnings
from collections import Counter
import copy

import numpy as np
import torch

REF = os.environ.get("SCD_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
warnings.filterwarnings("ignore")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs(mcf_solver):
    import joblib
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    _stub("sklearn.utils._joblib", Parallel=joblib.Parallel, delayed=joblib.delayed,
          effective_n_jobs=joblib.effective_n_jobs)
    nltk = _stub("nltk")
    corpus = _stub("nltk.corpus", wordnet=object())
    nltk.corpus = corpus
    _stub("clip", tokenize=lambda texts: texts)
    _stub("pyximport", install=lambda *a, **k: None)
    _stub("k_means_constrained")
    _stub("k_means_constrained.mincostflow_vectorized", SimpleMinCostFlowVectorized=mcf_solver)
    for p in (REF, os.path.join(REF, "gcd"), os.path.join(REF, "local_utils")):
        if p not in sys.path:
            sys.path.insert(0, p)


class NxMinCostFlow:
    """Stand-in with the SimpleMinCostFlowVectorized surface used at
    sskm_constrained.py:333-353, backed by networkx.network_simplex (exact).
    It exists only so the REST of the reference's constrained path (graph build,
    cost rounding, label extraction, M-step, restarts) can execute here."""
    OPTIMAL = 0

    def AddArcWithCapacityAndUnitCostVectorized(self, tail, head, cap, cost):
        self.arcs = (np.asarray(tail), np.asarray(head), np.asarray(cap), np.asarray(cost))

    def SetNodeSupplyVectorized(self, node, supply):
        self.supply = np.asarray(supply)

    def Solve(self):
        import networkx as nx
        g = nx.DiGraph()
        for i, s in enumerate(self.supply):
            g.add_node(i, demand=-int(s))
        t, h, c, w = self.arcs
        for a in range(len(t)):
            g.add_edge(int(t[a]), int(h[a]), capacity=int(c[a]), weight=int(w[a]))
        try:
            self.cost, self.flow = nx.network_simplex(g)
        except nx.NetworkXUnfeasible:
            return 1
        return self.OPTIMAL

    def FlowVectorized(self, arc):
        t, h, _, _ = self.arcs
        return np.array([self.flow[int(t[a])][int(h[a])] for a in arc], dtype=np.int32)


def ref_lines(relpath, lo, hi):
    """Lines lo..hi (1-based, inclusive) of a reference file, de-indented."""
    with open(os.path.join(REF, relpath)) as f:
        lines = f.readlines()[lo - 1:hi]
    pad = min(len(l) - len(l.lstrip()) for l in lines if l.strip())
    return "".join(l[pad:] if l.strip() else l for l in lines)


# ----------------------------------------------------------------------------- generators
def gen_munkres():
    from project_utils.cluster_utils import linear_assignment
    rs = np.random.RandomState(0)
    out = {}
    cases = [(4, 4, 3), (12, 12, 4), (12, 12, 2), (7, 11, 5), (11, 7, 5), (64, 64, 6), (64, 64, 50),
             (1, 1, 2), (40, 40, 1000), (96, 96, 3), (30, 50, 2)]
    for ci, (n, m, hi) in enumerate(cases):
        for rep in range(3):
            x = rs.randint(0, hi, size=(n, m)).astype(np.int64)
            out["cost_%d_%d" % (ci, rep)] = x
            out["ind_%d_%d" % (ci, rep)] = linear_assignment(x)
    # w.max()-w shaped like assign_name's input: sparse counts, zero padding
    for rep in range(3):
        d = 60
        w = np.zeros((d, d), dtype=np.int64)
        for i in range(25):
            cols = rs.choice(d, 4, replace=False)
            w[i, cols] += rs.randint(1, 40, size=4)
        x = w.max() - w
        out["cost_w_%d" % rep] = x
        out["ind_w_%d" % rep] = linear_assignment(x)
    # assign_name-shaped instances at sizes where the sparse solver's bookkeeping (row / column classes, step 6 on the
    # potentials) is exercised: K cluster rows with <= 4 voted names each, popular names shared between clusters (conflicts),
    # D - K all-zero padding rows.  Stored as the non-zero entries of w; solved by the reference on w.max() - w.
    for rep, (k, d, pool) in enumerate(((40, 160, 30), (60, 250, 45), (25, 300, 12), (90, 120, 100))):
        w = np.zeros((d, d), dtype=np.int64)
        for i in range(k):
            cols = np.unique(rs.zipf(1.4, size=4) % pool)
            w[i, cols] += rs.randint(1, 25, size=len(cols))
        r, c = np.nonzero(w)
        out["vote_rows_%d" % rep], out["vote_cols_%d" % rep], out["vote_vals_%d" % rep] = r, c, w[r, c]
        out["vote_d_%d" % rep] = np.array(d)
        out["vote_ind_%d" % rep] = linear_assignment(w.max() - w)
    np.savez_compressed(os.path.join(OUT, "munkres.npz"), **out)
    print("munkres:", len([k for k in out if k.startswith("ind_") or k.startswith("vote_ind_")]), "cases")


def gen_acc_v2():
    from project_utils.cluster_and_log_utils import split_cluster_acc_v2
    gt = np.array([0] * 5 + [1] * 5 + [2] * 5 + [3] * 5)
    mask = gt < 2
    preds = np.array([2] * 4 + [0] * 1 + [1] * 4 + [3] * 1 + [0] * 4 + [3] * 1 + [3] * 5)
    t, o, n, m = split_cluster_acc_v2(gt, preds, mask, return_ind_map=True)
    assert (t, o, n) == (0.85, 0.8, 0.9) and m == {2: 0, 1: 1, 0: 2, 3: 3}   # notebook cell 2 output
    rs = np.random.RandomState(3)
    y = rs.randint(0, 12, size=600)
    p = np.where(rs.rand(600) < 0.7, (y * 5 + 3) % 12, rs.randint(0, 12, size=600))
    mk = y < 6
    t2, o2, n2, m2 = split_cluster_acc_v2(y, p, mk, return_ind_map=True)
    np.savez_compressed(os.path.join(OUT, "acc_v2.npz"), gt=gt, mask=mask, preds=preds, res=np.array([t, o, n]),
                        map_k=np.array(list(m.keys())), map_v=np.array(list(m.values())),
                        y2=y, p2=p, mask2=mk, res2=np.array([t2, o2, n2]),
                        map2_k=np.array(list(m2.keys())), map2_v=np.array(list(m2.values())))
    print("acc_v2 ok", t2, o2, n2)


def _blob_case(n, d, k, seed):
    from oracle import synth
    return synth.blob_case(n, d, k, seed)


def gen_kmeans():
    import methods.clustering.faster_mix_k_means_pytorch as sskm   # gcd copy (the one the mains import)
    out = {}
    cases = [("a", 500, 8, 4, 1), ("b", 1500, 32, 10, 2), ("c", 3000, 768, 20, 3)]
    for tag, n, d, k, seed in cases:
        x, y, mask_lab = _blob_case(n, d, k, seed)
        l, u = torch.from_numpy(x[mask_lab]), torch.from_numpy(x[~mask_lab])
        lt = torch.from_numpy(y[mask_lab])
        km = sskm.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=3,
                          random_state=seed, n_jobs=None, pairwise_batch_size=1024)
        km.fit_mix(u, l, lt)
        out["%s_shape" % tag] = np.array([n, d, k, seed])        # inputs = synth.blob_case(n, d, k, seed)
        out["%s_cfg" % tag] = np.array([k, 10, 3, seed])
        out["%s_labels" % tag] = km.labels_.numpy()
        out["%s_centers" % tag] = km.cluster_centers_.numpy()
        out["%s_inertia" % tag] = np.array(float(km.inertia_))
        out["%s_n_iter" % tag] = np.array(int(km.n_iter_))
        # k-means++ picks of the FIRST restart: replay kpp with a fresh RandomState
        l_cent = torch.stack([l[lt == c].mean(0) for c in torch.unique(lt)])
        c = km.kpp(u, l_cent, k=k, random_state=np.random.RandomState(seed))
        added = c[len(l_cent):]
        picks = [int(torch.nonzero((u == row).all(dim=1))[0][0]) for row in added]
        out["%s_kpp_picks" % tag] = np.array(picks)
        # plain fit (unlabelled only)
        km2 = sskm.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=2,
                           random_state=seed + 1, n_jobs=None, pairwise_batch_size=512)
        km2.fit(u)
        out["%s_fit_labels" % tag] = km2.labels_.numpy()
        out["%s_fit_centers" % tag] = km2.cluster_centers_.numpy()
        out["%s_fit_inertia" % tag] = np.array(float(km2.inertia_))
        print("kmeans", tag, "inertia", float(km.inertia_), "n_iter", int(km.n_iter_), "fit", float(km2.inertia_))
    # pairwise_distance itself
    rs = np.random.RandomState(9)
    a = rs.randn(257, 40).astype(np.float32)
    b = rs.randn(13, 40).astype(np.float32)
    out["pd_a"], out["pd_b"] = a, b
    out["pd_batched"] = sskm.pairwise_distance(torch.from_numpy(a), torch.from_numpy(b), 100).numpy()
    out["pd_plain"] = sskm.pairwise_distance(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    np.savez_compressed(os.path.join(OUT, "kmeans_sskm.npz"), **out)


def gen_kmeans16():
    """The reference's K_Means (gcd copy) on inputs that are EXACT in fp16 - what an fp16 encoder hands the product path, the condition
    under which it runs the MFMA filters, the lock-step restarts and the incremental M-step - at the CLIP and DINO feature widths."""
    import methods.clustering.faster_mix_k_means_pytorch as sskm
    out = {}
    for tag, n, d, k, seed, n_init in [("h", 6000, 512, 30, 5, 4), ("i", 4000, 768, 12, 6, 3)]:
        x, y, mask_lab = _blob_case(n, d, k, seed)
        x = x.astype(np.float16).astype(np.float32)
        l, u = torch.from_numpy(x[mask_lab]), torch.from_numpy(x[~mask_lab])
        lt = torch.from_numpy(y[mask_lab])
        km = sskm.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=n_init,
                          random_state=seed, n_jobs=None, pairwise_batch_size=1024)
        km.fit_mix(u, l, lt)
        out["%s_shape" % tag] = np.array([n, d, k, seed])        # inputs = fp16(synth.blob_case(n, d, k, seed))
        out["%s_cfg" % tag] = np.array([k, 10, n_init, seed])
        out["%s_labels" % tag] = km.labels_.numpy()
        out["%s_centers" % tag] = km.cluster_centers_.numpy()
        out["%s_inertia" % tag] = np.array(float(km.inertia_))
        out["%s_n_iter" % tag] = np.array(int(km.n_iter_))
        km2 = sskm.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=2,
                           random_state=seed + 1, n_jobs=None, pairwise_batch_size=512)
        km2.fit(u)
        out["%s_fit_labels" % tag] = km2.labels_.numpy()
        out["%s_fit_centers" % tag] = km2.cluster_centers_.numpy()
        out["%s_fit_inertia" % tag] = np.array(float(km2.inertia_))
        print("kmeans16", tag, "inertia", float(km.inertia_), "n_iter", int(km.n_iter_), "fit", float(km2.inertia_))
    np.savez_compressed(os.path.join(OUT, "kmeans_f16.npz"), **out)


def gen_sklearn_kmeans():
    """sklearn.cluster.KMeans as the reference calls it for --cluster KM (main_unsup.py:362), pinned the way SURVEY.md 8c says:
    this container's scikit-learn with explicit init, n_init=1, algorithm='lloyd', so that only the Lloyd arithmetic is compared."""
    import sklearn
    from sklearn.cluster import KMeans
    from oracle import synth
    out = {"sklearn_version": np.array(sklearn.__version__)}
    cases = [("a", 600, 8, 5, 1, 0.8), ("b", 2500, 64, 12, 2, 0.9), ("c", 4000, 768, 20, 3, 0.8), ("e", 900, 16, 6, 4, 0.7)]
    for tag, n, d, k, seed, noise in cases:
        x, y, cent = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 40, noise=noise)
        rs = np.random.RandomState(seed)
        init = x[rs.choice(n, k, replace=False)].copy()
        if tag == "e":
            init[2] = 50.0                       # a centre nobody is closest to: exercises _relocate_empty_clusters_dense
        km = KMeans(n_clusters=k, init=init, n_init=1, algorithm="lloyd", random_state=0).fit(x)
        out["%s_shape" % tag] = np.array([n, d, k, seed])
        out["%s_noise" % tag] = np.array(noise)
        out["%s_init" % tag] = init
        out["%s_labels" % tag] = km.labels_
        out["%s_centers" % tag] = km.cluster_centers_
        out["%s_inertia" % tag] = np.array(float(km.inertia_))
        out["%s_n_iter" % tag] = np.array(int(km.n_iter_))
        print("sklearn KMeans", tag, "inertia", km.inertia_, "n_iter", km.n_iter_)
        # the call the reference makes (main_unsup.py:362, main_ptsup.py:381): k-means++ seeded, random_state=0.  Three pins:
        # the public seeding function, the default call of this scikit-learn (n_init='auto' -> 1 start) and ten starts on one
        # RandomState (n_init=10, the default of the scikit-learn 1.0.2 that requirements.txt pins)
        from sklearn.cluster import kmeans_plusplus
        _, picks = kmeans_plusplus(x, k, random_state=0)
        out["%s_kpp_picks" % tag] = picks.astype(np.int64)
        for name, kw in (("default", {}), ("n10", {"n_init": 10})):
            kd = KMeans(n_clusters=k, random_state=0, **kw).fit(x)
            out["%s_%s_labels" % (tag, name)] = kd.labels_
            out["%s_%s_inertia" % (tag, name)] = np.array(float(kd.inertia_))
            out["%s_%s_n_iter" % (tag, name)] = np.array(int(kd.n_iter_))
            print("   ", name, "inertia", kd.inertia_, "n_iter", kd.n_iter_)
    np.savez_compressed(os.path.join(OUT, "kmeans_sklearn.npz"), **out)


def _reference_k_init():
    """The reference's own k-means++ (`_k_init`, scikit-learn 0.19's pure-Python seeding, vendored under
    local_utils/k_means_constrained/sklearn_import/cluster/k_means_.py:33-132) with the vendored helpers it calls
    (metrics/pairwise.py `euclidean_distances` :20-114, `check_pairwise_arrays` :246-313, `_return_float_dtype` :556-577;
    utils/extmath.py `row_norms` :10-27, `stable_cumsum` :93-119, `safe_sparse_dot` :122-149), exec'd from their line ranges:
    the modules themselves import Cython extensions built for another Python.  Draw sequence = scikit-learn 1.0.2's
    `_kmeans_plusplus` (randint first centre, `random_sample(n_local_trials) * current_pot`, searchsorted on the stable cumsum)."""
    import scipy.sparse as sp
    from scipy.sparse import issparse, csr_matrix
    from sklearn.utils.validation import check_array as _check_array

    def check_array(a, **kw):           # input validation only; 0.19's keywords that today's validator dropped are ignored
        kw.pop("warn_on_dtype", None)
        kw.pop("estimator", None)
        return _check_array(a, **kw)
    base = "local_utils/k_means_constrained/sklearn_import/"
    ns = dict(np=np, sp=sp, issparse=issparse, csr_matrix=csr_matrix, warnings=warnings, check_array=check_array,
              np_version=tuple(int(v) for v in np.__version__.split(".")[:2]))
    for rel, lo, hi in (("utils/extmath.py", 10, 27), ("utils/extmath.py", 93, 119), ("utils/extmath.py", 122, 149),
                        ("metrics/pairwise.py", 556, 577), ("metrics/pairwise.py", 246, 313), ("metrics/pairwise.py", 20, 114),
                        ("cluster/k_means_.py", 33, 132)):
        exec(ref_lines(base + rel, lo, hi), ns)
    return ns["_k_init"], ns["row_norms"]


def gen_sklearn_kinit():
    """Pins for the DEFAULT mode of `--cluster KM` (scikit-learn 1.0.2 semantics, the reference's requirements.txt pin): the
    picks of the reference-held `_k_init` on the four cases of kmeans_sklearn.npz and on a CUB-shaped case (BASELINE configs[0]:
    4,500 x 768, K = 200), `RandomState(0)` as `KMeans(random_state=0)` seeds it, for the float32 rows `KMeans.fit` would pass
    and for their float64 copy.  Ten consecutive seedings on ONE RandomState = the stream of the n_init = 10 default.
    Stored next to the existing goldens in kmeans_sklearn.npz (keys `*_kinit_*`)."""
    from oracle import synth
    k_init, row_norms = _reference_k_init()
    path = os.path.join(OUT, "kmeans_sklearn.npz")
    out = dict(np.load(path))
    cases = [("a", 600, 8, 5, 1, 0.8), ("b", 2500, 64, 12, 2, 0.9), ("c", 4000, 768, 20, 3, 0.8), ("e", 900, 16, 6, 4, 0.7),
             ("c1", 4500, 768, 200, 5, 0.6)]
    for tag, n, d, k, seed, noise in cases:
        x, y, cent = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 40, noise=noise)
        out["%s_shape" % tag] = np.array([n, d, k, seed])
        out["%s_noise" % tag] = np.array(noise)
        for dt, name in ((np.float32, "f32"), (np.float64, "f64")):
            xs = x.astype(dt)
            rs = np.random.RandomState(0)
            picks = []
            for _ in range(10 if tag != "c1" else 2):
                centers = k_init(xs, k, row_norms(xs, squared=True), rs)
                # rows of X are distinct, so a centre identifies its row
                idx = [int(np.nonzero((xs == c).all(axis=1))[0][0]) for c in centers]
                picks.append(idx)
            out["%s_kinit_%s" % (tag, name)] = np.array(picks, dtype=np.int64)
            out["%s_kinit_%s_next" % (tag, name)] = np.array(rs.random_sample())      # position in the stream afterwards
        same = np.array_equal(out["%s_kinit_f32" % tag], out["%s_kinit_f64" % tag])
        print("_k_init", tag, "first start", out["%s_kinit_f32" % tag][0][:8], "f32 == f64 picks:", same)
    np.savez_compressed(path, **out)


def gen_constrained():
    import sskm_constrained as con           # local_utils/sskm_constrained.py
    out = {}
    # graph arrays (pure numpy in the reference)
    rs = np.random.RandomState(4)
    d = np.abs(rs.randn(9, 3)).astype(np.float32)
    e, c, cap, sup, n_c, n_x = con.minimum_cost_flow_problem_graph(np.zeros((9, 2)), np.zeros((3, 2)), d, 2, 5)
    out.update(g_d=d, g_edges=e, g_costs=c, g_caps=cap, g_sup=sup)
    # docstring KAT of the vendored estimator (k_means_constrained_.py:777-793): labels [0,0,0,1,1,1]
    x6 = np.array([[1, 2], [1, 4], [1, 0], [4, 2], [4, 4], [4, 0]], dtype=np.float32)
    km = con.K_Means(k=2, size_min=2, size_max=5, random_state=0, n_init=10, max_iterations=100)
    km.fit(torch.from_numpy(x6))
    out.update(kat_x=x6, kat_labels=km.labels_.numpy(), kat_centers=km.cluster_centers_.numpy())
    # the print-only script local_utils/test_kmeans_cons.py: 9x2 integer array
    x9 = np.array([[1, 2], [1, 4], [1, 0], [4, 2], [4, 4], [4, 0], [2, 2], [3, 3], [0, 1]], dtype=np.float32)
    km9 = con.K_Means(k=2, size_min=2, size_max=5, random_state=0)
    km9.fit(torch.from_numpy(x9))
    out.update(x9=x9, x9_labels=km9.labels_.numpy(), x9_inertia=np.array(float(km9.inertia_)))
    # fit_mix on blobs with tight size bounds
    x, y, mask_lab = _blob_case(400, 16, 6, 21)
    l, u = torch.from_numpy(x[mask_lab]), torch.from_numpy(x[~mask_lab])
    lt = torch.from_numpy(y[mask_lab])
    km = con.K_Means(k=6, tolerance=1e-4, max_iterations=5, init="k-means++", size_min=30, size_max=80,
                     n_init=2, random_state=5, n_jobs=None, pairwise_batch_size=128)
    km.fit_mix(u, l, lt)
    cnt = np.bincount(km.labels_.numpy()[len(lt):], minlength=6)
    out.update(m_shape=np.array([400, 16, 6, 21]), m_labels=km.labels_.numpy(),
               m_centers=km.cluster_centers_.numpy(), m_inertia=np.array(float(km.inertia_)), m_counts=cnt)
    # one raw constrained assignment with its optimal cost
    d2 = con.pairwise_distance(u, km.cluster_centers_, 64)
    dist = np.zeros(len(u), dtype=np.float32)
    lab, inertia = con._labels_constrained(u.numpy(), km.cluster_centers_.numpy(), torch.sqrt(d2).numpy(),
                                           30, 80, dist)
    cost = np.around(torch.sqrt(d2).numpy() * 1000, 0).astype(np.int32)
    out.update(a_d2=d2.numpy(), a_labels=lab, a_inertia=np.array(float(inertia)),
               a_total=np.array(int(cost[np.arange(len(lab)), lab].sum())))
    np.savez_compressed(os.path.join(OUT, "kmeans_constrained.npz"), **out)
    print("constrained: kat labels", km.labels_.numpy() if False else out["kat_labels"], "counts", cnt)


def gen_naming():
    import clip_lang_util as clu
    from oracle import synth
    out = {}
    # assign_name on synthetic counters
    rs = np.random.RandomState(2)
    for rep in range(3):
        ncl, nn = 8 + rep, 30
        c2c = {}
        for i in range(ncl):
            keys = rs.choice(nn, 6, replace=False)
            c2c[int(i * 3 + 1)] = Counter({int(k): int(v) for k, v in zip(keys, rs.randint(1, 20, size=6))})
        voted = list(set(k for c in c2c.values() for k, _ in c.most_common(5)))
        ind, w = clu.assign_name(voted, c2c, num_common=3)
        out["an%d_voted" % rep] = np.array(voted)
        out["an%d_keys" % rep] = np.array([[ck] * 6 for ck in c2c]).reshape(-1)
        out["an%d_names" % rep] = np.array([k for c in c2c.values() for k in c.keys()])
        out["an%d_counts" % rep] = np.array([v for c in c2c.values() for v in c.values()])
        out["an%d_ind" % rep] = ind
        out["an%d_w" % rep] = w
    # accuracy()
    logits = torch.from_numpy(rs.randn(50, 20).astype(np.float32))
    tgt = torch.from_numpy(rs.randint(0, 20, size=50))
    out["acc_logits"], out["acc_target"] = logits.numpy(), tgt.numpy()
    out["acc_res"] = np.array(clu.accuracy(logits, tgt, topk=(1, 5)))

    # zeroshot_classifier with a deterministic fake model (pins normalise->mean->normalise->stack(dim=1))
    class Fake:
        def encode_text(self, texts):
            return torch.stack([torch.from_numpy(np.random.RandomState(abs(hash(t)) % (2 ** 31)).randn(16)
                                                 .astype(np.float32)) for t in texts])
    sys.modules["clip"].tokenize = lambda texts: _NoCuda(texts)
    tmpl = clu.imagenet_templates[:7]
    names = ["alpha", "beta_gamma", "delta"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    zs = clu.zeroshot_classifier(names, tmpl, Fake())
    embs = np.stack([np.stack([np.random.RandomState(abs(hash(t.format(n))) % (2 ** 31)).randn(16).astype(np.float32)
                               for t in tmpl]) for n in names])
    out["zs_embs"], out["zs_out"] = embs, zs.numpy()
    out["n_templates"] = np.array(len(clu.imagenet_templates))

    # ---- top-k block of main_unsup.py:504-531 and vote loops, run from the reference text
    n, dclip, k, v = 1200, 64, 12, 400
    x, y, cent = synth.clustered_features(n, dclip, k, seed=31, center_seed=32, noise=0.9)
    w = synth.vocabulary(v, dclip, cent, seed=33, jitter=0.5, dtype=np.float32)
    nouns = synth.nouns_list(v)
    perm, mask_lab = synth.labelled_split(y, k, prop=0.5, seed=34)
    x, y = x[perm], y[perm]
    ns = dict(torch=torch, F=torch.nn.functional, tqdm=lambda z: z, clip_all_feats=torch.from_numpy(x),
              zeroshot_weights=torch.from_numpy(w), args=types.SimpleNamespace(topk=5))
    exec(ref_lines("main_unsup.py", 504, 531), ns)
    out["tk_x"], out["tk_w"] = x, w
    out["tk_idx_unsup"] = ns["name_idx_top5"].numpy()
    out["tk_val_unsup"] = ns["name_logits_top5"].numpy()
    ns2 = dict(ns)
    exec(ref_lines("main_ptsup.py", 526, 545), ns2)
    out["tk_idx_ptsup"] = ns2["name_idx_top5"].numpy()
    out["tk_val_ptsup"] = ns2["name_logits_top5"].numpy()

    # unsupervised vote loop (main_unsup.py:568-614) from imperfect initial clusters
    rs = np.random.RandomState(35)
    u_preds0 = np.where(rs.rand(n) < 0.8, (y * 7 + 2) % k, rs.randint(0, k, size=n))
    trace = []
    src = ref_lines("main_unsup.py", 568, 614) + "    _trace(voted_unique_name_idx, ind, cand_names, u_preds)\n"
    ns3 = dict(torch=torch, np=np, Counter=Counter, copy=copy, assign_name=clu.assign_name, print=lambda *a, **k: None,
               name_idx_top5=ns["name_idx_top5"], u_preds=u_preds0.copy(), clip_u_feats=torch.from_numpy(x),
               zeroshot_weights=torch.from_numpy(w), nouns=nouns, num_unlab_classes=k, top_k=5, it=0,
               cur_voted_names=[0], prev_voted_names=[1],
               args=types.SimpleNamespace(num_common_vote=10, num_common_linear=2),
               _trace=lambda vo, ind, cand, up: trace.append((np.array(vo), ind.copy(),
                                                              np.array([nouns.index(c) for c in cand]), up.copy())))
    exec(src, ns3)
    out["vu_preds0"] = u_preds0
    out["vu_cfg"] = np.array([k, 5, 10, 2])
    out["vu_iters"] = np.array(len(trace))
    for i, (vo, ind, cand, up) in enumerate(trace):
        out["vu_voted_%d" % i], out["vu_ind_%d" % i], out["vu_cand_%d" % i], out["vu_preds_%d" % i] = vo, ind, cand, up
    print("unsup vote loop iterations:", len(trace))

    # partially supervised vote loop (main_ptsup.py:588-676)
    n_lab_cls = k // 2
    lab_names = [nouns[c] for c in range(n_lab_cls)]          # gt name of class c is column c
    all_preds0 = np.where(rs.rand(n) < 0.85, y, rs.randint(0, k, size=n))
    all_preds0[mask_lab] = y[mask_lab]
    trace2 = []
    pre = ref_lines("main_ptsup.py", 588, 599) + ref_lines("main_ptsup.py", 602, 603) + \
        ref_lines("main_ptsup.py", 615, 618) + ref_lines("main_ptsup.py", 625, 625)
    body = ref_lines("main_ptsup.py", 629, 676) + \
        "    _trace(voted_unique_name_idx, ind, cand_names, u_preds, unlab_cluster_idx)\n"
    ns4 = dict(torch=torch, np=np, Counter=Counter, copy=copy, assign_name=clu.assign_name, print=lambda *a, **k: None,
               name_idx_top5=ns2["name_idx_top5"], name_logits_top5=ns2["name_logits_top5"],
               mask_lab=mask_lab, all_preds=all_preds0.copy(), clip_u_feats=torch.from_numpy(x[~mask_lab]),
               zeroshot_weights=torch.from_numpy(w), nouns=nouns,
               cidx_to_cname={c: nouns[c] for c in range(k)},
               args=types.SimpleNamespace(num_common_vote=10, num_common_linear=2, topk=5, n_cluster=k,
                                          train_classes=list(range(n_lab_cls))),
               _trace=lambda vo, ind, cand, up, uc: trace2.append(
                   (np.array(vo), ind.copy(), np.array([nouns.index(c) for c in cand]), up.copy(), np.array(uc))))
    exec(pre, ns4)
    exec(body, ns4)
    out["vp_mask_lab"], out["vp_all_preds0"] = mask_lab, all_preds0
    out["vp_cfg"] = np.array([k, n_lab_cls, 5, 10, 2])
    out["vp_iters"] = np.array(len(trace2))
    for i, (vo, ind, cand, up, uc) in enumerate(trace2):
        out["vp_voted_%d" % i], out["vp_ind_%d" % i], out["vp_cand_%d" % i] = vo, ind, cand
        out["vp_preds_%d" % i], out["vp_unlab_%d" % i] = up, uc
    print("ptsup vote loop iterations:", len(trace2))

    # ---- missing-name matching (row a7): the reference's own lines with a stand-in text classifier of the missing names.
    # Vocabulary = the 400 names above; 9 "class names": 4 are in the vocabulary, 5 are missing and sit near vocabulary columns
    # (two of them near the SAME column, so that the greedy top-5 of :459-469 has to move to a second choice).
    rs = np.random.RandomState(36)
    near = [20, 21, 300, 20, 77]
    mw = np.stack([w[:, c] + 0.35 * rs.randn(dclip).astype(np.float32) / np.sqrt(dclip) for c in near], axis=1)
    mw = (mw / np.linalg.norm(mw, axis=0, keepdims=True)).astype(np.float32)
    class_cols = [5, 9, 130, 399]                                  # class names that ARE vocabulary names
    miss_names = ["miss_%d" % i for i in range(len(near))]
    original_names = [nouns[c] for c in class_cols] + miss_names
    base = dict(torch=torch, zeroshot_classifier=lambda names, templates, model: torch.from_numpy(mw), imagenet_templates=None,
                model=None, zeroshot_weights=torch.from_numpy(w), nouns=nouns, miss_names=miss_names, print=lambda *a, **k: None)
    n1 = dict(base)
    exec(ref_lines("main_unsup.py", 402, 406), n1)                                        # cifar / aircraft: top-1 over all nouns
    n2 = dict(base, nouns_truncated=[n for n in nouns if n not in original_names])
    exec(ref_lines("main_unsup.py", 487, 491), n2)                                        # cub: top-1 over nouns_truncated
    n3 = dict(base, nouns_truncated=[n for n in nouns if n not in original_names])
    exec(ref_lines("main_unsup.py", 459, 469), n3)                                        # sdogs: greedy de-duplicated top-5
    out["mm_w"], out["mm_miss_w"], out["mm_class_cols"] = w, mw, np.array(class_cols)
    out["mm_top1_full"] = np.array([nouns.index(n) for n in n1["matched_names"]])
    out["mm_top1_trunc"] = np.array([nouns.index(n) for n in n2["matched_names"]])
    out["mm_greedy5_trunc"] = np.array([nouns.index(n) for n in n3["matched_names"]])
    print("missing names:", out["mm_top1_full"], out["mm_top1_trunc"], out["mm_greedy5_trunc"])
    np.savez_compressed(os.path.join(OUT, "naming.npz"), **out)


class _NoCuda(list):
    def cuda(self):
        return self


def gen_topk16():
    """The reference's own top-k blocks (main_unsup.py:504-531 softmax, main_ptsup.py:526-545 raw) and both vote loops
    (main_unsup.py:568-614, main_ptsup.py:588-676) on inputs that are EXACT in fp16 - what the HIP path is given - at the product
    kernel's shape class (d = 512, three batches of 1024 with a ragged last one).  Only the outputs are stored; the tests rebuild the
    inputs from the same seeds (oracle/synth.py)."""
    out = ref_topk_votes_f16(2100, 512, 25, 1500, (41, 42, 43, 44, 45))
    np.savez_compressed(os.path.join(OUT, "topk_f16.npz"), **out)
    print("topk_f16.npz written")


def ref_topk_votes_f16(n, d, k, v, seeds, noise=0.9, jitter=0.5):
    """The reference's top-k blocks and vote loops on fp16-exact synthetic inputs (gen_topk16; tools/ref_fuzz_naming.py runs it over many
    seeds against the oracle)."""
    import clip_lang_util as clu
    from oracle import synth
    s0, s1, s2, s3, s4 = seeds
    x, y, cent = synth.clustered_features(n, d, k, seed=s0, center_seed=s1, noise=noise)
    w = synth.vocabulary(v, d, cent, seed=s2, jitter=jitter, dtype=np.float32)
    perm, mask_lab = synth.labelled_split(y, k, prop=0.5, seed=s3)
    x, y = x[perm], y[perm]
    x16, w16 = x.astype(np.float16), w.astype(np.float16)
    xt, wt = torch.from_numpy(x16.astype(np.float32)), torch.from_numpy(w16.astype(np.float32))
    nouns = synth.nouns_list(v)
    ns = dict(torch=torch, F=torch.nn.functional, tqdm=lambda z: z, clip_all_feats=xt, zeroshot_weights=wt,
              args=types.SimpleNamespace(topk=5))
    exec(ref_lines("main_unsup.py", 504, 531), ns)
    ns2 = dict(ns)
    exec(ref_lines("main_ptsup.py", 526, 545), ns2)
    out = dict(shape=np.array([n, d, k, v]), seeds=np.array([s0, s1, s2, s3]),
               idx_unsup=ns["name_idx_top5"].numpy(), val_unsup=ns["name_logits_top5"].numpy(),
               idx_ptsup=ns2["name_idx_top5"].numpy(), val_ptsup=ns2["name_logits_top5"].numpy())
    # unsupervised vote loop from imperfect initial clusters
    rs = np.random.RandomState(s4)
    u_preds0 = np.where(rs.rand(n) < 0.8, (y * 7 + 2) % k, rs.randint(0, k, size=n))
    trace = []
    src = ref_lines("main_unsup.py", 568, 614) + "    _trace(voted_unique_name_idx, ind, cand_names, u_preds)\n"
    ns3 = dict(torch=torch, np=np, Counter=Counter, copy=copy, assign_name=clu.assign_name, print=lambda *a, **k: None,
               name_idx_top5=ns["name_idx_top5"], u_preds=u_preds0.copy(), clip_u_feats=xt, zeroshot_weights=wt, nouns=nouns,
               num_unlab_classes=k, top_k=5, it=0, cur_voted_names=[0], prev_voted_names=[1],
               args=types.SimpleNamespace(num_common_vote=10, num_common_linear=2),
               _trace=lambda vo, ind, cand, up: trace.append((np.array(vo), ind.copy(),
                                                              np.array([nouns.index(c) for c in cand]), up.copy())))
    exec(src, ns3)
    out["vu_preds0"], out["vu_cfg"], out["vu_iters"] = u_preds0, np.array([k, 5, 10, 2]), np.array(len(trace))
    for i, (vo, ind, cand, up) in enumerate(trace):
        out["vu_voted_%d" % i], out["vu_ind_%d" % i], out["vu_cand_%d" % i], out["vu_preds_%d" % i] = vo, ind, cand, up
    print("unsup vote loop iterations:", len(trace))
    # partially supervised vote loop
    n_lab_cls = k // 2
    all_preds0 = np.where(rs.rand(n) < 0.85, y, rs.randint(0, k, size=n))
    all_preds0[mask_lab] = y[mask_lab]
    trace2 = []
    pre = ref_lines("main_ptsup.py", 588, 599) + ref_lines("main_ptsup.py", 602, 603) + \
        ref_lines("main_ptsup.py", 615, 618) + ref_lines("main_ptsup.py", 625, 625)
    body = ref_lines("main_ptsup.py", 629, 676) + \
        "    _trace(voted_unique_name_idx, ind, cand_names, u_preds, unlab_cluster_idx)\n"
    ns4 = dict(torch=torch, np=np, Counter=Counter, copy=copy, assign_name=clu.assign_name, print=lambda *a, **k: None,
               name_idx_top5=ns2["name_idx_top5"], name_logits_top5=ns2["name_logits_top5"],
               mask_lab=mask_lab, all_preds=all_preds0.copy(), clip_u_feats=xt[torch.from_numpy(~mask_lab)],
               zeroshot_weights=wt, nouns=nouns, cidx_to_cname={c: nouns[c] for c in range(k)},
               args=types.SimpleNamespace(num_common_vote=10, num_common_linear=2, topk=5, n_cluster=k,
                                          train_classes=list(range(n_lab_cls))),
               _trace=lambda vo, ind, cand, up, uc: trace2.append(
                   (np.array(vo), ind.copy(), np.array([nouns.index(c) for c in cand]), up.copy(), np.array(uc))))
    exec(pre, ns4)
    exec(body, ns4)
    out["vp_mask_lab"], out["vp_all_preds0"] = mask_lab, all_preds0
    out["vp_cfg"], out["vp_iters"] = np.array([k, n_lab_cls, 5, 10, 2]), np.array(len(trace2))
    for i, (vo, ind, cand, up, uc) in enumerate(trace2):
        out["vp_voted_%d" % i], out["vp_ind_%d" % i], out["vp_cand_%d" % i] = vo, ind, cand
        out["vp_preds_%d" % i], out["vp_unlab_%d" % i] = up, uc
    print("ptsup vote loop iterations:", len(trace2))
    return out


def gen_encoders():
    from scd_amd.clip import weights as W
    out = {}
    # DINO ViT-B/16 skeleton from the reference file, 2 blocks to keep the fixture generator quick
    import models.vision_transformer as vits
    for layers, tag in ((2, "d2"), (12, "d12")):
        sd = W.synthetic_dino_state_dict(seed=1, layers=layers)
        from functools import partial
        m = vits.VisionTransformer(patch_size=16, embed_dim=768, depth=layers, num_heads=12, mlp_ratio=4,
                                   qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))   # = vit_base :257-261
        missing = m.load_state_dict(sd, strict=True)
        m.eval()
        img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(77))
        with torch.no_grad():
            out["%s_out" % tag] = m(img).numpy()
    # CLIP towers vs transformers.CLIPModel on shared weights
    from transformers import CLIPConfig, CLIPModel
    for layers, tag in ((2, "c2"), (12, "c12")):
        sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=layers, t_layers=layers))
        cfg = CLIPConfig(text_config=dict(hidden_size=512, intermediate_size=2048, num_hidden_layers=layers,
                                          num_attention_heads=8, max_position_embeddings=77, vocab_size=49408,
                                          hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=49407,
                                          bos_token_id=49406, pad_token_id=0),
                         vision_config=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=layers,
                                            num_attention_heads=12, image_size=224, patch_size=16,
                                            hidden_act="quick_gelu", layer_norm_eps=1e-5),
                         projection_dim=512)
        hf = CLIPModel(cfg).eval()
        hf.load_state_dict(_to_hf(sd, layers), strict=False)
        img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(78))
        tok = torch.zeros(3, 77, dtype=torch.long)
        g = torch.Generator().manual_seed(79)
        for i, ln in enumerate((3, 8, 20)):
            tok[i, 0] = 49406
            tok[i, 1:1 + ln] = torch.randint(1, 49405, (ln,), generator=g)
            tok[i, 1 + ln] = 49407
        with torch.no_grad():
            io = hf.get_image_features(pixel_values=img)
            to = hf.get_text_features(input_ids=tok, attention_mask=torch.ones_like(tok))
        io = io if torch.is_tensor(io) else io.pooler_output
        to = to if torch.is_tensor(to) else to.pooler_output
        out["%s_img" % tag], out["%s_txt" % tag], out["%s_tok" % tag] = io.numpy(), to.numpy(), tok.numpy()
    np.savez_compressed(os.path.join(OUT, "encoders.npz"), **out)
    print("encoders ok")


def _to_hf(sd, layers):
    """openai/CLIP key names -> transformers.CLIPModel key names (q/k/v split)."""
    o = {}
    o["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    o["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    o["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    o["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    o["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    o["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"]
    o["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"]
    o["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    o["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    o["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    o["text_model.final_layer_norm.weight"] = sd["ln_final.weight"]
    o["text_model.final_layer_norm.bias"] = sd["ln_final.bias"]
    o["text_projection.weight"] = sd["text_projection"].t().contiguous()
    o["logit_scale"] = sd["logit_scale"]
    for tower, src in (("vision_model", "visual.transformer.resblocks."), ("text_model", "transformer.resblocks.")):
        for i in range(layers):
            s, d = "%s%d." % (src, i), "%s.encoder.layers.%d." % (tower, i)
            w, b = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
            e = w.shape[1]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                o[d + "self_attn.%s.weight" % nm] = w[j * e:(j + 1) * e]
                o[d + "self_attn.%s.bias" % nm] = b[j * e:(j + 1) * e]
            o[d + "self_attn.out_proj.weight"] = sd[s + "attn.out_proj.weight"]
            o[d + "self_attn.out_proj.bias"] = sd[s + "attn.out_proj.bias"]
            o[d + "layer_norm1.weight"], o[d + "layer_norm1.bias"] = sd[s + "ln_1.weight"], sd[s + "ln_1.bias"]
            o[d + "layer_norm2.weight"], o[d + "layer_norm2.bias"] = sd[s + "ln_2.weight"], sd[s + "ln_2.bias"]
            o[d + "mlp.fc1.weight"], o[d + "mlp.fc1.bias"] = sd[s + "mlp.c_fc.weight"], sd[s + "mlp.c_fc.bias"]
            o[d + "mlp.fc2.weight"], o[d + "mlp.fc2.bias"] = sd[s + "mlp.c_proj.weight"], sd[s + "mlp.c_proj.bias"]
    return o


def main():
    if os.environ.get("PYTHONHASHSEED") != "0":
        print("re-run with PYTHONHASHSEED=0 (set-of-str iteration order at main_ptsup.py:664 depends on it)")
        sys.exit(2)
    os.makedirs(OUT, exist_ok=True)
    install_stubs(NxMinCostFlow)
    which = sys.argv[1:] or ["munkres", "acc", "kmeans", "sklearn", "kinit", "constrained", "naming", "topk16", "kmeans16", "encoders"]
    for w in which:
        dict(munkres=gen_munkres, acc=gen_acc_v2, kmeans=gen_kmeans, sklearn=gen_sklearn_kmeans, kinit=gen_sklearn_kinit, constrained=gen_constrained,
             naming=gen_naming, topk16=gen_topk16, kmeans16=gen_kmeans16, encoders=gen_encoders)[w]()


if __name__ == "__main__":
    main()
