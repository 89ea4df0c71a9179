"""`KMeans` with the call surface and the algorithm of sklearn.cluster.KMeans as the reference uses it for `--cluster KM`
(/root/reference/main_unsup.py:362, main_ptsup.py:381: `KMeans(n_clusters=args.n_cluster, random_state=0).fit(u_feats).labels_`,
the default of scripts/evaluate_unsupervised.sh), on the HIP k-means kernels instead of the host's Cython/OpenMP Lloyd.

Which scikit-learn: /root/reference/requirements.txt pins scikit-learn 1.0.2; this image has 1.7.2.  The two differ in what the
reference's call means, so the class takes `sklearn_compat` (default: the reference's pin; env SCD_SKLEARN_COMPAT overrides):
  * "1.0.2" (default): n_init defaults to 10 starts on ONE RandomState, the first k-means++ centre is `random_state.randint(n)`,
    a start replaces the best when `inertia < best * (1 - 1e-6)`.  That version is not installed here, but the reference vendors
    its seeding routine: `_k_init` of /root/reference/local_utils/k_means_constrained/sklearn_import/cluster/k_means_.py:33-132 (the
    0.19 source, same algorithm and stream consumption as 1.0.2's `_kmeans_plusplus`).  oracle/gen_golden.py runs THAT function on
    seeded inputs (ten starts per case, BASELINE configs[0]'s 4,500 x 768, K = 200 among them) and the picks are committed in
    tests/golden/kmeans_sklearn.npz: the oracle (tests/test_oracle_golden.py) and the HIP lock-step seeding
    (test_sklearn_102_seeding_matches_reference_k_init) reproduce them pick for pick.  The Lloyd rules of the mode are restated
    from the public 1.0.2 source;
  * "1.7.2": n_init 'auto' -> 1 start for k-means++, first centre `random_state.choice(n, p=uniform)`, a start replaces the
    best when its inertia is smaller and its clustering differs (`_is_same_clustering`).  Pinned against scikit-learn 1.7.2
    itself: `kmeans_plusplus` picks, the default call's labels and the n_init=10 call's labels (tests/golden/kmeans_sklearn.npz).
Common to both (sklearn/cluster/_kmeans.py, third-party, not under /root/reference):
  * `_kmeans_plusplus`: per added centre 2 + floor(ln k) candidates drawn with `uniform(size=L) * pot` -> searchsorted on the
    cumulative closest distances; the candidate with the smallest new potential wins; the RandomState is consumed as sklearn does;
  * `_kmeans_single_lloyd` (1.0.2's default `elkan` takes the same decisions in exact arithmetic): E-step, centre update, empty
    clusters re-seeded with the points farthest from their centres (`_relocate_empty_clusters_dense`), stop on unchanged labels
    (strict) or `sum_k ||dc_k||^2 <= tol * mean(var(X))`, one more E-step when the stop was not strict; max_iter 300, tol 1e-4.
sklearn subtracts the column means from a float32 X first, for the accuracy of its float32 GEMM-form distances; distances here are
decided on float64 values (DESIGN.md "decision semantics"), which are translation invariant, so X is used as it is.
"""
import os

import numpy as np
import torch

from . import ops
from .kmeans import check_random_state


def _same_clustering(a, b, k):
    """sklearn `_is_same_clustering`: every label of `a` maps to ONE label of `b` (device tensors, int32)."""
    pairs = torch.unique(a.long() * k + b.long()).numel()
    return pairs == torch.unique(a).numel()


class KMeans:
    def __init__(self, n_clusters=8, *, init="k-means++", n_init="auto", max_iter=300, tol=1e-4, verbose=0, random_state=None,
                 copy_x=True, algorithm="lloyd", sklearn_compat=None):
        compat = sklearn_compat or os.environ.get("SCD_SKLEARN_COMPAT", "1.0.2")
        if compat not in ("1.0.2", "1.7.2"):
            raise ValueError("sklearn_compat must be '1.0.2' (the reference's pin) or '1.7.2' (got %r)" % (compat,))
        self.sklearn_compat = compat
        if algorithm not in ("lloyd", "auto", "full", "elkan"):
            raise ValueError("algorithm must be 'lloyd' / 'full' / 'auto' / 'elkan' (all run the Lloyd kernels; got %r)" % (algorithm,))
        self.n_clusters = n_clusters
        self.init = init
        self.n_init = n_init
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.random_state = random_state
        self.copy_x = copy_x
        self.algorithm = algorithm

    # ------------------------------------------------------------------ seeding (sklearn `_kmeans_plusplus`)
    def _draws(self, rs, n, starts):
        """What `starts` consecutive k-means++ seedings take from the RandomState, in sklearn's order: per start the first centre
        (1.0.2 / the reference's vendored `_k_init`: `randint(n)`; 1.7.2: `choice(n, p=uniform)`), then 2 + int(ln k) uniforms per
        added centre.  The Lloyd iterations between two seedings draw nothing, so the stream can be consumed up front."""
        k = self.n_clusters
        trials = 2 + int(np.log(k))
        first = np.empty(starts, dtype=np.int64)
        u = np.empty((starts, k - 1, trials), dtype=np.float64)
        for j in range(starts):
            if self.sklearn_compat == "1.0.2":
                first[j] = rs.randint(n)
            else:
                p = np.ones(n, dtype=np.float32)
                first[j] = rs.choice(n, p=p / p.sum())
            for c in range(k - 1):
                u[j, c] = rs.uniform(size=trials)
        return first, u

    def _seed(self, data, x16, rs, starts):
        """The greedy k-means++ seedings of `starts` consecutive starts, advanced in lock-step behind one call
        (scd_kpp_greedy_lockstep).  Returns float32 [starts, k, d] on the device."""
        first, u = self._draws(rs, data.n, starts)
        out = []
        step = max(1, min(starts, 64, 256 // u.shape[2]))           # the filter's list format holds 256 candidates per round
        for a in range(0, starts, step):
            cent, _ = ops.kpp_greedy_lockstep(data.x, x16, first[a:a + step], u[a:a + step], self.n_clusters)
            out.append(cent)
        return out[0] if len(out) == 1 else torch.cat(out)

    def _kpp(self, data, rs):
        """One seeding (the reference-held `_k_init` / sklearn's `_kmeans_plusplus`): centres float32 [k, d]."""
        return self._seed(data, ops.f16_exact(data.x), rs, 1)[0]

    # ------------------------------------------------------------------ sklearn `_kmeans_single_lloyd`
    def _lloyd(self, data, centers, tol_abs, lb=None):
        """lb: ops.LloydBuffers over the rows' exact fp16 copy -> the loop runs in C (scd_kmeans_lloyd_run_sk); an empty cluster
        (sklearn relocates it) or rows without such a copy take the loop below."""
        if lb is not None and lb.inc and os.environ.get("SCD_LLOYD_RUN", "1") != "0":
            lb.c0.copy_(centers)
            got = lb.run_sk(self.max_iter, tol_abs)
            if got is not None:
                labels, cen, n_iter = got
                inertia = float(ops.sum_f32(data.rowdist(cen, labels)).item())
                return labels, inertia, cen, n_iter
        x = data.x
        n, k = x.shape[0], self.n_clusters
        labels_old = torch.full((n,), -1, dtype=torch.int32, device=x.device)
        strict = False
        it = 0
        for it in range(self.max_iter):
            labels = data.estep(centers)
            sums, counts, _ = ops.kmeans_mstep(x, labels, None, k, 0)
            # the centres on the assumption that no cluster is empty (the common case), so that the iteration's three host-side
            # decisions - empty clusters, changed labels, centre shift - come back in ONE read instead of two (4,500 rows: the device
            # work of an iteration is ~0.1 ms, a read-back's round trip about as much); with an empty cluster the step is redone
            new_centers, shift = ops.kmeans_finalize(sums, counts, centers, shift_mode=1, data=data)      # (+ the next E-step's centre operands)
            flags = torch.cat([(counts == 0).sum().reshape(1).to(torch.float64), ops.labels_changed(labels, labels_old).to(torch.float64),
                               shift.reshape(1).to(torch.float64)]).cpu().numpy()
            if flags[0] > 0:
                self._relocate_empty(data, centers, labels, sums, counts, int(flags[0]))
                new_centers, shift = ops.kmeans_finalize(sums, counts, centers, shift_mode=1, data=data)
                flags[2] = float(shift.item())
            centers = new_centers
            if flags[1] == 0:
                strict = True
                break
            if float(flags[2]) <= tol_abs:
                break
            labels_old = labels
        if not strict:
            labels = data.estep(centers)
        inertia = float(ops.sum_f32(data.rowdist(centers, labels)).item())
        return labels, inertia, centers, it + 1

    @staticmethod
    def _relocate_empty(data, centers_old, labels, sums, counts, n_empty):
        """`_relocate_empty_clusters_dense`: the n_empty points farthest from their centres each become the only member of one
        empty cluster (their labels stay as they are this iteration, as in sklearn)."""
        dist = data.rowdist(centers_old, labels).cpu().numpy()
        far = np.argpartition(dist, -n_empty)[:-n_empty - 1:-1]
        empty = torch.nonzero(counts == 0).reshape(-1)
        lab = labels.cpu().numpy()
        for j in range(n_empty):
            row = data.x[int(far[j])].double()
            old = int(lab[far[j]])
            sums[old] -= row
            sums[empty[j]] = row
            counts[empty[j]] = 1
            counts[old] -= 1

    def fit(self, X, y=None, sample_weight=None):
        if sample_weight is not None:
            raise ValueError("sample_weight is not supported on the HIP path")
        xt = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32) if not torch.is_tensor(X) else X)
        if not xt.is_cuda:
            xt = xt.cuda()
        data = ops.KMeansData(xt.float())
        x = data.x
        x16 = ops.f16_exact(x)                    # features that left an fp16 encoder: filter seeding, incremental M-step, C loops
        lb = ops.LloydBuffers(data, x, x16, self.n_clusters) if x16 is not None else None
        n, d = x.shape
        if n < self.n_clusters:
            raise ValueError(f"n_samples={n} should be >= n_clusters={self.n_clusters}.")
        rs = check_random_state(self.random_state)
        # _tolerance: mean(var(X, axis=0)) * tol = inertia of X around its column means / (n d) * tol
        zeros = torch.zeros(n, dtype=torch.int32, device=x.device)
        s1, c1, _ = ops.kmeans_mstep(x, zeros, None, 1, 0)
        mu, _ = ops.kmeans_finalize(s1, c1, None)
        _, _, tot = ops.kmeans_mstep(x, zeros, mu, 1, 0)
        tol_abs = float(tot.sum().item()) / (n * d) * self.tol
        explicit = not isinstance(self.init, str)
        n_init = self.n_init
        if n_init == "auto":
            n_init = 10 if self.sklearn_compat == "1.0.2" else (1 if (explicit or self.init == "k-means++") else 10)
        if explicit:
            n_init = 1
        best = None
        seeds = self._seed(data, x16, rs, n_init) if (not explicit and self.init == "k-means++") else None
        for j in range(n_init):
            if explicit:
                centers = torch.as_tensor(np.asarray(self.init, dtype=np.float32)).to(x.device).contiguous()
            elif self.init == "k-means++":
                centers = seeds[j]
            elif self.init == "random":
                seeds = rs.choice(n, size=self.n_clusters, replace=False)
                centers = x[torch.as_tensor(seeds, device=x.device)].contiguous()
            else:
                raise ValueError("init must be 'k-means++', 'random' or an array")
            labels, inertia, centers, n_iter = self._lloyd(data, centers, tol_abs, lb)
            if best is None:
                better = True
            elif self.sklearn_compat == "1.0.2":
                better = inertia < best[1] * (1 - 1e-6)
            else:
                better = inertia < best[1] and not _same_clustering(labels, best[0], self.n_clusters)
            if better:
                best = (labels, inertia, centers, n_iter)
        self.labels_ = best[0].cpu().numpy().astype(np.int32)
        self.inertia_ = best[1]
        self.cluster_centers_ = best[2].cpu().numpy()
        self.n_iter_ = best[3]
        self.n_features_in_ = d
        return self

    def fit_predict(self, X, y=None, sample_weight=None):
        return self.fit(X, sample_weight=sample_weight).labels_
