"""CPU oracle for the naming part of the SCD hot path.

TEST INFRASTRUCTURE ONLY - never imported from scd_amd/.

Restates (file:line under /root/reference):
  similarity + top-k          main_unsup.py:504-531, main_ptsup.py:526-545
  candidate re-classification main_unsup.py:601-614, main_ptsup.py:668-676
  vote histogram              main_unsup.py:573-586, main_ptsup.py:636-648
  assign_name                 local_utils/clip_lang_util.py:156-180
  linear_assignment (Munkres) gcd/project_utils/cluster_utils.py:234-493
  split_cluster_acc_v2        gcd/project_utils/cluster_and_log_utils.py:29-74
  zeroshot_classifier         local_utils/clip_lang_util.py:96-108
  accuracy                    local_utils/clip_lang_util.py:151-154

Decision semantics shared with the HIP path: logits are the float64 dot products
of the given (float16/float32) features, scaled by 100; top-k / argmax order is
(value descending, index ascending).  The reference evaluates the same
expression in the storage dtype (fp16 on GPU), where equal-after-rounding
logits are ordered arbitrarily by torch.topk; golden fixtures use separated
data so both agree.  Parity status: pinned (goldens from the reference run in
the build container: tests/golden/naming_*.npz, munkres.npz, acc_v2.npz).
"""
from collections import Counter
import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------- similarity
def logits64(f, w, scale=100.0):
    return scale * (np.asarray(f, dtype=F64) @ np.asarray(w, dtype=F64))


def topk_desc(vals, k):
    """Indices of the k largest per row, ordered (value desc, index asc)."""
    n, v = vals.shape
    # lexsort: last key is primary -> sort by -value then index
    idx = np.empty((n, k), dtype=np.int64)
    for i in range(n):
        row = vals[i]
        part = np.argpartition(-row, min(k + 8, v - 1))[: min(k + 9, v)] if v > k + 9 else np.arange(v)
        # ties on the boundary: widen to every element >= the k-th value
        kth = np.sort(row[part])[::-1][k - 1]
        cand = np.nonzero(row >= kth)[0]
        order = np.lexsort((cand, -row[cand]))
        idx[i] = cand[order[:k]]
    return idx


def sim_topk(f, w, k, mode="raw", scale=100.0):
    """(idx int64 [N,k], val float32 [N,k]).  mode 'softmax' returns softmax
    probabilities (main_unsup.py:527); indices are unchanged by the monotone map."""
    lg = logits64(f, w, scale)
    idx = topk_desc(lg, k)
    val = np.take_along_axis(lg, idx, axis=1)
    if mode == "softmax":
        m = lg.max(axis=1, keepdims=True)
        z = np.exp(lg - m).sum(axis=1, keepdims=True)
        val = np.exp(val - m) / z
    return idx, val.astype(np.float32)


def sim_argmax(f, w_sel, scale=100.0):
    lg = logits64(f, w_sel, scale)
    return np.argmax(lg, axis=1).astype(np.int64), lg.max(axis=1).astype(np.float32)


def l2norm_rows(x):
    x64 = np.asarray(x, dtype=F64)
    return (x64 / np.sqrt((x64 * x64).sum(-1, keepdims=True))).astype(np.asarray(x).dtype)


def accuracy(output, target, topk=(1,)):
    """clip_lang_util.accuracy (:151-154): counts of correct in top-k (not %)."""
    pred = topk_desc(np.asarray(output, dtype=F64), max(topk))
    correct = pred == np.asarray(target).reshape(-1, 1)
    return [float(correct[:, :k].sum()) for k in topk]


def zeroshot_classifier(classnames, templates, encode_text, tokenize):
    """clip_lang_util.zeroshot_classifier (:96-108): per name, normalise the prompt
    embeddings, mean, normalise; stack along dim=1 -> [D, n_names]."""
    cols = []
    for name in classnames:
        e = np.asarray(encode_text(tokenize([t.format(name) for t in templates])), dtype=F64)
        e = e / np.linalg.norm(e, axis=-1, keepdims=True)
        m = e.mean(axis=0)
        cols.append(m / np.linalg.norm(m))
    return np.stack(cols, axis=1)


# ----------------------------------------------------------------------------- Munkres
class _Munkres:
    """Kuhn-Munkres with the tie-breaking of the reference's vendored state machine
    (cluster_utils.py:316-493): zeros are starred / primed in row-major order."""

    def __init__(self, cost):
        cost = np.atleast_2d(np.asarray(cost))
        self.transposed = cost.shape[1] < cost.shape[0]
        self.c = (cost.T if self.transposed else cost).copy()
        n, m = self.c.shape
        self.ru = np.ones(n, dtype=bool)      # row uncovered
        self.cu = np.ones(m, dtype=bool)      # col uncovered
        self.mark = np.zeros((n, m), dtype=np.int8)   # 1 star, 2 prime

    def solve(self):
        c = self.c
        n, m = c.shape
        if n == 0 or m == 0:
            return np.zeros((0, 2), dtype=int)
        c -= c.min(axis=1)[:, None]
        ru, cu, mark = self.ru, self.cu, self.mark
        for i, j in zip(*np.nonzero(c == 0)):       # row-major
            if ru[i] and cu[j]:
                mark[i, j] = 1
                ru[i] = False
                cu[j] = False
        ru[:] = True
        cu[:] = True
        while True:
            stars = mark == 1
            cu[stars.any(axis=0)] = False
            if stars.sum() >= n:
                break
            z0 = None
            while z0 is None:
                zero = c == 0
                avail = zero & ru[:, None] & cu[None, :]
                while True:
                    flat = int(np.argmax(avail))
                    r, q = divmod(flat, m)
                    if not avail[r, q]:
                        break                       # no uncovered zero -> adjust
                    mark[r, q] = 2
                    sc = int(np.argmax(mark[r] == 1))
                    if mark[r, sc] != 1:
                        z0 = (r, q)
                        break
                    ru[r] = False
                    cu[sc] = True
                    avail[:, sc] = zero[:, sc] & ru
                    avail[r, :] = False
                if z0 is None:
                    if ru.any() and cu.any():
                        mv = c[ru][:, cu].min()
                        c[~ru] += mv
                        c[:, cu] -= mv
            # augment along the alternating path from z0
            path = [z0]
            while True:
                col = path[-1][1]
                r = int(np.argmax(mark[:, col] == 1))
                if mark[r, col] != 1:
                    break
                path.append((r, col))
                q = int(np.argmax(mark[r] == 2))
                path.append((r, q))
            for r, q in path:
                mark[r, q] = 0 if mark[r, q] == 1 else 1
            ru[:] = True
            cu[:] = True
            mark[mark == 2] = 0
        res = np.array(np.nonzero(mark == 1)).T
        if self.transposed:
            res = res[:, ::-1]
        return res


def linear_assignment(x):
    """cluster_utils.linear_assignment (:234-275): rows sorted, shape (-1,2)."""
    ind = _Munkres(x).solve().tolist()
    ind.sort()
    out = np.array(ind, dtype=int)
    out.shape = (-1, 2)
    return out


# ----------------------------------------------------------------------------- voting
def cluster_counters(name_idx, u_preds, clusters, top_k, known=None):
    """Counter per cluster in row-major insertion order (main_unsup.py:575-577;
    main_ptsup.py:637-638 drops names in `known`)."""
    name_idx = np.asarray(name_idx)
    u_preds = np.asarray(u_preds)
    out = {}
    for i in clusters:
        flat = name_idx[u_preds == i, :top_k].reshape(-1)
        if known is None:
            out[i] = Counter(x for x in flat)
        else:
            out[i] = Counter(x for x in flat if x not in known)
    return out


def assign_name(unique_name_idx, cluster_to_counter, num_common=4):
    """clip_lang_util.assign_name (:156-180)."""
    col = {u: j for j, u in enumerate(unique_name_idx)}
    keys = list(cluster_to_counter.keys())
    d = max(len(unique_name_idx), len(keys))
    w = np.zeros((d, d), dtype=int)
    for i, ck in enumerate(keys):
        for k, v in cluster_to_counter[ck].most_common(num_common):
            w[i, col[k]] += v
    return linear_assignment(w.max() - w), w


def vote_loop_unsup(name_idx, u_preds, f_u, w, nouns, n_cluster, topk, num_common_vote, num_common_linear,
                    max_iter=100):
    """main_unsup.py:568-614.  Returns the per-iteration trace."""
    name_idx = np.asarray(name_idx)
    u_preds = np.asarray(u_preds)
    top_k = 5                                        # main_unsup.py:561
    cur, prev = [0], [1]
    trace = []
    w64 = np.asarray(w, dtype=F64)
    f64 = np.asarray(f_u, dtype=F64)
    first = {}
    for j, n in enumerate(nouns):
        first.setdefault(n, j)                       # nouns.index -> first occurrence
    while set(cur) != set(prev) and len(trace) < max_iter:
        clusters = list(set(u_preds.tolist()))
        c2c = cluster_counters(name_idx, u_preds, clusters, top_k)
        voted = []
        for i in clusters:
            voted += [c[0] for c in c2c[i].most_common(num_common_vote)]
        voted = list(set(voted))
        ind, wmat = assign_name(voted, c2c, num_common=num_common_linear)
        prev = list(cur)
        cur = [nouns[voted[x[1]]] for x in ind[:n_cluster]]
        cand = list(cur)
        w_sel = np.stack([w64[:, first[n]] for n in cand], axis=1)
        u_preds = np.argmax(100.0 * (f64 @ w_sel), axis=-1).reshape(-1)
        trace.append(dict(voted=np.array(voted, dtype=np.int64), ind=ind.copy(),
                          cand=np.array([first[n] for n in cand], dtype=np.int64), u_preds=u_preds.copy()))
    return trace


def vote_loop_ptsup(name_idx, all_preds, mask_lab, f_u, w, nouns, lab_names, n_cluster, topk,
                    num_common_vote, num_common_linear, max_iter=100):
    """main_ptsup.py:588-676 (partially supervised variant),
    including the reference quirk that `known_name_idx` holds candidate positions
    after the first iteration but is still compared with vocabulary indices (:638,666)."""
    name_idx = np.asarray(name_idx)
    all_preds = np.asarray(all_preds)
    u_preds = all_preds[~mask_lab]
    l_preds = all_preds[mask_lab]
    lab_class_index = list(set(l_preds.tolist()))
    all_class_index = list(set(all_preds.tolist()))
    cand = nouns
    num_unlab = n_cluster - len(lab_names)
    known = [cand.index(n) for n in lab_names]
    unlab_cluster_idx = list(set(all_class_index) - set(lab_class_index))
    cur, prev = [0], [1]
    trace = []
    w64 = np.asarray(w, dtype=F64)
    f64 = np.asarray(f_u, dtype=F64)
    while set(cur) != set(prev) and len(trace) < max_iter:
        c2c = cluster_counters(name_idx, u_preds, unlab_cluster_idx, topk, known=known)
        voted = []
        for i in unlab_cluster_idx:
            voted += [c[0] for c in c2c[i].most_common(num_common_vote)]
        voted = list(set(voted))
        ind, wmat = assign_name(voted, c2c, num_common=num_common_linear)
        prev = list(cur)
        cur = [nouns[voted[x[1]]] for x in ind[:num_unlab]]
        cand = sorted(list(set(cur + lab_names)))
        lab_class_index = [cand.index(n) for n in lab_names]
        unlab_cluster_idx = [cand.index(n) for n in list(set(cand) - set(lab_names))]
        known = list(lab_class_index)
        w_sel = np.stack([w64[:, nouns.index(n)] for n in cand], axis=1)
        u_preds = np.argmax(100.0 * (f64 @ w_sel), axis=-1).reshape(-1)
        trace.append(dict(voted=np.array(voted, dtype=np.int64), ind=ind.copy(),
                          cand=np.array([nouns.index(n) for n in cand], dtype=np.int64),
                          u_preds=u_preds.copy(),
                          unlab_cluster_idx=np.array(unlab_cluster_idx, dtype=np.int64)))
    return trace


# ----------------------------------------------------------------------------- metrics
def split_cluster_acc_v2(y_true, y_pred, mask, return_ind_map=False):
    """cluster_and_log_utils.split_cluster_acc_v2 (:29-74)."""
    y_true = np.asarray(y_true).astype(int)
    y_pred = np.asarray(y_pred).astype(int)
    mask = np.asarray(mask, dtype=bool)
    old_gt = set(y_true[mask].tolist())
    new_gt = set(y_true[~mask].tolist())
    d = max(y_pred.max(), y_true.max()) + 1
    w = np.zeros((d, d), dtype=int)
    np.add.at(w, (y_pred, y_true), 1)
    ind = linear_assignment(w.max() - w)
    ind_map = {int(j): int(i) for i, j in ind}
    total = sum(w[i, j] for i, j in ind) * 1.0 / y_pred.size
    old_acc = sum(w[ind_map[i], i] for i in old_gt) / max(1, sum(w[:, i].sum() for i in old_gt))
    new_acc = sum(w[ind_map[i], i] for i in new_gt) / max(1, sum(w[:, i].sum() for i in new_gt))
    if return_ind_map:
        return total, old_acc, new_acc, ind_map
    return total, old_acc, new_acc


def evaluate_semantic_acc(u_targets, cidx_to_cname, u_preds, cand_names):
    """main_unsup.py:149-167: (per-class average sACC, overall sACC)."""
    per = {}
    hit_all = []
    for t, p in zip(u_targets, u_preds):
        name = cidx_to_cname[int(t)]
        h = 1 if name == cand_names[int(p)] else 0
        per.setdefault(name, []).append(h)
        hit_all.append(h)
    acc = {n: sum(v) / float(len(v)) for n, v in per.items()}
    return float(sum(acc.values())) / len(acc), sum(hit_all) / float(len(hit_all))


# ----------------------------------------------------------------------------- missing class names (row a7)
def match_missing_names(miss_w, w, nouns, nouns_truncated=None, mode="top1"):
    """main_unsup.py:402-406 (top-1 over the vocabulary), :487-491 (top-1 over nouns_truncated), :459-469 (greedy
    de-duplicated top-5 over nouns_truncated): miss_w [D, m] classifier of the missing names, w [D, V]."""
    f = np.asarray(miss_w, dtype=F64).T
    if nouns_truncated is None:
        pool, wp = nouns, np.asarray(w, dtype=F64)
    else:
        pool = nouns_truncated
        wp = np.stack([np.asarray(w, dtype=F64)[:, nouns.index(n)] for n in nouns_truncated], axis=1)
    lg = 100.0 * (f @ wp)
    if mode == "top1":
        return [pool[i] for i in topk_desc(lg, 1)[:, 0]]
    top5 = topk_desc(lg, 5)
    matched = []
    for i in range(f.shape[0]):
        j = 0
        idx = top5[i, j]
        while pool[idx] in matched:
            j += 1
            idx = top5[i, j]
        matched.append(pool[idx])
    return matched
