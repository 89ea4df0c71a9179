"""The naming loop of the reference mains on the HIP ops: full-vocabulary top-k (main_unsup.py:504-531,
main_ptsup.py:526-545) and the iterative vote (main_unsup.py:568-614, main_ptsup.py:629-676).

Device side: similarity + top-k, per-cluster vote histograms (most_common), candidate-column gather, argmax
re-classification.  Host side (tiny): Python set / list order semantics of the reference and the Munkres call.
"""
import copy

import numpy as np
import torch

from . import ops
from .local_utils.clip_lang_util import assign_name


class TopCounter:
    """most_common() view of one cluster's device histogram (what collections.Counter gives the reference)."""

    def __init__(self, keys, counts):
        self._items = [(int(k), int(c)) for k, c in zip(keys, counts) if k >= 0]

    def most_common(self, n=None):
        return self._items if n is None else self._items[:n]


def full_vocab_topk(clip_feats, zeroshot_weights, topk, softmax, wt=None):
    """name_idx_top5, name_logits_top5 of main_unsup.py:504-531 (softmax=True) / main_ptsup.py:526-545 (False)."""
    if wt is None:
        wt = ops.transpose_f16(zeroshot_weights.to(torch.float16))
    return ops.sim_topk(clip_feats, wt, topk, "softmax" if softmax else "raw", 100.0)


def cluster_counters(name_idx, top_k, u_preds, clusters, m, known=None):
    keys, counts = ops.vote_hist(name_idx, top_k, u_preds, clusters, m, known)
    keys, counts = keys.cpu().numpy(), counts.cpu().numpy()
    return {c: TopCounter(keys[i], counts[i]) for i, c in enumerate(clusters)}


def _first_index(nouns):
    first = {}
    for j, n in enumerate(nouns):
        first.setdefault(n, j)
    return first


def vote_loop_unsup(name_idx, u_preds, clip_u_feats, wt, nouns, n_cluster, num_common_vote, num_common_linear,
                    on_iter=None, max_iter=1000):
    """main_unsup.py:568-614.  name_idx int64 [N_u, TOP_K] (device), u_preds int64 [N_u] (device or numpy),
    wt = zeroshot_weights.T fp16 [V, 512] (device).  Returns (cand_names, u_preds numpy, trace)."""
    dev = name_idx.device
    u_preds = torch.as_tensor(u_preds, dtype=torch.int64, device=dev)
    first = _first_index(nouns)
    top_k = min(5, name_idx.shape[1])                 # `top_k = 5` slices a [N, TOP_K] tensor (:561,577)
    m = max(num_common_vote, num_common_linear)
    cur, prev, cand, trace = [0], [1], list(nouns), []
    while set(cur) != set(prev) and len(trace) < max_iter:
        clusters = list(set(u_preds.cpu().numpy().tolist()))
        c2c = cluster_counters(name_idx, top_k, u_preds, clusters, m)
        voted = []
        for i in clusters:
            voted += [c[0] for c in c2c[i].most_common(num_common_vote)]
        voted = list(set(voted))
        ind, w = assign_name(voted, c2c, num_common=num_common_linear)
        prev = copy.deepcopy(cur)
        cur = [nouns[voted[x[1]]] for x in ind[:n_cluster]]
        cand = copy.deepcopy(cur)
        cols = torch.tensor([first[n] for n in cand], dtype=torch.int64, device=dev)
        w_sel = ops.gather_rows_f16(wt, cols)
        u_preds, _ = ops.sim_argmax(clip_u_feats, w_sel)
        trace.append(dict(voted=np.array(voted, dtype=np.int64), ind=ind, cand=cols.cpu().numpy(),
                          u_preds=u_preds.cpu().numpy()))
        if on_iter:
            on_iter(len(trace), cand, trace[-1]["u_preds"])
    return cand, (trace[-1]["u_preds"] if trace else u_preds.cpu().numpy()), trace


def vote_loop_ptsup(name_idx, all_preds, mask_lab, clip_u_feats, wt, nouns, lab_names, n_cluster, topk,
                    num_common_vote, num_common_linear, on_iter=None, max_iter=1000):
    """main_ptsup.py:588-676, including the `known_name_idx` quirk (:638,666): after the first iteration it holds
    candidate positions but is still compared with vocabulary indices."""
    dev = name_idx.device
    all_preds = np.asarray(all_preds)
    mask_lab = np.asarray(mask_lab, dtype=bool)
    u_preds = torch.as_tensor(all_preds[~mask_lab], dtype=torch.int64, device=dev)
    lab_class_index = list(set(all_preds[mask_lab].tolist()))
    all_class_index = list(set(all_preds.tolist()))
    cand = nouns
    num_unlab = n_cluster - len(lab_names)
    known = [cand.index(n) for n in lab_names]
    unlab_cluster_idx = list(set(all_class_index) - set(lab_class_index))
    m = max(num_common_vote, num_common_linear)
    cur, prev, trace = [0], [1], []
    while set(cur) != set(prev) and len(trace) < max_iter:
        c2c = cluster_counters(name_idx, topk, u_preds, unlab_cluster_idx, m, known)
        voted = []
        for i in unlab_cluster_idx:
            voted += [c[0] for c in c2c[i].most_common(num_common_vote)]
        voted = list(set(voted))
        ind, w = assign_name(voted, c2c, num_common=num_common_linear)
        prev = copy.deepcopy(cur)
        cur = [nouns[voted[x[1]]] for x in ind[:num_unlab]]
        cand = sorted(copy.deepcopy(list(set(cur + lab_names))))
        lab_class_index = [cand.index(n) for n in lab_names]
        unlab_cluster_idx = [cand.index(n) for n in list(set(cand) - set(lab_names))]
        known = copy.deepcopy(lab_class_index)
        cols = torch.tensor([nouns.index(n) for n in cand], dtype=torch.int64, device=dev)
        w_sel = ops.gather_rows_f16(wt, cols)
        u_preds, _ = ops.sim_argmax(clip_u_feats, w_sel)
        trace.append(dict(voted=np.array(voted, dtype=np.int64), ind=ind, cand=cols.cpu().numpy(),
                          u_preds=u_preds.cpu().numpy(), unlab_cluster_idx=np.array(unlab_cluster_idx, dtype=np.int64)))
        if on_iter:
            on_iter(len(trace), cand, trace[-1]["u_preds"])
    return cand, (trace[-1]["u_preds"] if trace else u_preds.cpu().numpy()), trace


def evaluate_semantic_acc(u_targets, cidx_to_cname, u_preds, cand_names):
    """main_unsup.py:149-167 (same signature and return value): sACC = fraction of unlabelled rows whose ground-truth class
    NAME equals the name voted for their cluster, as (average over the class names that occur, overall).  The reference
    loops over the rows in Python; here names are interned once and the rest is two bincounts."""
    import numpy as np
    t = np.asarray(u_targets).astype(np.int64).reshape(-1)
    p = np.asarray(u_preds.cpu() if hasattr(u_preds, "cpu") else u_preds).astype(np.int64).reshape(-1)
    classes = np.unique(t)
    intern = {}
    def nid(name):
        return intern.setdefault(name, len(intern))
    tname = np.array([nid(cidx_to_cname[int(c)]) for c in classes], dtype=np.int64)          # per class
    cname = np.array([nid(n) for n in cand_names], dtype=np.int64)                           # per cluster
    row_name = tname[np.searchsorted(classes, t)]
    hit = (row_name == cname[p]).astype(np.float64)
    # two classes may share a name: the reference keys its per-class lists by NAME
    tot = np.bincount(row_name, minlength=len(intern)).astype(np.float64)
    hits = np.bincount(row_name, weights=hit, minlength=len(intern))
    present = tot > 0
    return float((hits[present] / tot[present]).sum() / present.sum()), float(hit.sum() / len(hit))
