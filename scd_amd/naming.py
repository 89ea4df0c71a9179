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


def full_vocab_topk_te(clip_feats, text_feats, wt, topk, softmax):
    """Top-k with textual enhancement (BASELINE configs[4]; README "Ours w/TE"): the released mains keep only the commented
    formula `logits = 100. * (clip_batch_feat @ W + closed_text_feat @ W) / 2` (main_unsup.py:518,523).  By linearity it is
    the plain similarity of the mean feature, so the same kernel re-ranks: one elementwise mean, then scd_sim_topk."""
    return ops.sim_topk(ops.mean2_f16(clip_feats, text_feats), wt, topk, "softmax" if softmax else "raw", 100.0)


def cluster_counters(name_idx, top_k, u_preds, clusters, m, known=None):
    keys, counts = ops.vote_hist(name_idx, top_k, u_preds, clusters, m, known)
    keys, counts = keys.cpu().numpy(), counts.cpu().numpy()
    return {c: TopCounter(keys[i], counts[i]) for i, c in enumerate(clusters)}


_FIRST = {}


def _first_index(nouns):
    """name -> its first position in `nouns` (`nouns.index(name)`), as a dict.  Built once per vocabulary list and kept (the last few
    lists): a 100,000-name list costs ~10 ms to index, every vote call needs it, and `nouns.index` itself - one linear search per candidate
    name and iteration, main_ptsup.py:664 - was 76 % of the partially supervised vote loop's time at K = 120, V = 21,000 (round 6).  A
    list changed in place since is noticed through the hash of its contents (the strings' own hashes are cached: ~0.3 ms at V = 21,000)."""
    key = (id(nouns), len(nouns), hash(tuple(nouns)))
    ent = _FIRST.get(key)
    if ent is not None and ent[0] is nouns:
        return ent[1]
    first = {}
    for j, n in enumerate(nouns):
        first.setdefault(n, j)
    if len(_FIRST) >= 4:
        _FIRST.clear()
    _FIRST[key] = (nouns, first)
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
    first = _first_index(nouns)                         # first[n] == nouns.index(n)
    num_unlab = n_cluster - len(lab_names)
    known = [first[n] if n in first else cand.index(n) for n in lab_names]     # (a missing name raises ValueError here, as .index does)
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
        cols = torch.tensor([first[n] for n in cand], dtype=torch.int64, device=dev)        # nouns.index(n): every candidate is a name of the list
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


# ----------------------------------------------------------------------------- missing class names
def match_missing_names(miss_names, nouns, wt, model, mode="top1", nouns_truncated=None, templates=None, miss_weights=None):
    """Closest vocabulary name for every data-set class name that is not in the vocabulary.

    mode "top1", nouns_truncated None   main_unsup.py:402-406 (cifar / aircraft): top-1 over the whole vocabulary
    mode "top1", nouns_truncated given  main_unsup.py:487-491 (cub): top-1 over the names that are not class names
    mode "greedy_top5"                  main_unsup.py:459-469 (sdogs + wikidog): top-5 over nouns_truncated, each class takes
                                        its best name that no earlier class has taken
    (the same code at main_ptsup.py:419-423, 477-487, 505-509).  wt = zeroshot_weights.T [V,512] fp16 on the device;
    the text classifier of the missing names is built by zeroshot_classifier on the HIP text tower unless `miss_weights`
    ([512, m]) is given.  Returns the list of matched names, one per missing name."""
    if len(miss_names) == 0:
        return []
    if miss_weights is None:
        from .local_utils.clip_lang_util import imagenet_templates, zeroshot_classifier
        miss_weights = zeroshot_classifier(list(miss_names), templates or imagenet_templates, model)
    f = miss_weights.t().contiguous().to(device=wt.device, dtype=torch.float16)                  # [m, 512]
    if nouns_truncated is None:
        pool, w_pool = nouns, wt
    else:
        first = _first_index(nouns)                                                                # nouns.index(n)
        cols = torch.tensor([first[n] for n in nouns_truncated], dtype=torch.int64, device=wt.device)
        pool, w_pool = nouns_truncated, ops.gather_rows_f16(wt, cols)
    if mode == "top1":
        idx, _ = ops.sim_topk(f, w_pool, 1, "raw", 100.0)
        return [pool[i] for i in idx[:, 0].cpu().numpy().tolist()]
    if mode != "greedy_top5":
        raise ValueError("mode must be 'top1' or 'greedy_top5'")
    top5 = ops.sim_topk(f, w_pool, 5, "raw", 100.0)[0].cpu().numpy()
    matched = []
    for i in range(len(miss_names)):
        j = 0
        idx = int(top5[i, j])
        while pool[idx] in matched:
            j += 1
            idx = int(top5[i, j])                          # IndexError past the fifth name, like the reference
        matched.append(pool[idx])
    return matched


def class_names_with_matches(class_to_idx, miss_names, matched_names):
    """cidx_to_cname of main_unsup.py:407-412: a class keeps its name when the vocabulary has it, else the matched one."""
    miss_names = list(miss_names)
    out = {}
    for name, idx in class_to_idx.items():
        out[idx] = name if name not in miss_names else matched_names[miss_names.index(name)]
    return out


def resolve_class_names(dataset_name, corpus, class_to_idx, nouns, wt, model):
    """The data-set branches of main_unsup.py:398-502 that need the text tower: `class_to_idx` {original class name: index}
    (what the reference reads off its dataset objects) -> cidx_to_cname with every name inside the vocabulary."""
    names = list(class_to_idx.keys())
    miss = [n for n in names if n not in nouns]
    if dataset_name in ('cifar10', 'cifar100', 'aircraft'):
        matched = match_missing_names(miss, nouns, wt, model, "top1")
    elif dataset_name == 'sdogs' and corpus == 'wikidog':
        trunc = [n for n in nouns if n not in names]
        matched = match_missing_names(miss, nouns, wt, model, "greedy_top5", nouns_truncated=trunc)
    elif dataset_name == 'cub':
        trunc = [n for n in nouns if n not in names]
        matched = match_missing_names(miss, nouns, wt, model, "top1", nouns_truncated=trunc)
    else:                                                   # imagenet_*: names come from the WordNet ids, none is missing
        return {idx: name for name, idx in class_to_idx.items()}
    print(f'Missed {len(miss)} names and matched {len(set(matched))} names ... ')
    return class_names_with_matches(class_to_idx, miss, matched)


# ----------------------------------------------------------------------------- zero-shot bounds of main_ptsup.py
def _as_wt(zeroshot_weights):
    """[512, V] classifier (any float dtype, host or device) -> name-major fp16 [V, 512] on the device."""
    w = torch.as_tensor(zeroshot_weights)
    if not w.is_cuda:
        w = w.cuda()
    return ops.transpose_f16(w.to(torch.float16).contiguous())


def _as_feats(clip_feats):
    f = torch.as_tensor(clip_feats)
    if not f.is_cuda:
        f = f.cuda()
    return f.to(torch.float16).contiguous()


def get_clip_preds_fast(clip_feats, targets, cidx_to_cname, nouns, zeroshot_weights):
    """main_ptsup.py:78-99: argmax_name 100 * f @ W for every row (the targets only feed a tensor the reference never uses,
    but an unknown class name still raises like its nouns.index)."""
    first = _first_index(nouns)
    for t in np.unique(np.asarray(targets)):
        if cidx_to_cname[t] not in first:
            raise ValueError("%r is not in list" % (cidx_to_cname[t],))
    idx, _ = ops.sim_argmax(_as_feats(clip_feats), _as_wt(zeroshot_weights))
    return idx


def evaluate_semantic_acc_ub_lb(clip_feats, targets, cidx_to_cname, nouns, zeroshot_weights, return_top5=False):
    """main_ptsup.py:102-129: top-1 accuracy (%) of the zero-shot classifier `zeroshot_weights` ([512, len(nouns)]) against
    the vocabulary index of each row's class name - the lower bound with the full vocabulary, the upper bound with the
    ground-truth names only (call sites :550-561)."""
    first = _first_index(nouns)
    t = np.asarray(targets)
    tgt = torch.tensor([first[cidx_to_cname[x]] if cidx_to_cname[x] in first else nouns.index(cidx_to_cname[x]) for x in t],
                       dtype=torch.int64).cuda()
    wt = _as_wt(zeroshot_weights)
    k = min(5, wt.shape[0])
    idx, _ = ops.sim_topk(_as_feats(clip_feats), wt, k, "raw", 100.0)
    hit = idx == tgt.view(-1, 1)
    n = float(len(t))
    top1 = float(hit[:, 0].sum().item()) / n * 100
    if return_top5:
        return top1, float(hit.sum().item()) / n * 100
    return top1


# ----------------------------------------------------------------------------- soft sACC
def get_topk_name_indices(loader, cidx_to_cname, nouns, zeroshot_weights, model, with_target=True, verbose=True):
    """main_unsup.py:43-75 (and, with_target=False, get_topk_name_indices_wotarget :77-111): the loader form of the naming pass -
    per batch encode_image -> L2-normalise -> 100 f @ W -> top-5 names and raw logits, plus the zero-shot top-1 / top-5 accuracy
    against the vocabulary index of each image's class name.  The reference reads a global `model`; here it is an argument.  The
    N x V logits are never materialised (scd_sim_topk); ties among equal logits go to the lower name index.
    Returns (name_idx_top5 int64 [N, 5], name_logits_top5 float32 [N, 5]) on the device, like the reference's torch.cat."""
    first = _first_index(nouns)
    wt = _as_wt(zeroshot_weights)
    k = min(5, wt.shape[0])
    idxs, vals = [], []
    top1 = top5 = n = 0.0
    with torch.no_grad():
        for batch in loader:
            images, target = batch[0], batch[1]
            f = model.encode_image(images.cuda())
            f = ops.l2norm_rows(f)
            idx, val = ops.sim_topk(f, wt, k, "raw", 100.0)
            idxs.append(idx)
            vals.append(val)
            if with_target:
                tgt = torch.tensor([first[cidx_to_cname[int(t)]] if cidx_to_cname[int(t)] in first else nouns.index(cidx_to_cname[int(t)])
                                    for t in np.asarray(target)], dtype=torch.int64, device=idx.device)
                hit = idx == tgt.view(-1, 1)
                top1 += float(hit[:, 0].sum().item())
                top5 += float(hit.any(1).sum().item())
            n += images.shape[0]
    if with_target and verbose:
        print(f"Top-1 accuracy: {top1 / n * 100:.2f}")
        print(f"Top-5 accuracy: {top5 / n * 100:.2f}")
    return torch.cat(idxs, dim=0), torch.cat(vals, dim=0)


def get_topk_name_indices_wotarget(loader, cidx_to_cname, nouns, zeroshot_weights, model):
    """main_unsup.py:77-111."""
    return get_topk_name_indices(loader, cidx_to_cname, nouns, zeroshot_weights, model, with_target=False)


def calucate_dis_between_names(pred_name, target_name, wnid_to_synset, name_to_wnids):
    """main_unsup.py:170-188 (the reference's spelling): max Leacock-Chodorow similarity over the synsets of two names."""
    pred_wnids = name_to_wnids[pred_name]
    target_wnids = name_to_wnids[target_name]
    if 0 == len(pred_wnids):
        print(f"pred_name: {pred_name}, {pred_wnids}")
        return
    elif 0 == len(target_wnids):
        print(f"pred_name: {target_name}, {target_wnids}")
        return
    return max(wnid_to_synset[t].lch_similarity(wnid_to_synset[p]) for p in pred_wnids for t in target_wnids)


def evaluate_soft_semantic_acc(u_targets, cidx_to_cname, u_preds, cand_names, wnid_to_synset, name_to_wnids, return_score=False,
                               cache=None):
    """main_unsup.py:191-199 / main_ptsup.py:208-219.  The reference walks WordNet once per SAMPLE (and the mains call this
    three times per vote iteration); the score depends only on the (predicted name, target name) pair, so each distinct
    pair - at most n_cluster x n_classes of them - is scored once and looked up per row.  `cache` (a dict) may be shared
    between calls: the voting loop re-scores mostly the same pairs every iteration."""
    t = np.asarray(u_targets)
    p = np.asarray(u_preds.cpu() if hasattr(u_preds, "cpu") else u_preds).astype(np.int64).reshape(-1)
    cache = {} if cache is None else cache
    classes, t_inv = np.unique(t, return_inverse=True)
    pair = t_inv.astype(np.int64) * len(cand_names) + p
    upair, inv = np.unique(pair, return_inverse=True)
    score = np.empty(len(upair), dtype=object)
    for i, q in enumerate(upair.tolist()):
        key = (cand_names[q % len(cand_names)], cidx_to_cname[classes[q // len(cand_names)]])
        if key not in cache:
            cache[key] = calucate_dis_between_names(key[0], key[1], wnid_to_synset, name_to_wnids)
        score[i] = cache[key]
    matched_all = score[inv]
    matched_all = np.array(list(matched_all)) / max(matched_all)          # a missing synset (None) fails here, as in the reference
    semantic_acc_all = sum(matched_all) / float(len(matched_all))
    if not return_score:
        return semantic_acc_all
    return semantic_acc_all, matched_all


# ----------------------------------------------------------------------------- feature extraction (row a2)
def extract_feature(model, loader, args):
    """main_unsup.py:114-147: loader yields (images, label, uq_idx, mask_lab) batches; features are L2-normalised inside the
    encoder's last kernel and leave the device once, at the end (the reference copies and np.appends per batch)."""
    train_classes = set(int(c) for c in args.train_classes)
    feats, targets, mask_lab = [], [], []
    for images, label, _, mask_lab_ in loader:
        images = images.cuda()
        if args.feat_model == 'clip':
            f = model.visual.enc.encode_image(images, normalize=True)
        elif hasattr(model, "features"):
            f = model.features(images, normalize=True)
        else:
            f = ops.l2norm_rows(model(images).float())
        feats.append(f)
        targets.append(np.asarray(label.cpu().numpy() if hasattr(label, "cpu") else label, dtype=np.float64))
        mask_lab.append(np.asarray(mask_lab_.cpu().numpy() if hasattr(mask_lab_, "cpu") else mask_lab_).astype(bool))
    targets = np.concatenate(targets) if targets else np.array([])
    return dict(all_feats=torch.cat(feats).cpu().numpy(), mask_lab=np.concatenate(mask_lab).astype(bool),
                mask_cls=np.array([int(x) in train_classes for x in targets], dtype=bool), targets=targets)
