"""End-to-end hot path (encode -> L2-norm -> full-vocab sim+top-k -> semi-supervised K-Means -> vote loop), i.e. the
stage order of /root/reference/main_unsup.py:298-641 with the I/O and eval prints removed.  Used by bench.py,
__graft_entry__.smoke() and main_unsup.py --synthetic.  Everything numeric inside `run` is a libscd_hip.so call (row selections
included: scd_select_rows); torch allocates, uploads the row numbers and carries the collectives.

Multi-GPU: images (and their features) are sharded over ranks; W is replicated; K-Means exchanges one packed
all-reduce per Lloyd iteration; the vote either histograms each rank's own rows into a dense [clusters, V] table and all-reduces it
(counts: sum, first-seen positions: min) or - cheaper at every BASELINE size - gathers the rows' name lists once and their
cluster ids per iteration to rank 0's histogram; rank 0 solves the assignment and broadcasts the K candidate names, and every rank
re-classifies its own shard.
"""
import numpy as np
import torch

from . import naming, ops
from .gcd.methods.clustering.faster_mix_k_means_pytorch import K_Means as SemiSupKMeans


def encode_images(model, images, batch, out=None):
    """extract_feature (main_unsup.py:114-147) without the host round trips: encode + F.normalize, features stay in HBM."""
    n = images.shape[0]
    if out is None:
        out = torch.empty((n, model.visual.output_dim), dtype=torch.float16, device=images.device)
    enc = model.visual.enc
    for s in range(0, n, batch):
        enc.encode_image(images[s:s + batch], normalize=True, out=out[s:s + batch])       # straight into the feature matrix's rows
    return out


def run(model, images, mask_lab, l_targets, wt, nouns, n_cluster, topk=3, num_common_vote=10, num_common_linear=2,
        batch=3990, kmeans_iters=10, n_init=10, seed=0, group=None, timers=None, cluster="SSKM", build_vocab=None, text_feats=None):
    """One pass over `images` (this rank's shard).  Returns dict(feats, labels, cand_names, u_preds, name_idx).
    build_vocab: a callable returning the name-major classifier W^T - the open-vocabulary build of BASELINE configs[4] (text tower over
    the names, clip_lang_util.zeroshot_classifier[_sharded]) then runs INSIDE the pass ("text_tower" stage) instead of `wt` being
    handed in.  text_feats [n, 512] fp16: per-image closed-set text features for the textual-enhancement re-ranking - top-k and the vote
    loop's re-classification then use 100 * (f @ W + t @ W) / 2 = 100 * mean(f, t) @ W (the formula the reference keeps commented at
    main_unsup.py:518,523,604,609); the clustering still sees the image features."""
    def mark(name):
        if timers is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timers.append((name, ev))
    mark("start")
    if build_vocab is not None:
        wt = build_vocab()
        mark("text_tower")
    feats = encode_images(model, images, batch)
    mark("encode")
    if text_feats is not None:
        nfeats = ops.mean2_f16(feats, text_feats)          # the naming feature; `feats` stays the clustering feature
        name_idx, name_val = ops.sim_topk(nfeats, wt, topk, "softmax", 100.0)
    else:
        nfeats = feats
        name_idx, name_val = naming.full_vocab_topk(feats, None, topk, True, wt=wt)
    mark("sim_topk")
    # `all_feats[~mask_lab]`, `all_feats[mask_lab]`, `name_idx[~mask_lab]` (main_unsup.py:318-321,561): the row numbers come from the
    # host mask (two small uploads), the selections are one launch each - no torch kernel in the step
    mask_h = np.asarray(mask_lab.cpu() if torch.is_tensor(mask_lab) else mask_lab, dtype=bool)
    iu = torch.from_numpy(np.flatnonzero(~mask_h)).to(feats.device)
    il = torch.from_numpy(np.flatnonzero(mask_h)).to(feats.device)
    if nfeats is feats:
        fu, u_feats, nidx_u = ops.select_rows(feats, iu, name_idx)
    else:
        fu, _, nidx_u = ops.select_rows(nfeats, iu, name_idx, want32=False)
        _, u_feats, _ = ops.select_rows(feats, iu, None, want16=False)
    _, l_feats, _ = ops.select_rows(feats, il, None, want16=False)
    if cluster == "KM":
        # the shipped default of scripts/evaluate_unsupervised.sh: `KMeans(n_clusters, random_state=0).fit(u_feats).labels_` (:362)
        from .cluster import KMeans
        assert group is None, "--cluster KM is a single-process fit (as in the reference)"
        km = KMeans(n_clusters=n_cluster, random_state=0).fit(u_feats)        # the call site's fixed seed (main_unsup.py:362)
        mark("kmeans")
        u_preds = torch.from_numpy(km.labels_).to(feats.device)
        cand, preds, trace = naming.vote_loop_unsup(nidx_u, u_preds, fu, wt, nouns, n_cluster, num_common_vote, num_common_linear, max_iter=50)
        mark("vote")
        return dict(feats=feats, labels=km.labels_, cand_names=cand, u_preds=preds, name_idx=name_idx, vote_iters=len(trace), kmeans=km, wt=wt)
    km = SemiSupKMeans(k=n_cluster, tolerance=1e-4, max_iterations=kmeans_iters, init='k-means++', n_init=n_init,
                       random_state=seed, n_jobs=None, pairwise_batch_size=1024, mode=None, group=group)
    km.fit_mix(u_feats, l_feats, torch.as_tensor(l_targets, device=feats.device))
    mark("kmeans")
    u_preds = km.labels_[l_feats.shape[0]:]
    if group is None:
        cand, preds, trace = naming.vote_loop_unsup(nidx_u, u_preds, fu, wt, nouns, n_cluster, num_common_vote,
                                                    num_common_linear, max_iter=50)
    else:
        cand, preds, trace = vote_loop_unsup_sharded(nidx_u, u_preds, fu, wt, nouns, n_cluster, num_common_vote,
                                                     num_common_linear, group, max_iter=50)
    mark("vote")
    return dict(feats=feats, labels=km.labels_, cand_names=cand, u_preds=preds, name_idx=name_idx, vote_iters=len(trace),
                kmeans=km, wt=wt)


def run_cached(cluster_feats, clip_feats, mask_lab, wt, nouns, n_cluster, topk=3, num_common_vote=10, num_common_linear=2, timers=None,
               cluster="KM", l_targets=None, seed=0):
    """The pass of BASELINE configs[0] (CUB-200 unsupervised on cached features, main_unsup.py:298-364 with `extract_feature` replaced
    by the cache files it writes): no encoder at all - full-vocabulary top-k of the cached CLIP features, the clustering of the cached
    DINO features' unlabelled rows (`--cluster KM`, the shipped flag: `KMeans(n_clusters, random_state=0).fit(u_feats)`, :362; or SSKM)
    and the vote loop (:568-614).  cluster_feats float32 [n, 768], clip_feats fp16 [n, 512], both resident in HBM."""
    def mark(name):
        if timers is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timers.append((name, ev))
    mark("start")
    name_idx, _ = naming.full_vocab_topk(clip_feats, None, topk, True, wt=wt)
    mark("sim_topk")
    mask_h = np.asarray(mask_lab.cpu() if torch.is_tensor(mask_lab) else mask_lab, dtype=bool)
    iu = torch.from_numpy(np.flatnonzero(~mask_h)).to(clip_feats.device)
    fu, _, nidx_u = ops.select_rows(clip_feats, iu, name_idx, want32=False)
    u_feats = cluster_feats.index_select(0, iu)
    if cluster == "KM":
        from .cluster import KMeans
        km = KMeans(n_clusters=n_cluster, random_state=0).fit(u_feats)
        u_preds = torch.from_numpy(km.labels_).to(clip_feats.device)
    else:
        il = torch.from_numpy(np.flatnonzero(mask_h)).to(clip_feats.device)
        km = SemiSupKMeans(k=n_cluster, tolerance=1e-4, max_iterations=10, init='k-means++', n_init=10, random_state=seed, n_jobs=None,
                           pairwise_batch_size=1024, mode=None)
        km.fit_mix(u_feats, cluster_feats.index_select(0, il), torch.as_tensor(l_targets, device=clip_feats.device))
        u_preds = km.labels_[int(mask_h.sum()):]
    mark("kmeans")
    cand, preds, trace = naming.vote_loop_unsup(nidx_u, u_preds, fu, wt, nouns, n_cluster, num_common_vote, num_common_linear, max_iter=50)
    mark("vote")
    return dict(feats=clip_feats, labels=km.labels_, cand_names=cand, u_preds=preds, name_idx=name_idx, vote_iters=len(trace), kmeans=km, wt=wt,
                u_feats=u_feats)


def run_ptsup(model, feat_model, images, mask_lab, l_targets, wt, nouns, lab_names, n_cluster, topk=2, num_common_vote=5,
              num_common_linear=2, size_min=50, size_max=1000, batch=3990, kmeans_iters=10, n_init=10, seed=0, timers=None):
    """The partially supervised path (main_ptsup.py:272-705 with the I/O and the eval prints removed) on one GPU: clustering features
    from the GCD / DINO tower and naming features from CLIP for every image (extract_feature, main_ptsup.py:132-166), raw-logit top-5
    over the vocabulary (:526-545), size-constrained semi-supervised K-Means with the call-site values (:356-366: ten restarts x ten
    iterations, bounds --cluster_size_min / --cluster_size_max) and the partially supervised vote (:588-676) with the labelled classes'
    names kept.  Rows are brought into the reference's labelled-first order (data_utils.py:27-32) by one row selection per matrix."""
    from .local_utils.sskm_constrained import K_Means as ConSemiSupKMeans

    def mark(name):
        if timers is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timers.append((name, ev))
    mark("start")
    feats = encode_images(model, images, batch)
    n = images.shape[0]
    if feat_model._enc is None:
        feat_model.cuda()
    gfeats = torch.empty((n, feat_model._enc.out_dim), dtype=torch.float16, device=images.device)
    for s in range(0, n, batch):
        feat_model._enc.encode_image(images[s:s + batch], normalize=True, out=gfeats[s:s + batch])
    mark("encode")
    name_idx, _ = naming.full_vocab_topk(feats, None, 5, False, wt=wt)           # TOP_K = 5, no softmax (main_ptsup.py:526-545)
    mark("sim_topk")
    mask_h = np.asarray(mask_lab.cpu() if torch.is_tensor(mask_lab) else mask_lab, dtype=bool)
    iu = torch.from_numpy(np.flatnonzero(~mask_h)).to(feats.device)
    il = torch.from_numpy(np.flatnonzero(mask_h)).to(feats.device)
    fu, _, nidx_u = ops.select_rows(feats, iu, name_idx, want32=False)
    _, gu, _ = ops.select_rows(gfeats, iu, None, want16=False)
    _, gl, _ = ops.select_rows(gfeats, il, None, want16=False)
    km = ConSemiSupKMeans(k=n_cluster, tolerance=1e-4, max_iterations=kmeans_iters, init='k-means++', size_min=size_min, size_max=size_max,
                          n_init=n_init, random_state=seed, n_jobs=None, pairwise_batch_size=1024)
    km.fit_mix(gu, gl, torch.as_tensor(l_targets, device=feats.device))
    mark("kmeans")
    all_preds = km.labels_.cpu().numpy()
    lab_first = np.arange(n) < int(mask_h.sum())
    cand, preds, trace = naming.vote_loop_ptsup(nidx_u, all_preds, lab_first, fu, wt, nouns, lab_names, n_cluster, topk, num_common_vote,
                                                num_common_linear, max_iter=50)
    mark("vote")
    return dict(feats=feats, labels=all_preds, cand_names=cand, u_preds=preds, name_idx=name_idx, vote_iters=len(trace), kmeans=km)


def vote_loop_unsup_sharded(name_idx, u_preds, f_u, wt, nouns, n_cluster, ncv, ncl, group, max_iter=50, be=None, exchange="auto"):
    """main_unsup.py:568-614 over row shards, with the exchange of SURVEY.md 8e.  Per iteration every rank histograms ITS rows into
    a dense [clusters, V] table (counts, first-seen position in global row order), the tables are all-reduced (sum / min), and
    most_common(m) of every cluster is read off the reduced table - what Counter.most_common gives on the concatenated rows.  Rank 0
    solves the assignment (Munkres, host) and broadcasts the voted names, the assignment and the K candidate columns; every rank
    re-classifies only its own rows.  O(N / world) device work per rank and iteration; the collectives carry 12 bytes per
    (cluster, name) pair, whatever N is.  `be` = the op set (default scd_amd.ops; tests/test_dist_gloo.py passes an oracle-backed
    stand-in).

    exchange = "table" is the form above; "rows" exchanges the ROWS instead: the top-k name rows never change during the loop, so
    they are all-gathered once (8 k bytes per row), every iteration all-gathers only the rows' current cluster ids (8 bytes per
    row) and rank 0 histograms all rows itself - the same counts and first-seen positions by construction.  "auto" takes "rows"
    while an iteration's ids are fewer bytes than its tables (N_global * 8 <= clusters * V * 12: every BASELINE config - C2: 6 MB
    against 25 MB per iteration at 8 x 95k rows, C4 (K = 1000): 10 MB against 252 MB) and "table" beyond (N in the tens of millions)."""
    import copy
    import torch.distributed as dist
    from .local_utils.clip_lang_util import assign_name
    be = be or ops
    dev = name_idx.device
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lens = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([name_idx.shape[0]], dtype=torch.int64, device=dev), group=group)
    row_offset = int(sum(int(x) for x in lens[:rank]))
    v = wt.shape[0]
    first = {}
    for j, n in enumerate(nouns):
        first.setdefault(n, j)
    top_k = min(5, name_idx.shape[1])
    m = max(ncv, ncl)
    cur, prev, cand, trace = [0], [1], list(nouns), []
    u_preds = u_preds.to(torch.int64)
    n_slots = max(n_cluster, int(u_preds.max().item()) + 1 if u_preds.numel() else 1)
    ns = torch.tensor([n_slots], dtype=torch.int64, device=dev)
    dist.all_reduce(ns, op=dist.ReduceOp.MAX, group=group)
    n_slots = int(ns.item())
    n_rows = [int(x) for x in lens]
    n_glob = sum(n_rows)
    if exchange == "auto":
        exchange = "rows" if n_glob * 8 <= n_slots * v * 12 else "table"
    assert exchange in ("rows", "table")
    if exchange == "rows":
        # the rows' top-k names, gathered ONCE in rank order (padded to the longest shard; a rank may own no row)
        mx = max(max(n_rows), 1)
        pad = torch.zeros((mx, name_idx.shape[1]), dtype=torch.int64, device=dev)
        pad[: name_idx.shape[0]] = name_idx
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        name_idx_all = torch.cat([parts[r][: n_rows[r]] for r in range(world)]).contiguous()
        ppad = torch.zeros(mx, dtype=torch.int64, device=dev)
        pparts = [torch.empty_like(ppad) for _ in range(world)]
    while set(cur) != set(prev) and len(trace) < max_iter:
        if exchange == "rows":
            ppad[: u_preds.shape[0]] = u_preds
            dist.all_gather(pparts, ppad, group=group)
            preds_all = torch.cat([pparts[r][: n_rows[r]] for r in range(world)]).contiguous()
            # the clusters present anywhere (python-set order of the ids, main_unsup.py:573): every rank has all ids
            present = torch.bincount(preds_all, minlength=n_slots)
            clusters = list(set(torch.nonzero(present).reshape(-1).cpu().numpy().tolist()))
            if rank == 0:
                counts, firsts = be.vote_table(name_idx_all, top_k, preds_all, clusters, n_slots, 0, v)
        else:
            # the clusters present anywhere
            present = torch.bincount(u_preds, minlength=n_slots).to(torch.int64)
            dist.all_reduce(present, group=group)
            clusters = list(set(torch.nonzero(present).reshape(-1).cpu().numpy().tolist()))
            counts, firsts = be.vote_table(name_idx, top_k, u_preds, clusters, n_slots, row_offset, v)
            dist.all_reduce(counts, group=group)
            dist.all_reduce(firsts, op=dist.ReduceOp.MIN, group=group)
        # rank 0: most_common(m) per cluster off the (reduced) tables, names voted on, assignment (Munkres); broadcast
        # [n_voted | voted | ind (pairs)].  (Only rank 0 consumes the top-m lists, so only rank 0 extracts them.)
        nc = len(clusters)
        if rank == 0:
            keys, cnts = be.vote_table_topm(counts, firsts, m)
            keys_h, cnts_h = keys.cpu().numpy(), cnts.cpu().numpy()
            c2c = {c: naming.TopCounter(keys_h[i], cnts_h[i]) for i, c in enumerate(clusters)}
            voted = []
            for i in clusters:
                voted += [c[0] for c in c2c[i].most_common(ncv)]
            voted = list(set(voted))
            ind, w = assign_name(voted, c2c, num_common=ncl)
            pack = np.concatenate([[len(voted), len(ind)], np.asarray(voted, dtype=np.int64), np.asarray(ind, dtype=np.int64).reshape(-1)])
        else:
            pack = None
        hdr = torch.tensor([0 if pack is None else len(pack)], dtype=torch.int64, device=dev)
        dist.broadcast(hdr, 0, group=group)
        buf = torch.empty(int(hdr.item()), dtype=torch.int64, device=dev) if pack is None else torch.from_numpy(pack.astype(np.int64)).to(dev)
        dist.broadcast(buf, 0, group=group)
        arr = buf.cpu().numpy()
        nv, ni = int(arr[0]), int(arr[1])
        voted = arr[2:2 + nv].tolist()
        ind = arr[2 + nv:2 + nv + 2 * ni].reshape(ni, 2)
        prev = copy.deepcopy(cur)
        cur = [nouns[voted[x[1]]] for x in ind[:n_cluster]]
        cand = copy.deepcopy(cur)
        cols = torch.tensor([first[n] for n in cand], dtype=torch.int64, device=dev)
        u_preds, _ = be.sim_argmax(f_u, be.gather_rows_f16(wt, cols))
        u_preds = u_preds.to(torch.int64)
        trace.append(dict(voted=np.array(voted, dtype=np.int64), ind=ind, cand=cols.cpu().numpy(), u_preds=u_preds.cpu().numpy()))
    return cand, u_preds.cpu().numpy(), trace


# ----------------------------------------------------------------------------- synthetic workload
def synthetic_images(n, n_classes, seed, device, noise=0.35, chunk=4096, dtype=torch.float16):
    """'Preprocessed' 224x224 images with class structure: img = base[y] + noise*randn (N(0,1)-scaled, so CLIP's
    mean/std normalisation is already applied).  Returns (images [n,3,224,224] in HBM, y int64 [n])."""
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    gb = torch.Generator(device=device).manual_seed(4321)
    base = torch.randn(n_classes, 3, 224, 224, generator=gb, device=device, dtype=torch.float32)
    y = torch.randint(0, n_classes, (n,), generator=g, device=device)
    imgs = torch.empty((n, 3, 224, 224), dtype=dtype, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        z = torch.randn(e - s, 3, 224, 224, generator=g, device=device, dtype=torch.float32)
        imgs[s:e] = (base[y[s:e]] + noise * z).to(dtype)
    return imgs, y, base


def synthetic_vocab(model, base, v, seed, device, jitter=0.05):
    """W^T [v,512] fp16: row c < K is the (jittered) CLIP feature of class c's base image (the 'true' name), the rest
    are random unit vectors.  Returns (wt, nouns)."""
    protos = model.visual.enc.encode_image(base.to(torch.float16), normalize=True).float()
    g = torch.Generator(device=device).manual_seed(7 + seed)
    w = torch.randn(v, protos.shape[1], generator=g, device=device)
    k = protos.shape[0]
    w[:k] = protos + jitter * torch.randn(k, protos.shape[1], generator=g, device=device) / protos.shape[1] ** 0.5
    wt = ops.l2norm_rows(w.contiguous()).to(torch.float16).contiguous()
    return ops.freeze_vocab(wt), ["name_%05d" % i for i in range(v)]         # (not written again: its filter norm once, not per call)


def labelled_split(y, n_classes, prop=0.5, seed=5):
    """Classes < K/2 are 'old' and `prop` of their rows are labelled (get_datasets.py:144-145).  Rows are NOT
    re-ordered: a boolean mask plays the role of the reference's labelled-first ordering."""
    r = np.random.RandomState(seed)
    yn = y.cpu().numpy()
    return (yn < n_classes // 2) & (r.rand(len(yn)) < prop)
