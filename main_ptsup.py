#!/usr/bin/env python
"""Partially supervised semantic category discovery on MI355X - the entry point of /root/reference/main_ptsup.py with
the same flags (:227-244) and stage order, on libscd_hip.so.  Differences from main_unsup.py follow the reference:
the cluster cache name has no n_cluster (:385), TOP_K = 5 with raw logits (no softmax, :526-545), the zero-shot sACC lower /
upper bounds (:548-585), votes only over clusters without labelled samples, names of the labelled classes excluded and then
re-added (:588-676).  See main_unsup.py for the data conventions (--class_names, --images_pt, cache files)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import main_unsup as mu  # noqa: E402  (installs the aliases)
import clip  # noqa: E402
from gcd.project_utils.cluster_and_log_utils import split_cluster_acc_v2  # noqa: E402
from scd_amd import naming, ops, pipeline  # noqa: E402


def build_parser():
    p = mu.build_parser()
    p.set_defaults(dataset_name='imagenet_100', feat_model='clip', extract_feat=False, cluster='ConSSKM', n_cluster=100,
                   cluster_size_max=1000)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert torch.cuda.is_available(), "main_ptsup.py needs a HIP device"
    dev = torch.device("cuda")
    if args.synthetic:
        clip.allow_synthetic()
    model, _ = clip.load("ViT-B/16")
    model.cuda().eval()
    k = args.n_cluster
    wn = None
    if args.synthetic:
        images, y, base = pipeline.synthetic_images(args.synthetic_images, k, 0, dev)
        clip_all = pipeline.encode_images(model, images, 256)
        wt, nouns = pipeline.synthetic_vocab(model, base, args.synthetic_vocab, 0, dev)
        mask_lab = pipeline.labelled_split(y, k, args.prop_train_labels)
        # the reference relies on labelled-first ordering (data_utils.py:27-32): reorder rows accordingly
        order = np.concatenate([np.nonzero(mask_lab)[0], np.nonzero(~mask_lab)[0]])
        clip_all = clip_all[torch.as_tensor(order, device=dev)]
        targets = y.cpu().numpy()[order].astype(np.float64)
        mask_lab = np.arange(len(order)) < int(mask_lab.sum())
        mask_cls = targets < k // 2
        all_feats = clip_all.float().cpu().numpy()
        train_classes = list(range(k // 2))
        cidx_to_cname = {c: nouns[c] for c in range(k)}
        zw = None
    else:
        if not args.class_names:
            raise SystemExit("main_ptsup.py needs --class_names (JSON {original class name: class index}): the vote keeps the names "
                             "of the labelled classes fixed (main_ptsup.py:597-603)")
        feat_model = mu.load_feat_model(args, model) if args.extract_feat else None
        data = mu.load_or_extract(args, feat_model, args.feat_model, f'{args.feat_model}_{args.dataset_name}_all.pt')
        cdata = mu.load_or_extract(args, model, 'clip', f'clip_{args.dataset_name}_all.pt')
        all_feats, mask_lab, mask_cls, targets = data['all_feats'], data['mask_lab'], data['mask_cls'], data['targets']
        clip_all = torch.as_tensor(cdata['all_feats']).to(dev).half()
        nouns, zw, wt = mu.load_vocabulary(args, dev)
        with open(args.class_names) as fh:
            class_to_idx = {kk: int(v) for kk, v in json.load(fh).items()}
        cidx_to_cname = naming.resolve_class_names(args.dataset_name, args.corpus, class_to_idx, nouns, wt, model)
        cidx_to_cname = {c: cidx_to_cname[c] for c in sorted(cidx_to_cname)}
        # args.train_classes (get_class_splits, out of scope) = the classes of the labelled rows
        train_classes = sorted(int(c) for c in set(np.asarray(targets)[np.asarray(mask_cls, dtype=bool)].tolist()))
        wn = mu.wordnet_tables(args) if args.dataset_name != 'cub' else None
    mask_lab = np.asarray(mask_lab, dtype=bool)
    l_feats, u_feats = all_feats[mask_lab], all_feats[~mask_lab]
    l_targets, u_targets = targets[mask_lab], targets[~mask_lab]
    mask = np.asarray(mask_cls, dtype=bool)[~mask_lab]

    cdir = os.path.join(args.root_dir, 'cluster')
    cpath = os.path.join(cdir, f'{args.cluster}_{args.feat_model}_{args.dataset_name}.pt')            # :385 (no n_cluster)
    if args.run_cluster or args.synthetic:
        print(f'Fitting {args.cluster} ...')
        all_preds, preds = mu.run_clustering(args, u_feats, l_feats, l_targets)
        if all_preds is None:
            raise SystemExit("--cluster KM gives no labels for the labelled rows (all_preds), which the partially supervised vote "
                             "needs (main_ptsup.py:591-625): use SSKM or ConSSKM")
        cluster_result = dict(all_preds=all_preds, u_preds=preds, u_targets=u_targets, mask=mask)
        if args.save_cluster:
            os.makedirs(cdir, exist_ok=True)
            torch.save(cluster_result, cpath)
    else:
        cluster_result = torch.load(cpath, weights_only=False)
    all_preds, preds = cluster_result['all_preds'], cluster_result['u_preds']
    a, o, n = split_cluster_acc_v2(y_true=u_targets, y_pred=preds, mask=mask)
    print(f"{args.cluster} Accuracies: All {a} | Old {o} | New {n}")

    name_idx, _ = naming.full_vocab_topk(clip_all, None, 5, False, wt=wt)           # TOP_K = 5, raw logits (:526-545)
    m = torch.as_tensor(~mask_lab, device=dev)
    clip_u = clip_all[m]
    gt_names = list(cidx_to_cname.values())

    # zero-shot sACC bounds (:548-561): full vocabulary = lower bound, ground-truth names only = upper bound
    tm = torch.as_tensor(mask, device=dev)
    w_full = wt.t() if zw is None else zw
    print('=====sACC lower bound=====')
    lb = [naming.evaluate_semantic_acc_ub_lb(f, t, cidx_to_cname, nouns, w_full)
          for f, t in ((clip_u, u_targets), (clip_u[tm], u_targets[mask]), (clip_u[~tm], u_targets[~mask]))]
    print(f"sACC all {lb[0]},sACC old {lb[1]}, sACC new {lb[2]}")
    print('=====sACC upper bound=====')
    ub_names = [nm.lower().replace('-', '_') for nm in gt_names]
    first = {}
    for j, nm in enumerate(nouns):
        first.setdefault(nm, j)
    w_sel = ops.gather_rows_f16(wt, torch.tensor([first[nm] for nm in ub_names], dtype=torch.int64, device=dev)).t()
    ub = [naming.evaluate_semantic_acc_ub_lb(f, t, cidx_to_cname, ub_names, w_sel)
          for f, t in ((clip_u, u_targets), (clip_u[tm], u_targets[mask]), (clip_u[~tm], u_targets[~mask]))]
    print(f"sACC all {ub[0]},sACC old {ub[1]}, sACC new {ub[2]}")
    soft_cache = {}
    if wn is not None:                                                                # :564-585
        for tag, names, wmat in (("lower", nouns, w_full), ("upper", ub_names, w_sel)):
            cp = naming.get_clip_preds_fast(clip_u, u_targets, cidx_to_cname, names, wmat).cpu().numpy()
            if tag == "upper":
                ca, co_, cn = split_cluster_acc_v2(y_true=u_targets, y_pred=cp, mask=mask)
                print(f"clip ACC: All {ca} | Old {co_} | New {cn}")
            sv = [naming.evaluate_soft_semantic_acc(u_targets[s], cidx_to_cname, cp[s], names, wn[0], wn[2], cache=soft_cache)
                  for s in (slice(None), mask, ~mask)]
            print(f"=====Soft sACC {tag} bound===== all {sv[0]},sACC old {sv[1]}, sACC new {sv[2]}")

    lab_names = [gt_names[c] for c in train_classes]

    def report(it, cand, u_preds):
        a, o, n = split_cluster_acc_v2(y_true=u_targets, y_pred=u_preds, mask=mask)
        print(f"iter {it}: Accuracies: All {a} | Old {o} | New {n}")
        for tag, sel, acc in (("All", slice(None), a), ("old", mask, o), ("new", ~mask, n)):
            s_avg, s_all = naming.evaluate_semantic_acc(u_targets[sel], cidx_to_cname, u_preds[sel], cand)
            print(f"ACC/sACC_avg/sACC_all: {tag} {round(acc * 100, 2)}/{round(s_avg * 100, 2)}/{round(s_all * 100, 2)} ")
        if wn is not None:
            for tag, sel, acc in (("All", slice(None), a), ("old", mask, o), ("new", ~mask, n)):
                soft = naming.evaluate_soft_semantic_acc(u_targets[sel], cidx_to_cname, u_preds[sel], cand, wn[0], wn[2], cache=soft_cache)
                print(f"ACC/Soft sACC: {tag} {round(acc * 100, 2)}/{round(soft * 100, 2)}")

    cand, u_preds, trace = naming.vote_loop_ptsup(name_idx[m], all_preds, mask_lab, clip_u, wt, nouns, lab_names, k,
                                                  args.topk, args.num_common_vote, args.num_common_linear, on_iter=report)
    print(f"voting converged after {len(trace)} iterations")
    inter, union = set(cand) & set(gt_names), set(cand) | set(gt_names)             # :708-712
    print(f'IoU: {len(inter) * 1.0 / len(union)}')
    return cand, u_preds


if __name__ == "__main__":
    main()
