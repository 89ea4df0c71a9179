#!/usr/bin/env python
"""Partially supervised semantic category discovery on MI355X - the entry point of /root/reference/main_ptsup.py with
the same flags (:227-244) and stage order, on libscd_hip.so.  Differences from main_unsup.py follow the reference:
TOP_K = 5 with raw logits (no softmax, :526-545), votes only over clusters without labelled samples, names of the
labelled classes excluded and then re-added (:588-676).  See main_unsup.py for the data conventions."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import main_unsup as mu  # noqa: E402  (installs the aliases)
import clip  # noqa: E402
from gcd.project_utils.cluster_and_log_utils import split_cluster_acc_v2  # noqa: E402
from local_utils.clip_lang_util import get_nouns  # noqa: E402
from scd_amd import naming, ops, pipeline  # noqa: E402


def main():
    p = mu.build_parser()
    p.set_defaults(cluster='ConSSKM', cluster_size_max=1000)
    args = p.parse_args()
    assert torch.cuda.is_available(), "main_ptsup.py needs a HIP device"
    dev = torch.device("cuda")
    model, _ = clip.load("ViT-B/16")
    model.cuda().eval()
    k = args.n_cluster
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
        gt_names = [nouns[c] for c in range(k)]
    else:
        fdir = os.path.join(args.root_dir, 'extracted_features')
        data = torch.load(os.path.join(fdir, f'{args.feat_model}_{args.dataset_name}_all.pt'), weights_only=False)
        cdata = torch.load(os.path.join(fdir, f'clip_{args.dataset_name}_all.pt'), weights_only=False)
        all_feats, mask_lab, mask_cls, targets = data['all_feats'], data['mask_lab'], data['mask_cls'], data['targets']
        clip_all = torch.as_tensor(cdata['all_feats']).to(dev).half()
        nouns = [n.lower().replace('-', '_') for n in get_nouns(corpus=args.corpus)]
        zname = {'wordnet': 'nouns', 'wikibird': 'wikibird', 'wikidog': 'wikidog'}[args.corpus]
        zw = torch.load(os.path.join(args.root_dir, 'zeroshot_weights', f'zeroshot_weights_all_{zname}_vit_b_16.pt'))
        wt = ops.transpose_f16(zw.to(dev).half())
        raise SystemExit("real-data mode needs the dataset's class-name table (gcd/data, out of scope); use --synthetic")
    mask_lab = np.asarray(mask_lab, dtype=bool)
    l_feats, u_feats = all_feats[mask_lab], all_feats[~mask_lab]
    l_targets, u_targets = targets[mask_lab], targets[~mask_lab]
    mask = np.asarray(mask_cls, dtype=bool)[~mask_lab]
    all_preds, preds = mu.run_clustering(args, u_feats, l_feats, l_targets)
    a, o, n = split_cluster_acc_v2(y_true=u_targets, y_pred=preds, mask=mask)
    print(f"{args.cluster} Accuracies: All {a} | Old {o} | New {n}")
    name_idx, _ = naming.full_vocab_topk(clip_all, None, 5, False, wt=wt)           # TOP_K = 5, raw logits (:526)
    m = torch.as_tensor(~mask_lab, device=dev)
    lab_names = [gt_names[c] for c in train_classes]

    def report(it, cand, u_preds):
        sacc_avg, sacc = naming.evaluate_semantic_acc(u_targets, gt_names, u_preds, cand)
        print(f"iter {it}: sACC_avg {round(sacc_avg * 100, 2)} | sACC_all {round(sacc * 100, 2)} with {len(cand)} candidate names")

    cand, u_preds, trace = naming.vote_loop_ptsup(name_idx[m], all_preds, mask_lab, clip_all[m], wt, nouns, lab_names, k,
                                                  args.topk, args.num_common_vote, args.num_common_linear, on_iter=report)
    print(f"voting converged after {len(trace)} iterations")
    return cand, u_preds


if __name__ == "__main__":
    main()
