#!/usr/bin/env python
"""Unsupervised semantic category discovery on MI355X - the entry point of /root/reference/main_unsup.py with the
same flags (:207-224), stage order (:298-641) and cache files, running on libscd_hip.so.

Data: the reference reads image folders through torchvision loaders (out of scope).  Here the
stages start from the reference's own cache files under --root_dir
  extracted_features/{feat_model}_{dataset}_all.pt   keys all_feats, mask_lab, mask_cls, targets   (:141-146,294-301)
  extracted_features/clip_{dataset}_all.pt                                                            (:306-311)
  cluster/{cluster}_{feat_model}_{dataset}_{n_cluster}.pt   keys all_preds,u_preds,u_targets,mask    (:366-374)
  zeroshot_weights/zeroshot_weights_all_{nouns|wikibird|wikidog}_vit_b_16.pt  [512,V]                 (:389-394)
or, with --synthetic, from seeded synthetic images encoded on the fly (no dataset / checkpoint needed).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import scd_amd  # noqa: E402
scd_amd.install()

import clip  # noqa: E402
from gcd.project_utils.cluster_and_log_utils import split_cluster_acc_v2  # noqa: E402
from local_utils.util import str2bool  # noqa: E402
from local_utils.sskm_constrained import K_Means as ConSemiSupKMeans  # noqa: E402
from gcd.methods.clustering.faster_mix_k_means_pytorch import K_Means as SemiSupKMeans  # noqa: E402
from local_utils.clip_lang_util import get_nouns  # noqa: E402
from scd_amd import naming, ops, pipeline  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description='cluster', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--batch_size', default=32, type=int)
    p.add_argument('--num_workers', default=2, type=int)
    p.add_argument('--root_dir', type=str, default='/Your_data_dir')
    p.add_argument('--dataset_name', type=str, default='imagenet_1000')
    p.add_argument('--feat_model', type=str, default='dino_vit')
    p.add_argument('--prop_train_labels', type=float, default=0.5)
    p.add_argument('--transform', type=str, default='imagenet')
    p.add_argument('--extract_feat', type=str2bool, default=False)
    p.add_argument('--run_cluster', type=str2bool, default=False)
    p.add_argument('--cluster', type=str, default='KM', help='options: KM, SSKM, ConSSKM')
    p.add_argument('--save_cluster', type=str2bool, default=False)
    p.add_argument('--n_cluster', type=int, default=1000)
    p.add_argument('--cluster_size_min', type=int, default=50)
    p.add_argument('--cluster_size_max', type=int, default=1200)
    p.add_argument('--corpus', type=str, default='wordnet', help='options: wordnet, wikibird, wikidog')
    p.add_argument('--topk', type=int, default=5)
    p.add_argument('--num_common_vote', type=int, default=20)
    p.add_argument('--num_common_linear', type=int, default=4)
    # additions
    p.add_argument('--synthetic', type=str2bool, default=False, help='seeded synthetic images / vocabulary')
    p.add_argument('--synthetic_images', type=int, default=8192)
    p.add_argument('--synthetic_vocab', type=int, default=21000)
    p.add_argument('--class_names', type=str, default='', help='JSON file {class index: class name} of the data set (the '
                   'reference takes it from its dataset objects, which are out of scope here): enables sACC on cached features')
    return p


def run_clustering(args, u_feats, l_feats, l_targets):
    """main_unsup.py:334-364.  Returns (all_preds or None, u_preds numpy)."""
    dev = torch.device("cuda")
    if args.cluster == 'ConSSKM':
        km = ConSemiSupKMeans(k=args.n_cluster, tolerance=1e-4, max_iterations=10, init='k-means++', size_min=args.cluster_size_min,
                              size_max=args.cluster_size_max, n_init=10, random_state=None, n_jobs=None, pairwise_batch_size=1024)
    elif args.cluster == 'SSKM':
        km = SemiSupKMeans(k=args.n_cluster, tolerance=1e-4, max_iterations=10, init='k-means++', n_init=10, random_state=None,
                           n_jobs=None, pairwise_batch_size=1024, mode=None)
    else:
        from sklearn.cluster import KMeans                      # --cluster KM stays sklearn on the host (:362)
        return None, KMeans(n_clusters=args.n_cluster, random_state=0).fit(np.asarray(u_feats, dtype=np.float32)).labels_
    u, l, lt = (torch.as_tensor(x).to(dev) for x in (u_feats, l_feats, l_targets))
    km.fit_mix(u.float(), l.float(), lt)
    all_preds = km.labels_.cpu().numpy()
    return all_preds, all_preds[len(l_targets):]


def main():
    args = build_parser().parse_args()
    assert torch.cuda.is_available(), "main_unsup.py needs a HIP device"
    dev = torch.device("cuda")
    model, _ = clip.load("ViT-B/16")
    model.cuda().eval()

    if args.synthetic:
        k = args.n_cluster
        images, y, base = pipeline.synthetic_images(args.synthetic_images, k, 0, dev)
        clip_all = pipeline.encode_images(model, images, 256)
        wt, nouns = pipeline.synthetic_vocab(model, base, args.synthetic_vocab, 0, dev)
        mask_lab = pipeline.labelled_split(y, k, args.prop_train_labels)
        targets = y.cpu().numpy().astype(np.float64)
        mask_cls = targets < k // 2
        all_feats = clip_all.float().cpu().numpy()
        cidx_to_cname = {c: nouns[c] for c in range(k)}
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
        cidx_to_cname = None
        if args.class_names:
            import json
            with open(args.class_names) as fh:
                cidx_to_cname = {int(k): v for k, v in json.load(fh).items()}
    mask_lab = np.asarray(mask_lab, dtype=bool)
    l_feats, u_feats = all_feats[mask_lab], all_feats[~mask_lab]
    l_targets, u_targets = targets[mask_lab], targets[~mask_lab]
    mask = np.asarray(mask_cls, dtype=bool)[~mask_lab]

    cdir = os.path.join(args.root_dir, 'cluster')
    cpath = os.path.join(cdir, f'{args.cluster}_{args.feat_model}_{args.dataset_name}_{args.n_cluster}.pt')
    if args.run_cluster or args.synthetic:
        print(f'Fitting {args.cluster} ...')
        all_preds, preds = run_clustering(args, u_feats, l_feats, l_targets)
        cluster_result = dict(all_preds=all_preds, u_preds=preds, u_targets=u_targets, mask=mask)
        if args.save_cluster:
            os.makedirs(cdir, exist_ok=True)
            torch.save(cluster_result, cpath)
    else:
        cluster_result = torch.load(cpath, weights_only=False)
    preds = cluster_result['u_preds']
    all_acc, old_acc, new_acc = split_cluster_acc_v2(y_true=u_targets, y_pred=preds, mask=mask)
    print(f"{args.cluster} Accuracies: All {all_acc} | Old {old_acc} | New {new_acc}")

    # CLIP voting (main_unsup.py:504-641)
    name_idx, _ = naming.full_vocab_topk(clip_all, None, args.topk, True, wt=wt)
    m = torch.as_tensor(~mask_lab, device=dev)

    def report(it, cand, u_preds):
        a, o, n = split_cluster_acc_v2(y_true=u_targets, y_pred=u_preds, mask=mask)
        line = f"iter {it}: ACC All {round(a * 100, 2)} | Old {round(o * 100, 2)} | New {round(n * 100, 2)}"
        if cidx_to_cname is not None:
            sacc_avg, sacc_all = naming.evaluate_semantic_acc(u_targets, cidx_to_cname, u_preds, cand)
            line += f" | sACC_avg {round(sacc_avg * 100, 2)} | sACC_all {round(sacc_all * 100, 2)}"
        print(line)

    cand, u_preds, trace = naming.vote_loop_unsup(name_idx[m], preds, clip_all[m], wt, nouns, args.n_cluster,
                                                  args.num_common_vote, args.num_common_linear, on_iter=report)
    print(f"voting converged after {len(trace)} iterations; {len(set(cand))} names")
    return cand, u_preds


if __name__ == "__main__":
    main()
