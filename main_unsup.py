#!/usr/bin/env python
"""Unsupervised semantic category discovery on MI355X - the entry point of /root/reference/main_unsup.py with the
same flags (:207-224), stage order (:298-641) and cache files, running on libscd_hip.so.

Data.  The reference reads image folders through torchvision dataset classes (gcd/data: out of scope).
Here every stage starts from the reference's own cache files under --root_dir, byte-compatible with the ones it writes:
  extracted_features/{feat_model}_{dataset}_all.pt   dict all_feats, mask_lab, mask_cls, targets      (:141-146,294-301)
  extracted_features/clip_{dataset}_all.pt           same dict, CLIP features                          (:306-311)
  cluster/{cluster}_{feat_model}_{dataset}_{n_cluster}.pt   dict all_preds, u_preds, u_targets, mask  (:366-374)
  zeroshot_weights/zeroshot_weights_all_{nouns|wikibird|wikidog}_vit_b_16.pt   tensor [512, V]        (:389-394)
The two things the reference takes from its dataset objects are passed as files instead:
  --class_names  JSON {original class name: class index}  (`datasets['test'].class_to_idx` / the sorted breed / wnid tables)
  --images_pt    torch file dict(images [N,3,224,224] preprocessed, targets [N], mask_lab [N]) for --extract_feat true
With --synthetic everything is generated from seeds (no dataset / checkpoint needed).
"""
import argparse
import json
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
from scd_amd.cluster import KMeans  # noqa: E402   (sklearn.cluster.KMeans surface on the HIP kernels)


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
    # additions (see the module docstring)
    p.add_argument('--synthetic', type=str2bool, default=False, help='seeded synthetic images / vocabulary')
    p.add_argument('--synthetic_images', type=int, default=8192)
    p.add_argument('--synthetic_vocab', type=int, default=21000)
    p.add_argument('--class_names', type=str, default='', help='JSON {original class name: class index} of the data set')
    p.add_argument('--images_pt', type=str, default='', help='preprocessed images for --extract_feat true')
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
    elif args.cluster == 'KM':
        # :362 `KMeans(n_clusters=args.n_cluster, random_state=0).fit(u_feats).labels_` - on the device, no host sklearn
        return None, KMeans(n_clusters=args.n_cluster, random_state=0).fit(np.asarray(u_feats, dtype=np.float32)).labels_
    else:
        raise NotImplementedError(args.cluster)
    u, l, lt = (torch.as_tensor(x).to(dev) for x in (u_feats, l_feats, l_targets))
    km.fit_mix(u.float(), l.float(), lt)
    all_preds = km.labels_.cpu().numpy()
    return all_preds, all_preds[len(l_targets):]


def load_or_extract(args, model, feat_model_name, out_name):
    """:294-313: extract with the HIP towers and save, or load the cache."""
    fdir = os.path.join(args.root_dir, 'extracted_features')
    path = os.path.join(fdir, out_name)
    if not args.extract_feat:
        return torch.load(path, weights_only=False)
    if not args.images_pt:
        raise SystemExit("--extract_feat true needs --images_pt (the reference's dataset classes are out of scope)")
    blob = torch.load(args.images_pt, weights_only=False)
    images, targets, mask_lab = blob['images'], np.asarray(blob['targets']), np.asarray(blob['mask_lab'])
    args_feat = argparse.Namespace(feat_model=feat_model_name,
                                   train_classes=blob.get('train_classes', sorted(set(targets[mask_lab.astype(bool)].tolist()))))

    def loader():
        for s in range(0, len(images), 256):          # (images, label, uq_idx, mask_lab) like MergedDataset (data_utils.py:12-37)
            yield images[s:s + 256], targets[s:s + 256], None, mask_lab[s:s + 256]
    data = naming.extract_feature(model, loader(), args_feat)
    os.makedirs(fdir, exist_ok=True)
    torch.save(data, path)
    return data


def load_feat_model(args, clip_model):
    """:240-264.  dino_vit / gcd weights come from $SCD_ROOT (no torch.hub offline)."""
    if args.feat_model == 'clip':
        return clip_model
    from scd_amd.clip import DinoViT
    path = os.path.join(os.environ.get("SCD_ROOT", args.root_dir), 'dino' if args.feat_model == 'dino_vit' else 'gcd',
                        f'{args.feat_model}_{args.dataset_name}.pt' if args.feat_model == 'gcd' else 'dino_vitbase16_pretrain.pth')
    if not os.path.exists(path):
        raise FileNotFoundError(f"{args.feat_model} weights not found at {path}")
    return DinoViT(torch.load(path, map_location='cpu')).cuda()


def load_vocabulary(args, dev):
    """:381-394: nouns + the [512, V] text classifier."""
    nouns = [n.lower().replace('-', '_') for n in get_nouns(corpus=args.corpus)]
    zname = {'wordnet': 'nouns', 'wikibird': 'wikibird', 'wikidog': 'wikidog'}[args.corpus]
    if args.corpus != 'wordnet':
        nouns = [n.lower().replace("'s", "").replace(' ', '_') for n in nouns]
    zw = torch.load(os.path.join(args.root_dir, 'zeroshot_weights', f'zeroshot_weights_all_{zname}_vit_b_16.pt'), weights_only=False)
    return nouns, zw, ops.transpose_f16(torch.as_tensor(zw).to(dev).half())


def wordnet_tables(args):
    """get_wordnet_dict() (:386) when nltk + its corpus are installed; soft sACC is skipped otherwise."""
    if args.corpus != 'wordnet':
        return None
    try:
        from local_utils.clip_lang_util import get_wordnet_dict
        return get_wordnet_dict()
    except Exception as e:          # nltk is an optional host-side dependency
        print(f"(soft sACC disabled: {type(e).__name__}: {e})")
        return None


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert torch.cuda.is_available(), "main_unsup.py needs a HIP device"
    dev = torch.device("cuda")
    if args.synthetic:
        clip.allow_synthetic()
    model, _ = clip.load("ViT-B/16")
    model.cuda().eval()
    wn = None

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
        feat_model = load_feat_model(args, model) if args.extract_feat else None
        data = load_or_extract(args, feat_model, args.feat_model, f'{args.feat_model}_{args.dataset_name}_all.pt')
        cdata = load_or_extract(args, model, 'clip', f'clip_{args.dataset_name}_all.pt')
        all_feats, mask_lab, mask_cls, targets = data['all_feats'], data['mask_lab'], data['mask_cls'], data['targets']
        clip_all = torch.as_tensor(cdata['all_feats']).to(dev).half()
        nouns, zw, wt = load_vocabulary(args, dev)
        cidx_to_cname = None
        if args.class_names:
            with open(args.class_names) as fh:
                class_to_idx = {k: int(v) for k, v in json.load(fh).items()}
            # :398-502: class names that the vocabulary lacks are matched to their closest names by the text tower (row a7)
            cidx_to_cname = naming.resolve_class_names(args.dataset_name, args.corpus, class_to_idx, nouns, wt, model)
        else:
            print("(no --class_names: the semantic accuracies sACC / soft sACC and the name IoU of main_unsup.py:616-647 need the "
                  "data set's class names and are not reported; cluster accuracies and the vote loop run as usual)")
        wn = wordnet_tables(args) if args.dataset_name != 'cub' else None
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
    soft_cache = {}

    def report(it, cand, u_preds):
        a, o, n = split_cluster_acc_v2(y_true=u_targets, y_pred=u_preds, mask=mask)
        print(f"iter {it}: Accuracies: All {a} | Old {o} | New {n}")
        if cidx_to_cname is None:
            return
        for tag, sel, acc in (("All", slice(None), a), ("old", mask, o), ("new", ~mask, n)):
            s_avg, s_all = naming.evaluate_semantic_acc(u_targets[sel], cidx_to_cname, u_preds[sel], cand)
            print(f"ACC/sACC_avg/sACC_all: {tag} {round(acc * 100, 2)}/{round(s_avg * 100, 2)}/{round(s_all * 100, 2)} ")
        if wn is not None:
            for tag, sel, acc in (("All", slice(None), a), ("old", mask, o), ("new", ~mask, n)):
                soft = naming.evaluate_soft_semantic_acc(u_targets[sel], cidx_to_cname, u_preds[sel], cand, wn[0], wn[2], cache=soft_cache)
                print(f"ACC/Soft sACC: {tag} {round(acc * 100, 2)}/{round(soft * 100, 2)}")

    cand, u_preds, trace = naming.vote_loop_unsup(name_idx[m], preds, clip_all[m], wt, nouns, args.n_cluster,
                                                  args.num_common_vote, args.num_common_linear, on_iter=report)
    print(f"voting converged after {len(trace)} iterations; {len(set(cand))} names")
    if cidx_to_cname is not None:       # :643-647 IoU of predicted names and GT names
        gt_names = list(cidx_to_cname.values())
        inter, union = set(cand) & set(gt_names), set(cand) | set(gt_names)
        print(f'IoU: {len(inter) * 1.0 / len(union)}')
    return cand, u_preds


if __name__ == "__main__":
    main()
