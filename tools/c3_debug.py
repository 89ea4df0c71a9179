"""bench.py --config c3: how many distinct names does the first vote iteration see, for several synonym jitters (debugging aid)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
from scd_amd import pipeline, naming, ops
from scd_amd.clip import weights as W
from scd_amd.clip.model import DinoViT
from scd_amd.local_utils.sskm_constrained import K_Means
dev = torch.device("cuda")
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
n, k, v = 12000, 120, 21000
images, y, base = pipeline.synthetic_images(n, k, seed=0, device=dev)
wt0, nouns = pipeline.synthetic_vocab(model, base, v, 0, dev)
protos = wt0[:k].float()
pc = (protos @ protos.t())
print("prototype cosines: off-diagonal mean %.5f max %.5f" % (float((pc.sum() - pc.diag().sum()) / (k * k - k)), float((pc - 2 * torch.eye(k, device=dev)).max())))
feats = pipeline.encode_images(model, images, 3990)
fm = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()
g = torch.cat([fm._enc.encode_image(images[s:s + 3990], normalize=True) for s in range(0, n, 3990)])
mask_lab = pipeline.labelled_split(y, k, seed=5)
m = torch.as_tensor(mask_lab, device=dev)
km = K_Means(k=k, tolerance=1e-4, max_iterations=10, init='k-means++', size_min=50, size_max=1000, n_init=10, random_state=0, pairwise_batch_size=1024)
km.fit_mix(g[~m].float(), g[m].float(), y[m])
all_preds = km.labels_.cpu().numpy()
n_l = int(mask_lab.sum())
yu = y[~m].cpu().numpy()
up = all_preds[n_l:]
print("ConSSKM: purity %.3f, unlabelled clusters with a majority class shared with another cluster: %d" % (
    sum(np.bincount(yu[up == c]).max() for c in np.unique(up)) / len(yu),
    len([1 for c in range(60, 120)]) - len(set(int(np.bincount(yu[up == c]).argmax()) for c in range(60, 120) if (up == c).any()))))
name_row = np.sort(np.random.RandomState(123).choice(np.arange(k, v), size=k, replace=False))
free = np.setdiff1d(np.arange(k, v), name_row)
syn_rows = np.sort(np.random.RandomState(124).choice(free, size=4 * k, replace=False))
for jit in (0.05, 0.06, 0.08, 0.15):
    wt = wt0.clone()
    rows_t = torch.as_tensor(name_row, device=dev)
    proto, other = wt[:k].clone(), wt[rows_t].clone()
    wt[rows_t] = proto
    wt[:k] = other
    gs = torch.Generator(device=dev).manual_seed(77)
    syn = proto.float().repeat_interleave(4, dim=0)
    syn = syn + jit * torch.randn(syn.shape, generator=gs, device=dev) / syn.shape[1] ** 0.5
    wt[torch.as_tensor(syn_rows, device=dev)] = torch.nn.functional.normalize(syn, dim=-1).to(wt.dtype)
    idx, _ = ops.sim_topk(feats, wt, 5, "raw")
    iu = idx[~m][:, :2].cpu().numpy()
    known = set(name_row[: k // 2].tolist())
    voted = set()
    per = []
    for c in range(60, 120):
        rows = iu[up == c].reshape(-1)
        rows = [x for x in rows.tolist() if x not in known]
        from collections import Counter
        mc = [a for a, _ in Counter(rows).most_common(5)]
        per.append(len(mc))
        voted |= set(mc)
    top1 = float((idx[:, 0].cpu().numpy() == name_row[y.cpu().numpy()]).mean())
    print("synonym jitter %.2f: top-1 = true name %.3f; names per unlabelled cluster min %d mean %.1f; distinct voted names %d (need >= 60)" % (jit, top1, min(per), np.mean(per), len(voted)), flush=True)
