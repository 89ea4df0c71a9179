"""Do the synthetic images separate in the GCD / DINO tower's feature space (bench.py --config c3)?  Purity of a plain SSKM fit per noise level."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import pipeline, ops
from scd_amd.clip import weights as W
from scd_amd.clip.model import DinoViT
from scd_amd.kmeans import KMeansEngine
dev = torch.device("cuda")
fm = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()
n, k = 12000, 120
for noise in (0.35, 0.2, 0.1):
    images, y, base = pipeline.synthetic_images(n, k, 0, dev, noise=noise)
    f = torch.cat([fm._enc.encode_image(images[s:s + 2000], normalize=True) for s in range(0, n, 2000)]).float()
    cls = torch.stack([f[y == c].mean(0) for c in range(k)])
    within = float((f - cls[y]).norm(dim=1).mean()); between = float(torch.pdist(cls).mean())
    mask = pipeline.labelled_split(y, k)
    m = torch.as_tensor(mask, device=dev)
    km = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=0)
    km.fit_mix(f[~m], f[m], y[m])
    lab = km.labels_[int(m.sum()):].cpu().numpy(); yu = y[~m].cpu().numpy()
    pur = sum(np.bincount(yu[lab == c]).max() for c in np.unique(lab)) / len(yu)
    print("noise %.2f: within-class spread %.4f, between-class distance %.4f, SSKM purity %.3f" % (noise, within, between, pur), flush=True)
