"""Stress run for the streaming E-step (not collected by pytest): repeats scd_kmeans_estep many times on full-size inputs and
checks that every repetition returns the labels of the first one, and that the first one matches the float64 oracle on a
sample of rows.  python tests/stress_estep.py [repeats]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
from oracle import kmeans_oracle as ko, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for (n, d, k, mode) in [(95000, 768, 100, "cent"), (95000, 768, 100, "data"), (126976, 512, 100, "data"), (60000, 768, 1000, "data"),
                        (40001, 300, 37, "data")]:
    x, y, cent = synth.clustered_features(n, d, min(k, 100), seed=3, center_seed=4, noise=0.8)
    X = torch.from_numpy(x).cuda()
    if mode == "cent" and k == cent.shape[0]:
        C = torch.from_numpy(cent).cuda()
    else:
        C = X[torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))[:k]].clone()
    data = ops.KMeansData(X)
    ref, refined = data.estep(C, return_refined=True)
    rows = np.random.RandomState(0).choice(n, 1024, replace=False)
    olab, _, _ = ko.estep(x[rows], C.cpu().numpy())
    ok0 = np.array_equal(ref.cpu().numpy()[rows], olab)
    mism = 0
    for _ in range(reps):
        lab = data.estep(C)
        if not torch.equal(lab, ref):
            mism += 1
    print("n=%d d=%d k=%d %s: oracle sample %s, refined %d, %d/%d repetitions differ" % (n, d, k, mode, "ok" if ok0 else "MISMATCH", int(refined), mism, reps))
    bad += (not ok0) + mism
print("STRESS", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
