"""Fuzz the k-means oracle against the REFERENCE itself (only where /root/reference exists: the build container): the reference's
K_Means.fit_mix / fit (gcd/methods/clustering/faster_mix_k_means_pytorch.py) on fp16-exact blob data of several shapes against
oracle/kmeans_oracle.py - labels, inertia, n_iter, NaN centre rows.  PYTHONHASHSEED=0 python tools/ref_fuzz_kmeans.py [first_seed] [n_cases]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gen_golden as gg, kmeans_oracle as ko, synth
gg.install_stubs(gg.NxMinCostFlow)
import methods.clustering.faster_mix_k_means_pytorch as sskm

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shapes = [(5000, 512, 40, 40), (4000, 768, 16, 16), (8000, 512, 60, 100), (3000, 256, 24, 24), (6000, 512, 100, 100), (2500, 128, 10, 14)]
bad = 0
for c in range(cases):
    n, d, blobs, k = shapes[c % len(shapes)]
    seed = s0 + c
    x, y, m = synth.blob_case(n, d, blobs, seed)
    x = x.astype(np.float16).astype(np.float32)
    u, l, lt = x[~m], x[m], y[m]
    t0 = time.time()
    for mode in ("fit_mix", "fit"):
        km = sskm.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=3, random_state=seed, n_jobs=None, pairwise_batch_size=1024)
        ok = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=seed, pairwise_batch_size=1024)
        if mode == "fit_mix":
            if k < len(np.unique(lt)):
                continue
            km.fit_mix(torch.from_numpy(u), torch.from_numpy(l), torch.from_numpy(lt)); ok.fit_mix(u, l, lt)
        else:
            km.fit(torch.from_numpy(u)); ok.fit(u)
        same = np.array_equal(km.labels_.numpy(), ok.labels_)
        cen = np.allclose(km.cluster_centers_.numpy(), ok.cluster_centers_, rtol=1e-5, atol=1e-6, equal_nan=True)
        ine = abs(float(km.inertia_) - float(ok.inertia_)) <= 1e-5 * abs(float(km.inertia_))
        nit = int(km.n_iter_) == int(ok.n_iter_)
        print("case %d n=%d d=%d blobs=%d k=%d seed=%d %-7s labels %s centres %s inertia %s n_iter %s (ref %.4f, nan rows %d, n_iter %d)  %.0fs" % (
            c, n, d, blobs, k, seed, mode, same, cen, ine, nit, float(km.inertia_), int(np.isnan(km.cluster_centers_.numpy()).any(axis=1).sum()), int(km.n_iter_), time.time() - t0), flush=True)
        bad += not (same and cen and ine and nit)
print("FUZZ", "MISMATCHES: %d" % bad if bad else "ok")
