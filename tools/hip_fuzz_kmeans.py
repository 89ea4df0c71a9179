"""Fuzz the HIP k-means (KMeansEngine: lock-step C loops, MFMA filters, incremental M-step) against the float64 oracle on fp16-exact blob
data of several shapes - incl. the plain fits whose restarts empty clusters (tools/ref_fuzz_kmeans.py pins the oracle to the reference on the
same kind of cases).  python tools/hip_fuzz_kmeans.py [first_seed] [n_cases]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kmeans_oracle as ko, synth
from scd_amd.kmeans import KMeansEngine

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shapes = [(5000, 512, 40, 40), (4000, 768, 16, 16), (6000, 512, 60, 100), (3000, 256, 24, 24), (6000, 512, 100, 100), (2500, 128, 10, 14), (7000, 768, 50, 50)]
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for c in range(cases):
    n, d, blobs, k = shapes[c % len(shapes)]
    seed = s0 + c
    x, y, m = synth.blob_case(n, d, blobs, seed)
    x = x.astype(np.float16).astype(np.float32)
    u, l, lt = x[~m], x[m], y[m]
    for mode in ("fit_mix", "fit"):
        if mode == "fit_mix" and k < len(np.unique(lt)):
            continue
        t0 = time.time()
        km = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=4, random_state=seed)
        ok = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, n_init=4, random_state=seed)
        if mode == "fit_mix":
            km.fit_mix(T(u), T(l), T(lt)); ok.fit_mix(u, l, lt)
        else:
            km.fit(T(u)); ok.fit(u)
        cen = km.cluster_centers_.cpu().numpy()
        same = (np.array_equal(km.labels_.cpu().numpy(), ok.labels_) and np.array_equal(cen, ok.cluster_centers_, equal_nan=True)
                and float(km.inertia_) == float(ok.inertia_) and int(km.n_iter_) == int(ok.n_iter_))
        print("case %d n=%d d=%d blobs=%d k=%d seed=%d %-7s %s (inertia %.4f, NaN centre rows %d, n_iter %d, lock-step fits %s)  %.0fs" % (
            c, n, d, blobs, k, seed, mode, "bit-identical" if same else "MISMATCH", float(km.inertia_), int(np.isnan(cen).any(axis=1).sum()), int(km.n_iter_),
            km.stats.get("lockstep_fits", 0), time.time() - t0), flush=True)
        bad += not same
print("FUZZ", "MISMATCHES: %d" % bad if bad else "ok")
sys.exit(1 if bad else 0)
