"""Stress: the restarts' Lloyd loops in lock-step over four streams (scd_kmeans_lloyd_run_multi) must give bit-identical labels, centres and
inertia on repeated fits of the same rows and seed - a missing stream dependency would show up as a diff (the exchange-buffer race of
round 5 did).  Also the merged multi-restart filter (SCD_ESTEP_MERGED=1).   python tests/stress_lockstep.py [repeats]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km
from oracle import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
# (the last case: no labelled rows and k = the number of blobs - every restart EMPTIES a cluster within its first iterations and ends there,
# as the reference's NaN arithmetic does; its winning centres carry NaN rows, hence the bit-wise comparison below)
for (n, d, k, tol) in [(95000, 512, 100, 1e-4), (60000, 768, 100, 5e-2), (30000, 512, 37, 1e-4), (30000, 768, 100, -1.0)]:
    x, y, _ = synth.clustered_features(n, d, k, seed=61 if tol < 0 else 3, center_seed=62 if tol < 0 else 4, noise=0.8)
    X = torch.from_numpy(x).cuda().half().float()
    yt = torch.from_numpy(y).cuda()
    mask = torch.from_numpy((y < k // 2) & (np.random.RandomState(5).rand(n) < 0.5)).cuda()
    for merged in ("0", "1"):
        os.environ["SCD_ESTEP_MERGED"] = merged
        ref = None
        mism = 0
        for r in range(reps):
            eng = km.KMeansEngine(k=k, tolerance=1e-4 if tol < 0 else tol, max_iterations=10, n_init=3 if tol < 0 else 10, random_state=2 if tol < 0 else 7)
            if tol < 0:              # (seed 61 / 62 below: the data of test_incremental_mstep_is_bit_identical's dying case)
                eng.fit(X)
                assert bool(torch.isnan(eng.cluster_centers_).any()) and eng.n_iter_ == 10, "the case no longer empties a cluster"
            else:
                eng.fit_mix(X[~mask], X[mask], yt[mask])
            cur = (eng.labels_.clone(), eng.cluster_centers_.clone().view(torch.int32), float(eng.inertia_))
            if ref is None:
                ref = cur
            elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and cur[2] == ref[2]):
                mism += 1
        print("n=%d d=%d k=%d tol=%g merged=%s: %d/%d repetitions differ (lock-step fits %s)" % (n, d, k, tol, merged, mism, reps - 1, eng.stats.get("lockstep_fits")), flush=True)
        bad += mism
    os.environ.pop("SCD_ESTEP_MERGED", None)
print("STRESS", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
