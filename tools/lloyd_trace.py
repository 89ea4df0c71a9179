"""Workload for a rocprofv3 kernel trace of the Lloyd loop: two SSKM fits (the first warms up) without scd_kmeans_timing.
python tools/lloyd_trace.py [n] [d] [k]; the trace is read by tools/trace_gaps.py."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km
from tools.kmeans_bench import clustered_features

n, d, k = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 95000), (2, 768), (3, 100)))
x, y, _ = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda().half().float()
for rep in range(2):
    eng = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=0)
    t = {"lloyd": 0.0, "iters": 0}
    orig = eng._lloyd
    def lloyd(*a, **kw):
        torch.cuda.synchronize(); t0 = time.time(); r = orig(*a, **kw); torch.cuda.synchronize(); t["lloyd"] += time.time() - t0
        t["iters"] += r[3]; return r
    eng._lloyd = lloyd
    eng.fit(X)
    print("fit %d: Lloyd %.2f ms over %d iterations (%.0f us each, no event timing)" % (rep, t["lloyd"] * 1e3, t["iters"], t["lloyd"] * 1e6 / max(t["iters"], 1)))
