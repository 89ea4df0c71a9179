"""An SSKM fit at the default bench's shape (126,976 x 512 clustered fp16-exact features, K = 100, half of the old classes' rows labelled,
ten restarts x ten iterations) for a rocprofv3 --kernel-trace --stats pass: which kernels the lock-step Lloyd loops spend their time in.
    python tools/lloyd_multi_prof.py [fits]      (SCD_ESTEP_MERGED / SCD_LLOYD_STREAMS / SCD_LLOYD_LOCKSTEP select the variant)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km
from tools.kmeans_bench import clustered_features

n, d, k = 126976, 512, 100
x, y, _ = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda().half().float()
yt = torch.from_numpy(y).cuda()
mask = torch.from_numpy((y < k // 2) & (np.random.RandomState(5).rand(n) < 0.5)).cuda()
fits = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rep in range(fits):
    eng = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=10, random_state=rep)
    torch.cuda.synchronize(); t0 = time.time()
    eng.fit_mix(X[~mask], X[mask], yt[mask])
    torch.cuda.synchronize()
    print("fit %d: %.2f ms, stats %s" % (rep, (time.time() - t0) * 1e3, eng.stats), flush=True)
