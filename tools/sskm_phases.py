"""Where the SSKM stage of the default bench goes: lock-step seeding vs the Lloyd iterations of the ten restarts (wall, with a device
synchronisation around each phase).  python tools/sskm_phases.py [n] [d] [k]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km, ops
from tools.kmeans_bench import clustered_features

n, d, k = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 126976), (2, 512), (3, 100)))
x, y, _ = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda().half().float()
eng = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0)
t = {"kpp": 0.0, "lloyd": 0.0, "iters": 0}
orig_kpp, orig_lloyd = eng.kpp_lockstep, eng._lloyd
def kpp(*a, **kw):
    torch.cuda.synchronize(); t0 = time.time(); r = orig_kpp(*a, **kw); torch.cuda.synchronize(); t["kpp"] += time.time() - t0; return r
def lloyd(*a, **kw):
    torch.cuda.synchronize(); t0 = time.time(); r = orig_lloyd(*a, **kw); torch.cuda.synchronize(); t["lloyd"] += time.time() - t0
    t["iters"] += r[3]; return r
eng.kpp_lockstep, eng._lloyd = kpp, lloyd
dp = (d + 127) // 128 * 128
algo = n * dp * 2 + 4 * n + ((k + 127) // 128 * 128) * dp * 2
for rep in range(3):
    t.update(kpp=0.0, lloyd=0.0, iters=0)
    ops.kmeans_timing(True)                     # HIP events around every streaming-filter launch (scd_kmeans_timing)
    torch.cuda.synchronize(); t0 = time.time()
    eng.fit(X)
    torch.cuda.synchronize(); tot = time.time() - t0
    smp = ops.kmeans_timing(False) * 1e3                    # us per launch, call order: restart r, iteration i = smp[10 r + i]
    late = np.array([smp[j] for j in range(len(smp)) if j % 10 >= 2]) if len(smp) % 10 == 0 else smp
    print("fit %.2f ms: seeding (lock-step, %d rounds) %.2f ms, Lloyd %.2f ms over %d iterations (%.0f us each); estep_stream_kernel "
          "inside the loop: %d launches, %.1f us average (iterations 0-1 of a restart: %.1f us; iterations "
          ">= 2: average %.1f / median %.1f / min %.1f us = %.0f GB/s on %.1f MB at the average; HIP-event brackets, dispatch latency included)"
          % (tot * 1e3, k - 1, t["kpp"] * 1e3, t["lloyd"] * 1e3, t["iters"], t["lloyd"] * 1e6 / max(t["iters"], 1), len(smp),
             smp.mean(), np.mean([smp[j] for j in range(len(smp)) if j % 10 < 2]) if len(smp) % 10 == 0 else float("nan"),
             late.mean(), np.median(late), late.min(), algo / late.mean() / 1e3, algo / 1e6))
    if rep == 2:
        print("   per launch (us), restart 0:", " ".join("%.1f" % v for v in smp[:10]))
