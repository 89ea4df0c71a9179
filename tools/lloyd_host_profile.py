"""Where the host time of one Lloyd iteration goes (KMeansEngine._lloyd_pipelined, fused path): wall per call site."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
from tools.kmeans_bench import clustered_features
n, d, k = 95000, 768, 100
x, y, cent = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda().half().float()
data = ops.KMeansData(X)
x16 = ops.f16_exact(X)
B = ops.LloydBuffers(data, X, x16, k)
B.c0.copy_(torch.from_numpy(cent).cuda())
ring = [torch.zeros(5, dtype=torch.float64).pin_memory() for _ in range(2)]
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for full in (True, False):
    T.clear()
    c = B.c0
    for w in range(3):
        B.step_delta(c, B.c[w & 1], B.stats[0], True, True); c = B.c[w & 1]
    torch.cuda.synchronize()
    N = 200
    t_all = time.perf_counter()
    for it in range(N):
        t0 = time.perf_counter(); B.step_delta(c, B.c[(it + 1) & 1], B.stats[it & 1], True, full); tick("step call", t0)
        t0 = time.perf_counter(); ring[it & 1].copy_(B.stats[it & 1], non_blocking=True); tick("stats D2H copy_", t0)
        t0 = time.perf_counter(); snap = B.lab32.clone(); tick("labels clone", t0)
        t0 = time.perf_counter(); ev = torch.cuda.Event(); ev.record(); tick("event", t0)
        c = B.c[(it + 1) & 1]
        if it:
            t0 = time.perf_counter(); pev.synchronize(); h = ring[(it - 1) & 1].numpy(); v = np.float32(np.float32(h[1]) + np.float32(h[0])); tick("settle (event sync + numpy)", t0)
        pev = ev
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_all
    print("full M-step" if full else "incremental M-step", ": %.1f us wall per iteration;" % (tot / N * 1e6), "; ".join("%s %.1f" % (k2, v / N * 1e6) for k2, v in T.items()))
