import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops, kmeans as km
from oracle import synth
n, d, k = 30000, 768, 100
x, y, _ = synth.clustered_features(n, d, k, seed=61, center_seed=62, noise=0.8)
X = torch.from_numpy(x.astype(np.float16).astype(np.float32)).cuda()
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["SCD_MSTEP_DELTA"] = mode
    log = []
    e = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=2)
    orig = e._lloyd_pipelined
    def wrap(*a, **kw):
        r = orig(*a, **kw)
        log.append((float(r[1]), r[3], r[2].double().sum().item()))
        return r
    e._lloyd_pipelined = wrap
    e.fit(X)
    print("mode", mode, "inertia %.10g n_iter %s" % (float(e.inertia_), e.n_iter_), "restarts:", [(("%.9g" % a), b, ("%.12g" % c)) for a, b, c in log], e.stats)
    res.setdefault(mode, []).append(e.cluster_centers_.clone())
print("delta vs full: centres equal", torch.equal(res["1"][0], res["0"][0]), "n diff", int((res["1"][0] != res["0"][0]).sum()), "max", (res["1"][0] - res["0"][0]).abs().max().item())
print("delta vs delta:", torch.equal(res["1"][0], res["1"][1]), " full vs full:", torch.equal(res["0"][0], res["0"][1]))
