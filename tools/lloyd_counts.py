"""Per-iteration counts of an SSKM fit: rows the E-step re-evaluated exactly and rows whose label changed (stats[3], stats[4] of
scd_kmeans_lloyd_step_delta) - what the SCD_ESTEP_FEW hint and the incremental M-step's threshold are chosen by.
python tools/lloyd_counts.py [n] [d] [k] [noise]"""
import os, sys
os.environ["SCD_LLOYD_RUN"] = "0"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km, ops
from tools.kmeans_bench import clustered_features

n, d, k = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 95000), (2, 768), (3, 100)))
noise = float(sys.argv[4]) if len(sys.argv) > 4 else 0.8
x, y, _ = clustered_features(n, d, k, seed=21, center_seed=22, noise=noise)
X = torch.from_numpy(x).cuda().half().float()
orig = ops.LloydBuffers.step_delta
log = []
def step(self, c_in, c_out, stats, few, full):
    orig(self, c_in, c_out, stats, few, full)
    torch.cuda.synchronize()
    s = stats.cpu().numpy()
    log.append((bool(full), bool(few), int(s[3]), int(s[4]), float(s[2])))
ops.LloydBuffers.step_delta = step
eng = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=0)
eng.fit(X)
for i, (full, few, ref, chg, sh) in enumerate(log):
    print("call %2d  %s %s  refined %6d  changed %6d  shift %.3e" % (i, "full " if full else "delta", "few" if few else "   ", ref, chg, sh))
