"""scd_kmeans_dist at the ConSSKM shape (9,000 x 768 against 120 centres, with integer costs): microseconds per call (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
n, d, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9000, 768, 120)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, device="cuda", generator=g)
c = torch.randn(k, d, device="cuda", generator=g)
data = ops.KMeansData(x)
for _ in range(5):
    data.dist(c, sqrt=True, with_cost=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    data.dist(c, sqrt=True, with_cost=True)
e1.record(); torch.cuda.synchronize()
print("SCD_DIST_CG=%s  n=%d d=%d k=%d: %.1f us per call" % (os.environ.get("SCD_DIST_CG", "8 (default)"), n, d, k, e0.elapsed_time(e1) * 1e3 / 50))
