"""`--cluster KM` at the headline shape: scd_amd.cluster.KMeans(n_clusters=100, random_state=0).fit on 95,000 x 768 fp16-exact clustered
features (main_unsup.py:362; n_init = 10 under the reference's scikit-learn 1.0.2 pin).  Prints the fit's wall time and its parts."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
from scd_amd.cluster import KMeans
from scd_amd.kmeans import check_random_state


def clustered(n, d, k, seed=13, noise=0.6):
    g = torch.Generator(device="cuda").manual_seed(seed)
    c = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
    y = torch.randint(0, k, (n,), device="cuda", generator=g)
    x = torch.nn.functional.normalize(c[y] + noise / d ** 0.5 * torch.randn(n, d, device="cuda", generator=g), dim=-1)
    return x.half().float().contiguous()


def main():
    n, d, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (95000, 768, 100)
    x = clustered(n, d, k)
    for compat in ("1.0.2", "1.7.2"):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            km = KMeans(n_clusters=k, random_state=0, sklearn_compat=compat).fit(x)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        print("KMeans(%d, random_state=0) compat %s on %d x %d: fit %.1f ms (n_iter of the kept start %d, inertia %.4f)" % (k, compat, n, d, (t1 - t0) * 1e3, km.n_iter_, km.inertia_), flush=True)
    # the parts, 1.0.2 mode: seeding of the ten starts, then the ten Lloyd runs
    km = KMeans(n_clusters=k, random_state=0)
    data = ops.KMeansData(x)
    x16 = ops.f16_exact(x)
    lb = ops.LloydBuffers(data, x, x16, k)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        seeds = km._seed(data, x16, check_random_state(0), 10)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        its = []
        for j in range(10):
            lab, inertia, cen, n_iter = km._lloyd(data, seeds[j], 1e-4 * 1.0 / d, lb)
            its.append(n_iter)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("  seeding of 10 starts (99 rounds x 60 candidates): %.1f ms; 10 Lloyd runs: %.1f ms (%s iterations)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, its))
    os.environ["SCD_KM_FILTER_FROM"] = "-1"


if __name__ == "__main__":
    main()
