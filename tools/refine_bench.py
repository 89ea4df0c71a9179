"""E-step call at mid-fit centres (a few thousand flagged rows) for several refine-grid shapes; needs the -DSCD_ABLATE build
(SCD_HIP_LIB=scd_amd/lib/libscd_hip_ablate.so).  python tools/refine_bench.py [n] [d] [k]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import kmeans as km, ops
from tools.kmeans_bench import clustered_features

n, d, k = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 95000), (2, 768), (3, 100)))
x, y, _ = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda().half().float()
eng = km.KMeansEngine(k=k, tolerance=1e-4, max_iterations=4, n_init=1, random_state=0)
eng.fit(X)
c = torch.nan_to_num(eng.cluster_centers_.float()).contiguous()
data = ops.KMeansData(X)
lab0, ref = data.estep(c, return_refined=True)
hdr = data._ws[("e", k)][:64].view(torch.int32).cpu()
print("flagged rows at these centres: %d (pair list %d, all-centres list %d)" % (int(ref.item()), int(hdr[1]), int(hdr[2])))
def timeit(f, reps=40):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for grid, pair in ((1024, 1024), (1024, 0), (1024, 256), (768, 256), (512, 256), (1024, 512), (2048, 1024), (2048, 512), (4096, 1024), (4096, 2048), (8192, 2048), (3072, 512), (1280, 256)):
    os.environ["SCD_REFINE_GRID"], os.environ["SCD_REFINE_PAIR"] = str(grid), str(pair)
    t = timeit(lambda: data.estep(c))
    lab = data.estep(c)
    print("grid %5d (pair blocks %4d): estep call %.1f us   labels equal: %s" % (grid, pair, t, bool(torch.equal(lab, lab0))))
