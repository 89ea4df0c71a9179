import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from scd_amd import ops
from kmeans_bench import clustered_features
n, d, k = 95000, 768, 100
x, y, cent = clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
X = torch.from_numpy(x).cuda()
C2 = X[torch.randperm(n, device="cuda")[:k]].clone()
data = ops.KMeansData(X)
lab, ref = data.estep(C2, return_refined=True)
torch.cuda.synchronize()
hdr = data._ws[("e", k)][:64].cpu().numpy().view(np.int32)
print("flag_cnt", hdr[1], "full_cnt", hdr[2])
