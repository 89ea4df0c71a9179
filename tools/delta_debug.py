"""Debug aid: the incremental Lloyd step against the fresh one, iteration by iteration (sums, counts, centres, inertia, changes)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
from oracle import synth
n, d, k = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 30000), (2, 768), (3, 100)))
x, y, _ = synth.clustered_features(n, d, k, seed=61, center_seed=62, noise=0.8)
X = torch.from_numpy(x.astype(np.float16).astype(np.float32)).cuda()
data = ops.KMeansData(X)
x16 = ops.f16_exact(X)
A, B = ops.LloydBuffers(data, X, x16, k), ops.LloydBuffers(data, X, x16, k)
c0 = X[torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))[:k]].contiguous()
for bufs in (A, B):
    bufs.c0.copy_(c0)
ca, cb = A.c0, B.c0
for it in range(10):
    A.step_delta(ca, A.c[it & 1], A.stats[it & 1], False, True)
    B.step_delta(cb, B.c[it & 1], B.stats[it & 1], False, it < 2)
    torch.cuda.synchronize()
    sa, sb = A.stats[it & 1].cpu().numpy(), B.stats[it & 1].cpu().numpy()
    print("it %d: labels equal %s  sums equal %s (max |diff| %.3g)  counts equal %s  centres equal %s | inertia full %.17g delta %.17g  f32 equal %s | shift %.6g %.6g  changed %d %d refined %d"
          % (it, torch.equal(A.lab32, B.lab32), torch.equal(A.sums, B.sums), (A.sums - B.sums).abs().max().item(), torch.equal(A.counts, B.counts),
             torch.equal(A.c[it & 1].nan_to_num(7), B.c[it & 1].nan_to_num(7)), sa[1], sb[1], np.float32(sa[1]) == np.float32(sb[1]), sa[2], sb[2], sa[4], sb[4], sa[3]))
    ca, cb = A.c[it & 1], B.c[it & 1]
