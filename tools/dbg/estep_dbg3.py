import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SCD_ESTEP_REFINE_SPLIT"] = "2"
from scd_amd import ops
from oracle import synth
x, y, mask = synth.blob_case(3000, 768, 20, 3)
x = x[~mask]
n, d = x.shape
rs = np.random.RandomState(1)
c = x[rs.choice(n, 20, replace=False)].copy()
data = ops.KMeansData(torch.from_numpy(x).cuda())
lab = data.estep(torch.from_numpy(c).cuda()).cpu().numpy()
def al(v): return (v + 255) // 256 * 256
prep = data.prep.cpu().numpy().view(np.uint8)
dp, kp = 768, 128
xn_off = al(64 + 8 * dp); xh_off = xn_off + al(4 * n)
xh = prep[xh_off:xh_off + 2 * n * dp].view(np.float16).reshape(n, dp).astype(np.float64)
ws = data._ws[("e", 20)].cpu().numpy().view(np.uint8)
cn = ws[64:64 + 4 * kp].view(np.float32)
ch = ws[64 + al(4 * kp):64 + al(4 * kp) + 2 * kp * dp].view(np.float16).reshape(kp, dp).astype(np.float64)
s = cn[None, :20].astype(np.float64) - 2 * xh @ ch[:20].T
best = s.argmin(1)
bad = np.nonzero(best != lab)[0]
srt = np.sort(s, 1)
print("n", n, "filter mismatch", len(bad), "score range", s.min(), s.max())
for i in bad[:40]:
    print(i, "blk", i // 128, "r", i % 32, "pb", (i % 128) // 32, "gpu", lab[i], "emu", best[i], "s_gpu %.3f s_best %.3f s_2nd %.3f" % (s[i, lab[i]] if lab[i] < 20 else np.nan, srt[i, 0], srt[i, 1]))
g32 = (n + 31) // 32
G = min((n + 127) // 128, 256)
starts = [32 * (g32 * b // G) for b in range(G + 1)]
import collections
hist = collections.Counter()
big = collections.Counter()
for i in bad:
    b = max(j for j in range(G) if starts[j] <= i)
    p = i - starts[b]
    hist[(b, p // 32)] += 1
    if s[i, lab[i]] - srt[i, 0] > 100: big[(b, p // 32)] += 1
print("starts", starts)
print("mismatch by (block,pb):", sorted(hist.items()))
print("gross (>100) by (block,pb):", sorted(big.items()))
