import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scd_amd import ops
from oracle import kmeans_oracle as ko
rs = np.random.RandomState(0)
x = rs.randn(3000, 64).astype(np.float32) + 5.0
c = np.repeat(rs.randn(6, 64).astype(np.float32) + 5.0, 2, axis=0)
c[1::2] += 1e-6
c[3] = c[2]
data = ops.KMeansData(torch.from_numpy(x).cuda())
lab, ref = data.estep(torch.from_numpy(c).cuda(), return_refined=True)
olab, omind, _ = ko.estep(x, c)
lab = lab.cpu().numpy()
bad = np.nonzero(lab != olab)[0]
print("refined", int(ref), "mismatch", len(bad))
d = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1)
for i in bad[:20]:
    print(i, "gpu", lab[i], "oracle", olab[i], "d2", np.sort(d[i])[:4], np.argsort(d[i], kind="stable")[:4])
ws = data._ws[("e", c.shape[0])].cpu().numpy().view(np.uint8)
def al(x): return (x + 255) // 256 * 256
n, d = x.shape; kp, dp = 128, 128
hdr = ws[:64].view(np.int32)
print("cmax_bits", hdr[0], "flag_cnt", hdr[1], "full_cnt", hdr[2])
off = 64 + al(4 * kp) + al(2 * kp * dp) + al(4 * kp * dp)
flags = ws[off:off + 4 * n].view(np.int32); off += al(4 * n)
fcand = ws[off:off + 4 * n].view(np.int32); off += al(4 * n)
fulls = ws[off:off + 4 * n].view(np.int32)
fl = flags[:hdr[1]]; fc = fcand[:hdr[1]]; fu = fulls[:hdr[2]]
print("unique flag rows", len(np.unique(fl)), "unique full", len(np.unique(fu)), "overlap", len(np.intersect1d(fl, fu)))
for i in bad[:20]:
    w = np.nonzero(fl == i)[0]
    print(i, "in flag:", [(int(fc[j]) & 0xffff, int(fc[j]) >> 16) for j in w], "in full:", int((fu == i).sum()))
