import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scd_amd import ops
from oracle import kmeans_oracle as ko, synth
x, y, mask = synth.blob_case(3000, 768, 20, 3)
x = x[~mask]
n, d = x.shape
rs = np.random.RandomState(1)
for trial in range(4):
    c = x[rs.choice(n, 20, replace=False)].copy()
    if trial == 3: c[5] = np.nan
    data = ops.KMeansData(torch.from_numpy(x).cuda())
    lab, ref = data.estep(torch.from_numpy(c).cuda(), return_refined=True)
    olab, omind, _ = ko.estep(x, c)
    lab = lab.cpu().numpy()
    bad = np.nonzero(lab != olab)[0]
    print("n", n, "refined", int(ref), "mismatch", len(bad))
    d2 = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1)
    ws = data._ws[("e", 20)].cpu().numpy().view(np.uint8)
    def al(v): return (v + 255) // 256 * 256
    kp, dp = 128, 768
    hdr = ws[:64].view(np.int32)
    off = 64 + al(4 * kp) + al(2 * kp * dp) + al(4 * kp * dp)
    flags = ws[off:off + 4 * n].view(np.int32); off += al(4 * n)
    fcand = ws[off:off + 4 * n].view(np.int32); off += al(4 * n)
    fulls = ws[off:off + 4 * n].view(np.int32)
    fl = flags[:hdr[1]]; fc = fcand[:hdr[1]]; fu = fulls[:hdr[2]]
    print("flag_cnt", hdr[1], "full_cnt", hdr[2])
    for i in bad[:10]:
        w = np.nonzero(fl == i)[0]
        print(i, "gpu", lab[i], "oracle", olab[i], np.sort(d2[i])[:3], np.argsort(d2[i], kind="stable")[:3],
              "in flag:", [(int(fc[j]) & 0xffff, int(fc[j]) >> 16) for j in w], "in full:", int((fu == i).sum()))
