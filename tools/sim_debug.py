"""Debug aid: dump the pass-1 candidate lists of scd_sim_topk and compare them with the exact top lists per (image, half)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops, _lib
from scd_amd._lib import ptr, check
n, v, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 2000, 512, 5
TM = 4 if k <= 3 else 8
g = torch.Generator(device="cuda").manual_seed(0)
f = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=-1).half()
wt = torch.nn.functional.normalize(torch.randn(v, d, device="cuda", generator=g), dim=-1).half()
L = _lib.load()
idx = torch.empty((n, k), dtype=torch.int64, device="cuda"); val = torch.empty((n, k), dtype=torch.float32, device="cuda")
fb = torch.zeros(1, dtype=torch.int32, device="cuda")
nb = L.scd_sim_topk_ws_bytes(n, d, v, k)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
ops._need_cuda(f)
check(L.scd_sim_topk(ops.handle(), ptr(f), ptr(wt), n, d, v, 100.0, k, 0, ptr(idx), ptr(val), ptr(fb), ptr(ws), nb, ops.stream_ptr()))
torch.cuda.synchronize()
csz = (n * 2 * 8 * 4 + 255) // 256 * 256
cval = ws[64:64 + n * 2 * TM * 4].view(torch.float32).reshape(n, 2, TM).cpu().numpy()
cidx = ws[64 + csz:64 + csz + n * 2 * TM * 4].view(torch.int32).reshape(n, 2, TM).cpu().numpy()
lg = (f.double() @ wt.double().t() * 100.0).cpu().numpy()
# names seen by half hh: (name % 8) // 4 == hh  (rows (i&3) + 8(i>>2) + 4hh of each 32-name unit)
name = np.arange(v)
bad = []
for i in range(n):
    for hh in range(2):
        mine = name[((name % 8) // 4) == hh]
        top = mine[np.argsort(-lg[i, mine], kind="stable")[:TM]]
        if set(top.tolist()) != set(cidx[i, hh].tolist()):
            bad.append((i, hh))
print("fallback rows", int(fb), "bad lists", len(bad), "of", 2 * n)
b = np.array(bad) if bad else np.zeros((0, 2), int)
if len(b):
    print("by half:", np.bincount(b[:, 1], minlength=2), "by (img%64)//32:", np.bincount((b[:, 0] % 64) // 32, minlength=2),
          "by img%32<16:", np.bincount((b[:, 0] % 32) // 16, minlength=2), "by wave:", np.bincount((b[:, 0] % 256) // 64, minlength=4))
    i, hh = bad[0]
    mine = name[((name % 8) // 4) == hh]
    top = mine[np.argsort(-lg[i, mine], kind="stable")[:TM]]
    print("first bad", i, hh, "got", cidx[i, hh], cval[i, hh], "want", top, lg[i, top])
    print("got idx decoded: unit", cidx[i, hh] // 32, "pos", cidx[i, hh] % 32)
print("final ok:", bool((idx.cpu().numpy() == np.argsort(-lg, axis=1, kind="stable")[:, :k]).all()))
