"""scd_sim_topk on 'crowded' logits: every image feature sits in one blob and P planted names sit near it (the bench's synthetic
CLIP features: the top P logits of every image lie within a fraction of a unit), the other names are random unit vectors.
python tools/sim_crowd.py [n] [k]  - prints the call time for P = 0 / 100 / 1000 and the rows that needed the exact pass."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 126976
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
v, d = 21000, 512
g = torch.Generator(device="cuda").manual_seed(1)
nz = lambda t: torch.nn.functional.normalize(t, dim=-1)
b = nz(torch.randn(1, d, device="cuda", generator=g))
for P, spread in ((0, 0.0), (100, 0.3), (1000, 0.3), (100, 1.0), (1000, 1.0)):
    f = nz(b + (spread if P else 1e3) / d ** 0.5 * torch.randn(n, d, device="cuda", generator=g)).half()
    wt = nz(torch.randn(v, d, device="cuda", generator=g))
    if P:
        wt[:P] = nz(b + spread / d ** 0.5 * torch.randn(P, d, device="cuda", generator=g))
    wt = wt.half().contiguous()
    for mode in ("raw", "softmax"):
        for _ in range(2): idx, val, fb = ops.sim_topk(f, wt, k, mode, return_fallback=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.sim_topk(f, wt, k, mode)
        e1.record(); torch.cuda.synchronize()
        print("planted %4d spread %.1f  %-7s k=%d: %.3f ms   exact-pass rows %d" % (P, spread, mode, k, e0.elapsed_time(e1) / 5, int(fb)))
