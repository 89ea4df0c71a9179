"""Microbenchmark of scd_sim_topk (N x 512 features against V=21000 names), HIP events, random unit vectors."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
n, v, d = int(sys.argv[1]) if len(sys.argv) > 1 else 126976, 21000, 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=-1).half()
wt = torch.nn.functional.normalize(torch.randn(v, d, device="cuda"), dim=-1).half()
for mode in ("raw", "softmax"):
    for _ in range(2): idx, val, fb = ops.sim_topk(f, wt, k, mode, return_fallback=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.sim_topk(f, wt, k, mode)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("sim_topk[%s] k=%d n=%d v=%d: %.3f ms  %.1f TFLOP/s  fallback rows %d" % (mode, k, n, v, ms, 2.0 * n * v * d / ms / 1e9, int(fb)))
if len(sys.argv) > 3:          # the same product on the same tensors through torch.mm (hipBLASLt), logits materialised, no top-k
    for _ in range(2): torch.mm(f, wt.t())
    e0.record()
    for _ in range(5): torch.mm(f, wt.t())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("torch.mm (hipBLASLt) n=%d v=%d d=%d: %.3f ms  %.1f TFLOP/s  (writes the %d x %d fp16 logits: %.1f GB)" % (n, v, d, ms, 2.0 * n * v * d / ms / 1e9, n, v, n * v * 2 / 1e9))
