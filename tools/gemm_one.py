"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py M N K [act] [res] [iters]."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops

m, n, k = (int(x) for x in sys.argv[1:4])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
res = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
a = (torch.randn(m, k, device="cuda") * 0.5).half()
w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
b = torch.randn(n, device="cuda")
r = torch.randn(m, n, device="cuda").half() if res else None
for _ in range(iters):
    ops.gemm_f16(a, w, b, r, act)
torch.cuda.synchronize()
