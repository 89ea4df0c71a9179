"""Same GEMM kernel, three operand contents: random normal, all zero, constant.  On MI355X the rate differs by ~40 %: with random
operands the chip is power-limited (DVFS), not schedule-limited.  python tools/gemm_data_power.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
def bench(m, n, k, mode, iters=10):
    if mode == "zero":
        a = torch.zeros(m, k, device="cuda").half(); w = torch.zeros(n, k, device="cuda").half()
    elif mode == "const":
        a = torch.full((m, k), 0.5, device="cuda").half(); w = torch.full((n, k), 0.25, device="cuda").half()
    else:
        a = (torch.randn(m, k, device="cuda") * 0.5).half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    for _ in range(3): ops.gemm_f16(a, w, None, None, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_f16(a, w, None, None, 0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-6s m=%d n=%d k=%d: %8.1f us %7.1f TFLOP/s" % (mode, m, n, k, us, 2.0 * m * n * k / us / 1e6), flush=True)
for mode in ("randn", "zero", "const", "randn"):
    bench(8192, 8192, 8192, mode)
    bench(100864, 3072, 768, mode)
