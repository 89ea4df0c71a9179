"""Microbenchmark of scd_gemm_f16 on the ViT-B/16 shapes (HIP events, random data)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops

def bench(m, n, k, act=0, bias=True, res=False, iters=20):
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    b = torch.randn(n, device="cuda") if bias else None
    r = torch.randn(m, n, device="cuda").half() if res else None
    for _ in range(3):
        ops.gemm_f16(a, w, b, r, act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_f16(a, w, b, r, act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("m=%6d n=%5d k=%5d act=%d res=%d : %8.1f us  %7.1f TFLOP/s" % (m, n, k, act, res, us, 2.0 * m * n * k / us / 1e6), flush=True)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = (B * 197 + 255) // 256 * 256
    bench(M, 2304, 768)
    bench(M, 768, 768, res=True)
    bench(M, 3072, 768, act=1)
    bench(M, 768, 3072, res=True)
    bench(4096, 4096, 4096, bias=False)
    bench(8192, 8192, 8192, bias=False, iters=5)
