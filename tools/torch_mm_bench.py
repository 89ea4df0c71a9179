"""Reference point only: torch.matmul (hipBLASLt/rocBLAS) on the same shapes as tools/gemm_bench.py."""
import sys, torch
def bench(m, n, k, iters=20):
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    for _ in range(3):
        torch.matmul(a, w.t())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("torch m=%6d n=%5d k=%5d : %8.1f us  %7.1f TFLOP/s" % (m, n, k, us, 2.0 * m * n * k / us / 1e6), flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = (B * 197 + 255) // 256 * 256
for shp in [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
    bench(*shp)
