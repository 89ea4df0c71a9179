"""estep_stream_kernel's duration against the row count (converged centres, nothing flagged: the steady state of a Lloyd loop):
T(n) = a + n / rate separates the launch's fixed cost (ramp, centre fragments, first unit's latency, tail) from the stream's rate.
HIP events around the kernel itself (scd_kmeans_timing).   python tools/estep_scaling.py [d] [k]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops

d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = torch.Generator(device="cuda").manual_seed(1)
cen = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
nmax = 512 * 1024
y = torch.randint(0, k, (nmax,), device="cuda", generator=g)
x = torch.nn.functional.normalize(cen[y] + (0.5 / d ** 0.5) * torch.randn(nmax, d, device="cuda", generator=g), dim=-1).half().float()
rows = []
for n in (8192, 16384, 32768, 65536, 98304, 131072, 196608, 262144, 393216, 524288):
    data = ops.KMeansData(x[:n].contiguous())
    lab, ref = data.estep(cen, return_refined=True)
    for _ in range(5):
        data.estep(cen, expect_few=True)
    torch.cuda.synchronize()
    ops.kmeans_timing(True)
    for _ in range(40):
        data.estep(cen, expect_few=True)
    torch.cuda.synchronize()
    t = ops.kmeans_timing(False) * 1e3
    dp = (d + 127) // 128 * 128
    mb = n * dp * 2 / 1e6
    rows.append((n, mb, float(np.median(t)), float(t.min()), int(ref)))
    print("n = %7d  %7.1f MB  kernel median %7.2f us  min %7.2f us  -> %5.0f GB/s  (rows flagged %d)" % (n, mb, np.median(t), t.min(), mb / np.median(t) * 1e3, int(ref)), flush=True)
a = np.array([[1.0, r[1]] for r in rows[3:]])
b = np.array([r[2] for r in rows[3:]])
(c0, c1), *_ = np.linalg.lstsq(a, b, rcond=None)
print("fit over n >= 65,536: T = %.1f us + bytes / %.2f TB/s   (1 MB/us = 1 TB/s)" % (c0, 1.0 / c1))
