"""Same GEMM binary, same launch, different operand CONTENTS: random normal, all zero, constant - plus torch.mm (hipBLASLt) on the
same tensors.  On MI355X the rate differs by 30-40 %: with random operands the chip lowers its clock under load (DVFS,
MI355X_MICROARCH.md 'DVFS give-back'), so the schedule is not what limits an fp16 GEMM on random data.

  python tools/gemm_data_power.py [out.json]
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d DIR -- python tools/gemm_data_power.py     (clock = GRBM_GUI_ACTIVE / 8 / duration)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops


def operands(m, n, k, mode):
    if mode == "zero":
        return torch.zeros(m, k, device="cuda").half(), torch.zeros(n, k, device="cuda").half()
    if mode == "const":
        return torch.full((m, k), 0.5, device="cuda").half(), torch.full((n, k), 0.25, device="cuda").half()
    return (torch.randn(m, k, device="cuda") * 0.5).half(), (torch.randn(n, k, device="cuda") * k ** -0.5).half()


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    rows = []
    for rnd in range(2):                                          # interleaved rounds in one process
        for mode in ("randn", "zero", "const"):
            for (m, n, k) in ((8192, 8192, 8192), (131072, 3072, 768)):
                a, w = operands(m, n, k, mode)
                wt = w.t()
                for impl, fn in (("scd_gemm_f16", lambda: ops.gemm_f16(a, w, None, None, 0)), ("torch.mm(hipBLASLt)", lambda: torch.mm(a, wt))):
                    us = timeit(fn, 10)
                    tf = 2.0 * m * n * k / us / 1e6
                    rows.append(dict(round=rnd, impl=impl, operands=mode, m=m, n=n, k=k, us=round(us, 1), tflops=round(tf, 1)))
                    print("%d %-20s %-6s m=%d n=%d k=%d: %8.1f us %7.1f TFLOP/s" % (rnd, impl, mode, m, n, k, us, tf), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(dict(device=torch.cuda.get_device_name(0), note="HIP events, 10 launches each after 3 warm-ups; same process", rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
