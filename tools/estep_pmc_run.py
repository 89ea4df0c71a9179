"""Launches for a rocprofv3 --pmc pass over the streaming E-step: converged centres (nothing flagged), K = 100, two regimes per width -
rows that stay in the 256-MB Infinity Cache between launches (n = 98,304) and rows that stream from HBM (n = 393,216 / 524,288).
    python tools/estep_pmc_run.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops

k = 100
for d in (512, 768):
    g = torch.Generator(device="cuda").manual_seed(1)
    cen = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
    nmax = 524288
    y = torch.randint(0, k, (nmax,), device="cuda", generator=g)
    x = torch.nn.functional.normalize(cen[y] + (0.5 / d ** 0.5) * torch.randn(nmax, d, device="cuda", generator=g), dim=-1).half().float()
    for n in (98304, 393216, 524288):
        data = ops.KMeansData(x[:n].contiguous())
        for _ in range(12):
            data.estep(cen, expect_few=True)
        torch.cuda.synchronize()
        del data
