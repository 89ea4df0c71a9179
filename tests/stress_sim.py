"""Stress run for scd_sim_topk (not collected by pytest): repeated full-size calls must return identical indices and values,
and the first call must match the float64 oracle on a sample of rows.  python tests/stress_sim.py [repeats]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops
from oracle import naming_oracle as no

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0
for (n, v, d, k, mode) in [(126976, 21000, 512, 5, "softmax"), (126976, 21000, 512, 5, "raw"), (50001, 9999, 512, 8, "raw"), (30000, 21000, 256, 3, "softmax")]:
    g = torch.Generator(device="cuda").manual_seed(n + v)
    f = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=-1).half()
    wt = torch.nn.functional.normalize(torch.randn(v, d, device="cuda", generator=g), dim=-1).half()
    idx0, val0, fb = ops.sim_topk(f, wt, k, mode, return_fallback=True)
    rows = np.random.RandomState(1).choice(n, 256, replace=False)
    oi, ov = no.sim_topk(f[rows].cpu().numpy(), wt.t().contiguous().cpu().numpy(), k, mode)
    ok0 = np.array_equal(idx0.cpu().numpy()[rows], oi)
    mism = 0
    for _ in range(reps):
        idx, val = ops.sim_topk(f, wt, k, mode)
        if not (torch.equal(idx, idx0) and torch.equal(val, val0)):
            mism += 1
    print("n=%d v=%d d=%d k=%d %s: oracle sample %s, fallback rows %d, %d/%d repetitions differ" % (n, v, d, k, mode, "ok" if ok0 else "MISMATCH", int(fb), mism, reps))
    bad += (not ok0) + mism
print("STRESS", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
