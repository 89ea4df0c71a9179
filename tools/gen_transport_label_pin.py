"""How tests/golden/transport_labels_r5.npz was made: the labels of ROUND 5's transport solver on five seeded problems, so that later
solvers can be held to the same tie-breaking (tests/test_cpu_abi_and_host.py::test_transport_labels_equal_round5_solver).
Needs round 5's library:  git worktree add /tmp/r5 acd1433 && (cd /tmp/r5 && python -m scd_amd.build), then
    python tools/gen_transport_label_pin.py /tmp/r5/scd_amd/lib/libscd_hip.so tests/golden/transport_labels_r5.npz"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import transport_oracle as to                       # noqa: E402  (int_costs only: the fixture's inputs)

lib = C.CDLL(sys.argv[1])
P = lambda a: C.c_void_p(a.ctypes.data)                          # noqa: E731
out = {}
cases = [(900, 12, 50, 110, 21, False), (2000, 40, 30, 80, 22, False), (1500, 20, 60, 90, 23, True), (3000, 120, 15, 60, 24, False),
         (700, 7, 100, 100, 25, True)]
for i, (n, k, smin, smax, seed, tie) in enumerate(cases):
    rs = np.random.RandomState(seed)
    pts, cen = rs.randn(n, 6), rs.randn(k, 6) * 1.3
    cost = to.int_costs(((pts[:, None] - cen[None]) ** 2).sum(-1).astype(np.float32))
    if tie:
        cost = (cost // 50 * 50).astype(np.int32)                # many equal costs: the tie-breaking rules decide the labels
    lab = np.zeros(n, dtype=np.int32)
    tot = C.c_int64(0)
    assert lib.scd_transport_solve(P(cost), C.c_int64(n), k, smin, smax, P(lab), C.byref(tot)) == 0
    out["lab%d" % i], out["tot%d" % i], out["bounds%d" % i] = lab, np.int64(tot.value), np.array([smin, smax])
np.savez_compressed(sys.argv[2], **out)
