"""Fuzz the naming oracle against the REFERENCE itself (build container only): the reference's top-k blocks (main_unsup.py:504-531,
main_ptsup.py:526-545) and both vote loops (main_unsup.py:568-614, main_ptsup.py:588-676) exec'd on fp16-exact synthetic inputs of many
seeds and shapes (oracle/gen_golden.py:ref_topk_votes_f16) against oracle/naming_oracle.py: index lists and every iteration of both traces.
PYTHONHASHSEED=0 python tools/ref_fuzz_naming.py [first_seed] [n_cases]"""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PYTHONHASHSEED") == "0", "set PYTHONHASHSEED=0 (set-of-str order at main_ptsup.py:664)"
import numpy as np
from oracle import gen_golden as gg, naming_oracle as no, synth
gg.install_stubs(gg.NxMinCostFlow)

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
shapes = [(600, 64, 8, 300, 0.9, 0.5), (900, 128, 12, 500, 1.1, 0.6), (1500, 512, 20, 800, 0.9, 0.5), (700, 64, 10, 250, 1.3, 0.8), (1200, 256, 16, 1000, 1.0, 0.4)]
bad = 0
for c in range(cases):
    n, d, k, v, noise, jitter = shapes[c % len(shapes)]
    seeds = tuple(s0 + 10 * c + j for j in range(5))
    with contextlib.redirect_stdout(io.StringIO()):
        g = gg.ref_topk_votes_f16(n, d, k, v, seeds, noise=noise, jitter=jitter)
    x, y, cent = synth.clustered_features(n, d, k, seed=seeds[0], center_seed=seeds[1], noise=noise)
    w = synth.vocabulary(v, d, cent, seed=seeds[2], jitter=jitter, dtype=np.float32)
    perm, mask_lab = synth.labelled_split(y, k, prop=0.5, seed=seeds[3])
    f16, w16 = x[perm].astype(np.float16), w.astype(np.float16)
    nouns = synth.nouns_list(v)
    res = []
    iu, _ = no.sim_topk(f16, w16, 5, "softmax")
    ip, _ = no.sim_topk(f16, w16, 5, "raw")
    res.append(("topk", np.array_equal(iu, g["idx_unsup"]) and np.array_equal(ip, g["idx_ptsup"])))
    kk, topk, ncv, ncl = g["vu_cfg"].tolist()
    tr = no.vote_loop_unsup(g["idx_unsup"], g["vu_preds0"], f16, w16, nouns, kk, topk, ncv, ncl)
    ok = len(tr) == int(g["vu_iters"]) and all(np.array_equal(np.asarray(t[key]), g[gk % i]) for i, t in enumerate(tr)
                                              for key, gk in (("voted", "vu_voted_%d"), ("ind", "vu_ind_%d"), ("cand", "vu_cand_%d"), ("u_preds", "vu_preds_%d")))
    res.append(("unsup votes (%d it)" % int(g["vu_iters"]), ok))
    kk, n_lab, topk, ncv, ncl = g["vp_cfg"].tolist()
    ml = g["vp_mask_lab"]
    tr = no.vote_loop_ptsup(g["idx_ptsup"][~ml], g["vp_all_preds0"], ml, f16[~ml], w16, nouns, [nouns[c2] for c2 in range(n_lab)], kk, topk, ncv, ncl)
    ok = len(tr) == int(g["vp_iters"]) and all(np.array_equal(np.asarray(t[key]), g[gk % i]) for i, t in enumerate(tr)
                                              for key, gk in (("voted", "vp_voted_%d"), ("ind", "vp_ind_%d"), ("cand", "vp_cand_%d"), ("u_preds", "vp_preds_%d"),
                                                              ("unlab_cluster_idx", "vp_unlab_%d")))
    res.append(("ptsup votes (%d it)" % int(g["vp_iters"]), ok))
    print("case %d n=%d d=%d k=%d v=%d seeds %s: %s" % (c, n, d, k, v, seeds[0], ", ".join("%s %s" % (a, "ok" if b else "MISMATCH") for a, b in res)), flush=True)
    bad += sum(not b for _, b in res)
print("FUZZ", "MISMATCHES: %d" % bad if bad else "ok")
