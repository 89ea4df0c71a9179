"""CPU-only checks of the product: the C-ABI library loads and exports every symbol include/scd_hip.h declares,
the host solvers (Munkres, transport) match the reference goldens / the LP optimum, the C oracle restatement matches
the numpy oracle.  No device compute is called."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle import kmeans_oracle as ko
from oracle import naming_oracle as no
from oracle import transport_oracle as to

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from scd_amd import _lib
    return _lib.load()


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "scd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(scd_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"scd_status", "scd_dtype", "scd_sim_mode"}
    from scd_amd import _lib
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.scd_version() >= 100


def test_no_torch_types_in_abi():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "scd_amd", "lib", "libscd_hip.so")],
                         capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l and "scd_" in l]
    assert all(not s.startswith("_Z") for s in syms if s.startswith("scd_"))     # extern "C" entry points
    assert "libtorch" not in subprocess.run(["ldd", os.path.join(ROOT, "scd_amd", "lib", "libscd_hip.so")],
                                            capture_output=True, text=True).stdout


def test_default_library_has_no_ablation_switches():
    """Timing ablations (kernels with pieces removed: wrong results) and the A/B kernels of earlier rounds exist only in the
    -DSCD_ABLATE build (`python -m scd_amd.build --ablate`): the shipped library neither reads those variables nor contains the
    code behind them."""
    blob = open(os.path.join(ROOT, "scd_amd", "lib", "libscd_hip.so"), "rb").read()
    for name in (b"SCD_GEMM_X", b"SCD_SIM_X", b"SCD_ATTN_X", b"SCD_ESTEP_DBG", b"SCD_ESTEP_REFINE_SPLIT", b"SCD_GEMM_MFMA", b"SCD_GEMM_TILE"):
        assert name not in blob, name
    for kern in (b"gemm_w8_kernel", b"gemm_dma16_kernel", b"sim_topk_w4_kernel", b"sim_topk_rb_kernel"):
        assert kern not in blob, kern


def test_product_does_not_import_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "scd_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt:
                    bad.append(f)
    assert not bad, bad


def test_missing_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    from scd_amd import ops, _lib
    with pytest.raises(_lib.ScdError):
        ops.l2norm_rows(torch.zeros(4, 8))
    with pytest.raises(_lib.ScdError):
        _lib.handle()


def test_munkres_matches_reference_goldens(golden):
    from scd_amd import ops
    g = golden("munkres.npz")
    keys = sorted(k[5:] for k in g.files if k.startswith("cost_"))
    for k in keys:
        assert np.array_equal(ops.munkres(g["cost_" + k]), g["ind_" + k]), k
    rs = np.random.RandomState(5)
    for d, hi in ((150, 2), (257, 3), (200, 40)):
        x = rs.randint(0, hi, size=(d, d))
        assert np.array_equal(ops.munkres(x), no.linear_assignment(x))
    assert ops.munkres(np.zeros((0, 0), dtype=int)).shape == (0, 2)


def test_munkres_sparse_matches_reference_and_dense(golden):
    """scd_munkres_sparse (assign_name's solve on the non-zero entries of w) against (i) the reference's own
    linear_assignment on vote-shaped instances (tests/golden/munkres.npz, vote_*), (ii) the dense state machine and the
    oracle on random matrices with heavy ties, negative entries, full rows and empty matrices."""
    from scd_amd import ops
    g = golden("munkres.npz")
    for rep in range(4):
        d = int(g["vote_d_%d" % rep])
        r, c, v = g["vote_rows_%d" % rep], g["vote_cols_%d" % rep], g["vote_vals_%d" % rep]
        assert np.array_equal(ops.munkres_sparse(d, r, c, v), g["vote_ind_%d" % rep]), rep
        w = np.zeros((d, d), dtype=np.int64)
        w[r, c] = v
        assert np.array_equal(ops.munkres(w.max() - w), g["vote_ind_%d" % rep]), rep
    rs = np.random.RandomState(11)
    for rep in range(250):
        d = rs.randint(1, 50)
        w = np.zeros((d, d), dtype=np.int64)
        for i in rs.choice(d, rs.randint(0, d + 1), replace=False):
            cols = rs.choice(d, min(rs.randint(1, 6), d), replace=False)
            w[i, cols] += rs.randint(1, rs.choice([2, 3, 5, 40]), size=len(cols))
            if rep % 3 == 0:
                w[i, cols] *= rs.choice([-1, 1], size=len(cols))
        if d > 3 and rep % 4 == 0:
            w[rs.randint(d)] = rs.randint(1, 4, size=d)                  # a row without background entries
        r, c = np.nonzero(w)
        sp = ops.munkres_sparse(d, r, c, w[r, c])
        assert np.array_equal(sp, ops.munkres(w.max() - w)), rep
        if rep % 10 == 0:
            assert np.array_equal(sp, no.linear_assignment(w.max() - w)), rep
    # duplicates add up like `w[i, col] += v`
    assert np.array_equal(ops.munkres_sparse(3, [0, 0, 1], [2, 2, 2], [1, 1, 3]), ops.munkres(3 - np.array([[0, 0, 2], [0, 0, 3], [0, 0, 0]])))
    assert ops.munkres_sparse(0, [], [], []).shape == (0, 2)
    assert np.array_equal(ops.munkres_sparse(4, [], [], []), np.stack([np.arange(4)] * 2, 1))


def test_munkres_sparse_c4_size_is_fast():
    """BASELINE configs[3]: K = 1000 clusters, num_common_vote 10 / 20 -> D = 10,000 / 20,000.  The dense machine is O(D^3)
    (16 s already at D = 4000); the sparse one must stay around a second and agree with it where the dense one is feasible."""
    import time
    from scd_amd import ops
    rs = np.random.RandomState(3)
    for k, per, d, limit in ((300, 4, 1500, None), (1000, 2, 10000, 20.0), (1000, 4, 20000, 30.0)):
        pop = rs.zipf(1.3, size=(k, per)) % min(d, 3 * k)
        rr, cc, vv = [], [], []
        for i in range(k):
            cols = np.unique(pop[i])
            rr += [i] * len(cols)
            cc += cols.tolist()
            vv += rs.randint(1, 200, size=len(cols)).tolist()
        t0 = time.time()
        sp = ops.munkres_sparse(d, rr, cc, vv)
        dt = time.time() - t0
        assert len(sp) == d and len(set(sp[:, 1].tolist())) == d            # a permutation
        if limit is None:
            w = np.zeros((d, d), dtype=np.int64)
            np.add.at(w, (rr, cc), vv)
            assert np.array_equal(sp, ops.munkres(w.max() - w))
        else:
            assert dt < limit, dt


def test_split_cluster_acc_v2_notebook_kat(golden):
    from scd_amd.gcd.project_utils.cluster_and_log_utils import split_cluster_acc_v2
    g = golden("acc_v2.npz")
    t, o, n, m = split_cluster_acc_v2(g["gt"], g["preds"], g["mask"], return_ind_map=True)
    assert (t, o, n) == (0.85, 0.8, 0.9) and m == {2: 0, 1: 1, 0: 2, 3: 3}


def test_assign_name_matches_reference(golden):
    from collections import Counter
    from scd_amd.local_utils import clip_lang_util as clu
    g = golden("naming.npz")
    for rep in range(3):
        c2c = {}
        for ck, nm, ct in zip(g["an%d_keys" % rep].tolist(), g["an%d_names" % rep].tolist(), g["an%d_counts" % rep].tolist()):
            c2c.setdefault(ck, Counter())[nm] = ct
        ind, w = clu.assign_name(g["an%d_voted" % rep].tolist(), c2c, num_common=3)
        assert np.array_equal(w, g["an%d_w" % rep]) and np.array_equal(ind, g["an%d_ind" % rep])
    assert len(clu.imagenet_templates) == 80


@pytest.mark.parametrize("n,k,smin,smax,seed", [(60, 4, 10, 20, 1), (300, 6, 40, 60, 2), (500, 10, 45, 55, 3), (400, 5, 0, 400, 4),
                                                (200, 8, 25, 25, 5), (1000, 12, 60, 120, 6), (37, 3, 1, 36, 7)])
def test_transport_optimal_and_feasible(n, k, smin, smax, seed):
    from scd_amd import ops
    rs = np.random.RandomState(seed)
    pts, cen = rs.randn(n, 3), rs.randn(k, 3) * 1.5
    d2 = ((pts[:, None] - cen[None]) ** 2).sum(-1).astype(np.float32)
    cost = to.int_costs(d2)
    lab, tot = ops.transport_solve(cost, smin, smax)
    _, tot_lp = to.solve_lp(cost, smin, smax)
    ok, tot_chk = to.check_assignment(cost, lab, smin, smax)
    assert ok and tot == tot_chk == tot_lp


def test_transport_batch_equals_single_solves():
    """scd_transport_solve_batch (the restarts' flow problems of one ConSSKM iteration on host threads, sskm_constrained.py:165-176):
    every problem's labels and total are scd_transport_solve's, whatever the thread count; an infeasible batch raises like one problem."""
    from scd_amd import ops
    rs = np.random.RandomState(9)
    b, n, k, smin, smax = 7, 900, 12, 50, 110
    pts = rs.randn(n, 4)
    costs = np.stack([to.int_costs(((pts[:, None] - rs.randn(k, 4)[None] * 1.5) ** 2).sum(-1).astype(np.float32)) for _ in range(b)])
    singles = [ops.transport_solve(costs[i], smin, smax) for i in range(b)]
    for threads in (1, 3, 16):
        labs, tots = ops.transport_solve_batch(costs, smin, smax, threads=threads)
        for i in range(b):
            assert np.array_equal(labs[i], singles[i][0]) and tots[i] == singles[i][1]
            ok, tot_chk = to.check_assignment(costs[i], labs[i], smin, smax)
            assert ok and tot_chk == tots[i] and to.check_optimal(costs[i], labs[i], smin, smax)
    with pytest.raises(Exception, match="There was an issue with the min cost flow input."):
        ops.transport_solve_batch(costs, 80, smax, threads=4)                 # 12 x 80 > 900


def test_transport_labels_equal_round5_solver():
    """The flow step's optimum is not unique, so which optimal labelling comes out is a property of the solver's tie-breaking rules
    (lowest index wins: nearest centre, cheapest member, next node of the search).  Round 6 rewrote the solver's loops for the host's
    vector units with the promise of the SAME augmentations in the same order: the labels of round 5's solver on five problems (two with
    costs quantised to multiples of 50, i.e. thousands of ties; one at k = 120) are kept in tests/golden/transport_labels_r5.npz
    (generated with round 5's library from the seeds below) and must come out again, bit for bit, single and batched
    (sskm_constrained.py:331-356)."""
    from scd_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "transport_labels_r5.npz"))
    cases = [(900, 12, 50, 110, 21, False), (2000, 40, 30, 80, 22, False), (1500, 20, 60, 90, 23, True), (3000, 120, 15, 60, 24, False),
             (700, 7, 100, 100, 25, True)]
    for i, (n, k, smin, smax, seed, tie) in enumerate(cases):
        rs = np.random.RandomState(seed)
        pts, cen = rs.randn(n, 6), rs.randn(k, 6) * 1.3
        cost = to.int_costs(((pts[:, None] - cen[None]) ** 2).sum(-1).astype(np.float32))
        if tie:
            cost = (cost // 50 * 50).astype(np.int32)
        assert list(g["bounds%d" % i]) == [smin, smax]
        lab, tot = ops.transport_solve(cost, smin, smax)
        assert tot == int(g["tot%d" % i]) and np.array_equal(lab, g["lab%d" % i]), i
        labs, tots = ops.transport_solve_batch(np.stack([cost, cost]), smin, smax, threads=2)
        assert np.array_equal(labs[0], lab) and np.array_equal(labs[1], lab) and tots[0] == tots[1] == tot
        assert to.check_optimal(cost, lab, smin, smax)


def test_first_index_cache_follows_the_list():
    """naming._first_index (the vote loops' `nouns.index(n)`, main_ptsup.py:664 / main_unsup.py:599): the dict is built once per vocabulary
    list, equals `.index` on duplicates, and a list changed in place gets a fresh one."""
    from scd_amd import naming
    nouns = ["a", "b", "c", "b", "d"]
    first = naming._first_index(nouns)
    assert all(first[n] == nouns.index(n) for n in nouns) and naming._first_index(nouns) is first
    nouns[0], nouns[4] = "d", "a"
    first2 = naming._first_index(nouns)
    assert first2 is not first and all(first2[n] == nouns.index(n) for n in nouns)
    assert naming._first_index(list(nouns)) == first2                      # another list object with the same names: its own entry


def test_hash_tokenizer_junction_rule():
    """The synthetic stand-in for the BPE merges pre-tokenises like the real tokenizer (letter runs, single digits, runs of other
    characters; '_' = space), so clip.tokenize_templates assembles prompts from pieces exactly where encode(x + y) == encode(x) +
    encode(y): checked on every pair of a junction alphabet, and the assembled 80-template prompt set equals prompt-by-prompt
    tokenisation (clip_lang_util.py:96-108)."""
    import scd_amd.clip as clip
    from scd_amd.local_utils.clip_lang_util import imagenet_templates
    tk = clip.HashTokenizer()
    chars = ["a", "Z", "7", ".", ",", " ", "_", "-", "'", "!", "é"]
    for a in chars:
        for b in chars:
            x, y = "xy" + a, b + "zw"
            if tk.separable(a, b):
                assert tk.encode(x + y) == tk.encode(x) + tk.encode(y), (a, b)
    old_tok, old_allow = clip._tokenizer, clip._allow_synthetic
    try:
        clip._tokenizer = tk
        names = ["name_%05d" % i for i in range(40)] + ["golden retriever", "x-ray", "o'neil", "a.b", "it's", "d_7", "42", "e."]
        a = clip.tokenize_templates(names, imagenet_templates).numpy()
        b = clip.tokenize([t.format(n) for n in names for t in imagenet_templates]).numpy()
        assert np.array_equal(a, b)
    finally:
        clip._tokenizer, clip._allow_synthetic = old_tok, old_allow


def test_transport_infeasible_raises_like_reference():
    from scd_amd import ops
    from scd_amd.local_utils.sskm_constrained import _labels_constrained
    with pytest.raises(Exception, match="There was an issue with the min cost flow input."):
        ops.transport_solve(np.ones((5, 2), dtype=np.int32), 3, 5)
    with pytest.raises(Exception, match="min cost flow"):
        _labels_constrained(None, None, np.ones((5, 2), dtype=np.float32), 0, 2, np.zeros(5, dtype=np.float32))


def test_labels_constrained_reference_golden(golden):
    from scd_amd.local_utils.sskm_constrained import _labels_constrained
    g = golden("kmeans_constrained.npz")
    dist = np.zeros(g["a_d2"].shape[0], dtype=np.float32)
    lab, inertia = _labels_constrained(None, None, np.sqrt(g["a_d2"]), 30, 80, dist)
    cost = to.int_costs(g["a_d2"])
    assert int(cost[np.arange(len(lab)), lab].sum()) == int(g["a_total"])          # same optimum as the reference run
    assert lab.dtype == np.int32 and abs(float(inertia) - float(g["a_inertia"])) <= 2e-3 * float(g["a_inertia"])


def test_c_oracle_matches_numpy_oracle():
    so = os.path.join(ROOT, "oracle", "c", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c")], check=True)
    lib = C.CDLL(so)
    P = lambda a: C.c_void_p(a.ctypes.data)
    rs = np.random.RandomState(0)
    x = rs.randn(700, 48).astype(np.float32)
    c = rs.randn(9, 48).astype(np.float32)
    lab = np.zeros(700, dtype=np.int64)
    mind = np.zeros(700, dtype=np.float32)
    lib.oracle_estep(P(x), P(c), C.c_int64(700), 48, 9, P(lab), P(mind))
    olab, omind, _ = ko.estep(x, c)
    assert np.array_equal(lab, olab) and np.allclose(mind, omind, rtol=1e-6)
    cen = np.zeros((9, 48), dtype=np.float32)
    lib.oracle_mstep(P(x), P(lab), C.c_int64(700), 48, 9, P(cen))
    oc, _ = ko.mstep(x, olab, 9)
    assert np.allclose(cen, oc, rtol=1e-6, atol=1e-7, equal_nan=True)
    f = (rs.randn(50, 64) / 8).astype(np.float16).astype(np.float32)
    w = (rs.randn(64, 333) / 8).astype(np.float16).astype(np.float32)
    wt = np.ascontiguousarray(w.T)
    idx = np.zeros((50, 5), dtype=np.int64)
    val = np.zeros((50, 5), dtype=np.float32)
    lib.oracle_sim_topk(P(f), P(wt), C.c_int64(50), 64, C.c_int64(333), C.c_double(100.0), 5, P(idx), P(val))
    oi, ov = no.sim_topk(f, w, 5, "raw")
    assert np.array_equal(idx, oi) and np.allclose(val, ov, rtol=1e-6)


def test_evaluate_semantic_acc_matches_reference_restatement():
    """scd_amd.naming.evaluate_semantic_acc (vectorised) against the loop of main_unsup.py:149-167 restated in the oracle,
    including two classes that share one name and clusters voted the same name."""
    from oracle import naming_oracle as no
    from scd_amd import naming
    rs = np.random.RandomState(5)
    n_cls, n_clu = 17, 11
    names = ["n%02d" % (i % 13) for i in range(n_cls)]              # classes 13..16 reuse names of 0..3
    cidx_to_cname = {c + 100: names[c] for c in range(n_cls)}        # class ids need not be 0..C-1
    cand = ["n%02d" % rs.randint(0, 15) for _ in range(n_clu)]
    t = rs.randint(0, n_cls, size=4000) + 100
    t = t[t != 105]                                                  # a class that never occurs
    p = rs.randint(0, n_clu, size=len(t))
    ref = no.evaluate_semantic_acc(t.astype(np.float64), cidx_to_cname, p, cand)
    got = naming.evaluate_semantic_acc(t.astype(np.float64), cidx_to_cname, p, cand)
    assert got[0] == pytest.approx(ref[0], rel=1e-12) and got[1] == pytest.approx(ref[1], rel=1e-12)


def test_soft_semantic_acc_memoised_equals_per_sample_loop():
    """naming.evaluate_soft_semantic_acc scores each distinct (predicted name, target name) pair once; the
    result must equal the reference's per-sample loop (main_unsup.py:170-199, restated here with a fake WordNet whose
    lch_similarity counts its calls)."""
    from scd_amd import naming
    calls = {"n": 0}

    class Syn:
        def __init__(self, depth):
            self.depth = depth

        def lch_similarity(self, other):
            calls["n"] += 1
            return 3.6 - 0.1 * abs(self.depth - other.depth) - 0.01 * min(self.depth, other.depth)
    rs = np.random.RandomState(9)
    names = ["n%02d" % i for i in range(30)]
    name_to_wnids = {n: ["w%s_%d" % (n, j) for j in range(1 + i % 3)] for i, n in enumerate(names)}
    wnid_to_synset = {wid: Syn(rs.randint(1, 20)) for ws in name_to_wnids.values() for wid in ws}
    cidx_to_cname = {float(c): names[c] for c in range(12)}
    cand = [names[rs.randint(30)] for _ in range(15)]
    t = rs.randint(0, 12, size=5000).astype(np.float64)
    p = rs.randint(0, 15, size=5000)
    # the reference's loop
    ref = []
    for ut, up in zip(t, p):
        pn, tn = cand[up], cidx_to_cname[ut]
        ref.append(max(wnid_to_synset[b].lch_similarity(wnid_to_synset[a]) for a in name_to_wnids[pn] for b in name_to_wnids[tn]))
    ref = np.array(ref) / max(ref)
    loop_calls, calls["n"] = calls["n"], 0
    cache = {}
    acc, scores = naming.evaluate_soft_semantic_acc(t, cidx_to_cname, p, cand, wnid_to_synset, name_to_wnids, return_score=True, cache=cache)
    assert acc == pytest.approx(ref.sum() / len(ref), rel=1e-12) and np.allclose(scores.astype(float), ref)
    assert calls["n"] < loop_calls / 10 and len(cache) <= 12 * 15
    calls["n"] = 0
    assert naming.evaluate_soft_semantic_acc(t[:100], cidx_to_cname, p[:100], cand, wnid_to_synset, name_to_wnids, cache=cache) > 0
    assert calls["n"] == 0                                     # a second call over the same pairs walks nothing


def test_bpe_tokenizer_against_independent_implementation(tmp_path, monkeypatch):
    """Rows a4 / N3: scd_amd.clip.SimpleTokenizer (byte-level BPE of the third-party `clip` package, absent here together with
    its 16e6 merges file) against the `tokenizers` library's BPE on the SAME merges table: a small table is learned on prompt-like
    text, written in the package's file format (gzip, header line, one merge per line), loaded by SimpleTokenizer, and the token
    ids of prompts with apostrophes, hyphens, digits and repeated words must be identical; clip.tokenize frames them with
    SOT / EOT and zero padding to 77."""
    import gzip
    import json
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Regex, Tokenizer, models, normalizers, pre_tokenizers, trainers
    import scd_amd.clip as clip
    words = ("red fox", "tabby cat", "kit fox", "zebra", "grey whale", "arctic fox", "sea lion", "golden retriever", "labrador retriever",
             "american black bear", "b-flat clarinet", "carpenter's kit", "soft-coated wheaten terrier", "4x4 truck", "don't")
    corpus = ["a photo of a %s." % w for w in words] * 3
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    alphabet = [chr(c) for c in cs]
    pat = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def make(model):
        tk = Tokenizer(model)
        tk.normalizer = normalizers.Lowercase()
        tk.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="removed", invert=True),
                                                    pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
        return tk
    learner = make(models.BPE(end_of_word_suffix="</w>"))
    learner.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=700, initial_alphabet=alphabet, end_of_word_suffix="</w>",
                                                            special_tokens=[], show_progress=False))
    merges = [tuple(m) for m in json.loads(learner.to_str())["model"]["merges"]]
    assert len(merges) > 50
    path = tmp_path / "bpe_simple_vocab_16e6.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    ours = clip.SimpleTokenizer(str(path))
    assert ours.sot == len(ours.encoder) - 2 and ours.eot == len(ours.encoder) - 1
    ref = make(models.BPE(vocab=dict(ours.encoder), merges=merges, end_of_word_suffix="</w>"))
    texts = ["a photo of a red fox.", "A bad photo of the Carpenter's kit!", "art of the b-flat clarinet.", "a 4x4 truck, don't", "the   origami zebra  ",
             "a photo of a soft-coated wheaten terrier.", "itap of a labrador retriever retriever.", "graffiti of a sea lion &amp; a kit fox"]
    for t in texts:
        import html
        want = ref.encode(" ".join(html.unescape(t).split()).strip()).ids
        assert ours.encode(t) == want, t
    monkeypatch.setattr(clip, "_tokenizer", ours)
    tok = clip.tokenize(texts[:2])
    assert tok.shape == (2, 77) and tok.dtype.is_floating_point is False
    ids = ours.encode(texts[0])
    assert tok[0, 0].item() == ours.sot and tok[0, 1:1 + len(ids)].tolist() == ids and tok[0, 1 + len(ids)].item() == ours.eot
    assert int(tok[0, 2 + len(ids):].abs().sum()) == 0
    with pytest.raises(RuntimeError, match="too long"):
        clip.tokenize(["fox " * 100])
    # the prompt set of zeroshot_classifier assembled from once-encoded pieces (tokenize_templates) == tokenize of every prompt,
    # for the BPE tokenizer and for the hash stand-in, including names / templates whose junctions are NOT separable
    from scd_amd.local_utils.clip_lang_util import imagenet_templates
    names = ["red fox", "b-flat clarinet", "carpenter's kit", "4x4", "c++", "st. bernard", "jack-o'-lantern", "name_with_underscore", "x&y",
             " lead", "trail ", "", "(paren)", "end.", "don't", "UPPER Case", "na\u00efve caf\u00e9", "\uff13\u3041", "a{b}", "it's", "'quoted'", "zebra" * 30]
    templates = list(imagenet_templates) + ["{}", "{} photo", "photo {}", "a {}'s toy", "({})", "x{}y", "a {}, a {{}}", "the {}&amp;co", "{}3", "3{}"]
    for tkz in (ours, clip.HashTokenizer()):
        monkeypatch.setattr(clip, "_tokenizer", tkz)
        want = clip.tokenize([t.format(c) for c in names for t in templates], truncate=True)
        got = clip.tokenize_templates(names, templates, truncate=True)
        assert got.shape == want.shape and torch.equal(got, want), type(tkz).__name__
    with pytest.raises(RuntimeError, match="too long"):
        clip.tokenize_templates(["fox " * 100], ["a photo of a {}."])


@pytest.mark.parametrize("mixed", [False, True])
def test_kpp_lockstep_equals_sequential_restarts(monkeypatch, mixed):
    """KMeansEngine draws the seedings of all n_init restarts in lock-step (kpp_lockstep) from one pre-drawn random stream; the
    reference runs kpp once per restart on the shared RandomState (sskm.py:190-204).  Same centres, labels, inertia, and the same
    position in the random stream afterwards."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_backend import OracleBackend
    from oracle import synth
    from scd_amd.kmeans import KMeansEngine
    x, y, mask_lab = synth.blob_case(500, 8, 7, 5)
    xt = torch.from_numpy(x)
    out = []
    for lock in ("1", "0"):
        monkeypatch.setenv("SCD_KPP_LOCKSTEP", lock)
        rs = np.random.RandomState(11)
        km = KMeansEngine(k=7, max_iterations=5, n_init=4, random_state=rs, backend=OracleBackend())
        if mixed:
            km.fit_mix(xt[~mask_lab], xt[mask_lab], torch.from_numpy(y[mask_lab]))
        else:
            km.fit(xt)
        out.append((km.labels_.numpy(), km.cluster_centers_.numpy(), float(km.inertia_), rs.rand()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]


def test_host_solvers_under_sanitizers(tmp_path):
    """munkres.cpp, munkres_sparse.cpp and transport.cpp compiled for the host with clang's AddressSanitizer + UndefinedBehaviorSanitizer
    and driven by tests/sanitize_host.cpp: random assignment problems against brute force, sparse = dense on vote-shaped matrices,
    transport labels inside their bounds, the threaded batch solver = the single solves (also under ThreadSanitizer) - and no sanitizer report (cluster_utils.py:234-493, clip_lang_util.py:167-178,
    sskm_constrained.py:277-356).  The GPU sanitizers are not available on this pool; the host code is what can be covered."""
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("clang++")
    if not clang:
        pytest.skip("no clang++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "scd_amd", "csrc")
    exe = str(tmp_path / "sanitize_host")
    cmd = [clang, "-x", "c++", "-std=c++17", "-O1", "-g", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(root, "include"), "-I", src,
           os.path.join(src, "munkres.cpp"), os.path.join(src, "munkres_sparse.cpp"), os.path.join(src, "transport.cpp"),
           os.path.join(root, "tests", "sanitize_host.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr
    # the same driver under ThreadSanitizer: scd_transport_solve_batch runs its problems on host threads (round 5)
    exe_t = str(tmp_path / "sanitize_host_tsan")
    cmd_t = [c for c in cmd if not c.startswith("-fsanitize=") and c != "-fno-sanitize-recover=undefined"]
    cmd_t[cmd_t.index(exe)] = exe_t
    cmd_t.insert(1, "-fsanitize=thread")
    subprocess.run(cmd_t, check=True, timeout=600)
    r = subprocess.run([exe_t], capture_output=True, text=True, timeout=600)
    if "FATAL: ThreadSanitizer" in r.stderr and "0 failures" not in r.stdout:      # (a kernel that refuses TSan's address-space layout)
        pytest.skip("ThreadSanitizer cannot run here: " + r.stderr.strip().splitlines()[0])
    assert r.returncode == 0 and "0 failures" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_double_double_add_of_the_sharded_sumsq():
    """ops._dd_add (the host-side sum of the ranks' double-double sums of squares, once per sharded fit): hi + lo reproduces the exact
    rational sum of the four parts to ~2^-100 where a plain float64 sum of the high parts loses the low ones."""
    from fractions import Fraction
    from scd_amd.ops import _dd_add
    rs = np.random.RandomState(4)
    acc, exact = (0.0, 0.0), Fraction(0)
    for _ in range(64):
        hi = float(rs.rand() * 10.0 ** rs.randint(-3, 9))
        lo = float((rs.rand() - 0.5) * hi * 2.0 ** -53)
        acc = _dd_add(acc, (hi, lo))
        exact += Fraction(hi) + Fraction(lo)
    got = Fraction(acc[0]) + Fraction(acc[1])
    assert abs(got - exact) <= abs(exact) * Fraction(1, 2 ** 96)
    assert abs(acc[1]) <= abs(acc[0]) * 2.0 ** -52


def test_bench_power_sampler_reads_hwmon(tmp_path, monkeypatch):
    """bench.PowerSampler: the rank's GPU is found through its PCI address, watts / cap / shader clock come from the hwmon files, and a
    box without them yields None (the bench line then carries "board_power": null)."""
    import importlib.util
    import time
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hw = tmp_path / "hwmon3"
    hw.mkdir()
    (hw / "power1_input").write_text("1388000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "freq1_input").write_text("1837000000\n")
    import glob as globmod
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x5a, pci_device_id=0))
    real_glob = globmod.glob
    monkeypatch.setattr(globmod, "glob", lambda pat, **kw: [str(hw)] if pat.startswith("/sys/bus/pci/devices/0000:5a:00.0/hwmon") else real_glob(pat, **kw))
    ps = bench.PowerSampler(0)
    ps.start()
    time.sleep(0.6)
    out = ps.stop()
    assert out and out["median_w"] == 1388.0 and out["cap_w"] == 1400.0 and out["sclk_mhz_median"] == 1837.0 and out["samples"] >= 2
    monkeypatch.setattr(globmod, "glob", lambda pat, **kw: [])
    ps = bench.PowerSampler(0)
    ps.start()
    assert ps.stop() is None
