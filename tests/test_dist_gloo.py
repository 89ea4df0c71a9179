"""world_size=2 gloo runs of the multi-GPU k-means logic on CPU (one process per 'GPU', sharded rows).  The compute
backend is the oracle-backed stand-in of tests/oracle_backend.py; what is under test is the exchange pattern of
scd_amd/kmeans.py: one packed all-reduce per Lloyd iteration, shard-aware k-means++ draws, gathered constrained E-step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import synth
        from oracle_backend import OracleBackend
        from scd_amd.kmeans import KMeansEngine, ConstrainedEngine
        n, d, k, seed = 900, 16, 6, 21
        x, y, mask_lab = synth.blob_case(n, d, k, seed)
        u, l, lt = x[~mask_lab], x[mask_lab], y[mask_lab]
        cut_u, cut_l = 400, 100                                     # uneven shards
        su = slice(0, cut_u) if rank == 0 else slice(cut_u, None)
        sl = slice(0, cut_l) if rank == 0 else slice(cut_l, None)
        if kind == "fit":                                            # no labelled rows: first centres are fetched across shards
            km = KMeansEngine(k=k, max_iterations=5, n_init=3, random_state=7, backend=OracleBackend(), group=dist.group.WORLD)
            sx = slice(0, 500) if rank == 0 else slice(500, None)
            km.fit(torch.from_numpy(x[sx]))
            q.put((rank, km.labels_.numpy(), km.cluster_centers_.numpy(), float(km.inertia_)))
            return
        if kind == "empty":                                          # rank 1 owns no unlabelled row: BOTH ranks must raise, nobody may hang
            km = KMeansEngine(k=k, max_iterations=3, n_init=2, random_state=3, backend=OracleBackend(), group=dist.group.WORLD)
            su = slice(0, len(u)) if rank == 0 else slice(0, 0)
            try:
                km.fit_mix(torch.from_numpy(u[su]), torch.from_numpy(l[sl]), torch.from_numpy(lt[sl]))
                q.put((rank, "no error", None, 0.0))
            except ValueError as e:
                q.put((rank, "ValueError: " + str(e), None, 0.0))
            return
        if kind == "sskm":
            km = KMeansEngine(k=k, max_iterations=6, n_init=2, random_state=3, backend=OracleBackend(), group=dist.group.WORLD)
        else:
            km = ConstrainedEngine(k=k, max_iterations=4, size_min=60, size_max=200, n_init=2, random_state=3,
                                   backend=OracleBackend(), group=dist.group.WORLD)
        km.fit_mix(torch.from_numpy(u[su]), torch.from_numpy(l[sl]), torch.from_numpy(lt[sl]))
        q.put((rank, km.labels_.numpy(), km.cluster_centers_.numpy(), float(km.inertia_)))
    finally:
        dist.destroy_process_group()


def _run(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, lab, cen, inertia = q.get(timeout=300)
        res[r] = (lab, cen, inertia)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_sskm_equals_single_process():
    from oracle import kmeans_oracle as ko, synth
    res = _run("sskm")
    n, d, k, seed = 900, 16, 6, 21
    x, y, mask_lab = synth.blob_case(n, d, k, seed)
    u, l, lt = x[~mask_lab], x[mask_lab], y[mask_lab]
    okm = ko.K_Means(k=k, max_iterations=6, n_init=2, random_state=3)
    okm.fit_mix(u, l, lt)
    n_l = len(lt)
    # each rank holds [its labelled rows ; its unlabelled rows]
    lab_l = np.concatenate([res[0][0][:100], res[1][0][: n_l - 100]])
    lab_u = np.concatenate([res[0][0][100:], res[1][0][n_l - 100:]])
    assert np.array_equal(lab_l, okm.labels_[:n_l]) and np.array_equal(lab_u, okm.labels_[n_l:])
    assert np.array_equal(res[0][1], res[1][1])                      # identical centroids on every rank
    assert np.allclose(res[0][1], okm.cluster_centers_, rtol=1e-6, atol=1e-7)
    assert res[0][2] == res[1][2] == pytest.approx(float(okm.inertia_), rel=1e-6)


def test_sharded_fit_equals_single_process():
    """K_Means.fit over two row shards (lock-step seeding with the first centres fetched from their owner ranks, three
    all-gathers per round for all restarts, one all-reduce per Lloyd iteration) == the single-process oracle run."""
    from oracle import kmeans_oracle as ko, synth
    res = _run("fit")
    x, _, _ = synth.blob_case(900, 16, 6, 21)
    okm = ko.K_Means(k=6, max_iterations=5, n_init=3, random_state=7)
    okm.fit(x)
    lab = np.concatenate([res[0][0], res[1][0]])
    assert np.array_equal(lab, okm.labels_)
    assert np.array_equal(res[0][1], res[1][1])
    assert np.allclose(res[0][1], okm.cluster_centers_, rtol=1e-6, atol=1e-7)
    assert res[0][2] == res[1][2] == pytest.approx(float(okm.inertia_), rel=1e-6)


def test_sharded_constrained_respects_global_bounds():
    res = _run("con")
    from oracle import synth
    _, y, mask_lab = synth.blob_case(900, 16, 6, 21)
    n_l = int(mask_lab.sum())
    lab_u = np.concatenate([res[0][0][100:], res[1][0][n_l - 100:]])
    cnt = np.bincount(lab_u, minlength=6)
    assert cnt.min() >= 60 and cnt.max() <= 200 and cnt.sum() == 900 - n_l
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]


def test_empty_shard_raises_on_every_rank():
    """A rank whose shard has no row to cluster: the precondition is agreed with one all-reduce before any rank-local check, so
    every rank raises the same ValueError instead of one rank raising while its peers wait inside the next collective."""
    res = _run("empty")
    assert res[0][0] == res[1][0] and res[0][0].startswith("ValueError: a rank of the process group owns no row")


def _vocab_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scd_amd.local_utils import clip_lang_util as clu
        names = ["name_%03d" % i for i in range(37)]                 # 37 names over 2 ranks: shards of 19 and 18
        w = clu.zeroshot_classifier_sharded(names, ["a {}.", "the {}."], None, dist.group.WORLD, build=_stub_build)
        q.put((rank, w.numpy()))
    finally:
        dist.destroy_process_group()


def _stub_build(names, templates, model, names_per_batch):
    """CPU stand-in for zeroshot_classifier: a deterministic unit column per name, [8, n]."""
    cols = []
    for nme in names:
        rs = np.random.RandomState(int(nme.split("_")[1]) + 1000 * len(templates))
        v = rs.randn(8).astype(np.float32)
        cols.append(v / np.linalg.norm(v))
    return torch.from_numpy(np.stack(cols, axis=1)).to(torch.float16)


def test_sharded_vocabulary_allgather_keeps_name_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_vocab_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = ["name_%03d" % i for i in range(37)]
    ref = _stub_build(names, ["a {}.", "the {}."], None, 16).numpy()
    assert res[0].shape == (8, 37) and np.array_equal(res[0], ref) and np.array_equal(res[1], ref)


def _vote_worker(rank, world, port, q, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleNamingOps
        from scd_amd import pipeline
        f, w, nouns, idx, preds0, k = _vote_case()
        cut = 700                                                    # uneven row shards
        sl = slice(0, cut) if rank == 0 else slice(cut, None)
        cand, up, tr = pipeline.vote_loop_unsup_sharded(torch.from_numpy(idx[sl]), torch.from_numpy(preds0[sl]), torch.from_numpy(f[sl]),
                                                        torch.from_numpy(np.ascontiguousarray(w.T)), nouns, k, 10, 2, dist.group.WORLD,
                                                        be=OracleNamingOps(), exchange=exchange)
        q.put((rank, cand, up, [(t["voted"], t["ind"], t["cand"]) for t in tr]))
    finally:
        dist.destroy_process_group()


def _vote_case():
    from oracle import naming_oracle as no, synth
    n, d, k, v = 1200, 64, 12, 400
    x, y, cent = synth.clustered_features(n, d, k, seed=31, center_seed=32, noise=0.9)
    w = synth.vocabulary(v, d, cent, seed=33, jitter=0.5)
    f = x.astype(np.float16)
    idx, _ = no.sim_topk(f, w, 5, "softmax")
    rs = np.random.RandomState(35)
    preds0 = np.where(rs.rand(n) < 0.8, (y * 7 + 2) % k, rs.randint(0, k, size=n))
    return f, w, synth.nouns_list(v), idx, preds0, k


@pytest.mark.parametrize("exchange", ["table", "rows", "auto"])
def test_sharded_vote_loop_equals_single_process(exchange):
    """pipeline.vote_loop_unsup_sharded over two row shards = the single-process oracle loop (main_unsup.py:568-614): same voted
    lists, assignments and candidate names on every iteration, and each rank's re-classified rows are its slice of the global
    predictions - with the dense [clusters, V] table exchange and with the rows exchange (names gathered once, cluster ids
    per iteration) that "auto" picks at these sizes."""
    from oracle import naming_oracle as no
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + {"table": 0, "rows": 2000, "auto": 4000}[exchange]
    procs = [ctx.Process(target=_vote_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, cand, up, tr = q.get(timeout=300)
        res[r] = (cand, up, tr)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f, w, nouns, idx, preds0, k = _vote_case()
    otr = no.vote_loop_unsup(idx, preds0, f, w, nouns, k, 5, 10, 2)
    assert res[0][0] == res[1][0] == [nouns[c] for c in otr[-1]["cand"]]
    assert len(res[0][2]) == len(res[1][2]) == len(otr)
    for r in (0, 1):
        for (voted, ind, cand), o in zip(res[r][2], otr):
            assert np.array_equal(voted, o["voted"]) and np.array_equal(ind, o["ind"]) and np.array_equal(cand, o["cand"])
    assert np.array_equal(np.concatenate([res[0][1], res[1][1]]), otr[-1]["u_preds"])
