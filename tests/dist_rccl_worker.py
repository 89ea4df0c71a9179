"""One rank of the multi-GPU check (tests/test_gpu_parity.py::test_multi_rank_rccl): started N times by torch.distributed.run,
one process per GPU, "nccl" (= RCCL) backend, every kernel through libscd_hip.so.  Each rank holds a contiguous, uneven row shard;
the sharded results must equal what the same rank computes alone from the full data:
  * sharded SSKM fit_mix (lock-step seeding over three all-gathers per round, one packed all-reduce per Lloyd iteration; both loops in
    C with the collectives as callbacks: scd_kpp_seed_lockstep_sharded, scd_kmeans_lloyd_run_sharded),
  * the sharded vote loop,
  * the C entry points scd_comm_init / scd_allreduce_centroids / scd_allgather_text over RCCL.
Prints "rank R ok" and exits 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard(n, rank, world):
    """contiguous, deliberately uneven: rank r owns [cut[r], cut[r+1])"""
    w = np.arange(1, world + 1, dtype=np.float64) + 2.0
    cut = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * n).astype(int)])
    return slice(int(cut[rank]), int(cut[rank + 1]))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", local % torch.cuda.device_count())
    backend = os.environ.get("SCD_TEST_BACKEND", "nccl")       # "gloo": several ranks sharing ONE GPU (RCCL refuses a duplicate device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from oracle import synth, naming_oracle as no
    from scd_amd import ops, naming, pipeline
    from scd_amd.kmeans import KMeansEngine
    grp = dist.group.WORLD
    # ---- (1) sharded SSKM == the same fit on one rank
    n, d, k = 6000, 64, 12
    x, y, mask_lab = synth.blob_case(n, d, k, 21)
    u, l, lt = x[~mask_lab], x[mask_lab], y[mask_lab]
    su, sl = shard(len(u), rank, world), shard(len(l), rank, world)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    one = KMeansEngine(k=k, max_iterations=6, n_init=3, random_state=3)
    one.fit_mix(T(u), T(l), T(lt))
    shd = KMeansEngine(k=k, max_iterations=6, n_init=3, random_state=3, group=grp)
    shd.fit_mix(T(u[su]), T(l[sl]), T(lt[sl]))
    n_l, n_ls = len(lt), sl.stop - sl.start
    full = one.labels_.cpu().numpy()
    mine = shd.labels_.cpu().numpy()
    assert np.array_equal(mine[:n_ls], full[:n_l][sl]), "labelled rows"
    assert np.array_equal(mine[n_ls:], full[n_l:][su]), "unlabelled rows: sharded labels differ from the single-rank fit"
    # 1-GPU and N-GPU centres: the same float64 sums in a different order
    assert np.allclose(shd.cluster_centers_.cpu().numpy(), one.cluster_centers_.cpu().numpy(), rtol=1e-6, atol=1e-7)
    cen = shd.cluster_centers_.contiguous()
    ref = cen.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(cen, ref), "centres must be bit-identical on every rank"
    # ---- (1b) fp16-exact rows (what the encoders deliver): the sharded fit runs its Lloyd loops behind scd_kmeans_lloyd_run_sharded (the
    # packed all-reduce handed in as a callback) and is BIT-identical to the single-rank fit - exact sums do not depend on the sharding
    x16 = x.astype(np.float16).astype(np.float32)
    u, l = x16[~mask_lab], x16[mask_lab]
    one = KMeansEngine(k=k, max_iterations=8, n_init=3, random_state=5)
    one.fit_mix(T(u), T(l), T(lt))
    shd = KMeansEngine(k=k, max_iterations=8, n_init=3, random_state=5, group=grp)
    shd.fit_mix(T(u[su]), T(l[sl]), T(lt[sl]))
    assert shd.stats.get("sharded_runs", 0) == 3, "the sharded C loop did not run: %r" % (shd.stats,)
    # the three restarts advance in lock-step behind ONE call (scd_kmeans_lloyd_run_multi) and share one all-reduce per iteration:
    # at most max_iterations exchanges for the whole fit (3 x 8 with one loop per restart)
    assert shd.stats.get("lockstep_fits", 0) == 1 and 1 <= shd.stats.get("lloyd_exchanges", 0) <= 8, shd.stats
    assert shd.stats.get("sharded_seedings", 0) == 1, "the sharded seeding rounds did not run behind scd_kpp_seed_lockstep_sharded: %r" % (shd.stats,)
    full, mine = one.labels_.cpu().numpy(), shd.labels_.cpu().numpy()
    assert np.array_equal(mine[:n_ls], full[:n_l][sl]) and np.array_equal(mine[n_ls:], full[n_l:][su]), "exact rows: sharded labels differ"
    assert torch.equal(shd.cluster_centers_, one.cluster_centers_), "exact rows: centres must be bit-identical to the single-rank fit"
    # (n_iter_ of fit_mix is the reference's stale labelled-row index, sskm_constrained.py:104,139 - a per-shard number by construction)
    assert float(shd.inertia_) == float(one.inertia_), (float(shd.inertia_), float(one.inertia_))
    # the unsupervised fit (no labelled rows) through the same loop
    one = KMeansEngine(k=k, max_iterations=8, n_init=2, random_state=6)
    one.fit(T(u))
    shd = KMeansEngine(k=k, max_iterations=8, n_init=2, random_state=6, group=grp)
    shd.fit(T(u[su]))
    assert shd.stats.get("sharded_runs", 0) == 2
    assert np.array_equal(shd.labels_.cpu().numpy(), one.labels_.cpu().numpy()[su]) and torch.equal(shd.cluster_centers_, one.cluster_centers_)
    assert float(shd.inertia_) == float(one.inertia_) and shd.n_iter_ == one.n_iter_, (float(shd.inertia_), float(one.inertia_), shd.n_iter_, one.n_iter_)
    # a restart that EMPTIES clusters (the unlabelled rows of tests/golden/kmeans_f16.npz case "h": its second restart does so in its second
    # iteration and still wins): the reference's NaN arithmetic ends it there (faster_mix_k_means_pytorch.py:140-160) - the ranks take that
    # decision from the exchanged counts, stop together, keep the NaN centre rows and report max_iterations, as the single-rank fit
    xh, yh, mh = synth.blob_case(6000, 512, 30, 5)
    uh = xh.astype(np.float16).astype(np.float32)[~mh]
    sh = shard(len(uh), rank, world)
    one = KMeansEngine(k=30, max_iterations=10, n_init=2, random_state=6)
    one.fit(T(uh))
    shd = KMeansEngine(k=30, max_iterations=10, n_init=2, random_state=6, group=grp)
    shd.fit(T(uh[sh]))
    assert bool(torch.isnan(one.cluster_centers_).any()) and one.n_iter_ == 10, "the case no longer empties a cluster"
    assert np.array_equal(shd.labels_.cpu().numpy(), one.labels_.cpu().numpy()[sh]), "emptied clusters: sharded labels differ"
    assert np.array_equal(shd.cluster_centers_.cpu().numpy(), one.cluster_centers_.cpu().numpy(), equal_nan=True)
    assert float(shd.inertia_) == float(one.inertia_) and shd.n_iter_ == one.n_iter_ == 10
    # restarts that converge at DIFFERENT iterations (a loose tolerance, eight restarts over the library's four streams): the running
    # restarts move up in the densely packed exchange buffer when one drops out - round 5 found a stream race there (a restart packed into a
    # region another stream's finalize was still reading; the ranks then disagreed about who was still running and the collective sizes
    # diverged).  Bit-identical to the single-rank fit, as above
    one = KMeansEngine(k=k, tolerance=5e-2, max_iterations=8, n_init=8, random_state=7)
    one.fit_mix(T(u), T(l), T(lt))
    os.environ["SCD_LLOYD_STREAMS"] = "4"          # (under a group the default is one stream)
    shd = KMeansEngine(k=k, tolerance=5e-2, max_iterations=8, n_init=8, random_state=7, group=grp)
    shd.fit_mix(T(u[su]), T(l[sl]), T(lt[sl]))
    del os.environ["SCD_LLOYD_STREAMS"]
    assert shd.stats.get("lockstep_fits", 0) == 1 and shd.stats.get("lloyd_exchanges", 0) <= 8, shd.stats
    full, mine = one.labels_.cpu().numpy(), shd.labels_.cpu().numpy()
    assert np.array_equal(mine[:n_ls], full[:n_l][sl]) and np.array_equal(mine[n_ls:], full[n_l:][su]), "restarts dropping out: sharded labels differ"
    assert torch.equal(shd.cluster_centers_, one.cluster_centers_) and float(shd.inertia_) == float(one.inertia_)
    # ---- (2) sharded vote loop == single-rank vote loop
    nv, dv, kv, vv = 4800, 512, 12, 2100
    xv, yv, cv = synth.clustered_features(nv, dv, kv, seed=31, center_seed=32, noise=0.9)
    w = synth.vocabulary(vv, dv, cv, seed=33, jitter=0.5)
    nouns = synth.nouns_list(vv)
    f = T(xv.astype(np.float16))
    wt = T(np.ascontiguousarray(w.T).astype(np.float16))
    idx, _ = ops.sim_topk(f, wt, 5, "softmax")
    rs = np.random.RandomState(35)
    preds0 = T(np.where(rs.rand(nv) < 0.8, (yv * 7 + 2) % kv, rs.randint(0, kv, size=nv)).astype(np.int64))
    cand1, up1, tr1 = naming.vote_loop_unsup(idx, preds0, f, wt, nouns, kv, 10, 2, max_iter=50)
    sv = shard(nv, rank, world)
    for exchange in ("table", "rows"):          # the dense-table exchange of SURVEY 8e and the rows exchange "auto" picks at these sizes
        cand2, up2, tr2 = pipeline.vote_loop_unsup_sharded(idx[sv], preds0[sv], f[sv], wt, nouns, kv, 10, 2, grp, max_iter=50, exchange=exchange)
        assert cand1 == cand2 and len(tr1) == len(tr2), "sharded vote loop (%s): candidate names differ" % exchange
        assert np.array_equal(np.asarray(up1)[sv], np.asarray(up2)), "sharded vote loop (%s): re-classified rows differ" % exchange
        for a, b in zip(tr1, tr2):
            assert np.array_equal(a["voted"], b["voted"]) and np.array_equal(np.asarray(a["ind"]), np.asarray(b["ind"])), exchange
    if backend != "nccl":
        torch.cuda.synchronize()
        dist.barrier()
        print("rank %d ok" % rank, flush=True)
        dist.destroy_process_group()
        return
    # ---- (3) the C entry points over RCCL
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid = torch.frombuffer(bytearray(ops.Comm.unique_id()), dtype=torch.uint8).to(dev)
    dist.broadcast(uid, 0)
    comm = ops.Comm(rank, world, bytes(uid.cpu().numpy().tobytes()))
    try:
        xk, yk, ck = synth.clustered_features(5000, 64, 7, seed=5)
        sk = shard(5000, rank, world)
        sums, counts, inertia = ops.kmeans_mstep(T(xk[sk]), T(yk[sk].astype(np.int32)), T(ck), 7, 0)
        s2, c2, i2 = comm.allreduce_centroids(sums, counts, inertia)
        fs, fc, fi = ops.kmeans_mstep(T(xk), T(yk.astype(np.int32)), T(ck), 7, 0)
        assert torch.equal(c2, fc) and torch.allclose(s2, fs, rtol=1e-12, atol=1e-12) and torch.allclose(i2, fi, rtol=1e-12)
        rows = 8 * world
        wfull = np.random.RandomState(0).randn(rows, 512).astype(np.float16)
        got = comm.allgather_text(T(wfull[8 * rank: 8 * rank + 8]))
        assert torch.equal(got, T(wfull)), "scd_allgather_text: rank order"
    finally:
        comm.close()
    torch.cuda.synchronize()
    dist.barrier()
    print("rank %d ok" % rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
