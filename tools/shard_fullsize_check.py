"""Full-size check of the sharded SSKM fit (BASELINE configs[1] shape: 95,232 unlabelled + 31,744 labelled rows x 512, K = 100): every rank
computes the single-process fit on ALL rows and the sharded fit on its shard (uneven shards); labels, centres and inertia must be
bit-identical (fp16-exact rows: exact sums do not depend on the sharding).  Ranks may share one GPU over gloo:
  SCD_TEST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/shard_fullsize_check.py"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nd = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % nd)
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group(os.environ.get("SCD_TEST_BACKEND", "nccl"), rank=rank, world_size=world)
    from scd_amd.kmeans import KMeansEngine
    n, d, k = 126976, 512, 100
    g = torch.Generator(device=dev).manual_seed(5)                      # the same data on every rank
    cen = torch.nn.functional.normalize(torch.randn(k, d, device=dev, generator=g), dim=-1)
    y = torch.randint(0, k, (n,), device=dev, generator=g)
    x = torch.nn.functional.normalize(cen[y] + (0.9 / d ** 0.5) * torch.randn(n, d, device=dev, generator=g), dim=-1).half().float()
    lab = (y < k // 2) & (torch.rand(n, device=dev, generator=g) < 0.5)
    u, l, lt = x[~lab].contiguous(), x[lab].contiguous(), y[lab].contiguous()
    w = np.arange(1, world + 1, dtype=np.float64) + 2.0
    cut = lambda m: np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * m).astype(int)])
    cu, cl = cut(len(u)), cut(len(l))
    su, sl = slice(int(cu[rank]), int(cu[rank + 1])), slice(int(cl[rank]), int(cl[rank + 1]))
    one = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0)
    t0 = time.perf_counter(); one.fit_mix(u, l, lt); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
    shd = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0, group=dist.group.WORLD)
    dist.barrier()
    t0 = time.perf_counter(); shd.fit_mix(u[su].contiguous(), l[sl].contiguous(), lt[sl].contiguous()); torch.cuda.synchronize(); t_shd = time.perf_counter() - t0
    full, mine = one.labels_.cpu().numpy(), shd.labels_.cpu().numpy()
    n_l, n_ls = len(lt), sl.stop - sl.start
    ok = (np.array_equal(mine[:n_ls], full[:n_l][sl]) and np.array_equal(mine[n_ls:], full[n_l:][su]) and
          torch.equal(shd.cluster_centers_, one.cluster_centers_) and float(shd.inertia_) == float(one.inertia_))
    print("rank %d of %d: %d + %d rows of %d + %d; single-process fit %.1f ms, sharded fit %.1f ms; sharded C loops %d, C seedings %d; labels / centres / "
          "inertia bit-identical: %s" % (rank, world, su.stop - su.start, n_ls, len(u), n_l, t_one * 1e3, t_shd * 1e3, shd.stats.get("sharded_runs", 0),
                                         shd.stats.get("sharded_seedings", 0), ok), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
