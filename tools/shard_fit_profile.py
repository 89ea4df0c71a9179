"""Where a sharded SSKM fit's wall time goes when the ranks share one GPU over gloo (the N = 2 rehearsal): number of collectives and
the wall time spent inside them, against the fit's wall time.  Run:
  SCD_TEST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/shard_fit_profile.py
Rank 0 prints one line per mode (collectives timed as issued / with a device synchronisation before and after each)."""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(os.environ.get("SCD_TEST_BACKEND", "gloo"), rank=rank, world_size=world)
    from scd_amd.kmeans import KMeansEngine
    n, d, k = 126976, 512, 100                       # the bench's per-rank shard: 95k unlabelled + 32k labelled rows
    g = torch.Generator(device=dev).manual_seed(11)
    cen = torch.nn.functional.normalize(torch.randn(k, d, device=dev, generator=g), dim=-1)
    g2 = torch.Generator(device=dev).manual_seed(100 + rank)
    y = torch.randint(0, k, (n,), device=dev, generator=g2)
    x = torch.nn.functional.normalize(cen[y] + (0.9 / d ** 0.5) * torch.randn(n, d, device=dev, generator=g2), dim=-1).half().float()
    lab = (y < k // 2) & (torch.rand(n, device=dev, generator=g2) < 0.5)
    u, l, lt = x[~lab].contiguous(), x[lab].contiguous(), y[lab].contiguous()
    stat = {"n": 0, "t": 0.0, "sync": False}
    for name in ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast"):
        orig = getattr(dist, name)

        def wrapped(*a, _o=orig, **kw):
            if stat["sync"]:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = _o(*a, **kw)
            if stat["sync"]:
                torch.cuda.synchronize()
            stat["t"] += time.perf_counter() - t0
            stat["n"] += 1
            return r
        setattr(dist, name, wrapped)
    for mode in ("warm-up", "as issued", "synchronised around every collective"):
        stat.update(n=0, t=0.0, sync=mode.startswith("sync"))
        km = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0, group=dist.group.WORLD)
        torch.cuda.synchronize(); dist.barrier()
        stat.update(n=0, t=0.0)
        t0 = time.perf_counter()
        km.fit_mix(u, l, lt)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if rank == 0 and mode != "warm-up":
            print("%-40s fit %.1f ms, %d collectives, %.1f ms inside them (%.3f ms each), sharded C loops: %d"
                  % (mode, wall * 1e3, stat["n"], stat["t"] * 1e3, stat["t"] * 1e3 / max(stat["n"], 1), km.stats.get("sharded_runs", 0)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
