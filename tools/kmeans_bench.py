"""Microbenchmark of the k-means kernels at BASELINE C2 size (N_u=95k, D=768, K=100), HIP events, clustered data."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd import ops


def clustered_features(n, d, k, seed, center_seed, noise):
    """k Gaussian blobs (unit-norm random centres, isotropic noise scaled so that neighbouring blobs overlap a little)."""
    rs = np.random.RandomState(center_seed)
    cent = rs.randn(k, d).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    rs = np.random.RandomState(seed)
    y = rs.randint(0, k, size=n)
    x = cent[y] + (noise / np.sqrt(d)) * rs.randn(n, d).astype(np.float32)
    return x.astype(np.float32), y, cent

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

if __name__ == "__main__":
    n, d, k = int(sys.argv[3]) if len(sys.argv) > 3 else 95000, int(sys.argv[1]) if len(sys.argv) > 1 else 768, int(sys.argv[4]) if len(sys.argv) > 4 else 100
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
    x, y, cent = clustered_features(n, d, k, seed=21, center_seed=22, noise=noise)
    X = torch.from_numpy(x).cuda(); C = torch.from_numpy(cent).cuda()
    C2 = X[torch.randperm(n, device="cuda")[:k]].clone()          # k-means++-like start: data points as centres
    data = ops.KMeansData(X)
    for name, cc in (("converged centres", C), ("data-point centres", C2)):
        lab, ref = data.estep(cc, return_refined=True)
        few = name.startswith("converged")
        us = timeit(lambda: data.estep(cc, expect_few=few))
        dp = (d + 127) // 128 * 128
        algo = n * dp * 2 + 4 * n + ((k + 127) // 128 * 128) * dp * 2
        print("estep  [%s] %8.1f us  refined rows %6d (%.2f%%)  algorithmic %.1f MB -> %.0f GB/s" % (name, us, int(ref), 100.0 * int(ref) / n, algo / 1e6, algo / us / 1e3))
    lab = data.estep(C)
    us = timeit(lambda: ops.kmeans_mstep(X, lab, C, k, 0))
    algo = n * d * 4 + 4 * n
    print("mstep  %8.1f us  algorithmic %.1f MB -> %.0f GB/s" % (us, algo / 1e6, algo / us / 1e3))
    X16 = ops.f16_exact(X.half().float())
    Xe = X.half().float().contiguous()
    us = timeit(lambda: ops.kmeans_mstep(Xe, lab, C, k, 0, x16=X16))
    algo = n * d * 2 + 4 * n
    print("mstep on the exact fp16 copy %8.1f us  algorithmic %.1f MB -> %.0f GB/s" % (us, algo / 1e6, algo / us / 1e3))
    d2 = data.rowdist(C, lab)
    us = timeit(lambda: data.min_update(X[17], d2))
    algo = n * d * 4 + 8 * n
    print("minupd %8.1f us  algorithmic %.1f MB -> %.0f GB/s" % (us, algo / 1e6, algo / us / 1e3))
    us = timeit(lambda: ops.kpp_draw(d2, 0.37))
    print("draw   %8.1f us" % us)
    R = 10
    d2m = d2.reshape(1, -1).expand(R, -1).contiguous()
    cm = X[17:17 + R].contiguous()
    us = timeit(lambda: ops.min_update_multi(X, cm, d2m))
    print("minupd for %d restarts in lock-step %8.1f us  (%.1f us per restart), X read once: %.1f MB -> %.0f GB/s"
          % (R, us, us / R, n * d * 4 / 1e6, n * d * 4 / us / 1e3))
    rv = np.linspace(0.05, 0.95, R)
    us = timeit(lambda: ops.kpp_draw_multi(d2m, rv))
    print("draw for %d restarts in lock-step   %8.1f us  (%.1f us per restart)" % (R, us, us / R))
    sums, counts, _ = ops.kmeans_mstep(X, lab, C, k, 0)
    us = timeit(lambda: ops.kmeans_finalize(sums, counts, C))
    print("final  %8.1f us" % us)
    us = timeit(lambda: ops.kmeans_finalize(sums, counts, C, data=data))
    print("final+prep (fused E-step operands) %8.1f us" % us)
    # one Lloyd iteration as scd_amd.kmeans runs it: E-step (operands prepared by the previous finalize) + M-step + finalize
    state = {"c": C, "x16": None}
    def lloyd():
        lab = data.estep(state["c"], expect_few=True)
        sums, counts, _ = ops.kmeans_mstep(X, lab, state["c"], k, 0, x16=state["x16"])
        state["c"], _ = ops.kmeans_finalize(sums, counts, state["c"], data=data)
    for tag, x16 in (("float32 rows", None), ("exact fp16 copy", X16)):
        state.update(c=C, x16=x16)
        us = timeit(lloyd)
        algo = n * dp * 2 + n * d * (4 if x16 is None else 2) + 8 * n
        print("lloyd iteration (estep + mstep[%s] + finalize, no host sync) %8.1f us   X bytes read %.1f MB -> %.0f GB/s" % (tag, us, algo / 1e6, algo / us / 1e3))
    # (round 2 printed `timeit(finalize + estep) - timeit(finalize)` here as "the E-step inside the loop": a difference of two launch-bound
    # loops, below the kernel's own duration - withdrawn; tools/sskm_phases.py under rocprofv3 and scd_kmeans_timing measure it directly)
    cc, _ = ops.kmeans_finalize(sums, counts, C, data=data)
    us = timeit(lambda: data.estep(cc, expect_few=True))          # the hand-over is consumed by the first call only: this times prep + stream
    print("estep call without hand-over %8.1f us" % us)
