#!/usr/bin/env python
"""bench.py - images/sec of the SCD embedding-and-naming hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config c2|c4]
  N > 1: one rank per GPU.  Launched by the driver through torch.distributed.run (WORLD_SIZE = N in the environment); started
  bare (`python bench.py --gpus N`), it spawns that launcher itself as a child process BEFORE touching the GPU.

Workload (BASELINE.json configs[1], "ImageNet-100 unsupervised, CLIP ViT-B/16 encode + 21k WordNet vocab"):
per GPU 126,976 synthetic 224x224 images (already resident in HBM, fp16), K=100 classes, V=21,000 names;
one step = CLIP ViT-B/16 encode + L2-norm -> full-vocab similarity + top-k -> semi-supervised K-Means
(k=100, max_iterations=10, n_init=10: the reference's call-site values, main_unsup.py:350) -> vote loop to
convergence (topk/num_common_vote/num_common_linear = 3/10/2, scripts/evaluate_unsupervised.sh).
Weak scaling: per-GPU images are fixed; `value` = images of all ranks / max-over-ranks step time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the fc1 GEMM of the ViT blocks, MFMA-bound,
timed live with HIP events) and `cpu_baseline` (the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAGES_PER_GPU = 126976
N_CLASSES = 100
VOCAB = 21000
# --config c4 = BASELINE configs[3]: ImageNet-1k-sized synthetic set sharded over 8 GPUs (1,281,167 / 8 images per GPU), K = 1000
CONFIGS = {"c2": dict(images=IMAGES_PER_GPU, n_cluster=N_CLASSES,
                      workload="ImageNet-100 unsupervised (BASELINE configs[1]): CLIP ViT-B/16 encode + V=%d vocab + SSKM k=100 "
                               "(10 restarts x 10 iters) + vote loop"),
           "c4": dict(images=160146, n_cluster=1000,
                      workload="ImageNet-1k synthetic shard (BASELINE configs[3], 1,281,167 images / 8 GPUs): CLIP ViT-B/16 encode + "
                               "V=%d vocab + SSKM k=1000 (10 restarts x 10 iters) + vote loop")}
CONFIGS["c3"] = dict(images=12000, n_cluster=120,
                     workload="Stanford Dogs partially supervised (BASELINE configs[2]): 12,000 images (~3,000 labelled), GCD/DINO ViT-B/16 "
                              "+ CLIP ViT-B/16 encode, V=%d vocab raw top-5, ConSSKM k=120 size 50/1000 (10 restarts x 10 iters, "
                              "main_ptsup.py defaults) + partially supervised vote loop")
# --config c1 = BASELINE configs[0]: CUB-200 unsupervised on CACHED features (5,994 rows, ~4,500 unlabelled; DINO 768-d for the clustering, CLIP
# 512-d for the naming), V = 1,000 names, the shipped `--cluster KM`, K = 200: no encoder in the step - what the k-means / vote path costs
# when it is the whole step.  --config c5 = BASELINE configs[4]: ImageNet-100 + a 100,000-name open vocabulary whose classifier is BUILT inside
# the timed step (text tower over V x 80 prompts; N > 1: name shards + one all-gather) + textual-enhancement re-ranking.
CONFIGS["c1"] = dict(images=5994, n_cluster=200, vocab=1000,
                     workload="CUB-200 unsupervised on cached features (BASELINE configs[0]): 5,994 rows (~4,500 unlabelled), cached DINO 768-d + CLIP "
                              "512-d features, V=%d vocab softmax top-3, sklearn-style KMeans (--cluster KM, n_init=10) k=200 + vote loop; no encoder")
CONFIGS["c5"] = dict(images=IMAGES_PER_GPU, n_cluster=N_CLASSES, vocab=100000,
                     workload="ImageNet-100 + open vocabulary (BASELINE configs[4]): text-tower build of the V=%d-name classifier (80 prompts per "
                              "name, sharded over the ranks + all-gather) INSIDE the step + CLIP ViT-B/16 encode + textual-enhancement top-3 "
                              "(100 * mean(f, t) @ W) + SSKM k=100 (10 restarts x 10 iters) + vote loop")
PEAK_F16_TFLOPS = 2500.0       # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0
FLOP_PER_IMAGE = 2 * 17563453440        # CLIP ViT-B/16 visual tower


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    p.add_argument("--images", type=int, default=None, help="images per GPU (default = the config's)")
    p.add_argument("--n-cluster", type=int, default=None, help="classes / clusters (default = the config's)")
    # 665*197 rows = 512 GEMM row tiles: every GEMM fills the 256 CUs exactly; six of those per launch (3,072 row tiles) amortise
    # the ramp-up / tail of the ~60 launches per batch: +1.5-1.9 % over 665 on the same box, features bit-identical (tools/bigbatch_check.py)
    p.add_argument("--batch", type=int, default=3990)
    p.add_argument("--vocab", type=int, default=None, help="names in the vocabulary (default = the config's: 21,000; c1 1,000; c5 100,000)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-only", action="store_true", help="print the cpu_baseline object and exit (no GPU work; how the main run obtains it)")
    p.add_argument("--cluster", choices=["SSKM", "KM"], default="SSKM",
                   help="clustering stage: SSKM (semi-supervised K-Means, the north-star path; default) or KM = the flag of the shipped "
                        "scripts/evaluate_unsupervised.sh, `KMeans(n_clusters, random_state=0).fit(u_feats)` (main_unsup.py:362)")
    a = p.parse_args()
    if a.cluster == "KM" and a.gpus > 1:
        p.error("--cluster KM is a single-process fit (as in the reference, main_unsup.py:362): use --gpus 1")
    if a.config == "c5" and a.cluster != "SSKM":
        p.error("--config c5 runs the SSKM path")
    if a.config == "c3" and (a.gpus > 1 or a.cluster != "SSKM"):
        p.error("--config c3 is the 1-GPU partially supervised path with its own clustering (ConSSKM): no --gpus / --cluster")
    if a.config == "c1" and a.gpus > 1:
        p.error("--config c1 is the 1-GPU cached-feature path (5,994 rows): --gpus 1")
    if a.config == "c1" and a.cluster == "SSKM" and "--cluster" not in sys.argv:
        a.cluster = "KM"                       # the shipped flag of scripts/evaluate_unsupervised.sh; --cluster SSKM still selects SSKM
    a.vocab = a.vocab or CONFIGS[a.config].get("vocab", VOCAB)
    a.images = a.images or CONFIGS[a.config]["images"]
    a.n_cluster = a.n_cluster or CONFIGS[a.config]["n_cluster"]
    return a


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process (never an
    exec, and before this process has made any HIP call) and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(seed=0):
    """The oracle ('port') timed on the host cores on a bounded sample of the same workload, one leg per stage of SURVEY.md 8(d)
    (an end-to-end CPU pass over 127k images would take hours); `value` composes the stages of the metric harmonically.
    Thread count: the fastest of a 32 / 64 / 128 / all-cores sweep per leg (on a 256-core host the 4096-row legs do not scale to all
    of them).  Legs: (i) CLIP image tower, (ii) similarity + top-5, (iii) one Lloyd iteration in the reference's own form
    (broadcast (A - B)^2 in 1024-row blocks, torch.min, per-cluster mean: sskm_constrained.py:189-224,125-128) and in GEMM form
    (C/OpenMP), (iv) sklearn KMeans on the same rows, (v) one pass of the vote loop (Counter + Munkres + argmax re-classification)."""
    import ctypes as C
    from oracle import clip_oracle as co
    from oracle import naming_oracle as no
    from scd_amd.clip import weights as W
    cores = os.cpu_count() or 1
    sweep = sorted({min(cores, t) for t in (32, 64, 128, cores)})
    t_budget = time.time()
    used = {}

    leg_s = {}

    def best_of(tag, fn):
        """fn(threads, divisor) -> seconds.  The thread count is chosen on a quarter-size sample, ascending, stopping at the first
        count that is slower than its predecessor (on a 256-core host every leg peaks at 32-64 threads); the leg is then timed on
        the full sample."""
        t_leg = time.time()
        best_t, thr = None, sweep[0]
        for t in sweep:
            sec = fn(t, 4)
            if best_t is not None and sec > best_t:
                break
            best_t, thr = sec, t
        used[tag] = thr
        sec = fn(thr, 1)
        leg_s[tag] = round(time.time() - t_leg, 1)
        return sec

    # (i) encode: the fp32 torch restatement of the CLIP visual tower
    sd = W.synthetic_clip_state_dict(seed=0, text=False)
    img = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(1))

    def enc(t, div):
        torch.set_num_threads(t)
        co.clip_encode_image(sd, img[:2])
        t0 = time.time()
        co.clip_encode_image(sd, img[: 32 // div])
        return time.time() - t0
    enc_ips = 32 / best_of("encode", enc)
    # (ii) + (iii, GEMM form): C restatement (OpenMP)
    so = os.path.join(ROOT, "oracle", "c", "liboracle.so")
    lib = C.CDLL(so)
    rs = np.random.RandomState(seed)
    n_s, d, v = 4096, 512, VOCAB
    f = (rs.randn(n_s, d) / np.sqrt(d)).astype(np.float32)
    wt = (rs.randn(v, d) / np.sqrt(d)).astype(np.float32)
    idx = np.zeros((n_s, 5), dtype=np.int64)
    val = np.zeros((n_s, 5), dtype=np.float32)
    P = lambda a: C.c_void_p(a.ctypes.data)

    def sim(t, div):
        lib.oracle_set_threads(t)
        t0 = time.time()
        lib.oracle_sim_topk(P(f), P(wt), C.c_int64(n_s // div), d, C.c_int64(v), C.c_double(100.0), 5, P(idx), P(val))
        return time.time() - t0
    sim_ips = n_s / best_of("sim", sim)
    n_k, k, dk = 20000, N_CLASSES, 768                       # N = 20k, K = 100, D = 768
    x = rs.randn(n_k, dk).astype(np.float32)
    c0 = x[rs.choice(n_k, k, replace=False)].copy()
    lab = np.zeros(n_k, dtype=np.int64)
    mind = np.zeros(n_k, dtype=np.float32)

    def lloyd_gemm(t, div):
        lib.oracle_set_threads(t)
        cc = c0.copy()
        t0 = time.time()
        lib.oracle_estep(P(x), P(cc), C.c_int64(n_k // div), dk, k, P(lab), P(mind))
        lib.oracle_mstep(P(x), P(lab), C.c_int64(n_k // div), dk, k, P(cc))
        return time.time() - t0
    it_gemm = best_of("lloyd_gemm", lloyd_gemm)
    xt, ct = torch.from_numpy(x), torch.from_numpy(c0)

    def lloyd_ref(t, div):                                  # the reference's own arithmetic (pairwise_distance with batch_size 1024)
        torch.set_num_threads(t)
        t0 = time.time()
        dist = torch.zeros(n_k, k)
        for i in range(0, n_k // div, 1024):
            a = xt[i:i + 1024].unsqueeze(1)
            dist[i:i + 1024] = ((a - ct.unsqueeze(0)) ** 2.0).sum(dim=-1)
        _, ul = torch.min(dist, 1)
        for j in range(k):
            sel = xt[ul == j]
            if len(sel):
                sel.mean(0)
        return time.time() - t0
    it_ref = best_of("lloyd_reference_form", lloyd_ref)
    # 10 restarts x (10 Lloyd iterations + ~50 k-means++ sweeps of one centre each ~ 0.5 iteration-equivalents)
    km_ips = n_k / (it_gemm * 10 * (10 + 0.5))
    km_ref_ips = n_k / (it_ref * 10 * (10 + 0.5))
    # (iv) sklearn KMeans (--cluster KM, main_unsup.py:362) on the same rows: explicit init, one start, lloyd
    sk_s, sk_it = None, 0
    t_leg = time.time()
    try:
        from sklearn.cluster import KMeans as SK
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=used.get("lloyd_gemm", 32)):
            t0 = time.time()
            skm = SK(n_clusters=k, init=c0, n_init=1, algorithm="lloyd", max_iter=10, random_state=0).fit(x)
            sk_s, sk_it = time.time() - t0, int(skm.n_iter_)
    except Exception:
        pass
    leg_s["sklearn"] = round(time.time() - t_leg, 1)
    # (v) one pass of the vote loop (main_unsup.py:568-614) on N_u = 95,000 rows: Counter per cluster, Munkres (the reference's
    # state machine restated in numpy) and the argmax re-classification
    n_u, kv = 95000, N_CLASSES
    cen = rs.randn(kv, 512).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    yv = rs.randint(0, kv, n_u)
    fu = cen[yv] + (0.6 / np.sqrt(512)) * rs.randn(n_u, 512).astype(np.float32)
    fu /= np.linalg.norm(fu, axis=1, keepdims=True)
    wv = np.concatenate([cen, (rs.randn(VOCAB - kv, 512) / np.sqrt(512)).astype(np.float32)]).T.copy()
    nidx = np.stack([yv, (yv + 1 + rs.randint(0, 50, n_u)) % VOCAB, rs.randint(0, VOCAB, n_u)], axis=1).astype(np.int64)
    nouns = ["n%d" % j for j in range(VOCAB)]
    torch.set_num_threads(min(cores, 32))
    t_leg = time.time()
    try:
        from threadpoolctl import threadpool_limits
        _lim = threadpool_limits(limits=min(cores, 32))                  # numpy's BLAS: 95,000 x 512 @ 512 x 100 per pass
    except Exception:
        _lim = None
    t0 = time.time()
    no.vote_loop_unsup(nidx, (yv + (rs.rand(n_u) < 0.1) * rs.randint(0, kv, n_u)) % kv, fu, wv, nouns, kv, 3, 10, 2, max_iter=1)
    vote_s = time.time() - t0
    if _lim is not None:
        _lim.restore_original_limits()
    leg_s["vote"] = round(time.time() - t_leg, 1)
    vote_ips = n_u / (vote_s * 3)                             # the bench's loops converge in ~3 passes
    total = 1.0 / (1.0 / enc_ips + 1.0 / sim_ips + 1.0 / km_ips + 1.0 / vote_ips)
    return {"value": round(total, 3), "unit": "images/sec", "cores": max(used.values()), "host_cores": cores, "kind": "port",
            "threads_per_leg": used, "seconds_per_leg_incl_sweep": leg_s,
            "legs_images_per_sec": {"encode": round(enc_ips, 2), "sim_topk": round(sim_ips, 1), "sskm_gemm_form": round(km_ips, 1),
                                    "sskm_reference_form": round(km_ref_ips, 1), "vote_loop": round(vote_ips, 1)},
            "sklearn_kmeans": None if sk_s is None else {"seconds": round(sk_s, 3), "n_iter": sk_it, "rows": n_k, "d": dk, "k": k},
            "sample": "oracle on host, best thread count of %s per leg: encode 32 imgs (torch fp32, %.2f img/s); sim+top-5 4096 rows x "
                      "V=21000 (C/OpenMP, %.0f img/s); one Lloyd iteration 20000x768xK=100 in GEMM form (C/OpenMP, %.3f s) and in the "
                      "reference's broadcast form (torch, %.3f s), scaled to 10 restarts x 10 iterations; sklearn KMeans (explicit "
                      "init, <= 10 iterations) %s s; one vote-loop pass on 95,000 rows (numpy Counter + Munkres + argmax, %.2f s) x 3 "
                      "passes; `value` = harmonic composition of encode, sim, GEMM-form SSKM and vote; %.0f s of CPU work"
                      % (sweep, enc_ips, sim_ips, it_gemm, it_ref, "n/a" if sk_s is None else "%.2f" % sk_s, vote_s,
                         time.time() - t_budget)}


def dominant_kernel_roofline(ms, launches, flop):
    """fc1 GEMM of the ViT blocks (gemm_w4_kernel<QuickGELU,bias,no-residual,LN-folded>): [B*197,768] x [3072,768]^T.  Every launch
    inside the timed steps is bracketed by HIP events on its launch stream (scd_encoder_timing); algorithmic FLOPs =
    2*M*N*K per launch (M = 197 * batch padded to 256 images)."""
    sec = ms / 1e3
    tf = flop / max(sec, 1e-12) / 1e12
    # HBM-side traffic per launch cannot be read without the profiler: it is taken from this round's committed PMC passes
    # (profiles/r05_pmc_fc1.json: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs of this script, tools/history/gpu_r05_final.sh)
    # at the default launch size (3,990 images = 786,432 rows); another --batch scales it by its rows per launch
    traffic, pmf = None, ""
    try:
        pmf = next(p for p in (os.path.join(ROOT, "profiles", "r0%d_pmc_fc1.json" % r) for r in (6, 5, 4)) if os.path.exists(p))
        with open(pmf) as f:
            pm = json.load(f)
        rows_per_launch = flop / max(launches, 1) / (2.0 * 3072 * 768)
        traffic = round(pm["traffic_bytes_per_launch"] * rows_per_launch / pm["rows"])
    except Exception:
        pass
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_F16_TFLOPS, 4), "traffic": traffic,
            "traffic_source": "replayed from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command), scaled by rows per launch; not measured in this run" % os.path.basename(pmf) if traffic is not None else None,
            "kernel": "gemm_w4_kernel<QuickGELU,bias,no-residual,LN-folded> (ViT fc1, n=3072 k=768)", "launches": launches,
            "avg_launch_us": round(ms * 1e3 / max(launches, 1), 1)}


def secondary_rooflines(out, wt, dev, km_fit=False):
    """Call-level rates of the two other kernels north_star names, measured after the timed region on this run's own
    tensors (HIP events on the current stream, 20 calls each): the similarity + top-k call (MFMA-bound) and the k-means
    E-step call (HBM-bound: centre prep + streaming filter + refine) on the run's features and on clustered features of the
    same shape.  Informational; `roofline` above stays the dominant kernel of the metric."""
    import torch
    from scd_amd import ops
    res = []

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters

    feats = out["feats"]
    n, d = feats.shape
    v = wt.shape[0]
    t = timeit(lambda: ops.sim_topk(feats, wt, 3, "softmax"), 5)
    fb_rows = int(ops.sim_topk(feats, wt, 3, "softmax", return_fallback=True)[2].item())
    fl = 2.0 * n * v * d
    res.append({"kernel": "scd_sim_topk_prenorm call (sim_init + sim_topk_rb8_kernel + sim_refine4_kernel + exact-pass launches; the frozen vocabulary's norm is not recomputed), softmax k = 3, %d x %d x %d" % (n, v, d), "bound": "mfma",
                "achieved": round(fl / t / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / t / 2.5e15, 4),
                "call_us": round(t * 1e6, 1), "rows_through_exact_pass": fb_rows})
    x = feats.float()
    c = torch.as_tensor(out["kmeans"].cluster_centers_).to(dev).to(torch.float32).contiguous()
    k = int(c.shape[0])
    data = ops.KMeansData(x)
    dp = (d + 127) // 128 * 128
    by = n * dp * 2 + 4 * n + 128 * dp * 2

    def estep_line(tag, dat, cen, few):
        _, ref = dat.estep(cen, return_refined=True)
        t = timeit(lambda: dat.estep(cen, expect_few=few))
        kern = "estep_rb_kernel" if (d == 512 and 128 < k <= 2048) else ("estep_stream_kernel" if k <= 128 else "estep_mfma_kernel")
        resident = "Infinity-Cache resident operand (%d MB <= 256 MB: repeated calls do not stream from HBM)" % (by >> 20) if by <= (256 << 20) else "operand streams from HBM"
        name = "scd_kmeans_estep call (centre prep + %s + refine), N=%d D=%d K=%d, %s; %s" % (kern, n, d, k, tag, resident)
        if k > 300:            # with 16-bit MFMA operands the E-step is matrix-bound beyond K ~ 300
            fl_e = 2.0 * n * k * d
            return {"kernel": name, "bound": "mfma", "achieved": round(fl_e / t / 1e12, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(fl_e / t / (PEAK_F16_TFLOPS * 1e12), 4), "call_us": round(t * 1e6, 1), "rows_refined": int(ref.item())}
        return {"kernel": name, "bound": "hbm", "achieved": round(by / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(by / t / 8e12, 4), "call_us": round(t * 1e6, 1), "rows_refined": int(ref.item())}
    # (i) the run's own features and final centres: CLIP features of SYNTHETIC images sit in one blob, many rows fall inside the
    # filter's error bound and are re-evaluated exactly; (ii) the same shape with cluster structure (SURVEY.md 8d generator), where
    # the filter decides every row - the regime of real features and of the north-star HBM target
    res.append(estep_line("this run's features / final centres", data, c, False))
    g = torch.Generator(device=dev).manual_seed(11)
    cen = torch.nn.functional.normalize(torch.randn(k, d, device=dev, generator=g), dim=-1)
    yy = torch.randint(0, k, (n,), device=dev, generator=g)
    xc = torch.nn.functional.normalize(cen[yy] + (0.8 / d ** 0.5) * torch.randn(n, d, device=dev, generator=g), dim=-1)
    dc = ops.KMeansData(xc)
    res.append(estep_line("clustered synthetic features / converged centres", dc, cen.contiguous(), True))
    # (iii) the kernel INSIDE the Lloyd loop: an SSKM fit (3 restarts x 10 iterations, the engine's fused scd_kmeans_lloyd_step)
    # on the clustered features with every streaming-filter launch bracketed by HIP events on its launch stream
    # (scd_kmeans_timing); the wall time of an iteration (host one iteration behind the device) beside it
    if k <= 128:
        from scd_amd import kmeans as km
        eng = km.KMeansEngine(k=k, tolerance=-1.0, max_iterations=10, n_init=3, random_state=0)   # tolerance < 0: all 10 iterations run
        eng.fit(xc)                                      # warm-up: allocations, kernel attributes
        ops.kmeans_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.fit(xc)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        smp = ops.kmeans_timing(False) * 1e-3                                  # seconds per launch, call order
        # iterations 0-1 of a restart run the filter with the refine in a launch of its own (many rows inside the bound right after
        # the seeding); from iteration 2 on the few flagged rows are re-evaluated in the filter kernel's tail
        late = np.array([smp[j] for j in range(len(smp)) if j % 10 >= 2]) if len(smp) == 30 else smp
        t = float(late.mean())
        res.append({"kernel": "estep_stream_kernel inside the Lloyd loop (HIP events around every launch of an SSKM fit, 3 restarts x 10 "
                              "iterations; iterations >= 2 of a restart: %d launches), N=%d D=%d K=%d, clustered synthetic features; %s"
                              % (len(late), n, d, k, "Infinity-Cache resident operand" if by <= (256 << 20) else "operand streams from HBM"),
                    "bound": "hbm", "achieved": round(by / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(by / t / 8e12, 4),
                    "kernel_us": round(t * 1e6, 1), "kernel_us_median": round(float(np.median(late)) * 1e6, 1),
                    "kernel_us_all_launches": round(float(smp.mean()) * 1e6, 1),
                    "fit_wall_ms_incl_seeding": round(wall * 1e3, 2),
                    "note": "HIP-event brackets include the dispatch latency of the bracketed launch (~5-8 us on a 30 us kernel); the "
                            "rocprofv3 kernel trace of a whole fit (tools/lloyd_multi_prof.py, profiles/r05_lloyd_fit_kernel_stats.csv) has the kernel alone"})
    # (iii') `--cluster KM` (the shipped script's flag): `KMeans(k, random_state=0).fit` on the clustered features' unlabelled share
    # (95,000 rows at C2) - greedy k-means++ of the ten starts in lock-step + ten Lloyd runs in C (scd_kpp_greedy_lockstep,
    # scd_kmeans_lloyd_run_sk); wall time of the whole fit
    if km_fit and k <= 2048:         # with --cluster KM only: the default command's kernel statistics stay the default path's
        from scd_amd.cluster import KMeans
        n_km = min(n, int(n * 0.75))
        xk = xc[:n_km].half().float().contiguous()
        KMeans(n_clusters=k, random_state=0).fit(xk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kmf = KMeans(n_clusters=k, random_state=0).fit(xk)
        torch.cuda.synchronize()
        res.append({"kernel": "scd_amd.cluster.KMeans(n_clusters=%d, random_state=0).fit on %d x %d clustered fp16-exact features "
                              "(--cluster KM, scikit-learn 1.0.2 rules: 10 starts, greedy k-means++, Lloyd to tol 1e-4)" % (k, n_km, d),
                    "bound": "latency", "fit_wall_ms": round((time.perf_counter() - t0) * 1e3, 2), "n_iter_kept_start": int(kmf.n_iter_)})
    # (iv) one round of the lock-step k-means++ seeding (N*D*s + 8*N bytes per added centre; the ten restarts'
    # centres of a round share ONE pass over the exact fp16 copy, s = 2): scd_kpp_seed_lockstep on the clustered features, rounds with
    # 20+ centres present (distance update through the MFMA filter), HIP events around the call / rounds
    x16 = ops.f16_exact(xc.half().float())
    if x16 is not None and d % 32 == 0:
        xe = xc.half().float().contiguous()
        rr, kk, m0, tr = 10, 64, 24, 32
        gg = torch.Generator(device=dev).manual_seed(3)
        buf = torch.zeros((rr, kk, d), dtype=torch.float32, device=dev)
        d2 = torch.full((rr, n), float("inf"), dtype=torch.float32, device=dev)
        for j in range(m0):                                   # distances to the first m0 (random) centres of every restart
            rows = xe[torch.randint(0, n, (rr,), device=dev, generator=gg)].contiguous()
            buf[:, j] = rows
            ops.min_update_multi(xe, rows, d2)
        rv = torch.rand((tr, rr), device=dev, generator=gg, dtype=torch.float32).contiguous()
        d2w = d2.clone()
        ops.kpp_seed_lockstep(xe, x16, d2w, rv, buf, m0)      # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d2w.copy_(d2)
        e0.record()
        ops.kpp_seed_lockstep(xe, x16, d2w, rv, buf, m0)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / tr
        byr = n * d * 2 + 8 * n * rr
        res.append({"kernel": "one round of scd_kpp_seed_lockstep (draw: 3 launches; fetch + centre operands; muf_filter_kernel + muf_exact_kernel), "
                              "10 restarts in lock-step, N=%d D=%d, %d-%d centres present, clustered synthetic features" % (n, d, m0, m0 + tr),
                    "bound": "hbm", "achieved": round(byr / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(byr / t / 8e12, 4),
                    "round_us": round(t * 1e6, 1),
                    "note": "the round's distance update alone (muf_filter_kernel 33 us + muf_exact_kernel 11 us at this size, "
                            "profiles/r05_lloyd_fit_kernel_stats.csv) moves these bytes at ~3.2 TB/s; the draw's three dependent launches add 21 us"})
    return res


class PowerSampler:
    """Board power / shader clock of THIS rank's GPU while the timed steps run, read from the amdgpu hwmon files (read-only sysfs: power1_input
    in microwatts, power1_cap, freq1_input = sclk in Hz) by a host thread every 0.25 s.  Context for the roofline fraction, not part of the
    metric: the MI355X holds the encoder at its board power cap and lowers the clock to do so (docs/design/encoder_time_budget.md).
    Returns None when the device's PCI address or its hwmon directory cannot be found."""

    def __init__(self, device_index):
        import glob
        import threading
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            cand = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
            if cand and os.path.exists(os.path.join(cand[0], "power1_input")):
                self.dir, self.bdf = cand[0], bdf
        except Exception:
            self.dir = None
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            w, hz = self._read("power1_input"), self._read("freq1_input")
            if w is not None:
                self.samples.append((w / 1e6, (hz or 0.0) / 1e6))
            self._stop.wait(0.25)

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(timeout=2.0)
        if not self.samples:
            return None
        w = sorted(x[0] for x in self.samples)
        f = [x[1] for x in self.samples if x[1] > 0]
        cap = self._read("power1_cap")
        return {"source": "amdgpu hwmon power1_input / freq1_input of %s, every 0.25 s over the timed steps" % self.bdf, "samples": len(w),
                "median_w": round(w[len(w) // 2], 1), "max_w": round(w[-1], 1), "cap_w": None if cap is None else round(cap / 1e6, 1),
                "sclk_mhz_median": round(sorted(f)[len(f) // 2], 0) if f else None}


def cpu_baseline_child():
    """cpu_baseline() in a child process of its own: the C restatement is a host library built elsewhere, and whatever goes wrong in
    it (an illegal instruction, a crash in OpenMP) must not take the measured line with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True, timeout=900)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": "cpu baseline child exited with code %d" % r.returncode, "stderr_tail": r.stderr[-400:]}
    except Exception as e:          # noqa: BLE001 - the baseline is reported beside the metric, never instead of it
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main_c1(args, dev):
    """BASELINE configs[0]: the cached-feature path (no encoder).  One JSON line; `roofline` is the E-step call of the fit's shape (the
    path's HBM-bound kernel), `secondary_rooflines` carries the fit itself - the step is launch- and latency-bound, and says so."""
    from scd_amd import ops, pipeline
    n, k, v = args.images, args.n_cluster, args.vocab
    g = torch.Generator(device=dev).manual_seed(2024)
    y = torch.randint(0, k, (n,), generator=g, device=dev)
    # cached "DINO" features: float32 [n, 768], L2-normalised rows around class centres (extract_feature + F.normalize, main_unsup.py:114-147)
    cen = torch.nn.functional.normalize(torch.randn(k, 768, generator=g, device=dev), dim=-1)
    xf = torch.nn.functional.normalize(cen[y] + (0.9 / 768 ** 0.5) * torch.randn(n, 768, generator=g, device=dev), dim=-1).contiguous()
    # cached CLIP features fp16 [n, 512] around the planted prototypes; vocabulary: rows < K the (jittered) prototypes, the rest random names
    proto = torch.nn.functional.normalize(torch.randn(k, 512, generator=g, device=dev), dim=-1)
    cf = torch.nn.functional.normalize(proto[y] + (0.9 / 512 ** 0.5) * torch.randn(n, 512, generator=g, device=dev), dim=-1).to(torch.float16).contiguous()
    w = torch.randn(v, 512, generator=g, device=dev)
    w[:k] = proto + 0.05 * torch.randn(k, 512, generator=g, device=dev) / 512 ** 0.5
    wt = ops.freeze_vocab(ops.l2norm_rows(w.contiguous()).to(torch.float16).contiguous())
    nouns = ["name_%05d" % i for i in range(v)]
    mask_lab = pipeline.labelled_split(y, k, seed=5)
    l_targets = y[torch.as_tensor(mask_lab, device=dev)]
    stage_ms = {}

    def step(timed):
        timers = [] if timed else None
        out = pipeline.run_cached(xf, cf, mask_lab, wt, nouns, k, topk=3, num_common_vote=10, num_common_linear=2, timers=timers,
                                  cluster=args.cluster, l_targets=l_targets)
        if timed:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(timers[:-1], timers[1:]):
                stage_ms[n1] = stage_ms.get(n1, 0.0) + e0.elapsed_time(e1)
        return out
    for _ in range(args.warmup):
        out = step(False)
    torch.cuda.synchronize()
    power = PowerSampler(dev.index or 0)
    power.start()
    ops.trace_mark(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    ops.trace_mark(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    power = power.stop()
    u_true = y.cpu().numpy()[~mask_lab]
    name_hits = float(np.mean(np.array([int(nm.split("_")[1]) for nm in out["cand_names"]])[out["u_preds"]] == u_true))
    # the E-step call on the fit's rows and final centres (HIP events, 50 calls): the HBM-bound kernel of this path
    xu = out["u_feats"].contiguous()
    cent = torch.as_tensor(out["kmeans"].cluster_centers_).to(dev).to(torch.float32).contiguous()
    data = ops.KMeansData(xu)
    for _ in range(5):
        data.estep(cent)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        data.estep(cent)
    e1.record()
    torch.cuda.synchronize()
    t_e = e0.elapsed_time(e1) * 1e-3 / 50
    nu, d = xu.shape
    dp = (d + 127) // 128 * 128
    by = nu * dp * 2 + 4 * nu + ((k + 127) // 128 * 128) * dp * 2
    sclk = (power or {}).get("sclk_mhz_median")
    line = {"metric": "images/sec end-to-end (encode+sim+k-means) on 224^2 synth, 21k vocab", "value": round(n * args.steps / dt, 2), "unit": "images/sec",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": CONFIGS["c1"]["workload"] % v if args.cluster == "KM" else (CONFIGS["c1"]["workload"] % v).replace(
                           "sklearn-style KMeans (--cluster KM, n_init=10)", "SSKM (10 restarts x 10 iters)"),
                       "cluster": args.cluster, "images_per_gpu": n, "vocab": v, "n_cluster": k, "encode_batch": None,
                       "weights": "none: cached features (synthetic, seeded)", "parallelism": "dp1"},
            "stage_ms_per_step": {kk: round(vv / args.steps, 3) for kk, vv in stage_ms.items()}, "encode_tflops": None,
            "vote_iters": out["vote_iters"], "synthetic_name_accuracy": round(name_hits, 4),
            "roofline": {"bound": "hbm", "achieved": round(by / t_e / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(by / t_e / (PEAK_HBM_GBS * 1e9), 4),
                         "traffic": None, "kernel": "scd_kmeans_estep call (centre prep + filter + refine), N=%d D=%d K=%d, this run's rows / final centres; "
                                                    "Infinity-Cache resident operand (%.1f MB): a latency-bound launch, not a streaming one" % (nu, d, k, by / 2 ** 20),
                         "launches": 50, "avg_launch_us": round(t_e * 1e6, 1),
                         "note": "the c1 step is launch- and latency-bound (4,500 rows: every kernel is a few microseconds; the fit is %d k-means++ rounds and "
                                 "Lloyd iterations driven from C): no kernel of it approaches a roofline, the stage times are the measurement" % k},
            "board_power": power, "sclk_mhz_median": sclk, "encode_cycles_per_image": None,
            "km_fit": {"n_iter_kept_start": int(getattr(out["kmeans"], "n_iter_", 0)), "fit_ms_per_step": round(stage_ms.get("kmeans", 0.0) / args.steps, 3)},
            "secondary_rooflines": []}
    line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_child()
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    local_rank = local_rank % torch.cuda.device_count()     # (debug: several ranks may share a GPU with SCD_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SCD_DIST_BACKEND", "nccl"), rank=rank, world_size=world)   # nccl == RCCL on ROCm
        group = dist.group.WORLD
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank)

    import scd_amd.clip as clip
    from scd_amd import pipeline
    n_cls = args.n_cluster
    if args.config == "c1":
        return main_c1(args, dev)
    clip.allow_synthetic()          # random-init weights + hash tokenizer: there is no checkpoint offline (`data: synthetic`)
    model, _ = clip.load("ViT-B/16", device="cuda")
    images, y, base = pipeline.synthetic_images(args.images, n_cls, seed=rank, device=dev)
    wt, nouns = pipeline.synthetic_vocab(model, base, args.vocab, 0, dev)
    if world > 1:
        # the open-vocabulary part of W is a real, SHARDED text-tower build: every rank encodes its contiguous range of
        # names (2 prompts each) and one RCCL all-gather assembles the [512, V] classifier on all ranks (setup, untimed);
        # the first N_CLASSES rows keep the planted prototypes so that the name accuracy stays meaningful
        from scd_amd.local_utils import clip_lang_util as clu
        w_text = clu.zeroshot_classifier_sharded(nouns, ["a photo of a {}.", "a {}."], model, group, names_per_batch=256)
        wt_text = w_text.t().contiguous()
        wt_text[:n_cls] = wt[:n_cls]
        wt = wt_text
    mask_lab = pipeline.labelled_split(y, n_cls, seed=5 + rank)
    l_targets = y[torch.as_tensor(mask_lab, device=dev)]
    feat_model = None
    name_row = np.arange(n_cls)                    # vocabulary row of class c's true name
    if args.config == "c3":
        from scd_amd.clip import weights as W
        from scd_amd.clip.model import DinoViT
        feat_model = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()      # the GCD tower = DINO ViT-B/16 (main_ptsup.py:263-287)
        # the true names sit at scattered vocabulary rows, as in a real corpus: with them at rows 0..K-1 the reference's `known_name_idx`
        # quirk (candidate POSITIONS compared with vocabulary indices from the second vote iteration on, main_ptsup.py:638,666) would
        # filter out exactly the true names
        name_row = np.sort(np.random.RandomState(123).choice(np.arange(n_cls, args.vocab), size=n_cls, replace=False))
        rows_t = torch.as_tensor(name_row, device=dev)
        proto, other = wt[:n_cls].clone(), wt[rows_t].clone()
        wt[rows_t] = proto
        wt[:n_cls] = other
        # four near-synonyms per class at other scattered rows (a dog corpus holds "collie", "border collie", "rough collie" ...): the
        # images of a class then disagree about their second name, as real ones do - with ONE near name per class every cluster votes
        # for the same two names and the vote has fewer distinct names than clusters (the reference indexes past `voted` then, :654)
        gs = torch.Generator(device=dev).manual_seed(77)
        free = np.setdiff1d(np.arange(n_cls, args.vocab), name_row)
        syn_rows = np.sort(np.random.RandomState(124).choice(free, size=4 * n_cls, replace=False))
        syn = proto.float().repeat_interleave(4, dim=0)
        syn = syn + 0.06 * torch.randn(syn.shape, generator=gs, device=dev) / syn.shape[1] ** 0.5      # (the true name sits at 0.05; the other classes at ~1 - cos = 0.008)
        wt[torch.as_tensor(syn_rows, device=dev)] = torch.nn.functional.normalize(syn, dim=-1).to(wt.dtype)
        lab_names = [nouns[name_row[c]] for c in range(n_cls // 2)]                       # the labelled classes' names are known (:597-603)

    build_vocab, text_feats = None, None
    if args.config == "c5":
        # open vocabulary: the classifier of all V names is built by the text tower INSIDE every step (80 prompts per name, the reference's
        # zeroshot_classifier call; N > 1: contiguous name shards per rank + one all-gather, clip_lang_util.zeroshot_classifier_sharded);
        # the first K rows then take the planted prototypes so that the vote has true names to find.  Textual enhancement: per-image
        # closed-set text features (the reference's `closed_text_feats`, loaded from disk there) = jittered class prototypes, resident
        from scd_amd.local_utils import clip_lang_util as clu
        protos = wt[:n_cls].clone()

        def build_vocab():
            if world > 1:
                w = clu.zeroshot_classifier_sharded(nouns, clu.imagenet_templates, model, group, names_per_batch=1024)
            else:
                w = clu.zeroshot_classifier(nouns, clu.imagenet_templates, model, names_per_batch=1024)
            wtb = _ops.transpose_f16(w)
            wtb[:n_cls] = protos
            return _ops.freeze_vocab(wtb)
        gt = torch.Generator(device=dev).manual_seed(900 + rank)
        text_feats = torch.nn.functional.normalize(protos.float()[y] + 0.5 * torch.randn((args.images, protos.shape[1]), generator=gt, device=dev)
                                                   / protos.shape[1] ** 0.5, dim=-1).to(torch.float16).contiguous()
    from scd_amd import ops as _ops
    wt = _ops.freeze_vocab(wt)          # written for the last time above: the similarity filter's vocabulary norm once, not per call

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    stage_ms = {}

    def step(i, timed):
        timers = [] if timed else None
        if args.config == "c3":
            out = pipeline.run_ptsup(model, feat_model, images, mask_lab, l_targets, wt, nouns, lab_names, n_cls, topk=2, num_common_vote=5,
                                     num_common_linear=2, size_min=50, size_max=1000, batch=args.batch, seed=i, timers=timers)
        else:
            out = pipeline.run(model, images, mask_lab, l_targets, wt, nouns, n_cls, topk=3, num_common_vote=10,
                               num_common_linear=2, batch=args.batch, seed=i, group=group, timers=timers, cluster=args.cluster,
                               build_vocab=build_vocab, text_feats=text_feats)
        if timed:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(timers[:-1], timers[1:]):
                stage_ms[n1] = stage_ms.get(n1, 0.0) + e0.elapsed_time(e1)
        return out

    for i in range(args.warmup):
        out = step(i, False)
    barrier()
    model.visual.enc.timing(True)
    if args.config == "c5":
        model._text.timing(True)        # the text tower's fc1 launches, bracketed by HIP events like the image tower's
    power = PowerSampler(local_rank) if rank == 0 else None       # a host thread reading sysfs: nothing enters the stream
    if power:
        power.start()
    _ops.trace_mark(False)              # an empty marker kernel in front of the timed steps and one behind them: a kernel trace of this
    t0 = time.perf_counter()            # command can be cut down to the timed region (tools/trace_window_stats.py); ~2 us each
    for i in range(args.steps):
        out = step(100 + i, True)
    _ops.trace_mark(True)
    barrier()
    dt = time.perf_counter() - t0
    power = power.stop() if power else None
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    # quality of the synthetic run (not part of the metric): cluster purity and names found
    yn = y.cpu().numpy()
    u_true = yn[~mask_lab]
    class_of_row = {int(r): c for c, r in enumerate(name_row)}
    name_hits = float(np.mean(np.array([class_of_row.get(int(n.split("_")[1]), -1) for n in out["cand_names"]])[out["u_preds"]] == u_true))

    if rank == 0:
        total_images = args.images * world * args.steps
        value = total_images / dt
        roof = dominant_kernel_roofline(*model.visual.enc.timing(False))
        enc_s = stage_ms.get("encode", 0.0) / 1e3 / args.steps
        line = {
            "metric": "images/sec end-to-end (encode+sim+k-means) on 224^2 synth, 21k vocab",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": (CONFIGS[args.config]["workload"] % args.vocab) if args.cluster == "SSKM" else
                       (CONFIGS[args.config]["workload"] % args.vocab).replace("SSKM k=", "sklearn-style KMeans (--cluster KM, n_init=10) k=").replace(" (10 restarts x 10 iters)", ""),
                       "cluster": args.cluster,
                       "images_per_gpu": args.images, "vocab": args.vocab, "n_cluster": n_cls, "encode_batch": args.batch,
                       "weights": "random-init (seeded), no checkpoint offline", "parallelism": "dp%d" % world},
            "stage_ms_per_step": {k: round(v / args.steps, 2) for k, v in stage_ms.items()},
            "encode_tflops": round(args.images * FLOP_PER_IMAGE / max(enc_s, 1e-9) / 1e12, 1),
            "vote_iters": out["vote_iters"], "synthetic_name_accuracy": round(name_hits, 4),
            "roofline": roof,
            "board_power": power,
        }
        # clock-normalised encode cost: boxes of this pool hold 1.69-1.88 GHz at the same 1,400-W cap, so images/s moves by +-3 % box to
        # box with the same binary; shader cycles per image (median sclk over the timed steps x encode seconds / images) does not
        sclk = (power or {}).get("sclk_mhz_median")
        line["sclk_mhz_median"] = sclk
        line["encode_cycles_per_image"] = round(sclk * 1e6 * enc_s / args.images) if (sclk and enc_s > 0) else None
        if args.config == "c3":
            line["encode_cycles_per_image_note"] = "two towers (DINO / GCD + CLIP) per image"
        if args.config == "c3":
            km = out["kmeans"]
            line["config"]["cluster"] = "ConSSKM"
            line["consskm"] = {"fit_ms_per_step": line["stage_ms_per_step"].get("kmeans"), "transport_solves_per_fit": km.stats.get("transport_solves"),
                               "host_threads": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                               "phase_ms_last_fit": km.stats.get("phase_ms"),       # SCD_CONSSKM_PROFILE=1 only (a device sync per iteration)
                               "cluster_sizes_min_max": [int(np.bincount(out["labels"][int(mask_lab.sum()):], minlength=n_cls).min()),
                                                         int(np.bincount(out["labels"][int(mask_lab.sum()):], minlength=n_cls).max())],
                               "note": "the restarts' flow problems of an iteration are solved on host threads (scd_transport_solve_batch); "
                                       "the step's images/s counts each image once although two towers encode it"}
            line["secondary_rooflines"] = []
        else:
            line["secondary_rooflines"] = secondary_rooflines(out, out.get("wt", wt), dev, km_fit=(args.cluster == "KM"))
        if args.config == "c5":
            # the text tower (row a4): GEMM FLOPs actually executed after the context trimming = 3 x its fc1 launches' FLOPs (per row and
            # layer QKV 3 w^2 + proj w^2 + fc1 4 w^2 + fc2 4 w^2 multiply-adds, fc1 is a third; attention and the pooling not counted) over
            # the "text_tower" stage's time (tokenisation on the host runs behind the device work and is inside it)
            t_ms, t_n, t_fl = model._text.timing(False)
            tt_s = stage_ms.get("text_tower", 0.0) / 1e3 / args.steps
            prompts = len(nouns) * 80 // world
            line["secondary_rooflines"].insert(0, {
                "kernel": "CLIP text tower inside the step: zeroshot_classifier%s over %d names x 80 prompts per rank (trimmed context, four length groups per 256 names)"
                          % ("_sharded + all-gather" if world > 1 else "", len(nouns) // world),
                "bound": "mfma", "achieved": round(3.0 * t_fl / args.steps / max(tt_s, 1e-9) / 1e12, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(3.0 * t_fl / args.steps / max(tt_s, 1e-9) / (PEAK_F16_TFLOPS * 1e12), 4), "stage_ms": round(tt_s * 1e3, 1),
                "prompts_per_s": round(prompts / max(tt_s, 1e-9)), "gemm_gflop_per_prompt_executed": round(3.0 * t_fl / args.steps / max(prompts, 1) / 1e9, 3),
                "fc1_launches_per_step": t_n // max(args.steps, 1), "fc1_kernel_frac_of_peak": round(t_fl / max(t_ms, 1e-9) / 1e9 / PEAK_F16_TFLOPS, 4)})
        line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_child()
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
