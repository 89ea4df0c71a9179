"""HIP path vs oracle on a real MI355X (run with -m gpu).  Every call goes through the C ABI (scd_amd.ops -> ctypes)."""
import os
from collections import Counter

import numpy as np
import pytest
import torch

from oracle import kmeans_oracle as ko
from oracle import naming_oracle as no
from oracle import transport_oracle as to
from oracle import clip_oracle as co
from oracle import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device; they must not be skipped on the GPU box"
    from scd_amd import ops as o
    return o


def dev(x):
    return torch.as_tensor(x).cuda()


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(128, 256, 128), (256, 512, 256), (512, 256, 512)])
def test_gemm_identity_asymmetric(ops, m, n, k):
    # A = I (padded), asymmetric integer W: catches any row/col or lane-map swap exactly (128- and 256-tile kernels)
    a = torch.zeros(m, k)
    idx = torch.arange(min(m, k))
    a[idx, idx] = 1.0
    w = (torch.arange(n).view(n, 1) * 3 + torch.arange(k).view(1, k) * 7) % 61 - 30.0
    c = ops.gemm_f16(a.half().cuda(), w.half().cuda())
    assert torch.equal(c.float().cpu(), (a @ w.t()))


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 768), (384, 768, 3072), (25216, 2304, 768), (256, 256, 64),
                                   (512, 768, 768), (1024, 2304, 3072), (50432, 768, 768)])
@pytest.mark.parametrize("variant", ["plain", "bias_qgelu", "bias_res", "gelu"])
def test_gemm_matches_fp32(ops, m, n, k, variant):
    if m > 10000 and variant != "bias_res":
        pytest.skip("big shape once")
    g = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).half().cuda()
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).half().cuda()
    bias = torch.randn(n, generator=g).cuda() if variant != "plain" else None
    res = torch.randn(m, n, generator=g).half().cuda() if variant == "bias_res" else None
    act = {"plain": 0, "bias_qgelu": 1, "bias_res": 0, "gelu": 2}[variant]
    c = ops.gemm_f16(a, w, bias, res, act).float()
    ref = a.float() @ w.float().t()
    if bias is not None:
        ref = ref + bias
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res.float()
    err = (c - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err     # fp16 output rounding


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (128, 128, 128)])
def test_gelu_epilogue_within_two_fp16_ulps_of_erf(ops, m, n, k):
    # the exact-GELU epilogue on its own: W = I, so the accumulator IS the fp16 input; every fp16 value of [-12, 12] (and the
    # specials around it) goes through and is compared with erf-GELU evaluated in float64 and rounded once.  256-tile (four-wave
    # kernel, value pairs) and 128-tile (scalar form) kernels.  nn.GELU: gcd/models/vision_transformer.py:48-64.
    from scipy.special import erfc
    vals = np.arange(-(1 << 15), 1 << 15, dtype=np.int32).astype(np.int16).view(np.float16)
    vals = vals[np.isfinite(vals) & (np.abs(vals.astype(np.float32)) <= 12.0)]
    per = m * k
    eye = torch.eye(k)[:n] if n <= k else torch.cat([torch.eye(k), torch.zeros(n - k, k)])
    worst = 0
    for lo in range(0, len(vals), per):
        chunk = np.zeros(per, dtype=np.float16)
        part = vals[lo:lo + per]
        chunk[:len(part)] = part
        a = torch.from_numpy(chunk.reshape(m, k)).cuda()
        c = ops.gemm_f16(a, eye.half().cuda(), torch.zeros(n).cuda(), None, 2).cpu().numpy()[:, :min(n, k)]
        x = chunk.reshape(m, k)[:, :min(n, k)].astype(np.float64)
        ref = np.where(x >= 0, x - 0.5 * x * erfc(x / np.sqrt(2.0)), 0.5 * x * erfc(-x / np.sqrt(2.0))).astype(np.float16)
        d = np.abs(c.view(np.int16).astype(np.int64) - ref.view(np.int16).astype(np.int64))
        d[(c == 0) & (ref == 0)] = 0            # -0 against +0
        worst = max(worst, int(d.max()))
        assert np.abs(c.astype(np.float64) - x * 0.5 * erfc(-x / np.sqrt(2.0))).max() <= 5e-4 * 12    # absolute: fp16 rounding only
    assert worst <= 2, worst


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (768, 512, 192), (2560, 1280, 320), (256, 2304, 64), (5120, 256, 1024), (1792, 768, 768)])
@pytest.mark.parametrize("variant", ["plain", "bias_qgelu", "bias_res"])
def test_gemm_four_wave_edge_shapes(ops, m, n, k, variant):
    # the persistent four-wave kernel at its corners: 1-5 chunks per tile, fewer tiles than CUs, tile counts that are not
    # multiples of 8 (XCD split), one tile column
    g = torch.Generator().manual_seed(7 * m + 3 * n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).half().cuda()
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).half().cuda()
    bias = torch.randn(n, generator=g).cuda() if variant != "plain" else None
    res = torch.randn(m, n, generator=g).half().cuda() if variant == "bias_res" else None
    act = 1 if variant == "bias_qgelu" else 0
    c = ops.gemm_f16(a, w, bias, res, act).float()
    ref = a.float() @ w.float().t()
    if bias is not None:
        ref = ref + bias
    if act == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    if res is not None:
        ref = ref + res.float()
    err = (c - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    assert torch.equal(c, ops.gemm_f16(a, w, bias, res, act).float())      # run-to-run reproducible


# ----------------------------------------------------------------------------------------------- k-means pieces
@pytest.mark.parametrize("n,d,k,seed", [(500, 8, 4, 1), (1500, 32, 10, 2), (3000, 768, 20, 3), (4097, 768, 100, 4),
                                        (2000, 768, 200, 5)])
def test_estep_bit_exact(ops, n, d, k, seed):
    x, y, cent = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 50, noise=0.9)
    rs = np.random.RandomState(seed)
    c = (cent + 0.05 * rs.randn(k, d)).astype(np.float32)
    data = ops.KMeansData(dev(x))
    lab, ref = data.estep(dev(c), return_refined=True)
    olab, omind, _ = ko.estep(x, c)
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), olab)
    assert int(ref.item()) < n                                   # the filter did decide most rows
    d2 = data.rowdist(dev(c), lab).cpu().numpy()
    assert np.array_equal(d2, omind)                             # float32(float64 sum): bit-exact


def test_estep_unstructured_data_takes_refine_path(ops):
    # no cluster structure and near-duplicate centres: margins are tiny, the exact path must carry the result
    rs = np.random.RandomState(0)
    x = rs.randn(3000, 64).astype(np.float32) + 5.0
    c = np.repeat(rs.randn(6, 64).astype(np.float32) + 5.0, 2, axis=0)
    c[1::2] += 1e-6
    c[3] = c[2]                                                   # exact duplicate: tie -> lowest index
    data = ops.KMeansData(dev(x))
    lab, ref = data.estep(dev(c), return_refined=True)
    olab, _, _ = ko.estep(x, c)
    assert np.array_equal(lab.cpu().numpy(), olab)
    assert int(ref.item()) > 0 and 3 not in set(lab.cpu().numpy().tolist())
    assert torch.equal(data.estep(dev(c), expect_few=True), lab)          # tail refine: same decisions


def test_estep_nan_centre_never_wins(ops):
    x, _, cent = synth.clustered_features(1000, 32, 5, seed=8)
    c = cent.copy()
    c[2] = np.nan                                                 # empty cluster after an M-step
    lab = ops.KMeansData(dev(x)).estep(dev(c)).cpu().numpy()
    olab, _, _ = ko.estep(x, c)
    assert np.array_equal(lab, olab) and 2 not in set(lab.tolist())


@pytest.mark.parametrize("n,d,k,seed", [(31, 16, 3, 11), (33, 200, 128, 12), (1000, 300, 1, 13), (2049, 512, 64, 14), (777, 640, 128, 15),
                                        (5000, 768, 128, 16), (1025, 100, 129, 17), (600, 896, 50, 18), (1500, 256, 1000, 19), (700, 128, 2048, 20),
                                        (900, 64, 2049, 21), (1300, 512, 129, 22), (2049, 512, 1000, 23), (515, 448, 2048, 24),
                                        (4000, 512, 300, 25)])
def test_estep_stream_kernel_shapes(ops, n, d, k, seed):
    """Every column-chunk count of estep_stream_kernel (D <= 768: 1..6 x 128), K = 1 / 128, ragged last 32-row unit, the
    multi-pass form (K = 129 / 1000 / 2048: 2 / 8 / 16 passes), the single-pass form of Dp = 512 with 128 < K <= 2048 (estep_rb_kernel:
    K = 129 / 300 / 1000 / 2048, ragged last 256-row block, D = 448 padded to 512) and the shapes outside both (K = 2049, D = 896:
    estep_mfma_kernel); centres are data points, so rows with distance 0 and many small margins (both refine lists) occur."""
    x, y, cent = synth.clustered_features(n, d, max(2, min(k, 20)), seed=seed, center_seed=seed + 7, noise=0.9)
    rs = np.random.RandomState(seed)
    c = x[rs.choice(n, k, replace=k > n)].copy()
    data = ops.KMeansData(dev(x))
    lab, ref = data.estep(dev(c), return_refined=True)
    olab, omind, _ = ko.estep(x, c)
    assert np.array_equal(lab.cpu().numpy().astype(np.int64), olab)
    assert np.array_equal(data.rowdist(dev(c), lab).cpu().numpy(), omind)
    # the same rows re-evaluated in the filter kernel's tail (scd_kmeans_estep_hint) instead of by the refine launch
    lab2, ref2 = data.estep(dev(c), return_refined=True, expect_few=True)
    assert torch.equal(lab2, lab) and int(ref2.item()) == int(ref.item())


def test_estep_stream_kernel_many_units_per_block(ops):
    """n above 256 blocks x 24 units x 32 rows: the grid grows beyond one block per CU (rows-per-block cap)."""
    n, d, k = 200003, 64, 16
    x, y, cent = synth.clustered_features(n, d, k, seed=31, center_seed=32, noise=0.7)
    data = ops.KMeansData(dev(x))
    lab = data.estep(dev(cent)).cpu().numpy()
    assert (lab == y).mean() > 0.99
    rows = np.random.RandomState(1).choice(n, 2048, replace=False)
    rows[:4] = [0, 31, n - 1, n - 33]
    olab, _, _ = ko.estep(x[rows], cent)
    assert np.array_equal(lab[rows], olab)


def test_estep_non_finite_rows_fall_back_to_exact(ops):
    x, y, cent = synth.clustered_features(700, 96, 7, seed=41)
    x[13, 5] = np.nan
    x[200, :] = np.inf
    lab = ops.KMeansData(dev(x)).estep(dev(cent)).cpu().numpy()
    olab, _, _ = ko.estep(x, cent)
    good = np.ones(700, bool); good[[13, 200]] = False
    assert np.array_equal(lab[good], olab[good])
    assert 0 <= lab[13] < 7 and 0 <= lab[200] < 7


@pytest.mark.parametrize("n,d,k", [(1, 8, 1), (1023, 64, 1), (1025, 48, 1000), (3000, 20, 9000), (4096, 768, 3)])
def test_mstep_counting_sort_edges(ops, n, d, k):
    """mstep_hist / scan / scatter: one block, k = 1, more labels than rows, the rocPRIM path (k > 8191), invalid labels."""
    rs = np.random.RandomState(n + k)
    x = rs.randn(n, d).astype(np.float32)
    labels = rs.randint(0, k, size=n).astype(np.int32)
    if n > 10:
        labels[3] = -1                                            # invalid labels are ignored
        labels[7] = k
    sums, counts, _ = ops.kmeans_mstep(dev(x), dev(labels), None, k, 0)
    ok = (labels >= 0) & (labels < k)
    ref = np.zeros((k, d), np.float64)
    np.add.at(ref, labels[ok], x[ok].astype(np.float64))
    assert np.array_equal(counts.cpu().numpy(), np.bincount(labels[ok], minlength=k))
    assert np.allclose(sums.cpu().numpy(), ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("n,d,k", [(5000, 768, 37), (3001, 512, 100), (777, 64, 5), (2000, 130, 9)])
def test_mstep_f16_copy_gives_the_same_partials(ops, n, d, k):
    """scd_kmeans_mstep_f16 on the exact fp16 copy (scd_f16_exact) = scd_kmeans_mstep on the float32 rows: same counts, float64
    sums equal to round-off of the atomic order, centres bit-equal after the single rounding to float32; a matrix with one value
    that fp16 cannot hold is refused."""
    x, y, cent = synth.clustered_features(n, d, k, seed=n)
    x = x.astype(np.float16).astype(np.float32)                   # what an fp16 encoder hands over
    rs = np.random.RandomState(2)
    labels = rs.randint(0, k, size=n).astype(np.int32)
    labels[3] = -1
    xd = dev(x)
    x16 = ops.f16_exact(xd)
    assert x16 is not None and x16.dtype == torch.float16 and torch.equal(x16.float(), xd)
    s32, c32, i32 = ops.kmeans_mstep(xd, dev(labels), dev(cent), k, n // 3)
    s16, c16, i16 = ops.kmeans_mstep(xd, dev(labels), dev(cent), k, n // 3, x16=x16)
    assert torch.equal(c32, c16)
    assert torch.allclose(s32, s16, rtol=1e-13, atol=1e-13) and torch.allclose(i32, i16, rtol=1e-12)
    ce32, _ = ops.kmeans_finalize(s32, c32, dev(cent))
    ce16, _ = ops.kmeans_finalize(s16, c16, dev(cent))
    assert torch.equal(ce32.nan_to_num(7.0), ce16.nan_to_num(7.0))
    x[5, 7] = 1.0 + 2.0 ** -12                                     # not an fp16 value
    assert ops.f16_exact(dev(x)) is None


def test_dist_and_costs(ops):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kmeans_sskm.npz"))
    a, b = g["pd_a"], g["pd_b"]
    data = ops.KMeansData(dev(a))
    d2 = data.dist(dev(b)).cpu().numpy()
    assert np.array_equal(d2, ko.pairwise_distance(a, b))
    assert np.allclose(d2, g["pd_batched"], rtol=2e-6, atol=1e-6)        # the reference's own output
    ds, cost = data.dist(dev(b), sqrt=True, with_cost=True)
    assert np.array_equal(cost.cpu().numpy(), to.int_costs(d2))
    from scd_amd.local_utils.sskm_constrained import pairwise_distance
    out = pairwise_distance(dev(a), dev(b), 100)
    assert out.device.type == "cpu" and np.array_equal(out.numpy(), d2)   # CPU result when batched (:209)


@pytest.mark.parametrize("n,d,k,seed", [(1, 5, 1, 0), (63, 64, 31, 1), (65, 65, 33, 2), (300, 130, 8, 3), (1000, 768, 120, 4), (4097, 512, 100, 5),
                                        (257, 3, 40, 6)])
def test_dist_tile_kernel_ragged_shapes(ops, n, d, k, seed):
    # scd_kmeans_dist at the corners of its tiling: fewer rows than a 64-row block, one row past a block, centre counts that do not fill
    # a wave's eight / a block's 32, row lengths that are not multiples of the 64-column chunk (or of 4): float32(float64 difference
    # form) bit for bit, sqrt and the ConSSKM integer costs from it (sskm_constrained.py:189-224, 277-287)
    rs = np.random.RandomState(seed)
    a = (rs.randn(n, d) * 3.0).astype(np.float32)
    b = (rs.randn(k, d) * 3.0).astype(np.float32)
    if n > 2 and k > 2:
        b[1] = a[2]                                               # an exact zero distance
    data = ops.KMeansData(dev(a))
    d2 = data.dist(dev(b)).cpu().numpy()
    ref = ko.pairwise_distance(a, b)
    assert d2.shape == (n, k) and np.array_equal(d2, ref)
    ds, cost = data.dist(dev(b), sqrt=True, with_cost=True)
    assert np.array_equal(ds.cpu().numpy(), np.sqrt(ref)) and np.array_equal(cost.cpu().numpy(), to.int_costs(ref))


def test_mstep_finalize_and_inertia(ops):
    x, y, cent = synth.clustered_features(5000, 768, 37, seed=9)
    rs = np.random.RandomState(1)
    labels = rs.randint(0, 37, size=5000).astype(np.int32)
    labels[labels == 5] = 6                                       # cluster 5 empty -> NaN centre
    sums, counts, inertia = ops.kmeans_mstep(dev(x), dev(labels), dev(cent), 37, 1234)
    oc, ocnt = ko.mstep(x, labels, 37)
    assert np.array_equal(counts.cpu().numpy(), ocnt)
    c, shift = ops.kmeans_finalize(sums, counts, dev(cent))
    cn = c.cpu().numpy()
    assert np.isnan(cn[5]).all()
    ok = ~np.isnan(oc)
    assert np.array_equal(cn[ok], oc[ok])                         # float32(float64 sum / n): bit-exact
    d = x.astype(np.float64) - cent[labels].astype(np.float64)
    rowd = (d * d).sum(1)
    ref = np.array([rowd[:1234].sum(), rowd[1234:].sum()])
    assert np.allclose(inertia.cpu().numpy(), ref, rtol=1e-12)
    assert np.isnan(shift.item())                                 # NaN centre => NaN shift (never "converged")
    # linearity: the partial sums of two halves add up to the whole
    s1, c1, _ = ops.kmeans_mstep(dev(x[:2500]), dev(labels[:2500]), None, 37, 0)
    s2, c2, _ = ops.kmeans_mstep(dev(x[2500:]), dev(labels[2500:]), None, 37, 0)
    assert torch.allclose(s1 + s2, sums, rtol=1e-13, atol=1e-13) and torch.equal(c1 + c2, counts)


def test_kpp_draw_and_sum(ops):
    rs = np.random.RandomState(3)
    for n in (7, 1000, 95001):
        d2 = (rs.rand(n) ** 3).astype(np.float32)
        d2[rs.randint(0, n, size=max(1, n // 10))] = 0.0
        t = dev(d2)
        assert float(ops.sum_f32(t).item()) == pytest.approx(float(d2.astype(np.float64).sum()), rel=1e-14)
        for r in (0.0, 1e-9, 0.25, 0.5, 0.999999, 1.0):
            idx, _ = ops.kpp_draw(t, r)
            assert int(idx.item()) == ko.kpp_draw(d2, r), (n, r)
    # shard-aware form: two shards reproduce the single-device draw
    d2 = rs.rand(5000).astype(np.float32)
    tot = ops.sum_f32(dev(d2))
    a, b = dev(d2[:2000]), dev(d2[2000:])
    _, pa = ops.kpp_draw(a, 0.5, total=tot, want_idx=False, want_probsum=True)
    for r in (0.1, 0.39, 0.41, 0.9):
        ia, _ = ops.kpp_draw(a, r, total=tot)
        ib, _ = ops.kpp_draw(b, r, total=tot, prefix=pa)
        want = ko.kpp_draw(d2, r)
        got = int(ia.item()) if int(ia.item()) >= 0 else 2000 + int(ib.item())
        assert got == want


def test_min_update_exact(ops):
    x, _, _ = synth.clustered_features(3001, 768, 10, seed=12)
    data = ops.KMeansData(dev(x))
    d2 = torch.full((3001,), 1e30, device="cuda")
    data.min_update(dev(x[17]), d2)
    ref = ko.pairwise_distance64(x, x[17:18])[:, 0].astype(np.float32)
    assert np.array_equal(d2.cpu().numpy(), ref) and d2[17].item() == 0.0


def test_kpp_multi_kernels_equal_single(ops):
    """The lock-step forms (R restarts per launch) give, row by row, the bits of the single-vector kernels."""
    rs = np.random.RandomState(8)
    for n, d, R in ((3001, 768, 3), (95001, 512, 10), (513, 64, 2), (777, 1024, 5)):
        x, _, _ = synth.clustered_features(n, d, 10, seed=n)
        xt = dev(x)
        data = ops.KMeansData(xt)
        pick = rs.randint(0, n, size=R)
        c_new = xt[torch.as_tensor(pick, device="cuda")].contiguous()
        d2 = dev((rs.rand(R, n) * 3).astype(np.float32))
        want = d2.clone()
        for j in range(R):
            data.min_update(c_new[j], want[j])
        ops.min_update_multi(data.x, c_new, d2)
        assert torch.equal(d2, want)
        assert all(d2[j, pick[j]].item() == 0.0 for j in range(R))
        rv = rs.rand(R)
        rv[0] = 0.0
        idx, _ = ops.kpp_draw_multi(d2, rv)
        for j in range(R):
            assert int(idx[j]) == int(ops.kpp_draw(d2[j], rv[j])[0]) == ko.kpp_draw(d2[j].cpu().numpy(), np.float32(rv[j]))
        sums = ops.sum_f32_multi(d2)
        assert all(float(sums[j]) == float(ops.sum_f32(d2[j])) for j in range(R))
        # shard-aware form
        h = n // 3
        a, b = d2[:, :h].contiguous(), d2[:, h:].contiguous()
        _, pa = ops.kpp_draw_multi(a, rv, total=sums, want_idx=False, want_probsum=True)
        ia, _ = ops.kpp_draw_multi(a, rv, total=sums)
        ib, _ = ops.kpp_draw_multi(b, rv, total=sums, prefix=pa)
        for j in range(R):
            got = int(ia[j]) if int(ia[j]) >= 0 else h + int(ib[j])
            assert got == int(idx[j])


@pytest.mark.parametrize("n,d,k,R,exact", [(20011, 512, 40, 10, True), (9000, 768, 25, 3, True), (5000, 96, 12, 4, True), (20011, 512, 30, 10, False),
                                           (3000, 256, 20, 16, True), (7000, 128, 24, 5, True), (4100, 384, 20, 10, True), (2500, 1024, 14, 3, True)])
def test_kpp_seed_lockstep_rounds_in_c(ops, monkeypatch, n, d, k, R, exact):
    """scd_kpp_seed_lockstep (draw, fetch, distance update of every round behind one call; with the exact fp16 copy the update goes
    through the MFMA filter + float64 pass) leaves the bits of the round-by-round calls it replaces: picks, centres and the final
    distance arrays (sskm_constrained.py:28-44 per restart)."""
    x, _, _ = synth.clustered_features(n, d, 12, seed=n + d, noise=0.7)
    if exact:
        x = x.astype(np.float16).astype(np.float32)
    xt = dev(x)
    x16 = ops.f16_exact(xt)
    assert (x16 is not None) == exact
    rs = np.random.RandomState(5)
    first = rs.randint(0, n, size=R)
    rv = dev(rs.rand(k - 1, R).astype(np.float32))
    rv[3, 0] = 0.0
    def start():
        buf = torch.zeros((R, k, d), dtype=torch.float32, device="cuda")
        buf[:, 0] = xt[torch.as_tensor(first, device="cuda")]
        d2 = torch.full((R, n), float("inf"), dtype=torch.float32, device="cuda")
        ops.min_update_multi(xt, buf[:, 0].contiguous(), d2)
        return buf, d2
    # the rounds one call at a time (what KMeansEngine.kpp_lockstep does under a process group)
    buf0, d20 = start()
    picks0 = []
    for t in range(k - 1):
        idx, _ = ops.kpp_draw_multi(d20, rv[t])
        picks0.append(idx)
        rows = xt.index_select(0, idx.clamp(min=0)).contiguous()
        buf0[:, 1 + t] = rows
        if t + 1 < k - 1:
            ops.min_update_multi(xt, rows, d20)
    res = {}
    for filt in ("1", "0"):
        monkeypatch.setenv("SCD_KPP_FILTER", filt)
        buf, d2 = start()
        picks = ops.kpp_seed_lockstep(xt, x16, d2, rv, buf, 1)
        assert torch.equal(picks, torch.stack(picks0)) and torch.equal(buf, buf0) and torch.equal(d2, d20), filt
        assert int((picks < 0).sum()) == 0


def test_incremental_mstep_needs_exact_sums(ops):
    """The incremental M-step (scd_kmeans_lloyd_step_delta) is offered only where every float64 cluster sum is exact: an exact fp16
    copy AND rows * max|x| < 2^29 (scd_f16_exact_max).  Unit-scale features qualify; fp16-exact values of magnitude 32,768 over
    20,000 rows, or an infinity, take the fresh M-step - and the fit still equals the oracle's."""
    from scd_amd.kmeans import KMeansEngine
    x, _, _ = synth.clustered_features(20000, 64, 6, seed=3, noise=0.8)
    xh = dev(x.astype(np.float16).astype(np.float32))
    x16 = ops.f16_exact(xh)
    assert x16 is not None and 0.0 < x16.scd_absmax <= 1.0
    assert ops.LloydBuffers(ops.KMeansData(xh), xh, x16, 6).inc
    big = np.round(x * 3.0) * 16384.0                                   # multiples of 2^14 up to 32,768 (49,152): exact in fp16
    xb = dev(big.astype(np.float32))
    b16 = ops.f16_exact(xb)
    assert b16 is not None and b16.scd_absmax * 20000 >= 2.0 ** 29
    assert not ops.LloydBuffers(ops.KMeansData(xb), xb, b16, 6).inc
    xi = xb.clone()
    xi[5, 7] = float("inf")
    i16 = ops.f16_exact(xi)
    assert i16 is not None and i16.scd_absmax == float("inf") and not ops.LloydBuffers(ops.KMeansData(xi), xi, i16, 6).inc
    eng = KMeansEngine(k=6, max_iterations=10, n_init=2, random_state=0)
    eng.fit(xb)
    okm = ko.K_Means(k=6, max_iterations=10, n_init=2, random_state=0)
    okm.fit(big.astype(np.float32))
    assert np.array_equal(eng.labels_.cpu().numpy(), okm.labels_) and np.array_equal(eng.cluster_centers_.cpu().numpy(), okm.cluster_centers_)


@pytest.mark.parametrize("n,d,R", [(5000, 512, 10), (3001, 768, 4), (2600, 128, 16), (4096, 256, 1)])
def test_kpp_update_filter_equals_tile_kernel(ops, monkeypatch, n, d, R):
    """scd_kpp_update_filter (one seeding round's distance update through the MFMA filter, the call the Python-driven rounds of a
    process group use) leaves the float32 bits of scd_kmeans_min_update_multi, round after round, also for a centre that is NOT a
    row of X (not exact in fp16); and KMeansEngine.kpp_lockstep driven from Python (SCD_KPP_SEED_RUN=0, filter updates from the
    eighth centre on) returns the centres of the C loop (sskm_constrained.py:28-44 per restart)."""
    x, _, _ = synth.clustered_features(n, d, 12, seed=n + d, noise=0.7)
    xt = dev(x.astype(np.float16).astype(np.float32))
    x16 = ops.f16_exact(xt)
    assert ops.UpdateFilter.serves(n, d, R)
    uf = ops.UpdateFilter(x16)
    g = torch.Generator(device="cuda").manual_seed(n)
    d2a = torch.full((R, n), float("inf"), dtype=torch.float32, device="cuda")
    ops.min_update_multi(xt, xt[torch.randint(0, n, (R,), device="cuda", generator=g)].contiguous(), d2a)
    d2b = d2a.clone()
    for t in range(12):
        rows = xt[torch.randint(0, n, (R,), device="cuda", generator=g)].contiguous()
        if t == 5:
            rows = (rows + 1e-3 * torch.randn(rows.shape, device="cuda", generator=g)).contiguous()
        ops.min_update_multi(xt, rows, d2a)
        uf.update(rows, d2b)
        assert torch.equal(d2a, d2b), t
    if R != 10:
        return
    from scd_amd.kmeans import KMeansEngine
    res = []
    for run in ("1", "0"):
        monkeypatch.setenv("SCD_KPP_SEED_RUN", run)
        eng = KMeansEngine(k=24, n_init=R, random_state=0)
        data = eng._be().prepare(xt)
        res.append(eng.kpp_lockstep(data, None, 24, np.random.RandomState(3), R, x16=x16))
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("mixed", [False, True])
def test_kpp_lockstep_equals_sequential_restarts_gpu(ops, monkeypatch, mixed):
    """KMeansEngine on the device: seedings of the n_init restarts drawn in lock-step == one kpp per restart (sskm.py:190-204)."""
    from scd_amd.kmeans import KMeansEngine
    x, y, mask_lab = synth.blob_case(6000, 64, 12, 5)
    xt = dev(x)
    m = torch.as_tensor(mask_lab, device="cuda")
    out = []
    for lock in ("1", "0"):
        monkeypatch.setenv("SCD_KPP_LOCKSTEP", lock)
        km = KMeansEngine(k=12, max_iterations=5, n_init=5, random_state=11)
        if mixed:
            km.fit_mix(xt[~m], xt[m], dev(y[mask_lab]))
        else:
            km.fit(xt)
        out.append((km.labels_.cpu().numpy(), km.cluster_centers_.cpu().numpy(), float(km.inertia_)))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_lloyd_step_full_size_properties(ops):
    """scd_kmeans_lloyd_step at the C2 size (126,976 x 512, K = 100; too large for the oracle), through properties that do not
    depend on the size: its labels are those of the stand-alone E-step, the counts add up to N, the float64 column totals of the
    per-cluster sums equal the column totals of X EXACTLY (fp16-exact rows: every partial sum is an integer multiple of 2^-24
    below 2^29), its centres / statistics are those of the three single calls, and a second step from the new centres keeps
    every label whose row did not move (shift -> 0 at a fixed point)."""
    n, d, k = 126976, 512, 100
    g = torch.Generator(device="cuda").manual_seed(3)
    cen = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
    y = torch.randint(0, k, (n,), device="cuda", generator=g)
    x = (cen[y] + (0.8 / d ** 0.5) * torch.randn(n, d, device="cuda", generator=g)).half().float().contiguous()
    data = ops.KMeansData(x)
    x16 = ops.f16_exact(x)
    assert x16 is not None
    c0 = (cen + 0.02 * torch.randn(k, d, device="cuda", generator=g)).contiguous()      # near the blobs: no cluster runs empty
    buf = ops.LloydBuffers(data, x, x16, k)
    buf.c0.copy_(c0)
    buf.step(buf.c0, buf.c[0], buf.stats[0], False)
    lab = buf.lab32.clone()
    assert torch.equal(lab, data.estep(c0))
    assert int(buf.counts.sum()) == n and torch.equal(buf.counts, torch.bincount(lab.long(), minlength=k))
    assert torch.equal(buf.sums.sum(0), x.double().sum(0))                       # checksum of checksums, exact
    sums, counts, inertia = ops.kmeans_mstep(x, lab, c0, k, 0, x16=x16)
    c1, shift = ops.kmeans_finalize(sums, counts, c0)
    assert torch.equal(buf.c[0], c1) and torch.equal(buf.counts, counts)
    st = buf.stats[0].cpu().numpy()
    assert st[0] == 0.0 and st[1] == pytest.approx(float(inertia[1]), rel=1e-12) and st[2] == pytest.approx(float(shift), rel=1e-12)
    # iterate to a fixed point: labels stop changing, the shift reaches exactly 0, and the step then reproduces itself
    prev, it = lab, 0
    while it < 60:
        buf.step(buf.c[it & 1], buf.c[1 - (it & 1)], buf.stats[1 - (it & 1)], it >= 1)
        cur = buf.lab32.clone()
        it += 1
        if torch.equal(cur, prev):
            break
        prev = cur
    assert it < 60
    buf.step(buf.c[it & 1], buf.c[1 - (it & 1)], buf.stats[1 - (it & 1)], True)
    assert torch.equal(buf.lab32, prev) and float(buf.stats[1 - (it & 1)][2]) == 0.0
    assert torch.equal(buf.c[0], buf.c[1])


# ----------------------------------------------------------------------------------------------- k-means end to end
@pytest.mark.parametrize("tag", ["a", "b", "c", "h", "i"])
def test_sskm_matches_reference_golden(ops, golden, tag):
    """The reference's own K_Means runs (faster_mix_k_means_pytorch.py:91-275, gcd copy): "a".."c" on float32 blobs, "h" / "i" (round
    5) on the same construction rounded to fp16 at the CLIP / DINO widths - the production regime: exact fp16 copy, MFMA filters for the
    seeding and the E-step, the restarts' Lloyd loops in lock-step, the incremental exact M-step.  Labels EQUAL the reference's."""
    from scd_amd.gcd.methods.clustering.faster_mix_k_means_pytorch import K_Means
    from test_oracle_golden import kmeans_golden_case
    g, x, y, mask_lab, n_init = kmeans_golden_case(golden, tag)
    n, d, k, seed = g[tag + "_shape"].tolist()
    u, l, lt = dev(x[~mask_lab]), dev(x[mask_lab]), dev(y[mask_lab])
    km = K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=n_init, random_state=seed, n_jobs=None,
                 pairwise_batch_size=1024)
    km.fit_mix(u, l, lt)
    if tag in ("h", "i"):
        assert km.stats.get("lockstep_fits", 0) >= 1, km.stats      # the fp16-exact regime did take the lock-step C loop
    assert km.labels_.dtype == torch.int64 and km.labels_.is_cuda
    assert np.array_equal(km.labels_.cpu().numpy(), g[tag + "_labels"])             # the reference's own labels
    assert np.allclose(km.cluster_centers_.cpu().numpy(), g[tag + "_centers"], rtol=1e-5, atol=1e-6, equal_nan=True)
    assert abs(float(km.inertia_) - float(g[tag + "_inertia"])) <= 1e-5 * float(g[tag + "_inertia"])
    assert km.n_iter_ == int(g[tag + "_n_iter"])
    # and bit-for-bit against the oracle (same float64 decision semantics)
    okm = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, n_init=n_init, random_state=seed)
    okm.fit_mix(x[~mask_lab], x[mask_lab], y[mask_lab])
    assert np.array_equal(km.labels_.cpu().numpy(), okm.labels_)
    assert np.array_equal(km.cluster_centers_.cpu().numpy(), okm.cluster_centers_, equal_nan=True)
    assert float(km.inertia_) == float(okm.inertia_)
    # plain fit.  In "h" and "i" the winning restart comes out of an iteration whose M-step left clusters EMPTY: the reference's centres of
    # those are NaN, its torch.min then sends every row to the first of them with a NaN inertia, and the restart keeps the best of the
    # iterations up to that one while it runs on to max_iterations - labels, NaN centre rows and n_iter_ are the reference's
    km2 = K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=2, random_state=seed + 1,
                  pairwise_batch_size=512)
    km2.fit(u)
    assert np.array_equal(km2.labels_.cpu().numpy(), g[tag + "_fit_labels"])
    assert np.allclose(km2.cluster_centers_.cpu().numpy(), g[tag + "_fit_centers"], rtol=1e-5, atol=1e-6, equal_nan=True)
    assert abs(float(km2.inertia_) - float(g[tag + "_fit_inertia"])) <= 1e-5 * float(g[tag + "_fit_inertia"])
    okm2 = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, n_init=2, random_state=seed + 1)
    okm2.fit(x[~mask_lab])
    assert np.array_equal(km2.labels_.cpu().numpy(), okm2.labels_) and km2.n_iter_ == okm2.n_iter_
    assert np.array_equal(km2.cluster_centers_.cpu().numpy(), okm2.cluster_centers_, equal_nan=True)
    if tag == "h":      # the restart that emptied two clusters in its second iteration still wins with that iteration
        assert bool(np.isnan(g[tag + "_fit_centers"]).any()) and km2.n_iter_ == 10
    if tag == "i":      # the restart that died loses to the one that converged (this build's earlier rule - a NaN centre simply never
        assert not np.isnan(g[tag + "_fit_centers"]).any() and km2.n_iter_ == 3      # wins a row - let it run on and win with 1393.6)


def _record_transport(monkeypatch):
    """Record every (cost, size bounds, labels, total) the constrained E-step hands to scd_transport_solve."""
    from scd_amd import ops as o
    calls = []
    real = o.transport_solve

    def spy(cost, size_min, size_max):
        lab, tot = real(cost, size_min, size_max)
        calls.append((np.array(cost, copy=True), size_min, size_max, lab.copy(), tot))
        return lab, tot
    real_batch = o.transport_solve_batch

    def spy_batch(costs, size_min, size_max, threads=None, labels_out=None):
        labs, tots = real_batch(costs, size_min, size_max, threads=threads, labels_out=labels_out)
        for b in range(len(tots)):
            calls.append((np.array(costs[b], copy=True), size_min, size_max, labs[b].copy(), int(tots[b])))
        return labs, tots
    monkeypatch.setattr(o, "transport_solve", spy)
    monkeypatch.setattr(o, "transport_solve_batch", spy_batch)
    calls_real.append(real)
    return calls


calls_real = []          # the un-patched solver (the checks below must not feed the recorder they iterate over)


def _unique_optimum(cost, smin, smax, labels, ops):
    """True when random tie-breaking perturbations of the costs (cost * 1024 + r, r < 8) AND their complements (cost * 1024 + 7 - r: every
    tie broken the other way round) all give `labels` back: the optimum is then unique (an alternative optimum survives a seed only where
    the perturbations of its moves cancel exactly, both ways), so every exact solver must return these labels."""
    for seed in range(3):
        r = np.random.RandomState(100 + seed).randint(0, 8, size=cost.shape)
        for pert in (r, 7 - r):
            lab, _ = calls_real[-1]((cost.astype(np.int64) * 1024 + pert).astype(np.int32), smin, smax)
            if not np.array_equal(lab, labels):
                return False
    return True


def test_constrained_matches_reference_golden(ops, golden, monkeypatch):
    from scd_amd.local_utils.sskm_constrained import K_Means, _labels_constrained
    g = golden("kmeans_constrained.npz")
    n, d, k, seed = g["m_shape"].tolist()
    x, y, mask_lab = synth.blob_case(n, d, k, seed)
    calls = _record_transport(monkeypatch)
    km = K_Means(k=k, tolerance=1e-4, max_iterations=5, init="k-means++", size_min=30, size_max=80, n_init=2,
                 random_state=5, n_jobs=None, pairwise_batch_size=128)
    km.fit_mix(dev(x[~mask_lab]), dev(x[mask_lab]), dev(y[mask_lab]))
    lab = km.labels_.cpu().numpy()
    cnt = np.bincount(lab[mask_lab.sum():], minlength=k)
    assert cnt.min() >= 30 and cnt.max() <= 80
    assert abs(float(km.inertia_) - float(g["m_inertia"])) <= 1e-3 * float(g["m_inertia"])
    # every flow problem of the run (one per Lloyd iteration and restart) was solved EXACTLY: feasible, total cost equal to the
    # LP optimum, no improving move left; where the optimum is unique the labels are the LP's labels too.  (The reference's
    # OR-Tools labels are unpinned - third-party solver absent - and its optimum is not unique under integer cost ties.)
    assert len(calls) >= 2
    n_unique = 0
    for cost, smin, smax, labs, tot in list(calls):
        ok, tot_chk = to.check_assignment(cost, labs, smin, smax)
        lp_lab, lp_tot = to.solve_lp(cost, smin, smax)
        assert ok and tot == tot_chk == lp_tot and to.check_optimal(cost, labs, smin, smax)
        if _unique_optimum(cost, smin, smax, labs, ops):
            n_unique += 1
            assert np.array_equal(labs, lp_lab)
    # the 6-point docstring example of the vendored estimator (k_means_constrained_.py:777-793)
    km6 = K_Means(k=2, size_min=2, size_max=5, random_state=0, n_init=10, max_iterations=100)
    km6.fit(dev(g["kat_x"]))
    assert np.array_equal(km6.labels_.cpu().numpy(), g["kat_labels"])
    # raw assignment: optimal total cost equals the reference run's
    dist = np.zeros(g["a_d2"].shape[0], dtype=np.float32)
    lab2, inertia = _labels_constrained(None, None, np.sqrt(g["a_d2"]), 30, 80, dist)
    cost = to.int_costs(g["a_d2"])
    assert int(cost[np.arange(len(lab2)), lab2].sum()) == int(g["a_total"])
    with pytest.raises(Exception, match="min cost flow"):
        _labels_constrained(None, None, np.ones((5, 2), dtype=np.float32), 3, 5, np.zeros(5, dtype=np.float32))


def test_constrained_tie_free_instance_labels_equal_oracle(ops, monkeypatch):
    """Label parity where it is well defined: features scaled by 400 spread the integer costs round(1000 * dist) over ~1e6
    values, so every flow problem of the run has a unique optimum (checked by perturbation) and the HIP path must return the
    oracle's (LP) labels on every iteration - hence the same centres, inertia and final labels."""
    from scd_amd.local_utils.sskm_constrained import K_Means
    x, y, mask_lab = synth.blob_case(420, 12, 5, 33)
    x = (x * 400.0).astype(np.float32)
    calls = _record_transport(monkeypatch)
    km = K_Means(k=5, tolerance=1e-4, max_iterations=6, init="k-means++", size_min=60, size_max=75, n_init=2, random_state=3,
                 pairwise_batch_size=128)
    km.fit_mix(dev(x[~mask_lab]), dev(x[mask_lab]), dev(y[mask_lab]))
    for cost, smin, smax, labs, tot in list(calls):
        assert _unique_optimum(cost, smin, smax, labs, ops)
        assert np.array_equal(labs, to.solve_lp(cost, smin, smax)[0])
    okm = to.K_Means(k=5, tolerance=1e-4, max_iterations=6, size_min=60, size_max=75, n_init=2, random_state=3)
    okm.fit_mix(x[~mask_lab], x[mask_lab], y[mask_lab])
    assert np.array_equal(km.labels_.cpu().numpy(), okm.labels_)
    assert np.array_equal(km.cluster_centers_.cpu().numpy(), okm.cluster_centers_)
    assert float(km.inertia_) == float(okm.inertia_)
    cnt = np.bincount(km.labels_.cpu().numpy()[mask_lab.sum():], minlength=5)
    assert cnt.min() >= 60 and cnt.max() <= 75 and ((cnt == 60).any() or (cnt == 75).any())      # the bounds bind


def test_constrained_c3_size(ops, monkeypatch):
    """BASELINE configs[2] (Stanford Dogs pt-sup): N_l = 3k + N_u = 9k rows, D = 768, K = 120, size_min / size_max = 50 / 1000
    (main_ptsup.py:239-240).  Every flow problem solved exactly (feasible + no negative residual cycle), sizes within bounds,
    labelled rows keep their classes."""
    from scd_amd.local_utils.sskm_constrained import K_Means
    n, d, k = 12000, 768, 120
    x, y, _ = synth.clustered_features(n, d, k, seed=61, center_seed=62, noise=0.9)
    mask_lab = (y < k // 2) & (np.random.RandomState(63).rand(n) < 0.5)
    order = np.concatenate([np.nonzero(mask_lab)[0], np.nonzero(~mask_lab)[0]])
    x, y = x[order], y[order]
    n_l = int(mask_lab.sum())
    calls = _record_transport(monkeypatch)
    km = K_Means(k=k, tolerance=1e-4, max_iterations=3, init="k-means++", size_min=50, size_max=1000, n_init=1, random_state=0,
                 pairwise_batch_size=1024)
    km.fit_mix(dev(x[n_l:]), dev(x[:n_l]), dev(y[:n_l]))
    lab = km.labels_.cpu().numpy()
    assert len(calls) >= 1 and calls[0][0].shape == (n - n_l, k)
    for cost, smin, smax, labs, tot in list(calls):
        ok, tot_chk = to.check_assignment(cost, labs, smin, smax)
        assert ok and tot == tot_chk and to.check_optimal(cost, labs, smin, smax)
    cnt = np.bincount(lab[n_l:], minlength=k)
    assert cnt.min() >= 50 and cnt.max() <= 1000
    classes = np.unique(y[:n_l])
    assert np.array_equal(lab[:n_l], np.searchsorted(classes, y[:n_l]))
    assert (lab[n_l:][y[n_l:] < k // 2] == y[n_l:][y[n_l:] < k // 2]).mean() > 0.9     # old classes land on their class ids


# ----------------------------------------------------------------------------------------------- similarity / vote
def test_sim_topk_golden_and_oracle(ops, golden):
    g = golden("naming.npz")
    f, w = g["tk_x"].astype(np.float16), g["tk_w"].astype(np.float16)
    wt = ops.transpose_f16(dev(w))
    assert torch.equal(wt.cpu(), torch.from_numpy(w).t())
    for mode in ("softmax", "raw"):
        idx, val = ops.sim_topk(dev(f), wt, 5, mode)
        oi, ov = no.sim_topk(f, w, 5, mode)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.allclose(val.cpu().numpy(), ov, rtol=2e-4, atol=1e-6)
    # fp32 features of the golden run rounded to fp16 still give the reference's own top-1 nearly everywhere
    idx, _ = ops.sim_topk(dev(f), wt, 5, "raw")
    assert (idx[:, 0].cpu().numpy() == g["tk_idx_ptsup"][:, 0]).mean() > 0.99


def test_sim_topk_equals_the_reference_blocks_on_f16_inputs(ops, golden):
    """The reference's own top-k blocks (main_unsup.py:504-531 softmax, main_ptsup.py:526-545 raw logits), run at fixture-generation
    time on fp16-exact features and names (tests/golden/topk_f16.npz, d = 512: the row-block kernel), against scd_sim_topk on the same
    values: the reference's index lists on all 2,100 rows (its float32 ranking and the exact one agree on this fixture, near-ties of
    1e-5 logit units included), same values."""
    from test_oracle_golden import topk16_case
    g, f16, w16 = topk16_case(golden)
    wt = ops.transpose_f16(dev(w16))
    for mode, ikey, vkey, tol in (("softmax", "idx_unsup", "val_unsup", dict(rtol=2e-4, atol=1e-7)),
                                  ("raw", "idx_ptsup", "val_ptsup", dict(rtol=1e-5, atol=1e-4))):
        idx, val = ops.sim_topk(dev(f16), wt, 5, mode)
        assert np.array_equal(idx.cpu().numpy(), g[ikey])
        assert np.allclose(val.cpu().numpy(), g[vkey], **tol)


@pytest.mark.parametrize("n,v,d,k", [(300, 21000, 512, 5), (129, 1000, 512, 3), (1000, 100, 512, 1), (64, 37, 64, 8),
                                     (33100, 1100, 512, 3), (66000, 1031, 512, 3), (300, 1031, 512, 2)])
def test_sim_topk_shapes(ops, n, v, d, k):
    """Ragged N and V, every list size, d < 512 (tile kernel), and both launch forms of the row-block kernel: up to 128 row blocks
    in the (partial) last round are served by two blocks each, one per vocabulary half, merged before the refine pass
    (sim_split_merge_kernel: n = 300 and 129 with V >= 1024 names; n = 66,000 = 256 whole-vocabulary blocks + 2 split ones); n = 1000 x
    100 names and n = 33,100 (130 row blocks: more than half a round) take the whole vocabulary in every block."""
    rs = np.random.RandomState(n + v)
    f = (rs.randn(n, d) / np.sqrt(d)).astype(np.float16)
    w = (rs.randn(d, v) / np.sqrt(d)).astype(np.float16)
    w[:, 5] = w[:, 3]                                             # duplicate names: exact ties -> lower index first
    wt = ops.transpose_f16(dev(w))
    idx, val, fb = ops.sim_topk(dev(f), wt, k, "raw", return_fallback=True)
    oi, ov = no.sim_topk(f, w, k, "raw")
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.allclose(val.cpu().numpy(), ov, rtol=1e-6, atol=1e-5)
    assert int(fb.item()) <= n // 20
    a, av = ops.sim_argmax(dev(f), wt)
    assert np.array_equal(a.cpu().numpy(), oi[:, 0])


def test_sim_topk_vocabulary_norm_is_computed_once_and_follows_the_tensor(ops):
    """ops.sim_topk hands the vocabulary's max ||w||^2 (the data-dependent part of the filter's error bound) to scd_sim_topk_prenorm.
    It is computed per call unless the caller froze the tensor (ops.freeze_vocab: once, reused); an in-place change of a frozen tensor
    (torch's `_version`) ends the promise and gets a fresh one - a stale, too small norm would certify rows it must not.  Indices
    equal the oracle's before and after the change, and equal the plain scd_sim_topk call's (main_unsup.py:504-531)."""
    n, v, d, k = 700, 5000, 512, 3
    rs = np.random.RandomState(17)
    f = (rs.randn(n, d) / np.sqrt(d)).astype(np.float16)
    w = (rs.randn(d, v) / np.sqrt(d)).astype(np.float16)
    wt = ops.transpose_f16(dev(w))
    assert ops.vocab_norm(wt) is not ops.vocab_norm(wt)            # not frozen: a fresh norm per call
    ops.freeze_vocab(wt)
    idx1, val1 = ops.sim_topk(dev(f), wt, k, "softmax")
    norm1 = ops.vocab_norm(wt)
    assert ops.vocab_norm(wt) is norm1                             # frozen: computed once
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                  # a consumer on another stream waits for the norm's event
        idx_s, _ = ops.sim_topk(dev(f), wt, k, "softmax")
    side.synchronize()
    assert torch.equal(idx_s, idx1)
    idx2, val2 = ops.sim_topk(dev(f), wt, k, "softmax")
    oi, ov = no.sim_topk(f, w, k, "softmax")
    assert np.array_equal(idx1.cpu().numpy(), oi) and torch.equal(idx1, idx2) and torch.equal(val1, val2)
    w32 = float(norm1.view(torch.float32).item())
    assert abs(w32 - float((w.astype(np.float64) ** 2).sum(0).max())) <= 1e-3 * w32
    # in place: 40 x larger names in the second half - with the old norm the bound would be 40 x too small
    wt[v // 2:] *= 40.0
    w2 = w.copy()
    w2[:, v // 2:] = (w2[:, v // 2:].astype(np.float32) * 40.0).astype(np.float16)
    assert torch.equal(wt.cpu(), torch.from_numpy(w2).t())
    idx3, _ = ops.sim_topk(dev(f), wt, k, "raw")
    norm3 = ops.vocab_norm(wt)
    assert norm3 is not norm1 and float(norm3.view(torch.float32).item()) > 1000 * w32
    oi3, _ = no.sim_topk(f, w2, k, "raw")
    assert np.array_equal(idx3.cpu().numpy(), oi3)


def test_ablation_variables_change_nothing_in_the_default_library():
    """SCD_GEMM_X / SCD_SIM_X / SCD_ATTN_X / SCD_ESTEP_DBG removed kernel pieces (wrong results) in rounds 1-2; the default build
    ignores them: a child process with all of them set returns the oracle's top-k and labels."""
    import subprocess
    import sys
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from scd_amd import ops
from oracle import naming_oracle as no, kmeans_oracle as ko, synth
rs = np.random.RandomState(5)
f = (rs.randn(300, 512) / 22.6).astype(np.float16); w = (rs.randn(512, 2100) / 22.6).astype(np.float16)
wt = ops.transpose_f16(torch.from_numpy(w).cuda())
idx, val = ops.sim_topk(torch.from_numpy(f).cuda(), wt, 3, "raw")
oi, ov = no.sim_topk(f, w, 3, "raw")
assert np.array_equal(idx.cpu().numpy(), oi)
x, y, c = synth.clustered_features(3000, 768, 20, seed=3, center_seed=4, noise=0.8)
lab = ops.KMeansData(torch.from_numpy(x).cuda()).estep(torch.from_numpy(c).cuda())
assert np.array_equal(lab.cpu().numpy(), ko.estep(x, c)[0])
print("same")
""" % (ROOT,)
    env = dict(os.environ, SCD_GEMM_X="16", SCD_SIM_X="1", SCD_ATTN_X="4", SCD_ESTEP_DBG="16", SCD_ESTEP_REFINE_SPLIT="2", SCD_SIM_RB="1")
    env.pop("SCD_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("n", [3, 100, 300])
@pytest.mark.parametrize("mode", ["raw", "softmax"])
def test_sim_topk_fallback_rows_exact(ops, n, mode):
    """Rows whose top-k cannot be certified from the candidate lists (here: 40 names that differ by single fp16 ulps, two of them
    identical) go through the exact float64 pass: the first 256 spread over the whole chip in chunks of 128 names
    (sim_exact_chunk / merge), the rest (n = 300) one block per row.  Same indices as the float64 oracle, ties to the lower index."""
    rs = np.random.RandomState(17 + n)
    d, v, k = 512, 3000, 5
    base = (rs.randn(d) / np.sqrt(d)).astype(np.float16)
    w = (rs.randn(d, v) / np.sqrt(d)).astype(np.float16)
    for j in range(40):
        col = base.copy()
        pos = rs.randint(0, d, size=3)
        col[pos] = np.nextafter(col[pos], np.float16(10), dtype=np.float16)
        w[:, 100 + 7 * j] = col
    w[:, 100 + 7 * 13] = w[:, 100 + 7 * 2]                       # an exact tie inside the cluster
    f = (base[None, :].astype(np.float32) * 2 + rs.randn(n, d) * 0.01).astype(np.float16)
    wt = ops.transpose_f16(dev(w))
    idx, val, fb = ops.sim_topk(dev(f), wt, k, mode, return_fallback=True)
    oi, ov = no.sim_topk(f, w, k, mode)
    assert int(fb.item()) >= min(n, 3)                            # the construction does defeat the certificate
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.allclose(val.cpu().numpy(), ov, rtol=2e-4, atol=1e-6)


def test_l2norm_and_gather(ops):
    rs = np.random.RandomState(1)
    x = rs.randn(77, 512).astype(np.float32) * 3
    out = ops.l2norm_rows(dev(x)).cpu().numpy()
    assert np.allclose(out, x / np.linalg.norm(x, axis=1, keepdims=True), atol=1e-6)
    xh = dev(x).half()
    outh = ops.l2norm_rows(xh).float().cpu().numpy()
    assert np.allclose(np.linalg.norm(outh, axis=1), 1.0, atol=2e-3)
    wt = dev(rs.randn(300, 512).astype(np.float16))
    idx = dev(np.array([5, 0, 299, 5]))
    assert torch.equal(ops.gather_rows_f16(wt, idx), wt[idx])


def test_vote_hist_matches_counter(ops):
    rs = np.random.RandomState(2)
    n, k, v = 20000, 23, 500
    preds = rs.randint(0, k + 3, size=n)                          # some rows belong to clusters we do not ask for
    names = (rs.zipf(1.5, size=(n, 5)) % v).astype(np.int64)
    clusters = [c for c in range(k) if c % 5 != 4]
    known = [0, 1, 7]
    for kn, top_k, m in ((None, 5, 20), (known, 3, 4)):
        keys, counts = ops.vote_hist(dev(names), top_k, dev(preds), clusters, m, kn)
        keys, counts = keys.cpu().numpy(), counts.cpu().numpy()
        ref = no.cluster_counters(names, preds, clusters, top_k, known=kn)
        for i, c in enumerate(clusters):
            mc = ref[c].most_common(m)
            got = [(int(a), int(b)) for a, b in zip(keys[i], counts[i]) if a >= 0]
            assert got == [(int(a), int(b)) for a, b in mc], c


def test_vote_loops_match_reference_traces(ops, golden):
    from scd_amd import naming
    g = golden("naming.npz")
    f, w = g["tk_x"], g["tk_w"]
    nouns = synth.nouns_list(w.shape[1])
    wt = ops.transpose_f16(dev(w.astype(np.float16)))
    fh = dev(f.astype(np.float16))
    # inputs rounded to fp16 (the dtype the GPU path stores): replay the oracle on the same rounded inputs ...
    k, topk, ncv, ncl = g["vu_cfg"].tolist()
    idx, _ = naming.full_vocab_topk(fh, None, 5, True, wt=wt)
    oidx, _ = no.sim_topk(f.astype(np.float16), w.astype(np.float16), 5, "softmax")
    assert np.array_equal(idx.cpu().numpy(), oidx)
    cand, up, tr = naming.vote_loop_unsup(idx, g["vu_preds0"], fh, wt, nouns, k, ncv, ncl)
    otr = no.vote_loop_unsup(oidx, g["vu_preds0"], f.astype(np.float16), w.astype(np.float16), nouns, k, topk, ncv, ncl)
    assert len(tr) == len(otr)
    for a, b in zip(tr, otr):
        for key in ("voted", "ind", "cand", "u_preds"):
            assert np.array_equal(a[key], b[key]), key
    # ... and the reference's own trace (fp32 inputs) has the same names at convergence
    last = int(g["vu_iters"]) - 1
    assert set(tr[-1]["cand"].tolist()) == set(g["vu_cand_%d" % last].tolist())
    assert (tr[-1]["u_preds"] == g["vu_preds_%d" % last]).mean() > 0.995


# ----------------------------------------------------------------------------------------------- encoders
def _cos(a, b):
    a, b = a.double(), b.double()
    return torch.nn.functional.cosine_similarity(a, b, dim=-1)


@pytest.mark.parametrize("layers", [2, 12])
def test_clip_towers_match_oracle(ops, layers):
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=layers, t_layers=layers))
    sd16 = {k: (v.half().float() if v.dim() >= 2 and "positional" not in k and "class_emb" not in k else v) for k, v in sd.items()}
    model = CLIP(sd).cuda().eval()
    img = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(78))
    out = model.encode_image(img.cuda()).float().cpu()
    ref = co.clip_encode_image(sd16, img.half().float())
    assert out.shape == (5, 512)
    assert _cos(out, ref).min().item() > 1 - 1e-3                 # north-star tolerance: cosine within 1e-3 (fp16)
    assert (out - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    tok = torch.zeros(4, 77, dtype=torch.int32)
    g = torch.Generator().manual_seed(79)
    for i, ln in enumerate((3, 8, 20, 75)):
        tok[i, 0] = 49406
        tok[i, 1:1 + ln] = torch.randint(1, 49405, (ln,), generator=g, dtype=torch.int32)
        tok[i, 1 + ln] = 49407
    tout = model.encode_text(tok.cuda()).float().cpu()
    tref = co.clip_encode_text(sd16, tok.long())
    assert _cos(tout, tref).min().item() > 1 - 1e-3


def test_encoder_batch_invariance(ops):
    # token rows (not images) are padded to the GEMM tile: an image's features must not depend on how many images share the
    # batch, nor on the zero rows behind them (every row's K loop runs in the same order whatever M is)
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    sd = W.synthetic_clip_state_dict(seed=3, cfg=dict(v_layers=2, t_layers=2))
    model = CLIP(sd).cuda().eval()
    img = torch.randn(13, 3, 224, 224, generator=torch.Generator().manual_seed(5)).cuda()
    all13 = model.encode_image(img).float().cpu()
    one = model.encode_image(img[:1]).float().cpu()
    five = model.encode_image(img[4:9]).float().cpu()
    assert torch.equal(one[0], all13[0])
    assert torch.equal(five, all13[4:9])


@pytest.mark.parametrize("kind", ["clip", "dino"])
def test_encoder_large_launch_equals_small_batches(ops, kind):
    """The launch sizes the bench runs at - several rounds of tiles per GEMM, non-temporal C stores (2 M N > 64 MB: from 53 images on
    for fc1), QKV as one n-group - against small batches of the same images, bit for bit: 1,400 images in one call (two 665-image row
    groups + a ragged rest) = the same images 13 at a time.  The small-batch tests never reach those kernels' big-launch paths (round 4:
    a store-data hazard inside inline asm corrupted rows of the large launches only, and only the bench noticed)."""
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP, DinoViT
    if kind == "clip":
        model = CLIP(W.synthetic_clip_state_dict(seed=3, cfg=dict(v_layers=2, t_layers=1))).cuda().eval()
        enc = lambda x: model.encode_image(x)
    else:
        model = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=2)).cuda()
        enc = lambda x: model(x)
    g = torch.Generator(device="cuda").manual_seed(9)
    img = torch.randn(1400, 3, 224, 224, generator=g, device="cuda", dtype=torch.float16)
    big = enc(img).float()
    assert bool(torch.isfinite(big).all())
    for s0 in (0, 652, 665, 1317, 1387):
        small = enc(img[s0:s0 + 13]).float()
        assert torch.equal(small, big[s0:s0 + 13]), s0


def test_attention_t197_specialisation_is_bit_identical():
    """At T = 197 (both ViT-B/16 towers) `attention_persist_kernel<true>` leaves out what the padding to 224 keys costs: 12 of the last
    key block's 16 exponentials per lane, its second pair of P V MFMAs, the ring fills of K rows >= 200 / V rows >= 208 (whose LDS rows
    then hold stale data that must never reach a result).  A child process with SCD_ATTN_T197=0 runs the generic kernel: the CLIP
    (image) and DINO features of the same images are the same bits, for a launch of several items per CU and a ragged small one."""
    import subprocess
    import sys
    import tempfile
    code = """
import sys, torch
sys.path.insert(0, %r)
from scd_amd.clip import weights as W
from scd_amd.clip.model import CLIP, DinoViT
clip = CLIP(W.synthetic_clip_state_dict(seed=3, cfg=dict(v_layers=3, t_layers=1))).cuda().eval()
dino = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=3)).cuda()
g = torch.Generator(device="cuda").manual_seed(11)
img = torch.randn(330, 3, 224, 224, generator=g, device="cuda", dtype=torch.float16)
out = {"clip": clip.encode_image(img).cpu(), "dino": dino(img).cpu(), "clip7": clip.encode_image(img[:7]).cpu()}
assert all(bool(torch.isfinite(v.float()).all()) for v in out.values())
torch.save(out, sys.argv[1])
print("done")
""" % (ROOT,)
    outs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("0", "1"):
            env = dict(os.environ, SCD_ATTN_T197=mode)
            env.pop("SCD_HIP_LIB", None)
            path = os.path.join(tmp, "f%s.pt" % mode)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "done" in r.stdout, r.stderr[-2000:]
            outs[mode] = torch.load(path)
    for key in ("clip", "dino", "clip7"):
        a, b = outs["0"][key], outs["1"][key]
        assert a.dtype == b.dtype and torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)), key


def test_encoder_front_end_variants_are_bit_identical(ops, tmp_path):
    """Round 6's front end - the patch GEMM gathering its operand from the fp16 image (gemm_w4_kernel<..., IMG>) and the token assembly
    with four images of a token position per wave - against the round-5 kernels it replaces (SCD_PATCH_FROM_IMAGE=0: im2col + plain
    GEMM; SCD_ASSEMBLE_ROWS=1: one row per wave): the CLIP and DINO features of 3 / 257 / 1,000 images (one row tile with padding rows,
    many tiles per block, a partly padded last tile) are equal bit for bit.  The switches are read once per process, hence the child
    processes (main_unsup.py:114-147)."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "patch_img_check.py")
    outs = {}
    for tag, extra in (("new", {}), ("im2col", {"SCD_PATCH_FROM_IMAGE": "0"}), ("rows1", {"SCD_ASSEMBLE_ROWS": "1"}), ("attn_block", {"SCD_ATTN_SHORT": "0"})):
        env = dict(os.environ, **extra)
        env.pop("SCD_HIP_LIB", None)
        outs[tag] = str(tmp_path / ("feat_%s.pt" % tag))
        r = subprocess.run([sys.executable, tool, "save", outs[tag], "--text"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
    a = torch.load(outs["new"])
    for tag in ("im2col", "rows1", "attn_block"):       # (attn_block: the text tower's short-context attention, a block per item instead of a wave)
        b = torch.load(outs[tag])
        assert set(a) == set(b) and len(a) == 8
        for key in a:
            assert torch.isfinite(a[key].float()).all() and torch.equal(a[key], b[key]), (tag, key)


@pytest.mark.parametrize("longest", [9, 32, 33, 64, 65, 76])
def test_text_tower_trimmed_context_is_bit_identical(ops, longest):
    """encode_text reads only the EOT position of a causal tower (clip model.py encode_text): computing the first
    ctx_len > max EOT position token positions instead of all 77 gives the same bits (scd_clip_encode_text_len), through the
    C ABI with device ids and through CLIP.encode_text with host ids (trimmed automatically)."""
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    sd = W.synthetic_clip_state_dict(seed=4, cfg=dict(v_layers=1, t_layers=3))
    model = CLIP(sd).cuda().eval()
    rs = np.random.RandomState(longest)
    b = 37
    tok = np.zeros((b, 77), dtype=np.int32)
    lens = rs.randint(2, longest + 1, size=b)
    lens[0] = longest                                      # EOT at position `longest`: ctx_len = longest + 1
    for i in range(b):
        tok[i, 0] = 49406
        tok[i, 1:lens[i]] = rs.randint(1, 49405, size=lens[i] - 1)
        tok[i, lens[i]] = 49407
    t_host = torch.from_numpy(tok)
    full = model.encode_text(t_host.cuda())                # device ids: all 77 positions
    trimmed = model.encode_text(t_host)                    # host ids: ctx_len = longest + 1
    explicit = model.encode_text(t_host.cuda(), ctx_len=min(77, longest + 3))
    assert torch.equal(full, trimmed) and torch.equal(full, explicit)
    assert bool(torch.isfinite(full.float()).all())


def test_dino_tower_matches_oracle(ops):
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import DinoViT
    sd = W.synthetic_dino_state_dict(seed=1, layers=12)
    sd16 = {k: (v.half().float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k else v) for k, v in sd.items()}
    img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    out = DinoViT(sd).cuda()(img.cuda()).cpu()
    ref = co.dino_forward(sd16, img.half().float())
    assert out.shape == (3, 768) and _cos(out, ref).min().item() > 1 - 1e-3


def test_dino_features_give_the_oracle_features_labels(ops):
    """The DINO tower runs with fp16 activations where the reference runs fp32 (vision_transformer.py:135-219); what the features
    are USED for is clustering (main_unsup.py:339-350).  On a class-structured synthetic image set the semi-supervised K-Means
    labels computed from the HIP features equal the labels computed from the fp32 oracle's features, row for row, and the two
    feature sets agree to cos > 1 - 1e-3 on every image."""
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import DinoViT
    from scd_amd.kmeans import KMeansEngine
    n_cls, per = 6, 12
    sd = W.synthetic_dino_state_dict(seed=1, layers=12)
    sd16 = {k: (v.half().float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(123)
    base = torch.randn(n_cls, 3, 224, 224, generator=g)
    y = np.repeat(np.arange(n_cls), per)
    img = (base[torch.from_numpy(y)] + 0.6 * torch.randn(n_cls * per, 3, 224, 224, generator=g)).half().float()
    hip = torch.nn.functional.normalize(DinoViT(sd).cuda()(img.cuda()).float(), dim=-1).cpu()
    ref = torch.nn.functional.normalize(torch.cat([co.dino_forward(sd16, img[i:i + 12]) for i in range(0, len(img), 12)]).float(), dim=-1)
    assert _cos(hip, ref).min().item() > 1 - 1e-3
    rs = np.random.RandomState(3)
    mask_lab = (y < n_cls // 2) & (rs.rand(len(y)) < 0.5)
    labels = []
    for feats in (hip, ref):
        km = KMeansEngine(k=n_cls, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0)
        km.fit_mix(feats[~mask_lab].cuda(), feats[mask_lab].cuda(), torch.from_numpy(y[mask_lab]).cuda())
        labels.append(km.labels_.cpu().numpy())
    assert np.array_equal(labels[0], labels[1])
    acc = (labels[0][int(mask_lab.sum()):] == y[~mask_lab]).mean()      # and the clustering is the planted one (cluster ids of the
    assert acc > 0.45                                                  # labelled half are class ids; the novel half is permuted)


def _assert_pathologies(st, sumsq_bits=43):
    """The planted pathologies are there (outlier channels >= 50 x the median magnitude, attention logits of several tens, a
    saturated MLP channel) and inside what the kernels' number formats hold: fp16 activations, and the LayerNorm row statistics
    the proj / fc2 epilogues accumulate as 64-bit fixed point (sum in 2^-24 units, sum of squares in 2^-20 units: gemm.hip LN = 2)."""
    assert st["max_abs"] >= 50 * st["median_abs"], st
    assert st["max_logit"] >= 25, st
    assert st["max_hidden"] >= 10, st
    assert st["max_abs"] < 65504 / 8, st
    assert st["max_sumsq"] < 2.0 ** sumsq_bits and st["max_abs_sum"] < 2.0 ** 39, st     # x 2^20 / x 2^24 stay below 2^63


@pytest.mark.parametrize("tower", ["clip_image", "clip_text", "dino"])
def test_towers_with_outlier_weights_match_oracle(ops, tower):
    """The offline surrogate for the real checkpoints (main_unsup.py:237 clip.load, :241 dino): seeded weights carrying a trained
    ViT's pathologies - outlier residual channels, LayerNorm gains over 2.6 decades, sharp attention heads, bias spikes
    (tests/outlier_weights.py) - through all twelve blocks of each tower, against the fp32 oracle, at the north-star tolerance."""
    import outlier_weights as ow
    from scd_amd.clip.model import CLIP, DinoViT
    img = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(78)).half().float()
    if tower == "dino":
        sd, _ = ow.dino_outlier_state_dict(seed=1, layers=12)
        sd16 = ow.round_like_the_device(sd)
        _assert_pathologies(ow.residual_stream_stats(sd16, "dino", img[:2]))
        out = DinoViT(sd).cuda()(img.cuda()).float().cpu()
        ref = co.dino_forward(sd16, img)
    else:
        sd, _, _ = ow.clip_outlier_state_dict(seed=0, layers=12)
        sd16 = ow.round_like_the_device(sd)
        model = CLIP(sd).cuda().eval()
        if tower == "clip_image":
            _assert_pathologies(ow.residual_stream_stats(sd16, "clip_visual", img[:2]))
            out = model.encode_image(img.cuda()).float().cpu()
            ref = co.clip_encode_image(sd16, img)
        else:
            tok = torch.zeros(6, 77, dtype=torch.int32)
            g = torch.Generator().manual_seed(79)
            for i, ln in enumerate((1, 3, 8, 20, 40, 75)):
                tok[i, 0] = 49406
                tok[i, 1:1 + ln] = torch.randint(1, 49405, (ln,), generator=g, dtype=torch.int32)
                tok[i, 1 + ln] = 49407
            _assert_pathologies(ow.residual_stream_stats(sd16, "clip_text", tok.long()))
            out = model.encode_text(tok.cuda()).float().cpu()
            ref = co.clip_encode_text(sd16, tok.long())
    assert bool(torch.isfinite(out).all())
    assert _cos(out, ref).min().item() > 1 - 1e-3
    assert (out - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()


def test_outlier_weight_features_give_the_oracle_features_labels_and_names(ops):
    """What the towers' features are USED for, on the outlier-weight checkpoints: on a class-structured image set the SSKM labels
    (main_unsup.py:339-350) from the HIP DINO features equal those from the fp32 oracle's features row for row, and the top-1 names
    (main_unsup.py:504-531) of the HIP CLIP image features against a vocabulary built by the HIP text tower equal the top-1 names
    of the oracle's image features against the oracle's text features."""
    import outlier_weights as ow
    from scd_amd.clip.model import CLIP, DinoViT
    from scd_amd.kmeans import KMeansEngine
    n_cls, per = 6, 10
    g = torch.Generator().manual_seed(321)
    base = torch.randn(n_cls, 3, 224, 224, generator=g)
    y = np.repeat(np.arange(n_cls), per)
    img = (base[torch.from_numpy(y)] + 0.6 * torch.randn(n_cls * per, 3, 224, 224, generator=g)).half().float()
    norm = lambda t: torch.nn.functional.normalize(t.float(), dim=-1)
    # clustering features
    sdd, _ = ow.dino_outlier_state_dict(seed=1, layers=12)
    sdd16 = ow.round_like_the_device(sdd)
    hip = norm(DinoViT(sdd).cuda()(img.cuda())).cpu()
    ref = norm(torch.cat([co.dino_forward(sdd16, img[i:i + 12]) for i in range(0, len(img), 12)]))
    assert _cos(hip, ref).min().item() > 1 - 1e-3
    rs = np.random.RandomState(3)
    mask_lab = (y < n_cls // 2) & (rs.rand(len(y)) < 0.5)
    labels = []
    for feats in (hip, ref):
        km = KMeansEngine(k=n_cls, tolerance=1e-4, max_iterations=10, n_init=10, random_state=0)
        km.fit_mix(feats[~mask_lab].cuda(), feats[mask_lab].cuda(), torch.from_numpy(y[mask_lab]).cuda())
        labels.append(km.labels_.cpu().numpy())
    assert np.array_equal(labels[0], labels[1])
    # names
    sd, _, _ = ow.clip_outlier_state_dict(seed=0, layers=12)
    sd16 = ow.round_like_the_device(sd)
    model = CLIP(sd).cuda().eval()
    f_hip = norm(model.encode_image(img.cuda())).cpu()
    f_ref = norm(torch.cat([co.clip_encode_image(sd16, img[i:i + 12]) for i in range(0, len(img), 12)]))
    assert _cos(f_hip, f_ref).min().item() > 1 - 1e-3
    v = 48
    tok = torch.zeros(v, 77, dtype=torch.int32)
    gt = torch.Generator().manual_seed(5)
    for i in range(v):
        ln = 2 + i % 9
        tok[i, 0] = 49406
        tok[i, 1:1 + ln] = torch.randint(1, 49405, (ln,), generator=gt, dtype=torch.int32)
        tok[i, 1 + ln] = 49407
    w_hip = norm(model.encode_text(tok.cuda())).cpu()
    w_ref = norm(co.clip_encode_text(sd16, tok.long()))
    assert _cos(w_hip, w_ref).min().item() > 1 - 1e-3
    idx, _ = ops.sim_topk(f_hip.half().cuda(), w_hip.half().cuda().contiguous(), 1, "raw")
    s_ref = f_ref.double() @ w_ref.double().t()
    top2 = s_ref.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 4e-3                     # both operands within 1e-3 in cosine: a margin no such error flips
    assert decided.float().mean().item() > 0.8
    assert np.array_equal(idx[:, 0].cpu().numpy()[decided.numpy()], s_ref.argmax(1).numpy()[decided.numpy()])


def test_zeroshot_classifier_pooling(ops):
    from scd_amd.local_utils import clip_lang_util as clu
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    import scd_amd.clip as clip
    clip.allow_synthetic()
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(t_layers=2), visual=False)
    model = CLIP(sd).cuda()
    names = ["red_fox", "tabby", "kit_fox", "zebra", "grey_whale"]
    tmpl = clu.imagenet_templates[:9]
    zs = clu.zeroshot_classifier(names, tmpl, model, names_per_batch=2)
    assert zs.shape == (512, 5) and zs.dtype == torch.float16
    sd16 = {k: (v.half().float() if v.dim() >= 2 and "positional" not in k else v) for k, v in sd.items()}
    ref = no.zeroshot_classifier(names, tmpl, lambda t: co.clip_encode_text(sd16, t.long()).numpy(), clip.tokenize)
    assert _cos(zs.float().cpu().t(), torch.from_numpy(ref).t()).min().item() > 1 - 1e-3
    assert np.allclose(np.linalg.norm(zs.float().cpu().numpy(), axis=0), 1.0, atol=2e-3)
    # batching and length grouping are invisible in the result: whole vocabulary in one step with the prompts encoded in four
    # length groups (each trimmed to its own longest prompt) == two names per step, one group == full-length encodes
    names = names + ["soft-coated wheaten terrier", "x", "american black bear cub of the year"]
    tmpl = clu.imagenet_templates
    a = clu.zeroshot_classifier(names, tmpl, model, names_per_batch=256, length_groups=4, min_group=64)
    b = clu.zeroshot_classifier(names, tmpl, model, names_per_batch=2, length_groups=1)
    full = []
    for c in names:
        e = model.encode_text(clip.tokenize([t.format(c) for t in tmpl]).cuda())          # device ids: all 77 positions
        o = torch.empty((e.shape[1], 1), dtype=torch.float16, device="cuda")
        ops.prompt_pool(e.contiguous(), 1, len(tmpl), o, 0)
        full.append(o)
    assert torch.equal(a, b) and torch.equal(a, torch.cat(full, 1))


# ----------------------------------------------------------------------------------------------- full-size properties
def test_full_size_estep_mstep_properties(ops):
    """BASELINE C2 size (N_u=95k, D=768, K=100): size-independent properties instead of a CPU all-pairs oracle."""
    n, d, k = 95000, 768, 100
    x, y, cent = synth.clustered_features(n, d, k, seed=21, center_seed=22, noise=0.8)
    data = ops.KMeansData(dev(x))
    c = dev(cent)
    lab, ref = data.estep(c, return_refined=True)
    lab_np = lab.cpu().numpy()
    assert (lab_np == y).mean() > 0.99                           # recovers the planted partition
    assert int(ref.item()) < n // 10
    # spot-check 512 rows against the float64 oracle
    rs = np.random.RandomState(0)
    rows = rs.choice(n, 512, replace=False)
    olab, omind, _ = ko.estep(x[rows], cent)
    assert np.array_equal(lab_np[rows], olab)
    d2 = data.rowdist(c, lab)
    assert np.array_equal(d2.cpu().numpy()[rows], omind)
    # idempotence: M-step then E-step on converged data keeps the labels; sums add up to the column totals
    sums, counts, inertia = ops.kmeans_mstep(data.x, lab, c, k, 0)
    assert int(counts.sum().item()) == n
    col = x.astype(np.float64).sum(0)
    assert np.allclose(sums.sum(0).cpu().numpy(), col, rtol=1e-12, atol=1e-9)
    assert float(inertia[1].item()) == pytest.approx(float(ops.sum_f32(d2).item()), rel=1e-6)
    c2, shift = ops.kmeans_finalize(sums, counts, c)
    lab2 = data.estep(c2)
    assert (lab2 == lab).float().mean().item() > 0.999


# ----------------------------------------------------------------------------------------------- drop-in entry points
def test_main_unsup_and_ptsup_synthetic(ops, monkeypatch):
    import importlib
    import sys
    monkeypatch.setattr(sys, "argv", ["main_unsup.py", "--synthetic", "true", "--synthetic_images", "1536", "--synthetic_vocab", "600",
                                      "--n_cluster", "8", "--cluster", "SSKM", "--topk", "3", "--num_common_vote", "10",
                                      "--num_common_linear", "2"])
    mu = importlib.import_module("main_unsup")
    cand, u_preds = mu.main()
    assert len(cand) == 8 and set(int(c.split("_")[1]) for c in cand) == set(range(8))      # the planted names are found
    # the shipped default of scripts/evaluate_unsupervised.sh: --cluster KM, on the device (no host sklearn)
    cand, u_preds = mu.main(["--synthetic", "true", "--synthetic_images", "1536", "--synthetic_vocab", "600", "--n_cluster", "8",
                             "--cluster", "KM", "--topk", "3", "--num_common_vote", "10", "--num_common_linear", "2"])
    assert set(int(c.split("_")[1]) for c in cand) == set(range(8))
    monkeypatch.setattr(sys, "argv", ["main_ptsup.py", "--synthetic", "true", "--synthetic_images", "1536", "--synthetic_vocab", "600",
                                      "--n_cluster", "8", "--cluster", "ConSSKM", "--cluster_size_min", "50", "--cluster_size_max", "400",
                                      "--topk", "5", "--num_common_vote", "10", "--num_common_linear", "2"])
    mp_ = importlib.import_module("main_ptsup")
    cand, u_preds = mp_.main()
    assert set(int(c.split("_")[1]) for c in cand) == set(range(8))


# ----------------------------------------------------------------------------------------------- --cluster KM (sklearn KMeans)
@pytest.mark.parametrize("tag", ["a", "b", "c", "e"])
def test_sklearn_kmeans_lloyd_matches_sklearn(ops, golden, tag):
    """scd_amd.cluster.KMeans from an explicit init against scikit-learn 1.7.2's own labels (golden) and the oracle."""
    from scd_amd.cluster import KMeans
    g = golden("kmeans_sklearn.npz")
    n, d, k, seed = g[tag + "_shape"].tolist()
    x, _, _ = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 40, noise=float(g[tag + "_noise"]))
    km = KMeans(n_clusters=k, init=g[tag + "_init"], n_init=1, algorithm="lloyd", random_state=0).fit(x)
    assert km.labels_.dtype == np.int32 and np.array_equal(km.labels_, g[tag + "_labels"])
    assert km.n_iter_ == int(g[tag + "_n_iter"])
    assert abs(km.inertia_ - float(g[tag + "_inertia"])) <= 1e-5 * float(g[tag + "_inertia"])
    assert np.allclose(km.cluster_centers_, g[tag + "_centers"], rtol=1e-5, atol=1e-6, equal_nan=True)
    olab, oin, ocent, oit = ko.sklearn_lloyd(x, g[tag + "_init"])
    assert np.array_equal(km.labels_, olab) and np.array_equal(km.cluster_centers_, ocent) and km.n_iter_ == oit


@pytest.mark.parametrize("tag", ["a", "b", "c", "e"])
def test_sklearn_kmeans_seeding_and_default_call(ops, golden, tag):
    """`KMeans(n_clusters, random_state=0).fit(u_feats).labels_` (main_unsup.py:362, main_ptsup.py:381) against scikit-learn
    1.7.2 ITSELF (golden): the device k-means++ picks the rows `sklearn.cluster.kmeans_plusplus(x, k, random_state=0)` picks,
    the default call (one start) and the n_init=10 call (ten starts on one RandomState, the 1.0.2 default) return sklearn's
    labels; in the reference's pinned 1.0.2 mode the result equals the oracle's, whose seeding stream the reference-held `_k_init`
    pins (test_sklearn_102_seeding_matches_reference_k_init)."""
    from scd_amd.cluster import KMeans
    from scd_amd import ops as o
    g = golden("kmeans_sklearn.npz")
    n, d, k, seed = g[tag + "_shape"].tolist()
    x, y, _ = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 40, noise=float(g[tag + "_noise"]))
    km = KMeans(n_clusters=k, random_state=0, sklearn_compat="1.7.2")
    cent = km._kpp(o.KMeansData(dev(x)), ko.check_random_state(0)).cpu().numpy()
    assert np.array_equal(cent, x[g[tag + "_kpp_picks"]])
    for name, kw in (("default", {}), ("n10", {"n_init": 10})):
        km = KMeans(n_clusters=k, random_state=0, sklearn_compat="1.7.2", **kw).fit(x)
        assert km.labels_.dtype == np.int32 and np.array_equal(km.labels_, g["%s_%s_labels" % (tag, name)])
        assert km.n_iter_ == int(g["%s_%s_n_iter" % (tag, name)])
        assert abs(km.inertia_ - float(g["%s_%s_inertia" % (tag, name)])) <= 1e-5 * km.inertia_
    if tag != "c":                                       # (the float64 numpy oracle takes minutes at 10 x 4000 x 768)
        km = KMeans(n_clusters=k, random_state=0).fit(x)                # the reference's pin: 1.0.2 semantics, ten starts
        assert km.sklearn_compat == "1.0.2"
        olab, oin, ocent, oit = ko.sklearn_kmeans(x, k, 0, "auto", compat="1.0.2")
        assert np.array_equal(km.labels_, olab) and km.n_iter_ == oit and np.array_equal(km.cluster_centers_, ocent)
        assert km.inertia_ <= float(g[tag + "_default_inertia"]) * (1 + 1e-6) or tag == "a"


# ----------------------------------------------------------------------------------------------- pt-sup vote loop
def test_vote_loop_ptsup_matches_reference_trace(ops, golden):
    """naming.vote_loop_ptsup (HIP) on the vp_* inputs of the golden: per iteration voted / ind / cand / u_preds /
    unlab_cluster_idx equal the oracle's on the same fp16-rounded inputs, and the reference's own trace at convergence
    (main_ptsup.py:629-676).  set-of-str order (:664) depends on the hash seed the golden was recorded under: replay there."""
    import subprocess
    import sys
    if os.environ.get("PYTHONHASHSEED") != "0":
        env = dict(os.environ, PYTHONHASHSEED="0")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__, "-k", "test_vote_loop_ptsup_matches_reference_trace"],
                           env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return
    from scd_amd import naming
    g = golden("naming.npz")
    k, n_lab, topk, ncv, ncl = g["vp_cfg"].tolist()
    f, w = g["tk_x"].astype(np.float16), g["tk_w"].astype(np.float16)
    nouns = synth.nouns_list(w.shape[1])
    mask_lab = g["vp_mask_lab"]
    lab_names = [nouns[c] for c in range(n_lab)]
    wt = ops.transpose_f16(dev(w))
    fh = dev(f)
    idx, _ = naming.full_vocab_topk(fh, None, 5, False, wt=wt)                      # TOP_K = 5, raw logits (:526-545)
    oidx, _ = no.sim_topk(f, w, 5, "raw")
    assert np.array_equal(idx.cpu().numpy(), oidx)
    m = dev(~mask_lab)
    cand, up, tr = naming.vote_loop_ptsup(idx[m], g["vp_all_preds0"], mask_lab, fh[m], wt, nouns, lab_names, k, topk, ncv, ncl)
    otr = no.vote_loop_ptsup(oidx[~mask_lab], g["vp_all_preds0"], mask_lab, f[~mask_lab], w, nouns, lab_names, k, topk, ncv, ncl)
    assert len(tr) == len(otr) >= 1
    for a, b in zip(tr, otr):
        for key in ("voted", "ind", "cand", "u_preds", "unlab_cluster_idx"):
            assert np.array_equal(a[key], b[key]), key
    last = int(g["vp_iters"]) - 1
    assert set(tr[-1]["cand"].tolist()) == set(g["vp_cand_%d" % last].tolist())
    assert (tr[-1]["u_preds"] == g["vp_preds_%d" % last]).mean() > 0.995


def test_vote_loops_equal_the_reference_traces_on_f16_inputs(ops, golden):
    """Both vote loops of the reference (main_unsup.py:568-614, main_ptsup.py:588-676) exec'd at fixture-generation time on the
    fp16-exact inputs of tests/golden/topk_f16.npz, against naming.vote_loop_unsup / vote_loop_ptsup on the HIP library: EVERY iteration of
    both traces - voted names, cluster-to-name assignment, candidate names, re-classified predictions, unlabelled cluster ids - equals the
    reference's own, starting from the HIP path's own top-k (rows a5-a9 against the reference itself, not only against the oracle)."""
    import subprocess
    import sys
    if os.environ.get("PYTHONHASHSEED") != "0":   # set-of-str order (main_ptsup.py:664) depends on the hash seed of the fixture's run
        env = dict(os.environ, PYTHONHASHSEED="0")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__, "-k",
                            "test_vote_loops_equal_the_reference_traces_on_f16_inputs"], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return
    from scd_amd import naming
    from test_oracle_golden import topk16_case
    g, f16, w16 = topk16_case(golden)
    nouns = synth.nouns_list(w16.shape[1])
    wt = ops.transpose_f16(dev(w16))
    fh = dev(f16)
    k, topk, ncv, ncl = g["vu_cfg"].tolist()
    idx, _ = naming.full_vocab_topk(fh, None, 5, True, wt=wt)
    assert np.array_equal(idx.cpu().numpy(), g["idx_unsup"])
    cand, up, tr = naming.vote_loop_unsup(idx, g["vu_preds0"], fh, wt, nouns, k, ncv, ncl)
    assert len(tr) == int(g["vu_iters"])
    for i, t in enumerate(tr):
        for key, gk in (("voted", "vu_voted_%d"), ("ind", "vu_ind_%d"), ("cand", "vu_cand_%d"), ("u_preds", "vu_preds_%d")):
            assert np.array_equal(np.asarray(t[key]), g[gk % i]), (i, key)
    k, n_lab, topk, ncv, ncl = g["vp_cfg"].tolist()
    mask_lab = g["vp_mask_lab"]
    lab_names = [nouns[c] for c in range(n_lab)]
    idx, _ = naming.full_vocab_topk(fh, None, 5, False, wt=wt)
    assert np.array_equal(idx.cpu().numpy(), g["idx_ptsup"])
    m = dev(~mask_lab)
    cand, up, tr = naming.vote_loop_ptsup(idx[m], g["vp_all_preds0"], mask_lab, fh[m], wt, nouns, lab_names, k, topk, ncv, ncl)
    assert len(tr) == int(g["vp_iters"])
    for i, t in enumerate(tr):
        for key, gk in (("voted", "vp_voted_%d"), ("ind", "vp_ind_%d"), ("cand", "vp_cand_%d"), ("u_preds", "vp_preds_%d"),
                        ("unlab_cluster_idx", "vp_unlab_%d")):
            assert np.array_equal(np.asarray(t[key]), g[gk % i]), (i, key)


# ----------------------------------------------------------------------------------------------- BASELINE configs[3] / [4] shapes
@pytest.mark.parametrize("ncv,ncl", [(10, 2), (20, 4)])
def test_c4_shape_vote_k1000(ops, ncv, ncl):
    """BASELINE configs[3] per-GPU shard: N = 160,146 rows, D = 512, K = 1000 clusters, V = 21,000 names; num_common_vote 10 / 20
    -> assign_name on D = 10,000-20,000 (scd_munkres_sparse).  Checked per vote iteration: the device histograms equal the
    oracle's Counters on sampled clusters, the voted list equals the oracle's, the assignment is a permutation whose weight is
    the optimum (scipy on the K x names block), the re-classification equals the oracle on sampled rows."""
    import time
    from scipy.optimize import linear_sum_assignment
    from scd_amd import naming
    n, d, k, v = 160146, 512, 1000, 21000
    x, y, cent = synth.clustered_features(n, d, k, seed=81, center_seed=82, noise=1.0)
    w = synth.vocabulary(v, d, cent, seed=83, jitter=0.6)
    nouns = synth.nouns_list(v)
    f16 = x.astype(np.float16)
    fh, wt = dev(f16), ops.transpose_f16(dev(w))
    idx, _ = naming.full_vocab_topk(fh, None, 3, True, wt=wt)
    rs = np.random.RandomState(84)
    preds0 = np.where(rs.rand(n) < 0.85, y, rs.randint(0, k, size=n))
    seen = []

    def on_iter(it, cand, u_preds):
        seen.append(time.time())
    t0 = time.time()
    cand, up, tr = naming.vote_loop_unsup(idx, preds0, fh, wt, nouns, k, ncv, ncl, on_iter=on_iter, max_iter=6)
    assert len(tr) >= 1 and (time.time() - t0) / len(tr) < 30.0                 # Munkres at D = 10k-20k is not an Amdahl wall
    name_idx = idx.cpu().numpy()
    u_prev = preds0
    for t in tr[:2]:
        clusters = list(set(u_prev.tolist()))
        sample = clusters[:: max(1, len(clusters) // 40)]
        ref = no.cluster_counters(name_idx, u_prev, sample, 5)
        keys, counts = ops.vote_hist(idx, 3, dev(u_prev), sample, max(ncv, ncl))
        keys, counts = keys.cpu().numpy(), counts.cpu().numpy()
        for i, c in enumerate(sample):
            got = [(int(a), int(b)) for a, b in zip(keys[i], counts[i]) if a >= 0]
            assert got == [(int(a), int(b)) for a, b in ref[c].most_common(max(ncv, ncl))], c
        full = naming.cluster_counters(idx, 3, dev(u_prev), clusters, max(ncv, ncl))
        voted = []
        for c in clusters:
            voted += [a for a, _ in full[c].most_common(ncv)]
        voted = list(set(voted))
        assert np.array_equal(t["voted"], np.array(voted, dtype=np.int64))
        dd = max(len(voted), len(clusters))
        ind = t["ind"]
        assert ind.shape == (dd, 2) and np.array_equal(ind[:, 0], np.arange(dd)) and len(set(ind[:, 1].tolist())) == dd
        col = {u: j for j, u in enumerate(voted)}
        wsm = np.zeros((len(clusters), dd), dtype=np.int64)
        for i, c in enumerate(clusters):
            for a, b in full[c].most_common(ncl):
                wsm[i, col[a]] += b
        r_, c_ = linear_sum_assignment(wsm, maximize=True)
        assert int(wsm[np.arange(len(clusters)), ind[:len(clusters), 1]].sum()) == int(wsm[r_, c_].sum())
        rows = rs.choice(n, 1024, replace=False)
        oi, _ = no.sim_argmax(f16[rows], w[:, t["cand"]])
        assert np.array_equal(t["u_preds"][rows], oi)
        u_prev = t["u_preds"]
    names_found = len(set(int(c.split("_")[1]) for c in cand) & set(range(k)))
    assert names_found > 0.9 * k


def test_sim_topk_v100k(ops):
    """BASELINE configs[4]: V = 100,000 names.  256 rows against the oracle; then the C2-sized image set (126,976 rows) through
    size-independent properties: sampled rows equal the oracle, values sorted, planted names found."""
    v, d, k = 100000, 512, 100
    x, y, cent = synth.clustered_features(126976, d, k, seed=91, center_seed=92, noise=0.9)
    w = synth.vocabulary(v, d, cent, seed=93, jitter=0.4)
    f16 = x.astype(np.float16)
    wt = ops.transpose_f16(dev(w))
    for mode in ("raw", "softmax"):
        idx, val = ops.sim_topk(dev(f16[:256]), wt, 5, mode)
        oi, ov = no.sim_topk(f16[:256], w, 5, mode)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.allclose(val.cpu().numpy(), ov, rtol=2e-4, atol=1e-6)
    idx, val, fb = ops.sim_topk(dev(f16), wt, 5, "raw", return_fallback=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert (np.diff(val, axis=1) <= 0).all() and idx.min() >= 0 and idx.max() < v
    assert (idx[:, 0] == y).mean() > 0.98 and int(fb.item()) < len(x) // 100
    rows = np.random.RandomState(0).choice(len(x), 300, replace=False)
    oi, _ = no.sim_topk(f16[rows], w, 5, "raw")
    assert np.array_equal(idx[rows], oi)


def test_textual_enhancement_rerank(ops):
    """BASELINE configs[4] 'textual-enhancement re-ranking': logits = 100 * (f @ W + t @ W) / 2 (the formula the reference
    leaves commented at main_unsup.py:518,523,604,609) = 100 * mean(f, t) @ W: one elementwise mean (fp16, rounded once), then
    the same exact top-k."""
    from scd_amd import naming
    v, d, k = 5000, 512, 40
    x, y, cent = synth.clustered_features(3000, d, k, seed=95, center_seed=96, noise=1.1)
    t, _, _ = synth.clustered_features(3000, d, k, seed=95, center_seed=96, noise=0.5)       # same classes, cleaner features
    w = synth.vocabulary(v, d, cent, seed=98, jitter=0.5)
    f16, t16 = x.astype(np.float16), t.astype(np.float16)
    wt = ops.transpose_f16(dev(w))
    g16 = ((f16.astype(np.float32) + t16.astype(np.float32)) * np.float32(0.5)).astype(np.float16)
    assert np.array_equal(ops.mean2_f16(dev(f16), dev(t16)).cpu().numpy(), g16)
    for softmax in (False, True):
        idx, val = naming.full_vocab_topk_te(dev(f16), dev(t16), wt, 5, softmax=softmax)
        oi, ov = no.sim_topk(g16, w, 5, "softmax" if softmax else "raw")
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.allclose(val.cpu().numpy(), ov, rtol=2e-4, atol=1e-6)
    plain, _ = naming.full_vocab_topk(dev(f16), None, 5, False, wt=wt)
    assert (idx[:, 0].cpu().numpy() == y).mean() >= (plain[:, 0].cpu().numpy() == y).mean()      # the text term helps


# ----------------------------------------------------------------------------------------------- a7, zero-shot bounds, caches
def test_match_missing_names_golden_and_oracle(ops, golden):
    """Row a7 (main_unsup.py:402-406, 487-491, 459-469): HIP path = oracle on the same fp16 inputs = the reference's own lines
    (golden, fp32 inputs) - top-1 over the vocabulary, top-1 over the truncated vocabulary, greedy de-duplicated top-5."""
    from scd_amd import naming
    g = golden("naming.npz")
    w, mw = g["mm_w"].astype(np.float16), g["mm_miss_w"].astype(np.float16)
    nouns = synth.nouns_list(w.shape[1])
    wt = ops.transpose_f16(dev(w))
    miss = ["miss_%d" % i for i in range(mw.shape[1])]
    original = [nouns[c] for c in g["mm_class_cols"].tolist()] + miss
    trunc = [n for n in nouns if n not in original]
    got = [naming.match_missing_names(miss, nouns, wt, None, "top1", miss_weights=dev(mw)),
           naming.match_missing_names(miss, nouns, wt, None, "top1", nouns_truncated=trunc, miss_weights=dev(mw)),
           naming.match_missing_names(miss, nouns, wt, None, "greedy_top5", nouns_truncated=trunc, miss_weights=dev(mw))]
    want = [no.match_missing_names(mw, w, nouns, None, "top1"), no.match_missing_names(mw, w, nouns, trunc, "top1"),
            no.match_missing_names(mw, w, nouns, trunc, "greedy_top5")]
    assert got == want
    for names, key in zip(got, ("mm_top1_full", "mm_top1_trunc", "mm_greedy5_trunc")):
        assert [nouns.index(n) for n in names] == g[key].tolist(), key
    assert got[2][3] != got[0][3] and len(set(got[2])) == len(got[2])          # the fourth class had to take its second choice
    cidx = naming.class_names_with_matches({n: i for i, n in enumerate(original)}, miss, got[0])
    assert [cidx[i] for i in range(len(original))] == original[:4] + got[0]
    # a duplicate vocabulary entry: `nouns.index` semantics (first occurrence) when the truncated columns are gathered
    nouns2 = list(nouns)
    nouns2[21] = nouns2[3]
    trunc2 = [n for n in nouns2 if n not in original]
    assert naming.match_missing_names(miss, nouns2, wt, None, "top1", nouns_truncated=trunc2, miss_weights=dev(mw)) == \
        no.match_missing_names(mw, w, nouns2, trunc2, "top1")


def test_match_missing_names_through_text_tower(ops):
    """The same call with the classifier of the missing names built on the HIP text tower (zeroshot_classifier, 80 templates)."""
    from scd_amd import naming
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    import scd_amd.clip as clip
    from scd_amd.local_utils import clip_lang_util as clu
    clip.allow_synthetic()
    model = CLIP(W.synthetic_clip_state_dict(seed=0, cfg=dict(t_layers=2), visual=False)).cuda()
    nouns = ["red_fox", "tabby", "kit_fox", "zebra", "grey_whale", "ox", "heron", "pug", "lynx", "newt"]
    zw = clu.zeroshot_classifier(nouns, clu.imagenet_templates, model)
    wt = ops.transpose_f16(zw)
    miss = ["arctic_fox", "sea_lion"]
    got = naming.match_missing_names(miss, nouns, wt, model, "top1")
    mw = clu.zeroshot_classifier(miss, clu.imagenet_templates, model)
    assert got == no.match_missing_names(mw.cpu().numpy(), zw.cpu().numpy(), nouns, None, "top1")
    cidx = naming.resolve_class_names("cifar10", "wordnet", {"zebra": 0, "arctic_fox": 1, "sea_lion": 2}, nouns, wt, model)
    assert cidx == {0: "zebra", 1: got[0], 2: got[1]}


def test_zero_shot_bounds_and_cidx_names(ops):
    from scd_amd import naming
    v, d, k = 700, 512, 9
    x, y, cent = synth.clustered_features(1500, d, k, seed=41, center_seed=42, noise=1.0)
    w = synth.vocabulary(v, d, cent, seed=43, jitter=0.6)
    nouns = synth.nouns_list(v)
    cidx = {c: nouns[c] for c in range(k)}
    f16 = x.astype(np.float16)
    lg = 100.0 * f16.astype(np.float64) @ w.astype(np.float64)
    t_idx = np.array([nouns.index(cidx[int(t)]) for t in y])
    want1 = no.accuracy(lg, t_idx, topk=(1, 5))
    top1, top5 = naming.evaluate_semantic_acc_ub_lb(f16, y.astype(np.float64), {float(c): n for c, n in cidx.items()}, nouns, w, return_top5=True)
    assert top1 == pytest.approx(want1[0] / len(y) * 100) and top5 == pytest.approx(want1[1] / len(y) * 100)
    cand = [nouns[c] for c in range(k)]
    wsel = w[:, :k]
    ub = naming.evaluate_semantic_acc_ub_lb(dev(f16), y, cidx, cand, dev(wsel))
    assert ub == pytest.approx(float((lg[:, :k].argmax(1) == y).mean() * 100))
    preds = naming.get_clip_preds_fast(f16, y, cidx, nouns, w).cpu().numpy()
    assert np.array_equal(preds, lg.argmax(1))


def _write_cache_tree(root, n=1600, k=8, v=500, seed=51, dataset="cifar10", corpus="wordnet", d_feat=512, noise=0.9, layers=1,
                      feat_model="dino_vit"):
    """The reference's on-disk boundary: feature dicts (main_unsup.py:141-146), the [512, V] classifier
    (:389-394), the vocabulary file get_nouns reads (clip_lang_util.py:139-149), a class-name table and a CLIP checkpoint."""
    import json
    from scd_amd.clip import weights as W
    x, y, cent = synth.clustered_features(n, 512, k, seed=seed, center_seed=seed + 1, noise=noise)
    perm, mask_lab = synth.labelled_split(y, k, prop=0.5, seed=seed + 2)
    x, y = x[perm], y[perm]
    xf = x                                                        # clustering features: the CLIP ones, or d_feat-wide "DINO" ones
    if d_feat != 512:
        xf, yf, _ = synth.clustered_features(n, d_feat, k, seed=seed, center_seed=seed + 5, noise=noise)
        assert np.array_equal(yf[perm], y)
        xf = xf[perm]
    w = synth.vocabulary(v, 512, cent, seed=seed + 3, jitter=0.4)
    nouns = ["noun-%03d" % i for i in range(v)]                   # get_nouns output is lower-cased and '-' -> '_' by the mains
    zname, vfile = {"wordnet": ("nouns", "wordnet_all_noun.txt"), "wikibird": ("wikibird", "wiki_birdclass_names.txt"),
                    "wikidog": ("wikidog", "wiki_dogclass_names.txt")}[corpus]
    os.makedirs(os.path.join(root, "extracted_features"))
    os.makedirs(os.path.join(root, "zeroshot_weights"))
    os.makedirs(os.path.join(root, "data"))
    os.makedirs(os.path.join(root, "clip"))
    feats = dict(all_feats=xf.astype(np.float32), mask_lab=mask_lab, mask_cls=(y < k // 2), targets=y.astype(np.float64))
    torch.save(feats, os.path.join(root, "extracted_features", "%s_%s_all.pt" % (feat_model, dataset)))
    torch.save(dict(feats, all_feats=x.astype(np.float16)), os.path.join(root, "extracted_features", "clip_%s_all.pt" % dataset))
    torch.save(torch.from_numpy(w), os.path.join(root, "zeroshot_weights", "zeroshot_weights_all_%s_vit_b_16.pt" % zname))
    with open(os.path.join(root, "data", vfile), "w") as f:
        f.write("\n".join(nouns) + "\n")
    # classes 0..k-2 carry vocabulary names, the last one a name the vocabulary lacks (row a7 runs on the text tower)
    class_to_idx = {("noun_%03d" % c if c < k - 1 else "not_a_noun"): c for c in range(k)}
    with open(os.path.join(root, "class_names.json"), "w") as f:
        json.dump(class_to_idx, f)
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=layers, t_layers=layers))
    torch.save({kk: (vv.half() if vv.dim() >= 2 else vv) for kk, vv in sd.items()}, os.path.join(root, "clip", "ViT-B-16.pt"))
    if d_feat != 512:
        return x, y, mask_lab, w, xf
    return x, y, mask_lab, w


def test_c1_shape_end_to_end(ops, tmp_path, monkeypatch, capsys):
    """BASELINE configs[0] at its own shape, through main_unsup.main() on the reference's cache files: CUB-200 unsupervised - 5,994
    images of which ~4,500 unlabelled, cached 768-wide float32 "DINO" features for the clustering, fp16 CLIP features and a 1,000-name
    vocabulary for the naming, K = 200, the flags main_unsup.py:222-224 name for cub (--topk 3 --num_common_vote 10
    --num_common_linear 2).  The whole chain k-means -> full-vocabulary top-k -> vote loop -> names equals the ORACLE chain run on the
    same files bit for bit, for the shipped default `--cluster KM` (`KMeans(n_clusters, random_state=0)`: the scikit-learn 1.0.2 rules
    the reference pins, ten starts, whose seeding stream is pinned by the reference-held `_k_init` at this shape; and the 1.7.2 rules
    pinned by scikit-learn itself) and for `--cluster SSKM` (reference: main_unsup.py:334-364, 504-531, 568-614)."""
    import importlib
    import scd_amd.clip as clip
    root = str(tmp_path)
    n, k, v = 5994, 200, 1000
    x, y, mask_lab, w, xf = _write_cache_tree(root, n=n, k=k, v=v, seed=61, dataset="cub", corpus="wikibird", d_feat=768)
    monkeypatch.setenv("SCD_ROOT", root)
    monkeypatch.setenv("SCD_DATA", os.path.join(root, "data"))
    monkeypatch.setattr(clip, "_tokenizer", None)
    monkeypatch.setenv("SCD_SYNTHETIC", "1")                     # hash tokenizer for the one class name the vocabulary lacks
    mu = importlib.import_module("main_unsup")
    common = ["--root_dir", root, "--dataset_name", "cub", "--corpus", "wikibird", "--feat_model", "dino_vit", "--n_cluster", str(k),
              "--topk", "3", "--num_common_vote", "10", "--num_common_linear", "2", "--run_cluster", "true",
              "--class_names", os.path.join(root, "class_names.json")]
    n_u = int((~mask_lab).sum())
    assert 4300 <= n_u <= 4700
    xu, xl, yl = xf[~mask_lab], xf[mask_lab], y[mask_lab]
    nouns = ["noun_%03d" % i for i in range(v)]                  # load_vocabulary: lower-cased, '-' -> '_' 
    f16, w16 = x.astype(np.float16), w.astype(np.float16)
    oidx, _ = no.sim_topk(f16, w16, 3, "softmax")

    def oracle_names(preds0):
        tr = no.vote_loop_unsup(oidx[~mask_lab], preds0, f16[~mask_lab], w16, nouns, k, 3, 10, 2)
        return [nouns[c] for c in tr[-1]["cand"].tolist()], tr[-1]["u_preds"], len(tr)

    for compat in ("1.0.2", "1.7.2"):
        monkeypatch.setenv("SCD_SKLEARN_COMPAT", compat)
        cand, u_preds = mu.main(common + ["--cluster", "KM", "--save_cluster", "true"])
        saved = torch.load(os.path.join(root, "cluster", "KM_dino_vit_cub_%d.pt" % k), weights_only=False)
        olab, _, _, _ = ko.sklearn_kmeans(xu, k, 0, "auto", compat=compat)
        assert saved["u_preds"].dtype == np.int32 and np.array_equal(saved["u_preds"], olab), compat
        ocand, opreds, oit = oracle_names(olab)
        assert list(cand) == ocand and np.array_equal(np.asarray(u_preds), opreds), compat
        out = capsys.readouterr().out
        assert "voting converged after %d iterations" % oit in out and "sACC_avg" in out
        assert len(set("noun_%03d" % c for c in range(k - 1)) & set(cand)) >= 150          # the planted names are found
    monkeypatch.delenv("SCD_SKLEARN_COMPAT")
    # SSKM: the reference passes random_state=None (main_unsup.py:350): both sides draw from numpy's global RandomState
    np.random.seed(7)
    cand, u_preds = mu.main(common + ["--cluster", "SSKM", "--save_cluster", "true"])
    saved = torch.load(os.path.join(root, "cluster", "SSKM_dino_vit_cub_%d.pt" % k), weights_only=False)
    np.random.seed(7)
    okm = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=10, random_state=None)
    okm.fit_mix(xu, xl, yl)
    assert np.array_equal(saved["all_preds"], okm.labels_)
    ocand, opreds, _ = oracle_names(okm.labels_[len(yl):])
    assert list(cand) == ocand and np.array_equal(np.asarray(u_preds), opreds)


def test_c3_shape_ptsup_end_to_end(ops, tmp_path, monkeypatch, capsys):
    """BASELINE configs[2] at its own shape, through main_ptsup.main() on the reference's cache files: Stanford Dogs partially
    supervised - 12,000 images of which ~3,000 labelled (classes < 60, half of their rows), cached 768-wide float32 GCD features, K =
    120, the flags main_unsup.py:222-224 name for sdogs (--topk 2 --num_common_vote 5 --num_common_linear 2), the dog-name corpus.
    `--cluster SSKM`: all_preds, the top-5 indices, every iteration of the partially supervised vote (names of the labelled classes
    kept, the rest voted) and the final names / u_preds equal the oracle chain on the same files (main_ptsup.py:526-545, 588-676);
    the class name the vocabulary lacks goes through the greedy top-5 matching of the sdogs branch (:459-469)."""
    import importlib
    import scd_amd.clip as clip
    from scd_amd import naming
    root = str(tmp_path)
    n, k, v = 12000, 120, 1000
    x, y, mask_lab, w, xf = _write_cache_tree(root, n=n, k=k, v=v, seed=71, dataset="sdogs", corpus="wikidog", d_feat=768, feat_model="gcd")
    monkeypatch.setenv("SCD_ROOT", root)
    monkeypatch.setenv("SCD_DATA", os.path.join(root, "data"))
    monkeypatch.setattr(clip, "_tokenizer", None)
    monkeypatch.setenv("SCD_SYNTHETIC", "1")
    mp_ = importlib.import_module("main_ptsup")
    assert 2700 <= int(mask_lab.sum()) <= 3300 and mask_lab[: int(mask_lab.sum())].all()          # labelled rows first
    np.random.seed(11)                                  # random_state=None at the call site (main_ptsup.py:369): numpy's global stream
    cand, u_preds = mp_.main(["--root_dir", root, "--dataset_name", "sdogs", "--corpus", "wikidog", "--feat_model", "gcd", "--n_cluster", str(k),
                              "--cluster", "SSKM", "--topk", "2", "--num_common_vote", "5", "--num_common_linear", "2", "--run_cluster", "true",
                              "--save_cluster", "true", "--class_names", os.path.join(root, "class_names.json")])
    out = capsys.readouterr().out
    assert "sACC lower bound" in out and "sACC upper bound" in out and "Missed 1 names" in out
    saved = torch.load(os.path.join(root, "cluster", "SSKM_gcd_sdogs.pt"), weights_only=False)
    np.random.seed(11)
    okm = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", n_init=10, random_state=None)
    okm.fit_mix(xf[~mask_lab], xf[mask_lab], y[mask_lab])
    assert np.array_equal(saved["all_preds"], okm.labels_)
    nouns = ["noun_%03d" % i for i in range(v)]
    f16, w16 = x.astype(np.float16), w.astype(np.float16)
    oidx, _ = no.sim_topk(f16, w16, 5, "raw")
    # the labelled classes' names as main_ptsup derives them (class k - 1 carries a name outside the vocabulary: matched by the text tower)
    model, _ = clip.load("ViT-B/16")
    import json
    with open(os.path.join(root, "class_names.json")) as fh:
        class_to_idx = {kk: int(vv) for kk, vv in json.load(fh).items()}
    wt = ops.transpose_f16(dev(w16))
    cname = naming.resolve_class_names("sdogs", "wikidog", class_to_idx, nouns, wt, model.cuda().eval())
    lab_names = [cname[c] for c in range(k // 2)]
    assert lab_names == ["noun_%03d" % c for c in range(k // 2)]
    otr = no.vote_loop_ptsup(oidx[~mask_lab], okm.labels_, mask_lab, f16[~mask_lab], w16, nouns, lab_names, k, 2, 5, 2)
    assert "voting converged after %d iterations" % len(otr) in out
    assert list(cand) == [nouns[c] for c in otr[-1]["cand"].tolist()] and np.array_equal(np.asarray(u_preds), otr[-1]["u_preds"])
    assert set(lab_names) <= set(cand) and len(cand) == k


def test_c3_shape_ptsup_end_to_end_consskm(ops, tmp_path, monkeypatch, capsys):
    """BASELINE configs[2] AS NAMED: Stanford Dogs partially supervised through main_ptsup.main() with its DEFAULT clustering -
    `--cluster ConSSKM`, size bounds 50 / 1000, ten restarts x ten iterations (main_ptsup.py:236,239-240,356-366) - on the
    reference's cache files at the config's shape (12,000 x 768 GCD features, ~3,000 labelled, K = 120, dog-name corpus).
    * every flow problem of the fit (one per restart and iteration) is feasible, inside the bounds and optimal (no negative cycle in
      the residual graph = the LP certificate of oracle/transport_oracle.py); where its optimum is unique (perturbation check) the
      labels are the LP's labels.  OR-Tools' own labels are third-party-unpinned;
    * the restarts advance in lock-step over host threads (ConstrainedEngine._run): all_preds equals the sequential fit's, bit for bit;
    * everything downstream of the clustering - top-5 indices, every iteration of the partially supervised vote, final names and
      u_preds - equals the oracle chain run from OUR all_preds (main_ptsup.py:526-545,588-676)."""
    import importlib
    import json
    import time
    import scd_amd.clip as clip
    from scd_amd import naming
    from scd_amd.local_utils.sskm_constrained import K_Means
    root = str(tmp_path)
    n, k, v = 12000, 120, 1000
    x, y, mask_lab, w, xf = _write_cache_tree(root, n=n, k=k, v=v, seed=71, dataset="sdogs", corpus="wikidog", d_feat=768, feat_model="gcd")
    monkeypatch.setenv("SCD_ROOT", root)
    monkeypatch.setenv("SCD_DATA", os.path.join(root, "data"))
    monkeypatch.setattr(clip, "_tokenizer", None)
    monkeypatch.setenv("SCD_SYNTHETIC", "1")
    mp_ = importlib.import_module("main_ptsup")
    calls = _record_transport(monkeypatch)
    np.random.seed(11)                                  # random_state=None at the call site (main_ptsup.py:369): numpy's global stream
    cand, u_preds = mp_.main(["--root_dir", root, "--dataset_name", "sdogs", "--corpus", "wikidog", "--feat_model", "gcd", "--n_cluster", str(k),
                              "--topk", "2", "--num_common_vote", "5", "--num_common_linear", "2", "--run_cluster", "true",
                              "--save_cluster", "true", "--class_names", os.path.join(root, "class_names.json")])
    out = capsys.readouterr().out
    assert "Fitting ConSSKM" in out
    saved = torch.load(os.path.join(root, "cluster", "ConSSKM_gcd_sdogs.pt"), weights_only=False)
    all_preds = np.asarray(saved["all_preds"])
    n_l = int(mask_lab.sum())
    cnt = np.bincount(all_preds[n_l:], minlength=k)
    assert cnt.min() >= 50 and cnt.max() <= 1000
    assert np.array_equal(all_preds[:n_l], np.searchsorted(np.unique(y[:n_l]), y[:n_l]))       # labelled rows keep their classes
    # the flow problems: 10 restarts x (up to) 10 iterations
    n_calls = len(calls)
    assert 10 <= n_calls <= 100 and all(c[0].shape == (n - n_l, k) for c in calls)
    n_unique = 0
    for i, (cost, smin, smax, labs, tot) in enumerate(list(calls)):
        ok, tot_chk = to.check_assignment(cost, labs, smin, smax)
        assert (smin, smax) == (50, 1000) and ok and tot == tot_chk and to.check_optimal(cost, labs, smin, smax)
        if i % 9 == 0 and _unique_optimum(cost, smin, smax, labs, ops):       # a sample: the uniqueness check is three more solves,
            n_unique += 1                                                      # and the LP takes 10-30 s per problem at this size
            if n_unique <= 2:
                lp_lab, lp_tot = to.solve_lp(cost, smin, smax)
                assert lp_tot == tot and np.array_equal(labs, lp_lab)
    # lock-step restarts = sequential restarts (same fit, same stream), and the time of the default fit
    fits = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCD_CONSSKM_LOCKSTEP", mode)
        np.random.seed(11)
        km = K_Means(k=k, tolerance=1e-4, max_iterations=10, init="k-means++", size_min=50, size_max=1000, n_init=10,
                     random_state=None, n_jobs=None, pairwise_batch_size=1024)
        torch.cuda.synchronize()
        t0 = time.time()
        km.fit_mix(dev(xf[~mask_lab]), dev(xf[mask_lab]), dev(y[mask_lab]))
        torch.cuda.synchronize()
        fits[mode] = (km.labels_.cpu().numpy(), km.cluster_centers_.cpu().numpy(), float(km.inertia_), time.time() - t0)
    monkeypatch.delenv("SCD_CONSSKM_LOCKSTEP")
    assert np.array_equal(fits["1"][0], fits["0"][0]) and np.array_equal(fits["1"][1], fits["0"][1]) and fits["1"][2] == fits["0"][2]
    assert np.array_equal(fits["1"][0], all_preds)
    print("C3 ConSSKM fit (10 restarts x 10 iterations, 9,000 x 120 flow problems): lock-step %.2f s, sequential %.2f s, %d solves, "
          "%d of the sampled optima unique" % (fits["1"][3], fits["0"][3], n_calls, n_unique))
    assert fits["1"][3] < 10.0
    # downstream of OUR labels: the oracle chain
    nouns = ["noun_%03d" % i for i in range(v)]
    f16, w16 = x.astype(np.float16), w.astype(np.float16)
    oidx, _ = no.sim_topk(f16, w16, 5, "raw")
    model, _ = clip.load("ViT-B/16")
    with open(os.path.join(root, "class_names.json")) as fh:
        class_to_idx = {kk: int(vv) for kk, vv in json.load(fh).items()}
    wt = ops.transpose_f16(dev(w16))
    cname = naming.resolve_class_names("sdogs", "wikidog", class_to_idx, nouns, wt, model.cuda().eval())
    lab_names = [cname[c] for c in range(k // 2)]
    otr = no.vote_loop_ptsup(oidx[~mask_lab], all_preds, mask_lab, f16[~mask_lab], w16, nouns, lab_names, k, 2, 5, 2)
    assert "voting converged after %d iterations" % len(otr) in out
    assert list(cand) == [nouns[c] for c in otr[-1]["cand"].tolist()] and np.array_equal(np.asarray(u_preds), otr[-1]["u_preds"])
    assert set(lab_names) <= set(cand) and len(cand) == k


@pytest.mark.parametrize("tag", ["a", "b", "e", "c", "c1"])
def test_sklearn_102_seeding_matches_reference_k_init(ops, golden, tag):
    """The default mode of scd_amd.cluster.KMeans (scikit-learn 1.0.2 rules, the reference's pin): scd_kpp_greedy_lockstep picks the
    rows the reference-held `_k_init` picks (tests/golden/kmeans_sklearn.npz `*_kinit_*`, oracle/gen_golden.py: ten consecutive
    seedings on RandomState(0) = the n_init = 10 stream) - on the float32 rows (dense candidate evaluation) and, for rows rounded to
    fp16 (features that left an fp16 encoder: MFMA lower-bound filter + exact pairs), the rows the oracle picks on the same rows."""
    from scd_amd.cluster import KMeans
    from scd_amd import ops as o
    g = golden("kmeans_sklearn.npz")
    n, d, k, seed = g[tag + "_shape"].tolist()
    x, _, _ = synth.clustered_features(n, d, k, seed=seed, center_seed=seed + 40, noise=float(g[tag + "_noise"]))
    gold = g[tag + "_kinit_f32"]
    km = KMeans(n_clusters=k, random_state=0)
    assert km.sklearn_compat == "1.0.2"
    rs = ko.check_random_state(0)
    first, u = km._draws(rs, n, gold.shape[0])
    assert rs.random_sample() == float(g[tag + "_kinit_f32_next"])
    cent, picks = o.kpp_greedy_lockstep(dev(x), None, first, u, k)
    assert np.array_equal(picks.cpu().numpy().T, gold)
    assert np.array_equal(cent.cpu().numpy(), x[gold])
    xh = x.astype(np.float16).astype(np.float32)
    x16 = o.f16_exact(dev(xh))
    assert x16 is not None
    starts = 2 if tag in ("c", "c1") else gold.shape[0]
    cent, picks = o.kpp_greedy_lockstep(dev(xh), x16, first[:starts], u[:starts], k)
    rs = ko.check_random_state(0)
    opicks = np.stack([ko.sklearn_kpp(xh, k, rs, compat="1.0.2") for _ in range(starts)])
    assert np.array_equal(picks.cpu().numpy().T, opicks)
    cent2, picks2 = o.kpp_greedy_lockstep(dev(xh), None, first[:starts], u[:starts], k)           # dense path, same rows
    assert torch.equal(picks2, picks) and torch.equal(cent2, cent)


@pytest.mark.parametrize("shape", [(3000, 64, 12, 0.9), (4500, 768, 200, 0.9), (2000, 512, 8, 1.2)])
def test_sklearn_kmeans_c_loops_equal_python_loops(ops, monkeypatch, shape):
    """`KMeans.fit` on fp16-exact rows takes the C loops (scd_kpp_greedy_lockstep with the filter, scd_kmeans_lloyd_run_sk with the
    incremental M-step); labels, centres, inertia and n_iter equal the oracle's and the Python-driven loop's (SCD_LLOYD_RUN=0)."""
    from scd_amd.cluster import KMeans
    n, d, k, noise = shape
    x, _, _ = synth.clustered_features(n, d, k, seed=n % 97, center_seed=d % 89, noise=noise)
    xh = x.astype(np.float16).astype(np.float32)
    for compat, n_init in (("1.0.2", 3), ("1.7.2", "auto")):
        km = KMeans(n_clusters=k, random_state=0, n_init=n_init, sklearn_compat=compat).fit(xh)
        olab, oin, ocent, oit = ko.sklearn_kmeans(xh, k, 0, n_init, compat=compat)
        assert np.array_equal(km.labels_, olab) and km.n_iter_ == oit and np.array_equal(km.cluster_centers_, ocent)
        assert abs(km.inertia_ - oin) <= 1e-9 * oin
        monkeypatch.setenv("SCD_LLOYD_RUN", "0")
        km2 = KMeans(n_clusters=k, random_state=0, n_init=n_init, sklearn_compat=compat).fit(xh)
        monkeypatch.delenv("SCD_LLOYD_RUN")
        assert np.array_equal(km2.labels_, km.labels_) and km2.n_iter_ == km.n_iter_ and km2.inertia_ == km.inertia_
        assert np.array_equal(km2.cluster_centers_, km.cluster_centers_)


def test_mains_run_on_reference_cache_files(ops, tmp_path, monkeypatch, capsys):
    """Real-data mode of both entry points on the reference's cache-file formats: features + classifier + vocabulary are read
    from disk, the cluster dict is written with the reference's keys and read back, class names missing from the vocabulary go
    through the text tower, --cluster KM runs on the device."""
    import importlib
    import scd_amd.clip as clip
    root = str(tmp_path)
    x, y, mask_lab, w = _write_cache_tree(root)
    monkeypatch.setenv("SCD_ROOT", root)
    monkeypatch.setenv("SCD_DATA", os.path.join(root, "data"))
    clip.allow_synthetic(False)
    monkeypatch.setattr(clip, "_tokenizer", None)               # an earlier test may have opted in to the hash tokenizer
    monkeypatch.setenv("SCD_SYNTHETIC", "")
    mu = importlib.import_module("main_unsup")
    common = ["--root_dir", root, "--dataset_name", "cifar10", "--n_cluster", "8", "--topk", "3", "--num_common_vote", "10",
              "--num_common_linear", "2", "--class_names", os.path.join(root, "class_names.json")]
    with pytest.raises(FileNotFoundError, match="BPE"):          # the real tokenizer file is absent: no silent stand-in
        mu.main(common + ["--feat_model", "dino_vit", "--cluster", "KM", "--run_cluster", "true"])
    monkeypatch.setattr(clip, "_tokenizer", None)
    monkeypatch.setenv("SCD_SYNTHETIC", "1")                     # opt in to the hash tokenizer (weights come from the checkpoint)
    cand, u_preds = mu.main(common + ["--feat_model", "dino_vit", "--cluster", "KM", "--run_cluster", "true", "--save_cluster", "true"])
    out = capsys.readouterr().out
    assert "KM Accuracies" in out and "sACC_avg" in out and "IoU" in out and "Missed 1 names" in out
    saved = torch.load(os.path.join(root, "cluster", "KM_dino_vit_cifar10_8.pt"), weights_only=False)
    assert set(saved) == {"all_preds", "u_preds", "u_targets", "mask"} and saved["all_preds"] is None
    assert saved["u_preds"].shape == ((~mask_lab).sum(),) and saved["u_preds"].dtype == np.int32
    assert np.array_equal(saved["u_targets"], y[~mask_lab].astype(np.float64))
    truth = set("noun_%03d" % c for c in range(7))
    assert len(truth & set(cand)) >= 6
    cand2, u_preds2 = mu.main(common + ["--feat_model", "dino_vit", "--cluster", "KM"])            # cluster cache read back
    assert cand2 == cand and np.array_equal(u_preds2, u_preds)
    mp_ = importlib.import_module("main_ptsup")
    pc = common + ["--feat_model", "clip", "--cluster", "SSKM", "--topk", "5"]
    cand3, u_preds3 = mp_.main(pc + ["--run_cluster", "true", "--save_cluster", "true"])
    out = capsys.readouterr().out
    assert "sACC lower bound" in out and "sACC upper bound" in out and "IoU" in out
    saved = torch.load(os.path.join(root, "cluster", "SSKM_clip_cifar10.pt"), weights_only=False)
    assert saved["all_preds"].shape == (len(y),) and set(saved) == {"all_preds", "u_preds", "u_targets", "mask"}
    lab_names = ["noun_%03d" % c for c in range(4)]
    assert set(lab_names) <= set(cand3) and len(cand3) == 8
    cand4, u_preds4 = mp_.main(pc)
    assert cand4 == cand3 and np.array_equal(u_preds4, u_preds3)
    with pytest.raises(SystemExit, match="KM"):
        mp_.main(common + ["--feat_model", "clip", "--cluster", "KM", "--run_cluster", "true"])


def test_extract_feature_writes_reference_dict(ops, tmp_path):
    """Row a2: naming.extract_feature over (images, label, uq_idx, mask_lab) batches = the dict of main_unsup.py:141-146; the
    fused normalise of the encoder's last kernel against the fp32 oracle."""
    import argparse
    from scd_amd import naming
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP, DinoViT
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=2, t_layers=1))
    model = CLIP(sd).cuda()
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn(10, 3, 224, 224, generator=g)
    labels = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2, 3, 4])
    mlab = torch.tensor([1, 1, 0, 0, 0, 1, 0, 0, 0, 0])
    loader = [(imgs[:6], labels[:6], None, mlab[:6]), (imgs[6:], labels[6:], None, mlab[6:])]
    d = naming.extract_feature(model, loader, argparse.Namespace(feat_model="clip", train_classes=range(2)))
    assert set(d) == {"all_feats", "mask_lab", "mask_cls", "targets"}
    assert d["all_feats"].shape == (10, 512) and d["all_feats"].dtype == np.float16
    assert d["mask_lab"].dtype == bool and np.array_equal(d["mask_lab"], mlab.numpy().astype(bool))
    assert np.array_equal(d["mask_cls"], labels.numpy() < 2) and d["targets"].dtype == np.float64
    sd16 = {k: (v.half().float() if v.dim() >= 2 and "positional" not in k and "class_emb" not in k else v) for k, v in sd.items()}
    ref = torch.nn.functional.normalize(co.clip_encode_image(sd16, imgs.half().float()), dim=-1)
    assert _cos(torch.from_numpy(d["all_feats"]).float(), ref).min().item() > 1 - 1e-3
    assert np.allclose(np.linalg.norm(d["all_feats"].astype(np.float32), axis=1), 1.0, atol=2e-3)
    dsd = W.synthetic_dino_state_dict(seed=1, layers=2)
    dd = naming.extract_feature(DinoViT(dsd).cuda(), loader, argparse.Namespace(feat_model="dino_vit", train_classes=[0, 1]))
    assert dd["all_feats"].shape == (10, 768) and dd["all_feats"].dtype == np.float32
    dsd16 = {k: (v.half().float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k else v) for k, v in dsd.items()}
    dref = torch.nn.functional.normalize(co.dino_forward(dsd16, imgs.half().float()), dim=-1)
    assert _cos(torch.from_numpy(dd["all_feats"]), dref).min().item() > 1 - 1e-3


class _OneRankExchange:
    """Stand-in for scd_amd.kmeans._Dist on a world of one (no process group): the exchanges are identities; `fail_after` makes the n-th
    all-reduce / all-gather raise, as a broken collective would."""
    rank, world = 0, 1

    def __init__(self, fail_after=None):
        self.calls, self.fail_after = 0, fail_after

    def _tick(self):
        self.calls += 1
        if self.fail_after is not None and self.calls > self.fail_after:
            raise RuntimeError("collective failed (test)")

    def allreduce_(self, t, op="sum"):
        self._tick()
        return t

    def allgather(self, t):
        self._tick()
        return t.unsqueeze(0).clone()

    def allgather_into(self, out, inp):
        self._tick()
        out.copy_(inp)


def test_sharded_loops_as_a_world_of_one_and_their_error_path(ops):
    """scd_kmeans_lloyd_run_sharded / scd_kpp_seed_lockstep_sharded with identity exchanges (a world of one, no process group) reproduce
    the single-process entry points bit for bit - the callback plumbing itself - and an exception raised inside a callback comes back as
    that exception after the C call has returned SCD_ERCCL (nothing unwinds through the C frames, nothing hangs)."""
    from scd_amd import ops as O
    n, d, k, rr = 20000, 128, 16, 3
    x, y, _ = synth.clustered_features(n, d, k, seed=81, center_seed=82, noise=0.6)
    X = dev(x.astype(np.float16).astype(np.float32))
    data = O.KMeansData(X)
    x16 = O.f16_exact(X)
    # seeding: three restarts, first centres = rows 5, 1700, 19000
    first = torch.tensor([5, 1700, 19000], device="cuda")
    rv = torch.from_numpy(np.random.RandomState(4).rand(k - 1, rr).astype(np.float32)).cuda().contiguous()

    def seed_inputs():
        buf = torch.zeros((rr, k, d), dtype=torch.float32, device="cuda")
        rows = X[first].contiguous()
        buf[:, 0] = rows
        d2 = torch.full((rr, n), float("inf"), dtype=torch.float32, device="cuda")
        O.min_update_multi(X, rows, d2)
        return buf, d2
    b1, d21 = seed_inputs()
    O.kpp_seed_lockstep(X, x16, d21, rv, b1, 1)
    b2, d22 = seed_inputs()
    pk = O.kpp_seed_lockstep_sharded(X, x16, d22, rv, b2, 1, _OneRankExchange())
    assert torch.equal(b1, b2) and torch.equal(d21, d22) and int((pk < 0).sum()) == 0
    b3, d23 = seed_inputs()
    with pytest.raises(RuntimeError, match="collective failed"):
        O.kpp_seed_lockstep_sharded(X, x16, d23, rv, b3, 1, _OneRankExchange(fail_after=7))
    # Lloyd loop
    lb1 = O.LloydBuffers(data, X, x16, k)
    lb1.c0.copy_(b1[0])
    r1 = lb1.run(10, 1e-4)
    ex = _OneRankExchange()
    lb2 = O.LloydBuffers(data, X, x16, k, ex)
    assert lb2.inc
    lb2.c0.copy_(b1[0])
    r2 = lb2.run(10, 1e-4)
    # (bit patterns: this seeding leaves one cluster empty, its centre is 0 / 0 = NaN as in the reference, and NaN != NaN)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[2].view(torch.int32), r2[2].view(torch.int32)) and r1[1] == r2[1] and r1[3] == r2[3] and ex.calls > 3
    lb3 = O.LloydBuffers(data, X, x16, k, _OneRankExchange(fail_after=6))
    lb3.c0.copy_(b1[0])
    with pytest.raises(RuntimeError, match="collective failed"):
        lb3.run(10, 1e-4)
    torch.cuda.synchronize()
    # the handle is still usable afterwards
    r4 = lb1.run(10, 1e-4)
    assert torch.equal(r1[0], r4[0])


def test_rccl_entry_points_single_rank(ops):
    """scd_comm_* / scd_allreduce_centroids / scd_allgather_text through the C ABI on a one-rank communicator (the GPU box has one
    GPU; the multi-rank pattern is covered by tests/test_dist_gloo.py on CPU and by the driver's 8-GPU bench)."""
    uid = ops.Comm.unique_id()
    assert len(uid) == 128
    comm = ops.Comm(0, 1, uid)
    try:
        x, y, cent = synth.clustered_features(2000, 64, 7, seed=5)
        lab = dev(y.astype(np.int32))
        sums, counts, inertia = ops.kmeans_mstep(dev(x), lab, dev(cent), 7, 0)
        s2, c2, i2 = comm.allreduce_centroids(sums, counts, inertia)
        assert torch.equal(s2, sums) and torch.equal(c2, counts) and torch.equal(i2, inertia)
        w = dev(np.random.RandomState(0).randn(33, 512).astype(np.float16))
        assert torch.equal(comm.allgather_text(w), w)
    finally:
        comm.close()


def test_multi_rank_rccl(ops):
    """One process per visible GPU (at most 8), RCCL: sharded SSKM, sharded vote loop and the C collectives equal the
    single-rank results (tests/dist_rccl_worker.py).  The ranks are fresh CHILD processes started through
    torch.distributed.run (never an exec of this process).  On a one-GPU box the same worker runs as a world of one: the
    script and the nccl backend are exercised, the exchange pattern itself is then covered by tests/test_dist_gloo.py."""
    import socket
    import subprocess
    import sys
    n = max(1, min(torch.cuda.device_count(), 8))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_rccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for rank in range(n):
        assert "rank %d ok" % rank in r.stdout


def test_bench_launcher_four_ranks_tiny_gloo(ops):
    """`python bench.py --gpus 4` end to end at a tiny size: bench.py starts torch.distributed.run as a child (the driver's own launch
    line), four ranks share this box's one GPU (gloo collectives on device tensors: RCCL refuses a duplicate device) and
    run the whole sharded job - sharded text-tower vocabulary build + all-gather, encode, similarity, the sharded k-means++ rounds and
    Lloyd loops behind their C calls, the sharded vote - through the barriers and the max-over-ranks timing to rank 0's JSON line.
    What it proves is that the launcher / ranks path of the N > 1 job neither deadlocks nor diverges before an 8-GPU node ever runs
    it.  Four ranks, not eight: a GPU box admits at most six processes on its card, this one included."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SCD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SCD_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--images", "1200", "--n-cluster", "12",
           "--vocab", "1024", "--batch", "665", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["parallelism"] == "dp4" and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vote_iters"] >= 1
    assert abs(d["value"] - 4 * 1200 / (d["ms_per_step"] / 1e3)) <= 1e-2 * d["value"]      # whole-job images over the max-over-ranks time
    assert d["synthetic_name_accuracy"] > 0.5


def test_bench_config_c3_small(ops):
    """`python bench.py --config c3` (BASELINE configs[2]: GCD / DINO + CLIP encode, raw top-5, ConSSKM with the restarts in lock-step on
    host threads, partially supervised vote) at a small size: the command runs to its JSON line, the size bounds hold and the planted
    names are found."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("SCD_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3", "--steps", "1", "--warmup", "1", "--images", "3000", "--n-cluster", "24",
           "--vocab", "2048", "--batch", "665", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["cluster"] == "ConSSKM" and d["n_gpus"] == 1 and d["value"] > 0
    assert d["consskm"]["transport_solves_per_fit"] >= 10 and d["consskm"]["cluster_sizes_min_max"][0] >= 50
    assert d["vote_iters"] >= 1 and d["synthetic_name_accuracy"] > 0.6


def test_bench_config_c1(ops):
    """`python bench.py --config c1` (BASELINE configs[0]: cached DINO + CLIP features, V = 1,000, the shipped `--cluster KM` with K = 200,
    vote loop; no encoder) at its full size - it is 5,994 rows: the command runs to its JSON line, the stages are all there, the planted
    names are found and the line carries the E-step's HBM roofline object (main_unsup.py:362,504-531,568-614)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("SCD_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["cluster"] == "KM" and d["config"]["images_per_gpu"] == 5994 and d["config"]["vocab"] == 1000 and d["config"]["n_cluster"] == 200
    assert set(d["stage_ms_per_step"]) == {"sim_topk", "kmeans", "vote"} and d["value"] > 0
    assert abs(d["value"] - 5994 / (d["ms_per_step"] / 1e3)) <= 1e-2 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["vote_iters"] >= 1
    assert d["synthetic_name_accuracy"] > 0.6


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_config_c5_small(ops, gpus):
    """`python bench.py --config c5` (BASELINE configs[4]: the open-vocabulary classifier built by the text tower INSIDE the step - 80
    prompts per name; two ranks: name shards + all-gather over gloo on the shared GPU -, textual-enhancement top-k and vote) at a small
    size: the JSON line carries the text tower's stage and its executed-FLOP rate, and the planted names are found
    (clip_lang_util.py:96-108, main_unsup.py:518,523)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SCD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SCD_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--images", "1500",
           "--n-cluster", "12", "--vocab", "1536", "--batch", "665", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["config"]["vocab"] == 1536 and "text_tower" in d["stage_ms_per_step"] and d["stage_ms_per_step"]["text_tower"] > 0
    tt = d["secondary_rooflines"][0]
    assert "text tower" in tt["kernel"] and tt["bound"] == "mfma" and 0 < tt["frac"] < 1 and tt["fc1_launches_per_step"] > 0
    assert 0.05 < tt["gemm_gflop_per_prompt_executed"] < 5.96          # trimmed: below the full-length 5.96 GFLOP per prompt
    assert d["vote_iters"] >= 1 and d["synthetic_name_accuracy"] > 0.6


def test_two_ranks_one_gpu_sharded_kmeans(ops):
    """The same worker as two ranks SHARING this box's one GPU (gloo collectives on device tensors; RCCL refuses a duplicate device):
    the sharded SSKM / K-Means fits - lock-step seeding over the three all-gathers, Lloyd loops behind scd_kmeans_lloyd_run_sharded
    with the group's all-reduce as the exchange callback - equal the single-rank fits bit for bit on fp16-exact rows, and the sharded
    vote loop reproduces the single-rank trace.  Two child processes (never an exec of this one)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", SCD_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_rccl_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for rank in range(2):
        assert "rank %d ok" % rank in r.stdout


def test_c4_shape_sskm_k1000(ops):
    """BASELINE configs[3] per-GPU shard (main_unsup.py:350 with --n_cluster 1000): N = 160,146 CLIP-width rows, K = 1000.
    Full-size properties of the single-pass E-step and of one fused Lloyd step - labels = the stand-alone E-step's, exact
    column-total checksum of the M-step sums, 2,048 sampled rows equal the float64 oracle - and, on a sub-sample, the lock-step
    k-means++ seeding of all restarts equal to the sequential one."""
    from scd_amd.kmeans import KMeansEngine
    n, d, k = 160146, 512, 1000
    g = torch.Generator(device="cuda").manual_seed(41)
    cen = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
    y = torch.randint(0, k, (n,), device="cuda", generator=g)
    x = torch.nn.functional.normalize(cen[y] + (0.8 / d ** 0.5) * torch.randn(n, d, device="cuda", generator=g), dim=-1).half().float().contiguous()
    c0 = x[torch.randperm(n, device="cuda", generator=g)[:k]].contiguous()          # data points as centres: zero distances, small margins
    data = ops.KMeansData(x)
    lab, ref = data.estep(c0, return_refined=True)
    assert int(ref.item()) < n // 4
    xs = x.cpu().numpy()
    cs = c0.cpu().numpy()
    rows = np.random.RandomState(2).choice(n, 2048, replace=False)
    rows[:4] = [0, 255, n - 1, n - 146]
    olab, omind, _ = ko.estep(xs[rows], cs)
    assert np.array_equal(lab.cpu().numpy()[rows].astype(np.int64), olab)
    assert np.array_equal(data.rowdist(c0, lab).cpu().numpy()[rows], omind)
    # one fused Lloyd step (E + M + finalize) from the same centres
    x16 = ops.f16_exact(x)
    assert x16 is not None
    bufs = ops.LloydBuffers(data, x, x16, k)
    bufs.c0.copy_(c0)
    bufs.step(bufs.c0, bufs.c[0], bufs.stats[0], False)
    assert torch.equal(bufs.lab32, lab)
    col = x.double().sum(0)
    assert torch.equal(bufs.sums.sum(0), col) or torch.allclose(bufs.sums.sum(0), col, rtol=1e-13, atol=1e-9)
    assert int(bufs.counts.sum().item()) == n
    sums, counts, inertia = ops.kmeans_mstep(x, lab, c0, k, 0, x16=x16)
    assert torch.equal(sums, bufs.sums) and torch.equal(counts, bufs.counts)
    newc, _ = ops.kmeans_finalize(sums, counts, c0)
    assert torch.equal(newc.nan_to_num(7.0), bufs.c[0].nan_to_num(7.0))
    # lock-step seeding == sequential seeding (sub-sample, k = 1000: 999 rounds)
    sub = x[:20000].contiguous()
    a = KMeansEngine(k=k, max_iterations=1, n_init=2, random_state=5)
    dsub = ops.KMeansData(sub)
    lock = a.kpp_lockstep(dsub, None, k, ko.check_random_state(5), 2)
    rs = ko.check_random_state(5)
    for j in range(2):
        seq = a.kpp(sub, k=k, random_state=rs, data=dsub)
        assert torch.equal(seq, lock[j])


@pytest.mark.parametrize("n,d,k,labelled,blobs", [(20000, 512, 40, True, 40), (30000, 768, 100, False, 100), (16000, 768, 60, False, 100),
                                                  (6000, 64, 7, True, 7)])
def test_incremental_mstep_is_bit_identical(ops, n, d, k, labelled, blobs, monkeypatch):
    """On fp16-exact rows the float64 cluster sums are exact, so updating them with the rows whose label changed
    (scd_kmeans_lloyd_step_delta) must give the SAME centres, labels, float32 inertia and iteration count as a fresh M-step per
    iteration - and both equal the float64 oracle's run (faster_mix_k_means_pytorch.py:187-214).
    The 30,000 x 768 case (k = the number of blobs, no labelled rows) is the one in which EVERY restart empties a cluster within its first
    iterations: every loop variant must end such a restart where the reference's NaN arithmetic does (:140-160), keeping the best of
    the iterations so far and reporting max_iterations - before any incremental step has run."""
    from scd_amd.kmeans import KMeansEngine
    x, y, _ = synth.clustered_features(n, d, blobs, seed=61, center_seed=62, noise=0.8)
    x = x.astype(np.float16).astype(np.float32)
    mask = (y < k // 2) & (np.random.RandomState(7).rand(n) < 0.5) if labelled else np.zeros(n, dtype=bool)
    res = {}
    # "1": scd_kmeans_lloyd_run_multi (all restarts' loops in lock-step behind one C call, the default); "streams": the same over four
    # streams; "seq": scd_kmeans_lloyd_run, one restart after the other; "py": the same steps driven from Python; "0": a fresh M-step
    # per iteration
    # "nomerge": lock-step, but every restart's filter launch of its own (at d = 512, K <= 256 the default serves all running restarts'
    # filters with ONE launch per iteration, estep_rbm_kernel)
    for mode in ("1", "streams", "nomerge", "seq", "py", "0"):
        monkeypatch.setenv("SCD_ESTEP_MERGED", "0" if mode == "nomerge" else "1")
        monkeypatch.setenv("SCD_MSTEP_DELTA", "0" if mode == "0" else "1")
        monkeypatch.setenv("SCD_LLOYD_RUN", "0" if mode == "py" else "1")
        monkeypatch.setenv("SCD_LLOYD_LOCKSTEP", "0" if mode == "seq" else "1")
        monkeypatch.setenv("SCD_LLOYD_STREAMS", "4" if mode == "streams" else "1")
        km = KMeansEngine(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=2)
        if labelled:
            km.fit_mix(dev(x[~mask]), dev(x[mask]), dev(y[mask]))
        else:
            km.fit(dev(x))
        st = dict(km.stats)
        lockstep = st.pop("lockstep_fits", 0)
        assert lockstep == (1 if mode in ("1", "streams", "nomerge") else 0)
        res[mode] = (km.labels_.cpu().numpy(), km.cluster_centers_.cpu().numpy(), float(km.inertia_), km.n_iter_, st)
    dying = (n, k, labelled) == (30000, 100, False)
    assert (res["1"][4].get("delta_steps", 0) > 0) == (not dying) and res["0"][4].get("delta_steps", 0) == 0
    if dying:
        assert res["1"][3] == 10 and bool(np.isnan(res["1"][1]).any()) and res["1"][4]["estep_calls"] <= 3 * 4
    # (an empty cluster's centre is NaN in the reference and here: equal_nan)
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1], equal_nan=True)
    assert res["1"][2] == res["0"][2] and res["1"][3] == res["0"][3]
    for other in ("streams", "nomerge", "seq", "py"):
        assert np.array_equal(res["1"][0], res[other][0]) and np.array_equal(res["1"][1], res[other][1], equal_nan=True), other
        assert res["1"][2] == res[other][2] and res["1"][3] == res[other][3] and res["1"][4] == res[other][4], other
    okm = ko.K_Means(k=k, tolerance=1e-4, max_iterations=10, n_init=3, random_state=2)
    if labelled:
        okm.fit_mix(x[~mask], x[mask], y[mask])
    else:
        okm.fit(x)
    assert np.array_equal(res["1"][0], okm.labels_) and np.array_equal(res["1"][1], okm.cluster_centers_, equal_nan=True)
    assert res["1"][2] == float(okm.inertia_)


@pytest.mark.parametrize("n,d,k,R,labelled,tol", [(12011, 512, 100, 10, True, 1e-4), (9000, 448, 130, 5, False, 1e-4), (16000, 512, 37, 12, True, 5e-2),
                                                   (4100, 512, 256, 2, True, 1e-4), (8000, 512, 60, 16, False, 1e30)])
def test_lockstep_merged_estep_equals_per_restart_filters(ops, monkeypatch, n, d, k, R, labelled, tol):
    """The restarts' Lloyd loops in lock-step (scd_kmeans_lloyd_run_multi) with ONE filter launch per iteration for all running restarts
    (estep_rbm_kernel: Dp = 512, Kp <= 256; segments of the unit stream, whole restarts per part) against the same loops with one filter
    launch per restart and against the float64 oracle (faster_mix_k_means_pytorch.py:187-214, :244-275): labels, centres, inertia and
    n_iter_ bit-identical.  Ragged row counts, Kp = 128 and 256, 2-16 restarts (more than eight: several parts), restarts that converge at
    different iterations (the set of running restarts shrinks) and a tolerance that stops every restart after its first iteration."""
    from scd_amd.kmeans import KMeansEngine
    x, y, _ = synth.clustered_features(n, d, k, seed=81, center_seed=82, noise=0.8)
    x = x.astype(np.float16).astype(np.float32)
    mask = (y < k // 2) & (np.random.RandomState(9).rand(n) < 0.5) if labelled else np.zeros(n, dtype=bool)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCD_ESTEP_MERGED", mode)
        km = KMeansEngine(k=k, tolerance=tol, max_iterations=7, n_init=R, random_state=4)
        if labelled:
            km.fit_mix(dev(x[~mask]), dev(x[mask]), dev(y[mask]))
        else:
            km.fit(dev(x))
        assert km.stats.get("lockstep_fits", 0) == 1
        res[mode] = (km.labels_.cpu().numpy(), km.cluster_centers_.cpu().numpy(), float(km.inertia_), km.n_iter_, km.stats["estep_calls"])
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1], equal_nan=True)
    assert res["1"][2:] == res["0"][2:]
    okm = ko.K_Means(k=k, tolerance=tol, max_iterations=7, n_init=R, random_state=4)
    if labelled:
        okm.fit_mix(x[~mask], x[mask], y[mask])
    else:
        okm.fit(x)
    assert np.array_equal(res["1"][0], okm.labels_) and np.array_equal(res["1"][1], okm.cluster_centers_, equal_nan=True)
    assert res["1"][2] == float(okm.inertia_)


def test_workspaces_are_not_overrun(ops, monkeypatch):
    """Every workspace the wrappers size with a *_ws_bytes function gets 4 KB of canary bytes behind it; after the similarity call, a
    C4-shaped and a C2-shaped E-step, M-steps, a whole SSKM fit (seeding rounds through the filter, Lloyd loops in C) and a vote
    histogram the canaries are intact: no kernel writes past the size the header promises (the GPU sanitizers are not available
    on this pool; this is the check that can be had)."""
    from scd_amd.kmeans import KMeansEngine
    guards = []
    real_ws = ops._ws

    def guarded(nbytes, device):
        t = torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=device)
        t[int(nbytes):] = 0xA5
        guards.append((t, int(nbytes)))
        return t

    monkeypatch.setattr(ops, "_ws", guarded)
    ops._kpp_ws.clear()
    g = torch.Generator(device="cuda").manual_seed(3)
    f = torch.nn.functional.normalize(torch.randn(3000, 512, device="cuda", generator=g), dim=-1).half()
    wt = torch.nn.functional.normalize(torch.randn(2100, 512, device="cuda", generator=g), dim=-1).half()
    ops.sim_topk(f, wt, 3, "softmax")
    ops.sim_topk(f[:257], wt, 5, "raw")
    for n, d, k in ((9000, 512, 1000), (7000, 768, 100), (5000, 96, 20)):
        x, y, _ = synth.clustered_features(n, d, min(k, 50), seed=n, noise=0.7)
        x = x.astype(np.float16).astype(np.float32)
        data = ops.KMeansData(dev(x))
        c = dev(x[np.random.RandomState(1).choice(n, k, replace=False)])
        lab = data.estep(c)
        ops.kmeans_mstep(dev(x), lab.to(torch.int32), c, k, 0)
    x, y, _ = synth.clustered_features(20000, 512, 30, seed=5, noise=0.7)
    x = x.astype(np.float16).astype(np.float32)
    mask = (y < 15) & (np.random.RandomState(7).rand(len(y)) < 0.5)
    km = KMeansEngine(k=30, tolerance=1e-4, max_iterations=6, n_init=3, random_state=1)
    km.fit_mix(dev(x[~mask]), dev(x[mask]), dev(y[mask]))
    nidx = torch.randint(0, 2100, (4000, 5), device="cuda", generator=g)
    preds = torch.randint(0, 30, (4000,), device="cuda", generator=g)
    ops.vote_hist(nidx, 5, preds, list(range(30)), 10)
    torch.cuda.synchronize()
    assert len(guards) >= 6
    for t, nb in guards:
        assert bool((t[nb:] == 0xA5).all()), "a kernel wrote past a %d-byte workspace" % nb
    ops._kpp_ws.clear()


def test_c_abi_host_program_without_torch(ops, tmp_path):
    """examples/c_abi_host.cpp - a C++ host that links libscd_hip.so and uses nothing but include/scd_hip.h and hipMalloc - builds
    with hipcc and reproduces float64 host loops: top-3 names + softmax probabilities (main_unsup.py:504-531) and E-step labels
    (faster_mix_k_means_pytorch.py:139-141); it then runs a restart's Lloyd loop over all rows and over two row shards (two host
    threads, two handles, a host-side sum as the exchange callback of scd_kmeans_lloyd_run_sharded) and finds them bit-identical, and
    does the same for the k-means++ rounds (scd_kpp_seed_lockstep against scd_kpp_seed_lockstep_sharded with a host-side all-gather).
    The boundary is usable from compiled code, not only through ctypes."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    lib = os.path.join(ROOT, "scd_amd", "lib")
    exe = str(tmp_path / "c_abi_host")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "examples", "c_abi_host.cpp"), "-L", lib, "-lscd_hip", "-lpthread", "-Wl,-rpath," + lib], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)          # a child process of its own: no exec from this one
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout and "0 label mismatches" in r.stdout and ": 0 differences" in r.stdout and ": 0 seed differences" in r.stdout, r.stdout


@pytest.mark.parametrize("tol,max_it", [(5e-2, 10), (1e30, 10), (1e-4, 1), (1e-4, 2), (0.0, 4)])
def test_lloyd_run_stops_like_the_reference_loop(ops, monkeypatch, tol, max_it):
    """scd_kmeans_lloyd_run's loop control: it stops after the first iteration whose centre shift is below the tolerance (the
    speculative iteration behind it is dropped), honours max_iterations, and keeps the least-inertia iteration - n_iter_, labels,
    centres and inertia equal the Python-driven loop's and the float64 oracle's (faster_mix_k_means_pytorch.py:187-214)."""
    from scd_amd.kmeans import KMeansEngine
    n, d, k = 12000, 128, 8
    x, y, _ = synth.clustered_features(n, d, k, seed=71, center_seed=72, noise=0.5)
    x = x.astype(np.float16).astype(np.float32)
    mask = (y < k // 2) & (np.random.RandomState(3).rand(n) < 0.5)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCD_LLOYD_RUN", mode)
        km = KMeansEngine(k=k, tolerance=tol, max_iterations=max_it, n_init=4, random_state=11)
        km.fit_mix(dev(x[~mask]), dev(x[mask]), dev(y[mask]))
        res[mode] = (km.labels_.cpu().numpy(), km.cluster_centers_.cpu().numpy(), float(km.inertia_), km.n_iter_, km.stats["estep_calls"])
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1], equal_nan=True)
    assert res["1"][2:] == res["0"][2:]
    okm = ko.K_Means(k=k, tolerance=tol, max_iterations=max_it, n_init=4, random_state=11)
    okm.fit_mix(x[~mask], x[mask], y[mask])
    assert np.array_equal(res["1"][0], okm.labels_) and np.array_equal(res["1"][1], okm.cluster_centers_, equal_nan=True)
    assert res["1"][2] == float(okm.inertia_)
    if max_it == 1 or tol == 1e30:
        assert res["1"][4] <= 2 * 4          # at most the speculative second iteration per restart was launched


@pytest.mark.parametrize("n,k", [(75700, 1000), (100000, 520)])
def test_estep_rb_split_last_round(ops, n, k):
    """estep_rb_kernel splits the row blocks of the chip's partial last round over the centres (n = 75,700: 40 of 296 row blocks,
    four parts of 8 units; n = 100,000, K = 520 -> 640 padded: 135 blocks left, no split at 20 units / 2 parts of 10) and merges
    their keys in estep_rb_merge_kernel: sampled rows, the first and the last rows of the split range equal the float64 oracle."""
    g = torch.Generator(device="cuda").manual_seed(n)
    d = 512
    cen = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=-1)
    y = torch.randint(0, k, (n,), device="cuda", generator=g)
    x = torch.nn.functional.normalize(cen[y] + (0.9 / d ** 0.5) * torch.randn(n, d, device="cuda", generator=g), dim=-1).contiguous()
    c0 = x[torch.randperm(n, device="cuda", generator=g)[:k]].contiguous()
    data = ops.KMeansData(x)
    lab = data.estep(c0).cpu().numpy()
    nfull = (n + 255) // 256 // 256 * 256
    rows = np.unique(np.concatenate([np.random.RandomState(1).choice(n, 1500, replace=False), np.arange(nfull * 256 - 3, nfull * 256 + 300),
                                     np.arange(n - 300, n)]))
    olab, _, _ = ko.estep(x.cpu().numpy()[rows], c0.cpu().numpy())
    assert np.array_equal(lab[rows].astype(np.int64), olab)


@pytest.mark.parametrize("n,k,v,topk,m", [(5000, 12, 2100, 3, 10), (20000, 100, 21000, 5, 20), (300, 7, 50, 2, 4)])
def test_vote_table_equals_vote_hist(ops, n, k, v, topk, m):
    """The sharded form of the vote histogram (dense per-rank table -> all-reduce -> most_common) on one rank, and
    split over two row shards with the tables added / min-ed by hand, gives scd_vote_hist's keys and counts - Counter.most_common
    order: count descending, ties in first-seen (global row-major) order."""
    rs = np.random.RandomState(n)
    idx = rs.randint(0, max(2, v // 40), size=(n, 5)).astype(np.int64)            # few distinct names: many ties in the counts
    preds = rs.randint(0, k, size=n).astype(np.int64)
    preds[preds == 3] = 4                                                         # a cluster id nobody predicts
    clusters = sorted(set(preds.tolist()))
    keys, cnt = ops.vote_hist(dev(idx), topk, dev(preds), clusters, m)
    c1, f1 = ops.vote_table(dev(idx), topk, dev(preds), clusters, k, 0, v)
    k1, n1 = ops.vote_table_topm(c1, f1, m)
    assert torch.equal(k1, keys) and torch.equal(n1, cnt)
    cut = n // 3
    ca, fa = ops.vote_table(dev(idx[:cut]), topk, dev(preds[:cut]), clusters, k, 0, v)
    cb, fb = ops.vote_table(dev(idx[cut:]), topk, dev(preds[cut:]), clusters, k, cut, v)
    k2, n2 = ops.vote_table_topm(ca + cb, torch.minimum(fa, fb), m)
    assert torch.equal(k2, keys) and torch.equal(n2, cnt)


def test_get_topk_name_indices_loader_form(ops, capsys):
    """main_unsup.py:43-111: the loader form (encode -> normalise -> sim -> top-5 per batch) returns what the cached-feature form
    returns on the same images, and prints the reference's two accuracy lines."""
    from scd_amd import naming
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import CLIP
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=2), text=False)
    model = CLIP(sd).cuda()
    g = torch.Generator().manual_seed(9)
    imgs = torch.randn(24, 3, 224, 224, generator=g).half()
    tgt = torch.arange(24) % 6
    loader = [(imgs[i:i + 8], tgt[i:i + 8], None, None) for i in range(0, 24, 8)]
    feats = ops.l2norm_rows(model.encode_image(imgs.cuda()))
    rs = np.random.RandomState(4)
    w = rs.randn(512, 300).astype(np.float32)
    w[:, :6] = feats.float().cpu().numpy()[:6].T + 0.05 * rs.randn(512, 6)
    w = (w / np.linalg.norm(w, axis=0, keepdims=True)).astype(np.float16)
    nouns = ["n%03d" % j for j in range(300)]
    cidx = {c: nouns[c] for c in range(6)}
    idx, val = naming.get_topk_name_indices(loader, cidx, nouns, dev(w), model)
    ri, rv = ops.sim_topk(feats, ops.transpose_f16(dev(w)), 5, "raw")
    assert torch.equal(idx, ri) and torch.equal(val, rv)
    out = capsys.readouterr().out
    assert "Top-1 accuracy:" in out and "Top-5 accuracy:" in out
    i2, v2 = naming.get_topk_name_indices_wotarget(loader, cidx, nouns, dev(w), model)
    assert torch.equal(i2, ri) and torch.equal(v2, rv)
