"""Tensor-level wrappers over the C ABI (include/scd_hip.h).

torch is plumbing here: it owns device memory and the current stream; every
computation below is a call into libscd_hip.so.  Nothing in this module falls
back to torch math - a missing library or device raises.
"""
import ctypes as C
import os

import numpy as np
import weakref

import torch

from . import _lib
from ._lib import check, ptr, SCD_F16, SCD_F32, SIM_RAW, SIM_SOFTMAX

_L = _lib.load
_dev = [None]           # device of the tensors of the op being issued: the handle and the stream follow the DATA, not torch's current device


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.ScdError(_lib.SCD_EINVAL, "tensor must live on the HIP device (got %s)" % t.device)
    for t in ts:
        if t is not None:
            _dev[0] = t.device.index
            break


def handle():
    return _lib.handle(_dev[0])


def stream_ptr():
    if _dev[0] is not None and _dev[0] != torch.cuda.current_device():
        raise _lib.ScdError(_lib.SCD_EINVAL, "tensors live on cuda:%d but the current device is cuda:%d (use torch.cuda.device / "
                            "set_device: kernels are launched on the current device's stream)" % (_dev[0], torch.cuda.current_device()))
    return _lib.stream_ptr()


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def trace_mark(end=False):
    """An empty marker kernel on the current stream (scd_mark_begin_kernel / scd_mark_end_kernel): brackets a measured region in a
    rocprofv3 kernel trace."""
    check(_L().scd_trace_mark(handle(), 1 if end else 0, stream_ptr()))


# ----------------------------------------------------------------------------- normalise / layout
def l2norm_rows(x):
    """F.normalize(x, dim=-1) (main_unsup.py:130)."""
    _need_cuda(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    dt = {torch.float32: SCD_F32, torch.float16: SCD_F16}[x.dtype]
    check(_L().scd_l2norm_rows(handle(), ptr(x), dt, x.shape[0], x.shape[1], ptr(out), stream_ptr()))
    return out


def transpose_f16(w):
    """[r,c] fp16 -> [c,r] (zeroshot_weights [512,V] -> name-major Wt [V,512])."""
    _need_cuda(w)
    w = w.contiguous()
    out = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float16, device=w.device)
    check(_L().scd_transpose_f16(handle(), ptr(w), w.shape[0], w.shape[1], ptr(out), stream_ptr()))
    return out


def gather_rows_f16(wt, idx):
    _need_cuda(wt, idx)
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), wt.shape[1]), dtype=torch.float16, device=wt.device)
    check(_L().scd_gather_rows_f16(handle(), ptr(wt), ptr(idx), idx.numel(), wt.shape[1], ptr(out), stream_ptr()))
    return out


def select_rows(feats16, idx, name_idx=None, want16=True, want32=True):
    """`all_feats[mask]` (main_unsup.py:318-321) for the rows `idx` (device int64) of the fp16 feature matrix, in one launch: returns
    (rows fp16 or None, rows float32 or None, name_idx[idx] or None)."""
    _need_cuda(feats16, idx)
    assert feats16.dtype == torch.float16 and feats16.is_contiguous() and idx.dtype == torch.int64
    m, d = idx.numel(), feats16.shape[1]
    dev = feats16.device
    o16 = torch.empty((m, d), dtype=torch.float16, device=dev) if want16 else None
    o32 = torch.empty((m, d), dtype=torch.float32, device=dev) if want32 else None
    k = 0
    on = None
    if name_idx is not None:
        assert name_idx.dtype == torch.int64 and name_idx.is_contiguous()
        k = name_idx.shape[1]
        on = torch.empty((m, k), dtype=torch.int64, device=dev)
    if m:
        check(_L().scd_select_rows(handle(), ptr(feats16), ptr(name_idx), ptr(idx), m, d, k, ptr(o16), ptr(o32), ptr(on), stream_ptr()))
    return o16, o32, on


def mean2_f16(a, b):
    """fp16((a + b) / 2): the textual-enhancement feature (main_unsup.py:518 commented formula)."""
    _need_cuda(a, b)
    a, b = a.to(torch.float16).contiguous(), b.to(torch.float16).contiguous()
    assert a.shape == b.shape
    out = torch.empty_like(a)
    check(_L().scd_mean2_f16(handle(), ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr()))
    return out


# ----------------------------------------------------------------------------- similarity
_frozen_vocabs = {}      # id(wt) -> (weakref, data_ptr, _version, shape, norm tensor, stream pointer, event): ops.freeze_vocab


def _vocab_norm_now(wt):
    out = torch.empty(1, dtype=torch.int32, device=wt.device)
    check(_L().scd_sim_vocab_norm(handle(), ptr(wt), wt.shape[0], wt.shape[1], ptr(out), stream_ptr()))
    return out


def freeze_vocab(wt):
    """The caller's promise that the name-major fp16 vocabulary `wt` will not be written any more (by anybody: torch, a kernel of this
    library writing through the raw pointer, another stream): its max ||w||^2 - the data-dependent part of the similarity filter's
    error bound - is computed once, here, and every later sim_topk on the tensor reuses it.  Without the promise the norm is computed
    per call (12 us beside a 2.7-ms call at C2): torch's `_version` does not see raw-pointer writes, and a stale, too small norm would
    certify rows it must not.  `unfreeze_vocab` takes the promise back.  Returns wt."""
    _need_cuda(wt)
    assert wt.dtype == torch.float16 and wt.is_contiguous() and wt.dim() == 2
    ev = torch.cuda.Event()
    norm = _vocab_norm_now(wt)
    ev.record()
    if len(_frozen_vocabs) > 64:
        for kk in [kk for kk, e in _frozen_vocabs.items() if e[0]() is None]:
            del _frozen_vocabs[kk]
    _frozen_vocabs[id(wt)] = (weakref.ref(wt), wt.data_ptr(), wt._version, tuple(wt.shape), norm, torch.cuda.current_stream().cuda_stream, ev)
    return wt


def unfreeze_vocab(wt):
    _frozen_vocabs.pop(id(wt), None)


def vocab_norm(wt):
    """max_v ||w_v||^2 of a name-major fp16 vocabulary (scd_sim_vocab_norm): the frozen tensor's stored value (ops.freeze_vocab; a
    consumer on another stream first waits for the event recorded behind its computation), a fresh one otherwise.  An in-place torch
    write after the freeze (`_version`) ends the promise as well."""
    ent = _frozen_vocabs.get(id(wt))
    if ent is not None and ent[0]() is wt and ent[1] == wt.data_ptr() and ent[2] == wt._version and ent[3] == tuple(wt.shape):
        if ent[5] != torch.cuda.current_stream().cuda_stream:
            ent[6].wait()                     # the current stream waits for the norm's kernel
        return ent[4]
    if ent is not None:
        del _frozen_vocabs[id(wt)]
    return _vocab_norm_now(wt)


def sim_topk(f, wt, k, mode="raw", scale=100.0, return_fallback=False):
    """Top-k of scale * f @ wt.T per row.  f [n,d] fp16, wt [v,d] fp16 (name-major).
    Returns (idx int64 [n,k], val float32 [n,k])."""
    _need_cuda(f, wt)
    f = f.to(torch.float16).contiguous()
    wt_in = wt
    wt = wt.to(torch.float16).contiguous()
    n, d = f.shape
    v = wt.shape[0]
    idx = torch.empty((n, k), dtype=torch.int64, device=f.device)
    val = torch.empty((n, k), dtype=torch.float32, device=f.device)
    fb = torch.empty(1, dtype=torch.int32, device=f.device) if return_fallback else None      # written by the library when asked for
    nb = _L().scd_sim_topk_ws_bytes(n, d, v, k)
    ws = _ws(nb, f.device)
    m = SIM_SOFTMAX if mode == "softmax" else SIM_RAW
    if wt is wt_in and id(wt) in _frozen_vocabs:
        # a vocabulary the caller froze (ops.freeze_vocab: the full vocabulary of main_unsup.py:504-531, reused by every call): its norm once
        check(_L().scd_sim_topk_prenorm(handle(), ptr(f), ptr(wt), n, d, v, float(scale), k, m, ptr(idx), ptr(val), ptr(fb),
                                        ptr(ws), nb, ptr(vocab_norm(wt)), stream_ptr()))
    else:
        check(_L().scd_sim_topk(handle(), ptr(f), ptr(wt), n, d, v, float(scale), k, m, ptr(idx), ptr(val), ptr(fb),
                                ptr(ws), nb, stream_ptr()))
    if return_fallback:
        return idx, val, fb
    return idx, val


def sim_argmax(f, wsel_t, scale=100.0):
    """argmax(scale * f @ wsel_t.T, -1) (main_unsup.py:603-614): scd_sim_argmax.  Returns (idx int64 [n], val float32 [n])."""
    _need_cuda(f, wsel_t)
    f = f.to(torch.float16).contiguous()
    wt = wsel_t.to(torch.float16).contiguous()
    n, d = f.shape
    v = wt.shape[0]
    idx = torch.empty(n, dtype=torch.int64, device=f.device)
    val = torch.empty(n, dtype=torch.float32, device=f.device)
    nb = _L().scd_sim_topk_ws_bytes(n, d, v, 1)
    ws = _ws(nb, f.device)
    check(_L().scd_sim_argmax(handle(), ptr(f), ptr(wt), n, d, v, float(scale), ptr(idx), ptr(val), ptr(ws), nb, stream_ptr()))
    return idx, val


def prompt_pool(emb, n_names, t_per, out, col0):
    """normalise -> mean -> normalise of each name's prompt embeddings into columns of out [d, V]."""
    _need_cuda(emb, out)
    check(_L().scd_prompt_pool(handle(), ptr(emb), n_names, t_per, emb.shape[1], col0, out.shape[1], ptr(out), stream_ptr()))


# ----------------------------------------------------------------------------- k-means
def kmeans_timing(enable, cap=4096):
    """HIP-event timing of the streaming E-step kernel inside scd_kmeans_estep / scd_kmeans_lloyd_step (measurement aid).
    Returns the durations (ms, numpy, call order) collected since the last call and switches the collection on / off."""
    buf = np.zeros(cap, dtype=np.float64)
    cnt = C.c_int(0)
    check(_L().scd_kmeans_timing(handle(), 1 if enable else 0, ptr(buf), cap, C.byref(cnt)))
    return buf[: min(cnt.value, cap)].copy()


ESTEP_FEW, ESTEP_CENTRES_FROM_FINALIZE, LLOYD_FULL = 1, 2, 8      # include/scd_hip.h
_LAST_FINALIZE = {}                                  # "c": (weakref to the centres kmeans_finalize returned, their _version, the KMeansData)


class KMeansData:
    """X (float32 [n,d], device) plus the prepared fp16 E-step operand."""

    def __init__(self, x):
        _need_cuda(x)
        self.x = x.to(torch.float32).contiguous()
        self.n, self.d = self.x.shape
        self.prep = _ws(_L().scd_kmeans_prep_bytes(self.n, self.d), self.x.device)
        check(_L().scd_kmeans_prepare(handle(), ptr(self.x), self.n, self.d, ptr(self.prep), stream_ptr()))
        self._ws = {}

    def ws(self, key, nbytes):
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            w = self._ws[key] = _ws(nbytes, self.x.device)
        return w

    def estep(self, centers, return_refined=False, expect_few=False):
        """expect_few: few rows are expected inside the filter's error bound (late Lloyd iterations): they are re-evaluated in
        the filter kernel's tail, no refine launch.  Same labels either way."""
        _need_cuda(self.x)
        k = centers.shape[0]
        # the hand-over of kmeans_finalize(data=self) is used only for the very tensor it returned, unmodified since (torch counts
        # in-place writes in `_version`): a matching address alone proves nothing, the allocator recycles addresses
        last = _LAST_FINALIZE.get("c")
        vouch = last is not None and last[0]() is centers and last[1] == centers._version and last[2] is self
        _LAST_FINALIZE.pop("c", None)
        flags = (ESTEP_FEW if expect_few else 0) | (ESTEP_CENTRES_FROM_FINALIZE if vouch else 0)
        check(_L().scd_kmeans_estep_hint(handle(), flags))
        centers = centers.to(torch.float32).contiguous()
        labels = torch.empty(self.n, dtype=torch.int32, device=self.x.device)
        ref = torch.zeros(1, dtype=torch.int32, device=self.x.device) if return_refined else None     # (a fill launch otherwise)
        nb = _L().scd_kmeans_estep_ws_bytes(self.n, self.d, k)
        ws = self.ws(("e", k), nb)
        check(_L().scd_kmeans_estep(handle(), ptr(self.x), ptr(self.prep), ptr(centers), self.n, self.d, k, ptr(labels),
                                    ptr(ref), ptr(ws), nb, stream_ptr()))
        return (labels, ref) if return_refined else labels

    def rowdist(self, centers, labels):
        _need_cuda(self.x)
        centers = centers.to(torch.float32).contiguous()
        out = torch.empty(self.n, dtype=torch.float32, device=self.x.device)
        check(_L().scd_kmeans_rowdist(handle(), ptr(self.x), ptr(centers), ptr(labels), self.n, self.d, centers.shape[0],
                                      ptr(out), stream_ptr()))
        return out

    def min_update(self, c_new, d2):
        _need_cuda(self.x)
        c_new = c_new.to(torch.float32).contiguous()
        check(_L().scd_kmeans_min_update(handle(), ptr(self.x), ptr(c_new), self.n, self.d, ptr(d2), stream_ptr()))

    def dist(self, centers, sqrt=False, with_cost=False):
        _need_cuda(self.x)
        centers = centers.to(torch.float32).contiguous()
        k = centers.shape[0]
        out = torch.empty((self.n, k), dtype=torch.float32, device=self.x.device)
        cost = torch.empty((self.n, k), dtype=torch.int32, device=self.x.device) if with_cost else None
        check(_L().scd_kmeans_dist(handle(), ptr(self.x), ptr(centers), self.n, self.d, k, 1 if sqrt else 0, ptr(out), ptr(cost),
                                   stream_ptr()))
        return (out, cost) if with_cost else out


def f16_exact(x):
    """fp16 copy of x if every value survives the round trip (features that left an fp16 encoder), else None.  One device
    read-back per data set; the copy halves the bytes the M-step streams per Lloyd iteration."""
    _need_cuda(x)
    x = x.to(torch.float32).contiguous()
    if x.numel() % 4 or x.shape[-1] % 2:
        return None
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    res = torch.empty(2, dtype=torch.int32, device=x.device)                  # [inexact blocks | max |x| as float bits]
    check(_L().scd_f16_exact_max(handle(), ptr(x), x.numel(), ptr(out), ptr(res), res.data_ptr() + 4, stream_ptr()))
    host = res.cpu()
    if int(host[0]) != 0:
        return None
    # max |x| rides along on the copy: the incremental M-step's exact-sums argument needs rows * max|x| < 2^29 as well
    # (LloydBuffers.inc); a plain attribute of the tensor object, not of its storage
    out.scd_absmax = float(host[1:].view(torch.float32)[0])
    return out


def kmeans_mstep(x, labels32, c_old, k, split=0, x16=None):
    """(sums float64 [k,d], counts int64 [k], inertia float64 [2]) partials of one rank.  x16: the exact fp16 copy of x
    (f16_exact), streamed instead of x."""
    _need_cuda(x, labels32)
    n, d = x.shape
    sums = torch.empty((k, d), dtype=torch.float64, device=x.device)
    counts = torch.empty(k, dtype=torch.int64, device=x.device)
    inertia = torch.empty(2, dtype=torch.float64, device=x.device)          # zeroed by the library (no torch fill launch)
    nb = _L().scd_kmeans_mstep_ws_bytes(n, d, k)
    ws = _ws(nb, x.device)
    if x16 is not None:
        check(_L().scd_kmeans_mstep_f16(handle(), ptr(x16), ptr(labels32), ptr(c_old), n, d, k, int(split), ptr(sums), ptr(counts),
                                        ptr(inertia), ptr(ws), nb, stream_ptr()))
    else:
        check(_L().scd_kmeans_mstep(handle(), ptr(x), ptr(labels32), ptr(c_old), n, d, k, int(split), ptr(sums), ptr(counts),
                                    ptr(inertia), ptr(ws), nb, stream_ptr()))
    return sums, counts, inertia


def _dd_add(a, b):
    """(hi, lo) + (hi, lo) in double-double (Knuth's two-sum on the high parts); host floats."""
    s_ = a[0] + b[0]
    bb = s_ - a[0]
    e = (a[0] - (s_ - bb)) + (b[0] - bb)
    e += a[1] + b[1]
    hi = s_ + e
    return hi, e - (hi - s_)


class LloydBuffers:
    """Device buffers of KMeansEngine's Lloyd loop through scd_kmeans_lloyd_step[_delta]: two sets (the host reads set i while the
    device fills set i + 1), allocated once per fit.  With an exact fp16 copy of the rows (`cat16`) the M-step can run
    incrementally (`step_delta`: sums / counts updated with the rows whose label changed, inertia from the sums)."""

    def __init__(self, data_u, cat, cat16, k, dd=None):
        """dd: the process group's exchange object (scd_amd.kmeans._Dist: allreduce_(t, op), allgather(t)) when `cat` is one rank's row
        shard - only `run` is shard-aware (scd_kmeans_lloyd_run_sharded), and only when EVERY rank's rows qualify (self.inc)."""
        dev = cat.device
        n_cat, d = cat.shape
        self.data, self.cat, self.cat16, self.k, self.dd = data_u, cat, cat16, k, dd
        self.lab32 = torch.empty(n_cat, dtype=torch.int32, device=dev)
        self.c = [torch.empty((k, d), dtype=torch.float32, device=dev) for _ in range(2)]
        # a run's initial centres get a buffer of their own: the hand-over of scd_kmeans_finalize is keyed on the centre POINTER, and
        # one of self.c may still be registered (with the previous run's last centres) when the next restart begins
        self.c0 = torch.empty((k, d), dtype=torch.float32, device=dev)
        self.sums = torch.empty((k, d), dtype=torch.float64, device=dev)
        self.counts = torch.empty(k, dtype=torch.int64, device=dev)
        # {inertia labelled, inertia unlabelled, centre shift, rows re-evaluated exactly, rows whose label changed}
        self.stats = [torch.zeros(5, dtype=torch.float64, device=dev) for _ in range(2)]
        # incremental exact M-step: every float64 cluster sum must be exact, i.e. rows * max|x| * 2^24 < 2^53 on top of the exact fp16
        # copy (unit-scale features: 1e5 * 1 against 5e8; fp16 values near 65504 in big clusters, or infinities, take the fresh M-step)
        amax = getattr(cat16, "scd_absmax", float("inf")) if cat16 is not None else float("inf")
        n_glob = n_cat
        if dd is not None:
            # the bound is on the GLOBAL cluster sums, and all ranks must take the same path (its collectives differ from the other's):
            # [rows] summed, [max|x|, "my rows do not qualify"] maximised
            tot = dd.allreduce_(torch.tensor([float(n_cat)], dtype=torch.float64, device=dev))
            bad = cat16 is None or not np.isfinite(amax) or data_u.n <= 0
            mx = dd.allreduce_(torch.tensor([0.0 if bad else float(amax), 1.0 if bad else 0.0], dtype=torch.float64, device=dev), op="max")
            tot_h, mx_h = tot.cpu().numpy(), mx.cpu().numpy()
            n_glob, amax = float(tot_h[0]), (float("inf") if mx_h[1] > 0 else float(mx_h[0]))
        # (rank-local sizing only after the ranks have agreed above: KMeansEngine.fit has also agreed that no shard is empty)
        self.nb_e = _L().scd_kmeans_estep_ws_bytes(data_u.n, data_u.d, k)
        self.ws_e = data_u.ws(("e", k), self.nb_e)
        self.nb_m = _L().scd_kmeans_mstep_ws_bytes(n_cat, d, k)
        self.ws_m = _ws(self.nb_m, dev)
        self.inc = (cat16 is not None and k <= 8192 and n_glob * amax < 2.0 ** 29 and os.environ.get("SCD_MSTEP_DELTA", "1") != "0")
        if self.inc:
            if dd is not None:
                self.xbuf = torch.empty(k * d + 2 * k, dtype=torch.float64, device=dev)     # [sums | counts as float64 | int64 counts]
                self._xch_err = None
                view = self.xbuf[: k * d + k]

                def _cb(ctx, buf, n_doubles, stream):
                    # called by scd_kmeans_lloyd_run_sharded between an iteration's M-step and its finalize launch, on the stream the
                    # kernels are enqueued on (torch's current stream, which is what dist.all_reduce orders itself against)
                    try:
                        dd.allreduce_(view)
                        return 0
                    except BaseException as e:          # never let an exception unwind through the C frames
                        self._xch_err = e
                        return 1
                self._xch_cb = _lib.EXCHANGE_FN(_cb)
            self.lab_prev = torch.full((n_cat,), -1, dtype=torch.int32, device=dev)
            self.sumsq = torch.empty(4, dtype=torch.float64, device=dev)
            self.sums_lab = self.counts_lab = None
            self._fit_ready = False
            # scd_kmeans_lloyd_run: iteration i's labels / centres in slot i % 3 of these rings
            self.c_ring = torch.empty((3, k, d), dtype=torch.float32, device=dev)
            self.stats_ring = torch.zeros((2, 5), dtype=torch.float64, device=dev)
            self.lab_ring = torch.empty((3, n_cat), dtype=torch.int32, device=dev)
            self.result = np.zeros(4, dtype=np.float64)

    def step(self, c_in, c_out, stats, expect_few):
        d = self.data
        # the hand-over is vouched for only when c_in is the buffer the previous step of THIS object wrote (private buffers, nothing
        # else writes them); c0 is a fresh seeding no finalize produced, and its address may be a recycled one
        flags = (ESTEP_FEW if expect_few else 0) | (ESTEP_CENTRES_FROM_FINALIZE if c_in is not self.c0 else 0)
        check(_L().scd_kmeans_lloyd_step(handle(), ptr(d.x), ptr(d.prep), d.n, ptr(self.cat), ptr(self.cat16), self.cat.shape[0],
                                         d.d, self.k, ptr(self.lab32), ptr(c_in), ptr(c_out), ptr(self.sums), ptr(self.counts),
                                         ptr(stats), flags, ptr(self.ws_e), self.nb_e, ptr(self.ws_m), self.nb_m,
                                         stream_ptr()))

    def _prepare_fit(self):
        """Once per fit: the rows' sums of squares (labelled | unlabelled, double-double) and the sums / counts of the labelled rows,
        whose labels (self.lab32[:l_num], set by the caller) never change."""
        n_cat, d = self.cat.shape
        l_num = n_cat - self.data.n
        check(_L().scd_kmeans_sumsq(handle(), ptr(self.cat16), None, n_cat, d, l_num, ptr(self.sumsq), stream_ptr()))
        dd = self.dd
        any_lab = l_num > 0
        if dd is not None:
            # global sums of squares: the ranks' double-double pairs added in rank order on the host (once per fit); global sums /
            # counts of the labelled rows (exact sums: any order)
            parts = dd.allgather(self.sumsq).cpu().numpy()
            acc = [(0.0, 0.0), (0.0, 0.0)]
            for r in range(parts.shape[0]):
                acc = [_dd_add(acc[0], (float(parts[r, 0]), float(parts[r, 1]))), _dd_add(acc[1], (float(parts[r, 2]), float(parts[r, 3])))]
            self.sumsq.copy_(torch.tensor([acc[0][0], acc[0][1], acc[1][0], acc[1][1]], dtype=torch.float64))
            any_lab = bool(dd.allreduce_(torch.tensor([float(l_num)], dtype=torch.float64, device=self.cat.device)).item() > 0)
        if any_lab:
            if l_num > 0:
                s, c, _ = kmeans_mstep(self.cat[:l_num], self.lab32[:l_num].contiguous(), None, self.k, 0, x16=self.cat16[:l_num])
            else:
                s = torch.zeros((self.k, d), dtype=torch.float64, device=self.cat.device)
                c = torch.zeros(self.k, dtype=torch.int64, device=self.cat.device)
            if dd is not None:
                dd.allreduce_(s)
                dd.allreduce_(c)
            self.sums_lab, self.counts_lab = s.contiguous(), c.contiguous()
        self._fit_ready = True

    def step_delta(self, c_in, c_out, stats, expect_few, full):
        """One Lloyd iteration with the incremental M-step (full=False) or with a fresh one that also re-bases the incremental
        state (full=True: the first iterations of a restart, or when many labels are moving)."""
        if not self._fit_ready:
            self._prepare_fit()
        d = self.data
        flags = (ESTEP_FEW if expect_few else 0) | (ESTEP_CENTRES_FROM_FINALIZE if c_in is not self.c0 else 0) | (LLOYD_FULL if full else 0)
        check(_L().scd_kmeans_lloyd_step_delta(handle(), ptr(d.x), ptr(d.prep), d.n, ptr(self.cat16), self.cat.shape[0], d.d, self.k,
                                               ptr(self.lab32), ptr(self.lab_prev), ptr(c_in), ptr(c_out), ptr(self.sums), ptr(self.counts),
                                               ptr(self.sums_lab), ptr(self.counts_lab), ptr(self.sumsq), ptr(stats), flags,
                                               ptr(self.ws_e), self.nb_e, ptr(self.ws_m), self.nb_m, stream_ptr()))


    def run(self, max_iter, tol):
        """A whole restart from self.c0 (the seeding; labelled rows' labels in self.lab32[:l_num]) behind one call:
        scd_kmeans_lloyd_run.  Returns (labels int32 [n_cat], float32 inertia, centres [k, d], iterations done, iterations with the
        incremental M-step, iterations launched) of the least-inertia iteration - fresh tensors."""
        if not self._fit_ready:
            self._prepare_fit()
        d = self.data
        n_cat = self.cat.shape[0]
        best_lab = torch.empty(n_cat, dtype=torch.int32, device=self.cat.device)
        best_c = torch.empty((self.k, d.d), dtype=torch.float32, device=self.cat.device)
        args = (handle(), ptr(d.x), ptr(d.prep), d.n, ptr(self.cat16), n_cat, d.d, self.k,
                ptr(self.lab32) if n_cat > d.n else None, ptr(self.lab_ring), ptr(self.lab_prev), ptr(self.c0),
                ptr(self.c_ring), ptr(self.sums), ptr(self.counts), ptr(self.sums_lab), ptr(self.counts_lab),
                ptr(self.sumsq), ptr(self.stats_ring), int(max_iter), float(tol), ptr(best_lab), ptr(best_c),
                self.result.ctypes.data, ptr(self.ws_e), self.nb_e, ptr(self.ws_m), self.nb_m, stream_ptr())
        if self.dd is None:
            check(_L().scd_kmeans_lloyd_run(*args))
        else:
            # a row shard: every iteration's [sums | counts] go through the group's all-reduce (the callback) before the centres are formed
            self._xch_err = None
            rc = _L().scd_kmeans_lloyd_run_sharded(*args, ptr(self.xbuf), self._xch_cb, None)
            if self._xch_err is not None:
                raise self._xch_err
            check(rc)
        r = self.result
        return best_lab, np.float32(r[0]), best_c, int(r[1]), int(r[2]), int(r[3])


    def run_multi(self, c_inits, max_iter, tol, n_streams=0):
        """ALL restarts of the fit in lock-step behind one call (scd_kmeans_lloyd_run_multi): c_inits float32 [R, k, d] are the restarts'
        seedings (labelled rows' labels in self.lab32[:l_num], shared).  Under a process group ONE all-reduce per iteration carries every
        running restart's [sums | counts].  Returns a list of run()'s tuples, restart by restart - the same bits as R calls of run()."""
        if not self._fit_ready:
            self._prepare_fit()
        d = self.data
        dev = self.cat.device
        n_cat, k, dim = self.cat.shape[0], self.k, d.d
        R = int(c_inits.shape[0])
        c_inits = c_inits.to(torch.float32).contiguous()
        hs = _lib.extra_handles(R)
        per = getattr(self, "_multi", None)
        if per is None or per["R"] != R:
            per = self._multi = dict(
                R=R, lab_ring=torch.empty((R, 3, n_cat), dtype=torch.int32, device=dev), lab_prev=torch.full((R, n_cat), -1, dtype=torch.int32, device=dev),
                c_ring=torch.empty((R, 3, k, dim), dtype=torch.float32, device=dev), sums=torch.empty((R, k, dim), dtype=torch.float64, device=dev),
                counts=torch.empty((R, k), dtype=torch.int64, device=dev), stats_ring=torch.zeros((R, 2, 5), dtype=torch.float64, device=dev),
                ws_m=[_ws(self.nb_m, dev) for _ in range(R)],
                result=np.zeros((R, 4), dtype=np.float64))
            # the restarts' E-step workspaces as equal strides of ONE buffer (and the label rings above as one tensor): the library then
            # serves all running restarts' filters with one launch per iteration (estep_rbm_kernel) where the shape allows
            stride = (int(self.nb_e) + 255) // 256 * 256
            per["ws_e_all"] = _ws(R * stride, dev)
            per["ws_e"] = [per["ws_e_all"][j * stride:(j + 1) * stride] for j in range(R)]
            if self.dd is not None:
                xb = per["xbuf"] = torch.empty(R * (k * dim + 2 * k), dtype=torch.float64, device=dev)
                dd = self.dd

                def _cb(ctx, buf, n_doubles, stream):
                    try:
                        dd.allreduce_(xb[: int(n_doubles)])          # the running restarts' [sums | counts], densely packed
                        self._xch_n = getattr(self, "_xch_n", 0) + 1
                        return 0
                    except BaseException as e:          # never let an exception unwind through the C frames
                        self._xch_err = e
                        return 1
                per["cb"] = _lib.EXCHANGE_FN(_cb)
        best_lab = torch.empty((R, n_cat), dtype=torch.int32, device=dev)
        best_c = torch.empty((R, k, dim), dtype=torch.float32, device=dev)
        arr = (_lib.LloydRestart * R)()
        for j in range(R):
            a = arr[j]
            a.h = hs[j].value
            a.lab_ring, a.labels_prev = per["lab_ring"][j].data_ptr(), per["lab_prev"][j].data_ptr()
            a.C_start, a.C_ring = c_inits[j].data_ptr(), per["c_ring"][j].data_ptr()
            a.sums, a.counts, a.stats_ring = per["sums"][j].data_ptr(), per["counts"][j].data_ptr(), per["stats_ring"][j].data_ptr()
            a.best_labels, a.best_C = best_lab[j].data_ptr(), best_c[j].data_ptr()
            a.result_host = per["result"][j].ctypes.data
            a.ws_e, a.ws_m = per["ws_e"][j].data_ptr(), per["ws_m"][j].data_ptr()
        self._xch_err = None
        rc = _L().scd_kmeans_lloyd_run_multi(C.cast(arr, C.c_void_p), R, ptr(d.x), ptr(d.prep), d.n, ptr(self.cat16), n_cat, dim, k,
                                             ptr(self.lab32) if n_cat > d.n else None, ptr(self.sums_lab), ptr(self.counts_lab), ptr(self.sumsq),
                                             int(max_iter), float(tol), self.nb_e, self.nb_m, stream_ptr(),
                                             ptr(per["xbuf"]) if self.dd is not None else None,
                                             C.cast(per["cb"], C.c_void_p) if self.dd is not None else None, None, int(n_streams))
        if self._xch_err is not None:
            raise self._xch_err
        check(rc)
        res = per["result"]
        return [(best_lab[j], np.float32(res[j, 0]), best_c[j], int(res[j, 1]), int(res[j, 2]), int(res[j, 3])) for j in range(R)]

    def run_sk(self, max_iter, tol):
        """sklearn's `_kmeans_single_lloyd` from self.c0 behind one call (scd_kmeans_lloyd_run_sk; no labelled rows).  Returns
        (labels int32 [n], centres [k, d], n_iter) - the E-step of the final centres and those centres, fresh tensors - or None when an
        iteration left a cluster empty (sklearn relocates it; the caller runs that start through its own loop)."""
        if not self._fit_ready:
            self._prepare_fit()
        d = self.data
        n = self.cat.shape[0]
        assert n == d.n and self.inc
        self.lab_prev.fill_(-1)
        lab = torch.empty(n, dtype=torch.int32, device=self.cat.device)
        cen = torch.empty((self.k, d.d), dtype=torch.float32, device=self.cat.device)
        check(_L().scd_kmeans_lloyd_run_sk(handle(), ptr(d.x), ptr(d.prep), d.n, ptr(self.cat16), d.d, self.k, ptr(self.lab_ring),
                                           ptr(self.lab_prev), ptr(self.c0), ptr(self.c_ring), ptr(self.sums), ptr(self.counts),
                                           ptr(self.sumsq), ptr(self.stats_ring), int(max_iter), float(tol), ptr(lab), ptr(cen),
                                           self.result.ctypes.data, ptr(self.ws_e), self.nb_e, ptr(self.ws_m), self.nb_m, stream_ptr()))
        r = self.result
        if r[0] != 0:
            return None
        return lab, cen, int(r[1])


def kpp_greedy_lockstep(x, x16, first, u, k):
    """scikit-learn's greedy k-means++ for R starts in lock-step (scd_kpp_greedy_lockstep).  x float32 [n, d]; x16 its exact fp16 copy
    or None; first: host int64 [R] (the starts' first centres); u: host float64 [R, k - 1, L] (the uniforms each start's RandomState
    slice provides, L per added centre).  Returns (centres float32 [R, k, d], picks int64 [k, R]) on the device."""
    _need_cuda(x)
    n, d = x.shape
    first = np.ascontiguousarray(first, dtype=np.int64)
    r = first.shape[0]
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(r, max(k - 1, 0), -1)
    ell = u.shape[2] if k > 1 else 1
    dev = x.device
    first_d = torch.from_numpy(first).to(dev)
    u_d = torch.from_numpy(np.ascontiguousarray(u.transpose(1, 0, 2))).to(dev) if k > 1 else None       # [k - 1][R][L]
    cbuf = torch.empty((r, k, d), dtype=torch.float32, device=dev)
    picks = torch.empty((k, r), dtype=torch.int64, device=dev)
    nb = _L().scd_kpp_greedy_ws_bytes(n, d, r, ell)
    ws = _ws(nb, dev)
    check(_L().scd_kpp_greedy_lockstep(handle(), ptr(x), ptr(x16), n, d, r, ell, k, ptr(first_d), ptr(u_d), ptr(cbuf), ptr(picks),
                                       ptr(ws), nb, stream_ptr()))
    return cbuf, picks


def kmeans_finalize(sums, counts, c_old=None, shift_mode=0, data=None):
    """shift_mode 0: (sum_k ||dc_k||)^2 (the reference's SSKM test); 1: sum_k ||dc_k||^2 (sklearn's center_shift_tot).
    data (a KMeansData): the centres' E-step operands are produced by the same launch, for the next data.estep(centres)."""
    _need_cuda(sums)
    k, d = sums.shape
    c = torch.empty((k, d), dtype=torch.float32, device=sums.device)
    shift = torch.empty(1, dtype=torch.float64, device=sums.device)
    prep, ws, nb, n = None, None, 0, 0
    if data is not None and data.d == d:
        nb = _L().scd_kmeans_estep_ws_bytes(data.n, d, k)
        prep, ws, n = data.prep, data.ws(("e", k), nb), data.n
    check(_L().scd_kmeans_finalize(handle(), ptr(sums), ptr(counts), k, d, ptr(c_old), ptr(c), ptr(shift), int(shift_mode),
                                   ptr(prep), ptr(ws), nb, n, stream_ptr()))
    if data is not None and data.d == d:
        _LAST_FINALIZE["c"] = (weakref.ref(c), c._version, data)
    else:
        _LAST_FINALIZE.pop("c", None)
    return c, shift


def labels_changed(a, b):
    """Number of rows where two int32 label vectors differ (device int64[1])."""
    _need_cuda(a, b)
    out = torch.empty(1, dtype=torch.int64, device=a.device)
    check(_L().scd_labels_changed(handle(), ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr()))
    return out


def kpp_searchsorted(d2, u):
    """sklearn's k-means++ candidate draw: searchsorted(cumsum_f64(d2), u * float32(sum d2)) for every uniform in u (host
    floats).  Returns (idx int64 [L] on the device, pot float64[1])."""
    _need_cuda(d2)
    uu = torch.as_tensor(np.asarray(u, dtype=np.float64)).to(d2.device)
    idx = torch.empty(uu.numel(), dtype=torch.int64, device=d2.device)
    pot = torch.empty(1, dtype=torch.float64, device=d2.device)
    nb = _L().scd_kpp_draw_ws_bytes(d2.numel())
    ws = _kpp_ws.get((d2.device, nb))
    if ws is None:
        ws = _kpp_ws[(d2.device, nb)] = _ws(nb, d2.device)
    check(_L().scd_kpp_searchsorted(handle(), ptr(d2), d2.numel(), ptr(uu), uu.numel(), ptr(idx), ptr(pot), ptr(ws), nb, stream_ptr()))
    return idx, pot


def kpp_draw(d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
    """Device-side k-means++ draw.  Returns (idx int64[1] or None, probsum float64[1] or None); idx -1 = no hit."""
    _need_cuda(d2)
    idx = torch.empty(1, dtype=torch.int64, device=d2.device) if want_idx else None
    ps = torch.empty(1, dtype=torch.float64, device=d2.device) if want_probsum else None
    nb = _L().scd_kpp_draw_ws_bytes(d2.numel())
    ws = _kpp_ws.get((d2.device, nb))
    if ws is None:
        ws = _kpp_ws[(d2.device, nb)] = _ws(nb, d2.device)
    check(_L().scd_kpp_draw(handle(), ptr(d2), d2.numel(), float(np.float32(r)), ptr(total), ptr(prefix), ptr(idx), ptr(ps),
                            ptr(ws), nb, stream_ptr()))
    return idx, ps


_kpp_ws = {}


def min_update_multi(x, c_new, d2):
    """d2[r] = min(d2[r], ||x - c_new[r]||^2) for the R rows of c_new at once (x read once).  d2 float32 [R, n], in place."""
    _need_cuda(x, c_new, d2)
    c_new = c_new.to(torch.float32).contiguous()
    n, d = x.shape
    check(_L().scd_kmeans_min_update_multi(handle(), ptr(x), ptr(c_new), n, d, c_new.shape[0], ptr(d2), d2.stride(0), stream_ptr()))


def kpp_draw_multi(d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
    """scd_kpp_draw for every row of d2 [R, n] with the uniforms r (host floats [R]).  Returns (idx int64 [R] or None,
    probsum float64 [R] or None)."""
    _need_cuda(d2)
    rr, n = d2.shape
    if torch.is_tensor(r):                                 # already on the device (kpp_lockstep uploads the whole stream once)
        rdev = r.to(device=d2.device, dtype=torch.float32).contiguous()
    else:
        rdev = torch.as_tensor(np.asarray(r, dtype=np.float32)).to(d2.device)
    assert rdev.numel() == rr
    idx = torch.empty(rr, dtype=torch.int64, device=d2.device) if want_idx else None
    ps = torch.empty(rr, dtype=torch.float64, device=d2.device) if want_probsum else None
    nb = rr * _L().scd_kpp_draw_ws_bytes(n)
    ws = _kpp_ws.get((d2.device, nb))
    if ws is None:
        ws = _kpp_ws[(d2.device, nb)] = _ws(nb, d2.device)
    check(_L().scd_kpp_draw_multi(handle(), ptr(d2), n, d2.stride(0), rr, ptr(rdev), ptr(total), ptr(prefix), ptr(idx), ptr(ps),
                                  ptr(ws), nb, stream_ptr()))
    return idx, ps


def kpp_seed_lockstep(x, x16, d2, rv, buf, m0):
    """The rounds of the lock-step k-means++ seeding behind one call (scd_kpp_seed_lockstep): for t < T: draw one row per restart
    from d2 [R, n] with the uniforms rv[t] (float32 [T, R] on the device), store it as centre m0 + t of buf [R, k, d], update d2.
    x16: the exact fp16 copy of x (or None).  Returns the picks, int64 [T, R] on the device (-1: no row drawn)."""
    _need_cuda(x, d2, rv, buf)
    t_rounds, rr = rv.shape
    n, d = x.shape
    assert d2.shape[0] == rr and buf.shape[0] == rr and buf.shape[2] == d and buf.is_contiguous() and rv.is_contiguous()
    picks = torch.empty((t_rounds, rr), dtype=torch.int64, device=x.device)
    nb = _L().scd_kpp_seed_ws_bytes(n, d, rr)
    ws = _kpp_ws.get((x.device, "seed", nb))
    if ws is None:
        ws = _kpp_ws[(x.device, "seed", nb)] = _ws(nb, x.device)
    check(_L().scd_kpp_seed_lockstep(handle(), ptr(x), ptr(x16), n, d, rr, ptr(d2), d2.stride(0), ptr(rv), t_rounds, ptr(buf), buf.shape[1],
                                     int(m0), ptr(picks), ptr(ws), nb, stream_ptr()))
    return picks


def kpp_seed_lockstep_sharded(x, x16, d2, rv, buf, m0, dd):
    """The same rounds over a ROW SHARD behind one call (scd_kpp_seed_lockstep_sharded): the three all-gathers of a round - shard sums,
    shard probability masses, candidate rows - go out through a callback that runs the process group's all-gather on views of one
    exchange buffer.  dd: scd_amd.kmeans._Dist (rank, world, allgather_bytes).  Returns the picks, int64 [T, R] (0, or -1 where no
    shard drew a row)."""
    _need_cuda(x, d2, rv, buf)
    t_rounds, rr = rv.shape
    n, d = x.shape
    assert d2.shape[0] == rr and buf.shape[0] == rr and buf.shape[2] == d and buf.is_contiguous() and rv.is_contiguous()
    dev = x.device
    picks = torch.empty((t_rounds, rr), dtype=torch.int64, device=dev)
    nb = _L().scd_kpp_seed_sharded_ws_bytes(n, d, rr)
    ws = _ws(nb, dev)
    xb = _L().scd_kpp_seed_sharded_xbuf_bytes(d, rr, dd.world)
    xbuf = torch.empty(int(xb), dtype=torch.uint8, device=dev)
    base = xbuf.data_ptr()
    err = []

    def _cb(ctx, send, recv, nbytes, stream):
        # rank w's nbytes at `send` -> recv + w * nbytes on every rank, in stream order (torch's current stream = the kernels' stream)
        try:
            so, ro = send - base, recv - base
            dd.allgather_into(xbuf[ro: ro + dd.world * nbytes], xbuf[so: so + nbytes])
            return 0
        except BaseException as e:            # never unwind through the C frames
            err.append(e)
            return 1
    cb = _lib.GATHER_FN(_cb)
    rc = _L().scd_kpp_seed_lockstep_sharded(handle(), ptr(x), ptr(x16), n, d, rr, ptr(d2), d2.stride(0), ptr(rv), t_rounds, ptr(buf),
                                            buf.shape[1], int(m0), ptr(picks), ptr(ws), nb, stream_ptr(), ptr(xbuf), int(xb), cb, None,
                                            dd.rank, dd.world)
    if err:
        raise err[0]
    check(rc)
    return picks


class UpdateFilter:
    """The distance update of one lock-step seeding round through the MFMA filter (scd_kpp_update_filter), for seedings whose rounds
    are driven from Python (process groups).  One object per seeding: it owns the workspace with the rows' norm table."""

    def __init__(self, x16):
        _need_cuda(x16)
        self.x16 = x16
        self.n, self.d = x16.shape
        self.nb = _L().scd_kpp_update_ws_bytes(self.n, self.d)
        self.ws = _ws(self.nb, x16.device)
        self.first = True

    @staticmethod
    def serves(n, d, restarts):
        dp = (d + 31) // 32 * 32
        return 1 <= restarts <= 16 and d % 32 == 0 and dp in (128, 256, 384, 512, 768)

    def update(self, c_new, d2):
        """d2 [R, ld] = min(d2, ||x - c_new[r]||^2); c_new float32 [R, d] contiguous."""
        assert c_new.dtype == torch.float32 and c_new.is_contiguous() and c_new.shape[1] == self.d and d2.shape[0] == c_new.shape[0]
        check(_L().scd_kpp_update_filter(handle(), ptr(self.x16), self.n, self.d, c_new.shape[0], ptr(c_new), ptr(d2), d2.stride(0),
                                         1 if self.first else 0, ptr(self.ws), self.nb, stream_ptr()))
        self.first = False


def sum_f32_multi(x):
    """float64 row sums of a float32 matrix [R, n] (deterministic order)."""
    _need_cuda(x)
    out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    check(_L().scd_sum_f32_multi(handle(), ptr(x), x.shape[1], x.stride(0), x.shape[0], ptr(out), stream_ptr()))
    return out


def sum_f32(x):
    _need_cuda(x)
    x = x.to(torch.float32).contiguous()
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    check(_L().scd_sum_f32(handle(), ptr(x), x.numel(), ptr(out), stream_ptr()))
    return out


# ----------------------------------------------------------------------------- vote
def vote_hist(name_idx, top_k, preds, clusters, m, known=None):
    """most_common(m) of every cluster's Counter: (keys int64 [C,m], counts int32 [C,m]); -1/0 padded."""
    _need_cuda(name_idx, preds)
    name_idx = name_idx.to(torch.int64).contiguous()
    preds = preds.to(torch.int64).contiguous()
    dev = name_idx.device
    clusters = torch.as_tensor(clusters, dtype=torch.int64, device=dev).contiguous()
    kn = None if known is None or len(known) == 0 else torch.as_tensor(known, dtype=torch.int64, device=dev).contiguous()
    n, ld = name_idx.shape
    nc = clusters.numel()
    keys = torch.empty((nc, m), dtype=torch.int64, device=dev)
    counts = torch.empty((nc, m), dtype=torch.int32, device=dev)
    nb = _L().scd_vote_hist_ws_bytes(n, top_k)
    ws = _ws(nb, dev)
    check(_L().scd_vote_hist(handle(), ptr(name_idx), n, ld, top_k, ptr(preds), ptr(clusters), nc, ptr(kn),
                             0 if kn is None else kn.numel(), m, ptr(keys), ptr(counts), ptr(ws), nb, stream_ptr()))
    return keys, counts


def vote_table(name_idx, top_k, preds, clusters, n_slots, row_offset, v):
    """This rank's dense vote tables for its row shard (scd_vote_table): counts int32 [C, V], first int64 [C, V]
    (global first-seen position, all ones = never).  clusters: the cluster ids voted on, in table-row order."""
    _need_cuda(name_idx, preds)
    name_idx = name_idx.to(torch.int64).contiguous()
    preds = preds.to(torch.int64).contiguous()
    dev = name_idx.device
    nc = len(clusters)
    slot = torch.full((n_slots,), -1, dtype=torch.int32)
    slot[torch.as_tensor(clusters, dtype=torch.int64)] = torch.arange(nc, dtype=torch.int32)
    slot = slot.to(dev)
    counts = torch.empty((nc, v), dtype=torch.int32, device=dev)
    first = torch.empty((nc, v), dtype=torch.int64, device=dev)
    n, ld = name_idx.shape                  # n == 0 (a shard without unlabelled rows) gives empty tables
    check(_L().scd_vote_table(handle(), ptr(name_idx), n, max(ld, top_k), top_k, ptr(preds), ptr(slot), n_slots, int(row_offset), int(v), nc,
                              ptr(counts), ptr(first), stream_ptr()))
    return counts, first


def vote_table_topm(counts, first, m):
    """most_common(m) per cluster from the (all-reduced) tables: (keys int64 [C, m], counts int32 [C, m]); -1 / 0 padded."""
    _need_cuda(counts, first)
    nc, v = counts.shape
    keys = torch.empty((nc, m), dtype=torch.int64, device=counts.device)
    cnt = torch.empty((nc, m), dtype=torch.int32, device=counts.device)
    check(_L().scd_vote_table_topm(handle(), ptr(counts), ptr(first), nc, v, m, ptr(keys), ptr(cnt), stream_ptr()))
    return keys, cnt


# ----------------------------------------------------------------------------- host solvers
def munkres(cost):
    """linear_assignment (cluster_utils.py:234): int array [n,m] -> sorted pairs [min(n,m),2]."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(cost)), dtype=np.int64)
    n, m = x.shape
    out = np.zeros((max(1, min(n, m)), 2), dtype=np.int64)
    npairs = C.c_int(0)
    check(_L().scd_munkres(ptr(x), n, m, ptr(out), C.byref(npairs)))
    res = out[: npairs.value].astype(int)
    res.shape = (-1, 2)
    return res


def munkres_sparse(d, rows, cols, vals):
    """linear_assignment(w.max() - w) for the d x d matrix w with entries w[rows, cols] += vals (assign_name,
    clip_lang_util.py:167-178) without building it: sorted pairs [d,2]."""
    r = np.ascontiguousarray(rows, dtype=np.int32)
    c = np.ascontiguousarray(cols, dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=np.int64)
    out = np.zeros((max(1, d), 2), dtype=np.int64)
    npairs = C.c_int(0)
    check(_L().scd_munkres_sparse(int(d), int(r.size), ptr(r), ptr(c), ptr(v), ptr(out), C.byref(npairs)))
    res = out[: npairs.value].astype(int)
    res.shape = (-1, 2)
    return res


def transport_solve(cost, size_min, size_max):
    """Size-constrained assignment; raises Exception('There was an issue with the min cost flow input.')
    when infeasible (sskm_constrained.py:349-350)."""
    c = np.ascontiguousarray(cost, dtype=np.int32)
    n, k = c.shape
    labels = np.empty(n, dtype=np.int32)
    total = C.c_int64(0)
    rc = _L().scd_transport_solve(ptr(c), n, k, int(size_min), int(size_max), ptr(labels), C.byref(total))
    if rc == _lib.SCD_EINFEASIBLE:
        raise Exception("There was an issue with the min cost flow input.")
    check(rc)
    return labels, total.value


def transport_solve_batch(costs, size_min, size_max, threads=None, labels_out=None):
    """`costs` int32 [B, n, k] (host, C-contiguous): the B problems on up to `threads` host threads (default: the cores this process
    may use, at most B).  Returns (labels int32 [B, n], totals int64 [B]); raises like transport_solve when a problem is infeasible."""
    c = np.ascontiguousarray(costs, dtype=np.int32)
    b, n, k = c.shape
    labels = labels_out if labels_out is not None else np.empty((b, n), dtype=np.int32)
    assert labels.dtype == np.int32 and labels.flags.c_contiguous and labels.shape == (b, n)
    totals = np.zeros(b, dtype=np.int64)
    if threads is None:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    rc = _L().scd_transport_solve_batch(ptr(c), n, k, b, int(size_min), int(size_max), ptr(labels), ptr(totals), int(max(1, threads)))
    if rc == _lib.SCD_EINFEASIBLE:
        raise Exception("There was an issue with the min cost flow input.")
    check(rc)
    return labels, totals


# ----------------------------------------------------------------------------- RCCL exchanges through the C ABI
class Comm:
    """One RCCL communicator on this rank's handle (scd_comm_*): the collectives of the sharded hot path for callers that do
    not go through torch.distributed.  `unique_id` = bytes from Comm.unique_id() on rank 0, shipped to every rank."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_L().scd_comm_unique_id_bytes())
        check(_L().scd_comm_unique_id(buf))
        return buf.raw

    def __init__(self, rank, world, unique_id, device=None):
        _dev[0] = torch.cuda.current_device() if device is None else device
        self.h = handle()
        self.rank, self.world = rank, world
        check(_L().scd_comm_init(self.h, rank, world, C.c_char_p(unique_id)))

    def allreduce_centroids(self, sums, counts, inertia):
        """In-place sum over ranks of the M-step partials; returns (sums, counts int64, inertia)."""
        packed = torch.cat([sums.reshape(-1), counts.to(torch.float64), inertia.reshape(-1)]).contiguous()
        check(_L().scd_allreduce_centroids(self.h, ptr(packed), packed.numel(), _lib.stream_ptr()))
        kd, k = sums.numel(), counts.numel()
        return packed[:kd].reshape(sums.shape), packed[kd:kd + k].round().to(torch.int64), packed[kd + k:]

    def allgather_text(self, wt_shard):
        """wt_shard fp16 [v_shard, d] (equal shards) -> [world * v_shard, d]."""
        wt_shard = wt_shard.to(torch.float16).contiguous()
        out = torch.empty((self.world * wt_shard.shape[0],) + tuple(wt_shard.shape[1:]), dtype=torch.float16, device=wt_shard.device)
        check(_L().scd_allgather_text(self.h, ptr(wt_shard), wt_shard.numel(), ptr(out), _lib.stream_ptr()))
        return out

    def close(self):
        check(_L().scd_comm_destroy(self.h))


# ----------------------------------------------------------------------------- encoders
def gemm_f16(a, w, bias=None, residual=None, act=0):
    _need_cuda(a, w)
    m, k = a.shape
    n = w.shape[0]
    c = torch.empty((m, n), dtype=torch.float16, device=a.device)
    check(_L().scd_gemm_f16(handle(), ptr(a), ptr(w), ptr(bias), ptr(residual), ptr(c), m, n, k, int(act), stream_ptr()))
    return c


class Encoder:
    """Owns the device weights (kept alive here) and the scd_encoder handle."""

    def __init__(self, desc, weights):
        self.desc = desc
        self.weights = weights              # list of tensors / None in the C order
        _need_cuda(next(w for w in weights if w is not None))
        arr = (C.c_void_p * len(weights))(*[None if w is None else w.data_ptr() for w in weights])
        enc = C.c_void_p()
        d = _lib.EncoderDesc(**desc)
        check(_L().scd_encoder_create(handle(), C.byref(d), arr, len(weights), C.byref(enc)))
        self._enc = enc
        self._ws = None
        self.out_dim = desc["out_dim"] if desc["out_dim"] > 0 else desc["width"]
        self.device = next(w for w in weights if w is not None).device

    def __del__(self):
        try:
            if getattr(self, "_enc", None):
                _L().scd_encoder_destroy(self._enc)
        except Exception:
            pass

    def timing(self, enable):
        """Enable/disable HIP-event timing of the fc1 GEMM launches; returns (ms, launches, flop) collected so far."""
        ms, n, fl = C.c_double(0), C.c_int(0), C.c_double(0)
        check(_L().scd_encoder_timing(self._enc, 1 if enable else 0, C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    def _workspace(self, batch):
        nb = _L().scd_encoder_ws_bytes(self._enc, batch)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = _ws(nb, self.device)
        return self._ws, nb

    def encode_image(self, pixels, normalize=False, out=None):
        """out: an fp16 [B, out_dim] contiguous tensor (e.g. a row slice of the feature matrix) the features are written into."""
        _need_cuda(pixels)
        pixels = pixels.contiguous()
        if pixels.dtype not in (torch.float16, torch.float32):
            pixels = pixels.float()
        b = pixels.shape[0]
        if out is None:
            out = torch.empty((b, self.out_dim), dtype=torch.float16, device=pixels.device)
        assert out.dtype == torch.float16 and out.shape == (b, self.out_dim) and out.is_contiguous() and out.device == pixels.device
        ws, nb = self._workspace(b)
        dt = SCD_F16 if pixels.dtype == torch.float16 else SCD_F32
        check(_L().scd_vit_encode_image(handle(), self._enc, ptr(pixels), dt, b, ptr(out), 1 if normalize else 0, ptr(ws), nb,
                                        stream_ptr()))
        return out

    def encode_text(self, tokens, normalize=False, ctx_len=None):
        """tokens int32 [B, 77] on the device.  ctx_len (> every row's EOT position): compute only the first ctx_len positions
        (scd_clip_encode_text_len: same bits, ctx_len / 77 of the work)."""
        _need_cuda(tokens)
        tokens = tokens.to(torch.int32).contiguous()
        b = tokens.shape[0]
        out = torch.empty((b, self.out_dim), dtype=torch.float16, device=tokens.device)
        ws, nb = self._workspace(b)
        if ctx_len is None or ctx_len >= tokens.shape[1]:
            check(_L().scd_clip_encode_text(handle(), self._enc, ptr(tokens), b, ptr(out), 1 if normalize else 0, ptr(ws), nb,
                                            stream_ptr()))
        else:
            check(_L().scd_clip_encode_text_len(handle(), self._enc, ptr(tokens), b, int(ctx_len), ptr(out), 1 if normalize else 0,
                                                ptr(ws), nb, stream_ptr()))
        return out
