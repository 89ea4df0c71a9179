"""Host orchestration of (semi-supervised / size-constrained) K-Means on the HIP ops.

Mirrors K_Means of /root/reference/gcd/methods/clustering/faster_mix_k_means_pytorch.py:47-275 (SSKM) and
/root/reference/local_utils/sskm_constrained.py:15-187 (ConSSKM): same constructor arguments, fit / fit_mix,
labels_ / cluster_centers_ / inertia_ / n_iter_, same RandomState consumption (one rand() per added centre).

Where the reference ships the N x K distance matrix to the host every iteration (:116), this keeps everything
on the device: E-step (MFMA filter + float64 refine), M-step partial sums, centre finalisation and the
k-means++ draws are C-ABI calls; only three scalars per Lloyd iteration (inertia x2, centre shift) reach the host.

Multi-GPU (one process per GPU): when `group` is given, X is this rank's shard; the per-iteration exchange is
ONE all-reduce of [sums | counts | inertia] (float64) - SURVEY.md 8(e) - and the k-means++ draw exchanges the
shard totals.  The compute backend is injectable (`backend=`) so that the collective logic is testable with
gloo on CPU; the default and only product backend is scd_amd.ops (HIP).
"""
import os
import threading
import time

import numpy as np
import torch


def check_random_state(seed):
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r cannot be used to seed a numpy.random.RandomState instance" % seed)


class HipBackend:
    """The product compute backend: every method is a libscd_hip.so call (scd_amd.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def prepare(self, x):
        return self.ops.KMeansData(x)

    def estep(self, data, centers, expect_few=False):
        return data.estep(centers, expect_few=expect_few)

    def rowdist(self, data, centers, labels):
        return data.rowdist(centers, labels)

    def min_update(self, data, c_new, d2):
        data.min_update(c_new, d2)

    def dist(self, data, centers, sqrt=False, with_cost=False):
        return data.dist(centers, sqrt=sqrt, with_cost=with_cost)

    def mstep(self, x, labels32, c_old, k, split, x16=None):
        return self.ops.kmeans_mstep(x, labels32, c_old, k, split, x16=x16)

    def exact_f16(self, x):
        return self.ops.f16_exact(x)

    def finalize(self, sums, counts, c_old, data=None):
        return self.ops.kmeans_finalize(sums, counts, c_old, data=data)

    def lloyd_buffers(self, data_u, cat, cat16, k, dd=None):
        return self.ops.LloydBuffers(data_u, cat, cat16, k, dd)

    def sum_f32(self, x):
        return self.ops.sum_f32(x)

    def kpp_draw(self, d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
        return self.ops.kpp_draw(d2, r, total, prefix, want_idx, want_probsum)

    def min_update_multi(self, data, c_new, d2):
        self.ops.min_update_multi(data.x, c_new, d2)

    def kpp_draw_multi(self, d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
        return self.ops.kpp_draw_multi(d2, r, total, prefix, want_idx, want_probsum)

    def sum_f32_multi(self, x):
        return self.ops.sum_f32_multi(x)

    def kpp_seed_lockstep(self, data, x16, d2, rv, buf, m0):
        return self.ops.kpp_seed_lockstep(data.x, x16, d2, rv, buf, m0)

    def kpp_seed_lockstep_sharded(self, data, x16, d2, rv, buf, m0, dd):
        return self.ops.kpp_seed_lockstep_sharded(data.x, x16, d2, rv, buf, m0, dd)

    def update_filter(self, data, x16, restarts):
        """The filtered distance update for Python-driven seeding rounds (process groups), or None when the shape is not served."""
        if x16 is None or not self.ops.UpdateFilter.serves(data.n, data.d, restarts):
            return None
        return self.ops.UpdateFilter(x16)

    def transport(self, cost, size_min, size_max):
        return self.ops.transport_solve(cost, size_min, size_max)

    def transport_batch(self, costs, size_min, size_max, labels_out=None):
        return self.ops.transport_solve_batch(costs, size_min, size_max, labels_out=labels_out)


class _Dist:
    """Thin wrapper over torch.distributed for the three exchanges k-means needs."""

    def __init__(self, group):
        import torch.distributed as dist
        self.d = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allreduce_(self, t, op="sum"):
        self.d.all_reduce(t, op={"sum": self.d.ReduceOp.SUM, "max": self.d.ReduceOp.MAX, "min": self.d.ReduceOp.MIN}[op], group=self.group)
        return t

    def allgather(self, t):
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.d.all_gather(out, t, group=self.group)
        return torch.stack(out)

    def allgather_cat(self, t):
        """Concatenate variable-length first dimensions in rank order; returns (cat, lens)."""
        lens = self.allgather(torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)).reshape(-1).tolist()
        mx = max(max(lens), 1)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        allp = self.allgather(pad)
        return torch.cat([allp[r, : lens[r]] for r in range(self.world)]), lens

    def allgather_into(self, out, inp):
        """rank w's `inp` (flat, equal sizes) -> out[w * len(inp): (w + 1) * len(inp)] on every rank.  Which form is issued follows from
        the group's backend NAME alone - RCCL ("nccl") has the flat all_gather_into_tensor, every other backend gets all_gather on views -
        so that every rank of every group issues the same collective without a probe: a probe's extra collectives, cached per process,
        would be issued by some ranks of a sub-group and not by others (advisor, round 5)."""
        if str(self.d.get_backend(self.group)) == "nccl":
            self.d.all_gather_into_tensor(out, inp, group=self.group)
        else:
            parts = list(out.view(self.world, -1).unbind(0))
            self.d.all_gather(parts, inp, group=self.group)

    def broadcast_(self, t, src):
        self.d.broadcast(t, src=self.d.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
        return t


class KMeansEngine:
    """Shared implementation; the two reference-named K_Means classes configure it."""

    constrained = False

    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, init="k-means++", n_init=10, random_state=None, n_jobs=None,
                 pairwise_batch_size=None, backend=None, group=None):
        self.k = k
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.init = init
        self.n_init = n_init
        self.random_state = random_state
        self.n_jobs = n_jobs
        self.pairwise_batch_size = pairwise_batch_size
        self.backend = backend
        self.group = group
        if max_iterations < 1:
            raise ValueError("max_iterations must be >= 1 (got %r)" % (max_iterations,))
        self.stats = {"estep_calls": 0, "refined_rows": 0}
        self._one_novel = False      # fit_mix with exactly one cluster without labelled rows (see _lloyd_pipelined)

    # ------------------------------------------------------------------ helpers
    def _be(self):
        if self.backend is None:
            self.backend = HipBackend()
        return self.backend

    def _dist(self):
        return _Dist(self.group) if self.group is not None else None

    def _agree_shards(self, x):
        """Under a process group: the one precondition every sharded path shares (each rank owns at least one row to cluster) is
        checked on ALL ranks at once, before any rank-local check, workspace sizing or collective - a rank that raised on its own
        would leave its peers inside the next collective.  One all-reduce (MIN) per fit."""
        dd = self._dist()
        if dd is None:
            return
        n_min = dd.allreduce_(torch.tensor([len(x)], dtype=torch.int64, device=x.device), op="min")
        if int(n_min) <= 0:
            raise ValueError("a rank of the process group owns no row to cluster: every shard needs at least one (re-balance the shards)")

    def _class_means(self, l, l_targets):
        """l_centers = per-class mean in torch.unique order (sskm_constrained.py:88-96)."""
        classes = torch.unique(l_targets)
        dd0 = self._dist()
        if dd0:
            classes = torch.unique(dd0.allgather_cat(classes)[0])
        rank = torch.searchsorted(classes, l_targets).to(torch.int32)
        sums, counts, _ = self._be().mstep(l, rank.contiguous(), None, int(classes.numel()), 0)
        dd = self._dist()
        if dd:
            dd.allreduce_(sums)
            dd.allreduce_(counts)
        cent, _ = self._be().finalize(sums, counts, None)
        return classes, rank.to(torch.int64), cent

    def kpp(self, X, pre_centers=None, k=10, random_state=None, data=None):
        """K_Means.kpp (sskm_constrained.py:28-44): incremental, all on device; one rand() per added centre."""
        rs = check_random_state(random_state)
        be = self._be()
        dd = self._dist()
        if data is None:
            data = be.prepare(X)
        x = data.x
        if pre_centers is not None:
            C = pre_centers.reshape(-1, x.shape[1]).to(torch.float32)
        else:
            first = rs.randint(0, self._global_len(x, dd))
            C = self._fetch_row(x, first, dd).reshape(1, -1)
        labels = be.estep(data, C)
        d2 = be.rowdist(data, C, labels)
        picks = []
        m = C.shape[0]
        if m < k:                                    # centres are appended in place (a torch.cat per centre re-copied all of C)
            buf = torch.empty((k, x.shape[1]), dtype=torch.float32, device=x.device)
            buf[:m] = C
        while m < k:
            r = rs.rand()
            if dd is None:
                idx, _ = be.kpp_draw(d2, r)
                row = x.index_select(0, idx.clamp(min=0)).reshape(1, -1)
                picks.append(idx)
            else:
                # three dependent exchanges per centre (sum -> normalised prefix -> row), each ONE small all-gather, and no
                # host synchronisation: prob = d2 / float32(total) needs the global total before the shard's probability
                # mass, the pick needs every earlier shard's mass, and the next sum needs the picked row.
                tot = dd.allgather(be.sum_f32(d2)).reshape(-1).sum().reshape(1)              # fixed rank order: same bits everywhere
                _, ps = be.kpp_draw(d2, r, total=tot, want_idx=False, want_probsum=True)
                allps = dd.allgather(ps).reshape(-1)
                prefix = allps[: dd.rank].sum().reshape(1) if dd.rank > 0 else torch.zeros_like(ps)
                idx, _ = be.kpp_draw(d2, r, total=tot, prefix=prefix.contiguous())
                cand = x.index_select(0, idx.clamp(min=0)).reshape(-1)
                pack = dd.allgather(torch.cat([(idx >= 0).to(torch.float32), cand]))          # [world, 1 + D]: (hit flag | row)
                hit = pack[:, 0] > 0
                firsthit = (hit & (torch.cumsum(hit.to(torch.int32), 0) == 1)).to(torch.int64)   # one-hot of the first owner
                row = pack[:, 1:].index_select(0, firsthit.argmax().reshape(1))
                picks.append(torch.where(hit.any(), 0, -1).reshape(1))
            buf[m] = row.reshape(-1)
            m += 1
            C = buf[:m]
            be.min_update(data, buf[m - 1], d2)
        if picks and bool((torch.cat([p.reshape(-1) for p in picks]) < 0).any()):
            # the reference indexes an empty nonzero() here (sskm_constrained.py:42)
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
        return C

    def kpp_lockstep(self, data, pre_centers, k, rs, restarts, x16=None):
        """The k-means++ seedings of all `restarts` restarts of one fit, advanced together: restart j's t-th centre depends only
        on restart j's earlier centres and on the (t)-th uniform of ITS slice of the random stream, and the reference's restarts
        consume the stream back to back (kpp draws k-m uniforms, Lloyd draws none; sskm.py:28-44, :190-204), so drawing the
        whole stream up front and adding centre t of every restart in one round gives the same centres as running kpp
        `restarts` times.  Per round: one pass over X for all restarts' distance updates and one batched draw; under a process
        group the same three all-gathers as kpp, each carrying `restarts` values.  Without a process group the rounds run behind
        one call (scd_kpp_seed_lockstep; x16 = the exact fp16 copy of data.x when there is one: the distance update then reads it
        through a filter).  Returns float32 [restarts, k, D]."""
        be = self._be()
        dd = self._dist()
        x = data.x
        n, d = x.shape
        dev = x.device
        if pre_centers is not None:
            c0 = pre_centers.reshape(-1, d).to(torch.float32)
            m = c0.shape[0]
            if m >= k:
                return c0.unsqueeze(0).expand(restarts, -1, -1).contiguous()
            rv = rs.rand(restarts, k - m)                                   # C order: restart 0's draws first
            buf = torch.empty((restarts, k, d), dtype=torch.float32, device=dev)
            buf[:, :m] = c0
            labels = be.estep(data, c0)
            d2 = be.rowdist(data, c0, labels).reshape(1, -1).expand(restarts, -1).contiguous()
        else:
            n_glob = self._global_len(x, dd)
            firsts, rv = [], np.empty((restarts, k - 1))
            for j in range(restarts):                                       # fit_once: randint, then k-1 uniforms, per restart
                firsts.append(rs.randint(0, n_glob))
                rv[j] = rs.rand(k - 1)
            m = 1
            buf = torch.empty((restarts, k, d), dtype=torch.float32, device=dev)
            rows = torch.stack([self._fetch_row(x, f, dd) for f in firsts]).to(torch.float32)
            buf[:, 0] = rows
            d2 = torch.full((restarts, n), float("inf"), dtype=torch.float32, device=dev)
            be.min_update_multi(data, rows.contiguous(), d2)
        if k - m == 0:                 # k = 1: the first centre is the whole seeding (no rounds, no draws)
            return buf
        picks = []
        ar = torch.arange(restarts, device=dev)
        # float32(uniform) is what the draw compares with (the reference's `cumsum(prob) >= r` promotes r to prob's float32)
        rv = torch.from_numpy(np.ascontiguousarray(rv.T.astype(np.float32))).to(dev)       # [k - m, restarts], one upload
        if dd is None and hasattr(be, "kpp_seed_lockstep") and os.environ.get("SCD_KPP_SEED_RUN", "1") != "0":
            pk = be.kpp_seed_lockstep(data, x16, d2, rv, buf, m)
            if pk.numel() and bool((pk < 0).any()):
                raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
            return buf
        if (dd is not None and hasattr(be, "kpp_seed_lockstep_sharded") and x.is_cuda and os.environ.get("SCD_KPP_SEED_RUN", "1") != "0"):
            # a row shard: the rounds behind one call as well, the three all-gathers of a round handed in as a callback
            pk = be.kpp_seed_lockstep_sharded(data, x16, d2, rv, buf, m, dd)
            self.stats["sharded_seedings"] = self.stats.get("sharded_seedings", 0) + 1
            if pk.numel() and bool((pk < 0).any()):
                raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
            return buf
        # Python-driven rounds (a process group: three all-gathers sit between a round's draw and its update): with an exact fp16 copy
        # the update still goes through the MFMA filter of the C loop (same float32 results, half the bytes, a fraction of the float64
        # work) once a new centre wins few rows
        filt = be.update_filter(data, x16, restarts) if (hasattr(be, "update_filter") and os.environ.get("SCD_KPP_FILTER", "1") != "0") else None
        for t in range(k - m):
            r = rv[t]
            if dd is None:
                idx, _ = be.kpp_draw_multi(d2, r)
                rows = x.index_select(0, idx.clamp(min=0))
                picks.append(idx)
            else:
                tot = dd.allgather(be.sum_f32_multi(d2)).sum(0).contiguous()                   # [restarts], fixed rank order
                _, ps = be.kpp_draw_multi(d2, r, total=tot, want_idx=False, want_probsum=True)
                allps = dd.allgather(ps)
                prefix = allps[: dd.rank].sum(0).contiguous() if dd.rank > 0 else torch.zeros_like(ps)
                idx, _ = be.kpp_draw_multi(d2, r, total=tot, prefix=prefix)
                cand = x.index_select(0, idx.clamp(min=0)).to(torch.float32)
                pack = dd.allgather(torch.cat([(idx >= 0).to(torch.float32).reshape(-1, 1), cand], 1))   # [world, restarts, 1 + D]
                hit = pack[:, :, 0] > 0
                firsthit = (hit & (torch.cumsum(hit.to(torch.int32), 0) == 1)).to(torch.int64)
                rows = pack[firsthit.argmax(0), ar, 1:]
                picks.append(torch.where(hit.any(0), 0, -1))
            rows = rows.to(torch.float32).contiguous()
            buf[:, m + t] = rows
            if t + 1 < k - m:
                if filt is not None and m + t >= 8:
                    filt.update(rows, d2)
                else:
                    be.min_update_multi(data, rows, d2)
        if picks and bool((torch.stack(picks) < 0).any()):
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
        return buf

    def _lockstep(self):
        import os
        be = self._be()
        return (self.init == "k-means++" and self.n_init > 1 and hasattr(be, "kpp_draw_multi")
                and os.environ.get("SCD_KPP_LOCKSTEP", "1") != "0")

    @staticmethod
    def _global_len(x, dd):
        if dd is None:
            return len(x)
        n = torch.tensor([len(x)], dtype=torch.int64, device=x.device)
        return int(dd.allreduce_(n))

    @staticmethod
    def _fetch_row(x, gidx, dd):
        if dd is None:
            return x[gidx].clone()
        lens = dd.allgather(torch.tensor([len(x)], dtype=torch.int64, device=x.device)).reshape(-1).tolist()
        owner, off = 0, gidx
        while off >= lens[owner]:
            off -= lens[owner]
            owner += 1
        row = x[off].clone() if dd.rank == owner else torch.empty(x.shape[1], dtype=x.dtype, device=x.device)
        return dd.broadcast_(row.contiguous(), owner)

    # E-step on the unlabelled rows -> (int32 labels on device, float32 inertia contribution or None)
    def _assign(self, data, centers, it=0):
        self.stats["estep_calls"] += 1
        return self._be().estep(data, centers, expect_few=it >= 2), None

    def _per_fit(self, data_u, cat):
        """What one fit()/fit_mix() call derives from its data and hands to every restart: the exact fp16 copy of the rows the
        M-step streams (None when a value does not survive the round trip) and the fused Lloyd-step buffers.  Built per CALL and
        passed down - never cached on the engine under a data pointer: X may be rewritten in place between two fits, and a freed
        `cat` is usually re-allocated at the same address."""
        be = self._be()
        cat16 = be.exact_f16(cat) if hasattr(be, "exact_f16") else None
        bufs = None
        if (not self.constrained and cat.is_cuda and hasattr(be, "lloyd_buffers")):
            # (under a process group too: the buffers then carry the group's exchange, and every rank learns whether ALL shards qualify
            # for the C-side loop - collectives inside, so every rank must get here)
            bufs = be.lloyd_buffers(data_u, cat, cat16, self.k, self._dist())
        return dict(cat16=cat16, bufs=bufs)

    def _lloyd(self, data_u, cat, labels, l_num, centers, cat16=None, bufs=None):
        """Iterations shared by fit_once / fit_mix_once (sskm_constrained.py:110-138)."""
        if not self.constrained and cat.is_cuda:
            return self._lloyd_pipelined(data_u, cat, cat16, labels, l_num, centers, bufs)
        return self._lloyd_sequential(data_u, cat, labels, l_num, centers, cat16)

    def _lloyd_sequential(self, data_u, cat, labels, l_num, centers, cat16=None):
        """One iteration at a time, the host reading every iteration's statistics: the constrained engine, host backends, and the loop
        that follows the reference through an EMPTIED cluster step by step.  The reference's centre of a cluster without rows is NaN
        (the mean of no rows, faster_mix_k_means_pytorch.py:147-150, :199-203) and its `torch.min(dist, dim=1)` (:140, :192) then returns
        NaN at the first NaN column for every row: all unlabelled rows go to the lowest-numbered empty cluster, the iteration's inertia
        is NaN (never the best), its centre shift NaN (never below the tolerance)."""
        be = self._be()
        dd = self._dist()
        best = (None, None, None)
        it = 0
        dead = -1                                    # lowest-numbered cluster the previous M-step left empty
        for it in range(self.max_iterations):
            old = centers
            if dead >= 0 and not self.constrained:
                u_lab = torch.full((len(cat) - l_num,), dead, dtype=labels.dtype, device=labels.device)
                u_inertia = np.float32("nan")
            else:
                u_lab, u_inertia = self._assign(data_u, old, it)
            labels[l_num:] = u_lab.to(labels.dtype)
            lab32 = labels.to(torch.int32).contiguous()
            sums, counts, inertia2 = be.mstep(cat, lab32, old, self.k, l_num, cat16) if cat16 is not None else be.mstep(cat, lab32, old, self.k, l_num)
            if dd:
                packed = torch.cat([sums.reshape(-1), counts.to(torch.float64), inertia2])
                dd.allreduce_(packed)
                kd = sums.numel()
                sums = packed[:kd].reshape(sums.shape)
                counts = packed[kd:kd + self.k].round().to(torch.int64)
                inertia2 = packed[kd + self.k:]
            centers, shift = be.finalize(sums, counts, old, data_u)
            empty = counts == 0
            first_empty = torch.where(empty.any(), empty.to(torch.int64).argmax(), torch.full((), -1, dtype=torch.int64, device=counts.device))
            host = torch.cat([inertia2, shift.reshape(1), first_empty.reshape(1).to(inertia2.dtype)]).cpu().numpy()   # the only per-iteration D2H
            ui = np.float32(host[1]) if u_inertia is None else np.float32(u_inertia)
            inertia = np.float32(ui + np.float32(host[0]))
            if dead >= 0 and not self.constrained:
                inertia = np.float32("nan")
            if best[1] is None or inertia < best[1]:
                best = (labels.clone(), inertia, centers.clone())
            dead = int(host[3])
            if host[2] < self.tolerance:
                break
        return best[0], best[1], best[2], it + 1

    def _lloyd_pipelined(self, data_u, cat, cat16, labels, l_num, centers, bufs=None):
        """The same iterations with the host one iteration behind the device: iteration i + 1 is launched (from iteration i's
        centres, which is what the sequential loop would use if i has not converged) before iteration i's {inertia, shift} are
        read back through a pinned buffer; if i turns out to have converged, i + 1 is dropped unseen.  The device never idles
        for the read-back (212 -> ~120 us per iteration at C2); labels / centres / inertia / n_iter are those of the sequential
        loop (every iteration's labels are snapshot on the device, 1 MB, so that the best-inertia bookkeeping can lag)."""
        be = self._be()
        dd = self._dist()
        dev = cat.device
        ring = getattr(self, "_host_ring", None)
        if ring is None:
            ring = self._host_ring = [torch.zeros(5, dtype=torch.float64).pin_memory() for _ in range(2)]
        best = (None, None, None)
        pending = None                      # (it, labels snapshot, new centres, host buffer, event)
        n_done = 0
        # without a process group the three calls of an iteration go out as ONE (scd_kmeans_lloyd_step): the Python call overhead
        # of an iteration (185 us) otherwise exceeds its device time (105 us)
        fused = None
        if self._one_novel:
            # exactly one cluster without labelled rows: the one configuration in which the reference's loop can RECOVER from an emptied
            # cluster (every row goes to it for one iteration, after which no centre is NaN any more).  Unreachable in practice - its seed
            # is a row that stays with it - and followed step by step rather than built into the fast loops
            return self._lloyd_sequential(data_u, cat, labels, l_num, centers, cat16)
        if (dd is not None and bufs is not None and getattr(bufs, "inc", False) and getattr(bufs, "dd", None) is not None
                and os.environ.get("SCD_LLOYD_RUN", "1") != "0"):
            # a row shard whose ranks ALL qualify for the exact incremental M-step: the loop below behind one C call per restart
            # (scd_kmeans_lloyd_run_sharded), the packed all-reduce of an iteration handed in as a callback; same decisions on every rank
            bufs.lab32[:l_num] = labels[:l_num]
            bufs.c0.copy_(centers)
            lab, inertia, cen, n_done, n_delta, n_launched = bufs.run(self.max_iterations, self.tolerance)
            self.stats["estep_calls"] += n_launched
            self.stats["delta_steps"] = self.stats.get("delta_steps", 0) + n_delta
            self.stats["sharded_runs"] = self.stats.get("sharded_runs", 0) + 1
            return lab.to(labels.dtype), inertia, cen, n_done
        if dd is None and hasattr(be, "lloyd_buffers"):
            fused = bufs if bufs is not None else be.lloyd_buffers(data_u, cat, cat16, self.k)
            fused.lab32[:l_num] = labels[:l_num]
            fused.c0.copy_(centers)
            centers = fused.c0              # iteration 0 reads the run's own start buffer and writes set 0
            if getattr(fused, "inc", False) and os.environ.get("SCD_LLOYD_RUN", "1") != "0":
                # the whole loop below behind one C call (same pipelining, same bookkeeping): the Python loop's ~45 us per
                # iteration were a third of an iteration's wall time
                lab, inertia, cen, n_done, n_delta, n_launched = fused.run(self.max_iterations, self.tolerance)
                self.stats["estep_calls"] += n_launched
                self.stats["delta_steps"] = self.stats.get("delta_steps", 0) + n_delta
                return lab.to(labels.dtype), inertia, cen, n_done

        # SCD_ESTEP_FEW (flagged rows re-evaluated in the filter kernel's tail) pays only when a handful of rows are flagged: the
        # one-block-per-CU tail takes ~100 us for a few thousand rows where the refine launch takes 20.  The cue is the count the
        # host has seen last (iteration i - 2 when launching iteration i: the host runs one iteration behind the device)
        refined_seen = [None]
        died = [False]
        # the incremental M-step (LloydBuffers.step_delta) pays while few labels move; same cue, the count of iteration i - 2
        changed_seen = [None, None]        # [seen last, the one before]
        n_u = data_u.n if hasattr(data_u, "n") else len(cat) - l_num

        def settle(p):
            nonlocal best
            p[4].synchronize()
            host = p[3].numpy()
            refined_seen[0] = float(host[3])
            changed_seen[1] = changed_seen[0]
            changed_seen[0] = float(host[4])
            inertia = np.float32(np.float32(host[1]) + np.float32(host[0]))
            if best[1] is None or inertia < best[1]:
                best = (p[1], inertia, p[2].clone() if fused is not None else p[2])
            if np.isnan(host[2]):
                # a NaN shift = a NaN centre = a cluster this iteration's M-step left empty: the reference's loop is dead from here on
                # (_lloyd_sequential's account; it runs on to max_iterations without ever improving or converging)
                died[0] = True
                return True
            return bool(host[2] < self.tolerance)

        for it in range(self.max_iterations):
            old = centers
            buf = ring[it & 1]
            if fused is not None:
                centers, stats = fused.c[it & 1], fused.stats[it & 1]
                self.stats["estep_calls"] += 1
                few = it >= 2 and refined_seen[0] is not None and refined_seen[0] <= 64
                if getattr(fused, "inc", False):
                    # (the same rule as scd_kmeans_lloyd_run: the count seen last is two iterations old, extrapolated with the last ratio)
                    pred = changed_seen[0]
                    if it >= 3 and changed_seen[1] is not None and changed_seen[1] > 0 and pred is not None and pred < changed_seen[1]:
                        pred = pred * (changed_seen[0] / changed_seen[1]) * (changed_seen[0] / changed_seen[1])
                    full = it < 2 or pred is None or pred > max(256, n_u // 32)
                    fused.step_delta(old, centers, stats, few, full)
                    self.stats["delta_steps"] = self.stats.get("delta_steps", 0) + (0 if full else 1)
                else:
                    fused.step(old, centers, stats, few)
                buf.copy_(stats, non_blocking=True)
                snap = fused.lab32.clone()
            else:
                u_lab, _ = self._assign(data_u, old, it)
                labels[l_num:] = u_lab.to(labels.dtype)
                lab32 = labels.to(torch.int32).contiguous()
                sums, counts, inertia2 = be.mstep(cat, lab32, old, self.k, l_num, cat16) if cat16 is not None else be.mstep(cat, lab32, old, self.k, l_num)
                if dd:
                    packed = torch.cat([sums.reshape(-1), counts.to(torch.float64), inertia2])
                    dd.allreduce_(packed)
                    kd = sums.numel()
                    sums = packed[:kd].reshape(sums.shape)
                    counts = packed[kd:kd + self.k].round().to(torch.int64)
                    inertia2 = packed[kd + self.k:]
                centers, shift = be.finalize(sums, counts, old, data_u)
                buf[:3].copy_(torch.cat([inertia2, shift.reshape(1)]), non_blocking=True)
                snap, centers = labels.clone(), centers.clone()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            cur = (it, snap, centers, buf, ev)
            if pending is not None:
                n_done = pending[0] + 1
                if settle(pending):          # iteration `it` was launched on speculation: drop it
                    pending = None
                    break
            pending = cur
        if pending is not None:
            n_done = pending[0] + 1
            settle(pending)
        if died[0]:
            n_done = self.max_iterations
        return best[0].to(labels.dtype), best[1], best[2], n_done

    # ------------------------------------------------------------------ reference API
    def fit_once(self, X, random_state, data=None, init_centers=None, cat16=None, bufs=None):
        be = self._be()
        if data is None:
            data = be.prepare(X)
        x = data.x
        if self.init != "k-means++" and self.group is not None:
            # rows would be drawn from the LOCAL shard: every rank would start from different centres
            raise NotImplementedError("init=%r is not shard-aware; use init='k-means++' with a process group" % (self.init,))
        if init_centers is not None:
            centers = init_centers
        elif self.init == "k-means++":
            centers = self.kpp(x, k=self.k, random_state=random_state, data=data)
        elif self.init == "random":
            rs = check_random_state(self.random_state)
            idx = rs.choice(len(x), self.k, replace=False)
            centers = x[torch.as_tensor(idx, device=x.device)].clone()
        else:
            centers = x[: self.k].clone()
        labels = torch.empty(len(x), dtype=torch.int64, device=x.device)
        return self._lloyd(data, x, labels, 0, centers, cat16, bufs)

    def fit_mix_once(self, u_feats, l_feats, l_targets, random_state, data=None, cat=None, init_centers=None, l_rank=None, cat16=None,
                     bufs=None):
        be = self._be()
        if data is None:
            data = be.prepare(u_feats)
        u = data.x
        l = l_feats.to(device=u.device, dtype=torch.float32).contiguous()
        l_targets = l_targets.to(u.device)
        if cat is None:
            cat = torch.cat((l, u)).contiguous()
        if init_centers is None:
            classes, l_rank, l_centers = self._class_means(l, l_targets)
        l_num = len(l_targets)
        labels = torch.empty(len(cat), dtype=torch.int64, device=u.device)
        labels[:l_num] = l_rank
        centers = init_centers if init_centers is not None else self.kpp(u, l_centers, k=self.k, random_state=random_state, data=data)
        lab, inertia, cent, _ = self._lloyd(data, cat, labels, l_num, centers, cat16, bufs)
        # reference quirk: returns `i + 1` with i the stale labelled-row index (sskm_constrained.py:104,139)
        return lab, inertia, cent, l_num

    def _run(self, once, *args, inits=None, **kw):
        """n_init restarts, best inertia kept (sskm.py:190-204).  `inits(rs)` -> per-restart keyword dicts when the seedings were
        drawn in lock-step (kpp_lockstep); otherwise every restart runs its own kpp on the shared stream."""
        rs = check_random_state(self.random_state)
        best_inertia = None
        per = inits(rs) if inits is not None else [{} for _ in range(self.n_init)]
        bufs = kw.get("bufs")
        dd = self._dist()
        if (inits is not None and len(per) > 1 and not self.constrained and not self._one_novel and bufs is not None and getattr(bufs, "inc", False)
                and hasattr(bufs, "run_multi") and (dd is None or getattr(bufs, "dd", None) is not None)
                and os.environ.get("SCD_LLOYD_RUN", "1") != "0" and os.environ.get("SCD_LLOYD_LOCKSTEP", "1") != "0"):
            # the restarts' Lloyd loops in lock-step behind ONE call (scd_kmeans_lloyd_run_multi): iteration i of every restart still running
            # is enqueued before iteration i - 1 of each is settled, and under a process group one all-reduce per iteration carries all of
            # them (n_init times fewer collectives per fit).  Every restart runs exactly the launches of its own loop: same bits
            mix = once == self.fit_mix_once
            l_num = len(args[2]) if mix else 0
            if mix:
                bufs.lab32[:l_num] = per[0]["l_rank"]
            c_inits = torch.stack([p["init_centers"] for p in per])
            x0 = getattr(bufs, "_xch_n", 0)
            # four library-owned streams on one GPU (the restarts' latency-bound launches overlap each other's filters: 11.3 -> 9.4-9.9 ms per
            # C2 stage); ONE stream under a process group unless asked otherwise - the collective's latency dominates there, and the
            # multi-stream exchange has only ever run over gloo (tests/dist_rccl_worker.py), never over RCCL with more than one rank
            n_streams = int(os.environ.get("SCD_LLOYD_STREAMS", "4" if dd is None else "1"))
            res = bufs.run_multi(c_inits, self.max_iterations, self.tolerance, n_streams=n_streams)
            self.stats["lockstep_fits"] = self.stats.get("lockstep_fits", 0) + 1
            if dd is not None:                  # all-reduces of the fit's Lloyd loops: one per lock-step iteration, whatever n_init is
                self.stats["lloyd_exchanges"] = self.stats.get("lloyd_exchanges", 0) + getattr(bufs, "_xch_n", 0) - x0
            for lab, inertia, cen, n_done, n_delta, n_launched in res:
                self.stats["estep_calls"] += n_launched
                self.stats["delta_steps"] = self.stats.get("delta_steps", 0) + n_delta
                if dd is not None:
                    self.stats["sharded_runs"] = self.stats.get("sharded_runs", 0) + 1
                if best_inertia is None or inertia < best_inertia:
                    self.labels_ = lab.to(torch.int64)
                    self.cluster_centers_ = cen.clone()
                    best_inertia = inertia
                    self.inertia_ = torch.tensor(float(inertia), dtype=torch.float32)
                    self.n_iter_ = l_num if mix else n_done
            return
        for extra in per:
            labels, inertia, centers, n_iters = once(*args, rs, **kw, **extra)
            if best_inertia is None or inertia < best_inertia:
                self.labels_ = labels.clone()
                self.cluster_centers_ = centers.clone()
                best_inertia = inertia
                self.inertia_ = torch.tensor(float(inertia), dtype=torch.float32)
                self.n_iter_ = n_iters

    def fit(self, X):
        self._one_novel = False
        self._agree_shards(X)
        data = self._be().prepare(X)
        per = self._per_fit(data, data.x)
        inits = None
        if self._lockstep():
            def inits(rs):
                c = self.kpp_lockstep(data, None, self.k, rs, self.n_init, x16=per["cat16"])
                return [dict(init_centers=c[j]) for j in range(self.n_init)]
        self._run(self.fit_once, X, data=data, inits=inits, **per)

    def fit_mix(self, u_feats, l_feats, l_targets):
        self._agree_shards(u_feats)
        data = self._be().prepare(u_feats)
        classes = torch.unique(torch.as_tensor(l_targets).to(data.x.device))
        if self._dist() is not None:                 # the labelled rows are sharded too: the class set is the union (as in _class_means)
            classes = torch.unique(self._dist().allgather_cat(classes)[0])
        self._one_novel = not self.constrained and self.k - int(classes.numel()) == 1
        l = l_feats.to(device=data.x.device, dtype=torch.float32).contiguous()
        cat = torch.cat((l, data.x)).contiguous()
        per = self._per_fit(data, cat)
        inits = None
        if self._lockstep():
            def inits(rs):
                _, l_rank, l_centers = self._class_means(l, l_targets.to(data.x.device))
                x16 = per["cat16"][len(l):] if per["cat16"] is not None else None       # the unlabelled rows' part of the copy
                c = self.kpp_lockstep(data, l_centers, self.k, rs, self.n_init, x16=x16)
                return [dict(init_centers=c[j], l_rank=l_rank) for j in range(self.n_init)]
        self._run(self.fit_mix_once, u_feats, l_feats, l_targets, data=data, cat=cat, inits=inits, **per)
        self.cluster_centers_ = self.cluster_centers_.type_as(u_feats) if torch.is_floating_point(u_feats) else self.cluster_centers_


_STAGING = {}            # (R, n_u, k) -> [host_cost, host_lab, in_use]: pinned staging of the lock-step ConSSKM fit, kept between fits
_STAGING_LOCK = threading.Lock()
_POOL = {}


def _pinned_staging(R, n_u, k):
    """Pinned host buffers for the restarts' cost matrices and labels.  Allocating 43 MB of pinned memory costs milliseconds per fit
    (C3: ten restarts x 9,000 x 120 int32), so one set per shape is kept and handed to one fit at a time; a second fit of the same
    shape running concurrently gets buffers of its own."""
    key = (R, n_u, k)
    with _STAGING_LOCK:
        ent = _STAGING.get(key)
        if ent is not None and not ent[2]:
            ent[2] = True
            return ent[0], ent[1], lambda: ent.__setitem__(2, False)
    hc = torch.empty((R, n_u, k), dtype=torch.int32).pin_memory()
    hl = torch.empty((R, n_u), dtype=torch.int32).pin_memory()
    with _STAGING_LOCK:
        if key not in _STAGING:
            if len(_STAGING) >= 4:
                _STAGING.clear()
            ent = _STAGING[key] = [hc, hl, True]
            return hc, hl, lambda: ent.__setitem__(2, False)
    return hc, hl, lambda: None


def _solver_pool(R):
    """Host threads of the lock-step fit's flow problems (one per restart), created once per process."""
    from concurrent.futures import ThreadPoolExecutor
    p = _POOL.get("p")
    if p is None or p._max_workers < R:
        p = _POOL["p"] = ThreadPoolExecutor(max_workers=max(R, 1), thread_name_prefix="scd-transport")
    return p


class ConstrainedEngine(KMeansEngine):
    """sskm_constrained.K_Means: the E-step is the min-cost-flow assignment (:116, :226-274).
    The flow problem has global capacity constraints, so it does not shard: with `group` set the int32 costs are
    gathered to rank 0, solved there, and the labels scattered back ('replicas only' sub-step)."""

    constrained = True

    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, size_min=100, size_max=1000, init="k-means++", n_init=10,
                 random_state=None, n_jobs=None, pairwise_batch_size=None, backend=None, group=None):
        super().__init__(k, tolerance, max_iterations, init, n_init, random_state, n_jobs, pairwise_batch_size, backend, group)
        self.size_min = size_min
        self.size_max = size_max

    def _assign(self, data, centers, it=0):
        be = self._be()
        d_sqrt, cost = be.dist(data, centers, sqrt=True, with_cost=True)
        dd = self._dist()
        if dd is None:
            labels_np, _ = be.transport(cost.cpu().numpy(), self.size_min, self.size_max)
            labels = torch.from_numpy(labels_np).to(cost.device)
        else:
            lens = dd.allgather(torch.tensor([cost.shape[0]], dtype=torch.int64, device=cost.device)).reshape(-1).tolist()
            mx = max(lens)
            pad = torch.zeros((mx, cost.shape[1]), dtype=torch.int32, device=cost.device)
            pad[: cost.shape[0]] = cost
            allc = dd.allgather(pad)
            lab_all = torch.zeros((dd.world, mx), dtype=torch.int32, device=cost.device)
            if dd.rank == 0:
                full = torch.cat([allc[r, : lens[r]] for r in range(dd.world)]).cpu().numpy()
                lab_np, _ = be.transport(full, self.size_min, self.size_max)
                off = 0
                for r in range(dd.world):
                    lab_all[r, : lens[r]] = torch.from_numpy(lab_np[off:off + lens[r]]).to(cost.device)
                    off += lens[r]
            dd.broadcast_(lab_all, 0)
            labels = lab_all[dd.rank, : cost.shape[0]].contiguous()
        # distances[:] = D[arange, labels] ** 2 in float32, inertia = sum (:271-272)
        picked = d_sqrt.gather(1, labels.to(torch.int64).reshape(-1, 1)).reshape(-1)
        sq = (picked * picked).contiguous()
        tot = be.sum_f32(sq)
        if dd:
            dd.allreduce_(tot)
        return labels.to(torch.int32), np.float32(float(tot))

    # ------------------------------------------------------------------ the restarts' Lloyd loops in lock-step
    def _run(self, once, *args, inits=None, **kw):
        """The n_init restarts of a fit (sskm_constrained.py:165-176) share nothing but X once their seedings are drawn (kpp_lockstep
        draws them all up front from the one random stream), and a restart's iteration is dominated by its flow problem on ONE host
        core (sskm_constrained.py:116 -> OR-Tools in the reference, scd_transport_solve here).  So the restarts advance together:
        per iteration the cost matrices of all restarts still running are formed on the device, copied out in one go, solved on as
        many host threads (scd_transport_solve_batch), and the M-steps follow.  Every restart performs exactly the operations of
        the sequential loop (KMeansEngine._lloyd) on its own state, so labels, centres, inertia and the winner are the same bits;
        `SCD_CONSSKM_LOCKSTEP=0` runs the restarts one after the other.  Under a process group the restarts stay sequential (the
        flow problem is gathered to rank 0 per iteration: _assign)."""
        be = self._be()
        data = kw.get("data")
        # (the lock-step form stages every running restart's int32 cost matrix in pinned host memory at once: beyond 2 GB - n_init x N_u x K x 4
        # bytes - the restarts run one after the other)
        if (inits is None or self._dist() is not None or data is None or not hasattr(be, "transport_batch")
                or not getattr(data.x, "is_cuda", False) or os.environ.get("SCD_CONSSKM_LOCKSTEP", "1") == "0"
                or self.n_init * data.n * self.k * 4 > (2 << 30)):
            return super()._run(once, *args, inits=inits, **kw)
        rs = check_random_state(self.random_state)
        per = inits(rs)
        mix = once == self.fit_mix_once
        cat = kw["cat"] if mix else data.x
        l_num = len(args[2]) if mix else 0
        l_rank = per[0]["l_rank"] if mix else None
        res = self._lloyd_lockstep(data, cat, kw.get("cat16"), l_num, l_rank, [p["init_centers"] for p in per])
        best_inertia = None
        for labels, inertia, centers, n_iters in res:        # restart order, strict < : the sequential loop's winner
            if best_inertia is None or inertia < best_inertia:
                self.labels_ = labels.clone()
                self.cluster_centers_ = centers.clone()
                best_inertia = inertia
                self.inertia_ = torch.tensor(float(inertia), dtype=torch.float32)
                self.n_iter_ = l_num if mix else n_iters     # (fit_mix: the reference's stale-index quirk, sskm_constrained.py:104,139)

    def _lloyd_lockstep(self, data_u, cat, cat16, l_num, l_rank, c_init):
        be = self._be()
        dev = cat.device
        R, k, n_u = len(c_init), self.k, data_u.n
        labels = []
        for _ in range(R):
            lab = torch.empty(len(cat), dtype=torch.int64, device=dev)
            if l_num:
                lab[:l_num] = l_rank
            labels.append(lab)
        centers = [c for c in c_init]
        best = [(None, None, None)] * R
        active = list(range(R))
        host_cost, host_lab, release = _pinned_staging(R, n_u, k)
        n_iters = [0] * R
        prof = os.environ.get("SCD_CONSSKM_PROFILE", "0") == "1"     # phase wall times into self.stats (adds a device sync per iteration)
        ph = self.stats.setdefault("phase_ms", {"cost_d2h": 0.0, "solve_wait": 0.0, "mstep": 0.0}) if prof else None
        pool = _solver_pool(R)

        def solve(a, ev):
            # a host thread per running restart: waits for ITS cost matrix only (the other restarts' distance kernels and copies are
            # still in flight), then solves; both the event wait and the C call release the GIL
            ev.synchronize()
            be.transport_batch(host_cost[a:a + 1].numpy(), self.size_min, self.size_max, labels_out=host_lab[a:a + 1].numpy())

        try:
            if prof or os.environ.get("SCD_CONSSKM_ASYNC", "1") == "0":
                return self._lockstep_iterations(be, dev, data_u, cat, cat16, l_num, k, labels, centers, best, active, n_iters, host_cost, host_lab,
                                                 pool, solve, prof, ph)
            return self._async_iterations(be, dev, data_u, cat, cat16, l_num, k, labels, centers, best, n_iters, host_cost, host_lab, pool, solve)
        finally:
            release()

    def _async_iterations(self, be, dev, data_u, cat, cat16, l_num, k, labels, centers, best, n_iters, host_cost, host_lab, pool, solve):
        """The restarts' loops without a common beat (round 6).  In lock-step every iteration ended with one read-back for all restarts:
        the device idled while the last solve finished, and the host threads idled while the device formed the next cost matrices
        (phases of a C3 fit: distances + copies 24 ms, waiting for solves 19 ms, M-steps 15 ms, one after the other).  Here a restart is
        a small state machine - cost matrix issued -> solving on its host thread -> M-step and its four statistics issued -> statistics
        back (an event) -> next iteration - and this thread drives all of them round robin, so that one restart's solve runs under
        another's distance kernel and a third's M-step.  Each restart still performs exactly the sequential loop's operations on its
        own state in its own order (the device work of all restarts shares the one stream, in issue order), so labels, centres,
        inertia and iteration counts are the same bits; `SCD_CONSSKM_ASYNC=0` (or the phase profile) selects the lock-step loop."""
        import concurrent.futures as cf
        R = len(centers)
        SOLVE, STATS, DONE = 0, 1, 2
        phase, it_of = [DONE] * R, [0] * R
        fut, ev, d_sqrt, c_new = [None] * R, [None] * R, [None] * R, [None] * R
        host_stats = torch.empty((R, 4), dtype=torch.float64).pin_memory()

        def start_iter(j):
            d_sqrt[j], cost = be.dist(data_u, centers[j], sqrt=True, with_cost=True)
            host_cost[j].copy_(cost, non_blocking=True)
            e = torch.cuda.Event()
            e.record()
            fut[j] = pool.submit(solve, j, e)
            phase[j] = SOLVE
            self.stats["transport_solves"] = self.stats.get("transport_solves", 0) + 1

        def after_solve(j):
            u_lab = host_lab[j].to(dev, non_blocking=True)
            # distances[:] = D[arange, labels] ** 2 in float32, inertia = sum (sskm_constrained.py:271-272)
            picked = d_sqrt[j].gather(1, u_lab.to(torch.int64).reshape(-1, 1)).reshape(-1)
            tot = be.sum_f32((picked * picked).contiguous())
            labels[j][l_num:] = u_lab.to(torch.int64)
            lab32 = labels[j].to(torch.int32).contiguous()
            old = centers[j]
            sums, counts, inertia2 = be.mstep(cat, lab32, old, k, l_num, cat16) if cat16 is not None else be.mstep(cat, lab32, old, k, l_num)
            c_new[j], shift = be.finalize(sums, counts, old, None)       # (no E-step follows: no centre operands to prepare)
            stats = torch.cat([inertia2.to(torch.float64), shift.reshape(1).to(torch.float64), tot.reshape(1).to(torch.float64)])
            host_stats[j].copy_(stats, non_blocking=True)
            ev[j] = torch.cuda.Event()
            ev[j].record()
            d_sqrt[j] = None
            phase[j] = STATS

        def after_stats(j):
            host = host_stats[j].numpy().copy()
            centers[j] = c_new[j]
            ui = np.float32(np.float32(float(host[3])))
            inertia = np.float32(ui + np.float32(host[0]))
            if best[j][1] is None or inertia < best[j][1]:
                best[j] = (labels[j].clone(), inertia, centers[j].clone())
            n_iters[j] = it_of[j] + 1
            if (not host[2] < self.tolerance) and it_of[j] + 1 < self.max_iterations:
                it_of[j] += 1
                start_iter(j)
            else:
                phase[j] = DONE

        try:
            for j in range(R):
                start_iter(j)
            while any(p != DONE for p in phase):
                progressed = False
                for j in range(R):
                    if phase[j] == SOLVE and fut[j].done():
                        fut[j].result()                                   # (a failed solve raises here, as the batch call did)
                        after_solve(j)
                        progressed = True
                    elif phase[j] == STATS and ev[j].query():
                        after_stats(j)
                        progressed = True
                if not progressed:
                    waiting = [fut[j] for j in range(R) if phase[j] == SOLVE]
                    if waiting:
                        cf.wait(waiting, timeout=2e-4, return_when=cf.FIRST_COMPLETED)
                    else:
                        time.sleep(2e-5)
        except BaseException:
            for f in fut:                                                 # no solver may still be writing the staging buffers when they are released
                if f is not None:
                    f.cancel()
            for f in fut:
                if f is not None and not f.cancelled():
                    try:
                        f.result()
                    except BaseException:
                        pass
            raise
        return [(best[j][0], best[j][1], best[j][2], n_iters[j]) for j in range(R)]

    def _lockstep_iterations(self, be, dev, data_u, cat, cat16, l_num, k, labels, centers, best, active, n_iters, host_cost, host_lab, pool, solve,
                             prof, ph):
        R = len(centers)
        for it in range(self.max_iterations):
            a_n = len(active)
            d_sqrts, futs = [], []
            t0 = time.perf_counter()
            for a, j in enumerate(active):
                d_sqrt, cost = be.dist(data_u, centers[j], sqrt=True, with_cost=True)
                host_cost[a].copy_(cost, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                futs.append(pool.submit(solve, a, ev))
                d_sqrts.append(d_sqrt)
            if prof:
                torch.cuda.current_stream(dev).synchronize()
            t1 = time.perf_counter()
            self.stats["transport_solves"] = self.stats.get("transport_solves", 0) + a_n
            stats, new_c = [], []
            wait_s = 0.0
            try:
                for a, j in enumerate(active):                   # restart order: restart a's M-step runs while the later solves finish
                    tw = time.perf_counter()
                    futs[a].result()                             # (a failed solve raises here, as the batch call did)
                    wait_s += time.perf_counter() - tw
                    u_lab = host_lab[a].to(dev, non_blocking=True)
                    # distances[:] = D[arange, labels] ** 2 in float32, inertia = sum (sskm_constrained.py:271-272)
                    picked = d_sqrts[a].gather(1, u_lab.to(torch.int64).reshape(-1, 1)).reshape(-1)
                    tot = be.sum_f32((picked * picked).contiguous())
                    labels[j][l_num:] = u_lab.to(torch.int64)
                    lab32 = labels[j].to(torch.int32).contiguous()
                    old = centers[j]
                    sums, counts, inertia2 = be.mstep(cat, lab32, old, k, l_num, cat16) if cat16 is not None else be.mstep(cat, lab32, old, k, l_num)
                    c_new, shift = be.finalize(sums, counts, old, data_u)
                    new_c.append(c_new)
                    stats.append(torch.cat([inertia2.to(torch.float64), shift.reshape(1).to(torch.float64), tot.reshape(1).to(torch.float64)]))
            except BaseException:
                for f in futs:                                   # no solver may still be writing the staging buffers when they are released
                    f.cancel()
                for f in futs:
                    if not f.cancelled():
                        try:
                            f.result()
                        except BaseException:
                            pass
                raise
            host = torch.stack(stats).cpu().numpy()            # the iteration's only read-back: [A, 4]
            if prof:
                ph["cost_d2h"] += (t1 - t0) * 1e3
                ph["solve_wait"] += wait_s * 1e3
                ph["mstep"] += (time.perf_counter() - t1 - wait_s) * 1e3
            still = []
            for a, j in enumerate(active):
                centers[j] = new_c[a]
                ui = np.float32(np.float32(float(host[a, 3])))
                inertia = np.float32(ui + np.float32(host[a, 0]))
                if best[j][1] is None or inertia < best[j][1]:
                    best[j] = (labels[j].clone(), inertia, centers[j].clone())
                n_iters[j] = it + 1
                if not host[a, 2] < self.tolerance:
                    still.append(j)
            active = still
            if not active:
                break
        return [(best[j][0], best[j][1], best[j][2], n_iters[j]) for j in range(R)]
