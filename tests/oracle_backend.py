"""TEST-ONLY compute backend for scd_amd.kmeans.KMeansEngine built on the numpy oracle, so that the multi-process
collective logic (all-reduce of centroid partials, shard-aware k-means++ draws, gathered constrained E-step) can be
exercised with gloo on CPU.  The product never uses this: its only backend is scd_amd.ops (HIP)."""
import numpy as np
import torch

from oracle import kmeans_oracle as ko
from oracle import transport_oracle as to


class _Data:
    def __init__(self, x):
        self.x = x.to(torch.float32).contiguous()
        self.n, self.d = self.x.shape


class OracleBackend:
    def prepare(self, x):
        return _Data(x)

    def estep(self, data, centers, expect_few=False):
        lab, _, _ = ko.estep(data.x.numpy(), centers.numpy())
        return torch.from_numpy(lab.astype(np.int32))

    def rowdist(self, data, centers, labels):
        d = ko.pairwise_distance64(data.x.numpy(), centers.numpy())
        return torch.from_numpy(d[np.arange(d.shape[0]), labels.numpy()].astype(np.float32))

    def min_update(self, data, c_new, d2):
        d = ko.pairwise_distance64(data.x.numpy(), c_new.numpy().reshape(1, -1))[:, 0].astype(np.float32)
        d2.copy_(torch.minimum(d2, torch.from_numpy(d)))

    def dist(self, data, centers, sqrt=False, with_cost=False):
        d2 = ko.pairwise_distance(data.x.numpy(), centers.numpy())
        out = np.sqrt(d2).astype(np.float32) if sqrt else d2
        if with_cost:
            return torch.from_numpy(out), torch.from_numpy(to.int_costs(d2))
        return torch.from_numpy(out)

    def mstep(self, x, labels32, c_old, k, split):
        xn, lab = x.numpy().astype(np.float64), labels32.numpy().astype(np.int64)
        sums = np.zeros((k, xn.shape[1]))
        np.add.at(sums, lab, xn)
        counts = np.bincount(lab, minlength=k).astype(np.int64)
        inertia = np.zeros(2)
        if c_old is not None:
            d = xn - c_old.numpy().astype(np.float64)[lab]
            row = (d * d).sum(1)
            inertia[:] = [row[:split].sum(), row[split:].sum()]
        return torch.from_numpy(sums), torch.from_numpy(counts), torch.from_numpy(inertia)

    def finalize(self, sums, counts, c_old, data=None):
        with np.errstate(invalid="ignore", divide="ignore"):
            c = (sums.numpy() / counts.numpy().astype(np.float64)[:, None]).astype(np.float32)
        shift = np.array([np.nan])
        if c_old is not None:
            shift = np.array([ko._center_shift_sq(c, c_old.numpy())])
        return torch.from_numpy(c), torch.from_numpy(shift)

    def sum_f32(self, x):
        return torch.tensor([float(np.sum(x.numpy().astype(np.float64)))], dtype=torch.float64)

    def kpp_draw(self, d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
        d = d2.numpy().astype(np.float32)
        tot = np.float32(np.sum(d.astype(np.float64))) if total is None else np.float32(float(total))
        prob = (d / tot).astype(np.float32).astype(np.float64)
        ps = torch.tensor([prob.sum()], dtype=torch.float64) if want_probsum else None
        idx = None
        if want_idx:
            cum = (np.cumsum(prob) + (0.0 if prefix is None else float(prefix))).astype(np.float32)
            hit = np.nonzero(cum >= np.float32(r))[0]
            idx = torch.tensor([int(hit[0]) if hit.size else -1], dtype=torch.int64)
        return idx, ps

    # lock-step variants (one row per restart): loops over the single-vector versions above
    def min_update_multi(self, data, c_new, d2):
        for j in range(c_new.shape[0]):
            self.min_update(data, c_new[j], d2[j])

    def sum_f32_multi(self, x):
        return torch.cat([self.sum_f32(x[j]) for j in range(x.shape[0])])

    def kpp_draw_multi(self, d2, r, total=None, prefix=None, want_idx=True, want_probsum=False):
        outs = [self.kpp_draw(d2[j], float(r[j]), None if total is None else total[j], None if prefix is None else prefix[j],
                              want_idx, want_probsum) for j in range(d2.shape[0])]
        idx = torch.cat([o[0] for o in outs]) if want_idx else None
        ps = torch.cat([o[1] for o in outs]) if want_probsum else None
        return idx, ps

    def transport(self, cost, size_min, size_max):
        from scd_amd import ops
        return ops.transport_solve(cost, size_min, size_max)      # host C++ solver (no device needed)


class OracleNamingOps:
    """TEST-ONLY stand-in for the naming ops pipeline.vote_loop_unsup_sharded calls (scd_amd.ops.vote_table,
    vote_table_topm, gather_rows_f16, sim_argmax; vote_hist for the single-rank form), on the numpy oracle - so that the row-sharded
    vote can run under gloo on CPU."""

    def vote_hist(self, name_idx, top_k, preds, clusters, m, known=None):
        from oracle import naming_oracle as no
        ref = no.cluster_counters(name_idx.numpy(), preds.numpy(), clusters, top_k, known=known)
        keys = np.full((len(clusters), m), -1, dtype=np.int64)
        counts = np.zeros((len(clusters), m), dtype=np.int32)
        for i, c in enumerate(clusters):
            for j, (a, b) in enumerate(ref[c].most_common(m)):
                keys[i, j], counts[i, j] = a, b
        return torch.from_numpy(keys), torch.from_numpy(counts)

    def vote_table(self, name_idx, top_k, preds, clusters, n_slots, row_offset, v):
        """numpy restatement of scd_vote_table: dense counts / first-seen tables of this rank's rows."""
        idx, pr = name_idx.numpy(), preds.numpy()
        slot = {c: i for i, c in enumerate(clusters)}
        counts = np.zeros((len(clusters), v), dtype=np.int32)
        first = np.full((len(clusters), v), 0x7F7F7F7F7F7F7F7F, dtype=np.int64)
        for i in range(idx.shape[0]):
            s = slot.get(int(pr[i]))
            if s is None:
                continue
            for j in range(top_k):
                nm = int(idx[i, j])
                counts[s, nm] += 1
                first[s, nm] = min(first[s, nm], (row_offset + i) * top_k + j)
        return torch.from_numpy(counts), torch.from_numpy(first)

    def vote_table_topm(self, counts, first, m):
        c, f = counts.numpy(), first.numpy()
        keys = np.full((c.shape[0], m), -1, dtype=np.int64)
        cnt = np.zeros((c.shape[0], m), dtype=np.int32)
        for s in range(c.shape[0]):
            nz = np.nonzero(c[s])[0]
            order = sorted(nz.tolist(), key=lambda nm: (-int(c[s, nm]), int(f[s, nm])))[:m]
            for j, nm in enumerate(order):
                keys[s, j], cnt[s, j] = nm, c[s, nm]
        return torch.from_numpy(keys), torch.from_numpy(cnt)

    def gather_rows_f16(self, wt, idx):
        return wt[idx]

    def sim_argmax(self, f, wsel_t, scale=100.0):
        from oracle import naming_oracle as no
        i, v = no.sim_argmax(f.numpy(), wsel_t.numpy().T, scale)
        return torch.from_numpy(i), torch.from_numpy(v)
