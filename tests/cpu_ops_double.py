"""CPU test double for the stage-kernel object (`gficf_amd.api.HipOps`) — tests only.

It lets the sharding / collective logic of gficf_amd.dist run on CPU tensors under the gloo
backend.  Arithmetic comes from the oracle and numpy; nothing here is part of the product.
"""
import numpy as np
import torch

import oracle


class CpuOpsDouble:
    @staticmethod
    def kpad(k):
        return 16 if k <= 16 else 32 if k <= 32 else 64 if k <= 64 else 128 if k <= 128 else 256

    @classmethod
    def row_words(cls, N_total, k):
        return cls.kpad(k)             # the double keeps plain int32 rows whatever the size

    def sync(self):
        if getattr(self, "_halo_overflow", False):
            self._halo_overflow = False
            from gficf_amd import GficfError

            raise GficfError(7, "halo request slots overflowed (test double)")

    # ---- the sharded build on local ids (mirrors csrc/halo.hip; numpy, tests only)
    @staticmethod
    def halo_workspace_bytes(N_total, P):
        return 8 * (N_total + 2) + 8 * (P + 2)          # the double keeps an int64 rank per id instead of a bitmap

    def halo_plan(self, idx_cm, n_local, k, N_total, cell_begin, P, rpr, cap, ws, req_out):
        ids = idx_cm.numpy().reshape(k, -1)[:, :n_local].reshape(-1).astype(np.int64)
        ids = ids[(ids >= 1) & (ids <= N_total)]
        need = np.unique(ids[(ids <= cell_begin) | (ids > cell_begin + n_local)])
        owner = (need - 1) // rpr
        req = req_out.numpy().reshape(P, cap)
        req[:] = 0
        local = np.zeros(N_total + 2, dtype=np.int64)    # global id -> local id of its halo slot (0: not requested)
        for r in range(P):
            mine = need[owner == r]
            if len(mine) > cap:
                self._halo_overflow = True
                mine = mine[:cap]
            req[r, :len(mine)] = mine
            local[mine] = n_local + r * cap + np.arange(len(mine)) + 1
        ws.numpy().view(np.int64)[:N_total + 2] = local

    def halo_serve(self, idx_cm, n_local, k, cell_begin, req_in, rows_out):
        rows = idx_cm.numpy().reshape(k, -1)[:, :n_local].T
        q = req_in.numpy().astype(np.int64)
        out = rows_out.numpy().reshape(-1, k)
        out[:] = 0
        sel = q != 0
        assert np.all((q[sel] > cell_begin) & (q[sel] <= cell_begin + n_local)), "asked for a row this rank does not own"
        out[sel] = rows[q[sel] - 1 - cell_begin]

    def halo_relabel(self, idx_cm, n_local, k, N_total, cell_begin, P, rpr, cap, ws, req_out, rows_in, idx_ext, l2g):
        local = ws.numpy().view(np.int64)[:N_total + 2].copy()
        own = np.arange(cell_begin + 1, cell_begin + n_local + 1)
        local[own] = np.arange(1, n_local + 1)
        rows = idx_cm.numpy().reshape(k, -1)[:, :n_local].T.astype(np.int64)
        assert rows.min() >= 1 and rows.max() <= N_total
        ext = idx_ext.numpy().reshape(k, -1)
        n_ext = n_local + P * cap
        ext[:, :n_local] = local[rows].T
        req = req_out.numpy().astype(np.int64)
        hal = rows_in.numpy().reshape(-1, k).astype(np.int64)
        hl = np.where((hal >= 1) & (hal <= N_total), local[np.clip(hal, 0, N_total + 1)], 0)
        hl[req == 0] = 0
        ext[:, n_local:n_ext] = hl.T
        g = l2g.numpy()
        g[:n_local] = own
        g[n_local:n_ext] = req

    def jaccard_ingest_local(self, idx_ext, n_ext, k, table):
        rows = idx_ext.numpy().reshape(k, -1)[:, :n_ext].T
        assert rows.min() >= 0 and rows.max() <= n_ext
        table[:n_ext, :k] = torch.from_numpy(np.ascontiguousarray(rows.astype(np.int32)))
        table[:n_ext, k:] = 0

    def jaccard_edges_mapped(self, table, n_ext, k, n_cells, src_offset, l2g, out3, u=None):
        tab = table.numpy()[:n_ext, :k].astype(np.int64)
        g = l2g.numpy().astype(np.int64)
        uu = np.zeros(n_cells * k, dtype=np.int64)
        for i in range(n_cells):
            a, ca = np.unique(tab[i][tab[i] != 0], return_counts=True)
            for j in range(k):
                nb = tab[i, j]
                if nb == 0:
                    continue
                b, cb = np.unique(tab[nb - 1][tab[nb - 1] != 0], return_counts=True)
                _, ia, ib = np.intersect1d(a, b, return_indices=True)
                uu[i * k + j] = int(np.minimum(ca[ia], cb[ib]).sum())          # multiset intersection (reference :41-46)
        src = np.repeat(np.arange(src_offset + 1, src_offset + n_cells + 1, dtype=np.float64), k)
        loc = tab[:n_cells].reshape(-1)
        dst = np.where(loc != 0, g[np.maximum(loc, 1) - 1], 0).astype(np.float64)
        w = uu / (2.0 * k - uu)
        posm = uu > 0
        out3.copy_(torch.from_numpy(np.stack([np.where(posm, src, 0.0), np.where(posm, dst, 0.0), np.where(posm, w, 0.0)])))
        if u is not None:
            u.copy_(torch.from_numpy(uu.astype(np.int32)))

    # ---- Jaccard
    def jaccard_ingest(self, idx_cm, n_rows, k, N_total, table_rows):
        rows = idx_cm.numpy().reshape(k, -1)[:, :n_rows].T          # n_rows x k
        assert rows.min() >= 1 and rows.max() <= N_total
        table_rows[:n_rows, :k] = torch.from_numpy(np.ascontiguousarray(rows.astype(np.int32)))
        table_rows[:n_rows, k:] = 0

    @staticmethod
    def _bits(N):
        b = 1
        while b < 31 and (1 << b) <= N:
            b += 1
        return b

    def packed_words(self, N_total, k):
        return (k * self._bits(N_total) + 1 + 31) // 32

    def jaccard_pack_rows(self, table_rows, n_rows, k, N_total, packed):
        bits, out = self._bits(N_total), packed.numpy().view(np.uint32)
        rows = table_rows.numpy().view(np.uint32)
        for r in range(n_rows):
            acc = 0
            for j in range(k):
                acc |= int(rows[r, j] & 0x7FFFFFFF) << (j * bits)
            acc |= int(rows[r, 0] >> 31) << (k * bits)
            for w in range(out.shape[1]):
                out[r, w] = (acc >> (32 * w)) & 0xFFFFFFFF

    def jaccard_unpack_rows(self, packed, n_rows, k, N_total, table_rows):
        bits, src = self._bits(N_total), packed.numpy().view(np.uint32)
        rows = table_rows.numpy().view(np.uint32)
        for r in range(n_rows):
            acc = 0
            for w in range(src.shape[1]):
                acc |= int(src[r, w]) << (32 * w)
            rows[r, :] = 0
            for j in range(k):
                rows[r, j] = (acc >> (j * bits)) & ((1 << bits) - 1)
            rows[r, 0] |= ((acc >> (k * bits)) & 1) << 31

    def jaccard_edges(self, table, N, k, cell_begin, cell_end, out3, u=None):
        # only the rows the block names are read (after a halo exchange the others are not filled): the oracle on the
        # sub-matrix of those rows, ids relabelled to its row numbers (a bijection: intersection counts are unchanged)
        tab = table.numpy()[:N, :k].astype(np.int64)
        rows = tab[cell_begin:cell_end]
        used = np.unique(np.concatenate([np.arange(cell_begin + 1, cell_end + 1), rows.reshape(-1)]))
        assert used.min() >= 1, "a named row was not fetched"
        sub = tab[used - 1]
        ids = np.unique(sub.reshape(-1))
        assert ids.min() >= 1, "a named row was not fetched"
        # ids that are not rows of the sub-matrix get fresh row numbers beyond it (they are never dereferenced by the rows we keep)
        label = {int(v): i + 1 for i, v in enumerate(used)}
        nxt = len(used) + 1
        for v in ids:
            if int(v) not in label:
                label[int(v)] = nxt
                nxt += 1
        lut = np.zeros(int(ids.max()) + 1, dtype=np.int64)
        for v, l in label.items():
            if v < len(lut):
                lut[v] = l
        rel = lut[sub]
        pad = np.tile(np.arange(1, k + 1, dtype=np.int64), (nxt - 1 - len(used), 1)) if nxt - 1 > len(used) else np.zeros((0, k), np.int64)
        full = np.concatenate([rel, pad]).astype(np.int32)
        rm, uu = oracle.jaccard(np.ascontiguousarray(full), nthreads=2)
        pos = np.searchsorted(used, np.arange(cell_begin + 1, cell_end + 1))
        sel = (pos[:, None] * k + np.arange(k)[None, :]).reshape(-1)
        uu = uu[sel]
        src = np.repeat(np.arange(cell_begin + 1, cell_end + 1, dtype=np.float64), k)
        dst = rows.reshape(-1).astype(np.float64)
        w = uu / (2.0 * k - uu)
        posm = uu > 0
        out = np.stack([np.where(posm, src, 0.0), np.where(posm, dst, 0.0), np.where(posm, w, 0.0)])
        out3.copy_(torch.from_numpy(out))
        if u is not None:
            u.copy_(torch.from_numpy(uu.astype(np.int32)))

    # ---- GF-ICF
    def csc_count(self, G, n_cells, colptr, rowidx, x, nt):
        ri, xv = rowidx.numpy(), x.numpy()
        nt += torch.from_numpy(np.bincount(ri[xv != 0], minlength=G).astype(np.int64))

    def csc_genes(self, G, N_total, nt, prop_min, prop_max, w_in, keep, genes, w, gkept):
        c = nt.numpy().astype(np.float64)
        kp = (c > N_total * prop_min) & (c <= N_total * prop_max)
        wv = np.where(kp, np.log((N_total + 1.0) / (c + 1.0)) if w_in is None else w_in.numpy(), 0.0)
        remap = np.where(kp, np.cumsum(kp) - 1, -1).astype(np.int32)
        keep.copy_(torch.from_numpy(kp.astype(np.uint8)))
        w.copy_(torch.from_numpy(wv))
        g = genes.numpy()[:2 * G].reshape(G, 2)
        g[:, 0] = wv
        g[:, 1] = remap.astype(np.float64)      # the double only needs to round-trip inside this double
        gkept[0] = int(kp.sum())

    def csc_colptr(self, G, n_cells, colptr, rowidx, keep, gkept, out_colptr):
        cp, ri, kp = colptr.numpy(), rowidx.numpy(), keep.numpy().astype(bool)
        cnt = np.add.reduceat(kp[ri].astype(np.int64), cp[:-1]) if len(ri) else np.zeros(n_cells, np.int64)
        cnt[np.diff(cp) == 0] = 0
        out_colptr.copy_(torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)))

    def csc_scale(self, G, n_cells, colptr, rowidx, x, genes, gkept, out_colptr, out_rowidx, out_x):
        cp, ri, xv = colptr.numpy(), rowidx.numpy(), x.numpy()
        g = genes.numpy()[:2 * G].reshape(G, 2)
        wv, remap = g[:, 0], g[:, 1].astype(np.int64)
        ocp = out_colptr.numpy()
        ori, ox = out_rowidx.numpy(), out_x.numpy()
        for c in range(n_cells):
            sl = slice(cp[c], cp[c + 1])
            kp = remap[ri[sl]] >= 0
            xs, gs = xv[sl][kp], ri[sl][kp]
            S = 0.0
            for v in xs:
                S += v
            v = (xs / S) * wv[gs] if S != 0 else np.zeros_like(xs)
            q = 0.0
            for t in v:
                q += t * t
            nv = 1.0 / np.sqrt(q) if q > 0 else 0.0
            ori[ocp[c]:ocp[c + 1]] = remap[gs]
            ox[ocp[c]:ocp[c + 1]] = nv * v

    # ---- exact kNN (N2)
    @staticmethod
    def knn_dpad(d):
        return (d + 3) & ~3

    def knn_workspace_bytes(self, n_queries, N, k):
        return 16

    def knn_prepare(self, X_cm, n_rows, d, metric, point_rows):
        X = X_cm.numpy().reshape(d, -1)[:, :n_rows].T.astype(np.float32)
        if metric == "correlation":
            X = X - X.mean(axis=1, keepdims=True)
        if metric in ("cosine", "correlation"):
            nrm = np.sqrt((X.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
            X = np.divide(X, nrm, out=np.zeros_like(X), where=nrm > 0)
        point_rows[:n_rows, :d] = torch.from_numpy(np.ascontiguousarray(X))
        point_rows[:n_rows, d:] = 0

    def knn_search(self, points, N, d, k, metric, q_begin, q_end, ws, idx_cm, dist_cm=None):
        # points are already prepared (normalised for cosine): a dot product / L1 / L2 brute force in float64
        P = points.numpy()[:N, :d].astype(np.float64)
        for q in range(q_begin, q_end):
            if metric == "manhattan":
                dv = np.abs(P - P[q]).sum(axis=1)
            elif metric == "euclidean":
                dv = np.sqrt(((P - P[q]) ** 2).sum(axis=1))
            else:
                dv = 1.0 - P @ P[q]
            order = np.lexsort((np.arange(N), dv))[:k]
            idx_cm[:, q - q_begin] = torch.from_numpy((order + 1).astype(np.int32))
            if dist_cm is not None:
                dist_cm[:, q - q_begin] = torch.from_numpy(dv[order].astype(np.float32))
