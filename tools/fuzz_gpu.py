#!/usr/bin/env python3
"""Randomised parity run of the HIP path against the oracle, longer than the test suite allows (minutes, not seconds).

Jaccard: random shapes (N 1 .. 6000, now and then 131 000 .. 180 000 for the wide-row table; k 1 .. 256, now and then 257 .. 700: the
sorted-row path; small shapes with k <= 16 take the one-launch form through the host entries and the distinct-ids mode), five kinds of index
matrix (windowed scrambled / windowed in order / uniform / few distinct ids = rows full of duplicates and self-references /
a window of 2), int32 or float64 input, through every entry that returns edges: the `.Call` entry (reference matrix), the
compact counts + host expansion, the filtered call-site form, the serial `jaccard_coeff` entry, the device-resident path
with counts, the single-process multi-device steps (peer copies; halo with nothing exchanged), the distinct-ids mode (the deferred GFICF_ERR_DUPLICATE_IDS must come exactly when a row repeats an id), and the
strict truncation mode on matrices with non-integer doubles.  Bit-exact or the run stops.
GF-ICF: random CSC matrices (G 1 .. 30 000, N 1 .. 3000; densities; empty cells, empty genes, explicit zeros, cells beyond
2048 entries), random filter bounds, supplied weights, icf types and norms; structure exact, values within 1e-12 (gficf() as the
reference runs it) / 1e-11 relative x the cell's condition number (the prob / smooth / l1 helper branches: an l1 sum over
weights of both signs can nearly cancel).
Every fourth case is one of the rows either side of the path (§8f): exact kNN (four metrics, plain and pruned form) against
the f32 oracle bit for bit, the adjacency matrix against scipy, cluster signatures and the transpose against restatements.
Usage: python tools/fuzz_gpu.py [seconds] [seed] [out.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import gficf_amd
import oracle
from gficf_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out_path = sys.argv[3] if len(sys.argv) > 3 else None
rng = np.random.default_rng(seed0)
NT = os.cpu_count() or 8
counts = {}


def bump(name):
    counts[name] = counts.get(name, 0) + 1


def knn_matrix(case):
    big = rng.random() < 0.04
    k = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 29, 30, 31, 32, 33, 50, 60, 61, 64, 65, 100, 120, 128, 129, 200, 255, 256])) if rng.random() < 0.6 \
        else int(rng.integers(1, 257))
    wide = (not big) and rng.random() < 0.06                         # round 5: k > 256, the sorted-row path (csrc/jaccard_sorted.h)
    if big:
        k = min(k, 64)
        N = int(rng.integers(131_000, 180_000))
    elif wide:
        k = int(rng.choice([257, 258, 300, 319, 320, 321, 400, 513, 700]))
        N = int(rng.integers(max(k // 3, 3), 1400))
    else:
        N = int(rng.integers(max(k + 2, 3), 6000)) if rng.random() < 0.9 else int(rng.integers(1, 40))
    kind = int(rng.integers(0, 5))
    if N <= 2 * k + 2 and kind in (0, 1):
        kind = 3
    if kind == 0:
        m = synth.knn_windowed(N, k, W=max(100, k), seed=case, perm_seed=case + 1)
    elif kind == 1:
        m = synth.knn_windowed(N, k, W=max(100, k), seed=case, perm_seed=None)
    elif kind == 2 and N - 1 >= k:
        m = synth.knn_uniform(N, k, seed=case)
    elif kind == 4 and N > 2 * k + 2:
        m = synth.knn_windowed(N, k, W=max((k + 1) // 2, 1), seed=case, perm_seed=case + 1)     # the tightest window: every row almost the same set
    else:                                                           # few distinct ids: duplicates inside rows, self-references
        distinct = int(rng.integers(1, max(1, min(N, 3 * k)) + 1))     # (ids must stay within [1, N])
        m = (synth.rand_u64(case, np.arange(N * k)).reshape(N, k) % np.uint64(distinct)).astype(np.int32) + 1
        kind = 3
    if rng.random() < 0.15 and N > 4:                               # a few duplicate ids planted into otherwise clean rows
        for _ in range(int(rng.integers(1, 6))):
            r = int(rng.integers(0, N))
            a, b = int(rng.integers(0, k)), int(rng.integers(0, k))
            m[r, a] = m[r, b]
    return m, N, k, kind, big


def jaccard_case(case):
    m, N, k, kind, big = knn_matrix(case)
    want, wu = oracle.jaccard(m, nthreads=NT)
    as_f64 = rng.random() < 0.3
    mat = m.astype(np.float64) if as_f64 else m
    entry = int(rng.integers(0, 9)) if not big else int(rng.choice([0, 1, 2, 6, 7, 8]))
    if entry == 8 and k > 64:
        entry = 7
    if entry == 5 and k > 256:                                      # (the truncating kernel for non-integer ids stops at k = 256)
        entry = 0
    tag = f"jaccard case {case}: N={N} k={k} kind={kind} f64={as_f64} entry={entry}"
    if entry == 0:
        got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
        assert np.array_equal(got, want), tag
        bump("jaccard .Call entry")
    elif entry == 1:
        u = gficf_amd.jaccard_counts(mat)
        assert np.array_equal(u.astype(np.int32).reshape(-1), wu), tag
        assert np.array_equal(gficf_amd.jaccard_expand(mat, u), want), tag
        bump("jaccard counts + expand")
    elif entry == 2:
        neigh = np.concatenate([np.arange(1, N + 1, dtype=mat.dtype)[:, None], mat], axis=1)
        e = gficf_amd.jaccard_edges(neigh, False)
        keep = want[:, 2] > 0
        assert np.array_equal(e["from"], want[keep, 0]) and np.array_equal(e["to"], want[keep, 1]) and np.array_equal(e["weight"], want[keep, 2]), tag
        bump("jaccard filtered call-site")
    elif entry == 3:
        got = gficf_amd.jaccard_coeff(mat, False)
        assert np.array_equal(got, oracle.jaccard_coeff(m)), tag
        bump("jaccard_coeff (serial entry)")
    elif entry == 4:
        import torch

        ops = gficf_amd.HipOps(0)
        idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
        table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
        rmat = torch.full((3, N * k), -7.0, dtype=torch.float64, device="cuda")
        u = torch.full((N * k,), -7, dtype=torch.int32, device="cuda")
        ops.jaccard(idx, N, k, table, rmat, u)
        ops.sync()
        assert np.array_equal(rmat.cpu().numpy().T, want) and np.array_equal(u.cpu().numpy(), wu), tag
        bump("jaccard device-resident with counts")
    elif entry == 6:
        # rows taken to hold distinct ids (gficf_ctx_set_jaccard_distinct): a matrix with a repeated id in any row must raise
        # the deferred error, a clean one must give the oracle's edges
        import torch

        ops = gficf_amd.HipOps(0)
        srt = np.sort(m, axis=1)
        has_dup = bool((srt[:, 1:] == srt[:, :-1]).any()) if k > 1 else False
        idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
        table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
        rmat = torch.full((3, N * k), -7.0, dtype=torch.float64, device="cuda")
        ops.set_jaccard_distinct(True)
        try:
            ops.jaccard(idx, N, k, table, rmat, None)
            try:
                ops.sync()
                raised = False
            except gficf_amd.GficfError as ex:
                # ABI 7: a row whose ids overflow its hash set beyond the checked list is reported as what it is (SET_OVERFLOW), not as a repeat
                assert ex.status in ("GFICF_ERR_DUPLICATE_IDS", "GFICF_ERR_SET_OVERFLOW"), tag + " " + ex.status
                raised = ex.status
        finally:
            ops.set_jaccard_distinct(False)
        if k > 256:
            assert not raised, tag + " (the sorted-row path is exact for every row: nothing to report)"
        elif raised == "GFICF_ERR_SET_OVERFLOW":
            assert k > 56, tag + " (only the general kernel's sets overflow)"
            ops.jaccard(idx, N, k, table, rmat, None)          # what a caller does next: the same call with the option off
            ops.sync()
            assert np.array_equal(rmat.cpu().numpy().T, want), tag + " (re-run after SET_OVERFLOW)"
        else:
            assert bool(raised) == has_dup or (raised and k >= 8), tag + f" raised={raised} has_dup={has_dup}"   # (with a second row overflowing, DUPLICATE_IDS may come from a row without a repeat only through the overflow list: k >= 8)
        if not raised:
            assert np.array_equal(rmat.cpu().numpy().T, want), tag
        bump("jaccard distinct-ids mode (error iff a row repeats an id)")
    elif entry in (7, 8):
        # the single-process multi-device steps on P contexts of the one GPU: peer copies of table slices (any ids), and the halo step
        # with nothing exchanged (int32, k <= 64: CAPACITY exactly when a block names more rows of one owner than there are slots)
        import torch

        from gficf_amd.api import MultiContext

        P = int(rng.integers(1, 5))
        mc = MultiContext([0] * P)
        try:
            bd = mc.cell_blocks(N)
            if entry == 7:
                rw = gficf_amd.HipOps.row_words(N, k)
                blocks = [torch.from_numpy(np.ascontiguousarray(mat[bd[r]:bd[r + 1]].T)).cuda() for r in range(P)]
                tables = [torch.zeros((N, rw), dtype=torch.int32, device="cuda") for _ in range(P)]
                outs = [torch.full((3, (bd[r + 1] - bd[r]) * k), -7.0, dtype=torch.float64, device="cuda") for r in range(P)]
                torch.cuda.synchronize()
                for _ in range(2):
                    mc.jaccard_device(blocks, N, k, tables, outs)
                mc.sync()
                assert np.array_equal(torch.cat(outs, dim=1).cpu().numpy().T, want), tag + f" P={P}"
                bump("multi-device step, peer copies")
            else:
                cap = int(rng.choice([1, 3, 64, 500])) if rng.random() < 0.5 else None
                bufs = mc.halo_buffers(N, k, cap)
                cap = bufs["cap"]
                rpr = -(-N // P)
                over = False
                for r in range(P):
                    ids = np.unique(m[bd[r]:bd[r + 1]])
                    ids = ids[(ids >= 1) & (ids <= N)]
                    own = (ids - 1) // rpr
                    for o in range(P):
                        if o != r and int((own == o).sum()) > cap:
                            over = True
                blocks = [torch.from_numpy(np.ascontiguousarray(m[bd[r]:bd[r + 1]].T)).cuda() for r in range(P)]
                torch.cuda.synchronize()
                raised = False
                try:
                    for _ in range(2):
                        mc.jaccard_halo_device(blocks, N, k, bufs)
                    mc.sync()
                except gficf_amd.GficfError as ex:
                    assert ex.status == "GFICF_ERR_CAPACITY", tag + " " + ex.status
                    raised = True
                assert raised == over, tag + f" P={P} cap={cap} raised={raised} over={over}"
                if not raised:
                    assert np.array_equal(torch.cat(bufs["out"], dim=1).cpu().numpy().T, want), tag + f" P={P} cap={cap}"
                bump("multi-device halo step, nothing exchanged" + (" (capacity error)" if raised else ""))
        finally:
            mc.close()
    else:
        md = m.astype(np.float64)
        frac = synth.rand_unit(case + 9, np.arange(N * k)).reshape(N, k)
        sel = synth.rand_unit(case + 10, np.arange(N * k)).reshape(N, k) < 0.2
        md = np.where(sel, np.minimum(md + 0.9 * frac, N + 0.999), md)
        wt, _ = oracle.jaccard(md, nthreads=NT)
        got = gficf_amd.rcpp_parallel_jaccard_coef(md, False, truncate_noninteger_ids=True)
        assert np.array_equal(got, wt), tag
        bump("jaccard strict truncation mode")


def gficf_case(case):
    G = int(rng.choice([1, 2, 50, 600, 5000, 20000, 30000])) if rng.random() < 0.5 else int(rng.integers(1, 30001))
    N = int(rng.integers(1, 3001)) if rng.random() < 0.9 else int(rng.integers(1, 12))
    mf = float(rng.choice([0.002, 0.02, 0.07, 0.3]))
    if G * mf * N > 6e6:
        mf = 6e6 / (G * N)
    sg = float(rng.choice([0.3, 0.5, 1.2]))
    cp, ri, x = synth.counts_csc(G, N, median_frac=mf, sigma=sg, seed=case)
    x = x.copy()
    zeros = rng.random() < 0.25
    if zeros and len(x):                                             # explicit zeros: the exact sequence must take over
        z = rng.integers(0, len(x), size=max(1, len(x) // 50))
        x[z] = 0.0
    if rng.random() < 0.3 and N > 3:                                 # empty cells
        M = sp.csc_matrix((x, ri, cp), shape=(G, N)).tolil()
        for c in rng.integers(0, N, size=max(1, N // 20)):
            M[:, int(c)] = 0
        M = M.tocsc()
        if not zeros:
            M.eliminate_zeros()
        M.sort_indices()
        cp, ri, x = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
    mn = float(rng.choice([0.0, 0.01, 0.05, 0.2]))
    mx = float(rng.choice([1.0, 1.0, 0.9, 0.5]))
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    tag = f"gficf case {case}: G={G} N={N} nnz={len(x)} zeros={zeros} min={mn} max={mx} median_frac={mf} sigma={sg}"
    mode = int(rng.integers(0, 3))
    if mode == 0:
        res = gficf_amd.gficf(M, cell_proportion_max=mx, cell_proportion_min=mn, normalize=False, verbose=False)
        ref = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
        keep = ref["keep"]
        g = res["gficf"]
        assert np.array_equal(res["genes"], np.flatnonzero(keep)) and np.array_equal(res["nt"], ref["nt"][keep]), tag
        assert np.array_equal(g.indptr, ref["colptr"]) and np.array_equal(g.indices, ref["rowidx"]), tag
        assert np.abs(g.data - ref["x"]).max(initial=0.0) < 1e-12 and np.allclose(res["w"], ref["w"][keep], rtol=1e-12, atol=1e-12), tag
        raw, want = res["rawCounts"], M[res["genes"], :]                  # $rawCounts = M[keep, ] (R/gficf.R:40,22), through the finish call
        assert np.array_equal(raw.indptr, want.indptr) and np.array_equal(raw.indices, want.indices) and np.array_equal(raw.data, want.data), tag
        bump("gficf()")
    elif mode == 1:
        w_in = 0.25 + synth.rand_unit(case + 3, np.arange(G))
        out, genes = gficf_amd.gficf_with_weights(M, w_in)
        ref = oracle.gficf_csc(G, N, cp, ri, x, 0.0, 2.0, w_in=w_in)
        assert np.array_equal(genes, np.flatnonzero(ref["keep"])), tag
        assert np.array_equal(out.indices, ref["rowidx"]) and np.array_equal(out.indptr, ref["colptr"]), tag
        assert np.abs(out.data - ref["x"]).max(initial=0.0) < 1e-12, tag
        bump("gficf with supplied weights")
    else:
        icf = str(rng.choice(["classic", "prob", "smooth"]))
        norm = str(rng.choice(["l2", "l1"]))
        res = gficf_amd.gficf(M, cell_proportion_max=mx, cell_proportion_min=mn, normalize=False, verbose=False, icf_type=icf, norm=norm)
        ref = oracle.gficf_csc(G, N, cp, ri, x, mn, mx, None, icf, norm)
        g = res["gficf"]
        assert np.array_equal(g.indptr, ref["colptr"]) and np.array_equal(g.indices, ref["rowidx"]), tag
        fin = np.isfinite(ref["x"])
        tag += f" icf={icf} norm={norm}"
        if not np.array_equal(np.isfinite(g.data), fin):
            bad = np.flatnonzero(np.isfinite(g.data) != fin)
            raise AssertionError(tag + f": finiteness differs at {bad[:5]}: got {g.data[bad[:5]]} want {ref['x'][bad[:5]]}")
        # l1 norm with prob weights (negative ones among them): the cell's sum can nearly cancel, and the result is then as
        # ill-conditioned for the reference as for anyone — the bound grows with the cell's condition number sum|v| / |sum v|
        cond = np.ones(len(ref["x"]))
        if norm == "l1":
            cellof = np.repeat(np.arange(N), np.diff(ref["colptr"]))
            v = ref["x"]                                            # v / sum(v): recover sum|.| / |sum .| of the unnormalised values
            sabs = np.bincount(cellof, weights=np.abs(np.where(np.isfinite(v), v, 0.0)), minlength=N)
            ssum = np.abs(np.bincount(cellof, weights=np.where(np.isfinite(v), v, 0.0), minlength=N))
            cond = np.maximum(1.0, (sabs / np.maximum(ssum, 1e-300)))[cellof]
        d = np.abs(g.data[fin] - ref["x"][fin]) / (np.maximum(1.0, np.abs(ref["x"][fin])) * cond[fin])
        if d.max(initial=0.0) >= 1e-11:
            i = int(np.argmax(d))
            raise AssertionError(tag + f": max scaled diff {d.max()} at {i}: got {g.data[fin][i]} want {ref['x'][fin][i]} cond {cond[fin][i]}")
        bump(f"gficf icf/norm variants")


def next_rows_case(case):
    """The rows either side of the path: exact kNN against the f32 oracle, adjacency against scipy,
    cluster signatures and the transpose against numpy / scipy restatements."""
    which = int(rng.integers(0, 5))
    if which == 4:
        # the chain of clustcells() (R/clustCells.R:57-86) two ways: step by step through the host entries (search, Jaccard + filter, adjacency,
        # Louvain) and fused on the device in one call (gficf_phenograph_host) — the same labels, cluster count, edge count and modularity; the
        # modularity the optimiser reports equal to the graph's modularity of its own labels, computed here in float64
        N = int(rng.integers(40, 4000)); d = int(rng.choice([2, 3, 10, 50])); k = int(min(N - 2, rng.choice([5, 15, 30, 33, 50])))
        metric = str(rng.choice(["manhattan", "euclidean", "cosine"]))
        res, algo, seed = float(rng.choice([0.5, 0.8, 1.0, 1.5])), int(rng.choice([1, 2])), int(rng.integers(0, 2**31 - 1))
        r2 = np.random.default_rng(case)
        C = int(rng.integers(1, 12))
        c = r2.normal(scale=5.0, size=(C, d)); lab = r2.integers(0, C, size=N)
        X = c[lab] + r2.normal(size=(N, d))
        tag = f"phenograph case {case}: N={N} d={d} k={k} {metric} resolution={res} algorithm={algo} seed={seed}"
        edges = gficf_amd.clustcells_graph(X, k, metric)
        A = gficf_amd.jaccard_adjacency(edges, N)
        step = gficf_amd.run_modularity_clustering(A, 1, res, algo, 3, 10, seed, False)
        fused = gficf_amd.phenograph(X, k, metric, res, algo, 3, 10, seed)
        assert np.array_equal(np.asarray(step), np.asarray(fused)), tag
        assert step.n_clusters == fused.n_clusters and fused.n_edges == len(edges["weight"]) and abs(step.modularity - fused.modularity) < 1e-12, tag
        co = A.tocoo()
        off = co.row != co.col                                           # (the optimiser ignores the diagonal)
        row, col, val = co.row[off], co.col[off], co.data[off]
        m2 = val.sum()                                                   # = 2 m
        deg = np.bincount(row, weights=val, minlength=N)
        labs = np.asarray(fused)
        inside = val[labs[row] == labs[col]].sum()
        q = inside / m2 - res * float((np.bincount(labs, weights=deg) ** 2).sum()) / (m2 * m2) if m2 > 0 else 0.0
        assert abs(q - fused.modularity) < 1e-6, (tag, q, fused.modularity)
        bump("phenograph fused against the chain step by step (N1, N2, N4)")
        return
    if which == 0:
        from oracle import oracle_np  # noqa: F401  (restatements live next to the oracle)

        N = int(rng.integers(2, 3000)); d = int(rng.choice([1, 2, 3, 7, 20, 50, 64, 100, 128])); k = int(min(N, rng.choice([1, 2, 5, 16, 31, 33, 51, 64, 100, 128])))
        metric = str(rng.choice(["manhattan", "euclidean", "cosine", "correlation"]))
        if metric == "correlation" and d < 2:
            metric = "euclidean"
        r2 = np.random.default_rng(case)
        c = r2.normal(scale=6.0, size=(8, d)); lab = r2.integers(0, 8, size=N)
        X = c[lab] + r2.normal(size=(N, d)) * r2.uniform(0.5, 2.0, size=(8, 1))[lab]
        if rng.random() < 0.3:
            os.environ["GFICF_KNN_PRUNE"] = "1"                     # the pruned form on small inputs too
        else:
            os.environ.pop("GFICF_KNN_PRUNE", None)
        tag = f"knn case {case}: N={N} d={d} k={k} {metric} prune={'GFICF_KNN_PRUNE' in os.environ}"
        got = gficf_amd.find_nn(X, k, True, metric)
        widx, wdist = oracle.knn(X, k, metric, nthreads=NT)
        os.environ.pop("GFICF_KNN_PRUNE", None)
        assert np.array_equal(got["idx"], widx), tag
        assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32)), tag
        bump("kNN (N2)")
    elif which == 1:
        N = int(rng.integers(1, 5000)); E = int(rng.integers(0, 40000))
        r2 = np.random.default_rng(case)
        f = r2.integers(1, N + 1, size=E).astype(np.float64); t = r2.integers(1, N + 1, size=E).astype(np.float64)
        w = r2.integers(1, 60, size=E) / 64.0                         # exactly representable: sums are exact in any order
        A = gficf_amd.jaccard_adjacency({"from": f, "to": t, "weight": w}, N)
        i, j = f.astype(np.int64) - 1, t.astype(np.int64) - 1
        W = sp.coo_matrix((w, (i, j)), shape=(N, N)).tocsc()          # igraph's undirected graph: A = W + W^T, a self edge once
        R = (W + W.T - sp.diags(W.diagonal())).tocsc()
        R.sum_duplicates(); R.sort_indices()
        assert A.shape == R.shape and np.array_equal(A.indptr, R.indptr) and np.array_equal(A.indices, R.indices) and np.array_equal(A.data, R.data), \
            f"adjacency case {case}: N={N} E={E}"
        bump("adjacency (N1)")
    else:
        G = int(rng.integers(1, 20000)); N = int(rng.integers(1, 2500))
        mf = float(rng.choice([0.005, 0.02, 0.07]))
        if G * mf * N > 4e6:
            mf = 4e6 / (G * N)
        cp, ri, x = synth.counts_csc(G, N, median_frac=mf, seed=case)
        if which == 2:
            C = int(rng.integers(1, min(N, 300) + 1))
            M = sp.csc_matrix((x * 0.25, ri, cp), shape=(G, N))
            lab = (synth.rand_u64(case, np.arange(N)) % np.uint64(C)).astype(np.int64)
            got, labels = gficf_amd.cluster_signatures(M, lab)
            onehot = sp.csr_matrix((np.ones(N), (np.arange(N), np.unique(lab, return_inverse=True)[1])), shape=(N, len(np.unique(lab))))
            want = np.asarray((M @ onehot).todense())
            col_of = {u: jj for jj, u in enumerate(np.unique(lab))}
            want = want[:, [col_of[u] for u in labels]]
            assert got.shape == want.shape and np.allclose(got, want, rtol=1e-9, atol=1e-9), f"signatures case {case}: G={G} N={N} C={C}"
            bump("cluster signatures (N3)")
        else:
            x = x.copy(); x[::17] = 0.0
            M = sp.csc_matrix((x, ri, cp), shape=(G, N))
            T = gficf_amd.transpose_gficf(M)
            S = M.T.tocsc(); S.sort_indices()
            assert T.shape == (N, G) and np.array_equal(T.indptr, S.indptr) and np.array_equal(T.indices, S.indices) and np.array_equal(T.data, S.data), \
                f"transpose case {case}: G={G} N={N}"
            bump("transpose (N3)")


t0 = time.time()
case = seed0 * 1_000_000
n = 0
last = t0
while time.time() - t0 < budget:
    if n % 4 == 2:
        gficf_case(case)
    elif n % 4 == 3:
        next_rows_case(case)
    else:
        jaccard_case(case)
    case += 1
    n += 1
    if time.time() - last > 60:
        print(f"... {n} cases, {time.time() - t0:.0f} s", flush=True)
        last = time.time()
lines = [f"tools/fuzz_gpu.py: {n} random cases in {time.time() - t0:.0f} s (seed {seed0}), every one equal to the oracle"]
lines += [f"  {v:6d}  {k}" for k, v in sorted(counts.items())]
print("\n".join(lines))
if out_path:
    open(out_path, "w").write("\n".join(lines) + "\n")
