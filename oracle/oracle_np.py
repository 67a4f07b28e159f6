"""Independent numpy/scipy restatement of the hot path — TEST INFRASTRUCTURE ONLY.

Second writing of the same reference lines as ``jaccard_oracle.cpp`` / ``gficf_oracle.cpp``,
deliberately through a different formulation (multiset keys + broadcasting for Jaccard;
scipy.sparse matrix algebra for GF-ICF) so that the two can cross-check each other.
PARITY UNPINNED for the same reason as the C++ oracle (no runnable reference here).

Citations relative to /root/reference.
"""
from __future__ import annotations

from collections import Counter

import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------- Jaccard
def jaccard_counts_py(mat: np.ndarray) -> np.ndarray:
    """Pure-Python multiset intersection counts (small cases only).

    u[i*k+j] = |sorted(row i) ∩ sorted(row mat[i,j])| with std::set_intersection
    semantics (src/rcpp_parallel_jaccard_coeff.cpp:38-46): every value counts
    min(multiplicity in row i, multiplicity in the neighbour row) times.
    """
    N, k = mat.shape
    rows = [Counter(int(v) for v in mat[i]) for i in range(N)]
    u = np.zeros(N * k, dtype=np.int32)
    for i in range(N):
        for j in range(k):
            kk = int(mat[i, j] - 1)                       # :28
            u[i * k + j] = sum((rows[i] & rows[kk]).values())
    return u


def jaccard_counts_np(mat: np.ndarray, block: int = 512) -> np.ndarray:
    """Vectorised multiset intersection counts.

    Each row element gets the key (value, occurrence-rank-within-row); keys are distinct
    within a row and |A ∩ B|_multiset == number of equal keys, so the count is a plain
    broadcast equality sum.
    """
    mat = np.asarray(mat)
    N, k = mat.shape
    ids = mat.astype(np.int64)
    order = np.argsort(ids, axis=1, kind="stable")
    srt = np.take_along_axis(ids, order, axis=1)
    same = np.zeros((N, k), dtype=np.int64)
    for t in range(1, k):
        same[:, t] = np.where(srt[:, t] == srt[:, t - 1], same[:, t - 1] + 1, 0)
    keys = srt * (k + 1) + same                      # N x k, sorted, distinct per row
    u = np.empty((N, k), dtype=np.int32)
    dst = ids - 1
    for b0 in range(0, N, block):
        b1 = min(N, b0 + block)
        A = keys[b0:b1]                               # B x k
        Bn = keys[dst[b0:b1]]                         # B x k(slot) x k
        eq = A[:, None, :, None] == Bn[:, :, None, :]  # B x slot x k x k
        u[b0:b1] = eq.sum(axis=(2, 3))
    return u.reshape(-1)


def jaccard_rmat(mat: np.ndarray, u: np.ndarray | None = None) -> np.ndarray:
    """(N*k) x 3 output of rcpp_parallel_jaccard_coef (:48-52, :67): rows with u == 0
    stay zero; others are (i+1, kk+1, u/(2.0*k-u))."""
    mat = np.asarray(mat)
    N, k = mat.shape
    if u is None:
        u = jaccard_counts_np(mat)
    E = N * k
    rm = np.zeros((E, 3), dtype=np.float64, order="F")
    src = np.repeat(np.arange(1, N + 1, dtype=np.float64), k)
    dst = np.trunc(mat.astype(np.float64) - 1.0).reshape(-1) + 1.0
    ud = u.astype(np.float64)
    pos = u > 0
    rm[pos, 0] = src[pos]
    rm[pos, 1] = dst[pos]
    rm[pos, 2] = ud[pos] / (2.0 * k - ud[pos])
    return rm


# ---------------------------------------------------------------------------- GF-ICF
def gficf_np(M: sp.csc_matrix, prop_min: float = 0.05, prop_max: float = 1.0, w_in=None, icf_type: str = "classic", norm: str = "l2"):
    """gficf(M, normalize=FALSE) on a scipy CSC genes x cells matrix (R/gficf.R:17-33).

    Returns dict(keep, nt, w (per ORIGINAL gene, 0 where dropped), gficf (CSC over kept
    genes)).  Written with sparse-matrix algebra, the way the R code is.
    """
    M = sp.csc_matrix(M, dtype=np.float64, copy=True)
    M.sort_indices()
    G, N = M.shape
    # R/gficf.R:40-41
    ix = np.asarray((M != 0).sum(axis=1)).ravel().astype(np.int64)
    keep = (ix.astype(np.float64) > N * prop_min) & (ix.astype(np.float64) <= N * prop_max)
    Mk = sp.csc_matrix(M[np.flatnonzero(keep), :])
    Mk.sort_indices()
    # R/gficf.R:59   t(t(M) / colSums(M))  — sequential storage-order sums
    cnt = np.diff(Mk.indptr)
    S = np.zeros(N, dtype=np.float64)
    for c in range(N):
        s = 0.0
        for v in Mk.data[Mk.indptr[c]:Mk.indptr[c + 1]]:
            s += v
        S[c] = s
    Srep = np.repeat(S, cnt)
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = np.where(Srep != 0.0, Mk.data / Srep, 0.0)   # defined-zero for S == 0 (see C++ header)
    TF = sp.csc_matrix((tf, Mk.indices, Mk.indptr), shape=Mk.shape)
    # R/gficf.R:88-89
    ntk = np.bincount(TF.indices[TF.data != 0], minlength=Mk.shape[0]).astype(np.int64)
    if w_in is None:
        with np.errstate(divide="ignore", invalid="ignore"):
            wk = {"classic": lambda: np.log((N + 1.0) / (ntk + 1.0)),          # R/gficf.R:89
                  "prob": lambda: np.log((N - ntk) / ntk),                      # R/gficf.R:90
                  "smooth": lambda: np.log(1.0 + N / ntk)}[icf_type]()          # R/gficf.R:91
    else:
        wk = np.asarray(w_in, dtype=np.float64)[keep]
    # R/gficf.R:79
    v = TF.data * wk[TF.indices]
    # R/gficf.R:100-103
    ss = np.zeros(N, dtype=np.float64)
    for c in range(N):
        s = 0.0
        for t in v[TF.indptr[c]:TF.indptr[c + 1]]:
            s += t if norm == "l1" else t * t
        ss[c] = s
    with np.errstate(divide="ignore"):
        nv = 1.0 / (ss if norm == "l1" else np.sqrt(ss))      # R/gficf.R:100
    nv[np.isinf(nv)] = 0.0
    out = np.repeat(nv, cnt) * v
    nt = np.zeros(G, dtype=np.int64)
    w = np.zeros(G, dtype=np.float64)
    nt[keep] = ntk
    w[keep] = wk
    return dict(keep=keep, nt=nt, w=w,
                gficf=sp.csc_matrix((out, TF.indices, TF.indptr), shape=Mk.shape))


# ---------------------------------------------------------------- cluster signatures
def cluster_signatures_np(gficf_mat: sp.csc_matrix, cluster):
    """R/clustCells.R:121-123: sapply(unique(cluster), function(x) rowSums(gficf[, cluster %in% x]))."""
    lab = np.asarray(cluster)
    _, first = np.unique(lab, return_index=True)
    labels = lab[np.sort(first)]                       # base::unique keeps first-appearance order
    M = sp.csc_matrix(gficf_mat)
    cols = [np.asarray(M[:, np.flatnonzero(lab == u)].sum(axis=1)).ravel() for u in labels]
    return np.stack(cols, axis=1) if cols else np.zeros((M.shape[0], 0)), labels


def transpose_np(G, N, colptr, rowidx, x):
    """t(data$gficf) (R/dimensinalityReduction.R:33, :100; Matrix::t): the CSC arrays of the cells x genes matrix.
    A stable sort of the stored entries by gene keeps the cells ascending within every gene; explicit zeros stay."""
    colptr = np.asarray(colptr, dtype=np.int64)
    rowidx = np.asarray(rowidx)
    cell = np.repeat(np.arange(N, dtype=np.int32), np.diff(colptr))
    order = np.argsort(rowidx, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(rowidx, minlength=G))]).astype(np.int64)
    return ptr, cell[order], np.asarray(x, dtype=np.float64)[order]


def modularity_np(A, labels, resolution: float = 1.0, function: int = 1) -> float:
    """VOSClusteringTechnique::calcQualityFunction (src/ModularityOptimizer.cpp:461-482) for modularityFunction = 1 on the
    network RunModularityClusteringCpp builds (diagonal dropped, src/RModularityOptimizer.cpp:73-75; node weight = the
    vertex's total edge weight, :185; resolution2 = resolution / 2W, :100):
        Q = ( sum_{ij, c_i = c_j} A_ij  -  resolution * sum_c K_c^2 / 2W ) / 2W."""
    A = sp.csr_matrix(A).astype(np.float64).copy()
    A.setdiag(0.0)
    A.eliminate_zeros()
    lab = np.asarray(labels)
    k = np.asarray(A.sum(axis=1)).ravel()
    two_w = k.sum()
    coo = A.tocoo()
    inside = coo.data[lab[coo.row] == lab[coo.col]].sum()
    if function == 2:                                   # alternative: node weight 1, resolution as given (:799-805, :100)
        n_c = np.bincount(lab).astype(np.float64)
        return float((inside - resolution * (n_c * n_c).sum()) / two_w)
    K = np.bincount(lab, weights=k)
    return float((inside - resolution * (K * K).sum() / two_w) / two_w)


def jaccard_coeff_np(mat: np.ndarray) -> np.ndarray:
    """The serial entry (src/jaccard_coeff.cpp:19-44) through numpy set algebra: u = |unique(row i) ∩ unique(row kk)|
    (``Rcpp::intersect``, :33), rows with u > 0 packed from the top (:34-39)."""
    N, k = mat.shape
    out = np.zeros((N * k, 3))
    r = 0
    for i in range(N):
        for j in range(k):
            kk = int(mat[i, j]) - 1
            u = len(np.intersect1d(mat[i], mat[kk]))
            if u > 0:
                out[r] = (i + 1, kk + 1, u / (2.0 * k - u))
                r += 1
    return out


# ------------------------------------------------------------------------------- kNN
def knn_np(X: np.ndarray, k: int, metric: str = "manhattan"):
    """Exact kNN in float64 (different arithmetic from ``knn_oracle.cpp``, which works in f32 like the
    HIP kernel): the k smallest (distance, index) pairs per row, the row itself included — the
    contract the approximate ``uwot:::find_nn(..., method="annoy", metric=...)`` call of
    R/clustCells.R:57,60 aims at.  Returns (idx N x k, 1-based; dist N x k)."""
    X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    idx = np.zeros((N, k), dtype=np.int32)
    dist = np.zeros((N, k), dtype=np.float64)
    if metric == "correlation":
        X = X - X.mean(axis=1, keepdims=True)
    if metric in ("cosine", "correlation"):
        nrm = np.sqrt((X * X).sum(axis=1, keepdims=True))
        Xn = np.divide(X, nrm, out=np.zeros_like(X), where=nrm > 0)
    for i in range(N):
        if metric == "manhattan":
            dv = np.abs(X - X[i]).sum(axis=1)
        elif metric == "euclidean":
            dv = np.sqrt(((X - X[i]) ** 2).sum(axis=1))
        elif metric in ("cosine", "correlation"):
            dv = 1.0 - Xn @ Xn[i]
        else:
            raise ValueError(metric)
        order = np.lexsort((np.arange(N), dv))[:k]
        idx[i] = order + 1
        dist[i] = dv[order]
    return idx, dist
