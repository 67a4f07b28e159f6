"""An answer that is DERIVED, not computed: Jaccard intersection counts of a ring of cells, used to check the oracle and every kernel
family of the HIP path without the oracle in the loop."""
import numpy as np


def cyclic_window_matrix(N, k):
    """Row i names the NEXT k cells on a ring of N >= 2k cells: ids (i + 1 + t) mod N + 1, t = 0 .. k-1.  Its intersection counts have a
    closed form that needs no oracle: the rows of cells i and j = i + d (1 <= d <= k) are the windows [i+1, i+k] and [j+1, j+k], which
    share exactly k - d ids; slot t names j = i + 1 + t, so u(i, t) = k - 1 - t — every cell the same, down to the zero row at t = k - 1."""
    i = np.arange(N, dtype=np.int64)[:, None]
    t = np.arange(k, dtype=np.int64)[None, :]
    return ((i + 1 + t) % N + 1).astype(np.int32)


def cyclic_window_expected(N, k):
    u = np.tile(np.arange(k - 1, -1, -1, dtype=np.int32), N)
    mat = cyclic_window_matrix(N, k)
    pos = u > 0
    src = np.where(pos, np.repeat(np.arange(1, N + 1), k), 0).astype(np.float64)
    dst = np.where(pos, mat.reshape(-1), 0).astype(np.float64)
    w = np.where(pos, u / (2.0 * k - u), 0.0)                      # reference src/rcpp_parallel_jaccard_coeff.cpp:51
    return np.stack([src, dst, w], axis=1), u


def circulant_counts(G, N, s, n_rare):
    """A genes x cells count matrix whose GF-ICF has a closed form.  Cell c holds the s genes (c + t) mod G, t = 0 .. s-1, with counts t + 1
    (N a multiple of G: every such gene lies in exactly N s / G cells), plus — in the first n_rare cells — one RARE gene G + c with count 7
    that no other cell has (nt = 1: dropped by any filter with min >= 1 / N).  With the rare genes dropped, every cell has
    S_c = s (s + 1) / 2 and every kept gene the same weight w = ln((N + 1) / (N s / G + 1)), so S and w cancel in the L2 step:
        gficf[(c + t) mod G, c] = (t + 1) / sqrt(s (s + 1) (2 s + 1) / 6)            (R/gficf.R:59, 88-89, 79, 100-103)
    Returns (scipy CSC matrix, expected dense-equivalent as (rowidx, colptr, x) of the compacted kept matrix, kept mask, nt, w)."""
    import scipy.sparse as sp

    assert N % G == 0 and 0 < s < G and n_rare <= N
    c = np.repeat(np.arange(N, dtype=np.int64), s)
    t = np.tile(np.arange(s, dtype=np.int64), N)
    rows = (c + t) % G
    vals = (t + 1).astype(np.float64)
    rc, rr = np.arange(n_rare, dtype=np.int64), G + np.arange(n_rare, dtype=np.int64)
    M = sp.csc_matrix((np.concatenate([vals, np.full(n_rare, 7.0)]), (np.concatenate([rows, rr]), np.concatenate([c, rc]))), shape=(G + n_rare, N))
    M.sort_indices()
    keep = np.concatenate([np.ones(G, dtype=bool), np.zeros(n_rare, dtype=bool)])
    nt = np.concatenate([np.full(G, N * s // G, dtype=np.int64), np.ones(n_rare, dtype=np.int64)])
    w = np.log((N + 1.0) / (N * s / G + 1.0))
    norm = np.sqrt(s * (s + 1) * (2 * s + 1) / 6.0)
    want = sp.csc_matrix((vals / norm, (rows, c)), shape=(G, N))
    want.sort_indices()
    return M, want, keep, nt, w


def cyclic_window_matrix_twice(N, k):
    """The same ring with every id named TWICE in a row (k even): slots 2 m and 2 m + 1 of row i both hold cell i + 1 + m.  Rows are multisets
    now: the rows of i and j = i + d share k/2 - d distinct ids, each with multiplicity two on both sides, so std::set_intersection
    (src/rcpp_parallel_jaccard_coeff.cpp:38-46) counts 2 (k/2 - d) and Rcpp::intersect of the serial entry (src/jaccard_coeff.cpp:33)
    k/2 - d; slot t names j = i + 1 + t // 2, i.e. d = 1 + t // 2."""
    assert k % 2 == 0 and N >= k
    i = np.arange(N, dtype=np.int64)[:, None]
    t = np.arange(k, dtype=np.int64)[None, :]
    return ((i + 1 + t // 2) % N + 1).astype(np.int32)


def cyclic_window_twice_expected(N, k, set_semantics=False):
    m = k // 2 - 1 - np.arange(k) // 2                                          # distinct ids shared with the neighbour of slot t
    u = np.tile((m if set_semantics else 2 * m).astype(np.int32), N)
    mat = cyclic_window_matrix_twice(N, k)
    pos = u > 0
    src = np.where(pos, np.repeat(np.arange(1, N + 1), k), 0).astype(np.float64)
    dst = np.where(pos, mat.reshape(-1), 0).astype(np.float64)
    w = np.where(pos, u / (2.0 * k - u), 0.0)
    return np.stack([src, dst, w], axis=1), u


def ring_of_cliques(c, m, weight=1.0):
    """c cliques of m vertices each (every pair inside joined by an edge of `weight`), clique q joined to clique q + 1 (mod c) by ONE edge, between
    the last vertex of q and the first of q + 1.  The answer of modularity optimisation at resolution g is derived, not computed:
    with K = m (m - 1) + 2 the total weight at a clique's vertices (in units of `weight`) and 2W = c K, merging two neighbouring cliques changes
    Q by 2 / (2W) - 2 g K^2 / (2W)^2, negative as long as c < g K; splitting a clique only loses.  So for c < g K the optimum is one community per
    clique and        Q = m (m - 1) / K - g / c        (reference quality function, src/ModularityOptimizer.cpp:461-482).
    Returns (symmetric scipy CSC adjacency, clique id of every vertex)."""
    import scipy.sparse as sp

    N = c * m
    q = np.repeat(np.arange(c, dtype=np.int64), m * m)
    a = np.tile(np.repeat(np.arange(m, dtype=np.int64), m), c)
    b = np.tile(np.tile(np.arange(m, dtype=np.int64), m), c)
    keep = a != b
    i, j = (q * m + a)[keep], (q * m + b)[keep]
    last, first = np.arange(c, dtype=np.int64) * m + (m - 1), ((np.arange(c, dtype=np.int64) + 1) % c) * m
    i, j = np.concatenate([i, last, first]), np.concatenate([j, first, last])
    A = sp.csc_matrix((np.full(len(i), float(weight)), (i, j)), shape=(N, N))
    A.sort_indices()
    return A, np.repeat(np.arange(c, dtype=np.int32), m)


def ring_of_cliques_modularity(c, m, resolution):
    return m * (m - 1) / (m * (m - 1) + 2.0) - resolution / c
