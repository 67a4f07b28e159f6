"""Portable synthetic inputs for the hot path.

Pure numpy, counter-based splitmix64 — no dependence on numpy's / libstdc++'s RNG
streams, so the same seeds give the same inputs everywhere (tests, bench, CPU baseline).

  * ``knn_windowed``  — N x k kNN index matrix (1-based ids, self excluded) with realistic
    neighbour overlap: cell i draws k distinct ids from the window [i-W, i+W] \\ {i}
    (mod N) in random order; all ids are then relabelled by a fixed random permutation so
    memory locality is destroyed the way real Annoy output does.  This is the shape
    ``uwot:::find_nn(...)$idx[, -1]`` hands to the Jaccard step
    (reference R/clustCells.R:57-63).
  * ``knn_uniform``   — k distinct uniformly random ids per cell (u ~ 0, cache-hostile).
  * ``counts_csc``    — dgCMatrix-like CSC genes x cells UMI count matrix (int32 row
    indices sorted within each column, float64 integer-valued counts).
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rand_u64(seed: int, a, b=0) -> np.ndarray:
    """Counter-based random uint64 for (seed, a, b)."""
    with np.errstate(over="ignore"):
        a = np.asarray(a, dtype=np.uint64)
        b = np.asarray(b, dtype=np.uint64)
        s = mix64(np.uint64(seed) * _GOLD + np.uint64(1))
        return mix64(mix64(s + (a + np.uint64(1)) * _GOLD) + (b + np.uint64(1)) * _M1)


def rand_unit(seed: int, a, b=0) -> np.ndarray:
    """Uniform doubles in [0, 1)."""
    return (rand_u64(seed, a, b) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def permutation(n: int, seed: int) -> np.ndarray:
    keys = rand_u64(seed, np.arange(n, dtype=np.uint64))
    return np.argsort(keys, kind="stable").astype(np.int64)


def knn_windowed(N: int, k: int, W: int = 100, seed: int = 42, perm_seed: int | None = 43,
                 dtype=np.int32, chunk: int = 65536, rows: tuple | None = None) -> np.ndarray:
    """N x k matrix (C order) of 1-based neighbour ids, self excluded, distinct per row.
    ``rows=(b, e)``: only rows [b, e) of that matrix (an (e-b) x k array, the same values) — what one rank of a sharded job
    holds; costs (e-b)/N of the full generation (plus the permutation)."""
    W = int(min(W, (N - 1) // 2))
    if 2 * W < k:
        raise ValueError(f"window 2*{W} smaller than k={k} (N={N})")
    offs = np.concatenate([np.arange(-W, 0), np.arange(1, W + 1)]).astype(np.int64)
    pi = permutation(N, perm_seed) if perm_seed is not None else np.arange(N, dtype=np.int64)
    if rows is None:
        b, e = 0, N
        src = None                                 # row pi[c] of the matrix belongs to (pre-permutation) cell c
    else:
        b, e = int(rows[0]), int(rows[1])
        if perm_seed is not None:
            inv = np.empty(N, dtype=np.int64)
            inv[pi] = np.arange(N, dtype=np.int64)
            src = inv[b:e]                         # the cells whose rows land in [b, e), in row order
        else:
            src = np.arange(b, e, dtype=np.int64)
    n_out = e - b
    out = np.empty((n_out, k), dtype=dtype)
    for c0 in range(0, N if src is None else n_out, chunk):
        c1 = min(N if src is None else n_out, c0 + chunk)
        n = c1 - c0
        cells = np.arange(c0, c1, dtype=np.int64) if src is None else src[c0:c1]
        cand = np.broadcast_to(offs, (n, 2 * W)).copy()
        rws = np.arange(n)
        for t in range(k):
            r = t + (rand_u64(seed, cells, t) % np.uint64(2 * W - t)).astype(np.int64)
            a = cand[rws, t].copy()
            cand[rws, t] = cand[rws, r]
            cand[rws, r] = a
        nb = (cells[:, None] + cand[:, :k]) % N
        if src is None:
            out[pi[c0:c1]] = (pi[nb] + 1).astype(dtype)
        else:
            out[c0:c1] = (pi[nb] + 1).astype(dtype)
    return out


def knn_uniform(N: int, k: int, seed: int = 44, dtype=np.int32) -> np.ndarray:
    """N x k matrix of k distinct uniformly random 1-based ids per row, self excluded."""
    if N - 1 < k:
        raise ValueError("need N-1 >= k")
    cells = np.arange(N, dtype=np.int64)
    if (N - 1) <= 4 * k and N * (N - 1) <= 64_000_000:
        # dense case: rank random keys of all N-1 offsets, keep the first k
        keys = rand_u64(seed, cells[:, None], np.arange(N - 1, dtype=np.int64)[None, :])
        off = 1 + np.argsort(keys, axis=1, kind="stable")[:, :k].astype(np.int64)
        return (((cells[:, None] + off) % N) + 1).astype(dtype)
    off = np.empty((N, k), dtype=np.int64)
    for t in range(k):
        off[:, t] = 1 + (rand_u64(seed, cells, t) % np.uint64(N - 1)).astype(np.int64)
    # redraw only the entries that repeat an earlier entry of their row (terminates quickly for any k < N-1)
    rnd = 1
    while True:
        order = np.argsort(off, axis=1, kind="stable")
        srt = np.take_along_axis(off, order, axis=1)
        dup_sorted = np.zeros_like(srt, dtype=bool)
        dup_sorted[:, 1:] = srt[:, 1:] == srt[:, :-1]
        if not dup_sorted.any():
            break
        rows, pos = np.nonzero(dup_sorted)
        cols = order[rows, pos]
        off[rows, cols] = 1 + (rand_u64(seed + 1000 * rnd, rows * k + cols, rnd) % np.uint64(N - 1)).astype(np.int64)
        rnd += 1
        if rnd > 10_000:
            raise RuntimeError("knn_uniform did not converge")
    return (((cells[:, None] + off) % N) + 1).astype(dtype)


def counts_csc(G: int, N: int, median_frac: float = 0.07, sigma: float = 0.5, zipf_s: float = 0.9,
               max_per_cell: int | None = None, seed: int = 7):
    """Synthetic UMI count matrix, CSC genes x cells.

    Per-cell number of gene draws ~ clipped lognormal (median ``median_frac*G``); genes
    drawn from a Zipf-like popularity (so per-gene cell counts span 1..N and the 5 %
    filter of gficf() removes a real fraction) and de-duplicated within a cell; values are
    1 + Geometric(0.5) integers stored as float64; row indices sorted within a column;
    every cell has at least one entry.

    Returns (colptr int64[N+1], rowidx int32[nnz], x float64[nnz]).
    """
    cells = np.arange(N, dtype=np.int64)
    z = np.sqrt(-2.0 * np.log(1.0 - rand_unit(seed, cells, 0))) * np.cos(2 * np.pi * rand_unit(seed, cells, 1))
    n_draw = np.clip(np.rint(median_frac * G * np.exp(sigma * z)), 1, G).astype(np.int64)
    if max_per_cell is not None:
        n_draw = np.minimum(n_draw, max_per_cell)
    pop = 1.0 / np.power(np.arange(1, G + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]
    start = np.concatenate([[0], np.cumsum(n_draw)])
    tot = int(start[-1])
    cell_of = np.repeat(cells, n_draw)
    within = np.arange(tot, dtype=np.int64) - np.repeat(start[:-1], n_draw)
    gene = np.searchsorted(cdf, rand_unit(seed + 1, cell_of, within), side="right").astype(np.int64)
    gene = np.minimum(gene, G - 1)
    key = np.unique(cell_of * G + gene)            # sorted by (cell, gene), de-duplicated
    col = key // G
    rowidx = (key - col * G).astype(np.int32)
    colptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(colptr, col + 1, 1)
    colptr = np.cumsum(colptr)
    u = rand_unit(seed + 2, key)
    x = 1.0 + np.floor(-np.log2(1.0 - u))          # 1 + Geometric(1/2) on {0,1,...}
    return colptr, rowidx, x.astype(np.float64)
