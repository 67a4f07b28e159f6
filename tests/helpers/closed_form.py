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
