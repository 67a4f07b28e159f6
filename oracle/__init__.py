"""CPU oracle for the GF-ICF / Jaccard hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / timed CPU baseline.  The product package
``gficf_amd`` never imports it.

PARITY UNPINNED: the reference (dibbelab/gficf) has no tests or golden vectors for this
path, R is absent from this image and the reference's Jaccard translation unit needs
Rcpp / RcppParallel headers that are absent too, so neither restatement could be checked
against the reference itself.  See ``jaccard_oracle.cpp`` / ``gficf_oracle.cpp`` headers.

One part of the reference DOES build here and is used as the checker of scope row N4 (community detection) only:
``src/ModularityOptimizer.cpp`` is plain C++ with its own ``main()`` under ``-DSTANDALONE``; ``build_ref()`` compiles it from
where it lies into ``oracle/_ref/modularity_optimizer`` and ``modularity_reference()`` runs it (nothing of it is copied).

Two independent writings of the same algorithm live here:
  * ``liboracle.so``  — C++ (``jaccard_oracle.cpp``, ``gficf_oracle.cpp``), built by
    ``make -C oracle``; this is also the timed CPU baseline.
  * ``oracle_np``     — numpy/scipy, used to cross-check the C++ one.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with g++ (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_LIB_PATH)
        for s in ("jaccard_oracle.cpp", "gficf_oracle.cpp", "knn_oracle.cpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


REFERENCE_ROOT = "/root/reference"
_REF_LOUVAIN = os.path.join(_HERE, "_ref", "modularity_optimizer")


def build_ref() -> str | None:
    """oracle/_ref/modularity_optimizer: the reference's own src/ModularityOptimizer.cpp compiled with -DSTANDALONE
    (``make -C oracle ref``).  Built only where /root/reference exists (the container); the GPU box uses the file that
    travelled with the snapshot.  Returns the path, or None when there is neither a source tree nor a built file."""
    src = os.path.join(REFERENCE_ROOT, "src", "ModularityOptimizer.cpp")
    if os.path.exists(src):
        subprocess.check_call(["make", "-C", _HERE, "ref", f"REFERENCE={REFERENCE_ROOT}"])
    return _REF_LOUVAIN if os.path.exists(_REF_LOUVAIN) else None


def modularity_reference(A, resolution: float = 0.8, algorithm: int = 1, n_start: int = 10, n_iter: int = 10, seed: int = 0, function: int = 1):
    """Runs the REFERENCE's modularity optimiser (oracle/_ref/modularity_optimizer, built from
    /root/reference/src/ModularityOptimizer.cpp) on the symmetric adjacency matrix ``A`` exactly as
    RunModularityClusteringCpp feeds it (src/RModularityOptimizer.cpp:66-84: strict lower triangle, node1 = column,
    node2 = row).  Returns (labels int32[N] ordered by decreasing cluster size, modularity as printed, 4 decimals)."""
    import re
    import tempfile

    import scipy.sparse as sp

    exe = _REF_LOUVAIN if os.path.exists(_REF_LOUVAIN) else build_ref()
    if exe is None:
        raise FileNotFoundError("oracle/_ref/modularity_optimizer is not built and /root/reference is absent")
    L = sp.tril(sp.csc_matrix(A), k=-1).tocoo()
    N = A.shape[0]
    if L.nnz == 0 or max(L.row.max(), L.col.max()) != N - 1:
        raise ValueError("the reference sizes the network by its largest vertex id: the last vertex needs an edge")
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "edges.txt"), os.path.join(d, "clusters.txt")
        try:                                                     # a few million lines: pandas' C writer, exact round trip (%.17g)
            import pandas as pd

            pd.DataFrame({"a": L.col, "b": L.row, "w": L.data}).to_csv(fin, sep="\t", header=False, index=False, float_format="%.17g")
        except ImportError:
            with open(fin, "w") as f:
                for c, r, v in zip(L.col.tolist(), L.row.tolist(), L.data.tolist()):
                    f.write(f"{c}\t{r}\t{v!r}\n")
        out = subprocess.run([exe, fin, fout, str(int(function)), repr(float(resolution)), str(int(algorithm)), str(int(n_start)), str(int(n_iter)),
                              str(int(seed)), "1"], check=True, capture_output=True, text=True).stdout
        labels = np.loadtxt(fout, dtype=np.int64).astype(np.int32).reshape(-1)
    m = re.findall(r"(?:^|\n)(?:Modularity|Maximum modularity in \d+ random starts): (-?[0-9.]+)", out)
    return labels, float(m[-1]) if m else float("nan")        # the last line printed is the final value


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_double
        vp = ctypes.c_void_p
        L.oracle_jaccard_f64.restype = i32
        L.oracle_jaccard_f64.argtypes = [vp, i64, i32, vp, vp, i32]
        L.oracle_jaccard_cells_f64.restype = i32
        L.oracle_jaccard_cells_f64.argtypes = [vp, i64, i32, i64, i64, vp, vp, i32]
        L.oracle_jaccard_i32.restype = i32
        L.oracle_jaccard_i32.argtypes = [vp, i64, i32, vp, vp, i32]
        L.oracle_jaccard_coeff_f64.restype = i32
        L.oracle_jaccard_coeff_f64.argtypes = [vp, i64, i32, vp]
        L.oracle_gficf_csc.restype = i32
        L.oracle_gficf_csc.argtypes = [i64, i64, vp, vp, vp, dbl, dbl, vp] + [vp] * 8
        L.oracle_gficf_csc_ex.restype = i32
        L.oracle_gficf_csc_ex.argtypes = [i64, i64, vp, vp, vp, dbl, dbl, vp, i32, i32] + [vp] * 8
        L.oracle_gficf_csc_mt.restype = i32
        L.oracle_gficf_csc_mt.argtypes = [i64, i64, vp, vp, vp, dbl, dbl, vp, i32, i32, i32] + [vp] * 8
        L.oracle_knn.restype = i32
        L.oracle_knn.argtypes = [vp, i64, i32, i64, i32, i32, vp, vp, i32]
        L.oracle_knn_block.restype = i32
        L.oracle_knn_block.argtypes = [vp, i64, i32, i64, i32, i32, i64, i64, vp, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def jaccard(mat: np.ndarray, nthreads: int = 1):
    """Reference-layout Jaccard: ``mat`` is N x k (ids 1-based; int32 or float64).

    Returns (rmat, u): rmat is the (N*k) x 3 float64 matrix of
    rcpp_parallel_jaccard_coef (reference src/rcpp_parallel_jaccard_coeff.cpp:59-80),
    u the N*k int32 intersection counts in edge order i*k+j.
    """
    N, k = mat.shape
    E = N * k
    rm = np.empty((3, E), dtype=np.float64)  # column-major (E x 3) == C-order (3 x E)
    u = np.empty(E, dtype=np.int32)
    if mat.dtype == np.int32:
        m = np.asfortranarray(mat)
        rc = lib().oracle_jaccard_i32(_p(m), N, k, _p(rm), _p(u), nthreads)
    else:
        m = np.asfortranarray(mat, dtype=np.float64)
        rc = lib().oracle_jaccard_f64(_p(m), N, k, _p(rm), _p(u), nthreads)
    if rc != 0:
        raise ValueError(f"oracle_jaccard: invalid input (rc={rc})")
    return rm.T, u  # view: (E x 3), Fortran-ordered like the R matrix


def jaccard_cells(mat: np.ndarray, begin: int, end: int, nthreads: int = 1):
    """The rows of ``jaccard(mat)`` that belong to the cells [begin, end): ((end-begin)*k) x 3 matrix and counts
    (same per-edge code; a bounded sample for checks at sizes where the whole matrix takes too long)."""
    N, k = mat.shape
    n = (end - begin) * k
    rm = np.empty((3, n), dtype=np.float64)
    u = np.empty(n, dtype=np.int32)
    m = np.asfortranarray(mat, dtype=np.float64)
    rc = lib().oracle_jaccard_cells_f64(_p(m), N, k, int(begin), int(end), _p(rm), _p(u), nthreads)
    if rc != 0:
        raise ValueError(f"oracle_jaccard_cells: invalid input (rc={rc})")
    return rm.T, u


def jaccard_coeff(mat: np.ndarray) -> np.ndarray:
    """The serial entry ``jaccard_coeff(idx, printOutput)`` (reference src/jaccard_coeff.cpp:19-44): set intersection,
    rows with u > 0 packed from the top of the (N*k) x 3 matrix."""
    N, k = mat.shape
    m = np.asfortranarray(mat, dtype=np.float64)
    w = np.empty((3, N * k), dtype=np.float64)
    rc = lib().oracle_jaccard_coeff_f64(_p(m), N, k, _p(w))
    if rc != 0:
        raise ValueError(f"oracle_jaccard_coeff: invalid input (rc={rc})")
    return w.T


def gficf_csc(G, N, colptr, rowidx, x, prop_min=0.05, prop_max=1.0, w_in=None, icf_type="classic", norm="l2", threads=1):
    """GF-ICF on a CSC genes x cells matrix (reference R/gficf.R:17-33, normalize=FALSE).

    ``threads > 1`` runs the multi-threaded restatement (cells cut into ranges; same sums in the same order, so the
    same bits).  Returns dict(keep, nt, w, colptr, rowidx, x, G_kept).
    """
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowidx = np.ascontiguousarray(rowidx, dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=np.float64)
    nnz = int(colptr[N])
    keep = np.zeros(G, dtype=np.uint8)
    nt = np.zeros(G, dtype=np.int64)
    w = np.zeros(G, dtype=np.float64)
    ocp = np.zeros(N + 1, dtype=np.int64)
    ori = np.zeros(max(nnz, 1), dtype=np.int32)
    ox = np.zeros(max(nnz, 1), dtype=np.float64)
    gk = ctypes.c_int64(0)
    nk = ctypes.c_int64(0)
    if w_in is not None:
        w_in = np.ascontiguousarray(w_in, dtype=np.float64)
    opts = ({"classic": 0, "prob": 1, "smooth": 2}[icf_type], {"l2": 0, "l1": 1}[norm])
    fn, extra = (lib().oracle_gficf_csc_mt, (int(threads),)) if threads > 1 else (lib().oracle_gficf_csc_ex, ())
    rc = fn(
        G, N, _p(colptr), _p(rowidx), _p(x), float(prop_min), float(prop_max), _p(w_in), *opts, *extra,
        _p(keep), _p(nt), _p(w), _p(ocp), _p(ori), _p(ox),
        ctypes.cast(ctypes.byref(gk), ctypes.c_void_p), ctypes.cast(ctypes.byref(nk), ctypes.c_void_p))
    if rc != 0:
        raise ValueError(f"oracle_gficf_csc: malformed CSC (rc={rc})")
    n = nk.value
    return dict(keep=keep.astype(bool), nt=nt, w=w, colptr=ocp, rowidx=ori[:n].copy(),
                x=ox[:n].copy(), G_kept=gk.value)


KNN_METRICS = {"manhattan": 0, "euclidean": 1, "cosine": 2, "correlation": 3}


def knn(X: np.ndarray, k: int, metric: str = "manhattan", nthreads: int = 1, queries: tuple | None = None):
    """Exact kNN in f32 (the contract the approximate ``uwot:::find_nn(..., method="annoy")`` call of
    reference R/clustCells.R:57,60 aims at): X is N x d, returns (idx N x k int32 1-based, dist N x k
    float64), the k smallest (distance, index) pairs per row, the row itself included.  ``queries=(b, e)``
    restricts the search to rows [b, e) (the other rows of the result stay zero)."""
    X = np.asfortranarray(X, dtype=np.float64)
    N, d = X.shape
    idx = np.zeros((k, N), dtype=np.int32)     # C-order (k, N) == column-major N x k
    dist = np.zeros((k, N), dtype=np.float64)
    qb, qe = queries if queries is not None else (0, N)
    rc = lib().oracle_knn_block(_p(X), N, d, max(N, 1), int(k), KNN_METRICS[metric], int(qb), int(qe), _p(idx), _p(dist), int(nthreads))
    if rc != 0:
        raise ValueError(f"oracle_knn: rc={rc}")
    return idx.T, dist.T
