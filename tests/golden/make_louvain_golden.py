"""Generates tests/golden/louvain_cases.npz: outputs of the REFERENCE's modularity optimiser on small graphs.

Unlike the Jaccard / GF-ICF fixtures these ARE reference outputs: src/ModularityOptimizer.cpp of dibbelab/gficf is
plain standard C++ with its own command-line main() under -DSTANDALONE, so `make -C oracle ref` compiles it as it is
(from /root/reference, into oracle/_ref/, nothing copied) and this script runs it.  Stored per case: the symmetric
adjacency matrix (CSC arrays), the parameters, the labels the reference returned and the modularity it printed.
Run from the repo root, in the container that holds /root/reference:  python tests/golden/make_louvain_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import oracle_np  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def planted(N, C, p_in, p_out, seed):
    rng = np.random.default_rng(seed)
    lab = np.sort(rng.integers(0, C, N))
    P = np.where(lab[:, None] == lab[None, :], p_in, p_out)
    M = np.triu(rng.random((N, N)) < P, 1) * rng.integers(1, 64, (N, N)) / 64.0
    return sp.csc_matrix(M + M.T)


def knn_jaccard_graph(N, d, k, C, seed):
    """The clustcells() graph through the CPU oracles: exact kNN -> Jaccard edges -> weight > 0 -> A = W + W^T."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(C, d))[rng.integers(0, C, N)] * 3.0 + rng.normal(size=(N, d))
    idx, _ = oracle.knn(X, k + 1, "manhattan", nthreads=8)
    rm, _ = oracle.jaccard(np.ascontiguousarray(idx[:, 1:]), nthreads=8)
    rm = rm[rm[:, 2] > 0]
    i, j = rm[:, 0].astype(np.int64) - 1, rm[:, 1].astype(np.int64) - 1
    W = sp.coo_matrix((rm[:, 2], (i, j)), shape=(N, N)).tocsc()
    A = (W + W.T - sp.diags(W.diagonal())).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return A


def main():
    cases = {
        "planted3": (planted(300, 3, 0.2, 0.01, 1), 1.0, 1, 3, 10, 0),
        "planted8_res08": (planted(800, 8, 0.15, 0.004, 2), 0.8, 1, 10, 10, 0),
        "knn_blobs": (knn_jaccard_graph(2000, 12, 15, 9, 3), 0.8, 1, 10, 10, 0),
        "knn_noise_alg2": (knn_jaccard_graph(1500, 10, 10, 1, 4), 1.0, 2, 5, 10, 7),
    }
    out = {}
    for name, (A, res, alg, n_start, n_iter, seed) in cases.items():
        A = sp.csc_matrix(A)
        A.sort_indices()
        labels, q = oracle.modularity_reference(A, res, alg, n_start, n_iter, seed)
        qn = oracle_np.modularity_np(A, labels, res)
        assert abs(q - qn) < 6e-5, (name, q, qn)               # the restated quality function against the reference's print-out
        print(f"{name}: N={A.shape[0]} nnz={A.nnz} clusters={labels.max() + 1} Q={qn:.6f} (printed {q})")
        out[name + "/indptr"] = A.indptr.astype(np.int64)
        out[name + "/indices"] = A.indices.astype(np.int32)
        out[name + "/data"] = A.data.astype(np.float64)
        out[name + "/params"] = np.array([res, alg, n_start, n_iter, seed], dtype=np.float64)
        out[name + "/labels"] = labels.astype(np.int32)
        out[name + "/printed_q"] = np.array([q])
    np.savez_compressed(os.path.join(OUT, "louvain_cases.npz"), **out)


if __name__ == "__main__":
    main()
