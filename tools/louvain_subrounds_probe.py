"""Quality of the device Louvain against the number of synchronous sub-rounds an iteration is cut into (GFICF_LOUVAIN_SUBROUNDS; default: 2 above
50 000 vertices of a component, 4 / 8 / 16 below 50 000 / 4 000 / 400) on graphs where simultaneous moves could hurt: small graphs, weak structure,
no structure.  Modularity by numpy on the returned labels, next to the REFERENCE binary's (oracle/_ref/modularity_optimizer).
Usage: python tools/louvain_subrounds_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import gficf_amd
import oracle
from oracle import oracle_np


def sbm(n, c, p_in, p_out, seed):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, c, n)
    m_in, m_out = int(p_in * n * n / c / 2), int(p_out * n * n / 2)
    i = rng.integers(0, n, 4 * m_in); j = rng.integers(0, n, 4 * m_in)
    keep = (lab[i] == lab[j]) & (i != j)
    i, j = i[keep][:m_in], j[keep][:m_in]
    a = rng.integers(0, n, m_out); b = rng.integers(0, n, m_out)
    k2 = a != b
    i, j = np.concatenate([i, a[k2]]), np.concatenate([j, b[k2]])
    w = rng.random(len(i)) * 0.9 + 0.1
    A = sp.coo_matrix((w, (i, j)), shape=(n, n)).tocsc()
    A = (A + A.T).tocsc()
    A.sum_duplicates(); A.sort_indices()
    return A


def knn_noise(n, k, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, 10))
    return gficf_amd.jaccard_adjacency(gficf_amd.clustcells_graph(X, k, "manhattan"), n)


graphs = [("SBM 300 x 6, strong", sbm(300, 6, 0.3, 0.005, 1)), ("SBM 1000 x 10, weak", sbm(1000, 10, 0.06, 0.01, 2)), ("SBM 5000 x 20, weak", sbm(5000, 20, 0.02, 0.002, 3)),
          ("random 2000, no structure", sbm(2000, 1, 0.0, 0.004, 4)), ("kNN graph of noise, 20000 x 15", knn_noise(20000, 15, 5)),
          ("SBM 60000 x 50, weak", sbm(60000, 50, 0.004, 0.0002, 6))]
for name, A in graphs:
    n = A.shape[0]
    ref_lab, ref_q = oracle.modularity_reference(A, 0.8, 1, 10, 10, 0)
    row = [f"{name:32s} n={n:6d} nnz={A.nnz:8d}  reference Q {oracle_np.modularity_np(A, ref_lab, 0.8):.4f} |"]
    for s in ("default", "1", "2", "4", "8", "16"):
        if s == "default":
            os.environ.pop("GFICF_LOUVAIN_SUBROUNDS", None)
        else:
            os.environ["GFICF_LOUVAIN_SUBROUNDS"] = s
        r = gficf_amd.run_modularity_clustering(A, 1, 0.8, 1, 10, 10, 180582)
        lab = np.asarray(r["labels"] if isinstance(r, dict) else r)
        row.append(f"S={s}: {oracle_np.modularity_np(A, lab, 0.8):.4f}")
    print("  ".join(row), flush=True)
