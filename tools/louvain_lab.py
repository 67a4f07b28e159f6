"""Quality / time of the device Louvain against the reference's optimiser (oracle/_ref/modularity_optimizer).
Usage: python tools/louvain_lab.py N k [data=blobs|uniform] [resolution] [n_iter]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd
import oracle
from oracle import oracle_np

N, k = int(sys.argv[1]), int(sys.argv[2])
data = sys.argv[3] if len(sys.argv) > 3 else "blobs"
res = float(sys.argv[4]) if len(sys.argv) > 4 else 0.8
n_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 10
alg = int(os.environ.get("LAB_ALG", "1"))
rng = np.random.default_rng(5)
d = 20
if data == "blobs":
    C = 25
    cen = rng.normal(size=(C, d)) * 4.0
    X = cen[rng.integers(0, C, N)] + rng.normal(size=(N, d))
else:
    X = rng.normal(size=(N, d))
edges = gficf_amd.clustcells_graph(X, k, "manhattan")
A = gficf_amd.jaccard_adjacency(edges, N)
print(f"graph: N={N} k={k} {data}: nnz={A.nnz}, max degree {np.diff(A.indptr).max()}")
t0 = time.perf_counter()
lab = gficf_amd.run_modularity_clustering(A, 1, res, alg, 1, n_iter, 0, False)
t1 = time.perf_counter()
lab2 = gficf_amd.run_modularity_clustering(A, 1, res, alg, 1, n_iter, 0, False)
t2 = time.perf_counter()
q = oracle_np.modularity_np(A, lab, res)
print(f"device: {lab.n_clusters} clusters, Q={lab.modularity:.6f} (numpy {q:.6f}), {(t2 - t1) * 1e3:.1f} ms host call (first {1e3 * (t1 - t0):.1f}), "
      f"deterministic={np.array_equal(lab, lab2)}")
if os.environ.get("LAB_REF", "1") != "0":
    t0 = time.perf_counter()
    rl, rq = oracle.modularity_reference(A, res, alg, 1, n_iter, 0)
    t1 = time.perf_counter()
    print(f"reference (1 start, {n_iter} iterations): {rl.max() + 1} clusters, Q={oracle_np.modularity_np(A, rl, res):.6f} (printed {rq}), {t1 - t0:.2f} s incl. file I/O")
