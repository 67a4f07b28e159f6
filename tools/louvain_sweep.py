"""Modularity of the device Louvain minus the reference optimiser's on a sweep of kNN -> Jaccard graphs (sizes, k, structure,
resolutions, both algorithms).  Prints every case and the extremes of the difference."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd
import oracle
from oracle import oracle_np

diffs = []
case = 0
for N, k, C, res, alg in itertools.product((1500, 6000, 20000), (10, 30), (1, 8), (0.5, 1.0, 2.0), (1, 2)):
    case += 1
    rng = np.random.default_rng(case)
    d = 8 + case % 9
    X = rng.normal(size=(C, d))[rng.integers(0, C, N)] * (2.5 if C > 1 else 0.0) + rng.normal(size=(N, d))
    A = gficf_amd.jaccard_adjacency(gficf_amd.clustcells_graph(X, k, "euclidean"), N)
    lab = gficf_amd.run_modularity_clustering(A, 1, res, alg, 1, 10, 0, False)
    ref, _ = oracle.modularity_reference(A, res, alg, 1, 10, 0)
    qd, qr = oracle_np.modularity_np(A, lab, res), oracle_np.modularity_np(A, ref, res)
    diffs.append(qd - qr)
    print(f"N={N:6d} k={k:2d} C={C} res={res} alg={alg}: device {qd:.5f} ({lab.n_clusters:4d} clusters)  reference {qr:.5f} ({ref.max() + 1:4d})  diff {qd - qr:+.5f}")
diffs = np.array(diffs)
print(f"{len(diffs)} cases: diff min {diffs.min():+.5f}  mean {diffs.mean():+.5f}  max {diffs.max():+.5f}")
