import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gficf_amd
rng = np.random.default_rng(3)
X = rng.normal(size=(30000, 10))
A = gficf_amd.jaccard_adjacency(gficf_amd.clustcells_graph(X, 20, "euclidean"), 30000)
ref = None
for i in range(20):
    for alg in (1, 2):
        lab = gficf_amd.run_modularity_clustering(A, 1, 1.0, alg, 3, 10, 7, False)
        key = (alg,)
        if ref is None: ref = {}
        if key not in ref: ref[key] = (lab.copy(), lab.modularity)
        assert np.array_equal(ref[key][0], lab) and ref[key][1] == lab.modularity, (i, alg)
print("40 runs identical:", {k: (int(v[0].max()) + 1, v[1]) for k, v in ref.items()})
