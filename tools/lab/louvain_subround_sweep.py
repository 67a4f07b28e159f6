import os, sys, itertools
sys.path.insert(0, os.getcwd())
import numpy as np
import gficf_amd, oracle
from oracle import oracle_np
rows = []
case = 0
for N, k, C, res, alg in itertools.product((1500, 6000, 20000), (10, 30), (1, 8), (0.5, 1.0, 2.0), (1, 2)):
    case += 1
    if N != 1500 and not (N == 6000 and C == 1): continue
    rng = np.random.default_rng(case)
    d = 8 + case % 9
    X = rng.normal(size=(C, d))[rng.integers(0, C, N)] * (2.5 if C > 1 else 0.0) + rng.normal(size=(N, d))
    A = gficf_amd.jaccard_adjacency(gficf_amd.clustcells_graph(X, k, "euclidean"), N)
    ref, _ = oracle.modularity_reference(A, res, alg, 1, 10, 0)
    qr = oracle_np.modularity_np(A, ref, res)
    out = []
    for S in ("", "16", "32", "64"):
        if S: os.environ["GFICF_LOUVAIN_SUBROUNDS"] = S
        else: os.environ.pop("GFICF_LOUVAIN_SUBROUNDS", None)
        lab = gficf_amd.run_modularity_clustering(A, 1, res, alg, 1, 10, 0, False)
        out.append(oracle_np.modularity_np(A, lab, res) - qr)
    rows.append(out)
    print(f"N={N} k={k} C={C} res={res} alg={alg}: " + "  ".join(f"{v:+.5f}" for v in out))
r = np.array(rows)
print("min per setting (default,16,32,64):", r.min(axis=0), "mean:", r.mean(axis=0))
