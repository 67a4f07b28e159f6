"""The stages of one gficf_phenograph_host call, timed by the library itself (GFICF_PHENOGRAPH_DEBUG=1: a stream synchronisation behind each stage).
Usage: python tools/phenograph_stages.py [N k n_start]   (default: the config-3 shape, 54000 x 50 points, k = 30, n.start = 10)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd

N, k, n_start = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (54000, 30, 10)))
rng = np.random.default_rng(1)
X = rng.normal(size=(30, 50))[rng.integers(0, 30, N)] * 3.0 + rng.normal(size=(N, 50))
for rep in range(3):
    if rep == 2:
        os.environ["GFICF_PHENOGRAPH_DEBUG"] = "1"
    r = gficf_amd.phenograph(X, k, "manhattan", 0.8, 1, n_start, 10, 180582)
print(f"N={N} k={k} n_start={n_start}: {r.n_clusters} clusters, Q {r.modularity:.6f}")
