#!/usr/bin/env python3
"""gficf_phenograph_host with the Jaccard stage on the caller's cell order (GFICF_PHENOGRAPH_ORDER=0) against cells renumbered in the
search's pivot order (=1; round 5's default from 2^17 cells on, off by default since round 6): same labels, edges and modularity; wall time of
the call (the first of each is a first call: the pool grows) and, with GFICF_PHENOGRAPH_DEBUG=1, the library's own stage times.  Usage: python tools/phenograph_order_ab.py [N] [d] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gficf_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 10
k = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(3)
centers = rng.normal(scale=8.0, size=(60, d))
lab = rng.integers(0, 60, size=N)
X = centers[lab] + rng.normal(size=(N, d))
res = {}
for order in ("0", "1", "0", "1"):
    os.environ["GFICF_PHENOGRAPH_ORDER"] = order
    t0 = time.perf_counter()
    r = gficf_amd.phenograph(X, k=k, n_start=1, n_iter=2)
    dt = time.perf_counter() - t0
    print(f"GFICF_PHENOGRAPH_ORDER={order}: {dt * 1e3:8.1f} ms  clusters {r.n_clusters}  edges {r.n_edges}  Q {r.modularity:.6f}", flush=True)
    res.setdefault(order, r)
same = (np.array_equal(np.asarray(res["0"]), np.asarray(res["1"])) and res["0"].n_edges == res["1"].n_edges
        and res["0"].modularity == res["1"].modularity)
print("identical labels, edge count and modularity:", same)
sys.exit(0 if same else 1)
