"""The exact search on the config-3 stand-in (54 k points, 30 Gaussian blobs in 50 dimensions, manhattan, k + 1 = 31): the form the device picks by
itself against the pruned form forced (GFICF_KNN_PRUNE=1) — is the "nothing to prune" decision right on this data?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd

rng = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 54000
X = rng.normal(size=(30, 50))[rng.integers(0, 30, N)] * 3.0 + rng.normal(size=(N, 50))
ref = None
for p in (os.environ.get("PROBE_FORMS", "0,1,auto,0,1,auto").split(",")):
    if p == "auto":
        os.environ.pop("GFICF_KNN_PRUNE", None)
    else:
        os.environ["GFICF_KNN_PRUNE"] = p
    gficf_amd.find_nn(X, 31, metric="manhattan")
    t0 = time.perf_counter()
    r = gficf_amd.find_nn(X, 31, metric="manhattan")
    t = time.perf_counter() - t0
    same = ref is None or bool(np.array_equal(ref, r["idx"]))
    ref = r["idx"] if ref is None else ref
    print(f"find_nn host call, GFICF_KNN_PRUNE={p}: {1e3 * t:.2f} ms   same neighbours as the first run: {same}", flush=True)
