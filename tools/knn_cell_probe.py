"""The search's cell size on data with MANY SMALL clusters (where cells larger than a cluster blur the bounds): N points in C clusters, d dimensions,
the default cell size against GFICF_KNN_PIVOT_CELL=256.  Usage: python tools/knn_cell_probe.py [N C d]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd

N, C, d = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (300000, 1000, 20)))
rng = np.random.default_rng(4)
X = np.asfortranarray(rng.normal(size=(C, d))[rng.integers(0, C, N)] * 3.0 + rng.normal(size=(N, d)))
ref = None
for cell in (os.environ.get("PROBE_CELL"), "256", os.environ.get("PROBE_CELL"), "256"):
    if cell is None:
        os.environ.pop("GFICF_KNN_PIVOT_CELL", None)
    else:
        os.environ["GFICF_KNN_PIVOT_CELL"] = cell
    gficf_amd.find_nn(X, 31, metric="manhattan")
    t0 = time.perf_counter()
    r = gficf_amd.find_nn(X, 31, metric="manhattan")
    t = time.perf_counter() - t0
    same = ref is None or bool(np.array_equal(ref, r["idx"]))
    ref = r["idx"] if ref is None else ref
    print(f"N={N} clusters={C} d={d}, cell {'default' if cell is None else cell}: find_nn host call {1e3 * t:.1f} ms   same neighbours: {same}", flush=True)
