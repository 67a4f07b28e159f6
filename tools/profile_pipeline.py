"""Where the host mirror's wall time goes at BASELINE config 3 (23 k genes x 54 k cells): cProfile of gficf() and clustcells() —
the library calls against the Python around them (scipy conversions, subsetting)."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import gficf_amd
from gficf_amd import synth

G, N, k = 23000, 54000, 30
cp, ri, x = synth.counts_csc(G, N, seed=7)
M = sp.csc_matrix((x, ri, cp), shape=(G, N))
rng = np.random.default_rng(1)
C = 30
pca = rng.normal(size=(C, 50))[rng.integers(0, C, N)] * 3.0 + rng.normal(size=(N, 50))
gficf_amd.gficf(M[:, :2000], normalize=False, verbose=False)
data = gficf_amd.gficf(M, normalize=False, verbose=False)          # warm: pools, pinned stages
data["pca"] = {"cells": pca}
gficf_amd.clustcells(data, k=k, community_algo="louvian 2", verbose=False)


def prof(label, fn):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = fn()
    pr.disable()
    ms = 1e3 * (time.perf_counter() - t0)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    print(f"==== {label}: {ms:.1f} ms")
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:40]))
    return out


data = prof("gficf(M)", lambda: gficf_amd.gficf(M, normalize=False, verbose=False))
data["pca"] = {"cells": pca}
prof("clustcells(k=30, louvian 2)", lambda: gficf_amd.clustcells(data, k=k, community_algo="louvian 2", verbose=False))
prof("clustcells(store_graph=False)", lambda: gficf_amd.clustcells(data, k=k, community_algo="louvian 2", verbose=False, store_graph=False))
