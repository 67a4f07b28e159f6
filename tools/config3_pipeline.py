"""BASELINE config 3 shape end to end through the host mirror: gficf() on a 23 k genes x 54 k cells synthetic count matrix, then
clustcells(k = 30) on a 50-component stand-in for the PCA space (PCA itself is third-party, RSpectra/irlba, and out of scope:
the stand-in is clustered Gaussian data).  Prints the wall time of every call (host entries: PCIe both ways included).
gficf() is timed as the reference calls it (storeRaw = TRUE: $rawCounts = M[keep, ], gathered by the library's host threads while the result comes
back; through round 4 and most of round 5 the mirror subset M with scipy: +150 ms) and without."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import gficf_amd
from gficf_amd import synth

G, N, k = 23000, 54000, 30
t0 = time.perf_counter()
cp, ri, x = synth.counts_csc(G, N, seed=7)
M = sp.csc_matrix((x, ri, cp), shape=(G, N))
rng = np.random.default_rng(1)
C = 30
pca = rng.normal(size=(C, 50))[rng.integers(0, C, N)] * 3.0 + rng.normal(size=(N, 50))
print(f"synthetic input: {G} x {N}, nnz {M.nnz}, built in {time.perf_counter() - t0:.1f} s (host)")
gficf_amd.gficf(M[:, :2000], normalize=False, verbose=False)                      # warm-up: context, library
data = cells = noraw = None
for rep in range(3):
    data = cells = noraw = None                      # the previous run's results are released OUTSIDE the timed calls (a GB of vectors: tens of ms)
    tn = time.perf_counter()
    noraw = gficf_amd.gficf(M, normalize=False, verbose=False, storeRaw=False)
    t0 = time.perf_counter()
    data = gficf_amd.gficf(M, normalize=False, verbose=False)
    t1 = time.perf_counter()
    data["pca"] = {"cells": pca}
    data = gficf_amd.clustcells(data, k=k, community_algo="louvian 2", verbose=False)
    t2 = time.perf_counter()
    cells = gficf_amd.transpose_gficf(data["gficf"])
    t3 = time.perf_counter()
    fused = gficf_amd.phenograph(pca, k, "manhattan", 0.8, 1, 10, 10, 180582)
    t4 = time.perf_counter()
    print(f"run {rep}: gficf() {1e3 * (t1 - t0):.1f} ms with $rawCounts, {1e3 * (t0 - tn):.1f} ms without ({data['gficf'].shape[0]} genes kept), clustcells(k={k}, louvian 2) {1e3 * (t2 - t1):.1f} ms "
          f"({len(set(data['cluster']))} clusters, Q {data['modularity']:.4f}), t(gficf) {1e3 * (t3 - t2):.1f} ms, "
          f"phenograph() in one call {1e3 * (t4 - t3):.1f} ms ({fused.n_clusters} clusters)")
t0 = time.perf_counter()
want = M[data["genes"], :]
t1 = time.perf_counter()
same = np.array_equal(want.indptr, data["rawCounts"].indptr) and np.array_equal(want.indices, data["rawCounts"].indices) and np.array_equal(want.data, data["rawCounts"].data)
print(f"$rawCounts against the row subset M[keep, ] by scipy (what the mirror did through round 4: {1e3 * (t1 - t0):.1f} ms on top of the call): identical {same}")
