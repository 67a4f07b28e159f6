"""Calls every host entry of the C ABI repeatedly and prints the device's free memory before and after (leak check)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

import gficf_amd
from gficf_amd import synth

rng = np.random.default_rng(0)
N, G, k = 4000, 900, 15
X = rng.normal(size=(N, 12))
cp, ri, x = synth.counts_csc(G, N, seed=2)
M = sp.csc_matrix((x, ri, cp), shape=(G, N))


big = synth.knn_windowed(40000, 30, seed=4, perm_seed=5)            # 1.2 M edges: the compact return of gficf_jaccard_host (pinned staging, host threads)
wide = synth.knn_windowed(1200, 300, W=200, seed=6, perm_seed=7)      # k > 256: the sorted-row path
ops = gficf_amd.HipOps(0)
ops.set_jaccard_distinct(True)
d_idx = torch.from_numpy(np.ascontiguousarray(synth.knn_windowed(3000, 15, seed=8).T)).cuda()
d_tab = torch.zeros((3000, ops.row_words(3000, 15)), dtype=torch.int32, device="cuda")
d_out = torch.zeros((3, 3000 * 15), dtype=torch.float64, device="cuda")


def rss_mib():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def once():
    gficf_amd.rcpp_parallel_jaccard_coef(big, False)
    gficf_amd.rcpp_parallel_jaccard_coef(wide, False)
    run = ops.jaccard_prepared(d_idx, 3000, 15, d_tab, d_out, None)     # a new prepared call every round (the one-launch form)
    for _ in range(5):
        run()
    ops.sync()
    data = gficf_amd.gficf(M, normalize=False, verbose=False)
    nn = gficf_amd.find_nn(X, k + 1, True, "manhattan")["idx"]
    rel = gficf_amd.rcpp_parallel_jaccard_coef(nn[:, 1:], False)
    gficf_amd.jaccard_coeff(nn[:200, 1:] % 200 + 1, False)
    edges = gficf_amd.jaccard_edges(nn)
    A = gficf_amd.jaccard_adjacency(edges, N)
    lab = gficf_amd.run_modularity_clustering(A, 1, 0.8, 2, 2, 3, 1, False)
    gficf_amd.cluster_signatures(data["gficf"], lab)
    gficf_amd.transpose_gficf(data["gficf"])
    gficf_amd.phenograph(X, k, "euclidean", 0.8, 1, 1, 2, 0)
    return rel.shape


once()
once()
torch.cuda.synchronize()
free0, rss0 = torch.cuda.mem_get_info()[0], rss_mib()
for i in range(40):
    once()
torch.cuda.synchronize()
free1, rss1 = torch.cuda.mem_get_info()[0], rss_mib()
print(f"device memory free before {free0 / 2**20:.1f} MiB, after 40 rounds {free1 / 2**20:.1f} MiB, delta {(free0 - free1) / 2**20:.2f} MiB; "
      f"host RSS {rss0:.0f} -> {rss1:.0f} MiB")
