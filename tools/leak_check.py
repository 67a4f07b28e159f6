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


def once():
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
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for i in range(40):
    once()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"free before {free0 / 2**20:.1f} MiB, after 40 rounds {free1 / 2**20:.1f} MiB, delta {(free0 - free1) / 2**20:.2f} MiB")
