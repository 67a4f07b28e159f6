"""Does the Louvain run faster when vertex ids have LOCALITY (neighbours close in memory)?  The same kNN -> Jaccard graph of the config-3
stand-in, once with the cells sorted by blob (ids contiguous per cluster) and once with the ids permuted at random: device-resident
gficf_louvain_device, 10 starts.  Usage: python tools/louvain_locality_probe.py [N k n_start]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

import gficf_amd

N, k, n_start = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (54000, 30, 10)))
rng = np.random.default_rng(1)
lab = np.sort(rng.integers(0, 30, N))
X = rng.normal(size=(30, 50))[lab] * 3.0 + rng.normal(size=(N, 50))
A0 = gficf_amd.jaccard_adjacency(gficf_amd.clustcells_graph(X, k, "manhattan"), N)
perm = rng.permutation(N)
P = sp.csc_matrix((np.ones(N), (perm, np.arange(N))), shape=(N, N))
A1 = (P @ A0 @ P.T).tocsc()
A1.sort_indices()
ops = gficf_amd.HipOps(0)
for name, A in (("ids sorted by cluster", A0), ("ids permuted", A1), ("ids sorted by cluster", A0), ("ids permuted", A1)):
    ptr = torch.from_numpy(A.indptr.astype(np.int64)).cuda()
    idx = torch.from_numpy(A.indices.astype(np.int32)).cuda()
    x = torch.from_numpy(A.data).cuda()
    ws = torch.zeros(ops.louvain_workspace_bytes(N, A.nnz, n_start), dtype=torch.uint8, device="cuda")
    out = torch.zeros(N, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nc, q = ops.louvain(N, ptr, idx, x, 0.8, 10, out, ws, 1, n_start, 180582)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:24s}: {min(ts):.2f} ms min, {sorted(ts)[2]:.2f} median ({nc} clusters, Q {q:.6f})", flush=True)
