import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gficf_amd, oracle
from gficf_amd import synth
N, k = int(sys.argv[1]), int(sys.argv[2])
mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k, perm_seed=7)
ops = gficf_amd.HipOps(0)
idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
rmat = torch.full((3, N * k), -7.0, dtype=torch.float64, device="cuda")
u = torch.full((N * k,), -7, dtype=torch.int32, device="cuda")
ops.jaccard(idx, N, k, table, rmat, u); ops.sync()
want, wu = oracle.jaccard(mat, nthreads=8)
got = rmat.cpu().numpy().T; gu = u.cpu().numpy()
bad = np.flatnonzero(gu != wu)
print("mismatching edges", bad.size, "of", N * k)
if bad.size:
    cells = np.unique(bad // k)
    print("cells", cells.size, cells[:20], "...", cells[-5:])
    print("hist of cell index /1024:", np.bincount(cells // 1024))
    for r in bad[:10]:
        print(r // k, r % k, "got", gu[r], "want", wu[r], got[r], want[r])
    unwritten = np.flatnonzero(gu == -7)
    print("unwritten", unwritten.size)
rb = np.flatnonzero((got != want).any(axis=1))
print("rmat row mismatches", rb.size)
