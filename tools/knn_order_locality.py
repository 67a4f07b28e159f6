#!/usr/bin/env python3
"""Rows a block of cells names outside itself in the kNN index of clustered points (the bench's 40 Gaussian blobs, cells in random
order), before and after renumbering the cells in the pruned search's pivot order (gficf_knn_pivot_order_device) — i.e. does
the halo form of the sharded Jaccard build fit real-looking kNN output?  One GPU; blocks of P equal ranks are evaluated on the
full index matrix.  Usage: python tools/knn_order_locality.py [N] [d] [k]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gficf_amd
from gficf_amd.dist import rows_per_rank, shard_bounds

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
ops = gficf_amd.HipOps(0)
rng = np.random.default_rng(11)
centers = rng.normal(scale=6.0, size=(40, d))
lab = rng.integers(0, 40, size=N)
Xh = centers[lab] + rng.normal(size=(N, d)) * rng.uniform(0.5, 2.0, size=(40, 1))[lab]
X = torch.from_numpy(np.ascontiguousarray(Xh.T)).cuda()
pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
ops.knn_prepare(X, N, d, "manhattan", pts)
ws = torch.zeros(ops.knn_workspace_bytes(N, N, k), dtype=torch.uint8, device="cuda")


def search(points):
    idx = torch.zeros((k, N), dtype=torch.int32, device="cuda")
    ops.knn_search(points, N, d, k, "manhattan", 0, N, ws, idx, None)
    return idx[1:].long()                      # (k-1, N) 1-based neighbour ids


order = torch.zeros(N, dtype=torch.int32, device="cuda")
wso = torch.zeros(ops.knn_workspace_bytes(N, N, 1), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.knn_pivot_order(pts, N, d, "manhattan", wso, order)
e1.record()
torch.cuda.synchronize()
print(f"N={N} d={d} k={k - 1}+self; pivot order: {e0.elapsed_time(e1):.3f} ms (first call)")
for name, idx in (("given order", search(pts)), ("pivot order", search(pts.index_select(0, order.long())))):
    for P in (2, 4, 8):
        rpr = rows_per_rank(N, P)
        cap = max(64, min(8192, ((1 << 17) - 1 - rpr) // P)) if (1 << 17) - 1 - rpr >= 64 * P else 1024
        worst_total, worst_owner = 0, 0
        for r in range(P):
            b, e = shard_bounds(N, P, r)
            ids = torch.unique(idx[:, b:e].reshape(-1))
            out = ids[(ids <= b) | (ids > e)]
            per_owner = torch.bincount((out - 1) // rpr, minlength=P)
            worst_total, worst_owner = max(worst_total, int(out.numel())), max(worst_owner, int(per_owner.max()))
        print(f"  {name}: P={P}: rows named outside a block (worst rank) {worst_total} of {N - rpr} remote; most from one owner {worst_owner}"
              f" (request slots per owner {cap}: {'fits' if worst_owner <= cap else 'OVERFLOW'})")
ops.sync()
