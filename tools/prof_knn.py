#!/usr/bin/env python3
"""Small driver for rocprofv3 passes over the exact kNN search (100 k x 50, k+1 = 31, manhattan by default)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gficf_amd  # noqa: E402

N, d, k = int(os.environ.get("PROF_N", 100000)), int(os.environ.get("PROF_D", 50)), int(os.environ.get("PROF_K", 31))
metric = os.environ.get("PROF_METRIC", "manhattan")
reps = int(os.environ.get("PROF_REPS", 2))
ops = gficf_amd.HipOps(0)
X = torch.from_numpy(np.random.default_rng(1).normal(size=(d, N))).cuda()
pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
ws = torch.zeros(ops.knn_workspace_bytes(N, N, k), dtype=torch.uint8, device="cuda")
idx = torch.zeros((k, N), dtype=torch.int32, device="cuda")
for _ in range(reps):
    ops.knn_prepare(X, N, d, metric, pts)
    ops.knn_search(pts, N, d, k, metric, 0, N, ws, idx, None)
ops.sync()
print("prof_knn done")
