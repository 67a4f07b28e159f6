#!/usr/bin/env python3
"""Times the exact kNN search (device-resident) for a few shapes; prints pair-distances/s and the
f32 VALU rate it corresponds to.  Usage: python tools/knn_bench.py [N d k metric]..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gficf_amd  # noqa: E402

ops = gficf_amd.HipOps(0)
cases = [(10000, 50, 31, "manhattan"), (54000, 50, 31, "manhattan"), (100000, 50, 31, "manhattan"),
         (100000, 50, 31, "euclidean"), (100000, 50, 31, "cosine"), (100000, 2, 31, "manhattan"), (100000, 50, 51, "manhattan")]
if len(sys.argv) > 4:
    a = sys.argv[1:]
    cases = [(int(a[i]), int(a[i + 1]), int(a[i + 2]), a[i + 3]) for i in range(0, len(a) - 3, 4)]
for N, d, k, metric in cases:
    rng = np.random.default_rng(1)
    if os.environ.get("KNN_DATA", "normal") == "blobs":        # clustered, like cells in PCA space (bench.py's recipe)
        centers = rng.normal(scale=float(os.environ.get("KNN_CENTER_SCALE", "6.0")), size=(40, d))
        lab = rng.integers(0, 40, size=N)
        X = torch.from_numpy(np.ascontiguousarray((centers[lab] + rng.normal(size=(N, d)) * rng.uniform(0.5, 2.0, size=(40, 1))[lab]).T)).cuda()
    else:
        X = torch.from_numpy(rng.normal(size=(d, N))).cuda()
    pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
    ws = torch.zeros(ops.knn_workspace_bytes(N, N, k), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((k, N), dtype=torch.int32, device="cuda")
    dist = torch.zeros((k, N), dtype=torch.float32, device="cuda")
    ops.knn_prepare(X, N, d, metric, pts)
    ops.knn_search(pts, N, d, k, metric, 0, N, ws, idx, dist)
    ops.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        ops.knn_prepare(X, N, d, metric, pts)
        ops.knn_search(pts, N, d, k, metric, 0, N, ws, idx, dist)
    e1.record()
    ops.sync()
    ms = e0.elapsed_time(e1) / reps
    ops_per = {"manhattan": 2, "euclidean": 2, "cosine": 1}[metric]
    print(f"N={N} d={d} k={k} {metric}: {ms:.2f} ms  {N / ms * 1e3:.3g} cells/s  {N * N / ms / 1e6:.1f} G pairs/s  "
          f"{N * N * d * ops_per / ms / 1e9:.1f} T lane-ops/s", flush=True)
