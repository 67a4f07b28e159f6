#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: a few launches of each hot-path kernel on the
bench workloads (Jaccard 100 k x 30; GF-ICF 23 k x 54 k), nothing else."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gficf_amd  # noqa: E402
from gficf_amd import synth  # noqa: E402

N, k = int(os.environ.get("PROF_N", 100000)), int(os.environ.get("PROF_K", 30))
reps = int(os.environ.get("PROF_REPS", 5))
ops = gficf_amd.HipOps(0)
if os.environ.get("PROF_JACCARD", "1") == "1":
    mat = synth.knn_windowed(N, k)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    rmat = torch.empty((3, N * k), dtype=torch.float64, device="cuda")
    for _ in range(reps):
        ops.jaccard(idx, N, k, table, rmat, None)
    ops.sync()
if os.environ.get("PROF_GFICF", "1") == "1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    G, Nc = bench.GFICF_G, bench.GFICF_N
    colptr, rowidx, x = bench.synth_counts_device(torch, G, Nc)
    ws = ops.csc_workspace(G, Nc, int(rowidx.numel()))
    for _ in range(reps):
        ops.gficf_csc(G, Nc, colptr, rowidx, x, 0.05, 1.0, None, ws)
    ops.sync()                                   # (synthetic counts hold no stored zeros: nothing to retry)
    if os.environ.get("PROF_META"):
        open(os.environ["PROF_META"], "w").write(str(int(rowidx.numel())))
print("prof_driver done")
