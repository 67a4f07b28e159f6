#!/usr/bin/env python3
"""A/B inside ONE process, on the same buffers: the GF-ICF pass with the scaling kernel's cells dealt by entries + LDS counter
(default) and dealt round-robin (GFICF_SCALE_STATIC_CELLS=1).  (Separate processes differ by up to 8 % on their own: the pass
time is bimodal from process to process — see DESIGN.md §4.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import gficf_amd

ops = gficf_amd.HipOps(0)
G, N = bench.GFICF_G, bench.GFICF_N
colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
ws = ops.csc_workspace(G, N, int(rowidx.numel()))
run = lambda: ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)


def timed(reps=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for r in range(4):
    os.environ.pop("GFICF_SCALE_STATIC_CELLS", None)
    a = timed()
    os.environ["GFICF_SCALE_STATIC_CELLS"] = "1"
    b = timed()
    print("round %d: by entries + counter %.4f ms   round-robin %.4f ms" % (r, a, b), flush=True)
