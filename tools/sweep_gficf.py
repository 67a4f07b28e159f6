#!/usr/bin/env python3
"""GF-ICF pass over the shapes of BASELINE.json's configs (synthetic counts generated on the device, device-resident pass,
5 % gene filter).  Usage: python tools/sweep_gficf.py [out.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import gficf_amd

ops = gficf_amd.HipOps(0)
lines = []
for name, G, N, kw in [("config 1: 3 k cells x 5 k genes", 5000, 3000, {}),
                       ("config 2: 10 k cells x 20 k genes", 20000, 10000, {}),
                       ("config 3: 54 k cells x 23 k genes", 23000, 54000, {}),
                       ("config 4: 100 k cells x 30 k genes", 30000, 100000, {}),
                       ("config 5: 1 M cells x 30 k genes (at most 2147 entries per cell)", 30000, 1000000,
                        dict(seed=5, max_per_cell=2147, chunk_cells=125_000))]:
    if os.environ.get("GFICF_SWEEP_CONFIGS") and name.split(":")[0].split()[-1] not in os.environ["GFICF_SWEEP_CONFIGS"].split(","):
        continue                                                    # (a kernel trace of one shape: GFICF_SWEEP_CONFIGS=1)
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N, **kw)
    nnz = int(rowidx.numel())
    ws = ops.csc_workspace(G, N, nnz)
    # one library call per pass with its arguments converted once (HipOps.gficf_csc_prepared); enough passes per bracket that the two
    # device syncs around them do not count (round 4 took 5 passes per bracket with eighteen ctypes conversions each: config 1's
    # "0.046 ms" was mostly that)
    run, _ = ops.gficf_csc_prepared(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
    reps = 400 if nnz < 5_000_000 else 100 if nnz < 50_000_000 else 10
    for _ in range(max(2, reps // 10)):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    ops.sync()
    kept, gk = int(ws["out_colptr"][N]), int(ws["gkept"][0])
    # the same pass with the result in the pointerB / pointerE form (no kept-count pass, no scan: three launches)
    run_be, _ = ops.gficf_csc_prepared(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws, form="begin_end")
    for _ in range(max(2, reps // 10)):
        run_be()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run_be()
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / reps
    ops.sync()
    lines.append("%-68s nnz %11d  kept genes %6d  kept entries %11d  %9.3f ms  %7.1f M cells/s  %5.2f TB/s algorithmic (24 B/entry) = %.3f of 8 TB/s  |  begin/end form %9.3f ms = %.3f"
                 % (name, nnz, gk, kept, t * 1e3, N / t / 1e6, 24 * nnz / t / 1e12, 24 * nnz / t / 8e12, tb * 1e3, 24 * nnz / tb / 8e12))
    print(lines[-1], flush=True)
    del colptr, rowidx, x, ws, run, run_be
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
