#!/usr/bin/env python3
"""Times the GF-ICF scaling pass alone (pointerB / pointerE form, config 3 shape) for the library GFICF_HIP_LIB names.
Usage: GFICF_HIP_LIB=... python tools/lab/scale_probe.py <label>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import gficf_amd

label = sys.argv[1] if len(sys.argv) > 1 else "?"
G, N = 23000, 54000
ops = gficf_amd.HipOps(0)
colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
nnz = int(rowidx.numel())
ws = ops.csc_workspace(G, N, nnz)
ops.gficf_csc_be(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)       # gene table, kept count
torch.cuda.synchronize()
run = lambda: ops.csc_scale_be(G, N, colptr, rowidx, x, ws["genes"], ws["gkept"], ws["out_end"], ws["out_rowidx"], ws["out_x"])
whole = lambda: ops.gficf_csc_be(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
s = torch.cuda.current_stream()
def timed(f, reps):
    for _ in range(5):
        f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            f()
        e1.record(s)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
for _ in range(200):
    run()
t_scale, t_whole = timed(run, 20), timed(whole, 20)
ops.sync()
kept = int((ws["out_end"] - colptr[:-1]).sum())
print("%-10s scale %7.1f us   whole pass %7.1f us   nnz %d kept %d   scale traffic %.2f TB/s" %
      (label, t_scale * 1e3, t_whole * 1e3, nnz, kept, (12 * nnz + 12 * kept) / (t_scale * 1e-3) / 1e12), flush=True)
