"""Times t(gficf) (gficf_csc_transpose_device) on the bench's GF-ICF output and checks it against a stable torch sort.
Usage: python tools/transpose_bench.py [G N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import gficf_amd

G, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (23000, 54000)
ops = gficf_amd.HipOps(0)
colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
res = ops.gficf_csc(G, N, colptr, rowidx, x, auto_exact=True)
ops.sync()
gk, nk = int(res["gkept"][0]), int(res["out_colptr"][N])
cp, ri, xv = res["out_colptr"], res["out_rowidx"][:nk], res["out_x"][:nk]
if os.environ.get("TR_RAW"):                      # the unfiltered count matrix instead of the GF-ICF output
    gk, nk, cp, ri, xv = G, int(rowidx.numel()), colptr, rowidx, x
ws = torch.zeros(ops.csc_transpose_workspace_bytes(gk, N), dtype=torch.uint8, device="cuda")
ptr = torch.zeros(gk + 1, dtype=torch.int64, device="cuda")
idx = torch.zeros(nk, dtype=torch.int32, device="cuda")
val = torch.zeros(nk, dtype=torch.float64, device="cuda")
run = lambda: ops.csc_transpose(gk, N, cp, ri, xv, ptr, idx, val, ws)
run(); ops.sync()
order = torch.sort(ri.long(), stable=True)[1]
cell = torch.repeat_interleave(torch.arange(N, device="cuda", dtype=torch.int32), cp[1:] - cp[:-1])
ok = bool(torch.equal(idx, cell[order]) and torch.equal(val, xv[order])
          and torch.equal(ptr[1:], torch.cumsum(torch.bincount(ri.long(), minlength=gk), 0)))
ms = bench.time_kernel_ms(torch, run, 20)
print(f"G={gk} (of {G}) N={N} nnz={nk}: {ms:.3f} ms/transpose, {28 * nk / ms / 1e6:.1f} GB/s algorithmic (28 B/entry), "
      f"{N / ms / 1e3:.2f} M cells/s, ok={ok}")
