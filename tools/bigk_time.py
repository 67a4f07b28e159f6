#!/usr/bin/env python3
"""What the k > 256 path (csrc/jaccard_sorted.h) costs: device-resident ingest + edges per call, next to the fast kernels at
k = 256 and to the oracle port on the host cores (the reference's own algorithm: two sorts + a merge per edge)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gficf_amd  # noqa: E402
import oracle  # noqa: E402

ops = gficf_amd.HipOps(0)
for N, k in ((5000, 256), (5000, 257), (5000, 300), (5000, 513), (20000, 513), (3200, 3000)):
    t_gen = time.perf_counter()
    mat = (np.argsort(np.random.default_rng(N + k).random((N, N), dtype=np.float32), axis=1)[:, :k] + 1).astype(np.int32) if N <= 6000 else \
        np.stack([np.random.default_rng(i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32)
    print(f"# N {N} k {k}: input made in {time.perf_counter() - t_gen:.1f} s", file=sys.stderr, flush=True)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    ops.jaccard(idx, N, k, table, out, None)
    ops.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.jaccard_ingest(idx, N, k, N, table)
    ops.sync()
    t_in = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.jaccard(idx, N, k, table, out, None)
    ops.sync()
    t = (time.perf_counter() - t0) / reps
    cells = min(N, 64)
    t0 = time.perf_counter()
    want, _ = oracle.jaccard_cells(mat, 0, cells, nthreads=os.cpu_count() or 1)
    tc = (time.perf_counter() - t0) * N / cells
    ok = bool(np.array_equal(out[:, :cells * k].cpu().numpy().T, want))
    print(f"N {N:6d} k {k:5d}: {t * 1e3:9.3f} ms per call (ingest {t_in * 1e3:7.3f} ms)  {N * k / t / 1e9:7.3f} G edges/s   "
          f"oracle port on {os.cpu_count()} host threads ~{tc * 1e3:9.1f} ms (from {cells} cells)   bit-exact sample: {ok}", flush=True)
