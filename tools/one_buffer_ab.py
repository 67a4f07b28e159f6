#!/usr/bin/env python3
"""Does a caller that reuses ONE table / ONE output matrix for every data set pay for it?  Round 5's N = 1 `pipelined` leg (one
shard, eight inputs in turn) measured 58 us per data set against `value`'s 47 (eight shards in turn) with the same kernels.
The step at 100 k x 30 with: 8 inputs x 8 tables x 8 outputs (the bench's `value`), 8 x 1 x 1, 1 x 1 x 1, 8 x 8 x 1, 8 x 1 x 8."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gficf_amd  # noqa: E402
from gficf_amd import synth  # noqa: E402

N, k, B = 100_000, 30, 8
ops = gficf_amd.HipOps(0)
ops.set_jaccard_distinct(True)
idx = [torch.from_numpy(np.ascontiguousarray(synth.knn_windowed(N, k, seed=42 + 7 * d, perm_seed=43 + 7 * d).T)).cuda() for d in range(B)]
tabs = [torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda") for _ in range(B)]
outs = [torch.zeros((3, N * k), dtype=torch.float64, device="cuda") for _ in range(B)]
for ni, nt, no in ((8, 8, 8), (8, 1, 1), (1, 1, 1), (8, 8, 1), (8, 1, 8), (8, 8, 8)):
    runs = [ops.jaccard_prepared(idx[d % ni], N, k, tabs[d % nt], outs[d % no], None) for d in range(B)]
    for _ in range(200):
        for r in runs:
            r()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        for r in runs:
            r()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 800
    print(f"inputs {ni}  tables {nt}  outputs {no}: {dt * 1e6:7.2f} us per data set  ({N * k / dt / 1e9:.1f} G edges/s)", flush=True)
ops.sync()
