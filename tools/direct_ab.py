#!/usr/bin/env python3
"""In-process A/B of the single-device Jaccard step at small shapes (round 5): two library calls on the table path (what rounds
1-4 timed), one prepared call on the table path, one prepared call on the one-launch form (csrc/jaccard_direct.h).
us per step = wall time of a long run of steps between two device syncs (the step as a caller sees it: host enqueue or device,
whichever bounds); us on device = one HIP-event pair around the same run.  Every form is checked against the oracle first."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gficf_amd  # noqa: E402
import oracle  # noqa: E402
from gficf_amd import synth  # noqa: E402

SHAPES = [(3000, 15), (10000, 15), (30000, 15), (10000, 30), (20000, 30), (30000, 30), (54000, 30), (100000, 30)]


def run(fn, steps):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6, e0.elapsed_time(e1) / steps * 1e3


def main():
    ops = gficf_amd.HipOps(0)
    ops.set_jaccard_distinct(True)
    print(f"{'N':>7} {'k':>3} | {'two calls, table':>22} | {'one call, table':>22} | {'one call, one launch':>22}   (us per step wall / device)")
    for N, k in SHAPES:
        mat = synth.knn_windowed(N, k, seed=42, perm_seed=43)
        want, _ = oracle.jaccard(mat, nthreads=os.cpu_count() or 1)
        idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
        table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
        out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
        steps = 3000 if N * k < 10**6 else 600

        def two():
            ops.jaccard_ingest(idx, N, k, N, table)
            ops.jaccard_edges(table, N, k, 0, N, out, None)

        res = []
        for name, limit, fn in (("two", 0, two), ("one_table", 0, None), ("one_direct", 10**9, None)):
            ops.set_jaccard_direct_max_edges(limit)
            f = fn or ops.jaccard_prepared(idx, N, k, table, out, None)
            out.fill_(-1.0)
            f()
            ops.sync()
            assert np.array_equal(out.cpu().numpy().T, want), (N, k, name)
            res.append(run(f, steps))
        print(f"{N:>7} {k:>3} | " + " | ".join(f"{w:>10.2f} / {d:>9.2f}" for w, d in res), flush=True)
    ops.set_jaccard_direct_max_edges(-1)


if __name__ == "__main__":
    main()
