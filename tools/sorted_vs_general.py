#!/usr/bin/env python3
"""For 56 < k <= 256: the general hash-set edge kernel (k_jaccard_edges) against the sorted-row path (csrc/jaccard_sorted.h: sort at
ingest + k binary searches per edge), device-resident ingest + edges per call.  The switch (GFICF_JACCARD_SORTED_FROM) is read once
per process, so every (setting, shape) is timed in a child process.  Usage: python tools/sorted_vs_general.py            (the table)
                                                                          python tools/sorted_vs_general.py child N k   (one cell)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(N, k):
    import numpy as np
    import torch

    import gficf_amd
    import oracle
    from gficf_amd import synth

    ops = gficf_amd.HipOps(0)
    ops.set_jaccard_distinct(True)
    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k, perm_seed=3)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    run = ops.jaccard_prepared(idx, N, k, table, out, None)
    run()
    try:
        ops.sync()
    except gficf_amd.GficfError as e:             # a row's ids overflowed the hash set: its own status since round 6 — what a device caller does: the option off
        if e.status != "GFICF_ERR_SET_OVERFLOW":
            raise
        ops.set_jaccard_distinct(False)
        run = ops.jaccard_prepared(idx, N, k, table, out, None)
        run()
        ops.sync()
    cells = min(N, 256)
    want, _ = oracle.jaccard_cells(mat, 0, cells, nthreads=os.cpu_count() or 1)
    ok = bool(np.array_equal(out[:, :cells * k].cpu().numpy().T, want))
    reps = max(3, min(200, int(2e8 / (N * k * k))))
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"{t * 1e3:.4f} {ops.row_words(N, k)} {ok}")


def main():
    print(f"{'N':>7} {'k':>4} | {'general kernel, ms':>19} {'row words':>9} | {'sorted path, ms':>16} {'row words':>9} | sorted / general   (both bit-exact on a 256-cell oracle sample)")
    for N in (5000, 50000, 200000):
        for k in (57, 64, 65, 80, 100, 128, 129, 160, 200, 256):
            res = []
            for frm in ("257", "57"):
                env = dict(os.environ, GFICF_JACCARD_SORTED_FROM=frm)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(N), str(k)], capture_output=True, text=True, env=env, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.strip()]
                if r.returncode != 0 or not line:
                    res.append((float("nan"), -1, f"FAILED {r.stderr[-200:]}"))
                    continue
                t, rw, ok = line[-1].split()
                res.append((float(t), int(rw), ok))
            (tg, rg, og), (ts, rs, os_) = res
            print(f"{N:>7} {k:>4} | {tg:>19.3f} {rg:>9} | {ts:>16.3f} {rs:>9} | {ts / tg:6.2f} x   {og} {os_}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
