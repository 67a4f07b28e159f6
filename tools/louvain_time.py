"""Device-resident time of gficf_louvain_device (graph already in HBM) and of the fused phenograph call, at the shapes VERDICT r05 names:
config 3 (54 000 cells x k = 30, n.start = 10 — the reference's clustcells() defaults, R/clustCells.R:46) and 100 000 x 50 with one start.
Usage: python tools/louvain_time.py [N k n_start [reps]] ...   (no arguments: both shapes).  LT_DEBUG=1: the library's per-iteration trace."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gficf_amd
from oracle import oracle_np


def graph(N, k, seed=1):
    rng = np.random.default_rng(seed)
    C = 30
    X = rng.normal(size=(C, 50))[rng.integers(0, C, N)] * 3.0 + rng.normal(size=(N, 50))
    edges = gficf_amd.clustcells_graph(X, k, "manhattan")
    return X, gficf_amd.jaccard_adjacency(edges, N)


def run(N, k, n_start, reps=5):
    X, A = graph(N, k)
    ops = gficf_amd.HipOps(0)
    dev = "cuda:0"
    ptr = torch.from_numpy(A.indptr.astype(np.int64)).to(dev)
    idx = torch.from_numpy(A.indices.astype(np.int32)).to(dev)
    x = torch.from_numpy(A.data).to(dev)
    ws = torch.zeros(ops.louvain_workspace_bytes(N, A.nnz, n_start), dtype=torch.uint8, device=dev)
    lab = torch.zeros(N, dtype=torch.int32, device=dev)
    ts, labs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nc, q = ops.louvain(N, ptr, idx, x, 0.8, 10, lab, ws, 1, n_start, 180582)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
        labs.append(lab.cpu().numpy().copy())
    same = all(np.array_equal(labs[0], l) for l in labs[1:])
    qn = oracle_np.modularity_np(A, labs[0], 0.8)
    print(f"louvain_device N={N} k={k} nnz={A.nnz} n_start={n_start}: {min(ts):.2f} ms min, {sorted(ts)[len(ts) // 2]:.2f} ms median of {reps} "
          f"({nc} clusters, Q {q:.6f}, numpy Q {qn:.6f}, reproducible {same}, workspace {ws.numel() / 1e6:.0f} MB)", flush=True)
    if reps >= 50:                                   # the spread of many calls: are there stalls?
        st = sorted(ts)
        print("  calls sorted: " + " ".join(f"p{p}={st[min(len(st) - 1, len(st) * p // 100)]:.2f}" for p in (0, 10, 50, 90, 99)) + f" max={st[-1]:.2f} ms; "
              f"calls above 1.5 x the median: {sum(t > 1.5 * st[len(st) // 2] for t in ts)} of {len(ts)}", flush=True)
    tp = []
    if os.environ.get("LT_NO_PHENOGRAPH"):
        return
    for _ in range(reps):
        t0 = time.perf_counter()
        ph = gficf_amd.phenograph(X, k, "manhattan", 0.8, 1, n_start, 10, 180582)
        tp.append(1e3 * (time.perf_counter() - t0))
    print(f"phenograph(X, {k}, manhattan, 0.8, 1, {n_start}, 10, seed) host call: {min(tp):.2f} ms min, {sorted(tp)[len(tp) // 2]:.2f} ms median "
          f"({ph.n_clusters} clusters, Q {ph.modularity:.6f})", flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if os.environ.get("LT_DEBUG"):
        os.environ["GFICF_LOUVAIN_DEBUG"] = "1"
    if not a:
        run(54000, 30, 10)
        run(54000, 30, 1)
        run(100000, 50, 1)
    else:
        while a:
            run(int(a[0]), int(a[1]), int(a[2]), int(a[3]) if len(a) > 3 and not a[3].startswith("-") else 5)
            a = a[4:] if len(a) > 3 else []
