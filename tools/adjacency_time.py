"""Device-resident time of the graph hand-off at N cells x k neighbours: Jaccard ingest + filtered edges, then the adjacency build alone.
Usage: python tools/adjacency_time.py [N k] ...      ADJ_GRAPH=blobs: the kNN matrix of a real search (30 Gaussian blobs in 50 dimensions,
manhattan — the config-3 stand-in) instead of the windowed synthetic one: in-degrees with a tail, rows of the matrix up to several hundred."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gficf_amd
from gficf_amd import synth


def run(N, k, reps=30):
    ops = gficf_amd.HipOps(0)
    kind = os.environ.get("ADJ_GRAPH", "windowed")
    if kind == "blobs":
        rng = np.random.default_rng(1)
        X = rng.normal(size=(30, 50))[rng.integers(0, 30, N)] * 3.0 + rng.normal(size=(N, 50))
        mat = gficf_amd.find_nn(X, k + 1, metric="manhattan")["idx"][:, 1:].astype(np.int32)
    else:
        mat = synth.knn_windowed(N, k, seed=42, perm_seed=43)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    cap = N * k
    u_ws = torch.zeros(cap, dtype=torch.int16, device="cuda")
    cell_ptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    out3 = torch.zeros((3, cap), dtype=torch.float64, device="cuda")
    ws = torch.zeros(ops.adjacency_workspace_bytes(N, cap), dtype=torch.uint8, device="cuda")
    indptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    indices = torch.zeros(2 * cap, dtype=torch.int32, device="cuda")
    x = torch.zeros(2 * cap, dtype=torch.float64, device="cuda")

    def edges():
        ops.jaccard_ingest(idx, N, k, N, table)
        ops.jaccard_edges_filtered(table, N, k, 0, N, u_ws, cell_ptr, out3)

    def adj():
        ops.adjacency(N, cap, cell_ptr[N:N + 1], out3, ws, indptr, indices, x, grouped_by_source=True)

    def timed(fn):
        for _ in range(3):
            fn()
        ops.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ops.sync()
        return 1e3 * (time.perf_counter() - t0) / reps

    te = timed(edges)
    ta = timed(adj)
    deg = (indptr[1:] - indptr[:-1])
    print(f"N={N} k={k} ({kind}; longest row {int(deg.max())}, {int((deg > 128).sum())} rows above 128 entries): ingest + filtered edges {te:.3f} ms, adjacency {ta:.3f} ms, together {te + ta:.3f} ms ({int(cell_ptr[N])} edges kept, {int(indptr[N])} entries, "
          f"workspace {ws.numel() / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if not a:
        a = [54000, 30, 100000, 50, 1000000, 30]
    for i in range(0, len(a), 2):
        run(a[i], a[i + 1])
