# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
"""Would a locality order of the cells pay at config 5 (1 M x 30, scrambled ids, 128 B rows)?  The matrix is relabelled on the
host by t rounds of min-label propagation (cells sorted by the smallest label within t hops) and the product kernels are timed
on it: with one line per row the numbering only matters through the order in which rows are touched."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, gficf_amd
from gficf_amd import synth
ops = gficf_amd.HipOps(0)
N, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 30
m = synth.knn_windowed(N, k, seed=42, perm_seed=43)
idx = m.astype(np.int64) - 1
def relabel(order):
    new = np.empty(N, np.int64); new[order] = np.arange(N)
    return (new[idx[order]] + 1).astype(np.int32)
def run(mat, name):
    d = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    rw = ops.row_words(N, k)
    table = torch.zeros((N, rw), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    for sw in (0, 1):
        if sw: os.environ["GFICF_JACCARD_XCD"] = "1"
        else: os.environ.pop("GFICF_JACCARD_XCD", None)
        ops.jaccard_ingest(d, N, k, N, table)
        for _ in range(2): ops.jaccard_edges(table, N, k, 0, N, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.jaccard_edges(table, N, k, 0, N, out)
        e1.record(); torch.cuda.synchronize()
        print("%-28s xcd %d  edges %.1f us" % (name, sw, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
run(m, "scrambled")
run(synth.knn_windowed(N, k, seed=42, perm_seed=None), "in order")
lab = np.minimum(idx.min(axis=1), np.arange(N))
for t in range(1, 4):
    order = np.argsort(lab, kind="stable")
    run(relabel(order), "min label within %d hop(s)" % t)
    lab = np.minimum(lab[idx].min(axis=1), lab)
