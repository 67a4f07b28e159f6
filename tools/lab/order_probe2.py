# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
"""Placement or walk?  1 M x 30: rows placed by a locality order (host relabel, 2 hops) or not, walked in order, in a
scrambled order (GFICF_JACCARD_ORDER_SCRAMBLE) or in the library's label order."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, gficf_amd
from gficf_amd import synth
ops = gficf_amd.HipOps(0)
N, k = 1_000_000, 30
m = synth.knn_windowed(N, k, seed=42, perm_seed=43)
idx = m.astype(np.int64) - 1
lab = np.minimum(idx.min(axis=1), np.arange(N)); lab = np.minimum(lab[idx].min(axis=1), lab)
order = np.argsort(lab, kind="stable")
new = np.empty(N, np.int64); new[order] = np.arange(N)
m2 = (new[idx[order]] + 1).astype(np.int32)
def run(mat, name, env):
    for kk in ("GFICF_JACCARD_ORDER", "GFICF_JACCARD_ORDER_SCRAMBLE", "GFICF_JACCARD_ORDER_HOPS"): os.environ.pop(kk, None)
    os.environ.update(env)
    d = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    ws = ops.order_workspace(N, N)
    ops.jaccard_ingest(d, N, k, N, table)
    f = lambda: ops.jaccard_edges_ordered(table, N, k, 0, N, ws, out)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("%-64s %.1f us" % (name, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
run(m, "rows scattered, plain walk (order off)", {"GFICF_JACCARD_ORDER": "0"})
run(m, "rows scattered, label walk (1 hop) [incl. ordering]", {})
run(m, "rows scattered, label walk (2 hops) [incl. ordering]", {"GFICF_JACCARD_ORDER_HOPS": "2"})
run(m2, "rows placed by locality, plain walk (order off)", {"GFICF_JACCARD_ORDER": "0"})
run(m2, "rows placed by locality, label walk (1 hop) [incl. ordering]", {})
run(m2, "rows placed by locality, scrambled walk [incl. ordering]", {"GFICF_JACCARD_ORDER_SCRAMBLE": "1"})
run(m, "rows scattered, scrambled walk [incl. ordering]", {"GFICF_JACCARD_ORDER_SCRAMBLE": "1"})
