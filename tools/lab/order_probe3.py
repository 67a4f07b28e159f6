# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
"""Is it the scattered 24 B/edge stores?  The counts-only output (2 B per edge) on the plain and on the label walk, 1 M x 30."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, gficf_amd
from gficf_amd import synth
ops = gficf_amd.HipOps(0)
N, k = 1_000_000, 30
m = synth.knn_windowed(N, k, seed=42, perm_seed=43)
d = torch.from_numpy(np.ascontiguousarray(m.T)).cuda()
table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
ops.jaccard_ingest(d, N, k, N, table)
u_ws = torch.zeros(N * k + 64, dtype=torch.int16, device="cuda")
cell_ptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
def run(name, env):
    for kk in ("GFICF_JACCARD_ORDER", "GFICF_JACCARD_ORDER_FILTERED", "GFICF_JACCARD_ORDER_HOPS"): os.environ.pop(kk, None)
    os.environ.update(env)
    f = lambda: ops.jaccard_edges_filtered(table, N, k, 0, N, u_ws, cell_ptr, out)
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("%-60s %.1f us   kept edges %d" % (name, e0.elapsed_time(e1) / 5 * 1e3, int(cell_ptr[N])), flush=True)
run("filtered edge build, plain walk", {})
run("filtered edge build, label walk 1 hop", {"GFICF_JACCARD_ORDER_FILTERED": "1"})
run("filtered edge build, label walk 2 hops", {"GFICF_JACCARD_ORDER_FILTERED": "1", "GFICF_JACCARD_ORDER_HOPS": "2"})
