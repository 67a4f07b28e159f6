# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, gficf_amd
from gficf_amd import synth
scen = os.environ.get("SCEN", "A")
ops = gficf_amd.HipOps(0)
N, k = 1_000_000, 30
m = synth.knn_windowed(N, k, seed=42, perm_seed=43)
if scen == "C":
    idx = m.astype(np.int64) - 1
    lab = np.minimum(idx.min(axis=1), np.arange(N)); lab = np.minimum(lab[idx].min(axis=1), lab)
    order = np.argsort(lab, kind="stable")
    new = np.empty(N, np.int64); new[order] = np.arange(N)
    m = (new[idx[order]] + 1).astype(np.int32)
if scen == "B": os.environ["GFICF_JACCARD_ORDER_HOPS"] = "2"
elif scen == "D": os.environ["GFICF_JACCARD_ORDER_HOPS"] = "1"
else: os.environ["GFICF_JACCARD_ORDER"] = "0"
d = torch.from_numpy(np.ascontiguousarray(m.T)).cuda()
table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
ws = ops.order_workspace(N, N)
ops.jaccard_ingest(d, N, k, N, table)
for _ in range(4): ops.jaccard_edges_ordered(table, N, k, 0, N, ws, out)
torch.cuda.synchronize()
