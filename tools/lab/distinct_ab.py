"""Headline step (8 data sets of 100 k x 30, ingest + edges each, one stream, in order) with and without the ingest's duplicate scan,
alternating inside one process: 6 rounds of 20 steps each."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, gficf_amd
from gficf_amd import synth
from gficf_amd.dist import JaccardShard
N, k, B = 100_000, int(sys.argv[1]) if len(sys.argv) > 1 else 30, 8 if len(sys.argv) < 3 else int(sys.argv[2])
ops = gficf_amd.HipOps(0)
idx = [torch.from_numpy(np.ascontiguousarray(synth.knn_windowed(N, k, seed=42 + 7 * d, perm_seed=43 + 7 * d).T)).cuda() for d in range(B)]
sh = [JaccardShard(ops, N, k, device="cuda", with_u=False) for _ in range(B)]
def step():
    for d in range(B): sh[d].step(idx[d])
for _ in range(10): step()
torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(6):
    for mode in (1, 0):
        ops.set_jaccard_distinct(bool(mode))
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res[mode].append(dt * 1e3)
ops.set_jaccard_distinct(False); ops.sync()
for mode in (1, 0):
    v = res[mode]
    print("k=%d  %-22s ms/step: %s   median %.4f  -> %.2f G edges/s" % (k, "no scan in the ingest" if mode else "ingest scans", " ".join("%.4f" % x for x in v), sorted(v)[len(v)//2], N * k * B / sorted(v)[len(v)//2] / 1e6))
