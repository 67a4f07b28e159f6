"""Experiment: consecutive edge kernels on alternating streams with two output buffers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gficf_amd
from gficf_amd import synth
N, k = 100000, 30
ops = gficf_amd.HipOps(0)
mat = synth.knn_windowed(N, k)
idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
kp = ops.kpad(k)
NB = 3
tables = [torch.zeros((N, kp), dtype=torch.int32, device="cuda") for _ in range(NB)]
outs = [torch.zeros((3, N * k), dtype=torch.float64, device="cuda") for _ in range(NB)]
side = torch.cuda.Stream(); mains = [torch.cuda.Stream() for _ in range(NB)]
ready = [torch.cuda.Event() for _ in range(NB)]; free = [torch.cuda.Event() for _ in range(NB)]
def run(K, nm):
    for t in range(K):
        p = t % nm
        if t >= nm: side.wait_event(free[p])
        with torch.cuda.stream(side):
            ops.jaccard_ingest(idx, N, k, N, tables[p]); ready[p].record(side)
        m = mains[p]
        m.wait_event(ready[p])
        with torch.cuda.stream(m):
            ops.jaccard_edges(tables[p], N, k, 0, N, outs[p]); free[p].record(m)
for nm in (1, 2, 3, 2, 3):
    run(12, nm); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(300, nm); torch.cuda.synchronize()
    print("streams=%d: %.2f us/step" % (nm, (time.perf_counter() - t0) / 300 * 1e6))
