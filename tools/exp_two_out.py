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
tables = [torch.zeros((N, kp), dtype=torch.int32, device="cuda") for _ in range(2)]
outs = [torch.zeros((3, N * k), dtype=torch.float64, device="cuda") for _ in range(2)]
side = torch.cuda.Stream(); mains = [torch.cuda.Stream(), torch.cuda.Stream()]
ready = [torch.cuda.Event(), torch.cuda.Event()]; free = [torch.cuda.Event(), torch.cuda.Event()]
def run(K, two_main):
    for t in range(K):
        p = t & 1
        if t >= 2: side.wait_event(free[p])
        with torch.cuda.stream(side):
            ops.jaccard_ingest(idx, N, k, N, tables[p]); ready[p].record(side)
        m = mains[p] if two_main else mains[0]
        m.wait_event(ready[p])
        with torch.cuda.stream(m):
            ops.jaccard_edges(tables[p], N, k, 0, N, outs[p] if two_main else outs[0]); free[p].record(m)
for two in (False, True, False, True):
    run(10, two); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200, two); torch.cuda.synchronize()
    print("two_main=%s: %.2f us/step" % (two, (time.perf_counter() - t0) / 200 * 1e6))
