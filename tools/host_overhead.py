import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gficf_amd
from gficf_amd import synth
from gficf_amd.dist import JaccardShard
N, k = 100000, 30
ops = gficf_amd.HipOps(0)
mat = synth.knn_windowed(N, k)
idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
for pipe in (False, True):
    sh = JaccardShard(ops, N, k, device="cuda", pipeline=pipe)
    for _ in range(5): sh.step(idx)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): sh.step(idx)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"pipeline={pipe}: host issue {t_issue/K*1e6:.1f} us/step, total {t_all/K*1e6:.1f} us/step")
# graph capture of one non-pipelined step
sh = JaccardShard(ops, N, k, device="cuda", pipeline=False)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): sh.step(idx)
s.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    sh.step(idx)
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter()-t0)/K*1e6:.1f} us/step")
