"""GF-ICF pass at the two small shapes (configs 1 and 2), for a rocprofv3 kernel trace: where do 0.04 / 0.13 ms go?
Usage: rocprofv3 --kernel-trace --stats -d DIR -o small -- python3 tools/lab/gficf_small_trace.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, gficf_amd
ops = gficf_amd.HipOps(0)
for name, G, N in (("c1", 5000, 3000), ("c2", 20000, 10000)):
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    ws = ops.csc_workspace(G, N, int(rowidx.numel()))
    run = lambda: ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(name, "nnz", int(rowidx.numel()), "%.4f ms" % (t * 1e3), "frac %.3f" % (24 * rowidx.numel() / t / 8e12), flush=True)
