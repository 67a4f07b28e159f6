import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, gficf_amd
ops = gficf_amd.HipOps(0)
for name, G, N in (("c3", 23000, 54000), ("c4", 30000, 100000)):
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    ws = ops.csc_workspace(G, N, int(rowidx.numel()))
    run = lambda: ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
    best = 1e9
    for rnd in range(3):
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    print(name, "nnz", int(rowidx.numel()), "%.4f ms" % (best * 1e3), "frac %.3f" % (24 * rowidx.numel() / best / 8e12), flush=True)
    del colptr, rowidx, x, ws; torch.cuda.empty_cache()
