"""x / S_c as a per-entry f64 division (GFICF_SCALE_TRUE_DIV=1) or as a multiplication by the cell's reciprocal (the default),
A/B inside one process on the same buffers, alternating."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, gficf_amd
ops = gficf_amd.HipOps(0)
for name, G, N in (("c3", 23000, 54000), ("c4", 30000, 100000)):
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    ws = ops.csc_workspace(G, N, int(rowidx.numel()))
    run = lambda: ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
    res = {0: [], 1: []}
    for rnd in range(4):
        for sw in (0, 1):
            if sw: os.environ["GFICF_SCALE_TRUE_DIV"] = "1"
            else: os.environ.pop("GFICF_SCALE_TRUE_DIV", None)
            for _ in range(2): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize(); res[sw].append((time.perf_counter() - t0) / 10)
    for sw in (0, 1):
        b = min(res[sw]); print(name, "true division" if sw else "reciprocal   ", "best %.4f ms  all %s  frac %.3f" % (b * 1e3, ["%.4f" % (t * 1e3) for t in res[sw]], 24 * rowidx.numel() / b / 8e12), flush=True)
    del colptr, rowidx, x, ws; torch.cuda.empty_cache()
