import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench, gficf_amd
ops = gficf_amd.HipOps(0)
for name, G, N in (("c1", 5000, 3000), ("c2", 20000, 10000), ("c3", 23000, 54000)):
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    ws = ops.csc_workspace(G, N, int(rowidx.numel()))
    res = []
    for ratio in ("0", "1", "2", "3", "4", "6", "8"):
        os.environ["GFICF_COUNT_FLUSH_RATIO"] = ratio
        run = lambda: ops.gficf_csc_be(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
        for _ in range(5): run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20): run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        res.append("%s:%.1f" % (ratio, best * 1e6))
    print(name, "BE pass us by flush ratio:", " ".join(res), flush=True)
