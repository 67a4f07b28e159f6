#!/usr/bin/env python3
"""bench.py — Jaccard edges/s (headline) and GF-ICF cells/s on MI355X.

A "step" is one pass of the Jaccard hot path over one batch of synthetic input that is already resident in HBM: for
every data set of the batch, the column-major int32 kNN index block of this rank's cells (what
`uwot:::find_nn(...)$idx[,-1]` hands to the reference, R/clustCells.R:63-65)
  -> ingest (transpose / validate into the row-major table)
  -> [N > 1: the exchange over xGMI — an RCCL all-gather of the table rows, or, when the blocks name few rows outside
      themselves (`--exchange auto` decides on data set 0 before anything is timed), the halo form on local ids: two
      equal-split all-to-alls of request slots and index rows, gficf_amd/dist.py JaccardHaloShard]
  -> edge kernel -> this rank's rows of the reference's (N*k) x 3 double matrix,
all on ONE stream, in order: every data set pays its own ingest + edge kernel (+ exchange), nothing of one data set
overlaps another (no software pipelining in `value`; `--pipeline` keeps the overlapped mode as a separate figure).

Workloads (`--config`, named in config.workload):
  north_star (default; what BASELINE.json's metric is quoted on): 100 000 cells x k = 30 per GPU, weak scaling — every
             rank owns 100 000 cells of a (100 000 x n_gpus)-cell data set; a batch of 8 independent data sets per step
             (different seeds), so that K = 20 steps time >= 10 ms and no launch re-reads what the one before left in
             the caches;
  c1 .. c5   the Jaccard halves of BASELINE configs 1 .. 5: ONE data set (3 000 x k = 15, 10 000 x 30, 54 000 x 30, 100 000 x 50,
             1 000 000 x 30) split over the ranks by cell block — strong scaling (c1 - c3 are single-GPU configs: a few
             microseconds of work per data set, bound by launches).

N > 1: ONE line carries the whole scaling answer — `value` (permuted ids, in order, exchange chosen from the data), `pipelined`
(the same, overlapped), `spatial_ids` (ids with locality: the halo form, in order and overlapped, rows named outside the block),
`peer` (the single-process form of the C ABI: every device pulls the other devices' table slices with hipMemcpyPeerAsync),
`single_gpu_step` (rank 0 timing the N = 1 step in the same process, after the N-rank region) and `efficiency` (each of
the above divided by n_gpus x the single-GPU rate).  N = 1: `value_from_idle` next to `value` (the K steps timed right after
start-up, without the clock-settling pre-warm: what one call from a cold session sees).

Launch: `python bench.py` (1 GPU); for N > 1 either
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`
or plain `python bench.py --gpus N`, which starts its own N ranks (gficf_amd/launch.py) before anything touches a GPU.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CELLS_PER_GPU = 100_000
K = 30
BATCH = 8                        # independent data sets per step (north_star)
CONFIGS = {                      # the Jaccard halves of BASELINE.json's configs: cells_total, k (ONE data set, split over the ranks: strong scaling)
    "c1": (3_000, 15),
    "c2": (10_000, 30),
    "c3": (54_000, 30),
    "c4": (100_000, 50),
    "c5": (1_000_000, 30),
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md "HBM3E peak BW")
HBM_COPY_GBS = 6300.0            # achievable streaming copy rate on the same part (MI355X_MICROARCH.md; SURVEY.md 8d asks for both)
JACCARD_BYTES_PER_EDGE = 28      # 4 B index entry read once + 24 B reference output row
GFICF_BYTES_PER_NNZ = 24         # 4 (count pass rowidx) + 12 (scale pass rowidx+x) + 8 (write x)  
GFICF_G, GFICF_N = 23_000, 54_000  # BASELINE config 3 shape (Tabula-Muris-sized synthetic stand-in)
KNN_N, KNN_D, KNN_K = 100_000, 50, 31  # north-star point: 100 k cells, 50 PCA components, k = 30 + the cell itself


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["north_star", "c1", "c2", "c3", "c4", "c5"], default="north_star")
    ap.add_argument("--cells-per-gpu", type=int, default=CELLS_PER_GPU, help="north_star only")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="independent data sets per step (default 8 for north_star, 1 for c4 / c5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gficf", action="store_true")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline line only (no stress / host ABI / GF-ICF / kNN sub-objects)")
    ap.add_argument("--pipeline", action="store_true", help="also report the software-pipelined mode (steps overlapped on side streams)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 code path on a one-GPU box: every rank uses device 0 and the gloo backend (numbers are meaningless)")
    ap.add_argument("--ids", choices=["permuted", "spatial"], default="permuted",
                    help="id model of `value`: permuted: ids relabelled by a random permutation (what Annoy output looks like; the default); "
                         "spatial: cells numbered in their spatial order (what the device kNN search's pivot order gives).  "
                         "N > 1 lines carry the other model too, as an object")
    ap.add_argument("--legs", default=None,
                    help="comma-separated legs to run besides `value` (one GPU: pipelined, cpu_baseline, stress, host_abi, gficf, knn; "
                         "N > 1: pipelined, other_ids, single_gpu_step, chain, peer, gficf); default: all")
    ap.add_argument("--budget-s", type=float, default=420.0,
                    help="wall budget of the whole run in seconds: a leg that would start with less than its reserve left is skipped and "
                         "named in `skipped_legs`; child-process and launcher limits are derived from it")
    ap.add_argument("--no-host-gficf", action="store_true", help="skip the GF-ICF host-ABI figure (plan + finish over PCIe) of the gficf leg")
    ap.add_argument("--no-peer", action="store_true", help="N > 1: skip the single-process peer-copy leg (`peer` object)")
    ap.add_argument("--no-chain", action="store_true", help="N > 1: skip the kNN -> Jaccard chain leg (`chain` object)")
    ap.add_argument("--pre-warm-ms", type=float, default=200.0,
                    help="run the step untimed for this long BEFORE the W warm-up steps, so that the K timed steps see settled clocks "
                         "(from idle the first ~20 ms of work run 10-12 %% slower: tools/lab/distinct_ab.py); 0 = off")
    ap.add_argument("--scan-dups", action="store_true",
                    help="N = 1: the ingest looks for duplicate ids inside a row itself (all-pairs scan) instead of leaving it to the edge kernel's "
                         "hash-set build + a deferred error (gficf_ctx_set_jaccard_distinct, what the host entries do)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed profiles/pmc_traffic.json instead of two rocprofv3 --pmc passes of this run (N = 1)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # internal: the process rocprofv3 wraps
    ap.add_argument("--peer-child", action="store_true", help=argparse.SUPPRESS)      # internal: the single-process peer-copy leg of an N > 1 line
    ap.add_argument("--exchange", choices=["auto", "allgather", "halo", "halo_generic"], default="auto",
                    help="N > 1: all-gather of every rank's table rows; halo: sub-problems in local ids, only the rows a block names travel "
                         "(fixed-capacity request slots, gficf_amd.dist.JaccardHaloShard); halo_generic: round 2's torch-side halo on global ids; "
                         "auto (default): halo when the warm-up step of data set 0 fits the request slots on every rank, all-gather otherwise")
    return ap.parse_args()


def time_kernel_ms(torch, fn, iters):
    """Average duration of `fn`'s launches with HIP events on torch's current stream
    (gficf_amd binds its context to that stream before every launch)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def synth_counts_device(torch, G, N, seed=7, median_frac=0.07, sigma=0.5, zipf_s=0.9, max_per_cell=None, chunk_cells=None, density="survey"):
    """Device-side generator of the BASELINE-shaped synthetic CSC count matrix (same recipe as
    gficf_amd.synth.counts_csc, torch RNG instead of splitmix64 so that ~1e8 draws take seconds).

    density="survey" (default): SURVEY.md §8d's density — the number of STORED entries of a cell follows the clipped
      lognormal (median 0.07 G, sigma 0.5, clipped to [0.25, 1.5] x the median; `max_per_cell` caps it: 2 147 for config 5 so
      that nnz < 2^31), the genes are drawn by Zipf-like popularity WITHOUT replacement: a cell over-draws by the expected
      collision rate of its target (distinct(m) = sum_g 1 - (1 - p_g)^m, inverted by interpolation), duplicates are
      collapsed and a random subset of the surplus is dropped.  Config 3: 88 M entries (survey: ~87 M).
    density="light": rounds 1-2's generator — the lognormal number is the number of DRAWS, duplicate (cell, gene) draws
      collapse (config 3: 59.8 M entries; kept for comparisons with profiles/r01*, r02*).
    chunk_cells generates the cells in blocks of that many (bounds the generator's temporaries at config 5's size)."""
    if chunk_cells is not None and N > chunk_cells:
        parts = [synth_counts_device(torch, G, min(chunk_cells, N - c0), seed + 1000003 * (i + 1), median_frac, sigma, zipf_s, max_per_cell, None, density)
                 for i, c0 in enumerate(range(0, N, chunk_cells))]
        colptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
        off, c0 = 0, 0
        for cp, ri, _ in parts:
            n = cp.numel() - 1
            colptr[c0 + 1:c0 + n + 1] = cp[1:] + off
            off, c0 = off + int(ri.numel()), c0 + n
        rows, xs = [p[1] for p in parts], [p[2] for p in parts]
        del parts
        rowidx = torch.cat(rows)
        del rows
        x = torch.cat(xs)
        return colptr, rowidx, x
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    med = median_frac * G
    n_target = torch.round(med * torch.exp(sigma * torch.randn(N, generator=g, device="cuda", dtype=torch.float64)))
    pop = 1.0 / torch.arange(1, G + 1, device="cuda", dtype=torch.float64).pow(zipf_s)
    p = pop / pop.sum()
    cdf = torch.cumsum(p, 0)
    cdf = cdf / cdf[-1]
    if density == "light":
        n_draw = torch.clamp(n_target, 1, G).long()
        if max_per_cell is not None:
            n_draw = torch.clamp(n_draw, max=int(max_per_cell))
        n_keep = None
    else:
        n_keep = torch.clamp(n_target, max(1.0, 0.25 * med), min(float(G), 1.5 * med))
        if max_per_cell is not None:
            n_keep = torch.clamp(n_keep, max=float(max_per_cell))
        n_keep = n_keep.long().clamp_(min=1)
        # draws m(n) such that the expected number of distinct genes is 1.03 n (the surplus is trimmed below)
        m_grid = torch.logspace(0, np.log10(40.0 * G), 512, device="cuda", dtype=torch.float64)
        d_grid = (1.0 - torch.exp(m_grid[:, None] * torch.log1p(-p)[None, :])).sum(1)            # distinct(m), increasing
        want = (1.03 * n_keep.double()).clamp_(max=float(d_grid[-1]) * 0.999)
        hi = torch.searchsorted(d_grid, want).clamp_(1, 511)
        t = (want - d_grid[hi - 1]) / (d_grid[hi] - d_grid[hi - 1])
        n_draw = torch.ceil(m_grid[hi - 1] + t * (m_grid[hi] - m_grid[hi - 1])).long().clamp_(min=1)
    cell_of = torch.repeat_interleave(torch.arange(N, device="cuda"), n_draw)
    gene = torch.searchsorted(cdf, torch.rand(cell_of.numel(), generator=g, device="cuda", dtype=torch.float64)).clamp_(max=G - 1)
    key = torch.unique(cell_of * G + gene)          # sorted by (cell, gene), de-duplicated
    del cell_of, gene
    col = torch.div(key, G, rounding_mode="floor")
    if n_keep is not None:
        # trim the cells that ended above their target: keep a random n_keep of a cell's entries
        cnt = torch.bincount(col, minlength=N)
        if bool((cnt > n_keep).any()):
            start = torch.cumsum(cnt, 0) - cnt
            order = torch.argsort(col.double() + torch.rand(key.numel(), generator=g, device="cuda", dtype=torch.float64))   # random order inside a cell
            rank = torch.empty_like(order)
            rank[order] = torch.arange(key.numel(), device="cuda") - start[col[order]]
            del order
            key = key[rank < n_keep[col]]
            del rank
            col = torch.div(key, G, rounding_mode="floor")
    rowidx = (key - col * G).to(torch.int32)
    colptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    colptr[1:] = torch.cumsum(torch.bincount(col, minlength=N), 0)
    x = 1.0 + torch.floor(-torch.log2(1.0 - torch.rand(key.numel(), generator=g, device="cuda", dtype=torch.float64)))
    return colptr, rowidx, x


def bench_knn(torch, ops, args):
    """The exact kNN search in front of the Jaccard build ("next" row N2; reference call-site R/clustCells.R:57,60):
    100 k cells x 50 PCA components, k+1 = 31 (k = 30 plus the cell itself), manhattan — the reference's default."""
    N, d, k, metric = KNN_N, KNN_D, KNN_K, "manhattan"
    rng = np.random.default_rng(11)
    centers = rng.normal(scale=6.0, size=(40, d))
    lab = rng.integers(0, 40, size=N)
    Xh = centers[lab] + rng.normal(size=(N, d)) * rng.uniform(0.5, 2.0, size=(40, 1))[lab]
    X = torch.from_numpy(np.ascontiguousarray(Xh.T)).cuda()            # (d, N) == column-major N x d, as R holds it
    pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
    ws = torch.zeros(ops.knn_workspace_bytes(N, N, k), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((k, N), dtype=torch.int32, device="cuda")

    def run():
        ops.knn_prepare(X, N, d, metric, pts)
        ops.knn_search(pts, N, d, k, metric, 0, N, ws, idx, None)

    run()
    torch.cuda.synchronize()
    reps = 3
    t1 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t1) / reps
    # the same search with tile pruning switched off: every query tile against every candidate tile
    os.environ["GFICF_KNN_PRUNE"] = "0"
    run()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    t_plain = (time.perf_counter() - t1) / reps
    del os.environ["GFICF_KNN_PRUNE"]
    # VALU-bound, not a contraction: |a-b| accumulation costs 1.5 lane-instructions per element on gfx950
    # (one packed subtract per pair + one add with |.| per element); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.
    # The roofline is quoted on the unpruned form, whose work is N^2 d; the pruned form does the same job with less.
    lane_ops = 1.5 * N * N * d
    peak = 256 * 4 * 16 * 2.4e9
    res = {"metric": "knn_cells_per_sec", "value": N / t, "unit": "cells/s", "ms_per_pass": t * 1e3, "dtype": "f32",
           "config": {"workload": f"exact kNN, {N} cells x {d} components (Gaussian blobs), {k} nearest incl. self, {metric}, device-resident"},
           "pair_distances_per_sec": N * N / t,
           "ms_per_pass_unpruned": t_plain * 1e3,
           "roofline": {"bound": "valu", "kernel": "k_knn_tiles, unpruned", "achieved": round(lane_ops / t_plain / 1e12, 2), "peak": round(peak / 1e12, 2),
                        "unit": "T lane-instr/s", "frac": round(lane_ops / t_plain / peak, 4), "traffic": None,
                        "note": "1.5 VALU lane-instructions per (pair, dimension) is the minimum for f32 |a-b| accumulation on gfx950; "
                                "value / ms_per_pass are the default (pruned, exact) search, which skips candidate tiles by a triangle-inequality bound"}}
    if not args.no_cpu_baseline:
        import oracle

        cores = os.cpu_count() or 1
        nq = 2048                                                      # bounded sample: the first 2048 queries against all N points
        t1 = time.perf_counter()
        widx, _ = oracle.knn(Xh, k, metric, nthreads=cores, queries=(0, nq))
        tc = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": nq / tc, "unit": "cells/s", "cores": cores, "kind": "port",
                               "sample": f"the first {nq} queries against all {N} points, oracle brute force (f32, g++ -O2, {cores} threads); "
                                         "the reference's own search is Annoy (approximate, third-party) and cannot run here",
                               "gpu_over_cpu": (N / t) / (nq / tc)}
        res["checked_vs_oracle"] = bool(np.array_equal(idx[:, :nq].cpu().numpy().T, widx[:nq]))
    # the whole graph build of clustcells() on the device (R/clustCells.R:57-69,80): kNN -> neigh[,-1] -> Jaccard ->
    # weight > 0 filter -> symmetric adjacency, no host round trip (the edge count stays on the device)
    kj = k - 1
    cap = N * kj
    table = torch.empty((N, ops.kpad(kj)), dtype=torch.int32, device="cuda")
    u_ws = torch.zeros(cap, dtype=torch.int16, device="cuda")
    cell_ptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    out3 = torch.zeros((3, cap), dtype=torch.float64, device="cuda")
    aws = torch.zeros(ops.adjacency_workspace_bytes(N, cap), dtype=torch.uint8, device="cuda")
    indptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    indices = torch.zeros(2 * cap, dtype=torch.int32, device="cuda")
    ax = torch.zeros(2 * cap, dtype=torch.float64, device="cuda")

    def graph():
        run()
        ops.jaccard_ingest(idx[1:], N, kj, N, table)
        ops.jaccard_edges_filtered(table, N, kj, 0, N, u_ws, cell_ptr, out3)
        ops.adjacency(N, cap, cell_ptr[N:N + 1], out3, aws, indptr, indices, ax, grouped_by_source=True)

    def graph_only():
        ops.jaccard_ingest(idx[1:], N, kj, N, table)
        ops.jaccard_edges_filtered(table, N, kj, 0, N, u_ws, cell_ptr, out3)
        ops.adjacency(N, cap, cell_ptr[N:N + 1], out3, aws, indptr, indices, ax, grouped_by_source=True)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n

    tgr = timed(graph, reps)
    tja = timed(graph_only, 4 * reps)        # timed by itself (through round 5: total minus search, the difference of two figures with a 0.3 ms spread)
    res["graph_build"] = {"ms_total": tgr * 1e3, "ms_knn": t * 1e3, "ms_jaccard_filter_adjacency": tja * 1e3,
                          "kept_edges": int(cell_ptr[N]), "adjacency_nnz": int(indptr[N]),
                          "note": "kNN -> Jaccard -> weight > 0 filter -> symmetric adjacency (CSC), device-resident, one stream; "
                                  "ms_jaccard_filter_adjacency is timed by itself on the search's output"}
    # next row N4: community detection on that adjacency matrix (R/clustCells.R:80: RunModularityClustering, resolution 0.8,
    # 10 iterations); relaxed contract — same objective, modularity compared with the reference optimiser's own
    nnz_a = int(indptr[N])
    lws = torch.zeros(ops.louvain_workspace_bytes(N, nnz_a), dtype=torch.uint8, device="cuda")
    labels = torch.zeros(N, dtype=torch.int32, device="cuda")
    louv = lambda: ops.louvain(N, indptr, indices[:nnz_a], ax[:nnz_a], 0.8, 10, labels, lws)
    louv()
    tls = []
    for _ in range(5):
        t1 = time.perf_counter()
        n_cl, q_dev = louv()
        tls.append(time.perf_counter() - t1)
    tl = sorted(tls)[2]                      # median of five calls (each call ends synchronised: it returns the number of clusters)
    lv = {"ms": tl * 1e3, "ms_min": min(tls) * 1e3, "ms_max": max(tls) * 1e3, "cells_per_sec": N / tl, "clusters": int(n_cl), "modularity": q_dev,
          "note": "deterministic parallel Louvain, ONE start, device-resident adjacency in, labels out (round 6: convergence decided on the device, "
                  "the host one iteration ahead: the call synchronises once per level of the hierarchy)"}
    # the reference's defaults run n.start = 10 starts (R/clustCells.R:46): independent problems on one graph, run TOGETHER as one launch set
    lws10 = torch.zeros(ops.louvain_workspace_bytes(N, nnz_a, 10), dtype=torch.uint8, device="cuda")
    louv10 = lambda: ops.louvain(N, indptr, indices[:nnz_a], ax[:nnz_a], 0.8, 10, labels, lws10, 1, 10, 180582)
    louv10()
    tls = []
    for _ in range(3):
        t1 = time.perf_counter()
        n_cl10, q10 = louv10()
        tls.append(time.perf_counter() - t1)
    tl10 = sorted(tls)[1]
    lv["ten_starts"] = {"ms": tl10 * 1e3, "ms_min": min(tls) * 1e3, "ms_max": max(tls) * 1e3, "ms_per_start": tl10 * 1e2, "clusters": int(n_cl10), "modularity": q10, "workspace_MB": round(lws10.numel() / 1e6),
                        "note": "n.start = 10, seed 180582 (clustcells' defaults): the ten starts as one problem on the disjoint union of ten copies of the graph"}
    del lws10
    if not args.no_cpu_baseline:
        import scipy.sparse as sp

        import oracle
        from oracle import oracle_np

        if oracle.build_ref() is not None:
            A = sp.csc_matrix((ax[:nnz_a].cpu().numpy(), indices[:nnz_a].cpu().numpy(), indptr.cpu().numpy()), shape=(N, N))
            t1 = time.perf_counter()
            ref_labels, _ = oracle.modularity_reference(A, 0.8, 1, 1, 10, 0)
            tr = time.perf_counter() - t1
            lv["cpu_baseline"] = {"value": N / tr, "unit": "cells/s", "cores": 1, "kind": "reference", "seconds": tr,
                                  "modularity": oracle_np.modularity_np(A, ref_labels, 0.8), "clusters": int(ref_labels.max()) + 1,
                                  "sample": "the same adjacency matrix through oracle/_ref/modularity_optimizer (the reference's own "
                                            "src/ModularityOptimizer.cpp, -DSTANDALONE, -O2): 1 random start, 10 iterations, edge file I/O included",
                                  "gpu_over_cpu": tr / tl}
    res["louvain"] = lv
    return res


def bench_chain(torch, dist, ops, args, world, rank, dev, fence, max_over_ranks):
    """The sharded kNN -> Jaccard CHAIN (what a sharded clustcells() runs, R/clustCells.R:57-68): every rank prepares its block of
    points, ONE all-gather replicates them, every rank derives the same pivot order and searches the cells at its positions of
    that order (KnnShard.step_ordered) — the index block comes out in the new numbering, whose locality is what the halo form
    wants —, then the Jaccard build on local ids (JaccardHaloShard; the all-gather form if the request slots overflow) and the
    edges mapped back to the original ids.  cells/s of the whole job, and the same chain on ONE GPU (rank 0, own cells only,
    after the N-rank region) for the efficiency.  Weak scaling: cells_per_gpu cells per rank."""
    import gficf_amd
    from gficf_amd.dist import JaccardHaloShard, JaccardShard, KnnShard, edges_to_original_ids, shard_bounds

    n_per, d, kk, metric = args.cells_per_gpu, KNN_D, KNN_K, "manhattan"
    N = n_per * world
    k = kk - 1
    b, e = shard_bounds(N, world, rank)
    centers = np.random.default_rng(11).normal(scale=6.0, size=(40, d))
    spread = np.random.default_rng(12).uniform(0.5, 2.0, size=(40, 1))

    def block(r):                                                      # rank r's block of the clustered point set (clusters scattered over the cell order)
        rng = np.random.default_rng(1000 + r)
        br, er = shard_bounds(N, world, r)
        lab = rng.integers(0, 40, size=er - br)
        return centers[lab] + rng.normal(size=(er - br, d)) * spread[lab]

    xl = torch.from_numpy(np.ascontiguousarray(block(rank).T)).to(dev)
    ks = KnnShard(ops, N, d, kk, metric, device=dev)
    # the exchange form of the Jaccard stage: decided once on the first result, by every rank alike
    idx_ord, order = ks.step_ordered(xl)
    probe = JaccardHaloShard(ops, N, k, device=dev)
    probe.step(idx_ord[1:].contiguous())
    fits = 1
    try:
        probe.sync()
    except gficf_amd.GficfError as ex:
        if ex.status != "GFICF_ERR_CAPACITY":
            raise
        fits = 0
    named = probe.rows_named_outside()
    t_fit = torch.tensor([fits], dtype=torch.int32, device=dev)
    dist.all_reduce(t_fit, op=dist.ReduceOp.MIN)
    halo = int(t_fit.item()) == 1
    del probe
    js = JaccardHaloShard(ops, N, k, device=dev) if halo else JaccardShard(ops, N, k, device=dev, with_u=False, pipeline=False)

    def step():
        io, od = ks.step_ordered(xl)
        out = js.step(io[1:].contiguous())
        return edges_to_original_ids(out, od), io, od

    reps = max(2, min(args.steps, 5))
    step()
    fence()
    t_search = t_all = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        ks.step_ordered(xl)
        torch.cuda.synchronize()
        t_search += time.perf_counter() - t0
    fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, io, od = step()
    fence()
    t_all = max_over_ranks((time.perf_counter() - t0) / reps)
    t_search = max_over_ranks(t_search / reps)
    js.sync()
    res = {"cells_per_sec": N / t_all, "ms_per_pass": t_all * 1e3, "ms_search_ordered": t_search * 1e3, "cells_total": N, "k": k, "components": d,
           "jaccard_exchange": "halo on local ids" if halo else "all-gather (request slots overflow)", "rows_named_outside_the_block": named,
           "rows_of_a_block": e - b,
           "note": "per pass: prepare + all-gather of the points + pivot order + search of the rank's positions (exact, pruned) + Jaccard build on "
                   "the ordered ids + edges mapped back to the original ids; in order, one stream"}
    # ---- checked against the oracle (after the timed region): the first cells of rank 0's block — the search on the ordered layout,
    # and the Jaccard stage on the index matrix the ranks produced (gathered to rank 0 for this check only)
    nq = 128
    blocks_idx = [torch.zeros((kk, shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0]), dtype=torch.int32, device=dev) for r in range(world)]
    if dist.get_backend() == "gloo":
        tmp = [t.cpu() for t in blocks_idx]
        dist.all_gather(tmp, io[:, :e - b].contiguous().cpu())
        blocks_idx = tmp
    else:
        dist.all_gather(blocks_idx, io[:, :e - b].contiguous())
    if rank == 0:
        import oracle

        cores = os.cpu_count() or 1
        mat_ord = np.ascontiguousarray(torch.cat([t.cpu() for t in blocks_idx], dim=1).numpy().T)     # N x kk, new numbering, column 0 = self
        X = np.concatenate([block(r) for r in range(world)], axis=0)
        Xq = X[od.cpu().numpy().astype(np.int64)]                       # the layout the ranks searched
        widx, _ = oracle.knn(Xq, kk, metric, nthreads=cores, queries=(0, nq))
        ok_knn = bool(np.array_equal(mat_ord[:nq], widx[:nq]))
        want, _ = oracle.jaccard_cells(np.ascontiguousarray(mat_ord[:, 1:]), 0, nq, nthreads=cores)
        o1 = np.concatenate([np.zeros(1), od.cpu().numpy().astype(np.float64) + 1.0])
        want[:, 0], want[:, 1] = o1[want[:, 0].astype(np.int64)], o1[want[:, 1].astype(np.int64)]     # new numbering -> original ids
        ok_j = bool(np.array_equal(out[:, :nq * k].cpu().numpy().T, want))
        res["checked_vs_oracle"] = ok_knn and ok_j
        res["oracle_check"] = {"cells": nq, "search_rows_equal": ok_knn, "edges_equal": ok_j}
        del X, Xq, mat_ord
    del js, ks
    # ---- the same chain on ONE GPU: rank 0, its n_per cells only (the N = 1 chain: search + Jaccard build, device-resident)
    if rank == 0:
        x1 = torch.from_numpy(np.ascontiguousarray(block(0)[:n_per].T)).to(dev)
        pts = torch.zeros((n_per, ops.knn_dpad(d)), dtype=torch.float32, device=dev)
        ws = torch.zeros(ops.knn_workspace_bytes(n_per, n_per, kk), dtype=torch.uint8, device=dev)
        idx1 = torch.zeros((kk, n_per), dtype=torch.int32, device=dev)
        tb = torch.zeros((n_per, ops.row_words(n_per, k)), dtype=torch.int32, device=dev)
        o1 = torch.zeros((3, n_per * k), dtype=torch.float64, device=dev)

        def chain1():
            ops.knn_prepare(x1, n_per, d, metric, pts)
            ops.knn_search(pts, n_per, d, kk, metric, 0, n_per, ws, idx1, None)
            ops.jaccard_ingest(idx1[1:], n_per, k, n_per, tb)
            ops.jaccard_edges(tb, n_per, k, 0, n_per, o1, None)

        chain1()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            chain1()
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / reps
        ops.sync()
        res["single_gpu_chain"] = {"cells_per_sec": n_per / t1, "ms_per_pass": t1 * 1e3, "cells": n_per}
        res["efficiency"] = round((N / t_all) / (world * n_per / t1), 4)
    fence()
    return res


def traffic_child(args):
    """What the live --pmc passes profile: one data set of the workload, ingested, four launches of the edge kernel."""
    import torch

    import gficf_amd
    from gficf_amd import synth

    N_total, k = CONFIGS[args.config] if args.config in CONFIGS else (args.cells_per_gpu, K)
    k = args.k or k
    ops = gficf_amd.HipOps(0)
    m = synth.knn_windowed(N_total, k, seed=42, perm_seed=43 if args.ids == "permuted" else None)
    idx = torch.from_numpy(np.ascontiguousarray(m.T)).cuda()
    table = torch.zeros((N_total, ops.row_words(N_total, k)), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N_total * k), dtype=torch.float64, device="cuda")
    ops.set_jaccard_distinct(True)                                      # what the timed step runs under
    if ops.jaccard_one_launch(N_total, k):                              # a small problem: the step is ONE kernel (csrc/jaccard_direct.h)
        for _ in range(4):
            ops.jaccard(idx, N_total, k, table, out, None)
    else:
        ops.jaccard_ingest(idx, N_total, k, N_total, table)
        for _ in range(4):
            ops.jaccard_edges(table, N_total, k, 0, N_total, out, None)
    ops.sync()


def live_traffic(args, kernel_name: str, timeout_s: float = 90.0):
    """HBM-side bytes per launch of the edge kernel, measured NOW: two `rocprofv3 --pmc` passes (counters in their own runs,
    as MI355X_MICROARCH.md's HBM section prescribes) over a child process that launches the kernel on this workload.
    Reads: the L2's read requests to the fabric priced by their width (32 / 64 / 128 B; profiles/r03_fetch_calibration.txt: a
    miss fills a whole 128 B line, also for a 64 B row gather); writes: WRITE_SIZE (KiB).  Returns (bytes, detail) or
    (None, reason); never raises — the caller falls back to the committed figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler (no profiler inside a profiler)"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="gficf_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--config", args.config, "--ids", args.ids,
             "--cells-per-gpu", str(args.cells_per_gpu)] + (["--k", str(args.k)] if args.k else [])
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for tag, ctrs in (("rd", ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]), ("wr", ["WRITE_SIZE"])):
            cmd = [exe, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, tag), "-o", "pmc", "--"] + child
            # own session: on a timeout the profiler AND the process it wraps are ended, by process group (exact ids, no patterns)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                out_txt, _ = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal

                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                raise
            if pr.returncode != 0:
                return None, f"rocprofv3 pass '{tag}' exited {pr.returncode}: {out_txt[-200:]}"
            acc = {}
            for f in glob.glob(os.path.join(tmp, tag, "**", "pmc_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_name + "<" in row["Kernel_Name"] or row["Kernel_Name"].split("(")[0].endswith(kernel_name):
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in ctrs:
                if not acc.get(c):
                    return None, f"counter {c} missing from the '{tag}' pass"
                vals[c] = sum(acc[c]) / len(acc[c])
                vals["launches"] = len(acc[c])
    except subprocess.TimeoutExpired:
        return None, f"rocprofv3 pass timed out after {timeout_s:.0f} s"
    except Exception as ex:                                            # a profiler that misbehaves must not cost the bench line
        return None, f"{type(ex).__name__}: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd = 32 * vals["TCC_EA0_RDREQ_32B_sum"] + 64 * vals["TCC_EA0_RDREQ_64B_sum"] + 128 * vals["TCC_EA0_RDREQ_128B_sum"]
    wr = 1024 * vals["WRITE_SIZE"]
    return int(rd + wr), {"read_bytes": int(rd), "written_bytes": int(wr), "launches_profiled": vals["launches"],
                          "read_requests": {"32B": vals["TCC_EA0_RDREQ_32B_sum"], "64B": vals["TCC_EA0_RDREQ_64B_sum"], "128B": vals["TCC_EA0_RDREQ_128B_sum"]}}


def peer_child(args):
    """The `peer` leg of an N > 1 line, in a process of its own (started by rank 0 once the N-rank region is over; a leg that
    failed or hung must not cost the line): the single-process form of the C ABI (gficf_multi_jaccard_device) — one context per
    GPU, every block of the kNN matrix already in HBM, every device ingests its block, pulls the other P - 1 table slices with
    hipMemcpyPeerAsync (all pairs at once, copy streams + events, no collective) and builds its block's edges; outputs stay on
    their devices.  Same workload, same in-order protocol and the same batch as `value`.  Prints one JSON object."""
    import concurrent.futures as cf

    import torch

    import gficf_amd
    from gficf_amd import synth
    from gficf_amd.api import MultiContext

    P = args.gpus
    strong = args.config in CONFIGS
    if strong:
        N_total, k = CONFIGS[args.config]
        k = args.k or k
        batch = args.batch or 1
    else:
        k = args.k or K
        N_total = args.cells_per_gpu * P
        batch = args.batch or BATCH
    devices = [0] * P if args.rehearse_one_gpu else list(range(P))
    mc = MultiContext(devices)
    if not args.scan_dups:
        mc.set_jaccard_distinct(True)
    bd = mc.cell_blocks(N_total)
    rw = gficf_amd.HipOps.row_words(N_total, k)
    gen = lambda d: synth.knn_windowed(N_total, k, seed=42 + 7 * d, perm_seed=(43 + 7 * d) if args.ids == "permuted" else None)
    with cf.ThreadPoolExecutor(max_workers=min(batch, 8)) as ex:
        mats = list(ex.map(gen, range(batch)))
    idx, tables, outs = [], [], []
    for d in range(batch):
        idx.append([torch.from_numpy(np.ascontiguousarray(mats[d][bd[r]:bd[r + 1]].T)).to(f"cuda:{devices[r]}") for r in range(P)])
        tables.append([torch.zeros((N_total, rw), dtype=torch.int32, device=f"cuda:{devices[r]}") for r in range(P)])
        outs.append([torch.zeros((3, (bd[r + 1] - bd[r]) * k), dtype=torch.float64, device=f"cuda:{devices[r]}") for r in range(P)])
    mat0 = mats[0]
    del mats
    for dv in set(devices):
        torch.cuda.synchronize(dv)

    def step():
        for d in range(batch):
            mc.jaccard_device(idx[d], N_total, k, tables[d], outs[d])

    step()
    mc.sync()
    t0 = time.perf_counter()
    step()
    mc.sync()
    one = max(time.perf_counter() - t0, 1e-5)
    if args.pre_warm_ms > 0 and one < 0.02:
        for _ in range(int(min(5000, max(1, args.pre_warm_ms * 1e-3 / one)))):
            step()
        mc.sync()
    for _ in range(args.warmup):
        step()
    mc.sync()
    # what the ENQUEUE of a step costs the host (the step is asynchronous: ingest + P - 1 peer copies + events + edges per device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0
    mc.sync()
    dt = time.perf_counter() - t0
    import oracle

    ok, checked = True, 0
    for r in sorted({0, P // 2, P - 1}):                                # oracle samples: the first cells of three blocks, data set 0
        run = min(512, bd[r + 1] - bd[r])
        want, _ = oracle.jaccard_cells(mat0, bd[r], bd[r] + run, nthreads=os.cpu_count() or 1)
        ok = ok and bool(np.array_equal(outs[0][r][:, :run * k].cpu().numpy().T, want))
        checked += run
    res = {"edges_per_sec": N_total * k * batch * args.steps / dt, "ms_per_data_set": dt / args.steps / batch * 1e3,
           "host_enqueue_us_per_data_set": t_enq / args.steps / batch * 1e6, "devices": devices, "ids": args.ids,
           "table_row_bytes": 4 * rw, "bytes_pulled_per_device_per_data_set": 4 * rw * (N_total - (bd[1] - bd[0])),
           "checked_vs_oracle": ok, "oracle_check_cells": checked,
           "form": "single process, one context per GPU (gficf_multi_jaccard_device): ingest -> every device pulls the other "
                   "P - 1 table slices with hipMemcpyPeerAsync on copy streams of its own, all pairs at once -> edges; "
                   "device-resident, in order, no collective"}
    print(json.dumps(res), flush=True)                                  # (kept if the leg below fails: the parent reads the last JSON line)
    if k > 64 or P > 16:
        return
    # The same protocol on ids with locality, with NOTHING exchanged (gficf_multi_jaccard_halo_device): every device reads the few rows
    # its block names outside where they lie, in the other devices' blocks of ids.
    del idx, tables, outs
    with cf.ThreadPoolExecutor(max_workers=min(batch, 8)) as ex:
        mats = list(ex.map(lambda d: synth.knn_windowed(N_total, k, seed=42 + 7 * d, perm_seed=None), range(batch)))
    idx = [[torch.from_numpy(np.ascontiguousarray(mats[d][bd[r]:bd[r + 1]].T)).to(f"cuda:{devices[r]}") for r in range(P)] for d in range(batch)]
    bufs = [mc.halo_buffers(N_total, k) for _ in range(batch)]
    mat0 = mats[0]
    del mats
    for dv in set(devices):
        torch.cuda.synchronize(dv)

    def hstep():
        for d in range(batch):
            mc.jaccard_halo_device(idx[d], N_total, k, bufs[d])

    hstep()
    mc.sync()
    t0 = time.perf_counter()
    hstep()
    mc.sync()
    one = max(time.perf_counter() - t0, 1e-5)
    if args.pre_warm_ms > 0 and one < 0.02:
        for _ in range(int(min(5000, max(1, args.pre_warm_ms * 1e-3 / one)))):
            hstep()
        mc.sync()
    for _ in range(args.warmup):
        hstep()
    mc.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hstep()
    t_enq = time.perf_counter() - t0
    mc.sync()
    dt = time.perf_counter() - t0
    ok, checked = True, 0
    for r in sorted({0, P // 2, P - 1}):
        run = min(512, bd[r + 1] - bd[r])
        want, _ = oracle.jaccard_cells(mat0, bd[r], bd[r] + run, nthreads=os.cpu_count() or 1)
        ok = ok and bool(np.array_equal(bufs[0]["out"][r][:, :run * k].cpu().numpy().T, want))
        checked += run
    named = [int((bufs[0]["req"][r] != 0).sum()) for r in range(P)]
    # ... and overlapped: a second context of the same devices (streams of its own), the data sets taken in turn — the front end of one
    # context's step runs under the edge kernel of the other's
    over = None
    try:
        mc2 = MultiContext(devices)
        if not args.scan_dups:
            mc2.set_jaccard_distinct(True)

        def ostep():
            for d in range(batch):
                (mc2 if d & 1 else mc).jaccard_halo_device(idx[d], N_total, k, bufs[d])

        for _ in range(max(args.warmup, 2)):
            ostep()
        mc.sync(); mc2.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ostep()
        mc.sync(); mc2.sync()
        dto = time.perf_counter() - t0
        oko = True
        run = min(256, bd[P] - bd[P - 1])
        for d in sorted({0, min(1, batch - 1)}):                        # one data set of either context
            want, _ = oracle.jaccard_cells(mat0 if d == 0 else synth.knn_windowed(N_total, k, seed=42 + 7 * d, perm_seed=None), bd[P - 1], bd[P - 1] + run,
                                           nthreads=os.cpu_count() or 1)
            oko = oko and bool(np.array_equal(bufs[d]["out"][P - 1][:, :run * k].cpu().numpy().T, want))
        over = {"edges_per_sec": N_total * k * batch * args.steps / dto, "ms_per_data_set": dto / args.steps / batch * 1e3, "checked_vs_oracle": oko,
                "form": "two contexts of the same devices take the data sets in turn"}
        mc2.close()
    except gficf_amd.GficfError as ex:
        over = {"error": str(ex)}
    res["spatial_ids"] = {"edges_per_sec": N_total * k * batch * args.steps / dt, "ms_per_data_set": dt / args.steps / batch * 1e3,
                          "overlapped": over,
                          "host_enqueue_us_per_data_set": t_enq / args.steps / batch * 1e6, "cap": bufs[0]["cap"],
                          "rows_named_outside_per_device": named, "checked_vs_oracle": ok, "oracle_check_cells": checked,
                          "form": "single process, one context per GPU (gficf_multi_jaccard_halo_device): plan -> own cells' table rows -> the rows "
                                  "named outside, read where they lie in the owners' blocks of ids (peer mapping) -> edges; four launches per "
                                  "device, nothing exchanged, no collective, no copy; in order"}
    print(json.dumps(res), flush=True)


def run_peer_leg(args, timeout_s=300.0, tick=None):
    """Rank 0 of an N > 1 run starts the peer leg as a child process and reads its JSON object; never raises.  `tick` is called
    every few seconds while the child runs (the caller's sign of life)."""
    import subprocess
    import threading

    cmd = [sys.executable, os.path.abspath(__file__), "--peer-child", "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--config", args.config, "--ids", args.ids, "--cells-per-gpu", str(args.cells_per_gpu), "--pre-warm-ms", str(args.pre_warm_ms)]
    cmd += (["--k", str(args.k)] if args.k else []) + (["--batch", str(args.batch)] if args.batch else [])
    cmd += (["--rehearse-one-gpu"] if args.rehearse_one_gpu else []) + (["--scan-dups"] if args.scan_dups else [])
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GFICF_SPAWNED_RANK")}
    try:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        stop = threading.Event()
        if tick is not None:
            def beat():
                while not stop.wait(5.0):
                    tick()
            threading.Thread(target=beat, daemon=True).start()
        try:
            so, se = pr.communicate(timeout=timeout_s)
            stop.set()
        except subprocess.TimeoutExpired:
            stop.set()
            import signal

            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except OSError:
                pass
            pr.communicate()
            return {"error": f"the peer leg did not finish within {timeout_s:.0f} s"}
        lines = [l for l in so.splitlines() if l.lstrip().startswith("{")]
        if not lines:
            return {"error": f"the peer leg exited {pr.returncode}: {se[-400:]}"}
        res = json.loads(lines[-1])
        if pr.returncode != 0:                                          # (the first form's figures were printed before the second one failed)
            res["error_after_this"] = f"the peer leg exited {pr.returncode}: {se[-400:]}"
        return res
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}


T_PROCESS_START = time.monotonic()

# Legs of a run, in order.  `value` (the timed region, its roofline, the exchange figures and the oracle check) always runs; the
# others are further objects of the same line and can be selected with --legs.  N > 1: every leg is entered by all ranks together.
LEGS_ONE_GPU = ["value", "pipelined", "spatial_ids", "cpu_baseline", "stress", "host_abi", "gficf", "knn"]
LEGS_MULTI = ["value", "pipelined", "other_ids", "single_gpu_step", "chain", "peer", "gficf"]
# seconds a leg is expected to need at most (a leg is skipped, and named in `skipped_legs`, when less than this is left of --budget-s)
LEG_RESERVE_S = {"pipelined": 10, "other_ids": 40, "single_gpu_step": 20, "chain": 60, "peer": 60, "gficf": 40, "cpu_baseline": 40,
                 "stress": 10, "host_abi": 15, "knn": 60, "spatial_ids": 30}


class Bench:
    """One rank of a bench run: the workload, the timing helpers and one method per leg.  `out` is the line (rank 0 prints it)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        import gficf_amd
        from gficf_amd import synth
        from gficf_amd.dist import JaccardHaloShard, JaccardShard, shard_bounds

        self.args, self.torch, self.dist, self.gficf_amd, self.synth = args, torch, dist, gficf_amd, synth
        self.JaccardShard, self.JaccardHaloShard = JaccardShard, JaccardHaloShard
        self.world = world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (plain `python bench.py --gpus N` starts its own ranks)")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (gficf_amd has no CPU fallback)")
        share0 = bool(os.environ.get("GFICF_BENCH_RANKS_SHARE_GPU0"))     # test hook: RCCL is ASKED for with every rank on device 0 — it refuses
        if args.rehearse_one_gpu or share0:                               # (duplicate GPU), which is how the gloo fallback below gets exercised
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                             "(one rank per GPU; --rehearse-one-gpu shares device 0 over gloo)")
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.backend_note = None
        self.rccl_log = None
        if world > 1:
            if args.rehearse_one_gpu:
                dist.init_process_group("gloo")
            else:
                try:
                    # which algorithm / protocol RCCL picks for the exchange (a direct all-gather over the 7 links or a ring?)
                    # goes into the line as `exchange.rccl_algo`: rank 0 asks RCCL for its INFO log in a file of its own and reads it back
                    self.rccl_log = None
                    if rank == 0 and "NCCL_DEBUG" not in os.environ:
                        import tempfile

                        self.rccl_log = os.path.join(tempfile.gettempdir(), f"gficf_bench_rccl_{os.getpid()}.log")
                        os.environ["NCCL_DEBUG"], os.environ["NCCL_DEBUG_SUBSYS"], os.environ["NCCL_DEBUG_FILE"] = "INFO", "INIT,COLL,TUNING,ENV", self.rccl_log
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                except Exception as ex:  # noqa: BLE001
                    # First contact with RCCL happens in the driver's own run: if the communicator cannot be built (every rank sees the
                    # same failure), the run goes on over gloo — device tensors staged through the host, as the one-GPU rehearsals do —
                    # and SAYS SO in the line (`exchange.backend`): a slow, labelled number instead of no line.  A rank that fails alone
                    # meets nobody at the gloo rendezvous and gives up after two minutes.
                    import datetime

                    sys.stderr.write(f"bench.py: rank {rank}: RCCL process group failed ({type(ex).__name__}: {ex}); falling back to gloo\n")
                    self.backend_note = f"RCCL init failed ({type(ex).__name__}: {str(ex)[:200]}); gloo fallback, device tensors staged through the host"
                    try:
                        if dist.is_initialized():
                            dist.destroy_process_group()
                    except Exception:  # noqa: BLE001
                        pass
                    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        self.dev = torch.device("cuda", local_rank)
        self.strong = strong = args.config in CONFIGS
        if strong:
            self.N_total, k = CONFIGS[args.config]
            self.k = args.k or k
            self.batch = args.batch or 1
        else:
            self.k = args.k or K
            self.N_total = args.cells_per_gpu * world
            self.batch = args.batch or BATCH
        self.ops = gficf_amd.HipOps(local_rank)
        self.b, self.e = shard_bounds(self.N_total, world, rank)
        self.n_local = self.e - self.b
        self.edges_per_step = self.N_total * self.k * self.batch
        self.distinct = not args.scan_dups
        self.out = {}
        self.legs_done, self.leg_seconds, self.skipped_legs = [], {}, []
        sel = LEGS_ONE_GPU if world == 1 else LEGS_MULTI
        if args.legs:
            asked = [x.strip() for x in args.legs.split(",") if x.strip()]
            unknown = [x for x in asked if x not in sel]
            if unknown:
                raise SystemExit(f"--legs: {unknown} not among the legs of an N {'=' if world == 1 else '>'} 1 run: {sel}")
            sel = [x for x in sel if x == "value" or x in asked]
        self.selected = sel
        self.guard = {"t": time.monotonic(), "line": None, "done": False}
        self.single = None

    # ------------------------------------------------------------------------------------------------ helpers
    def make_inputs(self, ids_kind, need_full0):
        """This rank's block of `batch` independent data sets ((k, n_local) device tensors == column-major blocks, 1-based
        global ids) and, where asked, the full matrix of data set 0 (oracle check, CPU baseline).  Only the block is generated
        (synth.knn_windowed(rows=...)): at 8 ranks a rank does an eighth of the work of the full matrix."""
        import concurrent.futures as cf

        def gen(d):
            perm = (43 + 7 * d) if ids_kind == "permuted" else None
            if d == 0 and need_full0:
                return self.synth.knn_windowed(self.N_total, self.k, seed=42, perm_seed=perm)
            return self.synth.knn_windowed(self.N_total, self.k, seed=42 + 7 * d, perm_seed=perm, rows=(self.b, self.e))

        # (the data sets of a batch are independent: generated side by side — numpy releases the GIL in its inner loops; at 8 ranks x 100 k
        # cells this is the longest host-side stretch in front of `value`)
        with cf.ThreadPoolExecutor(max_workers=min(self.batch, 8)) as ex:
            mats = list(ex.map(gen, range(self.batch)))
        full0 = mats[0] if need_full0 else None
        loc = []
        for d, m in enumerate(mats):
            blk = m[self.b:self.e] if (d == 0 and need_full0) else m
            loc.append(self.torch.from_numpy(np.ascontiguousarray(blk.T)).to(self.dev))
        return loc, full0

    def pick_exchange(self, idx0, asked):
        """The exchange form is a property of the input (do the blocks name few rows outside themselves?): decided once, on
        data set 0, before anything is timed; every rank reaches the same decision (one all-reduce)."""
        if self.world == 1:
            return "allgather", None                                    # nothing to exchange; the plain single-device path
        if asked not in ("auto", "halo"):
            return asked, None
        probe = self.JaccardHaloShard(self.ops, self.N_total, self.k, device=self.dev)
        probe.step(idx0)
        fits = 1
        try:
            probe.sync()
        except self.gficf_amd.GficfError as ex:
            if ex.status != "GFICF_ERR_CAPACITY":
                raise
            fits = 0
        named = probe.rows_named_outside()
        t_fit = self.torch.tensor([fits], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t_fit, op=self.dist.ReduceOp.MIN)
        del probe
        if int(t_fit.item()) == 0:
            if asked == "halo":
                raise SystemExit("--exchange halo: the blocks name more rows than the request slots hold (ids without locality); use allgather / auto")
            return "allgather", named
        return "halo", named

    def make_shards(self, exch, n, pipeline=False):
        if exch == "halo":
            return [self.JaccardHaloShard(self.ops, self.N_total, self.k, device=self.dev, pipeline=pipeline) for _ in range(n)]
        return [self.JaccardShard(self.ops, self.N_total, self.k, device=self.dev, with_u=False, pipeline=pipeline,
                                  exchange="halo" if exch == "halo_generic" else "allgather") for _ in range(n)]

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def settle(self, step_fn):
        """Settle the clocks: W = 5 steps are 2 ms of work and K = 20 another 9 ms — from idle the GPU is still ramping through
        all of that (the same K steps measured 0.42 ms right after start-up and 0.375-0.38 ms from the third repetition on).
        The step is run untimed for a fixed wall time (every rank the same number of steps: collectives stay matched)."""
        if self.args.pre_warm_ms <= 0:
            return 0
        step_fn()                                                  # (first call: allocations, occupancy queries)
        self.fence()
        t_pw = time.perf_counter()
        step_fn()
        self.fence()
        one = self.max_over_ranks(max(time.perf_counter() - t_pw, 1e-5))
        if one >= 0.02:                                            # (steps of tens of milliseconds — a rehearsal over gloo — settle the clocks by themselves)
            return 0
        n = int(min(5000, max(1, self.args.pre_warm_ms * 1e-3 / one)))
        for _ in range(n):
            step_fn()
        self.fence()
        return n

    def timed(self, step_fn, steps, warmup):
        """W untimed steps, then exactly K steps between two fences (barrier + device sync on both sides); MAX over ranks."""
        for _ in range(warmup):
            step_fn()
        self.fence()
        ev0, ev1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step_fn()
        ev1.record()
        self.fence()
        dt = time.perf_counter() - t0
        return self.max_over_ranks(dt), ev0.elapsed_time(ev1)

    def measure_overlapped(self, exch, idx):
        """The overlapped mode (never `value`): ingest / exchange of data set d+1 on a side stream under the edge kernel of d.
        One rank: there is no exchange to hide and JaccardShard(pipeline=True) runs in order (gficf_amd/dist.py says why)."""
        a = self.args
        psh = self.make_shards(exch, 1, pipeline=True)[0]
        for _ in range(3 * self.batch):
            psh.step(idx[0])
        self.fence()
        t1 = time.perf_counter()
        for i in range(a.steps * self.batch):
            psh.step(idx[i % self.batch])
        self.fence()
        tp = self.max_over_ranks(time.perf_counter() - t1)
        psh.sync()
        in_order = bool(getattr(psh, "pipeline_in_order", False))
        del psh
        return {"edges_per_sec": self.N_total * self.k * self.batch * a.steps / tp, "ms_per_data_set": tp / (a.steps * self.batch) * 1e3,
                "note": ("one rank: nothing to exchange, so the pipelined form is refused and the steps run in order on one stream "
                         "(the two-table / three-stream form measured 20 % SLOWER on one GPU: its ~10 runtime calls per step are host-bound)"
                         if in_order else
                         "software-pipelined over two tables and two output buffers; a steady-state rate over many data sets, not a call")}

    # ------------------------------------------------------------------------------------------------ progress, budget, printing
    def elapsed(self):
        return time.monotonic() - T_PROCESS_START

    def line(self, unfinished=None):
        o = dict(self.out, legs_done=list(self.legs_done), leg_seconds={k_: round(v, 2) for k_, v in self.leg_seconds.items()},
                 skipped_legs=list(self.skipped_legs), wall_s=round(self.elapsed(), 1), budget_s=self.args.budget_s)
        if unfinished:
            o["unfinished_leg"] = unfinished
        return json.dumps(o)

    def progress(self, stage=None):
        """A sign of life (the watchdog counts the seconds since the last one) and, once `value` exists, the line as it stands."""
        self.guard["t"] = time.monotonic()
        if self.rank == 0 and "value" in self.out:
            self.guard["line"] = self.line(stage)

    def emit(self):
        """N > 1: the cumulative line, printed as soon as `value` exists and again after every leg (the LAST line of stdout is
        the most complete one; a run cut short at any moment leaves a whole line behind).  One GPU: one line, at the end."""
        if self.rank == 0 and self.world > 1:
            sys.stdout.write(self.line() + "\n")
            sys.stdout.flush()

    def start_watchdog(self):
        """A leg that hangs (a rank lost in a collective: first contact with real RCCL happens in the driver's own run) must not cost
        the line: rank 0 re-prints what it has, marked `unfinished_leg`, and ends the process with a NON-ZERO code, so that the
        launcher tears the other ranks down at once (they are blocked in the collective) and a harness sees the failure."""
        if self.world == 1 or self.rank != 0:
            return
        import threading

        limit = float(os.environ.get("GFICF_BENCH_LEG_TIMEOUT", str(max(60.0, min(240.0, self.args.budget_s * 0.6)))))

        def watchdog():
            while not self.guard["done"]:
                time.sleep(2.0)
                stuck = time.monotonic() - self.guard["t"] > limit
                over = self.elapsed() > self.args.budget_s + 120.0
                if not self.guard["done"] and (stuck or over):
                    self.guard["done"] = True
                    if self.guard["line"] is not None:
                        sys.stderr.write(f"bench.py: no sign of life for {limit:.0f} s (or the budget long overrun); printing the line as it stands and leaving with code 3\n")
                        sys.stdout.write(self.guard["line"] + "\n")
                        sys.stdout.flush()
                    else:
                        sys.stderr.write(f"bench.py: no sign of life for {limit:.0f} s before `value` existed: nothing to print; leaving with code 3\n")
                    os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()

    def run_leg(self, name, fn):
        """Run one leg if it is selected and the wall budget allows (every rank takes rank 0's decision), record its seconds,
        re-print the line."""
        if name not in self.selected:
            return
        skip = 1 if (self.rank == 0 and self.elapsed() + LEG_RESERVE_S.get(name, 10) > self.args.budget_s) else 0
        if self.world > 1:
            t = self.torch.tensor([skip], dtype=self.torch.int32, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            skip = int(t.item())
        if skip:
            self.skipped_legs.append(name)
            self.progress()
            self.emit()
            return
        self.progress(name)
        t0 = time.monotonic()
        if self.world == 1:
            # one GPU: a leg that fails must not cost the line (`value` is measured and checked by then); N > 1: every rank runs the same
            # collectives, so a failure propagates (the launcher stops the job; the lines printed so far stand)
            try:
                fn()
            except Exception as ex:  # noqa: BLE001
                import traceback

                self.out.setdefault("leg_errors", {})[name] = f"{type(ex).__name__}: {ex}"
                sys.stderr.write(f"bench.py: leg {name} failed:\n{traceback.format_exc()}\n")
        else:
            fn()
        self.leg_seconds[name] = time.monotonic() - t0
        self.legs_done.append(name)
        self.progress()
        self.emit()

    # ------------------------------------------------------------------------------------------------ leg: value
    def assert_same_format(self):
        """N > 1: the table format is a function of (N, k) AND of the library build and its GFICF_JACCARD_* switches — a rank
        started under another environment would silently disagree on the row pitch.  Exchanged and compared once, before the
        first step (gficf_amd.dist.assert_same_format)."""
        if self.world > 1:
            from gficf_amd.dist import assert_same_format

            assert_same_format(self.ops, self.N_total, self.k, device=self.dev)

    def leg_value(self):
        a, torch, ops = self.args, self.torch, self.ops
        N_total, k, batch, world, rank = self.N_total, self.k, self.batch, self.world, self.rank
        self.assert_same_format()
        self.progress("value: inputs")
        self.idx_local, self.mat = self.make_inputs(a.ids, rank == 0)
        self.progress("value: exchange form")
        self.exchange, self.named_outside = self.pick_exchange(self.idx_local[0], a.exchange)
        self.shards = self.make_shards(self.exchange, batch)
        self.progress("value: timed region")

        def step():
            for d in range(batch):
                self.shards[d].step(self.idx_local[d])

        self.step = step
        # One context, all cells: rows are taken to hold distinct ids, as the library's host entries take them — the ingest does not
        # scan every row for a repeated id; the edge kernel meets one while it builds the row's hash set and raises a deferred
        # GFICF_ERR_DUPLICATE_IDS (surfaced by the sync behind the timed region; the exact sequence would then be re-run).  Sharded
        # by cell blocks with an all-gather of rows the check stays complete across the job — every row is the own row of a cell of
        # exactly one rank, whose sync raises (the job then fails loudly on that rank) —, and so it does in the halo form (the rank
        # that owns a cell inserts its row).
        if self.distinct:
            ops.set_jaccard_distinct(True)
        # N = 1: the same W + K steps FROM IDLE first (what rounds 1-3 reported as `value`, and what one call from a cold R session
        # sees: the clocks are still ramping) ...
        value_from_idle = None
        if world == 1 and a.pre_warm_ms > 0:
            dt_idle, _ = self.timed(step, a.steps, a.warmup)
            value_from_idle = self.edges_per_step * a.steps / dt_idle
        # ... then the clocks are settled, then W warm-up steps, then the K timed ones
        pre_warm_steps = self.settle(step)
        self.progress("value: timed region")
        dt, region_ms = self.timed(step, a.steps, a.warmup)
        self.progress("value: after the timed region")
        for sh in self.shards:
            sh.sync()                                                   # surfaces deferred validation errors
        self.value = value = self.edges_per_step * a.steps / dt
        # the same K steps with the ingest's own duplicate scan (the sequence of rounds 1-3, `--scan-dups`), for comparison
        value_scan = value
        if self.distinct:
            ops.set_jaccard_distinct(False)
            dts, _ = self.timed(step, a.steps, 2)
            value_scan = self.edges_per_step * a.steps / dts
            ops.set_jaccard_distinct(True)
        halo_form = self.exchange == "halo"
        wl = (f"BASELINE config {a.config[1]}: ONE data set of {N_total} cells x k={k} split over {world} GPU(s) by cell block (strong scaling)"
              if self.strong else
              f"north-star point: {a.cells_per_gpu} cells x k={k} per GPU; N_total={N_total}; {batch} independent data sets per step")
        one_launch = self.one_launch_form()
        self.out.update({
            "metric": "jaccard_edges_per_sec", "value": value, "unit": "edges/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if self.strong else "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": wl + f", windowed kNN (W=100) with {a.ids} ids; per data set: " +
                                   ("ONE launch straight from the column-major input (no table: csrc/jaccard_direct.h), one library call" if one_launch else
                                    "ingest + edge kernel, one library call" if world == 1 else
                                    "halo plan + 2 all-to-alls (request slots, rows) + relabel + ingest + edge kernel on local ids" if halo_form else
                                    "ingest + RCCL all-gather of table rows + edge kernel") +
                                   ", device-resident, one stream, in order (no overlap between data sets or steps)" +
                                   ("; rows taken to hold distinct ids: no duplicate scan in the ingest, the edge kernel's hash-set build reports a "
                                    "repeated id (deferred GFICF_ERR_DUPLICATE_IDS, exact re-run) — what the host entries do" if self.distinct else ""),
                       "cells_total": N_total, "k": k, "data_sets_per_step": batch, "edges_per_step": self.edges_per_step,
                       "partition": f"cell blocks x{world}" + ("" if world == 1 else f", exchange: {self.exchange}")},
            "timed_region_ms": round(region_ms, 4),
            "value_from_idle": value_from_idle,
            "value_from_idle_note": ("the same W warm-up + K timed steps run FIRST, right after start-up, before the clock-settling pre-warm: "
                                     "what rounds 1-3 reported as `value` and what one call from a cold session costs" if value_from_idle is not None else None),
            "pre_warm": {"ms_asked": a.pre_warm_ms, "steps": pre_warm_steps,
                         "note": "the step run untimed before the W warm-up steps so that the K timed steps see settled clocks; --pre-warm-ms 0 turns it off"},
            "value_with_ingest_duplicate_scan": value_scan,
            "ms_per_data_set": dt / a.steps / batch * 1e3,
        })
        self.guard["t"] = time.monotonic()
        self.out["roofline"] = self.roofline()
        self.guard["t"] = time.monotonic()
        if world > 1:
            self.out["exchange"] = self.exchange_figures()
        self.oracle_check()

    def one_launch_form(self):
        """Does the step of this workload take the one-launch form (small problem, one rank, distinct ids)?"""
        return self.world == 1 and self.exchange == "allgather" and self.ops.jaccard_one_launch(self.N_total, self.k)

    def roofline(self):
        """Roofline of the dominant kernel.  Times are ONE HIP-event pair around a long run of launches on the launch stream, divided
        by the launches (a pair per launch overstated every kernel by the events' own cost: round 4's 43.9 us against rocprofv3's
        41.2): `kernel_ms` = the edge kernel launched back to back over the batch's tables (what the rocprofv3 trace of the same
        command averages); `step_ms_per_data_set_device` = the step's own sequence, same bracket; `ingest_kernel_ms` = the
        difference — what the ingest adds to a data set inside the step, boundary included — so that
        data_sets x (kernel + ingest) is the device time of a step by construction."""
        a, torch, ops = self.args, self.torch, self.ops
        N_total, k, batch, world, rank, b, e, n_local = self.N_total, self.k, self.batch, self.world, self.rank, self.b, self.e, self.n_local
        shards, idx_local = self.shards, self.idx_local
        halo_form = self.exchange == "halo"
        one_launch = self.one_launch_form()
        rot = {"i": 0}

        def one_edges():
            d = rot["i"] % batch
            rot["i"] += 1
            sh = shards[d]
            if halo_form:
                ops.jaccard_edges_mapped(sh.table, sh.n_ext, k, n_local, b, sh.l2g, sh.out, None)
            else:
                ops.jaccard_edges(sh.table, N_total, k, b, e, sh.out, None)

        def one_ingest():
            d = rot["i"] % batch
            rot["i"] += 1
            sh = shards[d]
            if halo_form:
                # the step's own ingest: the own cells' rows (relabel + ingest) and the serve step in one launch (k <= 64) over what the
                # last step's plan and exchange left in the shard's buffers; the table rows it writes are the ones the step wrote
                if not ops.halo_serve_ingest(idx_local[d], n_local, k, N_total, b, world, sh.rpr, sh.cap, sh.ws, sh.req_out, sh.req_in, sh.rows_out, sh.table, sh.l2g):
                    ops.halo_relabel(idx_local[d], n_local, k, N_total, b, world, sh.rpr, sh.cap, sh.ws, sh.req_out, sh.rows_in, sh.idx_ext, sh.l2g)
                    ops.jaccard_ingest_local(sh.idx_ext, sh.n_ext, k, sh.table)
            else:
                ops.jaccard_ingest(idx_local[d], n_local, k, N_total, sh.table[rank * sh.rpr:(rank + 1) * sh.rpr])

        launches = max(a.steps * batch, 40)
        # the step's own sequence, bracketed once (world > 1: the exchange is inside; every rank runs the same number of steps)
        reps = max(a.steps, 40 // batch + 1)
        self.step()
        self.fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.step()
        e1.record()
        self.fence()
        t_step_ds = e0.elapsed_time(e1) / (reps * batch)
        if one_launch:
            t_edges_ms, t_ingest_ms, t_ingest_b2b, t_ingest_scan_ms = t_step_ds, 0.0, 0.0, 0.0
        else:
            # (one rank: the step ran the library's one-call sequence; the separate calls below read the table it left — the same table)
            t_edges_ms = time_kernel_ms(torch, one_edges, launches)
            t_ingest_b2b = time_kernel_ms(torch, one_ingest, launches)
            t_ingest_scan_ms = t_ingest_b2b
            if self.distinct:                                           # the scanning ingest next to it, and the option back on
                ops.set_jaccard_distinct(False)
                t_ingest_scan_ms = time_kernel_ms(torch, one_ingest, launches)
                ops.set_jaccard_distinct(True)
            t_ingest_ms = max(t_step_ds - t_edges_ms, 0.0) if world == 1 else t_ingest_b2b
        alg_bytes = JACCARD_BYTES_PER_EDGE * n_local * k
        achieved = alg_bytes / (t_edges_ms * 1e-3) / 1e9
        # HBM-side bytes per launch from the committed PMC passes (tools/pmc.sh + tools/make_traffic.py; FETCH_SIZE
        # correction documented there); null when no pass was taken for this workload
        pmc = {}
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                pmc = json.load(open(prof))
            except Exception:
                pmc = {}
        self.pmc = pmc
        traffic = None if (halo_form or one_launch) else (pmc.get(f"jaccard_edges_pipe_N{N_total}_k{k}") or pmc.get(f"jaccard_edges_bits_N{N_total}_k{k}")
                                                         or pmc.get(f"jaccard_edges_N{N_total}_k{k}") or {}).get("hbm_bytes_per_launch")
        edge_kernel = "k_jaccard_edges_pipe" if (k <= 32 and not os.environ.get("GFICF_JACCARD_NO_PIPE")) else "k_jaccard_edges"   # the name rocprofv3 shows
        row_words = shards[0].row_words if halo_form else ops.row_words(N_total, k)
        if 32 < k <= 55 and row_words == 64 and (shards[0].n_ext if halo_form else N_total) <= 131070:
            edge_kernel = "k_jaccard_edges_bits"                         # dual rows: the direct-address bit-set kernel
        if k > 256:
            edge_kernel = "k_jaccard_edges_sorted"
        if one_launch:
            edge_kernel = "k_jaccard_direct"
        traffic_source = "profiles/pmc_traffic.json (separate --pmc passes of tools/pmc_round.sh, read requests priced by their width: tools/make_traffic.py)" if traffic else None
        traffic_detail = None
        if world == 1 and not a.no_live_traffic and not a.no_extras:
            # measured in THIS run (after the timed region; the child is a fresh process under rocprofv3, nothing of it is timed)
            live, detail = live_traffic(a, edge_kernel, timeout_s=min(90.0, max(20.0, a.budget_s - self.elapsed() - 60.0)))
            if live is not None:
                traffic_detail = dict(detail, committed_figure=traffic)
                traffic = live
                traffic_source = ("live: two rocprofv3 --pmc passes of this run over a child process launching the same kernel on the same workload "
                                  "(TCC_EA0_RDREQ_{32,64,128}B priced by width + WRITE_SIZE)")
            else:
                traffic_detail = {"live_failed": detail}
        return {"bound": "hbm", "kernel": edge_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_copy_rate": round(achieved / HBM_COPY_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source, "traffic_detail": traffic_detail,
                # what the kernel really moves: a gathered row fills a whole 128 B line (profiles/r03_fetch_calibration.txt)
                "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                "traffic_GBps": round(traffic / (t_edges_ms * 1e-3) / 1e9, 1) if traffic else None,
                "traffic_frac_of_copy_rate": round(traffic / (t_edges_ms * 1e-3) / 1e9 / HBM_COPY_GBS, 4) if traffic else None,
                "kernel_ms": round(t_edges_ms, 5), "kernel_ms_back_to_back": round(t_edges_ms, 5),
                "ingest_kernel_ms": round(t_ingest_ms, 5), "ingest_kernel_ms_back_to_back": round(t_ingest_b2b, 5),
                "ingest_kernel_ms_with_duplicate_scan": round(t_ingest_scan_ms, 5),
                "step_ms_per_data_set_device": round(t_step_ds, 5),
                "algorithmic_bytes_per_launch": alg_bytes,
                "row_bytes": 4 * row_words,
                "note": "kernel_ms: ONE HIP-event pair on the launch stream around a run of back-to-back launches of the kernel over the batch's "
                        "tables, divided by the launches (a pair per launch overstated it); step_ms_per_data_set_device: the step's own sequence "
                        "in the same bracket; ingest_kernel_ms (one rank): their difference — what the ingest adds inside the step; its own "
                        "back-to-back figure is bound by the host's enqueue rate at these sizes, not by the device" +
                        ("; ONE-LAUNCH form: the step is one kernel, kernel_ms is the step" if one_launch else "")}

    def exchange_figures(self):
        fig = self._exchange_figures()
        fig["backend"] = self.dist.get_backend()
        if self.backend_note:
            fig["backend_note"] = self.backend_note
        fig["rccl_algo"] = self.rccl_algo()
        return fig

    def rccl_algo(self):
        """What RCCL says it runs the exchange with (its INFO log, rank 0): algorithm / protocol / channels of the collective's tuning lines.
        Best effort — the log's wording belongs to RCCL; the lines found are quoted as they are."""
        if self.dist.get_backend() != "nccl":
            return {"reported": False, "why": f"backend {self.dist.get_backend()}: no RCCL collective ran"}
        if not self.rccl_log or not os.path.exists(self.rccl_log):
            return {"reported": False, "why": "NCCL_DEBUG was set by the caller (its log is the caller's) or RCCL wrote no log file"}
        try:
            with open(self.rccl_log, errors="replace") as f:
                lines = f.read().splitlines()
        except OSError as ex:
            return {"reported": False, "why": f"log unreadable: {ex}"}
        import re

        pick = [ln.strip()[-220:] for ln in lines if re.search(r"(?i)(allgather|alltoall|all_gather|sendrecv).*(algo|proto)|(?i)(algo|algorithm)\s*[=:]?\s*(ring|tree|direct|collnet|nvls|pat)", ln)]
        algo = sorted({m.group(1).lower() for ln in pick for m in re.finditer(r"(?i)\b(ring|tree|direct|collnet\w*|nvls\w*|pat)\b", ln)})
        proto = sorted({m.group(1).upper() for ln in pick for m in re.finditer(r"(?i)\b(LL128|LL|SIMPLE)\b", ln)})
        chan = [ln.strip()[-160:] for ln in lines if re.search(r"(?i)\bchannels?\b.*\b(coll|p2p|nChannels)", ln)][:2]
        return {"reported": bool(pick), "algorithms_named": algo, "protocols_named": proto, "lines": pick[:4], "channel_lines": chan,
                "log_lines": len(lines)}

    def _exchange_figures(self):
        a, sh0 = self.args, self.shards[0]
        if self.exchange == "halo":
            return {"bytes_received_per_rank_per_data_set": int(sh0.bytes_received),
                    "rows_received_per_rank_per_data_set": int(sh0.rows_named_outside()),
                    "row_bytes_on_the_wire": 4 * self.k, "ids": a.ids, "request_slots_per_owner": sh0.cap,
                    "rows_of_the_sub_problem": sh0.n_ext, "table_row_bytes": 4 * sh0.row_words,
                    "rows_named_outside_the_block_data_set_0": self.named_outside, "chosen_by": a.exchange,
                    "form": "halo on local ids: P x cap request slots out, P x cap raw index rows back (two all-to-alls with equal splits, "
                            "no host round trip); bytes = the fixed slot traffic, rows = the slots in use"}
        row_b = 4 * (sh0.pw if sh0.packed is not None else sh0.row_words)
        return {"bytes_received_per_rank_per_data_set": int(sh0.bytes_received), "rows_received_per_rank_per_data_set": int(sh0.rows_received),
                "row_bytes_on_the_wire": row_b, "ids": a.ids, "rows_named_outside_the_block_data_set_0": self.named_outside,
                "chosen_by": a.exchange,
                "form": ("halo: unique-id request lists + the named rows, two all-to-alls" if self.exchange == "halo_generic" else
                         "all-gather of table rows" + (" (bit-packed)" if sh0.packed is not None else ""))}

    def oracle_check(self):
        """Every line carries a check against the oracle (checker only, after the timed region): the whole matrix of data
        set 0 when that takes seconds (one GPU, <= 200 k cells, extras on), otherwise a bounded sample of this rank's
        cells — three runs of 1024 consecutive source cells at the start, the middle and the end of its block."""
        if self.rank != 0:
            return
        import oracle

        a, N_total, k, b, e, n_local = self.args, self.N_total, self.k, self.b, self.e, self.n_local
        shard, mat = self.shards[0], self.mat
        cores = os.cpu_count() or 1
        extras = not a.no_extras and not self.strong
        if self.world == 1 and (extras or N_total * k <= 2_000_000) and N_total <= 200_000:
            want, _ = oracle.jaccard(mat, nthreads=cores)
            ok, checked = bool(np.array_equal(shard.out.cpu().numpy().T, want)), N_total
            del want
        else:
            run = min(1024, n_local)
            ok, checked = True, 0
            for c0 in sorted({b, b + (n_local - run) // 2, e - run}):
                want, _ = oracle.jaccard_cells(mat, c0, c0 + run, nthreads=cores)
                got = shard.out[:, (c0 - b) * k:(c0 - b + run) * k].cpu().numpy().T
                ok = ok and bool(np.array_equal(got, want))
                checked += run
        self.out["checked_vs_oracle"] = ok
        self.out["oracle_check"] = {"cells": checked, "of": n_local, "kind": "whole matrix" if checked == N_total else "sample of source cells",
                                    "what": "bit-exact rows of the reference's (N*k) x 3 matrix, data set 0"}

    # ------------------------------------------------------------------------------------------------ legs shared by N = 1 and N > 1
    def leg_pipelined(self):
        if self.world == 1:
            # one rank: nothing to exchange and nothing worth overlapping — JaccardShard(pipeline=True) refuses the two-table / three-stream
            # form and runs its steps in order on the caller's stream (gficf_amd/dist.py says why), i.e. it runs `value`'s own loop: the
            # figure IS `value`, no second measurement
            self.out["pipelined"] = {"edges_per_sec": self.value, "ms_per_data_set": self.out["ms_per_data_set"], "refused_on_one_rank": True,
                                     "note": "one rank: JaccardShard(pipeline=True) runs in order (pipeline_in_order): the steps of `value`, not a second measurement"}
            return
        self.out["pipelined"] = self.measure_overlapped(self.exchange, self.idx_local)

    def leg_spatial_ids_one_gpu(self):
        """One GPU, the big single-data-set configs (4 and 5): the same step on ids WITH locality — cells numbered in their spatial
        order, what the device kNN search's pivot order gives (gficf_knn_pivot_order_device; KnnShard.step_ordered) — next to `value`
        on permuted ids: what the walk order is worth to the edge kernel when it comes for free."""
        a, torch, ops = self.args, self.torch, self.ops
        other = "spatial" if a.ids == "permuted" else "permuted"
        idx_o, _ = self.make_inputs(other, False)
        sh = self.make_shards("allgather", 1)[0]
        step_o = lambda: sh.step(idx_o[0])
        if self.distinct:
            ops.set_jaccard_distinct(True)
        dt_o, _ = self.timed(step_o, a.steps, a.warmup)
        sh.sync()
        t_edges = time_kernel_ms(torch, lambda: ops.jaccard_edges(sh.table, self.N_total, self.k, 0, self.N_total, sh.out, None), max(a.steps, 100))
        import oracle

        mat_o = self.synth.knn_windowed(self.N_total, self.k, seed=42, perm_seed=43 if other == "permuted" else None)
        run = min(1024, self.n_local)
        want, _ = oracle.jaccard_cells(mat_o, 0, run, nthreads=os.cpu_count() or 1)
        alg = JACCARD_BYTES_PER_EDGE * self.N_total * self.k
        self.out[f"{other}_ids"] = {"ids": other, "edges_per_sec": self.N_total * self.k * a.steps / dt_o, "ms_per_data_set": dt_o / a.steps * 1e3,
                                    "kernel_ms": round(t_edges, 5), "roofline_frac": round(alg / (t_edges * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "checked_vs_oracle": bool(np.array_equal(sh.out[:, :run * self.k].cpu().numpy().T, want)),
                                    "note": "the same single-GPU step on the other id model (one data set, in order); kernel_ms: the edge kernel back to back, one event pair"}

    # ------------------------------------------------------------------------------------------------ legs of an N > 1 run
    def leg_other_ids(self):
        """The other id model: ids WITH locality (what the device kNN search's pivot order gives, gficf_knn_pivot_order_device)
        when `value` ran on permuted ids, and the other way round."""
        a, torch = self.args, self.torch
        other = "spatial" if a.ids == "permuted" else "permuted"
        self.other = other
        idx_o, _ = self.make_inputs(other, False)
        ex_o, named_o = self.pick_exchange(idx_o[0], "auto")
        shards_o = self.make_shards(ex_o, self.batch)

        def step_o():
            for d in range(self.batch):
                shards_o[d].step(idx_o[d])

        dt_o, _ = self.timed(step_o, a.steps, a.warmup)
        for sh in shards_o:
            sh.sync()
        sh0 = shards_o[0]
        obj = {"ids": other, "exchange": ex_o, "rows_named_outside_the_block_data_set_0": named_o,
               "in_order": {"edges_per_sec": self.edges_per_step * a.steps / dt_o, "ms_per_data_set": dt_o / a.steps / self.batch * 1e3},
               "bytes_received_per_rank_per_data_set": int(sh0.bytes_received),
               "table_row_bytes": 4 * sh0.row_words}
        if ex_o == "halo":
            obj.update({"rows_named_outside": int(sh0.rows_named_outside()), "request_slots_per_owner": sh0.cap, "rows_of_the_sub_problem": sh0.n_ext})
        # checked like `value`: a bounded oracle sample of this rank's block (rank 0), data set 0
        if self.rank == 0:
            import oracle

            mat_o = self.synth.knn_windowed(self.N_total, self.k, seed=42, perm_seed=43 if other == "permuted" else None)
            run = min(512, self.n_local)
            want, _ = oracle.jaccard_cells(mat_o, self.b, self.b + run, nthreads=os.cpu_count() or 1)
            obj["checked_vs_oracle"] = bool(np.array_equal(sh0.out[:, :run * self.k].cpu().numpy().T, want))
            del mat_o, want
        del shards_o, sh0
        self.out[f"{other}_ids"] = obj
        self.emit()                                                     # (the in-order figure is out before the overlapped one is taken)
        obj["overlapped"] = self.measure_overlapped(ex_o, idx_o)

    def leg_single_gpu_step(self):
        """The N = 1 step, timed by rank 0 in this same process once the N-rank region is over (the other ranks wait at the
        fence): one library call per data set (ingest + edge kernel), one stream, in order, on `cells_1` cells — what `python bench.py` times."""
        a, torch, ops, k, batch = self.args, self.torch, self.ops, self.k, self.batch
        cells_1 = self.N_total if self.strong else a.cells_per_gpu
        if self.rank == 0:
            tb1 = [torch.zeros((cells_1, ops.row_words(cells_1, k)), dtype=torch.int32, device=self.dev) for _ in range(batch)]
            o1 = [torch.zeros((3, cells_1 * k), dtype=torch.float64, device=self.dev) for _ in range(batch)]
            i1 = [torch.from_numpy(np.ascontiguousarray(self.synth.knn_windowed(cells_1, k, seed=42 + 7 * d, perm_seed=(43 + 7 * d) if a.ids == "permuted" else None).T)).to(self.dev)
                  for d in range(batch)]
            runs = [ops.jaccard_prepared(i1[d], cells_1, k, tb1[d], o1[d], None) for d in range(batch)]

            def step1():
                for r in runs:
                    r()

            for _ in range(max(a.warmup, 3)):
                step1()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                step1()
            torch.cuda.synchronize()
            t_1 = time.perf_counter() - t1
            ops.sync()
            self.single = cells_1 * k * batch * a.steps / t_1
            self.out["single_gpu_step"] = {"edges_per_sec": self.single, "ms_per_data_set": t_1 / a.steps / batch * 1e3, "cells": cells_1, "ids": a.ids,
                                           "note": "rank 0 alone, after the N-rank region of this same process (the other ranks wait): "
                                                   "ingest + edge kernel per data set, one stream, in order — the step `python bench.py` times"}
            del tb1, o1, i1, runs
        self.fence()

    def leg_chain(self):
        """The kNN -> Jaccard chain in the search's pivot order (what a sharded clustcells() runs)."""
        if self.strong:
            return
        try:
            ch = bench_chain(self.torch, self.dist, self.ops, self.args, self.world, self.rank, self.dev, self.fence, self.max_over_ranks)
        except self.gficf_amd.GficfError as ex:                          # (a failure of this leg must not cost the line; collectives stay matched: every rank runs the same code)
            ch = {"error": str(ex)}
        if self.rank == 0:
            self.out["chain"] = ch

    def leg_peer(self):
        """The single-process form with direct peer copies, in a process of its own (rank 0 starts it; the ranks wait)."""
        if self.rank == 0:
            left = self.args.budget_s - self.elapsed()
            self.out["peer"] = run_peer_leg(self.args, timeout_s=max(30.0, min(300.0, left - 10.0)), tick=lambda: self.progress("peer"))
        self.fence()

    def efficiency(self):
        if self.rank != 0 or self.world == 1 or not self.single:
            return
        a, out, world = self.args, self.out, self.world
        per = lambda v: round(v / (world * self.single), 4)
        eff = {"definition": "whole-job edges/s / (n_gpus x single_gpu_step.edges_per_sec), all measured in this run",
               f"in_order_{a.ids}": per(self.value)}
        if "pipelined" in out:
            eff[f"overlapped_{a.ids}"] = per(out["pipelined"]["edges_per_sec"])
        other = getattr(self, "other", None)
        oo = out.get(f"{other}_ids") if other else None
        if oo:
            eff[f"in_order_{other}"] = per(oo["in_order"]["edges_per_sec"])
            if "overlapped" in oo:
                eff[f"overlapped_{other}"] = per(oo["overlapped"]["edges_per_sec"])
        if "edges_per_sec" in out.get("peer", {}):
            eff[f"peer_in_order_{a.ids}"] = per(out["peer"]["edges_per_sec"])
        if "edges_per_sec" in out.get("peer", {}).get("spatial_ids", {}):
            eff["peer_halo_in_order_spatial"] = per(out["peer"]["spatial_ids"]["edges_per_sec"])
            if "edges_per_sec" in (out["peer"]["spatial_ids"].get("overlapped") or {}):
                eff["peer_halo_overlapped_spatial"] = per(out["peer"]["spatial_ids"]["overlapped"]["edges_per_sec"])
        out["efficiency"] = eff
        # ---- the model's figure beside every measured one (the constants and formulas of tools/project_scaling.py), so that ONE run on
        # the node says whether the projections hold: per exchange form {measured, projected, residual = measured - projected}
        LINK_GBS, LAT_US, GAP_US, PEER_LAT_US = 55.0, 20.0, 2.0, 6.0
        t1_us = (self.n_local * self.k) / self.single * 1e6                  # one data set on one GPU, measured in this run
        ex = out.get("exchange", {})
        shared = bool(a.rehearse_one_gpu or os.environ.get("GFICF_BENCH_RANKS_SHARE_GPU0"))
        how = ("time-sliced, meaningless: the ranks share ONE GPU (rehearsal) — only `projected` says anything" if shared else
               "measured on this node in this run")

        def entry(measured, ex_us, extra_launches, overlapped=False):
            step = max(t1_us, ex_us) if overlapped else t1_us + ex_us + extra_launches * GAP_US
            proj = round(t1_us / step, 4)
            e = {"measured": measured, "projected": proj, "measured_is": how}
            e["residual"] = None if measured is None else round(measured - proj, 4)
            return e

        forms = {}
        bytes_rx = float(ex.get("bytes_received_per_rank_per_data_set", 0))
        if self.exchange == "halo":
            # two all-to-alls with equal splits: a peer's share over its own link, one latency each
            ex_us = 2 * (bytes_rx / max(world - 1, 1) / 2 / (LINK_GBS * 1e3) + LAT_US)
            extra = 3
        else:
            # all-gather: every rank's block over each of its links once (+ pack / unpack launches when the rows travel bit-packed)
            ex_us = bytes_rx / max(world - 1, 1) / (LINK_GBS * 1e3) + LAT_US
            extra = 3 if "bit-packed" in ex.get("form", "") else 1
        forms["value"] = entry(eff.get(f"in_order_{a.ids}"), ex_us, extra)
        forms["value"]["exchange_model_us"] = round(ex_us, 1)
        if f"overlapped_{a.ids}" in eff:
            forms["pipelined"] = entry(eff[f"overlapped_{a.ids}"], ex_us, extra, overlapped=True)
        if oo and f"in_order_{other}" in eff:
            bo = float(oo.get("bytes_received_per_rank_per_data_set", 0) or 0)
            halo_o = oo.get("exchange") == "halo"
            exo = 2 * (bo / max(world - 1, 1) / 2 / (LINK_GBS * 1e3) + LAT_US) if halo_o else bo / max(world - 1, 1) / (LINK_GBS * 1e3) + LAT_US
            forms[f"{other}_ids"] = entry(eff[f"in_order_{other}"], exo, 3)
            if f"overlapped_{other}" in eff:
                forms[f"{other}_ids_pipelined"] = entry(eff[f"overlapped_{other}"], exo, 3, overlapped=True)
        if f"peer_in_order_{a.ids}" in eff:
            # peer copies: the other ranks' UNPACKED slices, one per link, all pairs at once; a copy's start-up instead of a collective's latency
            row_b = 4 * self.ops.row_words(self.N_total, self.k)
            exp = (self.N_total - self.n_local) / max(world - 1, 1) * row_b / (LINK_GBS * 1e3) + PEER_LAT_US
            forms["peer"] = entry(eff[f"peer_in_order_{a.ids}"], exp, 1)
        if "peer_halo_in_order_spatial" in eff:
            forms["peer_spatial"] = entry(eff["peer_halo_in_order_spatial"], 3.0, 2)      # nothing exchanged: a dependent read over xGMI inside one launch
        eff["forms"] = forms
        eff["model"] = {"link_GB_per_s_per_direction": LINK_GBS, "collective_latency_us": LAT_US, "launch_gap_us": GAP_US, "peer_copy_startup_us": PEER_LAT_US,
                        "single_gpu_data_set_us": round(t1_us, 1),
                        "formulas": "in order: t1 / (t1 + exchange + gaps); pipelined: t1 / max(t1, exchange); all-gather: block bytes per link / link rate + latency; "
                                    "halo: two equal-split all-to-alls; peer: unpacked slices per link + copy start-up (tools/project_scaling.py)"}

    def leg_gficf_sharded(self):
        """GF-ICF, cell-sharded: every rank owns a 54 k-cell block of a (54 k x n_gpus)-cell matrix; the only
        exchange is the all-reduce(sum) of the G per-gene cell counts between the count and the scale pass."""
        if self.strong:
            return
        from gficf_amd.dist import GficfShard

        torch, dist, ops, world = self.torch, self.dist, self.ops, self.world
        G, Nc = GFICF_G, GFICF_N
        colptr, rowidx, x = synth_counts_device(torch, G, Nc, seed=7 + self.rank)
        gs = GficfShard(ops, G, Nc * world, Nc, int(rowidx.numel()), device=self.dev)
        for _ in range(3):
            gs.step(colptr, rowidx, x, 0.05, 1.0)
        self.fence()
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            gs.step(colptr, rowidx, x, 0.05, 1.0)
        self.fence()
        tg = torch.tensor([(time.perf_counter() - t1) / reps], dtype=torch.float64, device=self.dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        nnz_all = torch.tensor([float(rowidx.numel())], dtype=torch.float64, device=self.dev)
        dist.all_reduce(nnz_all, op=dist.ReduceOp.SUM)
        ops.sync()
        tg, nnz_all = float(tg.item()), float(nnz_all.item())
        self.out["gficf"] = {"metric": "gficf_cells_per_sec", "value": Nc * world / tg, "unit": "cells/s", "ms_per_pass": tg * 1e3,
                             "config": {"workload": f"{G} genes x {Nc} cells per GPU (synthetic UMI CSC, one block per rank), "
                                                    f"{Nc * world} cells total; 1 all-reduce of {G} int64 gene counts per pass"},
                             "nnz_total": nnz_all, "dtype": "f64", "scaling": "weak",
                             "roofline": {"bound": "hbm", "kernel": "whole pass, all ranks", "achieved": round(GFICF_BYTES_PER_NNZ * nnz_all / tg / 1e9, 2),
                                          "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                          "frac": round(GFICF_BYTES_PER_NNZ * nnz_all / tg / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None}}

    # ------------------------------------------------------------------------------------------------ legs of a one-GPU run
    def leg_cpu_baseline(self):
        if self.args.no_cpu_baseline:
            return
        import oracle

        N_total, k = self.N_total, self.k
        cores = os.cpu_count() or 1
        mf = np.asfortranarray(self.mat.astype(np.float64))
        rmh = np.empty((3, N_total * k))
        uh = np.empty(N_total * k, dtype=np.int32)
        L = oracle.lib()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            L.oracle_jaccard_f64(mf.ctypes.data, N_total, k, rmh.ctypes.data, uh.ctypes.data, cores)
            ts.append(time.perf_counter() - t1)
        t1 = time.perf_counter()
        L.oracle_jaccard_f64(mf.ctypes.data, N_total, k, rmh.ctypes.data, uh.ctypes.data, 2)
        t_nt2 = time.perf_counter() - t1
        edges_ds = N_total * k                                      # one data set
        cpu_v = edges_ds / statistics.median(ts)
        self.out["cpu_baseline"] = {"value": cpu_v, "unit": "edges/s", "cores": cores, "kind": "port",
                                    "sample": f"the full workload ({N_total} cells x k={k}, same input), median of 3 passes of the oracle's "
                                              "faithful restatement of the RcppParallel worker (g++ -O2, dynamic chunks over std::thread)",
                                    "nt2_value": edges_ds / t_nt2, "nt2_note": "clustcells() default nt = 2 (reference R/clustCells.R:46)",
                                    "gpu_over_cpu": self.value / cpu_v}

    def leg_stress(self):
        """Stress row: uniformly random neighbour ids — u ~ 0, no overlap to exploit, same traffic."""
        torch, N_total, k = self.torch, self.N_total, self.k
        try:
            umat = self.synth.knn_uniform(N_total, k)
            uidx = torch.from_numpy(np.ascontiguousarray(umat.T)).to(self.dev)
            ush = self.JaccardShard(self.ops, N_total, k, device=self.dev, pipeline=False)
            t_u = time_kernel_ms(torch, lambda: ush.step(uidx), 40)
            self.out["stress_uniform_ids"] = {"edges_per_sec": N_total * k / (t_u * 1e-3), "ms_per_data_set": t_u,
                                              "nonzero_edge_fraction": float((ush.out[2] > 0).double().mean().item())}
        except Exception as ex:  # pragma: no cover
            self.out["stress_uniform_ids"] = {"error": str(ex)}

    def leg_host_abi(self):
        """End to end through the host C ABI (what the R glue calls): H2D + ingest + edges + D2H, device scratch from the
        context's pool.  PCIe-inclusive: reported, never `value`.  Two forms: the reference's 24 B/edge matrix, and the
        compact return (2 B/edge intersection counts; gficf_jaccard_expand_host rebuilds the matrix on the host)."""
        N_total, k = self.N_total, self.k
        try:
            import ctypes

            from gficf_amd import _lib

            L = _lib.load()
            hm = np.asfortranarray(self.mat)
            E1 = N_total * k
            hr = np.empty((3, E1), dtype=np.float64)
            hu = np.empty(E1, dtype=np.uint16)
            hctx = self.gficf_amd.default_context(self.local_rank)
            vp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)

            def timed(call, reps=5):
                call()
                t1 = time.perf_counter()
                for _ in range(reps):
                    rc = call()
                return (time.perf_counter() - t1) / reps, rc

            th, rc = timed(lambda: L.gficf_jaccard_host(hctx.handle, vp(hm), 0, N_total, k, N_total, vp(hr), 0))
            # ... and into a FRESH result matrix every call, as R allocates one (its pages are first touched inside the call)
            tf, rcf = timed(lambda: L.gficf_jaccard_host(hctx.handle, vp(hm), 0, N_total, k, N_total, vp(np.empty((3, E1), dtype=np.float64)), 0))
            # ... and the same two with the matrix itself copied back over PCIe (round 4's form; the switch is read per call)
            os.environ["GFICF_JACCARD_HOST_COMPACT_MIN_EDGES"] = str(1 << 62)
            try:
                tm, _ = timed(lambda: L.gficf_jaccard_host(hctx.handle, vp(hm), 0, N_total, k, N_total, vp(hr), 0))
                tmf, _ = timed(lambda: L.gficf_jaccard_host(hctx.handle, vp(hm), 0, N_total, k, N_total, vp(np.empty((3, E1), dtype=np.float64)), 0))
            finally:
                del os.environ["GFICF_JACCARD_HOST_COMPACT_MIN_EDGES"]
            self.out["host_abi"] = {"edges_per_sec": E1 / th, "ms_per_call": th * 1e3, "rc": rc, "ms_per_call_fresh_result_buffer": tf * 1e3,
                                    "matrix_over_pcie": {"ms_per_call": tm * 1e3, "ms_per_call_fresh_result_buffer": tmf * 1e3,
                                                         "note": "GFICF_JACCARD_HOST_COMPACT_MIN_EDGES=2^62: the 24 B/edge matrix copied back (round 4's form), same process"},
                                    "note": f"gficf_jaccard_host: pageable host buffers, device scratch from the context pool; from 2^20 edges on the "
                                            f"result returns as uint16 counts over PCIe ({4 * E1 / 1e6:.0f} MB in, {2 * E1 / 1e6:.0f} MB out) and the host cores "
                                            f"write the {24 * E1 / 1e6:.0f} MB matrix (round 4: the matrix itself crossed PCIe: 1.63 ms into a reused buffer, "
                                            "4.2 ms into a fresh one).  The `fresh_result_buffer` figures of THIS process include the first touch of a "
                                            "72 MB matrix after the earlier legs have fragmented its memory (~5 ms either way); fresh processes — an R "
                                            "session's situation — are measured by tools/host_compact_ab.py (profiles/r05_host_compact_ab.txt: 8.2 -> 2.0 ms)"}
            tc, rc = timed(lambda: L.gficf_jaccard_counts_host(hctx.handle, vp(hm), 0, N_total, k, N_total, vp(hu)))
            tx, rc2 = timed(lambda: L.gficf_jaccard_expand_host(vp(hm), 0, N_total, k, N_total, vp(hu), vp(hr), 0))
            want_full = self.shards[0].out.cpu().numpy()
            self.out["host_abi_counts"] = {"edges_per_sec": E1 / tc, "ms_per_call": tc * 1e3, "rc": rc, "expand_on_host_ms": tx * 1e3,
                                           "expanded_equals_device_matrix": bool(rc2 == 0 and np.array_equal(hr, want_full)),
                                           "note": "gficf_jaccard_counts_host: uint16 intersection counts only (2 B/edge out); "
                                                   "expand_on_host_ms = gficf_jaccard_expand_host rebuilding the reference matrix on the host cores"}
        except Exception as ex:  # pragma: no cover
            self.out["host_abi"] = {"error": str(ex)}

    def leg_gficf(self):
        a, torch, ops, dev = self.args, self.torch, self.ops, self.dev
        G, Nc = GFICF_G, GFICF_N
        pmc = getattr(self, "pmc", {})
        colptr, rowidx, x = synth_counts_device(torch, G, Nc)
        nnz = int(rowidx.numel())
        ws = ops.csc_workspace(G, Nc, nnz)
        run = lambda: ops.gficf_csc(G, Nc, colptr, rowidx, x, 0.05, 1.0, None, ws)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        reps = 10

        def batches(fn, n=5):
            """n batches of `reps` passes each, device-synchronised around every batch; the figure is the MEDIAN batch's mean (round 4
            quoted the fastest batch, and its record the fastest of four boxes on top: one selection too many); all batch means are
            reported."""
            ts = []
            for _ in range(n):
                t1 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) / reps)
            return statistics.median(ts), ts

        tg, tg_all = batches(run)
        ops.sync()
        t_scale = time_kernel_ms(torch, lambda: ops.csc_scale(G, Nc, colptr, rowidx, x, ws["genes"], ws["gkept"], ws["out_colptr"], ws["out_rowidx"], ws["out_x"]), 10)
        t_count = time_kernel_ms(torch, lambda: ops.csc_count(G, Nc, colptr, rowidx, x, ws["nt"]), 10)
        gf = {"metric": "gficf_cells_per_sec", "value": Nc / tg, "unit": "cells/s", "ms_per_pass": tg * 1e3,
              "config": {"workload": f"BASELINE config 3 shape: {G} genes x {Nc} cells synthetic UMI CSC (nnz={nnz}, SURVEY.md 8d density: "
                                     "clipped-lognormal stored entries per cell, Zipf genes drawn without replacement), "
                                     "gene filter 5 % + GF + ICF + L2, device-resident, compacted output",
                         "nnz": nnz, "nnz_per_cell_median": float((colptr[1:] - colptr[:-1]).double().median().item()),
                         "nnz_per_cell_max": int((colptr[1:] - colptr[:-1]).max().item())},
              "ms_per_pass_batches": [round(t * 1e3, 4) for t in tg_all], "ms_per_pass_is": "median of the batch means",
              "nnz": nnz, "kept_nnz": int(ws["out_colptr"][Nc]), "kept_genes": int(ws["gkept"][0]), "dtype": "f64",
              "roofline": {"bound": "hbm", "kernel": "whole pass (count + colptr + scale)",
                           "achieved": round(GFICF_BYTES_PER_NNZ * nnz / tg / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(GFICF_BYTES_PER_NNZ * nnz / tg / 1e9 / HBM_PEAK_GBS, 4),
                           "frac_of_copy_rate": round(GFICF_BYTES_PER_NNZ * nnz / tg / 1e9 / HBM_COPY_GBS, 4),
                           "traffic": (sum(pmc[kk]["hbm_bytes_per_launch"] for kk in ("gene_count", "nt_sum_table", "cell_kept_count", "scan_lookback", "scale_cells_lds") if kk in pmc)
                                       if all(kk in pmc for kk in ("gene_count", "cell_kept_count")) and pmc.get("gficf_nnz") == nnz else None),
                           "scale_kernel_ms": round(t_scale, 4), "count_exact_kernel_ms": round(t_count, 4),
                           "count_note": "count_exact_kernel_ms is gficf_csc_count_device (reads x: 12 B/nnz), the form the sharded and host entries use; the timed pass (gficf_csc_device) counts stored entries without reading x (about half that time, see the rocprof summary)",
                           "algorithmic_bytes_per_pass": GFICF_BYTES_PER_NNZ * nnz}}
        # the same pass with the result in the pointerB / pointerE form (cells compact inside their own input range: no global
        # positions, so the kept-count pass and its scan do not run — three launches instead of five), which is what the
        # device-resident chain hands on: t() of it, below, is an ordinary compact CSC again
        ops.gficf_csc_be(G, Nc, colptr, rowidx, x, 0.05, 1.0, None, ws)
        torch.cuda.synchronize()
        tb, tb_all = batches(lambda: ops.gficf_csc_be(G, Nc, colptr, rowidx, x, 0.05, 1.0, None, ws))
        ops.sync()
        be_end, be_ri, be_x = ws["out_end"].clone(), ws["out_rowidx"].clone(), ws["out_x"].clone()
        run()                                                   # the canonical result back in the workspace for what follows
        ops.sync()
        lens_be = be_end[:Nc] - colptr[:Nc]
        cell_b = torch.repeat_interleave(torch.arange(Nc, device=dev), lens_be)
        kn_b = int(cell_b.numel())
        pos_b = torch.arange(kn_b, device=dev) - ws["out_colptr"][:Nc][cell_b] + colptr[:Nc][cell_b]
        gf["roofline"]["frac_begin_end_form"] = round(GFICF_BYTES_PER_NNZ * nnz / tb / 1e9 / HBM_PEAK_GBS, 4)
        gf["begin_end_form"] = {"ms_per_pass": tb * 1e3, "ms_per_pass_batches": [round(t * 1e3, 4) for t in tb_all], "cells_per_sec": Nc / tb,
                                "roofline_frac": round(GFICF_BYTES_PER_NNZ * nnz / tb / 1e9 / HBM_PEAK_GBS, 4),
                                "achieved_GBps": round(GFICF_BYTES_PER_NNZ * nnz / tb / 1e9, 2),
                                "equals_canonical_result": bool(kn_b == int(ws["out_colptr"][Nc]) and torch.equal(be_ri[pos_b], ws["out_rowidx"][:kn_b])
                                                                and torch.equal(be_x[pos_b], ws["out_x"][:kn_b])),
                                "note": "gficf_csc_be_device: count + gene table + scale (3 launches); cell c's kept entries at [colptr[c], out_end[c]) of the "
                                        "output arrays — same entries, order and bits as the canonical compacted CSC (`value`), without the third read of "
                                        "rowidx that global output positions cost; read directly by gficf_csc_transpose_be_device / gficf_cluster_signatures_be_device"}
        del cell_b, pos_b, lens_be
        # next row N3: t(gficf), the PCA input (R/dimensinalityReduction.R:33), on the matrix just produced
        gk, kn = int(ws["gkept"][0]), int(ws["out_colptr"][Nc])
        tws = torch.zeros(ops.csc_transpose_workspace_bytes(gk, Nc), dtype=torch.uint8, device=dev)
        t_ptr = torch.zeros(gk + 1, dtype=torch.int64, device=dev)
        t_idx = torch.zeros(kn, dtype=torch.int32, device=dev)
        t_val = torch.zeros(kn, dtype=torch.float64, device=dev)
        run_t = lambda: ops.csc_transpose(gk, Nc, ws["out_colptr"], ws["out_rowidx"][:kn], ws["out_x"][:kn], t_ptr, t_idx, t_val, tws)
        t_tr = time_kernel_ms(torch, run_t, 10)
        order = torch.sort(ws["out_rowidx"][:kn].long(), stable=True)[1]
        cell = torch.repeat_interleave(torch.arange(Nc, device=dev, dtype=torch.int32), ws["out_colptr"][1:] - ws["out_colptr"][:-1])
        t_ptr2, t_idx2, t_val2 = torch.zeros_like(t_ptr), torch.zeros_like(t_idx), torch.zeros_like(t_val)
        run_tb = lambda: ops.csc_transpose_be(gk, Nc, colptr, be_end, be_ri, be_x, t_ptr2, t_idx2, t_val2, tws)
        t_trb = time_kernel_ms(torch, run_tb, 10)
        gf["begin_end_form"]["transpose_ms"] = round(t_trb, 4)
        gf["begin_end_form"]["transpose_equals_canonical"] = bool(torch.equal(t_ptr2, t_ptr) and torch.equal(t_idx2, t_idx) and torch.equal(t_val2, t_val))
        del t_ptr2, t_idx2, t_val2, be_end, be_ri, be_x
        gf["transpose"] = {"ms": round(t_tr, 4), "entries": kn, "cells_per_sec": Nc / (t_tr * 1e-3),
                           "algorithmic_GBps": round(28 * kn / t_tr / 1e6, 1),
                           "note": "t(gficf): kept genes x cells CSC -> cells x genes CSC, 28 B/entry (4 count + 12 read + 12 written)",
                           "checked_vs_stable_sort": bool(torch.equal(t_idx, cell[order]) and torch.equal(t_val, ws["out_x"][:kn][order]))}
        del order, cell, tws, t_idx, t_val
        # next row N3, first half: cluster signatures (R/clustCells.R:121-123) of the same matrix, 30 synthetic clusters
        n_cl = 30
        cl = (torch.arange(Nc, device=dev, dtype=torch.int64) * 2654435761 % n_cl).to(torch.int32)
        sig = torch.zeros((n_cl, gk), dtype=torch.float64, device=dev)
        run_s = lambda: (sig.zero_(), ops.cluster_signatures(gk, Nc, ws["out_colptr"], ws["out_rowidx"][:kn], ws["out_x"][:kn], cl, n_cl, sig))
        t_sig = time_kernel_ms(torch, run_s, 10)
        want = torch.zeros((n_cl, gk), dtype=torch.float64, device=dev)
        cell_of = torch.repeat_interleave(torch.arange(Nc, device=dev), ws["out_colptr"][1:] - ws["out_colptr"][:-1])
        want.index_put_((cl[cell_of].long(), ws["out_rowidx"][:kn].long()), ws["out_x"][:kn], accumulate=True)
        gf["cluster_signatures"] = {"ms": round(t_sig, 4), "clusters": n_cl, "algorithmic_GBps": round(12 * kn / t_sig / 1e6, 1),
                                    "note": "G x C sums of gficf[, cluster == c] (12 B/entry read; cells grouped by cluster, sums kept in LDS per workgroup)",
                                    "checked_vs_torch": bool(torch.allclose(sig, want, rtol=1e-9, atol=1e-12))}
        del cell_of, want, sig
        hcp = hri = hx = None
        if not a.no_cpu_baseline or not a.no_host_gficf:
            hcp, hri, hx = colptr.cpu().numpy(), rowidx.cpu().numpy(), x.cpu().numpy()
        ref = None
        if not a.no_cpu_baseline:
            import oracle

            t1 = time.perf_counter()
            ref = oracle.gficf_csc(G, Nc, hcp, hri, hx, 0.05, 1.0)
            tc1 = time.perf_counter() - t1
            cores = os.cpu_count() or 1
            t1 = time.perf_counter()
            ref_mt = oracle.gficf_csc(G, Nc, hcp, hri, hx, 0.05, 1.0, threads=cores)
            tc = time.perf_counter() - t1
            kn = int(ws["out_colptr"][Nc])
            gf["cpu_baseline"] = {"value": Nc / tc, "unit": "cells/s", "cores": cores, "kind": "port",
                                  "sample": "the full matrix, one pass of the oracle's multi-threaded restatement of R/gficf.R (cells cut into "
                                            "ranges of equal stored entries, one host thread each; same bits as the single-threaded one)",
                                  "single_thread_value": Nc / tc1,
                                  "single_thread_note": "the reference path itself is single-threaded R on the Matrix package (cannot run here)",
                                  "threads_give_same_bits": bool(np.array_equal(ref["x"], ref_mt["x"]) and np.array_equal(ref["nt"], ref_mt["nt"])),
                                  "gpu_over_cpu": (Nc / tg) / (Nc / tc)}
            gf["checked_vs_oracle"] = bool(kn == len(ref["x"]) and np.array_equal(ws["out_rowidx"][:kn].cpu().numpy(), ref["rowidx"])
                                           and np.allclose(ws["out_x"][:kn].cpu().numpy(), ref["x"], rtol=1e-6, atol=1e-6))
            del ref_mt
        if not a.no_host_gficf:
            # end to end through the host C ABI (what `.Call("_gficf_gficf_csc", ...)` binds): plan (H2D of @p / @i / @x, count, filter) +
            # finish (scale, D2H of the compacted matrix), pageable host buffers.  PCIe-inclusive: reported, never `value`.
            try:
                gf["host_abi"] = host_gficf_figure(self.gficf_amd, self.local_rank, G, Nc, hcp, hri, hx, ref)
            except Exception as ex:  # pragma: no cover
                gf["host_abi"] = {"error": f"{type(ex).__name__}: {ex}"}
        self.out["gficf"] = gf

    def leg_knn(self):
        self.out["knn"] = bench_knn(self.torch, self.ops, self.args)


def host_gficf_figure(gficf_amd, device, G, N, colptr, rowidx, x, ref=None, reps=2):
    """gficf_normalize_csc_host_plan + _finish on host arrays (int32 colptr when it fits, as a dgCMatrix holds it)."""
    import ctypes

    from gficf_amd import _lib

    L = _lib.load()
    ctx = gficf_amd.default_context(device)
    nnz = int(len(rowidx))
    i64 = nnz >= 2**31
    cp = np.ascontiguousarray(colptr, dtype=np.int64 if i64 else np.int32)
    vp = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
    gk, nk = ctypes.c_int64(0), ctypes.c_int64(0)
    ts, out = [], None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        rc = L.gficf_normalize_csc_host_plan(ctx.handle, G, N, vp(cp), 1 if i64 else 0, vp(rowidx), vp(x), ctypes.c_double(0.05), ctypes.c_double(1.0), None,
                                             ctypes.byref(gk), ctypes.byref(nk))
        if rc != 0:
            raise RuntimeError(f"gficf_normalize_csc_host_plan returned {rc}: {_lib.last_error()}")
        t_plan = time.perf_counter() - t0
        keep = np.empty(G, dtype=np.uint8)
        nt = np.empty(G, dtype=np.int64)
        w = np.empty(G, dtype=np.float64)
        ocp = np.empty(N + 1, dtype=cp.dtype)
        ori = np.empty(nk.value, dtype=np.int32)
        ox = np.empty(nk.value, dtype=np.float64)
        rc = L.gficf_normalize_csc_host_finish(ctx.handle, vp(keep), vp(nt), vp(w), vp(ocp), vp(ori), vp(ox))
        if rc != 0:
            raise RuntimeError(f"gficf_normalize_csc_host_finish returned {rc}: {_lib.last_error()}")
        ts.append((time.perf_counter() - t0, t_plan))
        out = (ori, ox)
    t_all, t_plan = min(ts[1:])                                         # (the first call grows the context's pool)
    res = {"ms_per_call": t_all * 1e3, "ms_plan": t_plan * 1e3, "ms_finish": (t_all - t_plan) * 1e3, "cells_per_sec": N / t_all,
           "bytes_in": int(cp.nbytes + rowidx.nbytes + x.nbytes), "bytes_out": int(12 * nk.value + cp.nbytes + 17 * G),
           "kept_nnz": int(nk.value), "kept_genes": int(gk.value),
           "note": "gficf_normalize_csc_host_plan + _finish: pageable host arrays in, freshly allocated host arrays out (as the R glue allocates them), "
                   "device scratch from the context pool, PCIe both ways; the faster of 2 calls after a first one that grows the pool"}
    if ref is not None:
        res["checked_vs_oracle"] = bool(len(ref["x"]) == nk.value and np.array_equal(out[0], ref["rowidx"]) and np.allclose(out[1], ref["x"], rtol=1e-6, atol=1e-6))
    ctx_trim = getattr(L, "gficf_ctx_trim", None)
    if ctx_trim is not None:
        ctx_trim(ctx.handle)                                            # (a GB of pooled scratch handed back before the next leg)
    return res


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and peer mappings across processes need on this driver (before any HIP call)
    if args.traffic_child:
        return traffic_child(args)
    if args.peer_child:
        return peer_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and never touches a
        # GPU itself; rank 0 prints its JSON lines straight to our stdout.  (Under torch.distributed.run the rank
        # environment is already there and this branch is not taken.)  The launcher's own limit follows the wall budget.
        from gficf_amd import launch

        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                            need_gpus=None if (args.rehearse_one_gpu or os.environ.get("GFICF_BENCH_RANKS_SHARE_GPU0")) else args.gpus,
                                            timeout_s=float(os.environ.get("GFICF_BENCH_LAUNCH_TIMEOUT", str(args.budget_s + 150.0)))))
    B = Bench(args)
    world, rank, ops = B.world, B.rank, B.ops
    extras = not args.no_extras
    B.start_watchdog()
    B.progress("value")
    t0 = time.monotonic()
    B.leg_value()
    B.leg_seconds["value"] = time.monotonic() - t0
    B.legs_done.append("value")
    # N > 1: from here on the line holds `value`, its roofline, the exchange and the oracle check: printed NOW, and again after every leg
    B.progress()
    B.emit()
    if args.pipeline or (extras and (world > 1 or not B.strong)):
        B.run_leg("pipelined", B.leg_pipelined)
    if world > 1 and extras:
        # ---- the rest of the scaling answer, in this one run (same box, same processes, same clocks)
        B.run_leg("other_ids", B.leg_other_ids)
        B.run_leg("single_gpu_step", B.leg_single_gpu_step)
        if not args.no_chain:
            B.run_leg("chain", B.leg_chain)
        if not args.no_peer:
            B.run_leg("peer", B.leg_peer)
        B.efficiency()
    if world == 1 and extras and B.strong and B.N_total >= 100_000:
        B.run_leg("spatial_ids", B.leg_spatial_ids_one_gpu)
    if B.distinct:
        ops.set_jaccard_distinct(False)
    if world == 1 and extras and not B.strong:
        B.run_leg("cpu_baseline", B.leg_cpu_baseline)
        B.run_leg("stress", B.leg_stress)
        B.run_leg("host_abi", B.leg_host_abi)
        if not args.no_gficf:
            B.run_leg("gficf", B.leg_gficf)
        if not args.no_knn:
            B.run_leg("knn", B.leg_knn)
    if world > 1 and extras and not args.no_gficf:
        B.run_leg("gficf", B.leg_gficf_sharded)
    if rank == 0 and not B.guard["done"]:
        B.guard["done"] = True
        print(B.line(), flush=True)
    if world > 1:
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
