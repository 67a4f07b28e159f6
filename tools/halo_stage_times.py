#!/usr/bin/env python3
"""Per-rank compute of the N > 1 Jaccard step, measured on ONE GPU for one rank of a P-rank job (weak scaling, 100 000
cells per rank, k = 30, N_total = P x 100 000): the stages of the two exchange forms, each timed with HIP events, the
exchange itself replaced by device copies of the right size (so only the kernels are timed; the bytes the real exchange
moves are printed next to them).  tools/project_scaling.py turns the output into the table of DESIGN.md.
Usage: python tools/halo_stage_times.py [P ...]   -> JSON lines"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gficf_amd
from gficf_amd import synth
from gficf_amd.dist import rows_per_rank, shard_bounds

ops = gficf_amd.HipOps(0)
ops.set_jaccard_distinct(True)          # the sequence bench.py and the host entries run: no duplicate scan in the ingests
n_per, k = 100_000, 30


_blk = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")


def t_ms(fn, iters=30):
    """GPU time per launch of `fn`, back to back: the launches are enqueued BEHIND a few milliseconds of other work (so the host,
    which needs ~10 us of Python + ctypes per call, is far ahead of the device by the time the first one starts) and timed by two
    events on the stream.  Without the blocker a kernel of a few microseconds measures the host's enqueue rate, not the device
    (round 3's stage times for plan / serve / ingest were inflated that way; rocprofv3's kernel trace agrees with this form)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(12):
        _blk.add_(1)                                                # ~12 x 0.1 ms: the device stays busy while the host enqueues
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for P in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    N = n_per * P
    r = P // 2                                                    # a middle rank
    b, e = shard_bounds(N, P, r)
    rpr = rows_per_rank(N, P)
    i32 = dict(dtype=torch.int32, device="cuda")
    res = {"P": P, "N_total": N, "k": k, "rank": r}
    for ids in ("spatial", "permuted"):
        mat = synth.knn_windowed(N, k, seed=42, perm_seed=None if ids == "spatial" else 43)
        idx = torch.from_numpy(np.ascontiguousarray(mat[b:e].T)).cuda()
        nl = e - b
        # ---- all-gather form: ingest own block, pack, [all-gather], unpack the other blocks, edges on the full table
        roww = ops.row_words(N, k)
        table = torch.zeros((P * rpr, roww), **i32)
        out = torch.zeros((3, nl * k), dtype=torch.float64, device="cuda")
        full = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
        ops.jaccard_ingest(full, N, k, N, table[:N])            # the other ranks' rows, so that the edge kernel gathers real rows
        ag = {"ingest_ms": t_ms(lambda: ops.jaccard_ingest(idx, nl, k, N, table[b:e]))}
        pw = ops.packed_words(N, k)
        ag["wire_bytes_per_row"] = 4 * min(pw, roww)
        if P > 1 and pw < roww:
            packed = torch.zeros((P * rpr, pw), **i32)
            ag["pack_ms"] = t_ms(lambda: ops.jaccard_pack_rows(table[b:e], nl, k, N, packed[b:e]))
            ops.jaccard_pack_rows(table[:N], N, k, N, packed[:N])

            def unpack_others():                      # as dist.JaccardShard does it: the rows in front of the own block, the rows behind it
                for lo, hi in ((0, b), (e, N)):
                    if hi > lo:
                        ops.jaccard_unpack_rows(packed[lo:hi], hi - lo, k, N, table[lo:hi])
            ag["unpack_ms"] = t_ms(unpack_others, 10)
        ag["edges_ms"] = t_ms(lambda: ops.jaccard_edges(table, N, k, b, e, out, None))
        ag["table_row_bytes"] = 4 * roww
        ag["bytes_received"] = (N - nl) * ag["wire_bytes_per_row"]
        ag["compute_ms"] = sum(v for kk, v in ag.items() if kk.endswith("_ms"))
        res[f"allgather_{ids}"] = ag
        del full
        # ---- halo form on local ids
        cap = max(64, min(8192, ((1 << 17) - 1 - rpr) // P))
        n_ext = nl + P * cap
        ws = torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda")
        req_out, rows_in = torch.zeros(P * cap, **i32), torch.zeros(P * cap * k, **i32)
        ops.halo_plan(idx, nl, k, N, b, P, rpr, cap, ws, req_out)
        fits = True
        try:
            ops.sync()
        except gficf_amd.GficfError:
            fits = False
        named = int((req_out != 0).sum())
        h = {"fits": fits, "rows_named_outside": named, "cap": cap, "n_ext": n_ext}
        if fits:
            # the rows the other ranks would serve (raw ids), straight from the matrix
            rq = req_out.cpu().numpy().astype(np.int64)
            rin = np.zeros((P * cap, k), dtype=np.int32)
            rin[rq != 0] = mat[rq[rq != 0] - 1]
            rows_in.copy_(torch.from_numpy(rin.reshape(-1)).cuda())
            idx_ext, l2g = torch.zeros((k, n_ext), **i32), torch.zeros(n_ext, **i32)
            tab2 = torch.zeros((n_ext, ops.row_words(n_ext, k)), **i32)
            # what the other ranks would ask of this one: as many of its own rows as it names of theirs (same slots)
            rqi = np.zeros(P * cap, dtype=np.int32)
            rqi[rq != 0] = b + 1 + (np.arange(int((rq != 0).sum())) * 7919) % nl
            req_in, rows_out = torch.from_numpy(rqi).cuda(), torch.zeros(P * cap * k, **i32)
            req_tmp = torch.zeros(P * cap, **i32)
            h["plan_ms"] = t_ms(lambda: ops.halo_plan(idx, nl, k, N, b, P, rpr, cap, ws, req_out))
            h["serve_ms"] = t_ms(lambda: ops.halo_serve(idx, nl, k, b, req_in, rows_out))
            h["relabel_unfused_us"] = 1e3 * t_ms(lambda: ops.halo_relabel(idx, nl, k, N, b, P, rpr, cap, ws, req_out, rows_in, idx_ext, l2g))
            h["ingest_unfused_us"] = 1e3 * t_ms(lambda: ops.jaccard_ingest_local(idx_ext, n_ext, k, tab2))
            h["serve_alone_us"] = 1e3 * h.pop("serve_ms")
            # round 4: serve + the own cells' rows in ONE launch between the exchanges, the halo slots in use behind the second one
            h["serve_ingest_ms"] = t_ms(lambda: ops.halo_serve_ingest(idx, nl, k, N, b, P, rpr, cap, ws, req_out, req_in, rows_out, tab2, l2g))
            h["slots_ms"] = t_ms(lambda: ops.halo_ingest_slots(idx, nl, k, N, b, P, rpr, cap, ws, req_out, rows_in, tab2, l2g))
            h["edges_ms"] = t_ms(lambda: ops.jaccard_edges_mapped(tab2, n_ext, k, nl, b, l2g, out, None))
            h["table_row_bytes"] = 4 * ops.row_words(n_ext, k)
            h["bytes_received"] = (P - 1) * cap * 4 * (1 + k)
            h["compute_ms"] = sum(v for kk, v in h.items() if kk.endswith("_ms"))
            # the whole chain back to back on one stream (launch gaps included), exchange replaced by two device copies
            def chain():
                ops.halo_plan(idx, nl, k, N, b, P, rpr, cap, ws, req_out)
                req_tmp.copy_(req_out)                      # (stands for the first all-to-all's local part)
                ops.halo_serve_ingest(idx, nl, k, N, b, P, rpr, cap, ws, req_out, req_in, rows_out, tab2, l2g)
                req_tmp.copy_(req_out)                      # (stands for the second all-to-all's local part)
                ops.halo_ingest_slots(idx, nl, k, N, b, P, rpr, cap, ws, req_out, rows_in, tab2, l2g)
                ops.jaccard_edges_mapped(tab2, n_ext, k, nl, b, l2g, out, None)
            h["chain_ms"] = t_ms(chain)
            want, _ = __import__("oracle").jaccard_cells(mat, b, b + 512, nthreads=os.cpu_count() or 1)
            out_halo = out[:, :512 * k].cpu().numpy().T.copy()
            out.zero_()
            # the peer halo form (gficf_multi_jaccard_halo_device): nothing exchanged — the whole table in ONE launch behind the plan, the rows
            # named outside read in the owners' blocks of ids (here: all on this GPU; over xGMI one dependent remote read more)
            owners = [torch.from_numpy(np.ascontiguousarray(mat[o * rpr:min(N, (o + 1) * rpr)].T)).cuda() for o in range(P)]
            tab3, l2g3 = torch.zeros_like(tab2), torch.zeros_like(l2g)
            h["peer_ingest_us"] = 1e3 * t_ms(lambda: ops.halo_ingest_peer(idx, nl, k, N, b, P, rpr, cap, ws, req_out, owners, tab3, l2g3))
            def peer_chain():
                ops.halo_plan(idx, nl, k, N, b, P, rpr, cap, ws, req_out)
                ops.halo_ingest_peer(idx, nl, k, N, b, P, rpr, cap, ws, req_out, owners, tab3, l2g3)
                ops.jaccard_edges_mapped(tab3, n_ext, k, nl, b, l2g3, out, None)
            h["peer_chain_us"] = 1e3 * t_ms(peer_chain)
            h["peer_checked_vs_oracle"] = bool(np.array_equal(out[:, :512 * k].cpu().numpy().T, want))
            del owners, tab3, l2g3
            h["checked_vs_oracle"] = bool(np.array_equal(out_halo, want))
        res[f"halo_{ids}"] = h
        ops.sync()
        del table, out, idx
        torch.cuda.empty_cache()
    print(json.dumps(res), flush=True)
