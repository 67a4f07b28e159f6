"""CPU, world_size 2, gloo: the N > 1 path of gficf_amd.dist (cell-block sharding, the
all-gather of kNN table rows, the all-reduce of per-gene counts, output placement) with a CPU
test double standing in for the HIP stage kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, k, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops_double import CpuOpsDouble
        from gficf_amd import synth
        from gficf_amd.dist import GficfShard, JaccardShard, shard_bounds

        ops = CpuOpsDouble()
        # ---- Jaccard: every rank holds only its block of the kNN matrix
        mat = synth.knn_windowed(N, k, seed=3)
        b, e = shard_bounds(N, world, rank)
        idx_local = torch.from_numpy(np.ascontiguousarray(mat[b:e].T))
        sh = JaccardShard(ops, N, k, with_u=True)
        assert (sh.b, sh.e) == (b, e)
        out = sh.step(idx_local)
        out2 = sh.step(idx_local).clone()            # buffers are reused across steps
        assert torch.equal(out, out2)
        np.save(os.path.join(outdir, f"jac_{rank}.npy"), out.numpy())
        np.save(os.path.join(outdir, f"u_{rank}.npy"), sh.u.numpy())
        # the gathered table must be identical on all ranks
        tabs = [torch.zeros_like(sh.table) for _ in range(world)]
        dist.all_gather(tabs, sh.table)
        assert all(torch.equal(tabs[0], t) for t in tabs)
        # ---- kNN in front of it: points sharded by the same cell blocks, the index block feeds the Jaccard shard
        from gficf_amd.dist import KnnShard

        rng = np.random.default_rng(17)
        X = rng.normal(size=(N, 6)) * 3.0
        ks = KnnShard(ops, N, 6, 9, "manhattan")
        assert (ks.b, ks.e) == (b, e)
        kidx = ks.step(torch.from_numpy(np.ascontiguousarray(X[b:e].T)))
        np.save(os.path.join(outdir, f"knn_{rank}.npy"), kidx.numpy())
        pts = [torch.zeros_like(ks.points) for _ in range(world)]
        dist.all_gather(pts, ks.points)
        assert all(torch.equal(pts[0], t) for t in pts)
        sh2 = JaccardShard(ops, N, 8)
        np.save(os.path.join(outdir, f"kjac_{rank}.npy"), sh2.step(kidx[1:].contiguous()).numpy())
        # ---- GF-ICF: every rank holds a block of cells (columns)
        G, Nc = 300, 101
        cp, ri, x = synth.counts_csc(G, Nc, seed=5)
        cb, ce = shard_bounds(Nc, world, rank)
        lcp = torch.from_numpy((cp[cb:ce + 1] - cp[cb]).astype(np.int64))
        lri = torch.from_numpy(ri[cp[cb]:cp[ce]].copy())
        lx = torch.from_numpy(x[cp[cb]:cp[ce]].copy())
        gs = GficfShard(ops, G, Nc, ce - cb, int(lri.numel()))
        ws = gs.step(lcp, lri, lx, 0.05, 1.0)
        n = int(ws["out_colptr"][ce - cb])
        np.savez(os.path.join(outdir, f"gf_{rank}.npz"), colptr=ws["out_colptr"].numpy(), rowidx=ws["out_rowidx"][:n].numpy(),
                 x=ws["out_x"][:n].numpy(), keep=ws["keep"].numpy(), nt=ws["nt"].numpy(), w=ws["w"].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,k,world", [(1001, 15, 2), (640, 30, 2), (1003, 15, 8)])
def test_sharded_path_world2_gloo(tmp_path, N, k, world):
    """(world 8: the rank count of the driver's largest scaling run, rehearsed on the CPU over gloo — blocks of unequal size
    (1003 = 7 x 126 + 121), eight all-gather chunks, eight GF-ICF column blocks of a 101-cell matrix.)"""
    import oracle
    from gficf_amd import synth
    from gficf_amd.dist import shard_bounds

    mp.spawn(_worker, args=(world, _free_port(), N, k, str(tmp_path)), nprocs=world, join=True)
    mat = synth.knn_windowed(N, k, seed=3)
    want, wu = oracle.jaccard(mat, nthreads=2)
    got = np.concatenate([np.load(tmp_path / f"jac_{r}.npy") for r in range(world)], axis=1).T
    gu = np.concatenate([np.load(tmp_path / f"u_{r}.npy") for r in range(world)])
    assert np.array_equal(gu, wu)
    assert np.array_equal(got, want)
    # kNN blocks == the float64 restatement on the whole matrix; chained Jaccard == oracle on those ids
    from oracle import oracle_np

    X = np.random.default_rng(17).normal(size=(N, 6)) * 3.0
    nidx, _ = oracle_np.knn_np(X.astype(np.float32), 9, "manhattan")
    kgot = np.concatenate([np.load(tmp_path / f"knn_{r}.npy") for r in range(world)], axis=1).T
    assert np.array_equal(kgot, nidx)
    kwant, _ = oracle.jaccard(np.ascontiguousarray(nidx[:, 1:]), nthreads=2)
    kj = np.concatenate([np.load(tmp_path / f"kjac_{r}.npy") for r in range(world)], axis=1).T
    assert np.array_equal(kj, kwant)
    # GF-ICF: concatenated blocks == single-shot oracle
    G, Nc = 300, 101
    cp, ri, x = synth.counts_csc(G, Nc, seed=5)
    ref = oracle.gficf_csc(G, Nc, cp, ri, x, 0.05, 1.0)
    parts = [np.load(tmp_path / f"gf_{r}.npz") for r in range(world)]
    assert all(np.array_equal(p["keep"].astype(bool), ref["keep"]) for p in parts)
    assert all(np.array_equal(p["nt"][ref["keep"]], ref["nt"][ref["keep"]]) for p in parts)
    assert np.array_equal(np.concatenate([p["rowidx"] for p in parts]), ref["rowidx"])
    assert np.allclose(np.concatenate([p["x"] for p in parts]), ref["x"], rtol=1e-12, atol=1e-15)
    offs = 0
    for r, p in enumerate(parts):
        cb, ce = shard_bounds(Nc, world, r)
        assert np.array_equal(p["colptr"] + offs, ref["colptr"][cb:ce + 1])
        offs += p["colptr"][-1]


def test_shard_bounds_cover_and_are_equal_pitch():
    from gficf_amd.dist import rows_per_rank, shard_bounds

    for n in (0, 1, 7, 100, 1001, 100000):
        for world in (1, 2, 3, 8):
            rpr = rows_per_rank(n, world)
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for r, (b, e) in enumerate(blocks):
                assert b == min(n, r * rpr) and 0 <= e - b <= rpr
                if r:
                    assert b == blocks[r - 1][1]


def test_nnz_balanced_column_blocks():
    from gficf_amd import synth
    from gficf_amd.dist import shard_bounds_by_nnz

    cp, _, _ = synth.counts_csc(400, 1000, seed=3)
    for world in (1, 2, 3, 8):
        blocks = shard_bounds_by_nnz(cp, world)
        assert blocks[0][0] == 0 and blocks[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        share = [cp[e] - cp[b] for b, e in blocks]
        assert sum(share) == cp[-1]
        # no block is further from the ideal share than the largest cell
        assert max(abs(s - cp[-1] / world) for s in share) <= np.diff(cp).max() * 2
    # degenerate inputs: empty matrix, more ranks than cells, all entries in one cell
    assert shard_bounds_by_nnz([0], 4) == [(0, 0)] * 4
    assert shard_bounds_by_nnz([0, 5], 3)[-1][1] == 1 and sum(e - b for b, e in shard_bounds_by_nnz([0, 5], 3)) == 1
    assert sum(e - b for b, e in shard_bounds_by_nnz([0, 0, 0, 9, 9], 2)) == 4


def _halo_worker(rank, world, port, N, k, perm, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops_double import CpuOpsDouble
        from gficf_amd import synth
        from gficf_amd.dist import JaccardShard, shard_bounds

        ops = CpuOpsDouble()
        mat = synth.knn_windowed(N, k, seed=5, perm_seed=43 if perm else None)
        b, e = shard_bounds(N, world, rank)
        idx_local = torch.from_numpy(np.ascontiguousarray(mat[b:e].T))
        res = {}
        for form in ("allgather", "halo"):
            sh = JaccardShard(ops, N, k, with_u=True, exchange=form)
            out = sh.step(idx_local).clone()
            res[form] = (out, sh.u.clone(), sh.rows_received, sh.bytes_received)
        assert torch.equal(res["halo"][0], res["allgather"][0]) and torch.equal(res["halo"][1], res["allgather"][1])
        np.save(os.path.join(outdir, f"halo_{rank}.npy"), res["halo"][0].numpy())
        np.save(os.path.join(outdir, f"halo_rows_{rank}.npy"), np.array([res["halo"][2], res["allgather"][2], res["halo"][3], res["allgather"][3]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("perm", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_matches_allgather_gloo(tmp_path, perm, world):
    """The halo form of the exchange (a rank fetches only the remote rows its block names) gives the bits of the
    all-gather form and of the oracle — on spatially ordered ids, where it moves a few hundred rows instead of the other
    ranks' whole blocks, and on scrambled ids, where it moves nearly all of them."""
    import oracle
    from gficf_amd import synth

    N, k = 1501, 15
    mp.spawn(_halo_worker, args=(world, _free_port(), N, k, perm, str(tmp_path)), nprocs=world, join=True)
    mat = synth.knn_windowed(N, k, seed=5, perm_seed=43 if perm else None)
    want, _ = oracle.jaccard(mat, nthreads=2)
    got = np.concatenate([np.load(tmp_path / f"halo_{r}.npy") for r in range(world)], axis=1).T
    assert np.array_equal(got, want)
    for r in range(world):
        halo_rows, ag_rows, halo_bytes, ag_bytes = np.load(tmp_path / f"halo_rows_{r}.npy")
        assert ag_rows == N - len(range(*__import__("gficf_amd.dist", fromlist=["shard_bounds"]).shard_bounds(N, world, r)))
        if perm:
            assert 0.8 * ag_rows <= halo_rows <= ag_rows           # scrambled ids: (nearly) every remote row is named
        else:
            assert halo_rows <= 2 * 2 * 100 and halo_rows < 0.5 * ag_rows    # ordered ids: the window either side of the block's two seams


def _local_worker(rank, world, port, N, k, perm, cap, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops_double import CpuOpsDouble
        from gficf_amd import GficfError, synth
        from gficf_amd.dist import JaccardHaloShard, shard_bounds

        ops = CpuOpsDouble()
        mat = synth.knn_windowed(N, k, seed=5, perm_seed=43 if perm else None)
        b, e = shard_bounds(N, world, rank)
        idx_local = torch.from_numpy(np.ascontiguousarray(mat[b:e].T))
        sh = JaccardHaloShard(ops, N, k, with_u=True, cap=cap)
        assert sh.n_ext == (e - b) + world * sh.cap
        out = sh.step(idx_local).clone()
        over = False
        try:
            sh.sync()
        except GficfError as ex:
            over = ex.status == "GFICF_ERR_CAPACITY"
        np.save(os.path.join(outdir, f"loc_{rank}.npy"), out.numpy())
        np.save(os.path.join(outdir, f"locmeta_{rank}.npy"), np.array([int(over), sh.rows_named_outside(), sh.bytes_received]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_local_id_halo_shard_matches_oracle_gloo(tmp_path, world):
    """JaccardHaloShard: every rank builds its block's edges on a sub-problem in local ids (own cells + the rows it names,
    fetched through fixed-capacity request slots: two all-to-alls with equal splits) — the oracle's bits on ordered ids."""
    import oracle
    from gficf_amd import synth

    N, k = 1501, 15
    mp.spawn(_local_worker, args=(world, _free_port(), N, k, False, 256, str(tmp_path)), nprocs=world, join=True)
    want, _ = oracle.jaccard(synth.knn_windowed(N, k, seed=5, perm_seed=None), nthreads=2)
    got = np.concatenate([np.load(tmp_path / f"loc_{r}.npy") for r in range(world)], axis=1).T
    assert np.array_equal(got, want)
    for r in range(world):
        over, named, moved = np.load(tmp_path / f"locmeta_{r}.npy")
        assert over == 0 and 0 < named <= 2 * 2 * 100
        assert moved == (world - 1) * 256 * 4 * (1 + k)


def test_local_id_halo_shard_reports_overflow_on_scrambled_ids(tmp_path):
    """Ids without locality name nearly every remote row: more than the request slots hold -> GFICF_ERR_CAPACITY at the sync
    (the caller then takes the all-gather form)."""
    world = 2
    mp.spawn(_local_worker, args=(world, _free_port(), 1501, 15, True, 64, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"locmeta_{r}.npy")[0] == 1


def _collective_sync_worker(rank, world, port, N, k, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops_double import CpuOpsDouble
        from gficf_amd import GficfError, synth
        from gficf_amd.dist import JaccardHaloShard, shard_bounds

        mat = synth.knn_windowed(N, k, seed=5, perm_seed=None)
        b1, e1 = shard_bounds(N, world, 1)
        # 120 cells in the middle of rank 0's block each name one more distinct row of rank 1's block: only rank 0's request slots overflow
        for t in range(120):
            mat[200 + t, 0] = b1 + 150 + t + 1
        b, e = shard_bounds(N, world, rank)
        sh = JaccardHaloShard(CpuOpsDouble(), N, k, cap=128)
        sh.step(torch.from_numpy(np.ascontiguousarray(mat[b:e].T)))
        msg = ""
        try:
            sh.sync(collective=True)
        except GficfError as ex:
            msg = f"{ex.status}|{ex}"
        # a clean step afterwards: nobody raises, and the collective is still matched on every rank
        sh2 = JaccardHaloShard(CpuOpsDouble(), N, k, cap=512)
        sh2.step(torch.from_numpy(np.ascontiguousarray(mat[b:e].T)))
        sh2.sync(collective=True)
        open(os.path.join(outdir, f"csync_{rank}.txt"), "w").write(msg)
    finally:
        dist.destroy_process_group()


def test_collective_sync_raises_the_same_error_on_every_rank(tmp_path):
    """a deferred error of the halo form is local to the rank whose block overflowed; a caller that switches forms on
    that rank alone would deadlock the job.  sync(collective=True) all-reduces the status: every rank raises, or none."""
    world = 3
    mp.spawn(_collective_sync_worker, args=(world, _free_port(), 1501, 15, str(tmp_path)), nprocs=world, join=True)
    msgs = [open(tmp_path / f"csync_{r}.txt").read() for r in range(world)]
    assert all(m.startswith("GFICF_ERR_CAPACITY|") for m in msgs), msgs
    assert "test double" in msgs[0]                                   # rank 0 saw it itself
    assert all("rank 0 reported GFICF_ERR_CAPACITY" in m for m in msgs[1:])


def _format_worker(rank, world, port, outdir, differ):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if differ and rank == 1:
        os.environ["GFICF_JACCARD_DUAL"] = "0"          # one rank started under another environment: it would lay 100 k x 50 rows out in 32 words, not 64
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gficf_amd
        from gficf_amd.dist import assert_same_format

        res = "same"
        try:
            assert_same_format(gficf_amd.HipOps, 100_000, 50)      # (row_words / kpad / packed_words are static: pure host functions of the library)
        except RuntimeError as ex:
            res = str(ex)
        with open(os.path.join(outdir, f"fmt_{rank}.txt"), "w") as f:
            f.write(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("differ", [False, True])
def test_ranks_compare_their_table_format_before_the_first_step(tmp_path, differ):
    """the table format is a function of (N, k) only while every rank has the same library and the same
    GFICF_JACCARD_* environment — exchanged and asserted once; a rank that differs makes EVERY rank raise, naming what differs."""
    world = 2
    mp.spawn(_format_worker, args=(world, _free_port(), str(tmp_path), differ), nprocs=world, join=True)
    msgs = [open(tmp_path / f"fmt_{r}.txt").read() for r in range(world)]
    if not differ:
        assert msgs == ["same", "same"]
    else:
        for m in msgs:
            assert "row_words" in m and "crc32(GFICF_JACCARD_* environment)" in m and "do not agree" in m
