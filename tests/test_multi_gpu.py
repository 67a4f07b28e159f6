"""GPU parity of the multi-GPU host entries (gficf_multi_* of the C ABI) and of the compact host return.
One GPU is enough: a device may be named more than once, one cell block each, which runs the whole sharded
sequence (per-block upload / ingest, table exchange, per-block edges, per-block download; per-block count, summed
gene counts, per-block scale) — bit-equal to the single-device entries and to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import gficf_amd
import oracle
from gficf_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]])
@pytest.mark.parametrize("N,k", [(5001, 30), (700, 100), (3, 2), (70000, 15), (140000, 30)])
def test_jaccard_host_multi_equals_single_device_and_oracle(devices, N, k):
    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N) if N > 300 else synth.knn_uniform(N, k)
    want = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False, devices=devices)
    assert np.array_equal(got, want)
    if N <= 70000:
        assert np.array_equal(got, oracle.jaccard(mat, nthreads=8)[0])
    # ids as doubles, and a leading dimension larger than N (a block of a bigger R matrix)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat.astype(np.float64), False, devices=devices), want)


def test_jaccard_host_multi_errors_and_empty():
    mc = [0, 0]
    mat = synth.knn_windowed(2000, 15)
    bad = mat.copy()
    bad[1500, 3] = 2001                       # lands in the second block
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.rcpp_parallel_jaccard_coef(bad, False, devices=mc)
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False, devices=mc), oracle.jaccard(mat, nthreads=4)[0])   # usable afterwards
    assert gficf_amd.rcpp_parallel_jaccard_coef(np.zeros((0, 5), dtype=np.int32), False, devices=mc).shape == (0, 3)
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.MultiContext([0, 99])


def test_jaccard_host_multi_without_peer_access_path():
    """GFICF_HIP_NO_PEER: the form taken when the devices cannot reach each other's memory (every device ingests the whole matrix)."""
    code = (
        "import numpy as np, gficf_amd, oracle\n"
        "from gficf_amd import synth\n"
        "for N, k in ((5001, 30), (900, 50)):\n"
        "    mat = synth.knn_windowed(N, k, seed=N)\n"
        "    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False, devices=[0, 0, 0]), oracle.jaccard(mat, nthreads=4)[0])\n"
        "print('ok')\n")
    env = dict(os.environ, GFICF_HIP_NO_PEER="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_env_variable_selects_the_multi_path():
    code = (
        "import numpy as np, gficf_amd, oracle\n"
        "from gficf_amd import synth, api\n"
        "assert api.env_devices() == [0, 0]\n"
        "mat = synth.knn_windowed(3000, 30)\n"
        "assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), oracle.jaccard(mat, nthreads=4)[0])\n"
        "assert (0, 0) in api._multi_ctx\n"
        "print('ok')\n")
    env = dict(os.environ, GFICF_HIP_DEVICES="0,0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("G,N,mn,mx,seed", [(600, 400, 0.05, 1.0, 7), (5000, 3000, 0.05, 1.0, 7), (300, 500, 0.0, 1.0, 8),
                                            (20000, 2000, 0.02, 0.6, 3), (50, 2, 0.0, 1.0, 6)])
def test_gficf_host_multi_equals_single_device_and_oracle(devices, G, N, mn, mx, seed):
    cp, ri, x = synth.counts_csc(G, N, seed=seed)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    one = gficf_amd.gficf(M, mx, mn, normalize=False, verbose=False)
    res = gficf_amd.gficf(M, mx, mn, normalize=False, verbose=False, devices=devices)
    g, h = res["gficf"], one["gficf"]
    assert np.array_equal(res["genes"], one["genes"]) and np.array_equal(res["nt"], one["nt"]) and np.array_equal(res["w"], one["w"])
    assert np.array_equal(g.indptr, h.indptr) and np.array_equal(g.indices, h.indices)
    assert np.array_equal(g.data, h.data)                     # a cell is never split: the same sums in the same order
    ref = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
    assert np.array_equal(g.indptr, ref["colptr"]) and np.array_equal(g.indices, ref["rowidx"])
    assert np.allclose(g.data, ref["x"], rtol=1e-6, atol=1e-6) and np.abs(g.data - ref["x"]).max(initial=0.0) < 1e-12
    # int32 colptr (what a dgCMatrix holds) and supplied weights (embedNewCells, R/cellClassifier.R:50-53)
    M32 = sp.csc_matrix((x, ri, cp.astype(np.int32)), shape=(G, N))
    r32 = gficf_amd.gficf(M32, mx, mn, normalize=False, verbose=False, devices=devices)
    assert np.array_equal(r32["gficf"].data, g.data) and np.array_equal(r32["gficf"].indptr, g.indptr)


def test_gficf_host_multi_bad_csc_and_empty():
    bad = sp.csc_matrix((np.ones(3), np.array([0, 7, 1]), np.array([0, 2, 3])), shape=(8, 2))
    bad.indices[1] = 99
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.gficf(bad, normalize=False, verbose=False, devices=[0, 0])
    assert ei.value.status == "GFICF_ERR_BAD_CSC"
    r = gficf_amd.gficf(sp.csc_matrix((5, 0)), normalize=False, verbose=False, devices=[0, 0])
    assert r["gficf"].shape == (0, 0)


@pytest.mark.parametrize("N,k", [(5000, 30), (3000, 50), (100000, 30), (200000, 15), (300, 256)])
def test_compact_return_counts_and_host_expansion(N, k):
    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N) if N > 600 else synth.knn_uniform(N, k)
    u = gficf_amd.jaccard_counts(mat)
    assert u.dtype == np.uint16 and u.shape == (N, k)
    full = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    assert np.array_equal(gficf_amd.jaccard_expand(mat, u), full)
    if N <= 100000:
        assert np.array_equal(u.reshape(-1).astype(np.int32), oracle.jaccard(mat, nthreads=8)[1])


def test_host_entries_do_not_allocate_per_call_and_trim_releases():
    """Device memory of the host entries comes from the context's pool: free memory is the same after the second call as after the first."""
    import ctypes

    import torch

    from gficf_amd import _lib

    ctx = gficf_amd.Context(0)
    mat = synth.knn_windowed(20000, 30)
    cp, ri, x = synth.counts_csc(3000, 2000)
    M = sp.csc_matrix((x, ri, cp), shape=(3000, 2000))

    def calls():
        gficf_amd.rcpp_parallel_jaccard_coef(mat, False, ctx=ctx)
        gficf_amd.jaccard_edges(np.concatenate([np.arange(1, 20001, dtype=np.int32)[:, None], mat], axis=1), ctx=ctx)
        gficf_amd.gficf(M, normalize=False, verbose=False, ctx=ctx)
        gficf_amd.transpose_gficf(M, ctx=ctx)

    calls()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        calls()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free1
    _lib.check(_lib.load().gficf_ctx_trim(ctx.handle))
    assert torch.cuda.mem_get_info()[0] > free1
    calls()                                                    # usable after a trim
    ctx.close()


@pytest.mark.parametrize("devices,N,k,f64", [([0, 0], 5001, 30, False), ([0, 0, 0], 20000, 15, False), ([0, 0], 3000, 50, True),
                                             ([0, 0, 0], 140000, 30, False), ([0], 4000, 30, False)])
def test_device_resident_multi_step_with_peer_copies(devices, N, k, f64):
    """gficf_multi_jaccard_device: every block already in HBM, every device (here: contexts on the one GPU of the box) ingests its
    block, pulls the other table slices with hipMemcpyPeerAsync on copy streams of its own and builds its block's edges — bit for
    bit the single-device result; a second step over the same buffers (ordered behind the first by events) gives it again."""
    import torch

    import oracle
    from gficf_amd.api import MultiContext

    mat = synth.knn_windowed(N, k, seed=4, perm_seed=5)
    mat2 = synth.knn_windowed(N, k, seed=6, perm_seed=7)
    want, _ = oracle.jaccard(mat, nthreads=8) if N <= 20000 else (gficf_amd.rcpp_parallel_jaccard_coef(mat, False), None)
    mc = MultiContext(devices)
    P = len(devices)
    bd = mc.cell_blocks(N)
    rw = gficf_amd.HipOps.row_words(N, k)
    dt = torch.float64 if f64 else torch.int32
    blocks = lambda m: [torch.from_numpy(np.ascontiguousarray(m[bd[r]:bd[r + 1]].T)).to("cuda:0").to(dt).contiguous() for r in range(P)]
    i1, i2 = blocks(mat), blocks(mat2)
    tables = [torch.zeros((N, rw), dtype=torch.int32, device="cuda:0") for _ in range(P)]
    outs = [torch.zeros((3, (bd[r + 1] - bd[r]) * k), dtype=torch.float64, device="cuda:0") for r in range(P)]
    torch.cuda.synchronize()
    for distinct in (False, True):
        mc.set_jaccard_distinct(distinct)
        mc.jaccard_device(i2, N, k, tables, outs)          # a step on other data first: the next one must wait for its pulls
        mc.jaccard_device(i1, N, k, tables, outs)
        mc.sync()
        got = torch.cat(outs, dim=1).cpu().numpy().T
        assert np.array_equal(got, want), (devices, N, k, distinct)
        for t in tables[1:]:
            assert torch.equal(t, tables[0])                # every device ends with the whole table
    # deferred errors come out of the sync: a bad id in the last block
    bad = blocks(mat)
    bad[-1][0, 0] = N + 1
    mc.jaccard_device(bad, N, k, tables, outs)
    with pytest.raises(gficf_amd.GficfError) as ei:
        mc.sync()
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    if not f64:
        dup = blocks(mat)
        dup[0][1, 3] = dup[0][0, 3]                          # a repeated id in a row: reported in distinct mode, exact otherwise
        mc.set_jaccard_distinct(True)
        mc.jaccard_device(dup, N, k, tables, outs)
        with pytest.raises(gficf_amd.GficfError) as ei:
            mc.sync()
        assert ei.value.status == "GFICF_ERR_DUPLICATE_IDS"
    mc.close()


@pytest.mark.parametrize("devices,N,k", [([0, 0], 5001, 30), ([0, 0, 0], 20000, 15), ([0, 0, 0, 0], 30001, 50), ([0, 0], 260000, 30), ([0], 4000, 30),
                                         ([0, 0, 0], 700, 64)])
def test_device_resident_halo_step_reads_the_named_rows_in_place(devices, N, k):
    """gficf_multi_jaccard_halo_device: blocks whose ids have locality, nothing exchanged — every device plans the rows its block names
    outside, reads them where they lie (the other devices' blocks of ids; here contexts on the one GPU of the box) and builds its
    sub-problem's table and edges: bit for bit the single-device result, step after step over the same buffers; ids without
    locality raise GFICF_ERR_CAPACITY at the sync (and the context stays usable)."""
    import torch

    import oracle
    from gficf_amd.api import MultiContext

    mat = synth.knn_windowed(N, k, W=max(100, k), seed=4, perm_seed=None)
    mat2 = synth.knn_windowed(N, k, W=max(100, k), seed=6, perm_seed=None)
    mat[N // 2, 0] = 1                                         # rows far outside the band too: a slot of owner 0 / of the last owner
    mat2[N // 3, k - 1] = N
    ref = lambda m: oracle.jaccard(m, nthreads=8)[0] if N <= 40000 else gficf_amd.rcpp_parallel_jaccard_coef(m, False)
    want, want2 = ref(mat), ref(mat2)
    mc = MultiContext(devices)
    P = len(devices)
    bd = mc.cell_blocks(N)
    blocks = lambda m: [torch.from_numpy(np.ascontiguousarray(m[bd[r]:bd[r + 1]].T)).to("cuda:0").contiguous() for r in range(P)]
    i1, i2 = blocks(mat), blocks(mat2)
    bufs = mc.halo_buffers(N, k)
    torch.cuda.synchronize()
    for distinct in (False, True):
        mc.set_jaccard_distinct(distinct)
        for blk, w in ((i2, want2), (i1, want), (i1, want)):
            mc.jaccard_halo_device(blk, N, k, bufs)
            mc.sync()
            got = torch.cat(bufs["out"], dim=1).cpu().numpy().T
            assert np.array_equal(got, w), (devices, N, k, distinct)
    mc.set_jaccard_distinct(False)
    if P > 1:
        # a block that names more rows of one owner than there are slots: the deferred CAPACITY error, then a good step again
        small = mc.halo_buffers(N, k, cap=2)
        mc.jaccard_halo_device(i1, N, k, small)
        with pytest.raises(gficf_amd.GficfError) as ei:
            mc.sync()
        assert ei.value.status == "GFICF_ERR_CAPACITY"
        mc.jaccard_halo_device(i1, N, k, bufs)
        mc.sync()
        assert np.array_equal(torch.cat(bufs["out"], dim=1).cpu().numpy().T, want)
    bad = blocks(mat)
    bad[-1][0, 0] = N + 1
    mc.jaccard_halo_device(bad, N, k, bufs)
    with pytest.raises(gficf_amd.GficfError) as ei:
        mc.sync()
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    with pytest.raises(gficf_amd.GficfError) as ei:           # k > 64: this form is the fused one
        mc.jaccard_halo_device([b[:1].repeat(65, 1).contiguous() for b in i1], N, 65, bufs)
    assert ei.value.status == "GFICF_ERR_UNSUPPORTED"
    mc.close()
