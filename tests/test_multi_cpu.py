"""CPU checks of the host-side pieces of the multi-GPU C ABI and of the compact-return expansion (no GPU):
the block arithmetic of gficf_multi_* equals the multi-process path's (gficf_amd/dist.py), and
gficf_jaccard_expand_host rebuilds the reference's edge matrix from intersection counts (against the oracle)."""
import ctypes

import numpy as np
import pytest

import gficf_amd
import oracle
from gficf_amd import _lib, synth
from gficf_amd.dist import shard_bounds, shard_bounds_by_nnz


def test_cell_blocks_match_the_multi_process_partition():
    L = _lib.load()
    for N, P in ((0, 1), (1, 3), (10, 3), (100000, 8), (7, 8), (1_000_000, 8), (54000, 5)):
        b = (ctypes.c_int64 * (P + 1))()
        assert L.gficf_multi_cell_blocks(N, P, b) == 0
        assert [(b[r], b[r + 1]) for r in range(P)] == [shard_bounds(N, P, r) for r in range(P)]
    assert L.gficf_multi_cell_blocks(-1, 2, (ctypes.c_int64 * 3)()) != 0
    assert L.gficf_multi_cell_blocks(5, 0, (ctypes.c_int64 * 3)()) != 0


def test_cell_blocks_by_nnz_match_the_multi_process_partition():
    L = _lib.load()
    rng = np.random.default_rng(0)
    for N, P in ((0, 2), (1, 2), (50, 4), (1000, 8), (13, 8), (300, 1)):
        for hi in (1, 40):           # hi = 1: every cell empty
            cp = np.concatenate([[0], np.cumsum(rng.integers(0, hi, size=N))]).astype(np.int64)
            for is64, arr in ((1, cp), (0, cp.astype(np.int32))):
                b = (ctypes.c_int64 * (P + 1))()
                assert L.gficf_multi_cell_blocks_by_nnz(N, arr.ctypes.data_as(ctypes.c_void_p), is64, P, b) == 0
                assert [(b[r], b[r + 1]) for r in range(P)] == shard_bounds_by_nnz(cp, P)


@pytest.mark.parametrize("N,k,dtype", [(3000, 30, np.int32), (500, 15, np.float64), (64, 5, np.int64), (200, 100, np.int32)])
def test_expand_host_rebuilds_the_reference_matrix(N, k, dtype):
    mat = synth.knn_windowed(N, k, W=max(100, k) if N > 2 * k + 2 else None or 25, seed=N) if N > 210 else synth.knn_uniform(N, k)
    want, u = oracle.jaccard(mat, nthreads=2)
    u16 = u.reshape(N, k).astype(np.uint16)
    for nt in (0, 1, 3):
        assert np.array_equal(gficf_amd.jaccard_expand(mat.astype(dtype), u16, n_threads=nt), want)
    with pytest.raises(ValueError):
        gficf_amd.jaccard_expand(mat, u16[:, :-1])


def test_env_device_list_parsing(monkeypatch):
    from gficf_amd import api

    monkeypatch.delenv("GFICF_HIP_DEVICES", raising=False)
    assert api.env_devices() is None
    monkeypatch.setenv("GFICF_HIP_DEVICES", "0")
    assert api.env_devices() is None
    monkeypatch.setenv("GFICF_HIP_DEVICES", "0, 1,2;3")
    assert api.env_devices() == [0, 1, 2, 3]


def test_ids_beyond_int32_are_rejected_before_the_narrowing_cast():
    from gficf_amd import api

    bad = np.array([[2 ** 32 + 1, 2], [1, 2]], dtype=np.int64)
    with pytest.raises(gficf_amd.GficfError) as ei:
        api._knn_matrix(bad)
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    m, f = api._knn_matrix(np.array([[2, 1], [1, 2]], dtype=np.uint16))
    assert m.dtype == np.int32 and f == 0 and m.flags.f_contiguous
