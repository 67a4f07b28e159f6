"""The R `.Call` glue (integration/gficf_hip_glue.c) EXECUTED: compiled against a runnable stand-in of the R C API
(tests/r_mock/r_runtime.c — NOT R; a tagged-heap SEXP model with a PROTECT stack, a torture collector that frees every
unprotected object at every allocation, Rf_error by longjmp, Rprintf capture and R_registerRoutines) and called through the
routine table it registers, by name and arity, as `.Call` does (reference: src/RcppExports.cpp:61-70 the entry, :85-97 the
table).  Results are compared with the oracle; the protect stack must balance and no collected object may be touched.
Without a GPU only the registration and the error path run (no compute calls)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import gficf_amd
from gficf_amd import synth
from tests.helpers import rmock


@pytest.fixture(scope="module")
def R():
    import shutil

    if not os.path.exists(rmock.SO) and shutil.which("gcc") is None:
        pytest.skip("no gcc and no pre-built glue stand-in: nothing to execute")
    rt = rmock.RMock()
    yield rt
    rt.unload()


def test_glue_registers_its_routines_by_name_and_arity(R):
    tab = R.routines()
    # the two entries that REPLACE reference entries keep name and arity (src/RcppExports.cpp:87,89)
    assert tab["_gficf_rcpp_parallel_jaccard_coef"] == 2 and tab["_gficf_jaccard_coeff"] == 2
    assert tab["_gficf_gficf_csc"] == 7 and tab["_gficf_gficf_csc_raw"] == 7
    with pytest.raises(KeyError):
        R.call("_gficf_no_such_entry")
    with pytest.raises(TypeError):                      # `.Call` with the wrong number of arguments is refused by the table
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(np.ones((2, 1), dtype=np.int32)))


def test_one_registration_table_survives_replace_semantics(R):
    """R_registerRoutines REPLACES a DllInfo's `.Call` table (R src/main/Rdynload.c), it does not append: the stand-in has the same
    semantics (self-test below).  The documented init path — the glue's own R_init_gficf, run by rmock_init() as dyn.load would — must
    therefore register ONE table holding the ten rows of the glue AND the three Rcpp routines that stay in the package
    (reference src/RcppExports.cpp:85-92: `_gficf_RunModularityClusteringCpp` 9, `_gficf_rcpp_WMU_test` 3, `_gficf_rcpp_parallel_WMU_test` 3),
    then switch dynamic lookup off (:96) — so that clustcells(community.algo = "louvian 2") and findClusterMarkers() still resolve."""
    assert R.L.rmock_selftest_second_registration_replaces() == 1        # two registrations: only the second table is left
    assert R.L.rmock_n_registrations() == 1                              # the init path registers exactly once
    assert R.L.rmock_use_dynamic_symbols() == 0                          # R_useDynamicSymbols(dll, FALSE)
    want = {"_gficf_RunModularityClusteringCpp": 9, "_gficf_rcpp_WMU_test": 3, "_gficf_rcpp_parallel_WMU_test": 3,     # the package's own
            "_gficf_rcpp_parallel_jaccard_coef": 2, "_gficf_jaccard_coeff": 2, "_gficf_gficf_csc": 7, "_gficf_gficf_csc_raw": 7,
            "_gficf_find_nn": 3, "_gficf_jaccard_adjacency": 4, "_gficf_cluster_signatures": 6, "_gficf_transpose_csc": 4,
            "_gficf_RunModularityClusteringHip": 9, "_gficf_phenograph": 8}
    assert R.routines() == want and len(want) == 13
    # the right function sits behind each of the package's names (the stand-ins return a number of their own)
    nil = R.null()
    for name, mark in (("_gficf_RunModularityClusteringCpp", 9009), ("_gficf_rcpp_WMU_test", 3003), ("_gficf_rcpp_parallel_WMU_test", 3103)):
        res = R.call(name, *([nil] * want[name]))
        assert res.type == rmock.INTSXP and int(res.numpy()[0]) == mark
        with pytest.raises(TypeError):
            R.call(name, nil)
    # and a package that is loaded again (R CMD INSTALL's test load, devtools::load_all) registers the same table again
    R.L.rmock_init()
    assert R.routines() == want and R.L.rmock_n_registrations() == 1


def test_gficf_entry_rejects_wrong_typed_slots_before_touching_them(R):
    """`_gficf_gficf_csc` reads @i / @p / @x / @Dim of a dgCMatrix: a pattern or logical matrix (ngCMatrix has no @x, lgCMatrix a logical
    one), a double @p, a short @i must reach Rf_error — not a read through the wrong type.  No device is needed to get there."""
    i = np.array([0, 1, 0], dtype=np.int32)
    p = np.array([0, 2, 3], dtype=np.int32)
    x = np.array([1.0, 2.0, 3.0])
    dim = np.array([2, 2], dtype=np.int32)
    one = lambda v: R.vector(np.array([v]))                                     # noqa: E731
    good = dict(i=R.vector(i), p=R.vector(p), x=R.vector(x), dim=R.vector(dim), w=R.null(), mn=one(0.0), mx=one(1.0))

    def call(entry="_gficf_gficf_csc", **kw):
        a = dict(good, **kw)
        return R.call(entry, a["i"], a["p"], a["x"], a["dim"], a["w"], a["mn"], a["mx"])

    for entry in ("_gficf_gficf_csc", "_gficf_gficf_csc_raw"):
        with pytest.raises(rmock.RError, match="x must be a double"):
            call(entry, x=R.vector(np.array([True, True, True])))               # lgCMatrix@x
        with pytest.raises(rmock.RError, match="x must be a double"):
            call(entry, x=R.null())                                             # ngCMatrix: no @x
        with pytest.raises(rmock.RError, match="x must be a double"):
            call(entry, x=R.vector(np.array([1, 2, 3], dtype=np.int32)))        # igCMatrix@x
        with pytest.raises(rmock.RError, match="p must be an integer"):
            call(entry, p=R.vector(p.astype(np.float64)))
        with pytest.raises(rmock.RError, match="p must be an integer"):
            call(entry, p=R.vector(p[:2]))                                      # not ncol + 1 long
        with pytest.raises(rmock.RError, match="i must be an integer"):
            call(entry, i=R.vector(i.astype(np.float64)))
        with pytest.raises(rmock.RError, match="must hold"):
            call(entry, i=R.vector(i[:2]))                                      # shorter than p[ncol + 1]
        with pytest.raises(rmock.RError, match="must hold"):
            call(entry, x=R.vector(x[:1]))
        with pytest.raises(rmock.RError, match="dim must be"):
            call(entry, dim=R.vector(dim.astype(np.float64)))
        with pytest.raises(rmock.RError, match="dim must be"):
            call(entry, dim=R.vector(np.array([2], dtype=np.int32)))
        with pytest.raises(rmock.RError, match="w must be"):
            call(entry, w=R.vector(np.array([1.0])))                            # not nrow long
        with pytest.raises(rmock.RError, match="w must be"):
            call(entry, w=R.vector(np.array([1, 1], dtype=np.int32)))
        with pytest.raises(rmock.RError, match="single numbers"):
            call(entry, mn=R.null())
        assert R.L.rmock_protect_depth() == 0 and R.L.rmock_dead_touched() == 0


def test_torture_collector_catches_a_missing_protect(R):
    R.L.rmock_selftest_missing_protect.restype = int
    assert R.L.rmock_selftest_missing_protect() == 1
    R.L.rmock_reset()                                    # (the self-test's dead object and its count do not stay for the tests that follow in this process)
    assert R.L.rmock_dead_touched() == 0


def test_argument_checks_reach_rf_error(R):
    with pytest.raises(rmock.RError, match="integer or numeric matrix"):
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.vector(np.arange(6, dtype=np.int32)), R.vector(np.array([False])))
    with pytest.raises(rmock.RError, match="integer or numeric matrix"):
        R.call("_gficf_jaccard_coeff", R.vector(np.arange(6.0)), R.vector(np.array([False])))
    assert R.L.rmock_protect_depth() == 0


@pytest.mark.skipif(gficf_amd.device_count() > 0, reason="GPU present")
def test_without_a_gpu_the_call_is_an_r_error_not_a_fallback(R):
    with pytest.raises(rmock.RError, match="no HIP device"):
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(np.array([[2], [1]], dtype=np.int32)), R.vector(np.array([False])))
    assert R.L.rmock_protect_depth() == 0               # (R's context unwinding resets the stack after an error)


# ------------------------------------------------------------------------------------------------------------- GPU
def _check_call_hygiene(R):
    assert R.last_protect_exit == 0 and R.L.rmock_protect_depth() == 0
    assert R.L.rmock_dead_touched() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("as_double", [False, True])
def test_dot_call_jaccard_entry_matches_the_oracle(R, as_double):
    import oracle

    mat = synth.knn_windowed(3000, 15, seed=5, perm_seed=6)
    want, _ = oracle.jaccard(mat, nthreads=4)
    arg = R.matrix(mat.astype(np.float64) if as_double else mat)        # REALSXP (what Rcpp coerces to) / INTSXP (what uwot returns)
    res = R.call("_gficf_rcpp_parallel_jaccard_coef", arg, R.vector(np.array([True])))
    _check_call_hygiene(R)
    assert res.type == rmock.REALSXP and res.dim == (3000 * 15, 3)      # reference :67
    assert np.array_equal(res.numpy(), want)
    # printOutput = TRUE: the reference's two banner lines (src/rcpp_parallel_jaccard_coeff.cpp:63,77) went through Rprintf
    assert "Running Parallell Jaccard Coefficient Estimation" in R.printed() and "Done" in R.printed()
    R.call("_gficf_rcpp_parallel_jaccard_coef", arg, R.vector(np.array([False])))
    assert R.printed() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("N", [600, 5000])
@pytest.mark.parametrize("k", [257, 300, 513])
def test_dot_call_jaccard_beyond_256_neighbours(R, N, k):
    """The reference's loop has no limit on mat.ncol() (src/rcpp_parallel_jaccard_coeff.cpp:26-36): neither has the `.Call`
    (k > 256 takes the sorted-row path, csrc/jaccard_sorted.h)."""
    import oracle

    if N > 2 * k + 2:
        mat = synth.knn_windowed(N, k, W=max(100, k), seed=k, perm_seed=N)
    else:
        mat = np.stack([np.random.default_rng(N + k + i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32)
    want, _ = oracle.jaccard(mat, nthreads=8)
    res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    _check_call_hygiene(R)
    assert res.type == rmock.REALSXP and res.dim == (N * k, 3) and np.array_equal(res.numpy(), want)


@pytest.mark.gpu
def test_dot_call_jaccard_edge_shapes(R):
    import oracle

    for N, k in ((1, 1), (5, 3), (64, 1), (200, 50), (300, 100)):
        rng = np.random.default_rng(N * 1000 + k)
        mat = rng.integers(1, N + 1, size=(N, k)).astype(np.int32)     # repeated ids, self ids: the multiset path
        want, _ = oracle.jaccard(mat, nthreads=2)
        res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
        _check_call_hygiene(R)
        assert res.dim == (N * k, 3) and np.array_equal(res.numpy(), want), (N, k)
    res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(np.zeros((0, 5), dtype=np.int32)), R.vector(np.array([False])))
    assert res.dim == (0, 3)


@pytest.mark.gpu
def test_dot_call_bad_id_is_an_r_error(R):
    mat = synth.knn_windowed(500, 10, seed=1)
    mat[17, 3] = 501                                                   # the reference reads outside the matrix here (:34)
    with pytest.raises(rmock.RError, match="gficf_hip: .*(id|ids)"):
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    assert R.L.rmock_protect_depth() == 0
    m2 = mat.astype(np.float64)
    m2[17, 3] = 4.5                                                    # non-integer double: rejected unless GFICF_HIP_TRUNCATE_IDS=1
    with pytest.raises(rmock.RError, match="gficf_hip"):
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(m2), R.vector(np.array([False])))
    # and the session goes on: the next call is served
    mat[17, 3] = 20
    res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    assert res.dim == (5000, 3)


@pytest.mark.gpu
def test_dot_call_serial_entry_matches_the_oracle(R):
    import oracle

    rng = np.random.default_rng(3)
    mat = rng.integers(1, 401, size=(400, 12)).astype(np.int32)
    res = R.call("_gficf_jaccard_coeff", R.matrix(mat), R.vector(np.array([False])))
    _check_call_hygiene(R)
    assert np.array_equal(res.numpy(), oracle.jaccard_coeff(mat))


def _counts(G, N, seed=7):
    colptr, rowidx, x = synth.counts_csc(G, N, seed=seed)
    return sp.csc_matrix((x, rowidx, colptr), shape=(G, N))


def _call_gficf(R, M, w=None, pmin=0.05, pmax=1.0, entry="_gficf_gficf_csc"):
    G, N = M.shape
    return R.call(entry, R.vector(M.indices.astype(np.int32)), R.vector(M.indptr.astype(np.int32)), R.vector(M.data),
                  R.vector(np.array([G, N], dtype=np.int32)), R.null() if w is None else R.vector(np.asarray(w, dtype=np.float64)),
                  R.vector(np.array([pmin])), R.vector(np.array([pmax])))


def _check_gficf(res, ref, G, N, n_elts=6):
    assert res.type == rmock.VECSXP and len(res) == n_elts
    oi, op, ox, keep, nt, w = (res.elt(i).numpy() for i in range(6))
    kb = ref["keep"].astype(bool)
    assert np.array_equal(keep.astype(bool), kb) and np.array_equal(nt[kb], ref["nt"][kb].astype(np.float64))   # (the oracle reports 0 for dropped genes)
    assert np.array_equal(op, ref["colptr"]) and np.array_equal(oi, ref["rowidx"])
    assert np.allclose(ox, ref["x"], rtol=1e-6, atol=1e-6) and np.allclose(w, ref["w"], rtol=1e-6, atol=1e-6)   # north_star's tolerance
    assert np.abs(ox - ref["x"]).max(initial=0.0) < 1e-12                                                        # (observed)


@pytest.mark.gpu
def test_dot_call_gficf_entry_matches_the_oracle(R):
    import oracle

    G, N = 1200, 700
    M = _counts(G, N)
    ref = oracle.gficf_csc(G, N, M.indptr.astype(np.int64), M.indices, M.data, 0.05, 1.0)
    res = _call_gficf(R, M)
    _check_call_hygiene(R)
    _check_gficf(res, ref, G, N)
    assert np.array_equal(res.elt(4).numpy(), np.asarray((M != 0).sum(axis=1)).ravel().astype(np.float64))      # nt of EVERY gene: rowSums(M != 0), R/gficf.R:40
    # embedNewCells(): the ICF weights supplied (R/cellClassifier.R:50-53), filter open (max = 2, min = 0)
    w_in = np.abs(np.random.default_rng(2).normal(size=G)) + 0.1
    ref2 = oracle.gficf_csc(G, N, M.indptr.astype(np.int64), M.indices, M.data, 0.0, 2.0, w_in=w_in)
    res2 = _call_gficf(R, M, w=w_in, pmin=0.0, pmax=2.0)
    _check_call_hygiene(R)
    _check_gficf(res2, ref2, G, N)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["_gficf_gficf_csc", "_gficf_gficf_csc_raw"])
def test_an_r_error_between_plan_and_finish_leaves_the_next_call_correct(R, entry):
    """The GF-ICF entry is two library calls with R allocations in between (`_plan` sizes the result, R allocates it, `_finish` fills it): an
    R error there — memory running out in Rf_allocVector is the one R itself raises — unwinds past the glue with a plan still pending in the
    context.  Every allocation of the call is made to fail in turn; after each failure the same entry is called again, on ANOTHER matrix, and
    must be right (the next `_plan` drops the stale one), with the protect stack balanced and nothing dead touched."""
    import oracle

    G, N = 900, 500
    M = _counts(G, N, seed=21)
    M2 = _counts(700, 650, seed=22)
    ref2 = oracle.gficf_csc(700, 650, M2.indptr.astype(np.int64), M2.indices, M2.data, 0.05, 1.0)
    n_elts = 7 if entry.endswith("_raw") else 6
    failed = 0
    for nth in range(1, 12):
        R.L.rmock_fail_allocation_after(nth)
        try:
            _call_gficf(R, M, entry=entry)
            R.L.rmock_fail_allocation_after(0)
            break                                                          # the call makes fewer than `nth` allocations: every one has been failed
        except rmock.RError as e:
            assert "cannot allocate" in str(e)
            failed += 1
        assert R.L.rmock_protect_depth() == 0
        res = _call_gficf(R, M2, entry=entry)
        _check_call_hygiene(R)
        _check_gficf(res, ref2, 700, 650, n_elts=n_elts)
    assert failed >= 6                                                     # out, i, p, x, keep, w (+ raw x, nt) — all between plan and finish
    # and the Jaccard entry next to it is unaffected by a stale GF-ICF plan
    mat = synth.knn_windowed(2000, 15, seed=5, perm_seed=6)
    R.L.rmock_fail_allocation_after(1)
    with pytest.raises(rmock.RError, match="cannot allocate"):
        R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    want, _ = oracle.jaccard(mat, nthreads=4)
    assert np.array_equal(R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False]))).numpy(), want)


@pytest.mark.gpu
def test_dot_call_gficf_raw_entry_returns_the_counts_of_the_kept_rows(R):
    """`_gficf_gficf_csc_raw`: the six elements of `_gficf_gficf_csc` + the x of the kept entries — (r[[1]], r[[2]], r[[7]]) is normCounts'
    `M[keep, ]` (reference R/gficf.R:40), what gficf(storeRaw = TRUE) keeps as $rawCounts (:22).  Explicitly stored zeros stay stored, as R's
    subsetting keeps them.  Also with the device list from the environment (the gather then runs behind the multi-GPU finish call)."""
    import oracle

    G, N = 1200, 700
    M = _counts(G, N)
    M.data[::97] = 0.0                                                   # explicitly stored zeros
    ref = oracle.gficf_csc(G, N, M.indptr.astype(np.int64), M.indices, M.data, 0.05, 1.0)

    def check():
        res = _call_gficf(R, M, entry="_gficf_gficf_csc_raw")
        _check_call_hygiene(R)
        _check_gficf(res, ref, G, N, n_elts=7)
        keep = res.elt(3).numpy().astype(bool)
        raw = sp.csc_matrix((res.elt(6).numpy(), res.elt(0).numpy(), res.elt(1).numpy()), shape=(int(keep.sum()), N))
        want = M[np.flatnonzero(keep), :]
        assert np.array_equal(raw.indptr, want.indptr) and np.array_equal(raw.indices, want.indices) and np.array_equal(raw.data, want.data)

    check()
    R.unload()
    os.environ["GFICF_HIP_DEVICES"] = "0,0"
    try:
        check()
    finally:
        del os.environ["GFICF_HIP_DEVICES"]
        R.unload()


@pytest.mark.gpu
def test_dot_call_device_list_from_the_environment_takes_the_multi_path(R):
    """GFICF_HIP_DEVICES=0,0: two contexts on the one GPU of the box — the same results, through gficf_*_host_multi."""
    import oracle

    mat = synth.knn_windowed(4001, 30, seed=9, perm_seed=10)
    want, _ = oracle.jaccard(mat, nthreads=4)
    M = _counts(900, 501, seed=3)
    ref = oracle.gficf_csc(900, 501, M.indptr.astype(np.int64), M.indices, M.data, 0.05, 1.0)
    R.unload()                                                          # R_unload_gficf: the device list is read again
    os.environ["GFICF_HIP_DEVICES"] = "0,0"
    try:
        res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([True])))
        _check_call_hygiene(R)
        assert np.array_equal(res.numpy(), want) and "Done" in R.printed()
        _check_gficf(_call_gficf(R, M), ref, 900, 501)
        _check_call_hygiene(R)
        R.unload()
        os.environ["GFICF_HIP_DEVICES"] = "0,99"                        # an ordinal that does not exist: an error at EVERY call, never a silent single-device run
        for _ in range(2):
            with pytest.raises(rmock.RError, match="GFICF_HIP_DEVICES"):
                R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    finally:
        del os.environ["GFICF_HIP_DEVICES"]
        R.unload()
    res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(mat), R.vector(np.array([False])))
    assert np.array_equal(res.numpy(), want)


@pytest.mark.gpu
def test_dot_call_optional_entries(R):
    """The other registered routines, each against the Python mirror of the same C ABI call (whose parity tests live in
    tests/test_knn_gpu.py, test_adjacency_gpu.py, test_gficf_gpu.py, test_louvain_gpu.py) — what is checked HERE is the glue:
    argument unpacking, allocation, list assembly, names, attributes."""
    rng = np.random.default_rng(0)
    centers = rng.normal(scale=5.0, size=(6, 8))
    X = centers[rng.integers(0, 6, 1500)] + rng.normal(size=(1500, 8))
    nn = R.call("_gficf_find_nn", R.matrix(X), R.vector(np.array([11], dtype=np.int32)), R.vector(np.array([0], dtype=np.int32)))
    _check_call_hygiene(R)
    assert nn.attr("names").strings() == ["idx", "dist"]
    ref = gficf_amd.find_nn(X, 11, metric="manhattan")
    assert np.array_equal(nn.elt(0).numpy(), ref["idx"]) and np.allclose(nn.elt(1).numpy(), ref["dist"])
    # edges -> adjacency
    ed = gficf_amd.jaccard_edges(ref["idx"], False)
    adj = R.call("_gficf_jaccard_adjacency", R.vector(ed["from"]), R.vector(ed["to"]), R.vector(ed["weight"]), R.vector(np.array([1500.0])))
    _check_call_hygiene(R)
    A = gficf_amd.jaccard_adjacency(ed, 1500)
    assert np.array_equal(adj.elt(0).numpy(), A.indices) and np.array_equal(adj.elt(1).numpy(), A.indptr) and np.array_equal(adj.elt(2).numpy(), A.data)
    # Louvain on it: an S4 object with the slots of a dgCMatrix, the reference's argument list (src/RcppExports.cpp:17)
    lab = R.call("_gficf_RunModularityClusteringHip", R.dgcmatrix(A), R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([0.8])),
                 R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([10], dtype=np.int32)),
                 R.vector(np.array([0], dtype=np.int32)), R.vector(np.array([True])), R.string(""))
    _check_call_hygiene(R)
    want_lab = gficf_amd.run_modularity_clustering(A, 1, 0.8, 1, 1, 10, 0, False)
    assert np.array_equal(lab.numpy(), np.asarray(want_lab)) and "Number of communities" in R.printed()
    with pytest.raises(rmock.RError, match="Modularity parameter"):
        R.call("_gficf_RunModularityClusteringHip", R.dgcmatrix(A), R.vector(np.array([3], dtype=np.int32)), R.vector(np.array([0.8])),
               R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([10], dtype=np.int32)),
               R.vector(np.array([0], dtype=np.int32)), R.vector(np.array([False])), R.string(""))
    # signatures + transpose of a normalised matrix
    M = _counts(600, 400, seed=11)
    g = gficf_amd.gficf(M, normalize=False, verbose=False)["gficf"]
    dim = R.vector(np.array(g.shape, dtype=np.int32))
    cl = (np.arange(400) * 7 % 5).astype(np.int32)
    sig = R.call("_gficf_cluster_signatures", R.vector(g.indices), R.vector(g.indptr.astype(np.int32)), R.vector(g.data), dim, R.vector(cl),
                 R.vector(np.array([5], dtype=np.int32)))
    _check_call_hygiene(R)
    want_sig = np.stack([np.asarray(g[:, cl == c].sum(axis=1)).ravel() for c in range(5)], axis=1)
    assert sig.dim == (g.shape[0], 5) and np.allclose(sig.numpy(), want_sig, rtol=1e-9, atol=1e-12)
    tr = R.call("_gficf_transpose_csc", R.vector(g.indices), R.vector(g.indptr.astype(np.int32)), R.vector(g.data), dim)
    _check_call_hygiene(R)
    gt = sp.csc_matrix(g.T)
    gt.sort_indices()
    assert np.array_equal(tr.elt(0).numpy(), gt.indices) and np.array_equal(tr.elt(1).numpy(), gt.indptr) and np.array_equal(tr.elt(2).numpy(), gt.data)
    # clustcells() lines 57-86 in one call: labels + two attributes
    ph = R.call("_gficf_phenograph", R.matrix(X), R.vector(np.array([10], dtype=np.int32)), R.vector(np.array([0], dtype=np.int32)), R.vector(np.array([0.8])),
                R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([1], dtype=np.int32)), R.vector(np.array([10], dtype=np.int32)),
                R.vector(np.array([0], dtype=np.int32)))
    _check_call_hygiene(R)
    want_ph = gficf_amd.phenograph(X, 10, "manhattan", 0.8, 1, 1, 10, 0)
    assert np.array_equal(ph.numpy(), np.asarray(want_ph))
    assert abs(float(ph.attr("modularity").numpy()[0]) - want_ph.modularity) < 1e-12 and float(ph.attr("n.edges").numpy()[0]) == want_ph.n_edges


@pytest.mark.gpu
def test_dot_call_entries_against_answers_derived_by_counting(R):
    """The two `.Call` entries of the hot path against closed forms, no oracle in the loop (tests/helpers/closed_form.py): the Jaccard
    entry on a ring of cells (u(i, t) = k - 1 - t: one small case that takes the one-launch form, one of 1.1 M edges that returns through
    uint16 counts + host-side expansion into the R matrix), and `_gficf_gficf_csc` on a circulant count matrix
    (gficf = (t + 1) / sqrt(s (s + 1) (2 s + 1) / 6), rare genes dropped by the 5 % filter)."""
    from tests.helpers.closed_form import circulant_counts, cyclic_window_expected, cyclic_window_matrix

    for N, k in ((3000, 15), (36_000, 30)):
        want, _ = cyclic_window_expected(N, k)
        res = R.call("_gficf_rcpp_parallel_jaccard_coef", R.matrix(cyclic_window_matrix(N, k)), R.vector(np.array([False])))
        _check_call_hygiene(R)
        assert res.dim == (N * k, 3) and np.array_equal(res.numpy(), want)
    G, N, s, n_rare = 400, 2000, 31, 300
    M, want, keep, nt, w = circulant_counts(G, N, s, n_rare)
    res = _call_gficf(R, M)
    _check_call_hygiene(R)
    oi, op, ox, kp, ntr, wr = (res.elt(i).numpy() for i in range(6))
    assert np.array_equal(kp.astype(bool), keep) and np.array_equal(ntr, nt.astype(np.float64))
    assert np.array_equal(op, want.indptr) and np.array_equal(oi, want.indices)
    assert np.allclose(ox, want.data, rtol=1e-12, atol=0) and np.allclose(wr[:G], w, rtol=1e-13) and not wr[G:].any()
