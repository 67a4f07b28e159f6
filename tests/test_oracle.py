"""CPU: the oracle against hand-derived known answers, its independent numpy restatement,
and the committed golden fixtures.  (The reference has no tests / golden vectors and cannot
run here: the oracle is 'parity unpinned' — see oracle/__init__.py.)"""
import hashlib
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import given, settings, strategies as st

import oracle
from gficf_amd import synth
from oracle import oracle_np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


def test_jaccard_known_answers(known):
    for case in known["jaccard"]:
        mat = np.array(case["mat"], dtype=np.int32)
        want_u = np.array(case["u"], dtype=np.int32).reshape(-1)
        N, k = mat.shape
        rm, u = oracle.jaccard(mat)
        assert np.array_equal(u, want_u), case["name"]
        assert np.array_equal(oracle_np.jaccard_counts_py(mat), want_u), case["name"]
        assert np.array_equal(oracle_np.jaccard_counts_np(mat), want_u), case["name"]
        # rows: (i+1, dst, u/(2k-u)) or zeros   (reference :48-52)
        for r in range(N * k):
            i, j = divmod(r, k)
            if want_u[r] == 0:
                assert tuple(rm[r]) == (0.0, 0.0, 0.0)
            else:
                assert tuple(rm[r]) == (i + 1.0, float(mat[i, j]), want_u[r] / (2.0 * k - want_u[r]))


def test_jaccard_toy_literals():
    mat = np.array([[2, 3, 4], [1, 3, 5], [1, 2, 4], [3, 5, 1], [4, 1, 2]], dtype=np.float64)
    rm, _ = oracle.jaccard(mat)
    assert rm.shape == (15, 3)
    assert rm[0].tolist() == [1.0, 2.0, 0.2]
    assert rm[1].tolist() == [1.0, 3.0, 0.5]
    assert rm[13].tolist() == [5.0, 1.0, 0.5]


def test_jaccard_int_and_double_inputs_agree():
    mat = synth.knn_windowed(500, 10, seed=3)
    a, ua = oracle.jaccard(mat)
    b, ub = oracle.jaccard(mat.astype(np.float64))
    assert np.array_equal(a, b) and np.array_equal(ua, ub)


def test_jaccard_rejects_out_of_range_ids():
    mat = np.array([[2, 3], [1, 3], [1, 4]], dtype=np.int32)   # 4 > N = 3
    with pytest.raises(ValueError):
        oracle.jaccard(mat)
    with pytest.raises(ValueError):
        oracle.jaccard(np.array([[0, 1], [1, 2]], dtype=np.int32))


def test_jaccard_threads_do_not_change_results():
    mat = synth.knn_windowed(3000, 15, seed=5)
    a, ua = oracle.jaccard(mat, nthreads=1)
    b, ub = oracle.jaccard(mat, nthreads=8)
    assert np.array_equal(a, b) and np.array_equal(ua, ub)


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 40), st.integers(1, 12), st.integers(0, 2**31 - 1))
def test_jaccard_cxx_vs_numpy_random_multisets(N, k, seed):
    r = synth.rand_u64(seed, np.arange(N * k)).reshape(N, k)
    mat = (r % np.uint64(N)).astype(np.int32) + 1       # duplicates and self ids allowed
    rm, u = oracle.jaccard(mat)
    assert np.array_equal(u, oracle_np.jaccard_counts_py(mat))
    assert np.array_equal(u, oracle_np.jaccard_counts_np(mat))
    assert np.array_equal(rm, oracle_np.jaccard_rmat(mat, u))


def test_jaccard_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "jaccard_cases.npz"))
    for nm in z["names"]:
        N, k, seed = (int(v) for v in z[nm + "_meta"])
        mat = synth.knn_windowed(N, k, seed=seed, perm_seed=seed + 100)
        rm, u = oracle.jaccard(mat, nthreads=4)
        assert np.array_equal(u.astype(np.uint8), z[nm + "_u"]), nm
        assert sha(np.asfortranarray(rm)) == bytes(z[nm + "_sha"]).decode(), nm
    for nm in ("uniform", "dupheavy"):
        rm, u = oracle.jaccard(z[nm + "_mat"], nthreads=2)
        assert np.array_equal(u.astype(np.uint8), z[nm + "_u"]), nm
        assert sha(np.asfortranarray(rm)) == bytes(z[nm + "_sha"]).decode(), nm


def test_synthetic_knn_statistics():
    # ~97 % of edges have u > 0 and the mean weight is ~0.06 at k = 30
    mat = synth.knn_windowed(20000, 30)
    assert all(len(set(r)) == 30 for r in mat[:200])
    assert (mat != np.arange(1, 20001)[:, None]).all()
    rm, u = oracle.jaccard(mat, nthreads=8)
    assert 0.95 < (u > 0).mean() < 0.99
    assert 0.05 < rm[u > 0, 2].mean() < 0.07


# --------------------------------------------------------------------------- GF-ICF
def _dense(res, N):
    return sp.csc_matrix((res["x"], res["rowidx"], res["colptr"]), shape=(res["G_kept"], N)).toarray()


def test_gficf_known_answers(known):
    g = known["gficf"]["basic"]
    M = sp.csc_matrix(np.array(g["M"], dtype=float))
    r = oracle.gficf_csc(4, 3, M.indptr, M.indices, M.data, g["min"], g["max"])
    assert r["nt"].tolist() == g["nt"]
    assert np.allclose(r["w"], g["w"], rtol=0, atol=1e-15)
    assert np.allclose(_dense(r, 3), np.array(g["dense"]), rtol=0, atol=1e-15)
    r2 = oracle_np.gficf_np(M, g["min"], g["max"])
    assert np.allclose(r2["gficf"].toarray(), np.array(g["dense"]), rtol=0, atol=1e-15)

    f = known["gficf"]["filter"]
    r = oracle.gficf_csc(4, 3, M.indptr, M.indices, M.data, f["min"], f["max"])
    assert r["keep"].astype(int).tolist() == f["keep"]
    assert np.allclose(_dense(r, 3), np.array(f["dense"]), rtol=0, atol=1e-15)

    z = known["gficf"]["w_zero_cell"]
    Mz = sp.csc_matrix(np.array(z["M"], dtype=float))
    r = oracle.gficf_csc(2, 2, Mz.indptr, Mz.indices, Mz.data, z["min"], z["max"])
    assert np.allclose(_dense(r, 2), np.array(z["dense"]), rtol=0, atol=1e-15)


def test_gficf_cxx_vs_numpy():
    for G, N, mn, mx, seed in [(400, 300, 0.05, 1.0, 1), (250, 350, 0.0, 1.0, 2), (300, 200, 0.1, 0.5, 3)]:
        cp, ri, x = synth.counts_csc(G, N, seed=seed)
        r = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
        r2 = oracle_np.gficf_np(sp.csc_matrix((x, ri, cp), shape=(G, N)), mn, mx)
        assert np.array_equal(r["keep"], r2["keep"])
        assert np.array_equal(r["nt"], r2["nt"])
        assert np.allclose(r["w"], r2["w"], rtol=1e-15, atol=0)
        assert np.array_equal(r["rowidx"], r2["gficf"].indices)
        assert np.array_equal(r["colptr"], r2["gficf"].indptr)
        assert np.allclose(r["x"], r2["gficf"].data, rtol=1e-13, atol=1e-16)


def test_gficf_supplied_weights_and_properties():
    G, N = 300, 200
    cp, ri, x = synth.counts_csc(G, N, seed=4)
    w_in = 0.5 + synth.rand_unit(5, np.arange(G))
    r = oracle.gficf_csc(G, N, cp, ri, x, 0.0, 2.0, w_in=w_in)
    r2 = oracle_np.gficf_np(sp.csc_matrix((x, ri, cp), shape=(G, N)), 0.0, 2.0, w_in=w_in)
    assert np.allclose(r["x"], r2["gficf"].data, rtol=1e-13, atol=1e-16)
    assert np.array_equal(r["w"][r["keep"]], w_in[r["keep"]])
    # every non-empty cell has unit L2 norm
    D = _dense(r, N)
    assert np.allclose(np.sqrt((D ** 2).sum(axis=0)), 1.0, rtol=1e-12)


def test_gficf_empty_cell_and_explicit_zero():
    # cell 1 empty; cell 2 holds an explicit zero (does not count in nt, R/gficf.R:40)
    colptr = np.array([0, 2, 2, 4], dtype=np.int64)
    rowidx = np.array([0, 1, 0, 1], dtype=np.int32)
    x = np.array([1.0, 3.0, 0.0, 2.0])
    r = oracle.gficf_csc(2, 3, colptr, rowidx, x, 0.0, 1.0)
    assert r["nt"].tolist() == [1, 2]
    assert r["colptr"].tolist() == [0, 2, 2, 4]
    D = _dense(r, 3)
    assert np.all(D[:, 1] == 0) and D[0, 2] == 0.0 and D[1, 2] == 1.0


def test_gficf_threads_do_not_change_results():
    # the multi-threaded restatement (bench.py's GF-ICF cpu_baseline) gives the single-threaded one's bits
    for G, N, mn, mx, seed, T in [(400, 300, 0.05, 1.0, 1, 3), (250, 350, 0.0, 1.0, 2, 8), (300, 7, 0.1, 0.9, 3, 16), (50, 1, 0.0, 1.0, 4, 4)]:
        cp, ri, x = synth.counts_csc(G, N, seed=seed)
        w_in = None if seed != 2 else 0.5 + synth.rand_unit(5, np.arange(G))
        a = oracle.gficf_csc(G, N, cp, ri, x, mn, mx, w_in=w_in)
        b = oracle.gficf_csc(G, N, cp, ri, x, mn, mx, w_in=w_in, threads=T)
        for key in ("keep", "nt", "w", "colptr", "rowidx", "x"):
            assert np.array_equal(a[key], b[key]), key
        assert a["G_kept"] == b["G_kept"]
    colptr = np.array([0, 2, 2, 4], dtype=np.int64)            # empty cell, explicit zero
    rowidx = np.array([0, 1, 0, 1], dtype=np.int32)
    x = np.array([1.0, 3.0, 0.0, 2.0])
    a, b = oracle.gficf_csc(2, 3, colptr, rowidx, x, 0.0, 1.0), oracle.gficf_csc(2, 3, colptr, rowidx, x, 0.0, 1.0, threads=2)
    assert np.array_equal(a["x"], b["x"]) and a["nt"].tolist() == b["nt"].tolist() == [1, 2]
    with pytest.raises(ValueError):
        oracle.gficf_csc(2, 3, colptr, np.array([0, 5, 0, 1], dtype=np.int32), x, 0.0, 1.0, threads=2)


def _boundary_matrix():
    """20 cells; gene g is present (value 1) in the first nt[g] cells.  With min = 0.05 (N*min = 1.0) and max = 0.5 (N*max = 10.0)
    the reference keeps a gene iff nt > 1.0 and nt <= 10.0 (R/gficf.R:41: strict below, inclusive above)."""
    N = 20
    nts = [0, 1, 2, 9, 10, 11, 20]
    M = np.zeros((len(nts), N))
    for g, n in enumerate(nts):
        M[g, :n] = 1.0
    keep = np.array([False, False, True, True, True, False, False])     # by hand from the two inequalities
    return sp.csc_matrix(M), np.array(nts), keep


def test_gficf_filter_boundaries_hand_derived():
    M, nts, keep = _boundary_matrix()
    G, N = M.shape
    r = oracle.gficf_csc(G, N, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, 0.05, 0.5)
    assert r["keep"].tolist() == keep.tolist()
    assert r["nt"].tolist() == [0, 0, 2, 9, 10, 0, 0]                 # nt of dropped genes is reported as 0
    # closed form: kept genes 2, 3, 4 with nt 2, 9, 10; w = log(21 / (nt + 1)); cell 0 holds all three with x = 1:
    # tf = 1/3 each, value_g = w_g / 3, then divided by the cell's L2 norm
    w = np.log(21.0 / (np.array([2, 9, 10]) + 1.0))
    assert np.allclose(r["w"][keep], w, rtol=1e-15)
    col0 = r["x"][r["colptr"][0]:r["colptr"][1]]
    assert np.allclose(col0, (w / 3) / np.sqrt(((w / 3) ** 2).sum()), rtol=1e-14)
    # cell 9 (10th) holds genes with nt >= 10 among the kept ones: only gene 4 -> a single entry of value 1
    col9 = r["x"][r["colptr"][9]:r["colptr"][10]]
    assert col9.tolist() == [1.0]
    # cells 10..19 hold no kept gene
    assert r["colptr"][10] == r["colptr"][20]
    r2 = oracle_np.gficf_np(M, 0.05, 0.5)
    assert np.array_equal(r2["keep"], keep)


def test_gficf_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "gficf_cases.npz"))
    for nm in ("g600_n400", "g300_n500_nofilter", "g500_n300_max"):
        G, N, seed = (int(v) for v in z[nm + "_meta"])
        mn, mx = z[nm + "_prop"]
        cp, ri, x = synth.counts_csc(G, N, seed=seed)
        r = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
        assert np.array_equal(r["keep"], z[nm + "_keep"])
        assert np.array_equal(r["nt"], z[nm + "_nt"])
        assert np.array_equal(r["colptr"], z[nm + "_colptr"])
        assert np.array_equal(r["rowidx"], z[nm + "_rowidx"])
        assert np.allclose(r["x"], z[nm + "_x"], rtol=1e-14, atol=0)
        assert np.allclose(r["w"], z[nm + "_w"], rtol=1e-14, atol=0)


# ------------------------------------------------------------------------------- kNN (N2)
@pytest.mark.parametrize("metric", ["manhattan", "euclidean", "cosine"])
def test_knn_oracle_cpp_vs_float64_restatement(metric):
    """The f32 C++ brute force against the float64 numpy restatement: same neighbours except where
    two float64 distances are within f32 rounding of each other; distances within 1e-5."""
    rng = np.random.default_rng(12)
    X = rng.normal(size=(400, 24)) * rng.uniform(0.3, 3.0, size=(1, 24))
    idx, dist = oracle.knn(X, 15, metric, nthreads=4)
    nidx, ndist = oracle_np.knn_np(X, 15, metric)
    assert np.allclose(dist, ndist, rtol=1e-5, atol=1e-5)
    assert (idx == nidx).mean() > 0.995
    assert np.array_equal(idx[:, 0], np.arange(1, 401)) or metric == "cosine"


def test_knn_oracle_known_answer_and_ties():
    # 1-d points 0, 1, 3, 6 and a duplicate of 1: distances are exact small integers
    X = np.array([[0.0], [1.0], [3.0], [6.0], [1.0]])
    idx, dist = oracle.knn(X, 3, "manhattan")
    assert idx.tolist() == [[1, 2, 5], [2, 5, 1], [3, 2, 5], [4, 3, 2], [2, 5, 1]]
    assert dist.tolist() == [[0, 1, 1], [0, 0, 1], [0, 2, 2], [0, 3, 5], [0, 0, 1]]
    idx2, dist2 = oracle.knn(X, 3, "euclidean")
    assert idx2.tolist() == idx.tolist() and dist2.tolist() == dist.tolist()


# ------------------------------------------------------------- the serial entry jaccard_coeff
def test_serial_jaccard_coeff_oracle_cpp_vs_numpy_and_vs_parallel_entry():
    """src/jaccard_coeff.cpp:19-44: set intersection, rows with u > 0 packed from the top.  C++ vs numpy set algebra;
    on rows without duplicate ids it holds the parallel entry's non-zero rows, in order."""
    mat = synth.knn_windowed(400, 9, seed=21)
    ser = oracle.jaccard_coeff(mat)
    assert np.array_equal(ser, oracle_np.jaccard_coeff_np(mat))
    full, _ = oracle.jaccard(mat)
    kept = full[full[:, 2] > 0]
    assert np.array_equal(ser[:len(kept)], kept) and not ser[len(kept):].any()
    # a row that holds an id three times: multiset (parallel entry) and set (serial entry) counts differ
    mat[3, :3] = mat[3, 0]
    mat[int(mat[3, 0]) - 1, :2] = mat[3, 0]
    ser = oracle.jaccard_coeff(mat)
    assert np.array_equal(ser, oracle_np.jaccard_coeff_np(mat))
    full, _ = oracle.jaccard(mat)
    kept = full[full[:, 2] > 0]
    assert not np.array_equal(ser[:len(kept), 2], kept[:, 2])


def test_serial_jaccard_coeff_known_answer():
    # 3 cells, k = 2: row 1 = {2,3}, row 2 = {1,3}, row 3 = {3,3} (an id twice: a set of one element)
    mat = np.array([[2, 3], [1, 3], [3, 3]], dtype=np.int32)
    got = oracle.jaccard_coeff(mat)
    # edges in (i, j) order with u: (1,2): {2,3}∩{1,3} = 1; (1,3): {2,3}∩{3} = 1; (2,1): 1; (2,3): {1,3}∩{3} = 1; (3,3): 1; (3,3): 1
    want = np.array([[1, 2, 1 / 3], [1, 3, 1 / 3], [2, 1, 1 / 3], [2, 3, 1 / 3], [3, 3, 1 / 3], [3, 3, 1 / 3]])
    assert np.allclose(got, want) and np.array_equal(got[:, :2], want[:, :2])


@pytest.mark.parametrize("icf_type", ["classic", "prob", "smooth"])
@pytest.mark.parametrize("norm", ["l2", "l1"])
def test_gficf_oracle_helper_branches_cpp_vs_numpy(icf_type, norm):
    """The other branches of the reference's helpers getIdfW(type) (R/gficf.R:89-91) and l.norm(norm) (R/gficf.R:100),
    which gficf() itself never takes: the two restatements agree, and one closed-form value is checked by hand."""
    cp, ri, x = synth.counts_csc(250, 180, seed=6)
    M = sp.csc_matrix((x, ri, cp), shape=(250, 180))
    a = oracle.gficf_csc(250, 180, cp, ri, x, 0.05, 1.0, None, icf_type, norm)
    b = oracle_np.gficf_np(M, 0.05, 1.0, None, icf_type, norm)
    assert np.array_equal(a["keep"], b["keep"]) and np.array_equal(a["rowidx"], b["gficf"].indices)
    assert np.allclose(a["x"], b["gficf"].data, rtol=1e-12, atol=1e-15)
    assert np.allclose(a["w"], b["w"], rtol=1e-14)
    # hand check on the 4 x 3 example of the golden known answers: nt = [2, 2, 3, 1], N = 3
    Mk = np.array([[1, 0, 3], [1, 2, 0], [2, 2, 1], [0, 4, 0]], dtype=np.float64)
    r = oracle_np.gficf_np(sp.csc_matrix(Mk), 0.0, 1.0, None, icf_type, norm)
    want_w = {"classic": np.log(4 / np.array([3, 3, 4, 2.0])), "prob": np.log((3 - np.array([2, 2, 3, 1.0])) / np.array([2, 2, 3, 1.0])),
              "smooth": np.log(1 + 3 / np.array([2, 2, 3, 1.0]))}[icf_type]
    with np.errstate(divide="ignore"):
        assert np.allclose(r["w"], want_w, equal_nan=True)


def test_transpose_restatement_matches_scipy():
    """t(data$gficf) (R/dimensinalityReduction.R:33): the numpy restatement against scipy's transpose, explicit zeros kept."""
    import scipy.sparse as sp

    from gficf_amd import synth
    from oracle import oracle_np

    G, N = 700, 450
    cp, ri, x = synth.counts_csc(G, N, seed=12)
    x = x.copy()
    x[::9] = 0.0
    ptr, idx, val = oracle_np.transpose_np(G, N, cp, ri, x)
    S = sp.csc_matrix((x, ri, cp), shape=(G, N)).T.tocsc()
    S.sort_indices()
    assert np.array_equal(S.indptr, ptr) and np.array_equal(S.indices, idx) and np.array_equal(S.data, val)
    assert len(val) == len(x)                                       # nothing dropped
    # known answer: the 4 x 3 matrix of SURVEY.md 8c
    M = sp.csc_matrix(np.array([[1, 0, 3], [1, 2, 0], [2, 2, 1], [0, 4, 0]], dtype=float))
    ptr, idx, val = oracle_np.transpose_np(4, 3, M.indptr, M.indices, M.data)
    assert ptr.tolist() == [0, 2, 4, 7, 8] and idx.tolist() == [0, 2, 0, 1, 0, 1, 2, 1]
    assert val.tolist() == [1, 3, 1, 2, 2, 2, 1, 4]


def test_modularity_restatement_against_reference_outputs(golden_dir):
    """calcQualityFunction (src/ModularityOptimizer.cpp:461-482) restated in numpy against the modularity the reference
    itself printed for its own labels (tests/golden/louvain_cases.npz: outputs of a build of the reference's optimiser),
    and against the binary run live when oracle/_ref holds it."""
    import os

    import scipy.sparse as sp

    from oracle import oracle_np

    z = np.load(os.path.join(golden_dir, "louvain_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) == 4
    for n in names:
        N = len(z[n + "/indptr"]) - 1
        A = sp.csc_matrix((z[n + "/data"], z[n + "/indices"], z[n + "/indptr"]), shape=(N, N))
        assert (abs(A - A.T)).nnz == 0
        res = float(z[n + "/params"][0])
        labels = z[n + "/labels"]
        assert abs(oracle_np.modularity_np(A, labels, res) - float(z[n + "/printed_q"][0])) < 6e-5
        sizes = np.bincount(labels)
        assert (np.diff(sizes) <= 0).all()                                  # orderClustersByNNodes
    if oracle.build_ref() is not None:                                      # live: the same numbers again
        n = "planted3"
        N = len(z[n + "/indptr"]) - 1
        A = sp.csc_matrix((z[n + "/data"], z[n + "/indices"], z[n + "/indptr"]), shape=(N, N))
        res, alg, n_start, n_iter, seed = z[n + "/params"]
        labels, q = oracle.modularity_reference(A, res, int(alg), int(n_start), int(n_iter), int(seed))
        assert np.array_equal(labels, z[n + "/labels"]) and q == float(z[n + "/printed_q"][0])


def test_jaccard_noninteger_doubles_reference_truncation_hand_derived():
    """What the reference does with non-integer double ids (src/rcpp_parallel_jaccard_coeff.cpp:28-46): the row is addressed by
    kk = (int)(v - 1), the rows are intersected as the doubles they hold.  Expectations by hand (3 cells, k = 2)."""
    mat = np.array([[2.0, 3.5], [1.0, 3.0], [1.2, 2.0]])
    rm, u = oracle.jaccard(mat)
    assert u.tolist() == [0, 1, 0, 0, 1, 0]
    third = 1.0 / (2.0 * 2 - 1)
    want = np.zeros((6, 3))
    want[1] = (1.0, 3.0, third)          # 3.5 -> row 3 = {1.2, 2.0}; shares 2.0 with row 1
    want[4] = (3.0, 1.0, third)          # 1.2 -> row 1 = {2.0, 3.5}; shares 2.0 with row 3
    assert np.array_equal(rm, want)
    # a value in (0, 1) truncates to row 1 like 1.0 does; 0 and N + 1 are outside what the reference can address
    rm2, _ = oracle.jaccard(np.array([[0.5, 2.0], [0.5, 1.0]]))
    assert rm2[0].tolist() == [1.0, 1.0, 1.0] and rm2[2].tolist() == [2.0, 1.0, 1.0 / 3.0]
    for bad in (0.0, 3.0, -1.5, np.nan):
        with pytest.raises(ValueError):
            oracle.jaccard(np.array([[bad, 2.0], [1.0, 2.0]]))


from tests.helpers.closed_form import cyclic_window_expected, cyclic_window_matrix  # noqa: E402


@pytest.mark.parametrize("N,k", [(40, 15), (64, 30), (700, 50), (300, 100), (620, 300)])
def test_oracle_against_the_closed_form_of_cyclic_windows(N, k):
    """An answer that is derived, not computed: pins the oracle's restatement of :24-55 at sizes beyond the hand-worked toys
    (and, in tests/test_jaccard_gpu.py, every kernel family of the HIP path against the same closed form)."""
    mat = cyclic_window_matrix(N, k)
    want, wu = cyclic_window_expected(N, k)
    got, u = oracle.jaccard(mat, nthreads=4)
    assert np.array_equal(u, wu) and np.array_equal(got, want)
    assert np.array_equal(oracle.jaccard(mat.astype(np.float64), nthreads=2)[0], want)


@pytest.mark.parametrize("G,N,s,n_rare", [(50, 200, 7, 20), (400, 2000, 31, 300)])
def test_gficf_oracle_against_the_closed_form_of_a_circulant_matrix(G, N, s, n_rare):
    """GF-ICF with an answer derived by algebra (tests/helpers/closed_form.py::circulant_counts): filter, GF, ICF, L2 and the compaction of
    the kept genes, both restatements of R/gficf.R:40-104."""
    from tests.helpers.closed_form import circulant_counts

    M, want, keep, nt, w = circulant_counts(G, N, s, n_rare)
    rc = oracle.gficf_csc(G + n_rare, N, M.indptr.astype(np.int64), M.indices, M.data, 0.05, 1.0)
    rn = oracle_np.gficf_np(M, 0.05, 1.0)
    gn = rn["gficf"].tocsc()
    gn.sort_indices()
    for ref, ri, cp, xv in ((rc, rc["rowidx"], rc["colptr"], rc["x"]), (rn, gn.indices, gn.indptr, gn.data)):
        assert np.array_equal(np.asarray(ref["keep"]).astype(bool), keep) and np.array_equal(np.asarray(ref["nt"])[:G], nt[:G])
        assert np.allclose(np.asarray(ref["w"])[:G], w, rtol=1e-14)
        assert np.array_equal(ri, want.indices) and np.array_equal(cp, want.indptr)
        assert np.allclose(xv, want.data, rtol=1e-13, atol=0)


@pytest.mark.parametrize("N,k", [(50, 16), (90, 30), (700, 300)])
def test_oracle_multiset_and_set_semantics_against_the_closed_form(N, k):
    """Rows that name every id twice: std::set_intersection's multiset count (parallel entry) is twice Rcpp::intersect's set count (serial
    entry), both derived by counting (tests/helpers/closed_form.py::cyclic_window_matrix_twice)."""
    from tests.helpers.closed_form import cyclic_window_matrix_twice, cyclic_window_twice_expected

    mat = cyclic_window_matrix_twice(N, k)
    want, wu = cyclic_window_twice_expected(N, k)
    got, u = oracle.jaccard(mat, nthreads=4)
    assert np.array_equal(u, wu) and np.array_equal(got, want)
    ws, _ = cyclic_window_twice_expected(N, k, set_semantics=True)
    ws = ws[ws[:, 2] > 0]                                                        # the serial entry packs its rows from the top (:36-41)
    gs = oracle.jaccard_coeff(mat)
    assert np.array_equal(gs[:len(ws)], ws) and not gs[len(ws):].any()


@pytest.mark.parametrize("N,k,metric", [(400, 7, "manhattan"), (3000, 16, "euclidean")])
def test_knn_oracle_against_the_derived_answer_on_a_line(N, k, metric):
    """The search oracle (oracle/knn_oracle.cpp, f32 brute force) held to an answer by counting: points at the integer positions of a line,
    the k nearest of i (itself included, ties to the smaller index) are i, i-1, i+1, i-2, ... (tests/test_knn_gpu.py holds the HIP search to
    the same)."""
    X = np.stack([np.arange(N, dtype=np.float64), np.full(N, 3.0)], axis=1)
    idx, _ = oracle.knn(X, k, metric, nthreads=4)
    i = np.arange(N, dtype=np.int64)[:, None]
    cand = i + np.arange(-k, k + 1, dtype=np.int64)[None, :]
    ok = (cand >= 0) & (cand < N)
    dist = np.where(ok, np.abs(cand - i), 10 * N)
    order = np.lexsort((np.where(ok, cand, 10 * N), dist), axis=1)[:, :k]
    assert np.array_equal(idx, (np.take_along_axis(cand, order, axis=1) + 1).astype(idx.dtype))


@pytest.mark.parametrize("c,m", [(8, 5), (60, 10), (200, 20)])
def test_modularity_restatement_and_reference_binary_against_the_ring_of_cliques(c, m):
    """The quality function's restatement, and — when oracle/_ref holds it — the REFERENCE's own optimiser, against an answer derived by counting
    (tests/helpers/closed_form.py::ring_of_cliques): one community per clique, Q = m (m - 1) / (m (m - 1) + 2) - resolution / c.  The device
    Louvain is held to the same answer in tests/test_louvain_gpu.py."""
    from tests.helpers import closed_form

    res = 0.8
    A, clique = closed_form.ring_of_cliques(c, m, 0.37)
    want = closed_form.ring_of_cliques_modularity(c, m, res)
    assert (abs(A - A.T)).nnz == 0 and abs(oracle_np.modularity_np(A, clique, res) - want) < 1e-12
    if oracle.build_ref() is not None:
        labels, q = oracle.modularity_reference(A, res, 1, 10, 10, 0)
        pairs = np.unique(np.stack([labels, clique], axis=1), axis=0)
        assert len(pairs) == c == len(np.unique(labels)) and abs(q - want) < 6e-5        # (the binary prints four decimals)
