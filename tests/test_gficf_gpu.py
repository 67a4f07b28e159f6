"""GPU parity: the HIP GF-ICF path (through the C ABI) against the CPU oracle.
Bar: keep mask, nt, structure (colptr / renumbered rowidx) exact; values and weights
within 1e-6 (absolute and relative) as BASELINE.json states — f64 throughout, observed
differences are ~1e-16 (summation order only)."""
import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import gficf_amd
import oracle
from gficf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-6


def check_against_oracle(res, ref, N):
    keep = ref["keep"]
    assert np.array_equal(res["genes"], np.flatnonzero(keep))
    assert np.array_equal(res["nt"], ref["nt"][keep])
    g = res["gficf"]
    assert g.shape == (int(keep.sum()), N)
    assert np.array_equal(g.indptr, ref["colptr"])
    assert np.array_equal(g.indices, ref["rowidx"])
    assert np.allclose(g.data, ref["x"], rtol=TOL, atol=TOL)
    assert np.allclose(res["w"], ref["w"][keep], rtol=TOL, atol=TOL)
    # much tighter in practice
    assert np.abs(g.data - ref["x"]).max(initial=0.0) < 1e-12


def test_known_answers(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        known = json.load(f)["gficf"]
    for nm in ("basic", "filter", "w_zero_cell"):
        c = known[nm]
        M = sp.csc_matrix(np.array(c["M"], dtype=float))
        res = gficf_amd.gficf(M, cell_proportion_max=c["max"], cell_proportion_min=c["min"], normalize=False, verbose=False)
        assert np.allclose(res["gficf"].toarray(), np.array(c["dense"]), rtol=0, atol=1e-14), nm
        if "w" in c:
            assert np.allclose(res["w"], np.array(c["w"]), rtol=0, atol=1e-15)
        if "keep" in c:
            assert res["genes"].tolist() == [i for i, kp in enumerate(c["keep"]) if kp]


@pytest.mark.parametrize("G,N,mn,mx,seed", [(600, 400, 0.05, 1.0, 7), (300, 500, 0.0, 1.0, 8), (500, 300, 0.02, 0.6, 9),
                                            (5000, 3000, 0.05, 1.0, 7), (20000, 2000, 0.05, 1.0, 3),
                                            (40000, 500, 0.01, 1.0, 4), (1, 50, 0.0, 1.0, 5), (50, 1, 0.0, 1.0, 6)])
def test_matches_oracle(G, N, mn, mx, seed):
    cp, ri, x = synth.counts_csc(G, N, seed=seed)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    res = gficf_amd.gficf(M, cell_proportion_max=mx, cell_proportion_min=mn, normalize=False, verbose=False)
    ref = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
    check_against_oracle(res, ref, N)
    assert res["param"] == {"cell_proportion_max": mx, "cell_proportion_min": mn, "normalized": False}
    assert (res["rawCounts"] != M[res["genes"], :]).nnz == 0


def test_verbose_prints_the_reference_progress_lines(capsys):
    """gficf(verbose = TRUE) prints the reference's tsmessage lines (R/gficf.R:58,87,68,99 through R/util.R:30-39: a %T stamp, then the text,
    on stderr like R's message()); verbose = FALSE prints nothing."""
    import re

    cp, ri, x = synth.counts_csc(300, 200, seed=1)
    M = sp.csc_matrix((x, ri, cp), shape=(300, 200))
    gficf_amd.gficf(M, normalize=False, verbose=True)
    err = capsys.readouterr().err.splitlines()
    assert [re.sub(r"^\d\d:\d\d:\d\d ", "", ln) for ln in err] == ["Apply GF transformation..", "Compute ICF weigth..", "Applay ICF..", "Apply l2"]
    assert all(re.match(r"^\d\d:\d\d:\d\d ", ln) for ln in err)
    gficf_amd.gficf(M, normalize=False, verbose=False)
    cap = capsys.readouterr()
    assert cap.err == "" and cap.out == ""


def test_raw_counts_come_back_with_the_result_and_are_the_row_subset():
    """gficf(storeRaw=True): $rawCounts = normCounts' M[keep, ] (reference R/gficf.R:40,22) comes from the finish call itself
    (gficf_normalize_csc_host_finish_raw: the result's structure, the counts gathered by host threads while the result crosses PCIe) — equal to
    scipy's row subsetting entry for entry, at a size that takes several host threads, explicit zeros kept, integer counts staying integers,
    index vectors of its own; a finish call handed another matrix than the plan's is refused."""
    import ctypes

    from gficf_amd import _lib
    from gficf_amd.api import _np_ptr

    G, N = 6000, 9000
    cp, ri, x = synth.counts_csc(G, N, median_frac=0.2, seed=11)
    assert len(x) > 6_000_000                                            # (three shares of host threads and more)
    x[::101] = 0.0
    # three more cells at the end that hold nothing but a gene of their own (dropped: one cell each) — the kept rows have nothing in the LAST cells
    # (the first version of the gather stored one slot past such a tail: tools/fuzz_gpu.py found it)
    tail = sp.csc_matrix((np.full(3, 2.0), (np.arange(G, G + 3), np.arange(3))), shape=(G + 3, 3))
    for dt in (np.float64, np.int32):
        M = sp.hstack([sp.vstack([sp.csc_matrix((x.astype(dt), ri, cp), shape=(G, N)), sp.csc_matrix((3, N), dtype=dt)], format="csc"), tail.astype(dt)], format="csc")
        res = gficf_amd.gficf(M, normalize=False, verbose=False)
        assert res["genes"].max() < G and np.diff(res["gficf"].indptr)[-3:].tolist() == [0, 0, 0]
        raw, want = res["rawCounts"], M[res["genes"], :]
        assert raw.dtype == dt and raw.shape == want.shape
        assert np.array_equal(raw.indptr, want.indptr) and np.array_equal(raw.indices, want.indices) and np.array_equal(raw.data, want.data)
        assert np.array_equal(raw.indices, res["gficf"].indices) and not np.shares_memory(raw.indices, res["gficf"].indices)
        assert not np.shares_memory(raw.indptr, res["gficf"].indptr)
    ref = oracle.gficf_csc(G + 3, N + 3, M.indptr.astype(np.int64), M.indices, M.data.astype(np.float64), 0.05, 1.0)
    check_against_oracle(res, ref, N + 3)
    assert gficf_amd.gficf(M, normalize=False, verbose=False, storeRaw=False).get("rawCounts") is None
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    want = M[gficf_amd.gficf(M, normalize=False, verbose=False, storeRaw=False)["genes"], :]
    # another matrix at the finish call: the kept entries of some cell do not match the plan's count
    L, ctx = _lib.load(), gficf_amd.default_context()
    gk, nk = ctypes.c_int64(0), ctypes.c_int64(0)
    ri32 = ri.astype(np.int32)
    assert L.gficf_normalize_csc_host_plan(ctx.handle, G, N, _np_ptr(cp), 1, _np_ptr(ri32), _np_ptr(x), 0.05, 1.0, None, ctypes.byref(gk), ctypes.byref(nk)) == 0
    other = ((ri32 + 1) % G).astype(np.int32)
    ocp, ori, ox, rx = np.zeros(N + 1, dtype=np.int64), np.empty(nk.value, dtype=np.int32), np.empty(nk.value), np.empty(nk.value)
    rc = L.gficf_normalize_csc_host_finish_raw(ctx.handle, None, None, None, _np_ptr(ocp), _np_ptr(ori), _np_ptr(ox), _np_ptr(other), _np_ptr(x), None, _np_ptr(rx))
    assert _lib.STATUS_NAMES[rc] == "GFICF_ERR_BAD_CSC" and b"not the matrix of the plan" in L.gficf_last_error()
    res2 = gficf_amd.gficf(M, normalize=False, verbose=False)              # the context is usable afterwards
    assert np.array_equal(res2["rawCounts"].data, want.data)


@pytest.mark.parametrize("shape", ["giants_first", "giants_last", "empty_blocks", "one_cell_has_it_all", "all_equal"])
def test_skewed_cell_lengths_above_the_range_search_threshold(shape):
    """The scaling and kept-count passes cut the cells into contiguous ranges of equal stored entries (a search on colptr by
    every workgroup, used from 16384 cells on) and hand a range's cells to the waves through a counter: cell-length
    distributions that put range boundaries inside runs of empty cells, on giant cells, or all in one place."""
    rng = np.random.default_rng(123)
    G, N = 4000, 20000
    lens = rng.integers(0, 6, size=N)
    if shape == "giants_first":
        lens[:40] = G
    elif shape == "giants_last":
        lens[-40:] = G
    elif shape == "empty_blocks":
        lens[:6000] = 0
        lens[9000:15000] = 0
        lens[-500:] = 0
        lens[7000:7010] = 3500
    elif shape == "one_cell_has_it_all":
        lens[:] = 0
        lens[12345] = G
        lens[3] = 1
    elif shape == "all_equal":
        lens[:] = 17
    cp = np.zeros(N + 1, dtype=np.int64)
    cp[1:] = np.cumsum(lens)
    ri = np.empty(int(cp[-1]), dtype=np.int32)
    for c in np.flatnonzero(lens):
        ri[cp[c]:cp[c + 1]] = np.sort(rng.choice(G, size=int(lens[c]), replace=False)) if lens[c] < G else np.arange(G)
    x = 1.0 + rng.integers(0, 9, size=ri.size).astype(np.float64)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    for mn in (0.0, 0.001):
        res = gficf_amd.gficf(M, 1.0, mn, normalize=False, verbose=False)
        ref = oracle.gficf_csc(G, N, cp, ri, x, mn, 1.0, threads=4)
        check_against_oracle(res, ref, N)


def test_filter_boundaries_hand_derived():
    """Strict below, inclusive above (R/gficf.R:41), on counts that sit exactly on N*min and N*max; expectations by hand,
    not from the oracle (tests/test_oracle.py holds the same case for the oracle)."""
    N = 20
    nts = [0, 1, 2, 9, 10, 11, 20]
    M = np.zeros((len(nts), N))
    for g, n in enumerate(nts):
        M[g, :n] = 1.0
    res = gficf_amd.gficf(sp.csc_matrix(M), 0.5, 0.05, normalize=False, verbose=False)
    assert res["genes"].tolist() == [2, 3, 4]
    assert res["nt"].tolist() == [2, 9, 10]
    w = np.log(21.0 / (np.array([2, 9, 10]) + 1.0))
    assert np.allclose(res["w"], w, rtol=1e-14)
    D = res["gficf"].toarray()
    assert np.allclose(D[:, 0], (w / 3) / np.sqrt(((w / 3) ** 2).sum()), rtol=1e-13)
    assert D[:, 9].tolist() == [0.0, 0.0, 1.0] and not D[:, 10:].any()


def test_golden_fixtures(golden_dir):
    z = np.load(os.path.join(golden_dir, "gficf_cases.npz"))
    for nm in ("g600_n400", "g300_n500_nofilter", "g500_n300_max"):
        G, N, seed = (int(v) for v in z[nm + "_meta"])
        mn, mx = z[nm + "_prop"]
        cp, ri, x = synth.counts_csc(G, N, seed=seed)
        res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(G, N)), mx, mn, normalize=False, verbose=False)
        assert np.array_equal(res["genes"], np.flatnonzero(z[nm + "_keep"]))
        assert np.array_equal(res["gficf"].indptr, z[nm + "_colptr"])
        assert np.array_equal(res["gficf"].indices, z[nm + "_rowidx"])
        assert np.allclose(res["gficf"].data, z[nm + "_x"], rtol=TOL, atol=TOL)


def test_int32_and_int64_colptr_agree():
    cp, ri, x = synth.counts_csc(800, 600, seed=2)
    M32 = sp.csc_matrix((x, ri, cp.astype(np.int32)), shape=(800, 600))
    M64 = sp.csc_matrix((x, ri, cp.astype(np.int64)), shape=(800, 600))
    M64.indptr = M64.indptr.astype(np.int64)
    a = gficf_amd.gficf(M32, normalize=False, verbose=False)["gficf"]
    b = gficf_amd.gficf(M64, normalize=False, verbose=False)["gficf"]
    assert np.array_equal(a.data, b.data) and np.array_equal(a.indices, b.indices)


def test_supplied_weights_second_caller():
    # embedNewCells(): reference R/cellClassifier.R:50-53 — ICF weights supplied, not recomputed
    G, N = 700, 300
    cp, ri, x = synth.counts_csc(G, N, seed=12)
    w_in = 0.25 + synth.rand_unit(13, np.arange(G))
    out, genes = gficf_amd.gficf_with_weights(sp.csc_matrix((x, ri, cp), shape=(G, N)), w_in)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.0, 2.0, w_in=w_in)
    assert np.array_equal(genes, np.flatnonzero(ref["keep"]))
    assert np.array_equal(out.indices, ref["rowidx"]) and np.array_equal(out.indptr, ref["colptr"])
    assert np.allclose(out.data, ref["x"], rtol=TOL, atol=TOL)


def test_empty_cells_explicit_zeros_and_empty_matrix():
    colptr = np.array([0, 2, 2, 4, 4], dtype=np.int32)
    rowidx = np.array([0, 1, 0, 1], dtype=np.int32)
    x = np.array([1.0, 3.0, 0.0, 2.0])
    M = sp.csc_matrix((x, rowidx, colptr), shape=(2, 4))
    res = gficf_amd.gficf(M, 1, 0.0, normalize=False, verbose=False)
    ref = oracle.gficf_csc(2, 4, colptr, rowidx, x, 0.0, 1.0)
    check_against_oracle(res, ref, 4)
    # nothing stored at all
    E = sp.csc_matrix((5, 3))
    r = gficf_amd.gficf(E, normalize=False, verbose=False)
    assert r["gficf"].shape == (0, 3) and r["gficf"].nnz == 0


def test_all_genes_filtered_out():
    cp, ri, x = synth.counts_csc(300, 200, seed=3)
    res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(300, 200)), 1, 1.0, normalize=False, verbose=False)
    assert res["gficf"].shape == (0, 200) and res["gficf"].nnz == 0


def test_malformed_csc_rejected():
    ctx = gficf_amd.default_context()
    from gficf_amd import api

    cp = np.array([0, 2, 1], dtype=np.int64)
    bad = sp.csc_matrix((3, 2))
    bad.indptr, bad.indices, bad.data = cp, np.array([0, 1], dtype=np.int32), np.array([1.0, 2.0])
    with pytest.raises(gficf_amd.GficfError) as ei:
        api._normalize_csc_host(bad, 0.0, 1.0, None, ctx)
    assert ei.value.status == "GFICF_ERR_BAD_CSC"
    bad2 = sp.csc_matrix((3, 2))
    bad2.indptr, bad2.indices, bad2.data = np.array([0, 1, 2], dtype=np.int64), np.array([0, 7], dtype=np.int32), np.array([1.0, 2.0])
    with pytest.raises(gficf_amd.GficfError) as ei:
        api._normalize_csc_host(bad2, 0.0, 1.0, None, ctx)
    assert ei.value.status == "GFICF_ERR_BAD_CSC"


def test_device_pipeline_and_cell_block_seam():
    """Device-resident stages; two column blocks + summed counts == the single-shot result
    (the multi-GPU seam exercised on one GPU)."""
    import torch

    ops = gficf_amd.HipOps(0)
    G, N = 3000, 2500
    cp, ri, x = synth.counts_csc(G, N, seed=17)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0)
    d = lambda a: torch.from_numpy(a).cuda()
    ws = ops.gficf_csc(G, N, d(cp), d(ri), d(x), 0.05, 1.0)
    ops.sync()
    nk = int(ws["out_colptr"][N])
    assert int(ws["gkept"][0]) == ref["G_kept"] and nk == len(ref["x"])
    assert np.array_equal(ws["out_colptr"].cpu().numpy(), ref["colptr"])
    assert np.array_equal(ws["out_rowidx"][:nk].cpu().numpy(), ref["rowidx"])
    assert np.allclose(ws["out_x"][:nk].cpu().numpy(), ref["x"], rtol=TOL, atol=TOL)
    assert np.array_equal(ws["keep"].cpu().numpy().astype(bool), ref["keep"])
    # two blocks of cells
    cut = 1100
    blocks = []
    nt = torch.zeros(G, dtype=torch.int64, device="cuda")
    for b, e in ((0, cut), (cut, N)):
        lcp = d((cp[b:e + 1] - cp[b]).astype(np.int64))
        lri, lx = d(ri[cp[b]:cp[e]]), d(x[cp[b]:cp[e]])
        blocks.append((e - b, lcp, lri, lx))
        ops.csc_count(G, e - b, lcp, lri, lx, nt)
    xs, rs = [], []
    for n, lcp, lri, lx in blocks:
        w2 = ops.csc_workspace(G, n, int(lri.numel()))
        ops.csc_genes(G, N, nt, 0.05, 1.0, None, w2["keep"], w2["genes"], w2["w"], w2["gkept"])
        ops.csc_colptr(G, n, lcp, lri, w2["keep"], w2["gkept"], w2["out_colptr"])
        ops.csc_scale(G, n, lcp, lri, lx, w2["genes"], w2["gkept"], w2["out_colptr"], w2["out_rowidx"], w2["out_x"])
        ops.sync()
        m = int(w2["out_colptr"][n])
        xs.append(w2["out_x"][:m].cpu().numpy())
        rs.append(w2["out_rowidx"][:m].cpu().numpy())
    assert np.array_equal(np.concatenate(rs), ref["rowidx"])
    assert np.allclose(np.concatenate(xs), ref["x"], rtol=TOL, atol=TOL)


def test_full_size_properties_config2():
    """BASELINE config 2 shape (10 k cells x 20 k genes): size-independent properties + oracle."""
    G, N = 20000, 10000
    cp, ri, x = synth.counts_csc(G, N)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    res = gficf_amd.gficf(M, normalize=False, verbose=False)
    g = res["gficf"]
    # every non-empty cell has unit L2 norm; values in [0, 1]
    nrm = np.sqrt(np.asarray(g.multiply(g).sum(axis=0)).ravel())
    nonempty = np.diff(g.indptr) > 0
    assert np.allclose(nrm[nonempty], 1.0, rtol=1e-12)
    assert g.data.min() >= 0.0 and g.data.max() <= 1.0 + 1e-12
    # idempotence of the per-cell scale: scaling the counts of a cell by a constant changes nothing
    M2 = M.copy()
    M2.data = M2.data * np.repeat(1.0 + (np.arange(N) % 7), np.diff(M2.indptr))
    g2 = gficf_amd.gficf(M2, normalize=False, verbose=False)["gficf"]
    assert np.allclose(g2.data, g.data, rtol=1e-12, atol=1e-15) and np.array_equal(g2.indices, g.indices)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0)
    check_against_oracle(res, ref, N)


def test_global_gather_variant_matches():
    """The scaling pass has two variants (gene tables staged in LDS when they fit, per-entry L2
    gathers otherwise).  Force the second through its test hook in a fresh process."""
    import subprocess
    import sys

    code = (
        "import numpy as np, scipy.sparse as sp, gficf_amd, oracle\n"
        "from gficf_amd import synth\n"
        "for G, N, mn in ((5000, 3000, 0.05), (20000, 1500, 0.0), (800, 600, 0.1)):\n"
        "    cp, ri, x = synth.counts_csc(G, N, seed=G)\n"
        "    res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(G, N)), 1, mn, normalize=False, verbose=False)\n"
        "    ref = oracle.gficf_csc(G, N, cp, ri, x, mn, 1.0)\n"
        "    assert np.array_equal(res['gficf'].indices, ref['rowidx']) and np.array_equal(res['gficf'].indptr, ref['colptr'])\n"
        "    assert np.allclose(res['gficf'].data, ref['x'], rtol=1e-6, atol=1e-6) and np.abs(res['gficf'].data - ref['x']).max() < 1e-12\n"
        "print('ok')\n")
    # GFICF_SCALE_FORCE_SEMI: the LDS variant with the weights read from global memory (what it does by itself when the row
    # ids fit LDS but the kept genes' weights do not, e.g. 23 k genes all kept; the second case below is that situation)
    # GFICF_SCALE_STATIC_CELLS: the round-robin deal of cells to waves (the A/B partner of the entry-balanced ranges)
    for hook in ("GFICF_SCALE_FORCE_GLOBAL", "GFICF_SCALE_FORCE_SEMI", "GFICF_SCALE_STATIC_CELLS"):
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env[hook] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (hook, r.stderr[-2000:])


def test_weights_that_do_not_fit_lds_next_to_the_row_ids():
    """23 k genes, none dropped: the 16-bit row ids fit the LDS, the 23 k weights next to them do not — the LDS variant of the
    scaling pass then reads the weights from global memory (no second kernel).  Long cells included (batched sweeps)."""
    G, N = 23000, 600
    cp, ri, x = synth.counts_csc(G, N, seed=77)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    res = gficf_amd.gficf(M, 1.0, 0.0, normalize=False, verbose=False)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.0, 1.0)
    assert int(ref["keep"].sum()) > 15000          # more kept genes than fit LDS as doubles next to 46 KB of row ids
    check_against_oracle(res, ref, N)
    w_in = 0.25 + synth.rand_unit(9, np.arange(G))
    out2, kept2 = gficf_amd.gficf_with_weights(M, w_in)
    ref2 = oracle.gficf_csc(G, N, cp, ri, x, 0.0, 2.0, w_in=w_in)
    assert np.array_equal(kept2, np.flatnonzero(ref2["keep"]))
    assert np.array_equal(out2.indices, ref2["rowidx"]) and np.allclose(out2.data, ref2["x"], rtol=TOL, atol=TOL)


def test_long_cells_take_the_batched_path():
    # dense-ish matrix: cells with more entries than a wave keeps in registers (28 x 64 = 1792) exercise the three-sweep path of both variants
    G, N = 6000, 300
    cp, ri, x = synth.counts_csc(G, N, median_frac=0.5, sigma=0.3, seed=23)
    assert np.diff(cp).max() > 2000
    res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(G, N)), 1, 0.05, normalize=False, verbose=False)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0)
    check_against_oracle(res, ref, N)


def test_degenerate_shapes():
    r = gficf_amd.gficf(sp.csc_matrix((0, 3)), normalize=False, verbose=False)
    assert r["gficf"].shape == (0, 3) and r["w"].size == 0
    r = gficf_amd.gficf(sp.csc_matrix((5, 0)), normalize=False, verbose=False)
    assert r["gficf"].shape == (0, 0) and r["gficf"].nnz == 0


def test_config4_shape_properties_device_resident():
    """BASELINE config 4 shape (100 k cells x 30 k genes, ~1.5e8 stored entries), device-resident:
    size-independent properties (no host oracle at this size)."""
    import torch

    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path)
    import bench

    ops = gficf_amd.HipOps(0)
    G, N = 30000, 100000
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N, seed=11)
    ws = ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0)
    ops.sync()
    nk = int(ws["out_colptr"][N])
    gk = int(ws["gkept"][0])
    ocp, ori, ox = ws["out_colptr"], ws["out_rowidx"][:nk], ws["out_x"][:nk]
    keep = ws["keep"].bool()
    assert gk == int(keep.sum()) and 0 < gk < G
    # structure: the kept entries of every cell, renumbered, in order
    kept_entry = keep[rowidx.long()]
    assert nk == int(kept_entry.sum())
    remap = torch.cumsum(keep.long(), 0) - 1
    assert torch.equal(ori.long(), remap[rowidx.long()[kept_entry]])
    cnt = torch.zeros(N, dtype=torch.int64, device="cuda")
    col_of = torch.repeat_interleave(torch.arange(N, device="cuda"), colptr[1:] - colptr[:-1])
    cnt.index_add_(0, col_of, kept_entry.long())
    assert torch.equal(ocp[1:] - ocp[:-1], cnt)
    # nt = cells with a non-zero entry per gene; w = log((N+1)/(nt+1))
    nt = torch.bincount(rowidx.long()[x != 0], minlength=G)
    assert torch.equal(ws["nt"], nt)
    w_ref = torch.log((N + 1.0) / (nt.double() + 1.0))
    assert torch.allclose(ws["w"][keep], w_ref[keep], rtol=1e-12, atol=0)
    # values: every non-empty cell has unit L2 norm, values in [0, 1]
    ocol = torch.repeat_interleave(torch.arange(N, device="cuda"), cnt)
    sq = torch.zeros(N, dtype=torch.float64, device="cuda").index_add_(0, ocol, ox * ox)
    nonempty = cnt > 0
    has_w = torch.zeros(N, dtype=torch.float64, device="cuda").index_add_(0, ocol, (ws["w"][keep][ori.long()] > 0).double()) > 0
    assert torch.allclose(sq[nonempty & has_w], torch.ones_like(sq[nonempty & has_w]), rtol=1e-12, atol=0)
    assert float(ox.min()) >= 0.0 and float(ox.max()) <= 1.0 + 1e-12
    # and the closed form on a sample of cells
    for c in (0, 1, N // 2, N - 1):
        sl = slice(int(colptr[c]), int(colptr[c + 1]))
        kp = kept_entry[sl]
        xs, gsel = x[sl][kp], rowidx[sl][kp].long()
        v = (xs / xs.sum()) * ws["w"][gsel]
        ref = v / torch.sqrt((v * v).sum())
        got = ox[int(ocp[c]):int(ocp[c + 1])]
        assert torch.allclose(got, ref, rtol=1e-10, atol=1e-14)


def _device_properties(torch, ops, G, N, colptr, rowidx, x, ws, chunk):
    """Size-independent checks of a device-resident GF-ICF result, cell block by cell block (bounded temporaries):
    structure (kept entries of every cell, renumbered, in order), nt, w, unit L2 norm, value range."""
    nk = int(ws["out_colptr"][N])
    gk = int(ws["gkept"][0])
    ocp, ori, ox = ws["out_colptr"], ws["out_rowidx"], ws["out_x"]
    keep = ws["keep"].bool()
    assert gk == int(keep.sum()) and 0 < gk < G
    remap = torch.cumsum(keep.long(), 0) - 1
    wk = ws["w"]
    nt = torch.zeros(G, dtype=torch.int64, device="cuda")
    kept_total = 0
    xmin, xmax = 1.0, 0.0
    for c0 in range(0, N, chunk):
        c1 = min(N, c0 + chunk)
        p0, p1 = int(colptr[c0]), int(colptr[c1])
        q0, q1 = int(ocp[c0]), int(ocp[c1])
        ri, xv = rowidx[p0:p1].long(), x[p0:p1]
        nt += torch.bincount(ri[xv != 0], minlength=G)
        kept_entry = keep[ri]
        assert q1 - q0 == int(kept_entry.sum())
        kept_total += q1 - q0
        assert torch.equal(ori[q0:q1].long(), remap[ri[kept_entry]])
        lens = colptr[c0 + 1:c1 + 1] - colptr[c0:c1]
        col_of = torch.repeat_interleave(torch.arange(c1 - c0, device="cuda"), lens)
        cnt = torch.zeros(c1 - c0, dtype=torch.int64, device="cuda").index_add_(0, col_of, kept_entry.long())
        assert torch.equal(ocp[c0 + 1:c1 + 1] - ocp[c0:c1], cnt)
        ocol = torch.repeat_interleave(torch.arange(c1 - c0, device="cuda"), cnt)
        oxs = ox[q0:q1]
        sq = torch.zeros(c1 - c0, dtype=torch.float64, device="cuda").index_add_(0, ocol, oxs * oxs)
        has_w = torch.zeros(c1 - c0, dtype=torch.float64, device="cuda").index_add_(0, ocol, (wk[keep][ori[q0:q1].long()] > 0).double()) > 0
        sel = (cnt > 0) & has_w
        assert torch.allclose(sq[sel], torch.ones_like(sq[sel]), rtol=1e-12, atol=0)
        if q1 > q0:
            xmin, xmax = min(xmin, float(oxs.min())), max(xmax, float(oxs.max()))
        del ri, xv, kept_entry, col_of, cnt, ocol, sq, has_w, sel
    assert kept_total == nk
    assert torch.equal(ws["nt"], nt)
    w_ref = torch.log((N + 1.0) / (nt.double() + 1.0))
    assert torch.allclose(wk[keep], w_ref[keep], rtol=1e-9, atol=1e-14)      # (w ~ 1e-6 for genes present in nearly every cell)
    lo, hi = float(N) * 0.05, float(N) * 1.0
    assert torch.equal(keep, (nt.double() > lo) & (nt.double() <= hi))
    assert xmin >= 0.0 and xmax <= 1.0 + 1e-12
    # and the closed form of R/gficf.R:59,79,100-103 on a sample of cells
    for c in (0, 1, N // 3, N // 2, N - 2, N - 1):
        sl = slice(int(colptr[c]), int(colptr[c + 1]))
        kp = keep[rowidx[sl].long()]
        xs, gsel = x[sl][kp], rowidx[sl][kp].long()
        v = (xs / xs.sum()) * wk[gsel]
        ref = v / torch.sqrt((v * v).sum())
        got = ox[int(ocp[c]):int(ocp[c + 1])]
        assert torch.allclose(got, ref, rtol=1e-10, atol=1e-14)
    return nk, gk


def _import_bench():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    return bench


def test_config3_shape_54k_cells_23k_genes_vs_oracle():
    """BASELINE config 3 (Tabula-Muris-sized stand-in: 23 k genes x 54 k cells, ~6e7 stored entries): the device
    pipeline and the host C ABI against the oracle — keep mask, nt, structure exact; values and weights <= 1e-6."""
    import torch

    bench = _import_bench()
    ops = gficf_amd.HipOps(0)
    G, N = 23000, 54000
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    ws = ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0)
    ops.sync()
    cp, ri, xv = colptr.cpu().numpy(), rowidx.cpu().numpy(), x.cpu().numpy()
    ref = oracle.gficf_csc(G, N, cp, ri, xv, 0.05, 1.0)
    nk = int(ws["out_colptr"][N])
    assert nk == len(ref["x"]) and int(ws["gkept"][0]) == int(ref["keep"].sum())
    assert np.array_equal(ws["keep"].cpu().numpy().astype(bool), ref["keep"])
    assert np.array_equal(ws["nt"].cpu().numpy()[ref["keep"]], ref["nt"][ref["keep"]])      # (the oracle reports 0 for dropped genes)
    assert np.array_equal(ws["out_colptr"].cpu().numpy(), ref["colptr"])
    assert np.array_equal(ws["out_rowidx"][:nk].cpu().numpy(), ref["rowidx"])
    got = ws["out_x"][:nk].cpu().numpy()
    assert np.allclose(got, ref["x"], rtol=TOL, atol=TOL) and np.abs(got - ref["x"]).max() < 1e-12
    assert np.allclose(ws["w"].cpu().numpy()[ref["keep"]], ref["w"][ref["keep"]], rtol=TOL, atol=TOL)
    # the reference-shaped host entry (gficf(), through gficf_normalize_csc_host_plan/_finish) on the same matrix
    M = sp.csc_matrix((xv, ri, cp), shape=(G, N))
    res = gficf_amd.gficf(M, normalize=False, verbose=False, storeRaw=False)
    check_against_oracle(res, ref, N)


def test_config5_shape_1M_cells_30k_genes_properties_device_resident():
    """BASELINE config 5, GF-ICF half (1 M cells x 30 k genes, at most 2 147 stored entries per cell so that
    nnz < 2^31 — SURVEY.md 8d; ~1.8e9 entries, ~22 GB of input), device-generated and device-resident:
    the size-independent properties and the sampled closed form (no host oracle at this size)."""
    import torch

    bench = _import_bench()
    ops = gficf_amd.HipOps(0)
    G, N = 30000, 1_000_000
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N, seed=5, max_per_cell=2147, chunk_cells=125_000)
    nnz = int(rowidx.numel())
    assert 1.0e9 < nnz < 2 ** 31 and int(colptr[N]) == nnz
    assert int((colptr[1:] - colptr[:-1]).max()) <= 2147
    ws = ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0)
    ops.sync()
    nk, gk = _device_properties(torch, ops, G, N, colptr, rowidx, x, ws, chunk=125_000)
    assert 0 < nk <= nnz
    # idempotence of the per-cell scale (x / colSums(x) is invariant to scaling a cell's counts): same values
    first = ws["out_x"][:nk].clone()
    x *= torch.repeat_interleave(1.0 + (torch.arange(N, device="cuda") % 7).double(), colptr[1:] - colptr[:-1])
    ws = ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, None, ws)
    ops.sync()
    assert int(ws["out_colptr"][N]) == nk
    assert torch.allclose(ws["out_x"][:nk], first, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("n_zero", [0, 1, 500])
def test_device_path_with_explicit_zeros_is_reported_and_the_exact_entry_handles_them(n_zero):
    """gficf_csc_device counts stored entries (no x read); the scaling pass notices an explicitly stored zero and the next
    sync reports GFICF_ERR_EXPLICIT_ZEROS; gficf_csc_exact_device then gives R's rowSums(M != 0) rule."""
    import torch

    ops = gficf_amd.HipOps(0)
    G, N = 4000, 2500
    cp, ri, x = synth.counts_csc(G, N, seed=31)
    x = x.copy()
    if n_zero:
        pos = (synth.rand_u64(5, np.arange(n_zero)) % np.uint64(len(x))).astype(np.int64)
        x[pos] = 0.0
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0)
    d = lambda a: torch.from_numpy(a).cuda()
    for rep in range(2):                        # twice: the deferred status is cleared by the sync that reports it
        ws = ops.gficf_csc(G, N, d(cp), d(ri), d(x), 0.05, 1.0)
        if n_zero:
            with pytest.raises(gficf_amd.GficfError) as ei:
                ops.sync()
            assert ei.value.status == "GFICF_ERR_EXPLICIT_ZEROS"
            ws = ops.gficf_csc(G, N, d(cp), d(ri), d(x), 0.05, 1.0, None, ws, exact=True)
        ops.sync()
        nk = int(ws["out_colptr"][N])
        assert np.array_equal(ws["keep"].cpu().numpy().astype(bool), ref["keep"])
        assert np.array_equal(ws["nt"].cpu().numpy()[ref["keep"]], ref["nt"][ref["keep"]])
        assert nk == len(ref["x"]) and np.array_equal(ws["out_rowidx"][:nk].cpu().numpy(), ref["rowidx"])
        assert np.allclose(ws["out_x"][:nk].cpu().numpy(), ref["x"], rtol=TOL, atol=TOL)
    # the self-healing form: one call, synchronised, the exact pass taken only when the status asks for it
    ws = ops.gficf_csc(G, N, d(cp), d(ri), d(x), 0.05, 1.0, auto_exact=True)
    nk = int(ws["out_colptr"][N])
    assert np.array_equal(ws["nt"].cpu().numpy()[ref["keep"]], ref["nt"][ref["keep"]])
    assert nk == len(ref["x"]) and np.array_equal(ws["out_rowidx"][:nk].cpu().numpy(), ref["rowidx"])
    assert np.allclose(ws["out_x"][:nk].cpu().numpy(), ref["x"], rtol=TOL, atol=TOL)
    ops.sync()                                  # nothing left pending
    if n_zero:
        # another deferred error of the same sync is neither masked nor replaced by the explicit-zero bit: a row index
        # outside [0, G) next to the stored zeros surfaces as BAD_CSC, from the plain and from the self-healing call
        bad = ri.copy()
        bad[len(bad) // 2] = G + 5
        ops.gficf_csc(G, N, d(cp), d(bad), d(x), 0.05, 1.0)
        with pytest.raises(gficf_amd.GficfError) as ei:
            ops.sync()
        assert ei.value.status == "GFICF_ERR_BAD_CSC"
        with pytest.raises(gficf_amd.GficfError) as ei:
            ops.gficf_csc(G, N, d(cp), d(bad), d(x), 0.05, 1.0, auto_exact=True)
        assert ei.value.status == "GFICF_ERR_BAD_CSC"
        ops.sync()                              # and the status word is clean again
    # the host entry (what gficf() binds) always counts exactly
    res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(G, N)), normalize=False, verbose=False)
    assert np.array_equal(res["gficf"].indices, ref["rowidx"]) and np.allclose(res["gficf"].data, ref["x"], rtol=TOL, atol=TOL)


def test_cluster_signatures_next_row_n3():
    """N3: data$cluster.gene.rnk (R/clustCells.R:121-123) on the GF-ICF matrix; labels in first-appearance order."""
    from oracle import oracle_np

    G, N = 3000, 2000
    cp, ri, x = synth.counts_csc(G, N, seed=41)
    res = gficf_amd.gficf(sp.csc_matrix((x, ri, cp), shape=(G, N)), normalize=False, verbose=False)
    lab = np.array([f"c{v}" for v in (synth.rand_u64(3, np.arange(N)) % np.uint64(7)).astype(int)])
    got, labels = gficf_amd.cluster_signatures(res["gficf"], lab)
    want, wl = oracle_np.cluster_signatures_np(res["gficf"], lab)
    assert list(labels) == list(wl) and got.shape == want.shape == (res["gficf"].shape[0], 7)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6) and np.abs(got - want).max() < 1e-10
    # an empty matrix / a single cluster
    got1, l1 = gficf_amd.cluster_signatures(res["gficf"], np.zeros(N, dtype=int))
    assert got1.shape[1] == 1 and np.allclose(got1[:, 0], np.asarray(res["gficf"].sum(axis=1)).ravel(), rtol=1e-10)


@pytest.mark.parametrize("G,N,C", [(800, 500, 5), (20000, 3000, 11), (2500, 30000, 6000), (3000, 20000, 40)])
def test_cluster_signatures_both_forms(G, N, C):
    """The plain form (few cells, or more genes than LDS holds) and the grouped form (cells sorted by cluster, sums kept in LDS;
    more clusters than the grouping bins at C = 6000) against scipy; labels that are not 0..C-1 strings."""
    cp, ri, x = synth.counts_csc(G, N, seed=G + C, median_frac=0.02)
    M = sp.csc_matrix((x * 0.25, ri, cp), shape=(G, N))
    lab = np.array([f"t{v}" for v in (synth.rand_u64(5, np.arange(N)) % np.uint64(C)).astype(int)])
    got, labels = gficf_amd.cluster_signatures(M, lab)
    assert got.shape == (G, len(labels)) and len(labels) == len(set(lab))
    onehot = sp.csr_matrix((np.ones(N), (np.arange(N), np.unique(lab, return_inverse=True)[1])), shape=(N, len(labels)))
    want = np.asarray((M @ onehot).todense())
    uniq = np.unique(lab)
    col_of = {u: j for j, u in enumerate(uniq)}
    want = want[:, [col_of[u] for u in labels]]
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("G,N,seed", [(3000, 2000, 41), (500, 7, 3), (40000, 300, 5), (1200, 5000, 9)])
def test_transpose_next_row_n3(G, N, seed):
    """N3: data$pca$cells = t(data$gficf) (R/dimensinalityReduction.R:33,100): exact structure and values, cells ascending
    within every gene; G = 40 000 takes two gene ranges of the LDS counters."""
    from oracle import oracle_np

    cp, ri, x = synth.counts_csc(G, N, seed=seed)
    x = x.copy()
    x[::17] = 0.0                                                     # explicit zeros are entries like any other
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    T = gficf_amd.transpose_gficf(M)
    ptr, idx, val = oracle_np.transpose_np(G, N, cp, ri, x)
    assert T.shape == (N, G)
    assert np.array_equal(T.indptr, ptr) and np.array_equal(T.indices, idx) and np.array_equal(T.data, val)
    S = M.T.tocsc()                                                   # scipy's own transpose as a second opinion
    S.sort_indices()
    assert np.array_equal(S.indptr, ptr) and np.array_equal(S.indices, idx) and np.array_equal(S.data, val)


def test_transpose_device_resident_and_edge_cases():
    import torch

    ops = gficf_amd.HipOps(0)
    dev = "cuda:0"
    from oracle import oracle_np

    G, N = 2500, 1500
    cp, ri, x = synth.counts_csc(G, N, seed=77)
    res = ops.gficf_csc(G, N, torch.from_numpy(cp.astype(np.int64)).to(dev), torch.from_numpy(ri).to(dev), torch.from_numpy(x).to(dev))
    ops.sync()
    gk, nk = int(res["gkept"][0]), int(res["out_colptr"][N])
    ws = torch.zeros(ops.csc_transpose_workspace_bytes(gk, N), dtype=torch.uint8, device=dev)
    ptr = torch.zeros(gk + 1, dtype=torch.int64, device=dev)
    idx = torch.zeros(nk, dtype=torch.int32, device=dev)
    val = torch.zeros(nk, dtype=torch.float64, device=dev)
    for _ in range(2):                                                # the scratch is reusable as it is
        ops.csc_transpose(gk, N, res["out_colptr"], res["out_rowidx"][:nk], res["out_x"][:nk], ptr, idx, val, ws)
        ops.sync()
        wp, wi, wv = oracle_np.transpose_np(gk, N, res["out_colptr"].cpu().numpy(), res["out_rowidx"][:nk].cpu().numpy(),
                                            res["out_x"][:nk].cpu().numpy())
        assert np.array_equal(ptr.cpu().numpy(), wp) and np.array_equal(idx.cpu().numpy(), wi)
        assert np.array_equal(val.cpu().numpy(), wv)
    # empty matrices
    E = gficf_amd.transpose_gficf(sp.csc_matrix((5, 4)))
    assert E.shape == (4, 5) and E.nnz == 0
    E = gficf_amd.transpose_gficf(sp.csc_matrix((0, 4)))
    assert E.shape == (4, 0)
    # a row index outside [0, G) is reported, not followed
    bad = ri.copy(); bad[5] = G + 3
    with pytest.raises(gficf_amd.GficfError):
        out_p, out_i, out_x = np.zeros(G + 1, np.int64), np.zeros(len(ri), np.int32), np.zeros(len(ri))
        from gficf_amd import _lib
        from gficf_amd.api import _np_ptr, check, default_context
        check(_lib.load().gficf_csc_transpose_host(default_context().handle, G, N, _np_ptr(cp.astype(np.int64)), 1, _np_ptr(bad),
                                                   _np_ptr(x), _np_ptr(out_p), _np_ptr(out_i), _np_ptr(out_x)))


@pytest.mark.parametrize("icf_type", ["classic", "prob", "smooth"])
@pytest.mark.parametrize("norm", ["l2", "l1"])
def test_helper_branches_icf_type_and_norm(icf_type, norm):
    """getIdfW(type = prob / smooth) and l.norm(norm = "l1") (reference R/gficf.R:89-91,100; gficf() itself never takes
    them): structure exact, values and weights within 1e-6 of the oracle; the context falls back to the defaults."""
    G, N = 1800, 900
    cp, ri, x = synth.counts_csc(G, N, seed=31)
    M = sp.csc_matrix((x, ri, cp), shape=(G, N))
    res = gficf_amd.gficf(M, normalize=False, verbose=False, icf_type=icf_type, norm=norm)
    ref = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0, None, icf_type, norm)
    assert np.array_equal(res["genes"], np.flatnonzero(ref["keep"]))
    assert np.array_equal(res["gficf"].indices, ref["rowidx"]) and np.array_equal(res["gficf"].indptr, ref["colptr"])
    assert np.allclose(res["gficf"].data, ref["x"], rtol=1e-6, atol=1e-6)
    assert np.allclose(res["w"], ref["w"][ref["keep"]], rtol=1e-6, atol=1e-6)
    # options do not leak into the next call
    d = gficf_amd.gficf(M, normalize=False, verbose=False)
    ref0 = oracle.gficf_csc(G, N, cp, ri, x, 0.05, 1.0)
    assert np.allclose(d["gficf"].data, ref0["x"], rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        gficf_amd.gficf(M, normalize=False, verbose=False, icf_type="bm25")


def test_config4_shape_as_eight_cell_blocks_matches_the_single_shot_pass():
    """BASELINE config 4's GF-ICF half is an 8-GPU config (100 k cells x 30 k genes): eight cell blocks cut by stored entries,
    each counted on its own, the per-gene counts summed (the all-reduce of the sharded path), gene table / colptr / scale per
    block — the seam the multi-GPU forms run through — against the single-shot device pass on the same matrix: identical
    structure, values equal to 1e-12."""
    import torch

    from gficf_amd.dist import shard_bounds_by_nnz

    bench = _import_bench()
    ops = gficf_amd.HipOps(0)
    G, N, P = 30000, 100000, 8
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N, seed=11)
    ws = ops.gficf_csc(G, N, colptr, rowidx, x, 0.05, 1.0, exact=True)
    ops.sync()
    nk = int(ws["out_colptr"][N])
    blocks = shard_bounds_by_nnz(colptr.cpu().numpy(), P)
    nt = torch.zeros(G, dtype=torch.int64, device="cuda")
    parts = []
    for b, e in blocks:
        p0, p1 = int(colptr[b]), int(colptr[e])
        lcp = (colptr[b:e + 1] - colptr[b]).contiguous()
        parts.append((e - b, lcp, rowidx[p0:p1], x[p0:p1], int(ws["out_colptr"][b]), int(ws["out_colptr"][e])))
        ops.csc_count(G, e - b, lcp, rowidx[p0:p1], x[p0:p1], nt)
    ops.sync()
    assert torch.equal(nt, ws["nt"])
    share = [p[2].numel() for p in parts]
    assert max(share) - min(share) < 3 * int((colptr[1:] - colptr[:-1]).max())           # blocks balanced by entries
    for n, lcp, lri, lx, q0, q1 in parts:
        w2 = ops.csc_workspace(G, n, int(lri.numel()))
        ops.csc_genes(G, N, nt, 0.05, 1.0, None, w2["keep"], w2["genes"], w2["w"], w2["gkept"])
        ops.csc_colptr(G, n, lcp, lri, w2["keep"], w2["gkept"], w2["out_colptr"])
        ops.csc_scale(G, n, lcp, lri, lx, w2["genes"], w2["gkept"], w2["out_colptr"], w2["out_rowidx"], w2["out_x"])
        ops.sync()
        m = int(w2["out_colptr"][n])
        assert m == q1 - q0 and torch.equal(w2["keep"], ws["keep"])
        assert torch.equal(w2["out_rowidx"][:m], ws["out_rowidx"][q0:q1])
        assert torch.allclose(w2["out_x"][:m], ws["out_x"][q0:q1], rtol=1e-12, atol=1e-15)
        del w2
    assert sum(p[5] - p[4] for p in parts) == nk


@pytest.mark.parametrize("G,N,mn,mx,seed,env", [(1500, 900, 0.05, 1.0, 1, {}), (5000, 3000, 0.05, 1.0, 2, {}), (23000, 6000, 0.0, 2.0, 3, {}),
                                               (800, 400, 0.3, 0.9, 4, {}), (3000, 2000, 0.05, 1.0, 5, {"GFICF_SCALE_FORCE_GLOBAL": "1"}),
                                               (70000, 300, 0.0, 2.0, 6, {}), (64, 1, 0.0, 2.0, 7, {}), (2000, 1200, 0.99, 1.0, 8, {})])
def test_pointer_begin_end_form_equals_the_canonical_pass(G, N, mn, mx, seed, env):
    """The device-resident chain's form of the result (gficf_csc_be_device): every cell compacts inside its own input range —
    cell c = out[colptr[c] : out_end[c]] — so no global positions, no kept-count pass and no scan are needed (three launches
    instead of five).  Entry for entry the canonical compacted CSC (same kernels, same arithmetic: the values are bit-equal),
    which in turn is checked against the oracle; t() and the cluster sums of the form equal those of the canonical matrix."""
    import subprocess

    if env:                      # (the variant switches are read once per process)
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
                f"import test_gficf_gpu as t; t.test_pointer_begin_end_form_equals_the_canonical_pass({G}, {N}, {mn}, {mx}, {seed}, {{}})\n")
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return
    import torch

    ops = gficf_amd.HipOps(0)
    colptr, rowidx, x = synth.counts_csc(G, N, seed=seed)
    if N > 2:                    # an empty cell, and one whose entries are all dropped or all kept
        colptr = colptr.copy()
        cut = colptr[2] - colptr[1]
        rowidx, x = np.delete(rowidx, slice(colptr[1], colptr[2])), np.delete(x, slice(colptr[1], colptr[2]))
        colptr[2:] -= cut
    cp, ri, xv = torch.from_numpy(colptr.astype(np.int64)).cuda(), torch.from_numpy(rowidx.astype(np.int32)).cuda(), torch.from_numpy(x).cuda()
    nnz = int(ri.numel())
    can = ops.gficf_csc(G, N, cp, ri, xv, mn, mx, None, None, exact=True)
    ops.sync()
    ref = oracle.gficf_csc(G, N, colptr.astype(np.int64), rowidx, x, mn, mx)
    kn = int(can["out_colptr"][N])
    assert kn == len(ref["x"]) and np.array_equal(can["out_rowidx"][:kn].cpu().numpy(), ref["rowidx"])
    assert np.allclose(can["out_x"][:kn].cpu().numpy(), ref["x"], rtol=1e-6, atol=1e-6)
    be = ops.gficf_csc_be(G, N, cp, ri, xv, mn, mx, None, None, exact=True)
    ops.sync()
    end = be["out_end"][:N]
    lens = end - cp[:N]
    assert torch.equal(lens, can["out_colptr"][1:N + 1] - can["out_colptr"][:N])            # kept entries per cell
    assert bool((end <= cp[1:N + 1]).all()) and bool((lens >= 0).all())
    # gather the form into a compact matrix: entry for entry the canonical one, bit-equal values
    if kn:
        cell = torch.repeat_interleave(torch.arange(N, device="cuda"), lens)
        pos = torch.arange(kn, device="cuda") - can["out_colptr"][:N][cell] + cp[:N][cell]
        assert torch.equal(be["out_rowidx"][pos], can["out_rowidx"][:kn]) and torch.equal(be["out_x"][pos], can["out_x"][:kn])
    assert torch.equal(be["keep"], can["keep"]) and torch.equal(be["nt"], can["nt"]) and torch.equal(be["w"], can["w"])
    gk = int(can["gkept"][0])
    if gk == 0 or kn == 0:
        return
    # t(): the result is an ordinary compact CSC, equal to the transpose of the canonical matrix
    tws = torch.zeros(ops.csc_transpose_workspace_bytes(gk, N), dtype=torch.uint8, device="cuda")
    outs = []
    for form in ("canonical", "be"):
        t_ptr = torch.zeros(gk + 1, dtype=torch.int64, device="cuda")
        t_idx, t_val = torch.full((nnz,), -7, dtype=torch.int32, device="cuda"), torch.zeros(nnz, dtype=torch.float64, device="cuda")
        if form == "canonical":
            ops.csc_transpose(gk, N, can["out_colptr"], can["out_rowidx"][:kn], can["out_x"][:kn], t_ptr, t_idx, t_val, tws)
        else:
            ops.csc_transpose_be(gk, N, cp, be["out_end"], be["out_rowidx"], be["out_x"], t_ptr, t_idx, t_val, tws)
        ops.sync()
        outs.append((t_ptr, t_idx[:kn], t_val[:kn]))
    assert int(outs[1][0][gk]) == kn and all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    # cluster sums
    C = 7
    cl = (torch.arange(N, device="cuda", dtype=torch.int64) * 2654435761 % C).to(torch.int32)
    s_can, s_be = torch.zeros((C, gk), dtype=torch.float64, device="cuda"), torch.zeros((C, gk), dtype=torch.float64, device="cuda")
    ops.cluster_signatures(gk, N, can["out_colptr"], can["out_rowidx"][:kn], can["out_x"][:kn], cl, C, s_can)
    ops.cluster_signatures_be(gk, N, cp, be["out_end"], be["out_rowidx"], be["out_x"], cl, C, s_be)
    ops.sync()
    assert torch.allclose(s_can, s_be, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("G,N,s,n_rare", [(400, 2000, 31, 300), (2000, 60000, 300, 5000), (5000, 20000, 2500, 100)])
def test_gficf_against_the_closed_form_of_a_circulant_matrix(G, N, s, n_rare):
    """GF-ICF with an answer derived by algebra, no oracle in the loop (tests/helpers/closed_form.py::circulant_counts): every cell holds s
    genes of a ring with counts 1 .. s plus, in some cells, a rare gene the 5 % filter drops; S_c and the common ICF weight cancel, so
    gficf[(c + t) mod G, c] = (t + 1) / sqrt(s (s + 1) (2 s + 1) / 6).  Through gficf() (host entry: plan + finish) — keep mask, nt, the
    compacted structure exact, values and w to 1e-12 — at up to 18 M stored entries and cells of 2 500 entries (the two-sweep path)."""
    from tests.helpers.closed_form import circulant_counts

    M, want, keep, nt, w = circulant_counts(G, N, s, n_rare)
    res = gficf_amd.gficf(M, normalize=False, verbose=False)
    assert np.array_equal(res["genes"], np.flatnonzero(keep)) and np.array_equal(res["nt"], nt[keep])
    assert np.allclose(res["w"], w, rtol=1e-13, atol=0)
    got = res["gficf"]
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.allclose(got.data, want.data, rtol=1e-12, atol=0)
    # every cell has unit L2 norm (R/gficf.R:100-103) — a property, checked on the result itself
    sq = np.add.reduceat(got.data ** 2, got.indptr[:-1])
    assert np.allclose(sq, 1.0, rtol=1e-12)
