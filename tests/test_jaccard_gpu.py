"""GPU parity: the HIP Jaccard path (through the C ABI) against the CPU oracle.
Bar: intersection counts, edge list and zero rows bit-exact; weights bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import gficf_amd
import oracle
from gficf_amd import _lib, synth

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ops():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return gficf_amd.HipOps(0)


def device_jaccard(ops, mat, with_u=True):
    """Device-resident pipeline: column-major idx on the GPU -> (E x 3) matrix + counts."""
    import torch

    N, k = mat.shape
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()        # (k, N) == column-major N x k
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    rmat = torch.full((3, N * k), -7.0, dtype=torch.float64, device="cuda")   # poison: must be fully overwritten
    u = torch.full((N * k,), -7, dtype=torch.int32, device="cuda") if with_u else None
    ops.jaccard(idx, N, k, table, rmat, u)
    ops.sync()
    return rmat.cpu().numpy().T, (u.cpu().numpy() if with_u else None)


def test_known_answers(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        known = json.load(f)
    for case in known["jaccard"]:
        mat = np.array(case["mat"], dtype=np.int32)
        k = mat.shape[1]
        want_u = np.array(case["u"], dtype=np.int32).reshape(-1)
        got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
        want, _ = oracle.jaccard(mat)
        assert np.array_equal(got, want), case["name"]
        pos = want_u > 0
        assert np.array_equal(got[:, 2], np.where(pos, want_u / (2.0 * k - want_u), 0.0)), case["name"]


@pytest.mark.parametrize("N,k", [(64, 5), (1000, 15), (1000, 16), (1000, 17), (1000, 30), (999, 32), (700, 33),
                                 (1200, 50), (500, 64), (400, 65), (300, 100), (300, 128), (300, 129), (600, 200),
                                 (520, 256), (3000, 15), (10000, 30)])
def test_host_api_matches_oracle(N, k):
    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k, perm_seed=7)
    got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    want, _ = oracle.jaccard(mat, nthreads=8)
    assert got.shape == (N * k, 3) and got.dtype == np.float64
    assert np.array_equal(got, want)


def test_double_input_matches_int_input():
    mat = synth.knn_windowed(2000, 30, seed=5)
    a = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    b = gficf_amd.rcpp_parallel_jaccard_coef(mat.astype(np.float64), False)     # what Rcpp coerces to
    c = gficf_amd.rcpp_parallel_jaccard_coef(np.asfortranarray(mat), False)
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_golden_fixtures(golden_dir, ops):
    z = np.load(os.path.join(golden_dir, "jaccard_cases.npz"))
    for nm in z["names"]:
        N, k, seed = (int(v) for v in z[nm + "_meta"])
        mat = synth.knn_windowed(N, k, seed=seed, perm_seed=seed + 100)
        rm, u = device_jaccard(ops, mat)
        assert np.array_equal(u.astype(np.uint8), z[nm + "_u"]), nm
        assert sha(np.asfortranarray(rm)) == bytes(z[nm + "_sha"]).decode(), nm
    for nm in ("uniform", "dupheavy"):
        rm, u = device_jaccard(ops, z[nm + "_mat"])
        assert np.array_equal(u.astype(np.uint8), z[nm + "_u"]), nm
        assert sha(np.asfortranarray(rm)) == bytes(z[nm + "_sha"]).decode(), nm


def test_uniform_random_mostly_empty_intersections(ops):
    mat = synth.knn_uniform(20000, 30)
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=8)
    assert np.array_equal(u, wu) and np.array_equal(rm, want)
    assert (u == 0).mean() > 0.9
    assert not rm[u == 0].any()           # zero rows stay zero


@pytest.mark.parametrize("k,mod", [(12, 40), (30, 50), (40, 45), (100, 130)])
def test_duplicate_and_self_ids_multiset_semantics(ops, k, mod):
    # ids drawn from a small range: rows hold duplicates and their own id (std::set_intersection multiset counts)
    N = 600
    r = synth.rand_u64(k, np.arange(N * k)).reshape(N, k)
    mat = (r % np.uint64(mod)).astype(np.int32) + 1
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=8)
    assert np.array_equal(u, wu)
    assert np.array_equal(rm, want)


def test_mixed_duplicate_rows_only_some_cells(ops):
    # mostly clean rows, a few rows with duplicates: exercises the per-row duplicate flag on
    # both the source row and a gathered neighbour row
    mat = synth.knn_windowed(5000, 30, seed=21)
    mat[17, 3] = mat[17, 4]
    mat[4000, 0] = mat[4000, 29]
    mat[123, :] = mat[123, 0]
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=8)
    assert np.array_equal(u, wu) and np.array_equal(rm, want)


def test_hash_collision_heavy_rows(ops):
    # ids congruent modulo a large power of two: stresses the LDS hash set's overflow path
    N, k = 1 << 16, 30
    base = synth.knn_windowed(64, k, W=31, seed=2)                 # ids 1..64
    mat = ((base.astype(np.int64) - 1) * 1024 + 1).astype(np.int32)   # ids 1, 1025, 2049, ...
    full = synth.knn_windowed(N, k, seed=3)
    rows = (np.arange(64) * 1024)
    full[rows] = mat
    rm, u = device_jaccard(ops, full)
    want, wu = oracle.jaccard(full, nthreads=8)
    assert np.array_equal(u, wu) and np.array_equal(rm, want)


def test_out_of_range_ids_are_rejected():
    mat = synth.knn_windowed(1000, 15, seed=1)
    for bad in (0, 1001, -3):
        m = mat.copy()
        m[500, 7] = bad
        with pytest.raises(gficf_amd.GficfError) as ei:
            gficf_amd.rcpp_parallel_jaccard_coef(m, False)
        assert ei.value.status == "GFICF_ERR_BAD_ID"
    m = mat.astype(np.float64)
    m[3, 3] = 2.5
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.rcpp_parallel_jaccard_coef(m, False)
    m[3, 3] = np.nan
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.rcpp_parallel_jaccard_coef(m, False)
    # the context stays usable afterwards
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), oracle.jaccard(mat)[0])


def test_unsupported_k_and_empty_inputs():
    # k beyond 256 is supported since round 5 (the sorted-row path); what stays unsupported is k beyond the uint16 counts
    got = gficf_amd.rcpp_parallel_jaccard_coef(np.ones((300, 257), dtype=np.int32), False)     # every row names cell 1, 257 times
    want, _ = oracle.jaccard(np.ones((300, 257), dtype=np.int32), nthreads=4)
    assert np.array_equal(got, want) and np.all(got[:, 2] == 1.0)
    assert _lib.load().gficf_jaccard_row_words(100, 65536) == -1
    assert gficf_amd.rcpp_parallel_jaccard_coef(np.zeros((0, 5), dtype=np.int32), False).shape == (0, 3)
    assert gficf_amd.rcpp_parallel_jaccard_coef(np.zeros((7, 0), dtype=np.int32), False).shape == (0, 3)


def test_print_output_banners(capfd):
    mat = synth.knn_windowed(200, 5, seed=1)
    gficf_amd.rcpp_parallel_jaccard_coef(mat, True)
    out = capfd.readouterr().out
    assert "Running Parallell Jaccard Coefficient Estimation..." in out and "Done!!" in out


def test_clustcells_call_site_filter():
    # reference R/clustCells.R:63-68: drop the self column, keep weight > 0 rows in order
    mat = synth.knn_uniform(3000, 15)
    neigh = np.concatenate([np.arange(1, 3001, dtype=np.int32)[:, None], mat], axis=1)
    rel = gficf_amd.jaccard_edges(neigh)
    want, _ = oracle.jaccard(mat, nthreads=4)
    want = want[want[:, 2] > 0]
    assert np.array_equal(rel["from"], want[:, 0]) and np.array_equal(rel["to"], want[:, 1])
    assert np.array_equal(rel["weight"], want[:, 2])


def test_cell_block_seam_matches_full(ops):
    """The multi-GPU seam on one GPU: two ingests into table blocks + per-block edges == one shot."""
    import torch

    N, k = 5001, 30
    mat = synth.knn_windowed(N, k, seed=8)
    full, fu = device_jaccard(ops, mat)
    kp = ops.row_words(N, k)
    table = torch.zeros((N, kp), dtype=torch.int32, device="cuda")
    cut = 2600
    outs = []
    for b, e in ((0, cut), (cut, N)):
        blk = torch.from_numpy(np.ascontiguousarray(mat[b:e].T)).cuda()
        ops.jaccard_ingest(blk, e - b, k, N, table[b:e])
    for b, e in ((0, cut), (cut, N)):
        out = torch.empty((3, (e - b) * k), dtype=torch.float64, device="cuda")
        ops.jaccard_edges(table, N, k, b, e, out)
        outs.append(out)
    ops.sync()
    got = torch.cat(outs, dim=1).cpu().numpy().T
    assert np.array_equal(got, full)


@pytest.mark.parametrize("k", [1, 3, 15, 17, 29, 31, 32])
def test_quad_stores_tails_and_seams_all_output_forms(ops, k):
    """The k <= 32 kernel writes its edges four consecutive cells at a time; the last quad of a wave may be short and, for odd
    k, end in the middle of a lane's pair of edges.  Every output form (matrix, matrix + uint32 counts, uint16 counts), cell
    counts with every remainder mod 4, and cell blocks that start at odd cells, against the oracle."""
    import torch

    for N in (k + 2 if k + 2 > 4 else 5, 101, 102, 103, 1001, 4099):
        if N <= k + 1:
            continue
        mat = synth.knn_uniform(N, k, seed=N + k) if N < 2 * k + 3 else synth.knn_windowed(N, k, W=max(100, k), seed=N + k)
        want, want_u = oracle.jaccard(mat, nthreads=4)
        got, got_u = device_jaccard(ops, mat, with_u=True)            # matrix + uint32 counts
        assert np.array_equal(got, want), (N, k)
        assert np.array_equal(got_u, want_u.reshape(-1)), (N, k)
        got2, _ = device_jaccard(ops, mat, with_u=False)              # matrix only
        assert np.array_equal(got2, want), (N, k)
        cnt = gficf_amd.jaccard_counts(mat)                           # uint16 counts through the host entry
        assert np.array_equal(np.asarray(cnt).reshape(-1).astype(np.int64), want_u.reshape(-1).astype(np.int64)), (N, k)
        # blocks that begin at odd cells
        table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
        ops.jaccard_ingest(torch.from_numpy(np.ascontiguousarray(mat.T)).cuda(), N, k, N, table)
        cuts = [0, 1, min(7, N), min(50, N), N]
        outs = []
        for b, e in zip(cuts[:-1], cuts[1:]):
            if e <= b:
                continue
            out = torch.full((3, (e - b) * k), -7.0, dtype=torch.float64, device="cuda")
            ops.jaccard_edges(table, N, k, b, e, out)
            outs.append(out)
        ops.sync()
        assert np.array_equal(torch.cat(outs, dim=1).cpu().numpy().T, want), (N, k, "blocks")


def test_full_size_north_star_point_bit_exact(ops):
    """100 k cells x k = 30 (the north-star point): whole edge matrix bit-exact vs the oracle."""
    mat = synth.knn_windowed(100000, 30)
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=os.cpu_count() or 8)
    assert np.array_equal(u, wu)
    assert np.array_equal(rm, want)


def test_full_size_properties_1M_cells(ops):
    """BASELINE config 5 scale on one GPU (1 M cells x k = 30): size-independent properties."""
    import torch

    N, k = 1_000_000, 30
    mat = synth.knn_windowed(N, k, seed=11)
    rm, u = device_jaccard(ops, mat)
    E = N * k
    src, dst, w = rm[:, 0], rm[:, 1], rm[:, 2]
    pos = u > 0
    # edge list: src = cell id, dst = the input id, zero rows exactly where u == 0
    assert np.array_equal(src, np.where(pos, np.repeat(np.arange(1, N + 1, dtype=np.float64), k), 0.0))
    assert np.array_equal(dst, np.where(pos, mat.reshape(-1).astype(np.float64), 0.0))
    assert np.array_equal(w, np.where(pos, u / (2.0 * k - u), 0.0))
    assert u.min() >= 0 and u.max() <= k
    # symmetry: whenever j is in row i and i is in row j, u(i->j) == u(j->i) — on every 10th edge (3 M of them; the first version sorted
    # all 30 M edge keys on the host, a third of this test's 37 s)
    e = np.arange(0, E, 10, dtype=np.int64)
    ii, slot = e // k, e % k
    jj = mat.reshape(-1)[e].astype(np.int64) - 1
    back = mat[jj] == (ii + 1)[:, None]                      # where row j names i
    mutual = back.any(axis=1)
    assert mutual.mean() > 0.1
    u2 = u.reshape(N, k)
    assert np.array_equal(u2[ii[mutual], slot[mutual]], u2[jj[mutual], back[mutual].argmax(axis=1)])
    # a 20 k-cell sample of source cells against the oracle's counting rule (numpy restatement)
    sample = np.arange(0, N, 50)[:2000]
    rows = mat[sample]
    nb = mat[rows.reshape(-1) - 1].reshape(len(sample), k, k)
    cnt = (rows[:, None, :, None] == nb[:, :, None, :]).sum(axis=(2, 3)).astype(np.int32)
    assert np.array_equal(cnt.reshape(-1), u.reshape(N, k)[sample].reshape(-1))


def test_wide_offset_variant_matches(tmp_path):
    """The 64-bit-offset / 32-bit-hash kernel variant (used for N >= 2^24 or tables >= 4 GiB), forced
    through its test hook in a fresh process, gives the same bits."""
    import subprocess
    import sys

    code = (
        "import numpy as np, gficf_amd, oracle\n"
        "from gficf_amd import synth\n"
        "for N, k in ((5000, 30), (3000, 50), (2000, 15), (700, 100)):\n"
        "    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N)\n"
        "    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), oracle.jaccard(mat, nthreads=4)[0])\n"
        "print('ok')\n")
    env = dict(os.environ, GFICF_JACCARD_FORCE_BIG="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_wide_rows_for_small_data_sets_match(tmp_path):
    """Data sets of fewer than 2^17 cells store their table rows compactly (16-bit low halves + a bitmap of high
    bits); GFICF_JACCARD_COMPACT=0 keeps the 32-bit rows that larger data sets use.  Same bits either way, duplicate
    rows (slow path), the filtered build and ids above 2^16 included."""
    import subprocess
    import sys

    code = (
        "import numpy as np, gficf_amd, oracle\n"
        "from gficf_amd import synth\n"
        "for N, k in ((5000, 30), (70000, 30), (3000, 50), (2000, 17), (700, 100), (1000, 31), (400, 130)):\n"
        "    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N)\n"
        "    mat[5, 2] = mat[5, 0]\n"
        "    want = oracle.jaccard(mat, nthreads=4)[0]\n"
        "    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), want)\n"
        "    ed = gficf_amd.jaccard_edges(np.concatenate([np.arange(1, N + 1, dtype=mat.dtype)[:, None], mat], axis=1))\n"
        "    kp = want[:, 2] > 0\n"
        "    assert np.array_equal(ed['from'], want[kp, 0]) and np.array_equal(ed['to'], want[kp, 1]) and np.array_equal(ed['weight'], want[kp, 2])\n"
        "print('ok')\n")
    for compact in ("0", "1"):
        env = dict(os.environ, GFICF_JACCARD_COMPACT=compact, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (compact, r.stderr[-2000:])


def test_config3_shape_54k_cells_k30_bit_exact(ops):
    """BASELINE config 3 (Tabula-Muris-sized: ~54 k cells, k = 30 Phenograph graph): whole edge matrix and
    intersection counts bit-exact vs the oracle, through the device pipeline and through the host C ABI."""
    mat = synth.knn_windowed(54000, 30, seed=3)
    want, wu = oracle.jaccard(mat, nthreads=os.cpu_count() or 8)
    rm, u = device_jaccard(ops, mat)
    assert np.array_equal(u, wu)
    assert np.array_equal(rm, want)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), want)
    # ids as doubles (what Rcpp hands the reference after its INTSXP -> REALSXP copy)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat.astype(np.float64), False), want)


def test_pipelined_steps_give_the_same_bits(ops):
    """JaccardShard(pipeline=True): ingest of step s+1 overlaps edges of step s (two tables, two
    streams); different inputs on consecutive steps must not bleed into each other."""
    import torch
    from gficf_amd.dist import JaccardShard

    N, k = 20000, 30
    mats = [synth.knn_windowed(N, k, seed=s) for s in (1, 2, 3)]
    idx = [torch.from_numpy(np.ascontiguousarray(m.T)).cuda() for m in mats]
    want = [oracle.jaccard(m, nthreads=8)[0] for m in mats]
    sh = JaccardShard(ops, N, k, device="cuda", with_u=False, pipeline=True)
    got = []
    for rep in range(3):
        for i in range(3):
            out = sh.step(idx[i])
            sh.wait()                             # current stream waits for this step's edge kernel
            got.append((i, out.clone()))
    sh.sync()
    torch.cuda.synchronize()
    for i, g in got:
        assert np.array_equal(g.cpu().numpy().T, want[i])


@pytest.mark.parametrize("mode", [True, "force"])
def test_pipelined_contract_holds_when_one_rank_runs_in_order(ops, mode):
    """JaccardShard(pipeline=True) on ONE rank runs in order (pipeline_in_order) — and must still honour the contract the class states for
    the pipelined mode: the result of step s stays valid until the step after next, wait() orders ANOTHER stream behind the step, release()
    gates the reuse of a buffer.  A caller that issues step s + 1 and only then reads step s, on a stream of its own, gets step s's result;
    and fresh input blocks handed over every step are not kept alive by the shard (advisor, round 5)."""
    import gc
    import weakref

    import torch
    from gficf_amd.dist import JaccardShard

    N, k = 20000, 30
    mats = [synth.knn_windowed(N, k, seed=s) for s in (11, 12, 13, 14)]
    want = [oracle.jaccard(m, nthreads=8)[0] for m in mats]
    sh = JaccardShard(ops, N, k, device="cuda", with_u=False, pipeline=mode)
    assert sh.pipeline_in_order == (mode is True)
    reader = torch.cuda.Stream()
    copies, refs = [], []
    prev = None
    for rep in range(3):
        for i in range(4):
            blk = torch.from_numpy(np.ascontiguousarray(mats[i].T)).cuda()       # a FRESH block every step
            refs.append(weakref.ref(blk))
            out = sh.step(blk)
            with torch.cuda.stream(reader):
                sh.wait()                                                        # the reader's stream behind this step
                c = torch.empty_like(out)
                c.copy_(out, non_blocking=True)
                sh.release()
            if prev is not None:                                                 # the result of the step BEFORE is still its own (two buffers)
                with torch.cuda.stream(reader):
                    copies.append((prev[0], prev[1].clone()))
            prev = (i, out)
            copies.append((i, c))
            del blk
    sh.sync()
    torch.cuda.synchronize()
    for i, g in copies:
        assert np.array_equal(g.cpu().numpy().T, want[i]), i
    gc.collect()
    alive = sum(r() is not None for r in refs)
    assert alive <= 4, f"{alive} of {len(refs)} input blocks are still held"


@pytest.mark.parametrize("gen,N,k", [("win", 5000, 30), ("uni", 20000, 15), ("win", 700, 100), ("uni", 3000, 50)])
def test_device_edge_filter_matches_reference_filter(ops, gen, N, k):
    """N1: edges with weight > 0 only, in order == relations[relations[,3] > 0, ] (R/clustCells.R:66)."""
    import torch

    mat = synth.knn_windowed(N, k, W=max(100, k), seed=9) if gen == "win" else synth.knn_uniform(N, k, seed=9)
    want, _ = oracle.jaccard(mat, nthreads=8)
    want = want[want[:, 2] > 0]
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest(idx, N, k, N, table)
    # two cell blocks, concatenated, must equal the single filter
    parts = []
    for b, e in ((0, N // 3), (N // 3, N)):
        n = (e - b) * k
        u_ws = torch.empty(n, dtype=torch.int16, device="cuda")
        ptr = torch.empty(e - b + 1, dtype=torch.int64, device="cuda")
        out3 = torch.full((3, n), -1.0, dtype=torch.float64, device="cuda")
        ops.jaccard_edges_filtered(table, N, k, b, e, u_ws, ptr, out3)
        ops.sync()
        m = int(ptr[-1])
        parts.append(out3[:, :m].cpu().numpy().T)
        cnt = np.diff(ptr.cpu().numpy())
        assert cnt.min() >= 0 and cnt.max() <= k
    got = np.concatenate(parts, axis=0)
    assert np.array_equal(got, want)
    # host form through the clustcells() call-site mirror
    neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], mat], axis=1)
    rel = gficf_amd.jaccard_edges(neigh)
    assert np.array_equal(rel["from"], want[:, 0]) and np.array_equal(rel["to"], want[:, 1]) and np.array_equal(rel["weight"], want[:, 2])


def test_randomised_shapes_against_oracle(ops):
    """Fuzz: random N, k in [1, 256], windowed / uniform / duplicate-heavy ids, int32 and float64 input."""
    rng = np.random.default_rng(20261003)
    for case in range(40):
        k = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 30, 31, 32, 33, 50, 63, 64, 65, 100, 128, 129, 200, 255, 256]))
        N = int(rng.integers(max(k + 2, 4), 3000))
        mode = case % 3
        if mode == 0 and N > 2 * k + 2:
            mat = synth.knn_windowed(N, k, W=max(100, k), seed=case)
        elif mode == 1:
            mat = synth.knn_uniform(N, k, seed=case)
        else:
            mat = (synth.rand_u64(case, np.arange(N * k)).reshape(N, k) % np.uint64(min(N, 3 * k))).astype(np.int32) + 1
        if case % 4 == 0:
            mat = mat.astype(np.float64)
        got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
        want, _ = oracle.jaccard(mat, nthreads=8)
        assert np.array_equal(got, want), (case, N, k, mode)


def test_rccl_single_rank_collectives_and_bench_launch_path():
    """RCCL path smoke on the 1-GPU box: a 1-rank NCCL group accepts the in-place flat all-gather of
    table rows and the int64 all-reduce that the N > 1 path issues (gficf_amd/dist.py), and bench.py
    runs under torch.distributed.run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, numpy as np, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "import gficf_amd, oracle\n"
        "from gficf_amd import synth\n"
        "from gficf_amd.dist import JaccardShard, GficfShard, _all_gather_rows\n"
        "ops = gficf_amd.HipOps(0)\n"
        "N, k = 4000, 30\n"
        "mat = synth.knn_windowed(N, k, seed=2)\n"
        "sh = JaccardShard(ops, N, k, device='cuda', pipeline=True)\n"
        "idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()\n"
        "for _ in range(3): out = sh.step(idx)\n"
        "t = sh.table; mine = t[0:sh.rpr]\n"
        "_all_gather_rows(t.view(-1), mine.reshape(-1), None)\n"
        "sh.sync(); torch.cuda.synchronize()\n"
        "assert np.array_equal(out.cpu().numpy().T, oracle.jaccard(mat, nthreads=4)[0])\n"
        "cp, ri, x = synth.counts_csc(900, 500, seed=3)\n"
        "d = lambda a: torch.from_numpy(a).cuda()\n"
        "gs = GficfShard(ops, 900, 500, 500, len(ri), device='cuda')\n"
        "ws = gs.step(d(cp), d(ri), d(x), 0.05, 1.0)\n"
        "dist.all_reduce(ws['nt'], op=dist.ReduceOp.SUM)\n"
        "ops.sync(); ref = oracle.gficf_csc(900, 500, cp, ri, x, 0.05, 1.0)\n"
        "n = int(ws['out_colptr'][500]); assert np.allclose(ws['out_x'][:n].cpu().numpy(), ref['x'], rtol=1e-6, atol=1e-6)\n"
        # the halo form's two all-to-alls (equal splits, int32) and the bench's checker gather (list form), on RCCL
        "from gficf_amd.dist import _all_to_all\n"
        "a = torch.arange(4096, dtype=torch.int32, device='cuda'); b2 = torch.zeros_like(a)\n"
        "_all_to_all(b2, a, None, None, None); assert torch.equal(a, b2)\n"
        "b2.zero_(); _all_to_all(b2, a, [4096], [4096], None); assert torch.equal(a, b2)\n"
        "c64 = torch.tensor([7], dtype=torch.int64, device='cuda'); r64 = torch.zeros_like(c64)\n"
        "_all_to_all(r64, c64, None, None, None); assert int(r64) == 7\n"
        "lst = [torch.zeros(3, 5, dtype=torch.int32, device='cuda')]\n"
        "dist.all_gather(lst, torch.ones(3, 5, dtype=torch.int32, device='cuda')); assert int(lst[0].sum()) == 15\n"
        "dist.barrier(); dist.destroy_process_group(); print('ok')\n")
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-gficf"], env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert r.returncode == 0 and '"metric": "jaccard_edges_per_sec"' in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_config4_shape_100k_cells_k50_bit_exact(ops):
    """BASELINE config 4 shape (100 k cells, k = 50): whole edge matrix bit-exact vs the oracle."""
    mat = synth.knn_windowed(100000, 50, seed=4)
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=os.cpu_count() or 8)
    assert np.array_equal(u, wu) and np.array_equal(rm, want)


@pytest.mark.parametrize("N,k", [(5000, 30), (70000, 15), (3000, 50), (900, 100), (200000, 30)])
def test_packed_transport_rows_roundtrip(ops, N, k):
    """Transport form of table rows (bit-packed ids + duplicate flag): pack -> unpack is the identity,
    and a table rebuilt block-wise from packed rows yields the same edges."""
    import torch

    mat = synth.knn_windowed(N, k, W=max(100, k), seed=13)
    mat[7, 1] = mat[7, 0]                       # a row with duplicate ids: its flag must survive
    mat[N - 1, :] = mat[N - 1, 0]
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    kp, pw = ops.row_words(N, k), ops.packed_words(N, k)
    bits = int(np.ceil(np.log2(N + 1)))
    assert pw == (k * bits + 1 + 31) // 32 and pw * 4 <= ops.kpad(k) * 4
    dual = 32 < k <= 55 and N <= 131070        # compact row + planar copy for the bit-set kernel: 64 words
    assert kp == (64 if dual else ops.kpad(k) // 2 if N < 2 ** 17 and ops.kpad(k) >= 32 and k <= ops.kpad(k) * 15 // 16 else ops.kpad(k))
    table = torch.empty((N, kp), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest(idx, N, k, N, table)
    packed = torch.empty((N, pw), dtype=torch.int32, device="cuda")
    ops.jaccard_pack_rows(table, N, k, N, packed)
    back = torch.full((N, kp), -1, dtype=torch.int32, device="cuda")
    cut = N // 3
    ops.jaccard_unpack_rows(packed[:cut], cut, k, N, back[:cut])
    ops.jaccard_unpack_rows(packed[cut:], N - cut, k, N, back[cut:])
    ops.sync()
    assert torch.equal(back, table)
    flag_word = 31 if dual else kp - 1 if kp < ops.kpad(k) else 0   # compact rows keep the flag in their last word (dual rows: of the compact part ...)
    assert int(table[7, flag_word]) < 0 and int(table[N - 1, flag_word]) < 0      # duplicate flags set and preserved
    if dual:                                                        # ... and in the header of the planar part, with the groups of the first plane
        assert int(table[7, 63]) < 0 and int(table[N - 1, 63]) < 0 and 0 <= (int(table[0, 63]) & 15) <= 7
    if N <= 70000:
        out = torch.empty((3, N * k), dtype=torch.float64, device="cuda")
        ops.jaccard_edges(back, N, k, 0, N, out)
        ops.sync()
        assert np.array_equal(out.cpu().numpy().T, oracle.jaccard(mat, nthreads=8)[0])


@pytest.mark.parametrize("N,k", [(64, 5), (1000, 15), (3000, 30), (700, 33), (300, 100)])
def test_serial_entry_jaccard_coeff_matches_oracle(N, k):
    """The package's serial entry (reference src/jaccard_coeff.cpp:19-44): set intersection (Rcpp::intersect), rows
    with u > 0 packed from the top.  Bit-exact, including rows that hold an id several times and int32 / double input."""
    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k)
    mat[3, : min(3, k)] = mat[3, 0]                              # duplicates: here set and multiset counts differ
    mat[int(mat[3, 0]) - 1, : min(2, k)] = mat[3, 0]
    mat[N - 1, :] = 7
    for m in (mat, mat.astype(np.float64)):
        got = gficf_amd.jaccard_coeff(m, False)
        assert got.shape == (N * k, 3)
        assert np.array_equal(got, oracle.jaccard_coeff(mat))
    # the parallel entry counts the same rows as multisets: different weights on the duplicate rows, same elsewhere
    par = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    kept = par[par[:, 2] > 0]
    assert len(kept) == int((got[:, 2] > 0).sum())
    assert not np.array_equal(kept[:, 2], got[:len(kept), 2])


def test_serial_entry_empty_and_error_paths():
    assert gficf_amd.jaccard_coeff(np.zeros((0, 4), dtype=np.int32)).shape == (0, 3)
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.jaccard_coeff(np.array([[2, 9], [1, 1]], dtype=np.int32))
    assert ei.value.status == "GFICF_ERR_BAD_ID"


@pytest.mark.parametrize("N,k,P,cap,perm", [(30011, 30, 3, 512, False), (140000, 30, 2, 4096, False), (20000, 50, 4, 1024, False),
                                            (9000, 100, 2, 512, False), (6000, 15, 2, 4096, True), (5000, 30, 1, 64, False)])
def test_local_id_sub_problems_of_emulated_ranks(N, k, P, cap, perm):
    """The halo form of the sharded build (csrc/halo.hip + the mapped edge kernels), all ranks emulated one after the other on
    this GPU: plan -> (the two all-to-alls done by slicing) -> serve -> relabel -> ingest in local ids -> edges written through
    the local -> global map.  Bit-exact against the oracle on the whole matrix: pipelined and general kernels, compact rows
    for a data set of more than 2^17 cells (sub-problems below 2^17), wide rows (k = 100), scrambled ids with slots to spare."""
    import torch

    from gficf_amd.dist import rows_per_rank, shard_bounds

    ops = gficf_amd.HipOps(0)
    mat = synth.knn_windowed(N, k, seed=11, perm_seed=(5 if perm else None))
    want, _ = oracle.jaccard(mat, nthreads=8)
    rpr = rows_per_rank(N, P)
    blocks = [shard_bounds(N, P, r) for r in range(P)]
    idx = [torch.from_numpy(np.ascontiguousarray(mat[b:e].T)).cuda() for b, e in blocks]
    i32 = dict(dtype=torch.int32, device="cuda")
    req_out = [torch.zeros(P * cap, **i32) for _ in range(P)]
    wss = [torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda") for _ in range(P)]
    for r, (b, e) in enumerate(blocks):
        ops.halo_plan(idx[r], e - b, k, N, b, P, rpr, cap, wss[r], req_out[r])
    ops.sync()                                                   # (no overflow)
    # all-to-all #1: rank r's slots for owner p arrive at p as its slots from r
    req_in = [torch.cat([req_out[r][p * cap:(p + 1) * cap] for r in range(P)]) for p in range(P)]
    rows_out = [torch.zeros(P * cap * k, **i32) for _ in range(P)]
    for p, (b, e) in enumerate(blocks):
        ops.halo_serve(idx[p], e - b, k, b, req_in[p], rows_out[p])
    rows_in = [torch.cat([rows_out[p][r * cap * k:(r + 1) * cap * k] for p in range(P)]) for r in range(P)]
    got = []
    for r, (b, e) in enumerate(blocks):
        nl = e - b
        n_ext = nl + P * cap
        idx_ext, l2g = torch.zeros((k, n_ext), **i32), torch.zeros(n_ext, **i32)
        ops.halo_relabel(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], rows_in[r], idx_ext, l2g)
        table = torch.zeros((n_ext, ops.row_words(n_ext, k)), **i32)
        ops.jaccard_ingest_local(idx_ext, n_ext, k, table)
        # the fused pair (k <= 64: the own rows with the serve step, the halo slots behind the replies): the same table and the same map
        t2, g2 = torch.full_like(table, -1), torch.full_like(l2g, -1)
        scratch = torch.zeros_like(rows_out[r])
        if ops.halo_serve_ingest(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], req_in[r], scratch, t2, g2):
            ops.halo_ingest_slots(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], rows_in[r], t2, g2)
            assert k <= 64 and torch.equal(scratch, rows_out[r])
            used = int((req_out[r] != 0).sum())                 # (slots that no request fills stay what they were: compare the rows in use)
            ops.sync()
            rows_used = torch.cat([torch.arange(nl, device="cuda"), nl + torch.nonzero(req_out[r] != 0).flatten()])
            assert torch.equal(t2[rows_used], table[rows_used]) and torch.equal(g2[rows_used], l2g[rows_used]) and used == len(rows_used) - nl
        else:
            assert k > 64
        out = torch.zeros((3, nl * k), dtype=torch.float64, device="cuda")
        ops.jaccard_edges_mapped(table, n_ext, k, nl, b, l2g, out)
        ops.sync()
        if N >= (1 << 17):
            assert n_ext < (1 << 17) and ops.row_words(n_ext, k) < ops.row_words(N, k)      # compact rows for the sub-problem
        named = int((req_out[r] != 0).sum())
        if not perm and P > 1:
            assert 0 < named <= 4 * 100                         # the window either side of the block's seams
        got.append(out.cpu().numpy())
    assert np.array_equal(np.concatenate(got, axis=1).T, want)


def test_local_id_request_slots_overflow_is_reported():
    import torch

    ops = gficf_amd.HipOps(0)
    N, k, P, cap = 8000, 30, 2, 256
    mat = synth.knn_windowed(N, k, seed=3)                       # scrambled ids: the block names ~ every remote row
    idx = torch.from_numpy(np.ascontiguousarray(mat[:4000].T)).cuda()
    ws = torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda")
    req = torch.zeros(P * cap, dtype=torch.int32, device="cuda")
    ops.halo_plan(idx, 4000, k, N, 0, P, 4000, cap, ws, req)
    with pytest.raises(gficf_amd.GficfError) as ei:
        ops.sync()
    assert ei.value.status == "GFICF_ERR_CAPACITY"
    assert int((req[cap:] != 0).sum()) == cap and int((req[:cap] != 0).sum()) == 0     # the first cap ids of the other owner, none of its own
    ops.sync()


def test_truncate_noninteger_ids_strict_drop_in_mode():
    """gficf_ctx_set_jaccard_options: non-integer double ids handled as the reference does (row addressed by truncation,
    rows intersected as doubles, src/rcpp_parallel_jaccard_coeff.cpp:28-46) — against the oracle, which restates exactly
    that; rejected (GFICF_ERR_BAD_ID) by default; an all-integer double matrix takes the ordinary kernels either way."""
    rng = np.random.default_rng(8)
    N, k = 700, 12
    base = synth.knn_windowed(N, k, seed=2).astype(np.float64)
    frac = base + np.where(rng.random(base.shape) < 0.3, rng.choice([0.25, 0.5, 0.75], size=base.shape), 0.0)
    frac[5, 3] = 0.5                                              # (0, 1) -> row 1
    frac[7, :4] = frac[7, 0]                                      # the same double four times: multiset counting
    want, _ = oracle.jaccard(frac)
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.rcpp_parallel_jaccard_coef(frac, False)
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    got = gficf_amd.rcpp_parallel_jaccard_coef(frac, False, truncate_noninteger_ids=True)
    assert np.array_equal(got, want)
    assert (got[:, 2] > 0).sum() > 100 and np.any(got[:, 1] != np.floor(frac.reshape(-1)))      # truncated ids in column 2
    # integer-valued doubles: same result with and without the option (the ordinary path)
    w2, _ = oracle.jaccard(base)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(base, False, truncate_noninteger_ids=True), w2)
    # values the reference cannot address are still an error in strict mode
    bad = frac.copy()
    bad[0, 0] = N + 1.5
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.rcpp_parallel_jaccard_coef(bad, False, truncate_noninteger_ids=True)
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    # k beyond 256 is the sorted-row path (exact, see test_k_beyond_256_*); non-integer ids there are refused, and the message says so
    frac_wide = np.tile(frac[:, :1], (1, 300))
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.rcpp_parallel_jaccard_coef(frac_wide, False, truncate_noninteger_ids=True)
    assert ei.value.status == "GFICF_ERR_UNSUPPORTED" and "256" in str(ei.value)


@pytest.mark.parametrize("N,k", [(100_000, 50), (1_000_000, 30)])
def test_configs_4_and_5_as_eight_blocks_on_local_ids(N, k):
    """BASELINE configs 4 (100 k x k = 50) and 5 (1 M x k = 30) are 8-GPU configs: the data set as eight cell blocks, every
    block built as a sub-problem in local ids (plan / serve / fused relabel-and-ingest / mapped edge kernel), ranks emulated one
    after the other on this GPU, ids in spatial order (what the halo form is for).  Each block against the oracle on a sample
    of its cells (start, middle, end), and the whole by properties: source column, zero rows exactly where u = 0."""
    import torch

    from gficf_amd.dist import rows_per_rank, shard_bounds

    ops = gficf_amd.HipOps(0)
    P = 8
    mat = synth.knn_windowed(N, k, seed=4, perm_seed=None)
    rpr = rows_per_rank(N, P)
    cap = max(64, min(8192, ((1 << 17) - 1 - rpr) // P))
    blocks = [shard_bounds(N, P, r) for r in range(P)]
    i32 = dict(dtype=torch.int32, device="cuda")
    idx = [torch.from_numpy(np.ascontiguousarray(mat[b:e].T)).cuda() for b, e in blocks]
    req_out = [torch.zeros(P * cap, **i32) for _ in range(P)]
    wss = [torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda") for _ in range(P)]
    for r, (b, e) in enumerate(blocks):
        ops.halo_plan(idx[r], e - b, k, N, b, P, rpr, cap, wss[r], req_out[r])
    ops.sync()
    req_in = [torch.cat([req_out[r][p * cap:(p + 1) * cap] for r in range(P)]) for p in range(P)]
    rows_out = [torch.zeros(P * cap * k, **i32) for _ in range(P)]
    for p, (b, e) in enumerate(blocks):
        ops.halo_serve(idx[p], e - b, k, b, req_in[p], rows_out[p])
    for r, (b, e) in enumerate(blocks):
        nl, n_ext = e - b, e - b + P * cap
        rows_in = torch.cat([rows_out[p][r * cap * k:(r + 1) * cap * k] for p in range(P)])
        table = torch.zeros((n_ext, ops.row_words(n_ext, k)), **i32)
        l2g = torch.zeros(n_ext, **i32)
        idx_ext = torch.zeros((k, n_ext), **i32)
        ops.halo_relabel(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], rows_in, idx_ext, l2g)
        ops.jaccard_ingest_local(idx_ext, n_ext, k, table)
        out = torch.zeros((3, nl * k), dtype=torch.float64, device="cuda")
        ops.jaccard_edges_mapped(table, n_ext, k, nl, b, l2g, out)
        ops.sync()
        assert n_ext < (1 << 17) and 4 * ops.row_words(n_ext, k) == (64 if k <= 30 else 256)      # compact (k = 50: dual) rows at any N_total
        run = 512
        for c0 in sorted({b, b + (nl - run) // 2, e - run}):
            want, _ = oracle.jaccard_cells(mat, c0, c0 + run, nthreads=8)
            assert np.array_equal(out[:, (c0 - b) * k:(c0 - b + run) * k].cpu().numpy().T, want), (r, c0)
        src, dst, w = out[0], out[1], out[2]
        pos = w > 0
        cells = torch.repeat_interleave(torch.arange(b + 1, e + 1, device="cuda", dtype=torch.float64), k)
        assert torch.equal(src[pos], cells[pos]) and not bool(src[~pos].any()) and not bool(dst[~pos].any())
        assert torch.equal(dst[pos], idx[r].T.reshape(-1).double()[pos])
        assert float(pos.double().mean()) > 0.9
        del table, out, rows_in


def _emulated_halo_build(ops, mat, P, cap, split=False, wss=None):
    """All P ranks of the halo form, one after the other on this GPU; returns the full edge matrix and the rows named per rank.
    split: round 4's step — serve + the own cells' rows in one launch between the exchanges, the halo slots behind the second
    (k <= 64; beyond it the unfused calls).  wss: workspaces of an earlier build (the plan hands its bitmap back clear)."""
    import torch

    from gficf_amd.dist import rows_per_rank, shard_bounds

    N, k = mat.shape
    rpr = rows_per_rank(N, P)
    blocks = [shard_bounds(N, P, r) for r in range(P)]
    i32 = dict(dtype=torch.int32, device="cuda")
    idx = [torch.from_numpy(np.ascontiguousarray(mat[b:e].T)).cuda() if e > b else torch.zeros((k, 0), **i32) for b, e in blocks]
    req_out = [torch.zeros(P * cap, **i32) for _ in range(P)]
    if wss is None:
        wss = [torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda") for _ in range(P)]
    for r, (b, e) in enumerate(blocks):
        ops.halo_plan(idx[r], e - b, k, N, b, P, rpr, cap, wss[r], req_out[r])
    ops.sync()
    req_in = [torch.cat([req_out[r][p * cap:(p + 1) * cap] for r in range(P)]) for p in range(P)]
    rows_out = [torch.zeros(P * cap * k, **i32) for _ in range(P)]
    tables = [torch.zeros((e - b + P * cap, ops.row_words(e - b + P * cap, k)), **i32) for b, e in blocks]
    l2gs = [torch.zeros(e - b + P * cap, **i32) for b, e in blocks]
    done = []
    for p, (b, e) in enumerate(blocks):
        done.append(split and ops.halo_serve_ingest(idx[p], e - b, k, N, b, P, rpr, cap, wss[p], req_out[p], req_in[p], rows_out[p], tables[p], l2gs[p]))
        if not done[p]:
            ops.halo_serve(idx[p], e - b, k, b, req_in[p], rows_out[p])
    got, named = [], []
    for r, (b, e) in enumerate(blocks):
        nl, n_ext = e - b, e - b + P * cap
        rows_in = torch.cat([rows_out[p][r * cap * k:(r + 1) * cap * k] for p in range(P)])
        table, l2g = tables[r], l2gs[r]
        if done[r]:
            ops.halo_ingest_slots(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], rows_in, table, l2g)
        else:
            idx_ext = torch.zeros((k, n_ext), **i32)
            ops.halo_relabel(idx[r], nl, k, N, b, P, rpr, cap, wss[r], req_out[r], rows_in, idx_ext, l2g)
            ops.jaccard_ingest_local(idx_ext, n_ext, k, table)
        out = torch.zeros((3, nl * k), dtype=torch.float64, device="cuda")
        ops.jaccard_edges_mapped(table, n_ext, k, nl, b, l2g, out)
        ops.sync()
        got.append(out.cpu().numpy())
        named.append(int((req_out[r] != 0).sum()))
    _emulated_halo_build.last_wss = wss
    return np.concatenate(got, axis=1).T, named


def test_local_id_sub_problems_fuzz():
    """Random small cases of the halo form against the oracle: N not a multiple of the ranks, more ranks than cells (empty
    blocks), k from 1 to beyond the fused ingest's range, rows with duplicate ids and self references, ids without any
    locality with slots that are exactly enough, one slot too few (-> GFICF_ERR_CAPACITY)."""
    rng = np.random.default_rng(2024)
    ops = gficf_amd.HipOps(0)
    for case in range(24):
        N = int(rng.integers(1, 400))
        k = int(rng.choice([1, 2, 7, 15, 30, 33, 64, 65, 100]))
        P = int(rng.choice([1, 2, 3, 5, 8]))
        mat = rng.integers(1, N + 1, size=(N, k)).astype(np.int32)        # no locality, duplicates and self references included
        want, _ = oracle.jaccard(mat, nthreads=4)
        got, named = _emulated_halo_build(ops, mat, P, cap=max(N, 1))
        assert np.array_equal(got, want), (case, N, k, P)
        # round 4's step (serve riding with the own rows' ingest, the slots behind the second exchange), on the workspaces the
        # build above has used — the plan hands its bitmap back clear — and then once more on other data of the same shape
        got2, named2 = _emulated_halo_build(ops, mat, P, cap=max(N, 1), split=True, wss=_emulated_halo_build.last_wss)
        assert np.array_equal(got2, want) and named2 == named, (case, N, k, P, "split")
        mat_b = rng.integers(1, N + 1, size=(N, k)).astype(np.int32)
        got3, _ = _emulated_halo_build(ops, mat_b, P, cap=max(N, 1), split=True, wss=_emulated_halo_build.last_wss)
        assert np.array_equal(got3, oracle.jaccard(mat_b, nthreads=4)[0]), (case, N, k, P, "split, reused workspace")
    # the slots exactly suffice / are one short
    import torch

    from gficf_amd.dist import rows_per_rank

    N, k, P = 300, 10, 2
    mat = rng.integers(1, N + 1, size=(N, k)).astype(np.int32)
    _, named = _emulated_halo_build(ops, mat, P, cap=N)
    need = max(named)
    got, _ = _emulated_halo_build(ops, mat, P, cap=need)
    assert np.array_equal(got, oracle.jaccard(mat)[0])
    idx0 = torch.from_numpy(np.ascontiguousarray(mat[:150].T)).cuda() if named[0] >= named[1] else torch.from_numpy(np.ascontiguousarray(mat[150:].T)).cuda()
    b0 = 0 if named[0] >= named[1] else 150
    ws = torch.zeros(ops.halo_workspace_bytes(N, P), dtype=torch.uint8, device="cuda")
    req = torch.zeros(P * (need - 1), dtype=torch.int32, device="cuda")
    ops.halo_plan(idx0, 150, k, N, b0, P, rows_per_rank(N, P), need - 1, ws, req)
    with pytest.raises(gficf_amd.GficfError) as ei:
        ops.sync()
    assert ei.value.status == "GFICF_ERR_CAPACITY"


def test_truncation_mode_small_and_degenerate_shapes():
    """Strict truncation mode on the smallest shapes: one cell, k = 1, a value that truncates to the cell itself."""
    for mat in (np.array([[1.5]]), np.array([[1.0, 1.9]]), np.array([[2.5], [1.25]]), np.array([[0.25, 2.75, 3.0], [1.5, 1.5, 2.0], [3.9, 0.9, 1.0]])):
        want, _ = oracle.jaccard(mat)
        got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False, truncate_noninteger_ids=True)
        assert np.array_equal(got, want), mat


def _scramble16(lo):
    """The stored form of an id's low half in a compact row (csrc/jaccard.hip: scramble16)."""
    s = (lo * 0x9E37) & 0xFFFF
    return ((s >> 5) | (s << 11)) & 0xFFFF


def test_rows_taken_to_hold_distinct_ids_and_the_deferred_duplicate_error(ops):
    """gficf_ctx_set_jaccard_distinct: the ingest does not scan rows for a repeated id; the edge kernel, which inserts every
    row of its cell range into a hash set, must report EVERY row that repeats one (deferred GFICF_ERR_DUPLICATE_IDS) — wherever
    the two copies sit, in every row format, also when both copies overflow the set's two-slot bucket — and must leave clean
    input alone (same edges, no error).  With the option off the same matrices give the oracle's multiset result."""
    rng = np.random.default_rng(123)
    shapes = [(3000, 30), (5000, 15), (3000, 31), (2000, 50), (2000, 64), (900, 100), (700, 200), (400, 256), (140_000, 30), (50, 3)]
    for N, k in shapes:
        mat = synth.knn_windowed(N, k, W=max(100, k) if N > 600 else max((k + 1) // 2, 2), seed=N + k, perm_seed=N + k + 1) if N > 2 * k + 2 \
            else synth.knn_uniform(N, k, seed=N + k)
        want, wu = oracle.jaccard(mat, nthreads=os.cpu_count() or 8)
        ops.set_jaccard_distinct(True)
        try:
            rm, u = device_jaccard(ops, mat)                      # clean input: no error at the sync inside, same bits
            assert np.array_equal(rm, want) and np.array_equal(u, wu), (N, k)
            trials = 3 if N > 100_000 else 25
            for t in range(trials):
                m2 = mat.copy()
                r = int(rng.integers(0, N))
                a, b = rng.choice(k, size=2, replace=False) if k > 1 else (0, 0)
                if k == 1:
                    continue
                m2[r, a] = m2[r, b]
                with pytest.raises(gficf_amd.GficfError) as ei:
                    device_jaccard(ops, m2)
                assert ei.value.status == "GFICF_ERR_DUPLICATE_IDS", (N, k, r, a, b, ei.value.status)
        finally:
            ops.set_jaccard_distinct(False)
        m2 = mat.copy()
        m2[N // 2, 0] = m2[N // 2, k - 1]
        rm, u = device_jaccard(ops, m2)                           # option off: the scan flags the row, the exact path runs
        w2, wu2 = oracle.jaccard(m2, nthreads=os.cpu_count() or 8)
        assert np.array_equal(rm, w2) and np.array_equal(u, wu2), (N, k)
    # compact rows, SIX ids of a row in one bucket of the hash set (two slots): four of them overflow — all distinct: no error;
    # a repeated id among the overflowed ones: reported.  (More than six overflowed ids are reported as a repeat without
    # being compared — never seen on real input, it only costs the exact re-run: the host entry still returns the oracle's edges.)
    N, k = 60_000, 30
    ids = np.arange(1, N + 1)
    bucket = np.array([_scramble16(int(i) & 0xFFFF) & 0x7F8 for i in ids])
    group = ids[bucket == bucket[0]]
    others = ids[bucket != bucket[0]]
    mat = synth.knn_windowed(N, k, seed=5, perm_seed=6)
    ops.set_jaccard_distinct(True)
    try:
        for t in range(20):
            m2 = mat.copy()
            r = int(rng.integers(0, N))
            cand = rng.permutation(others)
            _, first = np.unique(bucket[cand - 1], return_index=True)      # the other ids: no two in one bucket
            row = np.concatenate([rng.permutation(group)[:6], cand[np.sort(first)][:k - 6]])
            pos = rng.permutation(k)
            m2[r] = row[pos]
            device_jaccard(ops, m2)                               # distinct ids, six in one bucket: no error
            where = np.flatnonzero(pos < 6)                       # slots holding the six ids of the crowded bucket
            a, b = rng.choice(where, size=2, replace=False)
            m2[r, a] = m2[r, b]
            with pytest.raises(gficf_amd.GficfError) as ei:
                device_jaccard(ops, m2)
            assert ei.value.status == "GFICF_ERR_DUPLICATE_IDS", (t, r, a, b)
    finally:
        ops.set_jaccard_distinct(False)
    m2 = mat.copy()
    m2[123] = rng.permutation(group)[:k]                          # thirty distinct ids in one bucket: reported, healed by the host entry
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(m2, False), oracle.jaccard(m2, nthreads=8)[0])


def test_host_entries_take_the_fast_sequence_and_heal_themselves(monkeypatch):
    """The `.Call` entry, the counts, the filtered form and the serial entry on a matrix with repeated ids in some rows: the
    reference's multiset / set results (the exact sequence re-run behind the deferred error), banner lines printed once."""
    N, k = 4000, 30
    mat = synth.knn_windowed(N, k, seed=31, perm_seed=32)
    mat[17, 3] = mat[17, 9]
    mat[N - 1, 0] = mat[N - 1, k - 1]
    want, wu = oracle.jaccard(mat, nthreads=8)
    lines = []
    ctx = gficf_amd.default_context()
    got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    assert np.array_equal(got, want)
    assert np.array_equal(gficf_amd.jaccard_counts(mat).astype(np.int32).reshape(-1), wu)
    neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], mat], axis=1)
    e = gficf_amd.jaccard_edges(neigh)
    keep = want[:, 2] > 0
    assert np.array_equal(e["from"], want[keep, 0]) and np.array_equal(e["to"], want[keep, 1]) and np.array_equal(e["weight"], want[keep, 2])
    assert np.array_equal(gficf_amd.jaccard_coeff(mat, False), oracle.jaccard_coeff(mat))
    ctx.sync()                                                    # nothing deferred is left behind


def _uniform_rows(N, k, seed):
    """k distinct ids per row drawn uniformly from all N cells (NOT what a kNN search returns: no neighbourhood)."""
    rng = np.random.default_rng(seed)
    return (np.argsort(rng.random((N, N), dtype=np.float32), axis=1)[:, :k] + 1).astype(np.int32)


@pytest.mark.parametrize("k", [200, 230, 256])
def test_uniformly_spread_ids_at_k_near_256_no_false_duplicate_report_and_no_cliff(ops, k):
    """Round 6.  With ids spread uniformly over thousands of cells at k near 256 more than six ids of a row find their hash-set bucket full.
    Through round 5 that was reported as GFICF_ERR_DUPLICATE_IDS (no id repeats) and, with the duplicate scan on, sent the cell down the
    all-pairs path (22 ms at 5 000 x 256 against 1.7 ms at k = 257).  Now: (a) rows taken to hold distinct ids -> a deferred
    GFICF_ERR_SET_OVERFLOW, its own status; (b) the host entry re-runs on the sorted-row path by itself: the reference's matrix, bit for bit;
    (c) the exact device sequence walks the overflow list instead of falling to all-pairs: bit-exact and within a few milliseconds."""
    import time

    import torch

    N = 5000
    mat = _uniform_rows(N, k, seed=k)
    want, _ = oracle.jaccard(mat, nthreads=16)
    # (b) the `.Call` entry's mirror
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), want)
    # (a) distinct mode on the device entries
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
    out = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    ops.set_jaccard_distinct(True)
    try:
        ops.jaccard(idx, N, k, table, out, None)
        try:
            ops.sync()
            overflowed = False
        except gficf_amd.GficfError as e:
            overflowed = True
            assert e.status == "GFICF_ERR_SET_OVERFLOW", e.status          # never GFICF_ERR_DUPLICATE_IDS: no id repeats
        if not overflowed:                                                 # (no row happened to overflow: then the result must be right)
            assert np.array_equal(out.cpu().numpy().T, want)
    finally:
        ops.set_jaccard_distinct(False)
    # (c) the exact device sequence
    ops.jaccard(idx, N, k, table, out, None)
    ops.sync()
    t0 = time.perf_counter()
    ops.jaccard(idx, N, k, table, out, None)
    ops.sync()
    dt = time.perf_counter() - t0
    assert np.array_equal(out.cpu().numpy().T, want)
    assert dt < 0.008, f"{1e3 * dt:.1f} ms: the all-pairs cliff is back"


@pytest.mark.parametrize("N,k,P", [(30_000, 30, 3), (20_000, 50, 2), (9_000, 15, 4)])
def test_halo_form_with_rows_taken_to_hold_distinct_ids(ops, N, k, P):
    """The scan-less sequence in the sharded build on local ids (emulated ranks): clean input gives the oracle's edges with no
    error on any rank; a repeated id planted in a cell raises GFICF_ERR_DUPLICATE_IDS at the sync of the rank that OWNS the
    cell (its mapped edge kernel inserts the row) — the job's signal to re-run with the option off."""
    from gficf_amd.dist import shard_bounds

    mat = synth.knn_windowed(N, k, seed=N + k, perm_seed=None)      # ids with locality: the halo form fits
    want, _ = oracle.jaccard(mat, nthreads=8)
    ops.set_jaccard_distinct(True)
    try:
        got, named = _emulated_halo_build(ops, mat, P, 1024)
        assert np.array_equal(got, want) and max(named) < 1024
        rng = np.random.default_rng(N)
        for owner in range(P):
            b, e = shard_bounds(N, P, owner)
            m2 = mat.copy()
            r = int(rng.integers(b, e))
            a, c = rng.choice(k, size=2, replace=False)
            m2[r, a] = m2[r, c]
            with pytest.raises(gficf_amd.GficfError) as ei:
                _emulated_halo_build(ops, m2, P, 1024)
            assert ei.value.status == "GFICF_ERR_DUPLICATE_IDS", (owner, r)
    finally:
        ops.set_jaccard_distinct(False)
    m2 = mat.copy()
    m2[N // 2, 0] = m2[N // 2, k - 1]
    got, _ = _emulated_halo_build(ops, m2, P, 1024)                  # option off: flags, exact path, the oracle's multiset result
    assert np.array_equal(got, oracle.jaccard(m2, nthreads=8)[0])


@pytest.mark.parametrize("N,k", [(200, 33), (5000, 40), (5000, 41), (70000, 48), (70000, 49), (66000, 55), (131070, 50), (131071, 50),
                                 (65535, 50), (65536, 50), (65537, 35), (3000, 56)])
def test_dual_rows_and_the_bit_set_kernel_across_their_boundaries(ops, N, k):
    """Round 4: 32 < k <= 55 below 131 071 cells takes dual rows (compact row + planar copy) and k_jaccard_edges_bits — 5 / 6 / 7
    gather steps (k = 40 | 41, 48 | 49), ids in one plane only (N <= 65 535), the first id of the second plane (65 536), the last
    N the format serves (131 070) and the first it does not, k = 56 back on plain compact rows — every output form (matrix, matrix +
    counts, uint16 counts through the edge filter), cell ranges that do not start at 0, both settings of the distinct-ids option and
    the A/B switch GFICF_JACCARD_DUAL=0 on the same input: bit-exact against the oracle on a sample of cells."""
    import torch

    mat = synth.knn_windowed(N, k, W=max(100, k), seed=N % 1000 + k, perm_seed=11)
    mat[5] = [N, N - 1] + list(range(1, k - 1))             # the largest ids in a row and the smallest: both ends of both planes (distinct)
    dual = 32 < k <= 55 and N <= 131070
    assert ops.row_words(N, k) == (64 if dual else 32)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    cb, ce = (0, N) if N <= 5000 else (N // 3, N // 3 + 4001)
    want, wu = oracle.jaccard_cells(mat, cb, ce, nthreads=8)
    n = (ce - cb) * k
    for env in ({}, {"GFICF_JACCARD_DUAL": "0"}):
        os.environ.update(env)
        try:
            rw = ops.row_words(N, k)
            for distinct in (False, True):
                ops.set_jaccard_distinct(distinct)
                table = torch.zeros((N, rw), dtype=torch.int32, device="cuda")
                ops.jaccard_ingest(idx, N, k, N, table)
                out, u = torch.zeros((3, n), dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
                ops.jaccard_edges(table, N, k, cb, ce, out, None)
                ops.sync()
                assert np.array_equal(out.cpu().numpy().T, want), (env, distinct, "matrix")
                out.zero_()
                ops.jaccard_edges(table, N, k, cb, ce, out, u)
                ops.sync()
                assert np.array_equal(out.cpu().numpy().T, want) and np.array_equal(u.cpu().numpy(), wu), (env, distinct, "matrix + counts")
                u16 = torch.zeros(n, dtype=torch.int16, device="cuda")
                ptr = torch.zeros(ce - cb + 1, dtype=torch.int64, device="cuda")
                out.zero_()
                ops.jaccard_edges_filtered(table, N, k, cb, ce, u16, ptr, out)
                ops.sync()
                kept = want[want[:, 2] > 0]
                assert int(ptr[-1]) == len(kept) and np.array_equal(out[:, :len(kept)].cpu().numpy().T, kept), (env, distinct, "filtered")
        finally:
            ops.set_jaccard_distinct(False)
            for key in env:
                del os.environ[key]


# ---------------------------------------------------------------- k > 256: the sorted-row path (csrc/jaccard_sorted.h; round 5)
# The reference loops over mat.ncol() without a limit (src/rcpp_parallel_jaccard_coeff.cpp:26-46); the fast kernels stop at 256
# slots, beyond that rows are sorted once at ingest and an edge is k binary searches.  Same entries, same bits.
@pytest.mark.parametrize("N", [600, 5000])
@pytest.mark.parametrize("k", [257, 300, 513])
def test_k_beyond_256_matches_oracle(N, k):
    if N > 2 * k + 2:
        mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k, perm_seed=7)
    else:                                                       # 600 cells cannot hold a window of 2 x 513 + 1: distinct random ids
        mat = np.stack([np.random.default_rng(N * 1000 + k + i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32)
    got = gficf_amd.rcpp_parallel_jaccard_coef(mat, False)
    want, _ = oracle.jaccard(mat, nthreads=8)
    assert got.shape == (N * k, 3) and np.array_equal(got, want)
    assert (want[:, 2] > 0).mean() > 0.5
    if N == 600:                                                # REALSXP input (what Rcpp coerces to) and Fortran order
        assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(np.asfortranarray(mat.astype(np.float64)), False), want)


@pytest.mark.parametrize("k,mod", [(257, 300), (300, 90), (400, 1000)])
def test_k_beyond_256_multiset_and_set_semantics(ops, k, mod):
    """Rows that repeat ids (and name themselves): std::set_intersection's multiset counts for the parallel entry,
    Rcpp::intersect's set counts for the serial one (src/jaccard_coeff.cpp:33), the weight > 0 filter and the compact return."""
    N = 1100
    mat = (synth.rand_u64(k, np.arange(N * k)).reshape(N, k) % np.uint64(mod)).astype(np.int32) + 1
    rm, u = device_jaccard(ops, mat)
    want, wu = oracle.jaccard(mat, nthreads=8)
    assert np.array_equal(u, wu) and np.array_equal(rm, want)
    assert wu.max() > 1
    assert np.array_equal(gficf_amd.jaccard_coeff(mat, False), oracle.jaccard_coeff(mat))
    cnt = gficf_amd.jaccard_counts(mat)
    assert np.array_equal(cnt.astype(np.int32).reshape(-1), wu)
    assert np.array_equal(gficf_amd.jaccard_expand(mat, cnt), want)
    neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], mat], axis=1)
    rel = gficf_amd.jaccard_edges(neigh)
    keep = want[:, 2] > 0
    assert np.array_equal(rel["from"], want[keep, 0]) and np.array_equal(rel["to"], want[keep, 1]) and np.array_equal(rel["weight"], want[keep, 2])


def test_k_beyond_256_cell_ranges_bad_ids_and_devices(ops):
    import torch

    N, k = 900, 320
    mat = np.stack([np.random.default_rng(77 + i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32)
    want, wu = oracle.jaccard(mat, nthreads=8)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    rw = ops.row_words(N, k)
    assert rw == 640 == ops.kpad(k)
    table = torch.zeros((N, rw), dtype=torch.int32, device="cuda")
    # ingest in two blocks (the seam a sharded caller uses), edges of a cell range in the middle, with counts
    ops.jaccard_ingest(idx[:, :500].contiguous(), 500, k, N, table[:500])
    ops.jaccard_ingest(idx[:, 500:].contiguous(), 400, k, N, table[500:])
    b, e = 123, 777
    out = torch.full((3, (e - b) * k), -7.0, dtype=torch.float64, device="cuda")
    u = torch.full(((e - b) * k,), -7, dtype=torch.int32, device="cuda")
    ops.jaccard_edges(table, N, k, b, e, out, u)
    ops.sync()
    assert np.array_equal(out.cpu().numpy().T, want[b * k:e * k]) and np.array_equal(u.cpu().numpy(), wu[b * k:e * k])
    # the sorted half of a row is the row sorted (what the edge kernel searches)
    t = table.cpu().numpy().view(np.uint32)
    assert np.array_equal(t[:, :k], mat.astype(np.uint32)) and np.array_equal(t[:, 320:320 + k], np.sort(mat, axis=1).astype(np.uint32))
    # transport form of the rows: the rows themselves
    packed = torch.zeros((N, ops.packed_words(N, k)), dtype=torch.int32, device="cuda")
    ops.jaccard_pack_rows(table, N, k, N, packed)
    t2 = torch.zeros_like(table)
    ops.jaccard_unpack_rows(packed, N, k, N, t2)
    ops.sync()
    assert torch.equal(t2, table)
    # an id outside [1, N] is an error, as below 256
    bad = mat.copy()
    bad[5, 300] = N + 1
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.rcpp_parallel_jaccard_coef(bad, False)
    assert ei.value.status == "GFICF_ERR_BAD_ID"
    # several devices behind the host entry (blocks + peer copies of sorted rows): same matrix
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False, devices=[0, 0, 0]), want)


def test_k_in_the_thousands_rows_searched_in_the_table(ops):
    """k = 3000 (the edge kernel no longer stages rows in LDS) against the oracle on a few cells (the whole matrix would take it
    minutes), and k = 17000 > N (rows sorted in place in the table; every row repeats ids) against the multiset rule written out
    with numpy: u = sum over ids of min(multiplicity in row i, multiplicity in the neighbour's row)."""
    import torch

    N, k, (b, e) = 3200, 3000, (0, 3)
    mat = np.stack([np.random.default_rng(5 + i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32)
    mat[1, :40] = mat[1, 40]                                         # one row that repeats an id
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((N, ops.row_words(N, k)), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest(idx, N, k, N, table)
    out = torch.zeros((3, (e - b) * k), dtype=torch.float64, device="cuda")
    ops.jaccard_edges(table, N, k, b, e, out, None)
    ops.sync()
    want, _ = oracle.jaccard_cells(mat, b, e, nthreads=8)
    assert np.array_equal(out.cpu().numpy().T, want)
    del table, idx, out

    N, k, cell = 1500, 17000, 700
    mat = (synth.rand_u64(k, np.arange(N * k)).reshape(N, k) % np.uint64(N)).astype(np.int32) + 1
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    kp = (k + 63) // 64 * 64
    assert ops.row_words(N, k) == 2 * kp
    table = torch.zeros((N, 2 * kp), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest(idx, N, k, N, table)
    out = torch.zeros((3, k), dtype=torch.float64, device="cuda")
    u = torch.zeros(k, dtype=torch.int32, device="cuda")
    ops.jaccard_edges(table, N, k, cell, cell + 1, out, u)
    ops.sync()
    t = table[:64].cpu().numpy().view(np.uint32)
    assert np.array_equal(t[:, kp:kp + k], np.sort(mat[:64], axis=1).astype(np.uint32)) and (t[:, kp + k:] == 0xFFFFFFFF).all()
    cnt = np.stack([np.bincount(mat[r], minlength=N + 1) for r in range(N)])             # multiplicity of every id in every row
    wu = np.minimum(cnt[cell][None, :], cnt[mat[cell] - 1]).sum(axis=1).astype(np.int32)
    assert np.array_equal(u.cpu().numpy(), wu) and wu.min() > 0
    got = out.cpu().numpy()
    assert np.array_equal(got[0], np.full(k, cell + 1.0)) and np.array_equal(got[1], mat[cell].astype(np.float64))
    assert np.array_equal(got[2], wu / (2.0 * k - wu))                                    # reference :51, same IEEE division


# ---------------------------------------------------------------- small problems in ONE launch (csrc/jaccard_direct.h; round 5)
@pytest.mark.parametrize("N,k", [(1, 1), (2, 1), (64, 5), (500, 1), (777, 16), (777, 17), (1000, 32), (3000, 15), (10000, 30), (40000, 9)])
def test_one_launch_form_matches_oracle_and_the_table_path(ops, N, k):
    """gficf_jaccard_device under set_jaccard_distinct for N * k below the threshold: one kernel straight from the column-major
    input (no table).  Bit for bit the oracle's matrix and the table path's, int32 and double ids, with and without counts."""
    import torch

    if N > 2 * k + 2:
        mat = synth.knn_windowed(N, k, W=max(100, k), seed=N + k, perm_seed=3)
    else:
        mat = np.stack([np.random.default_rng(i).permutation(N)[:k] + 1 for i in range(N)]).astype(np.int32).reshape(N, k)
    want, wu = oracle.jaccard(mat, nthreads=8)
    ops.set_jaccard_distinct(True)
    try:
        for limit, dt in ((10**7, torch.int32), (10**7, torch.float64), (0, torch.int32)):          # 0: the table path, for comparison
            ops.set_jaccard_direct_max_edges(limit)
            idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda().to(dt)
            table = torch.full((N, ops.kpad(k)), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            rmat = torch.full((3, N * k), -7.0, dtype=torch.float64, device="cuda")
            u = torch.full((N * k,), -7, dtype=torch.int32, device="cuda")
            ops.jaccard(idx, N, k, table, rmat, u)
            ops.sync()
            assert np.array_equal(rmat.cpu().numpy().T, want) and np.array_equal(u.cpu().numpy(), wu), (limit, dt)
            assert bool((table == 0x5A5A5A5A).all()) == (limit > 0)                               # one launch: the table is not touched
            run = ops.jaccard_prepared(idx, N, k, table, rmat.fill_(-7.0), None)                   # the prepared single-call form
            run()
            ops.sync()
            assert np.array_equal(rmat.cpu().numpy().T, want)
    finally:
        ops.set_jaccard_direct_max_edges(-1)
        ops.set_jaccard_distinct(False)


def test_one_launch_form_reports_bad_and_repeated_ids(ops):
    import torch

    N, k = 2000, 15
    mat = synth.knn_windowed(N, k, seed=4)
    table = torch.zeros((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    rmat = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    ops.set_jaccard_distinct(True)
    ops.set_jaccard_direct_max_edges(10**7)
    try:
        for r, c, v, status in ((17, 3, N + 1, "GFICF_ERR_BAD_ID"), (1999, 14, 0, "GFICF_ERR_BAD_ID"), (5, 2, None, "GFICF_ERR_DUPLICATE_IDS")):
            m = mat.copy()
            m[r, c] = m[r, c + 1 if c + 1 < k else 0] if v is None else v
            ops.jaccard(torch.from_numpy(np.ascontiguousarray(m.T)).cuda(), N, k, table, rmat, None)
            with pytest.raises(gficf_amd.GficfError) as ei:
                ops.sync()
            assert ei.value.status == status
        # a non-integer double is a bad id here too
        md = mat.astype(np.float64)
        md[100, 0] += 0.5
        ops.jaccard(torch.from_numpy(np.ascontiguousarray(md.T)).cuda(), N, k, table, rmat, None)
        with pytest.raises(gficf_amd.GficfError) as ei:
            ops.sync()
        assert ei.value.status == "GFICF_ERR_BAD_ID"
    finally:
        ops.set_jaccard_direct_max_edges(-1)
        ops.set_jaccard_distinct(False)
    # the host entries take the one-launch form first and re-run the exact sequence for a matrix whose rows repeat ids
    dup = (synth.rand_u64(3, np.arange(900 * 12)).reshape(900, 12) % np.uint64(40)).astype(np.int32) + 1
    want, wu = oracle.jaccard(dup, nthreads=4)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(dup, False), want)
    assert np.array_equal(gficf_amd.jaccard_counts(dup).astype(np.int32).reshape(-1), wu)


def test_k_beyond_256_rows_in_local_ids_with_empty_slots_and_the_mapped_neighbour_column(ops):
    """The sorted-row path behind the sub-problem entries (gficf_jaccard_ingest_local_device: ids in [0, n_ext], 0 = no id in the slot;
    gficf_jaccard_edges_mapped_device: column 1 = src_offset + cell + 1, column 2 through the local -> global map): an empty slot gives
    a zero row and is no element of any set.  Expected counts from the multiset rule written out with numpy."""
    import torch

    n_ext, n_cells, k, src_off = 700, 500, 300, 12345
    rng = np.random.default_rng(8)
    mat = np.stack([rng.permutation(n_ext)[:k] + 1 for _ in range(n_ext)]).astype(np.int32)
    mat[rng.random(mat.shape) < 0.1] = 0                                  # empty slots (also in the own rows)
    mat[3, :5] = mat[3, 5] if mat[3, 5] else 7                             # ... and a row that repeats an id
    l2g = (rng.permutation(50_000)[:n_ext] + 1).astype(np.int32)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.zeros((n_ext, ops.row_words(n_ext, k)), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest_local(idx, n_ext, k, table)
    out = torch.full((3, n_cells * k), -7.0, dtype=torch.float64, device="cuda")
    u = torch.full((n_cells * k,), -7, dtype=torch.int32, device="cuda")
    ops.jaccard_edges_mapped(table, n_ext, k, n_cells, src_off, torch.from_numpy(l2g).cuda(), out, u)
    ops.sync()
    cnt = np.stack([np.bincount(r[r > 0], minlength=n_ext + 1) for r in mat])            # multiplicity of every id (0 left out) per row
    wu = np.zeros((n_cells, k), dtype=np.int32)
    for i in range(n_cells):
        nb = mat[i]
        wu[i] = np.where(nb > 0, np.minimum(cnt[i][None, :], cnt[np.maximum(nb, 1) - 1]).sum(axis=1), 0)
    wu = wu.reshape(-1)
    assert np.array_equal(u.cpu().numpy(), wu)
    pos = wu > 0
    got = out.cpu().numpy()
    cell = np.repeat(np.arange(n_cells), k)
    assert np.array_equal(got[0], np.where(pos, src_off + cell + 1.0, 0.0))
    assert np.array_equal(got[1], np.where(pos, l2g[np.maximum(mat[:n_cells].reshape(-1), 1) - 1].astype(np.float64), 0.0))
    assert np.array_equal(got[2], np.where(pos, wu / (2.0 * k - wu), 0.0))


@pytest.mark.parametrize("as_double", [False, True])
def test_host_entry_compact_return_from_a_million_edges_on(as_double):
    """gficf_jaccard_host from 2^20 edges on: uint16 counts cross PCIe and the host cores write the reference's (N*k) x 3 matrix
    (round 5; 4.2 -> 1.8 ms per call at 100 k x 30 into a fresh R matrix).  The same matrix, bit for bit — also for a matrix with a row
    that repeats an id (the fast sequence raises, the entry re-runs the exact one) and with counts above 255."""
    N, k = 36_000, 30
    mat = synth.knn_windowed(N, k, seed=77, perm_seed=78)
    assert N * k >= (1 << 20)
    m = mat.astype(np.float64) if as_double else mat
    want, _ = oracle.jaccard(mat, nthreads=8)
    got = gficf_amd.rcpp_parallel_jaccard_coef(m, False)
    assert got.flags["F_CONTIGUOUS"] or got.flags["C_CONTIGUOUS"]
    assert np.array_equal(got, want)
    dup = mat.copy()
    dup[123, 7] = dup[123, 8]
    dup[N - 1, :] = dup[N - 1, 0]
    wd, _ = oracle.jaccard(dup, nthreads=8)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(dup.astype(np.float64) if as_double else dup, False), wd)
    # the call-site mirror (gficf_jaccard_filtered_host_plan / _finish) takes the same compact form: kept rows written on the host, in order
    for src, w_ in ((mat, want), (dup, wd)):
        neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], src], axis=1)
        rel = gficf_amd.jaccard_edges(neigh.astype(np.float64) if as_double else neigh)
        keep = w_[:, 2] > 0
        assert np.array_equal(rel["from"], w_[keep, 0]) and np.array_equal(rel["to"], w_[keep, 1]) and np.array_equal(rel["weight"], w_[keep, 2])
    if not as_double:                                             # k = 300 (sorted rows), counts up to 300 > 255: still uint16 on the wire
        N2, k2 = 3600, 300
        m2 = synth.knn_windowed(N2, k2, W=151, seed=5, perm_seed=6)          # the tightest window: neighbouring rows are almost the same set
        w2, u2 = oracle.jaccard(m2, nthreads=8)
        assert u2.max() > 255 and N2 * k2 >= (1 << 20)
        assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(m2, False), w2)


@pytest.mark.parametrize("N,k", [(3000, 15), (70_000, 15), (5000, 30), (140_000, 30), (4000, 50), (1500, 100), (900, 256), (700, 300), (1100, 513)])
def test_every_kernel_family_against_the_closed_form_of_cyclic_windows(N, k):
    """Row i names the next k cells of a ring: u(i, t) = k - 1 - t for every cell, by counting — no oracle involved
    (tests/helpers/closed_form.py).  One shape per kernel: the one-launch form (3 000 x 15), the pipelined kernel on
    wide 16-slot and compact / wide 32-slot rows, the bit-set kernel (k = 50), the general kernel (k = 100, 256), the sorted-row path
    (k = 300, 513); through the host entry, so from 2^20 edges on through the compact return too."""
    from tests.helpers.closed_form import cyclic_window_expected, cyclic_window_matrix

    mat = cyclic_window_matrix(N, k)
    want, wu = cyclic_window_expected(N, k)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), want)
    assert np.array_equal(gficf_amd.jaccard_counts(mat).astype(np.int32).reshape(-1), wu)


@pytest.mark.parametrize("N,k", [(3000, 16), (4000, 30), (2500, 50), (900, 128), (800, 300)])
def test_multiset_and_set_semantics_against_the_closed_form(N, k):
    """Every id named twice in a row (the exact paths: the fast sequence raises, the entry re-runs; beyond 256 the sorted rows): the parallel
    entry counts multisets, 2 (k/2 - d), the serial entry sets, k/2 - d — derived by counting, no oracle in the loop."""
    from tests.helpers.closed_form import cyclic_window_matrix_twice, cyclic_window_twice_expected

    mat = cyclic_window_matrix_twice(N, k)
    want, wu = cyclic_window_twice_expected(N, k)
    assert np.array_equal(gficf_amd.rcpp_parallel_jaccard_coef(mat, False), want)
    assert np.array_equal(gficf_amd.jaccard_counts(mat).astype(np.int32).reshape(-1), wu)
    ws, _ = cyclic_window_twice_expected(N, k, set_semantics=True)
    ws = ws[ws[:, 2] > 0]
    gs = gficf_amd.jaccard_coeff(mat, False)
    assert np.array_equal(gs[:len(ws)], ws) and not gs[len(ws):].any()
