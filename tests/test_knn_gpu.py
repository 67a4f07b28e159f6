"""GPU parity of the exact kNN search ("next" row N2, the step in front of the Jaccard build:
reference R/clustCells.R:57,60) against the CPU oracle.  Bar: index matrix bit-exact (the oracle
evaluates the same f32 chain, ties broken by index), distances equal as f32; against the float64
numpy restatement distances within 1e-5 relative."""
import numpy as np
import pytest

import gficf_amd
import oracle
from oracle import oracle_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return gficf_amd.HipOps(0)


def blobs(N, d, seed, centers=12):
    """Clustered points (like cells in PCA space): Gaussian blobs with unequal spreads."""
    rng = np.random.default_rng(seed)
    c = rng.normal(scale=6.0, size=(centers, d))
    lab = rng.integers(0, centers, size=N)
    return c[lab] + rng.normal(size=(N, d)) * rng.uniform(0.5, 2.0, size=(centers, 1))[lab]


@pytest.mark.parametrize("metric", ["manhattan", "euclidean", "cosine"])
@pytest.mark.parametrize("N,d,k", [(300, 2, 5), (1000, 50, 16), (2500, 50, 31), (2000, 7, 31), (1500, 64, 51),
                                   (700, 128, 33), (900, 20, 100), (130, 3, 128), (129, 50, 1), (4000, 50, 31)])
def test_host_api_matches_oracle(metric, N, d, k):
    X = blobs(N, d, seed=N + d + k)
    got = gficf_amd.find_nn(X, k, True, metric)
    widx, wdist = oracle.knn(X, k, metric, nthreads=8)
    assert got["idx"].shape == (N, k) and got["idx"].dtype == np.int32
    assert np.array_equal(got["idx"], widx)
    assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32))
    # the row itself comes first (distinct points)
    if metric != "cosine":
        assert np.array_equal(got["idx"][:, 0], np.arange(1, N + 1))
        assert np.all(got["dist"][:, 0] == 0.0)


@pytest.mark.parametrize("metric", ["manhattan", "euclidean", "cosine"])
@pytest.mark.parametrize("N,d,k", [(300, 2, 5), (1000, 50, 16), (2500, 50, 31), (2000, 7, 31), (1500, 64, 51), (700, 128, 33),
                                   (900, 20, 100), (257, 3, 65), (4000, 50, 31), (9000, 10, 31)])
def test_pruned_search_gives_the_same_bits(metric, N, d, k, monkeypatch):
    """The pruned form (reordered points, tile bounds, best-first order with early exit; taken by default from 20 k
    points) forced on small inputs: the answer must not change by a bit."""
    monkeypatch.setenv("GFICF_KNN_PRUNE", "1")
    X = blobs(N, d, seed=N + d + k)
    got = gficf_amd.find_nn(X, k, True, metric)
    widx, wdist = oracle.knn(X, k, metric, nthreads=8)
    assert np.array_equal(got["idx"], widx)
    assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32))


@pytest.mark.parametrize("metric,k", [("manhattan", 31), ("euclidean", 16), ("cosine", 51)])
def test_default_path_at_size_clustered_and_not(metric, k):
    """60 k points: the pruned form is the default here (and falls back to the plain one on the device when the data
    has nothing to prune).  Blocks of queries against the oracle, clustered and unclustered data."""
    N, d = 60000, 30
    rng = np.random.default_rng(5)
    for name, X in (("blobs", blobs(N, d, seed=77, centers=25)), ("normal", rng.normal(size=(N, d)))):
        got = gficf_amd.find_nn(X, k, True, metric)
        for qb, qe in ((0, 700), (31000, 31500), (N - 300, N)):
            widx, wdist = oracle.knn(X, k, metric, nthreads=16, queries=(qb, qe))
            assert np.array_equal(got["idx"][qb:qe], widx[qb:qe]), (name, qb)
            assert np.array_equal(got["dist"][qb:qe].astype(np.float32), wdist[qb:qe].astype(np.float32)), (name, qb)


def test_pruned_search_on_unclustered_and_degenerate_data(monkeypatch):
    monkeypatch.setenv("GFICF_KNN_PRUNE", "1")
    rng = np.random.default_rng(8)
    cases = {"uniform": rng.uniform(size=(3000, 6)), "all equal": np.ones((600, 4)), "two points repeated": np.repeat(np.array([[0.0, 0.0], [5.0, 1.0]]), 400, axis=0),
             "line": np.linspace(0, 1, 2000)[:, None] * np.ones((1, 3))}
    for name, X in cases.items():
        for metric in ("manhattan", "euclidean"):
            got = gficf_amd.find_nn(X, 17, True, metric)
            widx, wdist = oracle.knn(X, 17, metric, nthreads=8)
            assert np.array_equal(got["idx"], widx), (name, metric)
            assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32)), (name, metric)


@pytest.mark.parametrize("metric", ["manhattan", "euclidean", "cosine"])
def test_against_float64_restatement(metric):
    X = blobs(600, 30, seed=5)
    got = gficf_amd.find_nn(X, 20, True, metric)
    nidx, ndist = oracle_np.knn_np(X, 20, metric)
    assert np.allclose(got["dist"], ndist, rtol=1e-5, atol=1e-5)
    # ids agree wherever the float64 distances are not within rounding of the next one
    agree = got["idx"] == nidx
    assert agree.mean() > 0.995


def test_duplicates_and_ties_break_by_index():
    rng = np.random.default_rng(3)
    base = rng.integers(0, 4, size=(40, 3)).astype(np.float64)       # a small grid: many exact ties and duplicates
    X = np.concatenate([base, base, base[:17]])
    for metric in ("manhattan", "euclidean"):
        got = gficf_amd.find_nn(X, 12, True, metric)
        widx, wdist = oracle.knn(X, 12, metric)
        assert np.array_equal(got["idx"], widx)
        assert np.array_equal(got["dist"], wdist)
        # a duplicated point's nearest is the copy with the smallest index, at distance 0
        assert np.all(got["dist"][:, 0] == 0.0)
        assert np.all(got["idx"][40:80, 0] <= np.arange(1, 41))


@pytest.mark.parametrize("N,d,k", [(5, 1, 5), (64, 1, 64), (100, 3, 100), (127, 50, 33), (200, 5, 64), (257, 2, 65)])
def test_small_and_degenerate_shapes(N, d, k):
    # fewer points than one tile, one dimension, every point a neighbour (k = N), list lengths at the 32 / 64 seams
    X = blobs(N, d, seed=N * 7 + k, centers=3)
    for metric in ("manhattan", "euclidean", "cosine"):
        got = gficf_amd.find_nn(X, k, True, metric)
        widx, wdist = oracle.knn(X, k, metric)
        assert np.array_equal(got["idx"], widx), metric
        assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32)), metric
    if k == N:
        assert all(sorted(r) == list(range(1, N + 1)) for r in gficf_amd.find_nn(X, k)["idx"].tolist())


def test_float32_input_and_zero_rows_cosine(ops):
    import torch

    N, d, k = 900, 12, 10
    X = blobs(N, d, seed=4).astype(np.float32)
    X[17] = 0.0                                                     # a zero vector: cosine distance 1 to everything
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()         # f32 column-major block
    pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
    ops.knn_prepare(Xd, N, d, "cosine", pts)
    ws = torch.zeros(ops.knn_workspace_bytes(N, N, k), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((k, N), dtype=torch.int32, device="cuda")
    dist = torch.zeros((k, N), dtype=torch.float32, device="cuda")
    ops.knn_search(pts, N, d, k, "cosine", 0, N, ws, idx, dist)
    ops.sync()
    widx, wdist = oracle.knn(X.astype(np.float64), k, "cosine", nthreads=4)
    assert np.array_equal(idx.cpu().numpy().T, widx)
    assert np.array_equal(dist.cpu().numpy().T, wdist.astype(np.float32))
    assert np.all(dist.cpu().numpy().T[17] == 1.0)


@pytest.mark.parametrize("N,d,k", [(1200, 50, 16), (3000, 9, 31)])
def test_correlation_metric(N, d, k, monkeypatch):
    """uwot's "correlation": cosine distance of the rows with their mean removed — plain and pruned form."""
    X = blobs(N, d, seed=N + k) + 5.0                               # an offset that the centring has to remove
    widx, wdist = oracle.knn(X, k, "correlation", nthreads=8)
    nidx, ndist = oracle_np.knn_np(X, k, "correlation")
    assert np.allclose(wdist, ndist, rtol=1e-4, atol=2e-6) and (widx == nidx).mean() > 0.99
    for prune in ("0", "1"):
        monkeypatch.setenv("GFICF_KNN_PRUNE", prune)
        got = gficf_amd.find_nn(X, k, True, "correlation")
        assert np.array_equal(got["idx"], widx)
        assert np.array_equal(got["dist"].astype(np.float32), wdist.astype(np.float32))


def test_include_self_false_drops_own_id():
    X = blobs(500, 10, seed=9)
    a = gficf_amd.find_nn(X, 11, True, "manhattan")
    b = gficf_amd.find_nn(X, 10, False, "manhattan")
    assert np.array_equal(a["idx"][:, 1:], b["idx"]) and np.array_equal(a["dist"][:, 1:], b["dist"])


def test_argument_errors():
    X = blobs(50, 4, seed=1)
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.find_nn(X, 51, True, "manhattan")            # more neighbours than points
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.find_nn(blobs(50, 129, seed=1), 5, True, "manhattan")
    with pytest.raises(ValueError):
        gficf_amd.find_nn(X, 5, True, "hamming")
    for poison in (np.nan, np.inf, 1e300):                          # not representable as a finite f32
        Y = X.copy()
        Y[7, 2] = poison
        with pytest.raises(gficf_amd.GficfError) as ei:
            gficf_amd.find_nn(Y, 5, True, "manhattan")
        assert ei.value.status == "GFICF_ERR_BAD_VALUE"
    assert gficf_amd.find_nn(X, 5, True, "manhattan")["idx"].shape == (50, 5)     # the context is usable afterwards


def test_device_query_blocks_and_split_seams(ops, monkeypatch):
    """Queries in two blocks (the multi-GPU seam) and a forced 5-way candidate split give the same bits."""
    import torch

    N, d, k = 3000, 50, 31
    X = blobs(N, d, seed=11)
    widx, _ = oracle.knn(X, k, "manhattan", nthreads=8)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()                     # (d, N) == column-major N x d
    pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
    # prepare in two row blocks, as two ranks would before their all-gather
    h = 1700
    ops.knn_prepare(Xd[:, :h].contiguous(), h, d, "manhattan", pts[:h])
    ops.knn_prepare(Xd[:, h:].contiguous(), N - h, d, "manhattan", pts[h:])
    for split in (None, "5", "prune"):
        if split == "prune":
            monkeypatch.delenv("GFICF_KNN_SPLIT")
            monkeypatch.setenv("GFICF_KNN_PRUNE", "1")
        elif split:
            monkeypatch.setenv("GFICF_KNN_SPLIT", split)
        out = []
        for b, e in ((0, 1300), (1300, N)):
            ws = torch.zeros(ops.knn_workspace_bytes(e - b, N, k), dtype=torch.uint8, device="cuda")
            idx = torch.full((k, e - b), -7, dtype=torch.int32, device="cuda")
            dist = torch.full((k, e - b), -7.0, dtype=torch.float32, device="cuda")
            ops.knn_search(pts, N, d, k, "manhattan", b, e, ws, idx, dist)
            ops.sync()
            out.append(idx.cpu().numpy().T)
        assert np.array_equal(np.concatenate(out), widx)


def test_knn_feeds_jaccard_on_device(ops):
    """find_nn -> neigh[,-1] -> Jaccard without leaving the device: the index matrix the search writes
    (column-major, 1-based) is what the ingest reads, skipping the self column."""
    import torch

    N, d, k = 2000, 20, 15
    X = blobs(N, d, seed=21)
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    pts = torch.zeros((N, ops.knn_dpad(d)), dtype=torch.float32, device="cuda")
    ops.knn_prepare(Xd, N, d, "manhattan", pts)
    ws = torch.zeros(ops.knn_workspace_bytes(N, N, k + 1), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((k + 1, N), dtype=torch.int32, device="cuda")
    ops.knn_search(pts, N, d, k + 1, "manhattan", 0, N, ws, idx, None)
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    rmat = torch.zeros((3, N * k), dtype=torch.float64, device="cuda")
    ops.jaccard(idx[1:], N, k, table, rmat, None)                               # neigh[,-1]  (R/clustCells.R:63)
    ops.sync()
    widx, _ = oracle.knn(X, k + 1, "manhattan", nthreads=8)
    want, _ = oracle.jaccard(np.ascontiguousarray(widx[:, 1:]), nthreads=8)
    assert np.array_equal(rmat.cpu().numpy().T, want)
    # and the one-call host form of the same lines
    edges = gficf_amd.clustcells_graph(X, k, "manhattan")
    keep = want[:, 2] > 0
    assert np.array_equal(edges["from"], want[keep, 0]) and np.array_equal(edges["to"], want[keep, 1])
    assert np.array_equal(edges["weight"], want[keep, 2])


@pytest.mark.parametrize("c,m,d", [(12, 9, 3), (300, 31, 6), (1500, 31, 50)])
def test_phenograph_against_a_derived_answer_on_far_apart_groups(c, m, d):
    """The whole fused call — search, Jaccard, filter, adjacency, Louvain — on an input whose answer is derived, no oracle in the loop: c groups of
    m points, the groups 1000 apart, the points of a group a small step apart, k = m - 1.  The k nearest of a point are exactly the rest of its group
    (whatever the ties inside), so two points of a group share m - 2 neighbours: u = m - 2, weight (m - 2) / m, every pair mutual — the graph is c
    disjoint cliques, n_edges = c m (m - 1), the communities are the groups and Q = 1 - resolution / c (reference chain R/clustCells.R:57-86)."""
    rng = np.random.default_rng(c)
    g = np.repeat(np.arange(c), m)
    X = np.zeros((c * m, d))
    X[:, 0] = 1000.0 * g + 0.01 * np.tile(np.arange(m), c)
    X[:, 1:] = 0.001 * rng.random((c * m, d - 1))
    perm = rng.permutation(c * m)                       # the caller's order carries no hint
    lab = gficf_amd.phenograph(X[perm], k=m - 1, dist_method="manhattan", resolution=0.8, n_start=3, n_iter=4, random_seed=5)
    assert lab.n_edges == c * m * (m - 1) and lab.n_clusters == c
    assert abs(lab.modularity - (1.0 - 0.8 / c)) < 1e-9
    pairs = np.unique(np.stack([np.asarray(lab), g[perm]], axis=1), axis=0)
    assert len(pairs) == c


def test_phenograph_on_cells_renumbered_in_pivot_order_gives_the_same_graph_and_labels(monkeypatch):
    """gficf_phenograph_host with GFICF_PHENOGRAPH_ORDER=1 (round 5's default from 2^17 cells on; off by default since round 6, where the
    stage times showed it costs more than it saves): the Jaccard stage runs on cells renumbered in the search's pivot order and hands the
    kept edges over in the ORIGINAL ids, grouped by source but not ascending — the same neighbour lists, the same graph, the same labels
    as the chain on the caller's order, below and above 2^17 cells (reference chain R/clustCells.R:57-68)."""
    rng = np.random.default_rng(21)
    for N in (60_000, 140_000):
        centers = rng.normal(scale=8.0, size=(40, 8))
        X = centers[rng.integers(0, 40, size=N)] + rng.normal(size=(N, 8))
        monkeypatch.delenv("GFICF_PHENOGRAPH_ORDER", raising=False)
        a = gficf_amd.phenograph(X, k=15, n_start=1, n_iter=2)
        monkeypatch.setenv("GFICF_PHENOGRAPH_ORDER", "1")
        b = gficf_amd.phenograph(X, k=15, n_start=1, n_iter=2)
        assert a.n_edges == b.n_edges > 0 and a.modularity == b.modularity and a.n_clusters == b.n_clusters
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("N,k,metric", [(3000, 7, "manhattan"), (150_000, 31, "manhattan"), (20_000, 16, "euclidean")])
def test_exact_search_against_the_derived_answer_on_a_line(N, k, metric):
    """An answer by counting, no oracle in the loop: points at the integer positions 0 .. N-1 of a line (a second, constant coordinate
    so that d > 1).  The k nearest of point i, itself included, ties to the smaller index: i, i-1, i+1, i-2, i+2, ... — clipped at the
    ends, where the list continues on the side that exists.  Integers below 2^24 are exact in f32, so are the distances.  150 000 points
    take the pruned search (pivot cells, tile bounds)."""
    X = np.stack([np.arange(N, dtype=np.float64), np.full(N, 3.0)], axis=1)
    res = gficf_amd.find_nn(X, k, True, metric)
    i = np.arange(N, dtype=np.int64)[:, None]
    cand = i + np.arange(-k, k + 1, dtype=np.int64)[None, :]                    # the k nearest lie within +-k positions
    ok = (cand >= 0) & (cand < N)
    dist = np.where(ok, np.abs(cand - i), 10 * N)
    order = np.lexsort((np.where(ok, cand, 10 * N), dist), axis=1)[:, :k]       # by (distance, index)
    want = np.take_along_axis(cand, order, axis=1) + 1
    assert np.array_equal(res["idx"], want.astype(np.int32))
    assert np.array_equal(res["dist"], np.take_along_axis(dist, order, axis=1).astype(np.float64))
