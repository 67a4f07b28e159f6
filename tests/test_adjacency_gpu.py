"""GPU parity of the graph hand-off (second half of "next" row N1): kept Jaccard edges -> symmetric weighted
adjacency matrix, as igraph::as_adjacency_matrix(graph.data.frame(relations, directed = FALSE), attr = "weight")
builds it for the modularity optimiser (reference R/clustCells.R:69,80,86).  Checker: scipy sparse algebra
(A = W + W^T, duplicates summed, a self edge once).  Structure exact; values exact (sums of at most a few equal
doubles)."""
import numpy as np
import pytest
import scipy.sparse as sp

import gficf_amd
import oracle
from gficf_amd import synth

pytestmark = pytest.mark.gpu


def reference_adjacency(f, t, w, N):
    i, j = f.astype(np.int64) - 1, t.astype(np.int64) - 1
    W = sp.coo_matrix((w, (i, j)), shape=(N, N)).tocsc()
    A = (W + W.T - sp.diags(W.diagonal())).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return A


def same(A, B):
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(A.data, B.data))


@pytest.mark.parametrize("N,E,seed", [(10, 0, 0), (1, 1, 1), (50, 400, 2), (1000, 30000, 3), (70000, 200000, 4)])
def test_random_edge_lists(N, E, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(1, N + 1, size=E).astype(np.float64)            # duplicates, self edges and mutual pairs all occur
    t = rng.integers(1, N + 1, size=E).astype(np.float64)
    w = rng.integers(1, 60, size=E) / 64.0                            # exactly representable: sums are exact in any order
    A = gficf_amd.jaccard_adjacency({"from": f, "to": t, "weight": w}, N)
    assert same(A, reference_adjacency(f, t, w, N))
    assert (abs(A - A.T)).nnz == 0


@pytest.mark.parametrize("N,E,grouped", [(3_000_000, 200_000, False), (3_000_000, 300_000, True), (150_000, 4_600_000, True), (150_000, 4_400_000, False),
                                         (1023, 9_000, True), (1024, 9_000, True), (5, 40_000, False)])
def test_every_form_of_the_sort(N, E, grouped):
    """The build orders the transposed half with its own radix passes (digits of up to 10 bits over the bits of N, least significant first): one
    pass up to 1023 cells, two up to 2^20 - 1, three beyond (3 M cells: 22 bits); from 4.2 M edges on a workgroup walks more than one tile and
    carries the digits' places along.  Random lists, exactly summable weights, against scipy."""
    rng = np.random.default_rng(N + E)
    f = rng.integers(1, N + 1, size=E)
    if grouped:
        f = np.sort(f)
    t = rng.integers(1, N + 1, size=E)
    if N > 1_000_000:                                                     # hubs too: long rows next to the three-pass sort
        t[: E // 50] = 7
    f, t = f.astype(np.float64), t.astype(np.float64)
    w = rng.integers(1, 60, size=E) / 64.0
    A = gficf_amd.jaccard_adjacency({"from": f, "to": t, "weight": w}, N)
    assert same(A, reference_adjacency(f, t, w, N))


@pytest.mark.parametrize("grouped", [False, "any order", "ascending"])
@pytest.mark.parametrize("hub_degrees", [(129, 300), (4096, 4097), (5000, 20000), (60000,)])
def test_rows_of_every_length_class(hub_degrees, grouped):
    """Round 6: the matrix is built row by row — a wave merges the two halves of a row of up to 128 entries by ranking, a workgroup one of up
    to 4096 in LDS, longer ones in global scratch; only the transposed half goes through a (stable, 32-bit) sort.  Hubs of every class next
    to ordinary rows and REPEATED edges whose weights are NOT exactly summable: a run of one column is added up in emission order (what the
    stable sort of the old path gave), so the sums must match a sequential addition in edge order bit for bit.  grouped = False: edges in
    random order (the build finds out and orders them by source as well); "any order": the edges of one source lie together, the sources in
    random order (what the edge kernel writes from renumbered cells) — half W needs no ordering; "ascending": the sources ascend as well (what
    the edge kernel writes otherwise) — the long rows are then MERGED from their two halves instead of ordered whole."""
    rng = np.random.default_rng(len(hub_degrees) * 1000 + hub_degrees[0])
    N = 70000
    f, t = [], []
    for h, deg in enumerate(hub_degrees):
        leaves = rng.choice(np.arange(len(hub_degrees), N), deg, replace=False)
        out = rng.random(deg) < 0.5                                        # some edges leave the hub, some arrive
        f += [np.where(out, h, leaves)]; t += [np.where(out, leaves, h)]
    bg = 150000
    f += [rng.integers(0, N, bg)]; t += [rng.integers(0, N, bg)]
    f, t = np.concatenate(f), np.concatenate(t)
    rep = rng.integers(0, len(f), 4000)                                    # repeated edges (some reversed): runs of 2, 3, 4 ... of one column
    f, t = np.concatenate([f, f[rep], t[rep[:1500]]]), np.concatenate([t, t[rep], f[rep[:1500]]])
    perm = rng.permutation(len(f))
    f, t = f[perm], t[perm]
    if grouped:
        by_source = np.argsort(rng.permutation(N)[f] if grouped == "any order" else f, kind="stable")
        f, t = f[by_source], t[by_source]
        assert len(np.flatnonzero(np.diff(f))) + 1 == len(np.unique(f))   # one run per source
    w = rng.random(len(f)) * 0.9 + 0.05                                    # arbitrary doubles
    A = gficf_amd.jaccard_adjacency({"from": (f + 1).astype(np.float64), "to": (t + 1).astype(np.float64), "weight": w}, N)
    # the checker: entries (row, col, emission position) sorted, runs summed sequentially in that order
    r = np.concatenate([f, t[f != t]]); c = np.concatenate([t, f[f != t]])
    pos = np.concatenate([2 * np.arange(len(f)), 2 * np.flatnonzero(f != t) + 1]); ww = np.concatenate([w, w[f != t]])
    order = np.lexsort((pos, r, c))                                        # CSC: by column, then row, then position (the matrix is symmetric)
    r, c, ww = r[order], c[order], ww[order]
    head = np.ones(len(r), dtype=bool); head[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    starts = np.flatnonzero(head)
    sums = np.empty(len(starts))
    ends = np.append(starts[1:], len(r))
    for q, (a, b) in enumerate(zip(starts, ends)):                         # (sequential addition; np.add.reduceat pairs differently)
        acc = ww[a]
        for z in range(a + 1, b):
            acc = acc + ww[z]
        sums[q] = acc
    want = sp.csc_matrix((sums, (r[head], c[head])), shape=(N, N))
    want.sort_indices()
    assert same(A, want)
    assert np.diff(A.indptr).max() >= max(hub_degrees)


def test_clustcells_graph_to_adjacency():
    N, k = 4000, 15
    mat = synth.knn_windowed(N, k)
    neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], mat], axis=1)
    edges = gficf_amd.jaccard_edges(neigh)
    A = gficf_amd.jaccard_adjacency(edges, N)
    assert same(A, reference_adjacency(edges["from"], edges["to"], edges["weight"], N))
    # u is symmetric, so a mutual pair carries exactly twice the edge weight
    want, _ = oracle.jaccard(mat, nthreads=4)
    want = want[want[:, 2] > 0]
    W = sp.coo_matrix((want[:, 2], (want[:, 0].astype(int) - 1, want[:, 1].astype(int) - 1)), shape=(N, N)).tocsr()
    i, j = 0, int(mat[0, 0]) - 1
    mutual = (i + 1) in mat[j]
    assert A[i, j] == W[i, j] * (2 if mutual else 1)


@pytest.mark.parametrize("promise", [True, False])
def test_device_chain_with_device_side_edge_count(promise):
    """Filtered edge build -> adjacency without a host round trip: the edge count stays on the device (promise: the caller says that the
    list is grouped by source, which the filtered build's is — no check, no synchronisation inside the call)."""
    import torch

    ops = gficf_amd.HipOps(0)
    N, k = 3000, 30
    mat = synth.knn_windowed(N, k, seed=9)
    idx = torch.from_numpy(np.ascontiguousarray(mat.T)).cuda()
    table = torch.empty((N, ops.kpad(k)), dtype=torch.int32, device="cuda")
    ops.jaccard_ingest(idx, N, k, N, table)
    cap = N * k
    u_ws = torch.zeros(cap, dtype=torch.int16, device="cuda")
    cell_ptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    out3 = torch.zeros((3, cap), dtype=torch.float64, device="cuda")
    ops.jaccard_edges_filtered(table, N, k, 0, N, u_ws, cell_ptr, out3)
    ws = torch.zeros(ops.adjacency_workspace_bytes(N, cap), dtype=torch.uint8, device="cuda")
    indptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    indices = torch.zeros(2 * cap, dtype=torch.int32, device="cuda")
    x = torch.zeros(2 * cap, dtype=torch.float64, device="cuda")
    ops.adjacency(N, cap, cell_ptr[N:N + 1], out3, ws, indptr, indices, x, grouped_by_source=promise)
    ops.sync()
    n = int(cell_ptr[N])
    f, t, w = (out3[r, :n].cpu().numpy() for r in range(3))
    nnz = int(indptr[N])
    A = sp.csc_matrix((x[:nnz].cpu().numpy(), indices[:nnz].cpu().numpy(), indptr.cpu().numpy()), shape=(N, N))
    assert same(A, reference_adjacency(f, t, w, N))


def test_a_broken_promise_is_reported_not_computed():
    """grouped_by_source = 1 on a list that is not grouped: GFICF_ERR_INVALID_ARG at the next sync (and nothing written out of bounds: the
    rows come out empty); the same buffers with the promise dropped give the matrix."""
    import torch

    ops = gficf_amd.HipOps(0)
    N, E = 500, 6000
    rng = np.random.default_rng(5)
    f = rng.integers(1, N + 1, E).astype(np.float64); t = rng.integers(1, N + 1, E).astype(np.float64); w = rng.integers(1, 60, E) / 64.0
    out3 = torch.from_numpy(np.stack([f, t, w])).cuda()
    ws = torch.zeros(ops.adjacency_workspace_bytes(N, E), dtype=torch.uint8, device="cuda")
    indptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    indices = torch.zeros(2 * E, dtype=torch.int32, device="cuda")
    x = torch.zeros(2 * E, dtype=torch.float64, device="cuda")
    ops.adjacency(N, E, None, out3, ws, indptr, indices, x, grouped_by_source=True)
    with pytest.raises(gficf_amd.GficfError) as ei:
        ops.sync()
    assert ei.value.status == "GFICF_ERR_INVALID_ARG" and "grouped" in str(ei.value)
    assert int(indptr[N]) == 0
    ops.adjacency(N, E, None, out3, ws, indptr, indices, x)
    ops.sync()
    nnz = int(indptr[N])
    A = sp.csc_matrix((x[:nnz].cpu().numpy(), indices[:nnz].cpu().numpy(), indptr.cpu().numpy()), shape=(N, N))
    assert same(A, reference_adjacency(f, t, w, N))


def test_bad_ids_are_rejected():
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.jaccard_adjacency({"from": np.array([1.0, 7.0]), "to": np.array([2.0, 1.0]), "weight": np.array([0.5, 0.5])}, 5)
    assert ei.value.status == "GFICF_ERR_BAD_ID"


@pytest.mark.parametrize("N,k", [(500, 15), (40_000, 30)])
def test_graph_of_a_ring_against_the_derived_adjacency_matrix(N, k):
    """The whole hand-off on an input whose answer is derived by counting (tests/helpers/closed_form.py): row i names the next k cells of a
    ring, so the edge i -> i + d (slot d - 1) has u = k - d and weight (k - d) / (k + d); nothing points backwards, so the symmetric
    adjacency matrix holds A[i, i +- d] = (k - d) / (k + d) for d = 1 .. k - 1 and nothing else — 2 (k - 1) entries per column.
    Edge filter (the u = 0 rows of slot k - 1 dropped, R/clustCells.R:66) and adjacency build (R/clustCells.R:69,80), no scipy algebra
    and no oracle in the loop."""
    from tests.helpers.closed_form import cyclic_window_matrix

    mat = cyclic_window_matrix(N, k)
    neigh = np.concatenate([np.arange(1, N + 1, dtype=np.int32)[:, None], mat], axis=1)
    edges = gficf_amd.jaccard_edges(neigh)
    assert len(edges["weight"]) == N * (k - 1)
    A = gficf_amd.jaccard_adjacency(edges, N)
    d = np.arange(1, k, dtype=np.int64)
    wd = (k - d) / (k + d)                                                       # u / (2 k - u) with u = k - d
    cols = np.repeat(np.arange(N, dtype=np.int64), 2 * (k - 1))
    offs = np.tile(np.concatenate([-d, d]), N)
    want = sp.csc_matrix((np.tile(np.concatenate([wd, wd]), N), ((cols + offs) % N, cols)), shape=(N, N))
    want.sort_indices()
    assert same(A, want)
