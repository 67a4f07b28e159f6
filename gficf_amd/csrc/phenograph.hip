// phenograph.hip — the whole of clustcells() in one call, nothing crossing PCIe between its steps.
//
//   neigh = uwot:::find_nn(X, k + 1, include_self = T, metric)$idx; neigh = neigh[,-1]      (R/clustCells.R:57-63)
//   relations = rcpp_parallel_jaccard_coef(neigh); relations[relations[,3] > 0, ]            (:65-66)
//   g = graph.data.frame(relations, directed = FALSE); as_adjacency_matrix(g, attr = "weight")  (:69,80)
//   community = RunModularityClustering(adjacency, 1, resolution, algorithm, n.start, n.iter, seed)   (:80,86)
// as the chain gficf_knn_* -> gficf_jaccard_ingest/edges_filtered -> gficf_adjacency -> gficf_louvain on device buffers:
// one upload of the N x d point matrix, one download of N labels.  Every stage keeps its own contract (exact search in
// place of Annoy, bit-exact Jaccard edges, relaxed contract of the Louvain stage: see include/gficf_hip.h).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace {

// One grow-only block of the context's pool carved into the call's buffers (no allocation once the pool has grown).
struct Carver {
  char* base = nullptr;
  size_t off = 0;
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

// ---- the Jaccard stage on cells renumbered by locality (round 5; OFF by default since round 6, GFICF_PHENOGRAPH_ORDER=1 turns it on).
// Measured stage by stage in round 6 (profiles/r06_phenograph_order_ab.txt): the renumbering needs a pass of its own over the points
// (every cell's nearest pivot: 10 ms at 1 M x 50) to save 0.4 ms in the edge kernel, and it hands the adjacency build sources that do
// not ascend (its slower form) — 12.4 against 0.94 ms for the Jaccard stage at 1 M cells, 0.92 against 0.18 at 200 k.  Round 5 read a
// first-call effect (the pool growing) as its gain.  What it does:  from 2^17 cells on the table takes 128 B rows and, with ids in
// the order the caller's matrix happens to have, nearly every gathered row is an L2 miss (1 M x 30: 697 us, 5.5 x the algorithmic
// bytes).  The search has just computed an order in which neighbours sit next to each other — its (coarse, fine) pivot order — so
// the index matrix is relabelled into that numbering (row p = original cell order[p], ids through the inverse), the edge kernel
// walks cells whose rows its XCD's L2 already holds, and the kept edges come out in the ORIGINAL ids
// (gficf_jaccard_edges_filtered_mapped); the adjacency build sorts them anyway.  The neighbour lists are the plain
// search's own (the relabelling happens behind it): the graph, and so the labels, are those of the unordered chain for every input.
__global__ __launch_bounds__(256) void k_invert_order(const int32_t* __restrict__ order, int32_t* __restrict__ inv, int64_t N) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < N) inv[order[p]] = (int32_t)p;
}

// out (k x N column-major, new numbering) from idx (kk x N column-major, original ids, column 0 = the cell itself: dropped)
__global__ __launch_bounds__(256) void k_relabel_idx(const int32_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ inv, int32_t* __restrict__ out, int64_t N, int k) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int64_t c = order[p];
  for (int j = 0; j < k; ++j) {
    const int32_t id = idx[(int64_t)(j + 1) * N + c];
    out[(int64_t)j * N + p] = (id >= 1 && id <= N) ? inv[id - 1] + 1 : id;      // (an id the search cannot produce stays what it is: the ingest reports it)
  }
}

bool phenograph_ordered(int64_t N) {
  (void)N;
  if (const char* e = getenv("GFICF_PHENOGRAPH_ORDER")) return atoi(e) != 0;    // A/B switch, read per call
  return false;
}

}  // namespace

extern "C" int gficf_phenograph_host(gficf_ctx* ctx, const double* X, int64_t N, int d, int64_t ld, int k, int metric, double resolution,
                                     int algorithm, int n_start, int n_iter, int seed, int32_t* labels, int64_t* n_clusters,
                                     double* modularity, int64_t* n_edges) {
  GFICF_CTX_ENTER(ctx);
  if (n_clusters) *n_clusters = 0;
  if (modularity) *modularity = 0.0;
  if (n_edges) *n_edges = 0;
  if (N < 0 || d <= 0 || k < 1) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size, no dimensions or k < 1");
  if (N == 0) return GFICF_OK;
  if (!X || !labels || !n_clusters) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  if (k + 1 > N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "k + 1 = %d neighbours asked of %lld cells", k + 1, (long long)N);
  if (k + 1 > GFICF_KNN_MAX_K) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k + 1 > %d", GFICF_KNN_MAX_K);
  const int dpad = gficf_knn_dpad(d), kpad = gficf_jaccard_kpad(k);
  if (dpad < 0 || kpad < 0) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "d or k beyond what the search / the edge kernel take");
  const int kk = k + 1;
  const int64_t cap = N * (int64_t)k;
  const size_t knn_ws = gficf_knn_workspace_bytes(ctx, N, N, kk), adj_ws = gficf_adjacency_workspace_bytes(N, cap);
  const bool ordered = phenograph_ordered(N);
  const size_t order_ws = ordered ? gficf_knn_workspace_bytes(ctx, N, N, 1) : 0;
  void *d_X, *d_P, *d_kws, *d_idx, *d_table, *d_u, *d_cptr, *d_e3, *d_aws, *d_indptr, *d_indices, *d_ax, *d_lab;
  void *d_order = nullptr, *d_inv = nullptr, *d_idx2 = nullptr;
  Carver cv;
  for (int pass = 0; pass < 2; ++pass) {               // pass 0 sizes the block, pass 1 hands the pointers out
    cv.off = 0;
    d_X = cv.take<double>((size_t)ld * (size_t)d);
    d_P = cv.take<float>((size_t)N * (size_t)dpad);
    d_kws = cv.take<char>(knn_ws > order_ws ? knn_ws : order_ws);
    d_idx = cv.take<int32_t>((size_t)N * (size_t)kk);
    d_table = cv.take<int32_t>((size_t)N * (size_t)kpad);
    d_u = cv.take<uint16_t>((size_t)cap);
    d_cptr = cv.take<int64_t>((size_t)N + 1);
    d_e3 = cv.take<double>(3 * (size_t)cap);
    d_aws = cv.take<char>(adj_ws);
    d_indptr = cv.take<int64_t>((size_t)N + 1);
    d_indices = cv.take<int32_t>(2 * (size_t)cap);
    d_ax = cv.take<double>(2 * (size_t)cap);
    d_lab = cv.take<int32_t>((size_t)N);
    if (ordered) {
      d_order = cv.take<int32_t>((size_t)N);
      d_inv = cv.take<int32_t>((size_t)N);
      d_idx2 = cv.take<int32_t>((size_t)N * (size_t)k);
    }
    if (pass == 0) {
      void* blk = nullptr;
      const hipError_t e0 = gficf_pool_get(ctx, 0, cv.off + 256, &blk);
      if (e0 != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_phenograph_host: %s", hipGetErrorString(e0));
      cv.base = (char*)blk;
    }
  }
  // GFICF_PHENOGRAPH_DEBUG=1: the stages timed on the host, each behind a stream synchronisation of its own (lab; stderr)
  const bool dbg = getenv("GFICF_PHENOGRAPH_DEBUG") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!dbg) return;
    (void)hipStreamSynchronize(ctx->stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[gficf_phenograph_host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  hipError_t e = hipMemcpyAsync(d_X, X, sizeof(double) * (size_t)ld * (size_t)d, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_phenograph_host: %s", hipGetErrorString(e));
  mark("upload of the points");

  // neighbours (column 0 = the cell itself is dropped by starting at column 1), edges with weight > 0, adjacency matrix
  double* from = (double*)d_e3;
  int rc = gficf_knn_prepare_device(ctx, d_X, 1, N, d, ld, metric, (float*)d_P);
  if (!rc) rc = gficf_knn_search_device(ctx, (const float*)d_P, N, d, kk, metric, 0, N, d_kws, knn_ws, (int32_t*)d_idx, nullptr, N);
  mark("neighbour search");
  if (!rc && ordered) {
    rc = gficf_knn_pivot_order_device(ctx, (const float*)d_P, N, d, metric, d_kws, knn_ws > order_ws ? knn_ws : order_ws, (int32_t*)d_order);
    if (!rc) {
      const unsigned nb = (unsigned)gficf_ceil_div(N, 256);
      hipLaunchKernelGGL(k_invert_order, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t*)d_order, (int32_t*)d_inv, N);
      hipLaunchKernelGGL(k_relabel_idx, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t*)d_idx, (const int32_t*)d_order, (const int32_t*)d_inv,
                         (int32_t*)d_idx2, N, k);
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) { gficf_set_error("HIP failure in gficf_phenograph_host (relabel launches): %s", hipGetErrorString(le)); rc = GFICF_ERR_HIP; }
    }
    if (!rc) rc = gficf_jaccard_ingest_device(ctx, (const int32_t*)d_idx2, 0, N, k, N, N, (int32_t*)d_table);
    if (!rc) rc = gficf_jaccard_edges_filtered_mapped(ctx, (const int32_t*)d_table, N, k, 0, N, (uint16_t*)d_u, (int64_t*)d_cptr, from, from + cap,
                                                             from + 2 * cap, (const int32_t*)d_order);
  } else {
    if (!rc) rc = gficf_jaccard_ingest_device(ctx, (const int32_t*)d_idx + N, 0, N, k, N, N, (int32_t*)d_table);
    if (!rc) rc = gficf_jaccard_edges_filtered_device(ctx, (const int32_t*)d_table, N, k, 0, N, (uint16_t*)d_u, (int64_t*)d_cptr, from, from + cap,
                                                      from + 2 * cap);
  }
  mark("Jaccard edges (filtered)");
  if (!rc) rc = gficf_adjacency_device(ctx, N, cap, (const int64_t*)d_cptr + N, from, from + cap, from + 2 * cap, 1, d_aws, adj_ws,
                                       (int64_t*)d_indptr, (int32_t*)d_indices, (double*)d_ax);
  mark("adjacency matrix");
  int64_t h_cnt[2] = {0, 0};                      // kept edges, adjacency entries
  if (!rc) {
    GFICF_HIP_CHECK(hipMemcpyAsync(&h_cnt[0], (const int64_t*)d_cptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    GFICF_HIP_CHECK(hipMemcpyAsync(&h_cnt[1], (const int64_t*)d_indptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    rc = gficf_ctx_sync(ctx);                     // also surfaces non-finite coordinates before the graph is used
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  if (rc) return rc;
  if (n_edges) *n_edges = h_cnt[0];

  // communities
  const size_t lws = gficf_louvain_workspace_bytes(N, h_cnt[1], n_start);      // all the starts in one launch set
  void* d_lws = nullptr;
  e = gficf_pool_get(ctx, 1, lws, &d_lws);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_phenograph_host: %s", hipGetErrorString(e));
  rc = gficf_louvain_device(ctx, N, (const int64_t*)d_indptr, (const int32_t*)d_indices, (const double*)d_ax, h_cnt[1], resolution, algorithm,
                            n_start, n_iter, seed, (int32_t*)d_lab, n_clusters, modularity, d_lws, lws);
  if (rc) return rc;
  mark("communities");
  GFICF_HIP_CHECK(hipMemcpyAsync(labels, d_lab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
  rc = gficf_ctx_sync(ctx);
  mark("labels to the host");
  return rc;
}
