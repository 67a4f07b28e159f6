// phenograph.hip — the whole of clustcells() in one call, nothing crossing PCIe between its steps.
//
//   neigh = uwot:::find_nn(X, k + 1, include_self = T, metric)$idx; neigh = neigh[,-1]      (R/clustCells.R:57-63)
//   relations = rcpp_parallel_jaccard_coef(neigh); relations[relations[,3] > 0, ]            (:65-66)
//   g = graph.data.frame(relations, directed = FALSE); as_adjacency_matrix(g, attr = "weight")  (:69,80)
//   community = RunModularityClustering(adjacency, 1, resolution, algorithm, n.start, n.iter, seed)   (:80,86)
// as the chain gficf_knn_* -> gficf_jaccard_ingest/edges_filtered -> gficf_adjacency -> gficf_louvain on device buffers:
// one upload of the N x d point matrix, one download of N labels.  Every stage keeps its own contract (exact search in
// place of Annoy, bit-exact Jaccard edges, relaxed contract of the Louvain stage: see include/gficf_hip.h).
#include "common.h"

namespace {

struct DevBufs {                 // everything this call allocates, released on every way out
  void* p[16];
  int n = 0;
  hipError_t get(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
  ~DevBufs() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

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
  DevBufs bufs;
  void *d_X = nullptr, *d_P = nullptr, *d_kws = nullptr, *d_idx = nullptr, *d_table = nullptr, *d_u = nullptr, *d_cptr = nullptr, *d_e3 = nullptr;
  void *d_aws = nullptr, *d_indptr = nullptr, *d_indices = nullptr, *d_ax = nullptr, *d_lab = nullptr, *d_lws = nullptr;
  hipError_t e = bufs.get(&d_X, sizeof(double) * (size_t)ld * (size_t)d);
  if (e == hipSuccess) e = bufs.get(&d_P, sizeof(float) * (size_t)N * (size_t)dpad);
  if (e == hipSuccess) e = bufs.get(&d_kws, knn_ws);
  if (e == hipSuccess) e = bufs.get(&d_idx, sizeof(int32_t) * (size_t)N * (size_t)kk);
  if (e == hipSuccess) e = bufs.get(&d_table, sizeof(int32_t) * (size_t)N * (size_t)kpad);
  if (e == hipSuccess) e = bufs.get(&d_u, sizeof(uint16_t) * (size_t)cap);
  if (e == hipSuccess) e = bufs.get(&d_cptr, sizeof(int64_t) * ((size_t)N + 1));
  if (e == hipSuccess) e = bufs.get(&d_e3, sizeof(double) * 3 * (size_t)cap);
  if (e == hipSuccess) e = bufs.get(&d_aws, adj_ws);
  if (e == hipSuccess) e = bufs.get(&d_indptr, sizeof(int64_t) * ((size_t)N + 1));
  if (e == hipSuccess) e = bufs.get(&d_indices, sizeof(int32_t) * 2 * (size_t)cap);
  if (e == hipSuccess) e = bufs.get(&d_ax, sizeof(double) * 2 * (size_t)cap);
  if (e == hipSuccess) e = bufs.get(&d_lab, sizeof(int32_t) * (size_t)N);
  if (e == hipSuccess) e = hipMemcpyAsync(d_X, X, sizeof(double) * (size_t)ld * (size_t)d, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_phenograph_host: %s", hipGetErrorString(e));

  // neighbours (column 0 = the cell itself is dropped by starting at column 1), edges with weight > 0, adjacency matrix
  double* from = (double*)d_e3;
  int rc = gficf_knn_prepare_device(ctx, d_X, 1, N, d, ld, metric, (float*)d_P);
  if (!rc) rc = gficf_knn_search_device(ctx, (const float*)d_P, N, d, kk, metric, 0, N, d_kws, knn_ws, (int32_t*)d_idx, nullptr, N);
  if (!rc) rc = gficf_jaccard_ingest_device(ctx, (const int32_t*)d_idx + N, 0, N, k, N, N, (int32_t*)d_table);
  if (!rc) rc = gficf_jaccard_edges_filtered_device(ctx, (const int32_t*)d_table, N, k, 0, N, (uint16_t*)d_u, (int64_t*)d_cptr, from, from + cap,
                                                    from + 2 * cap);
  if (!rc) rc = gficf_adjacency_device(ctx, N, cap, (const int64_t*)d_cptr + N, from, from + cap, from + 2 * cap, d_aws, adj_ws,
                                       (int64_t*)d_indptr, (int32_t*)d_indices, (double*)d_ax);
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
  const size_t lws = gficf_louvain_workspace_bytes(N, h_cnt[1]);
  e = bufs.get(&d_lws, lws);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_phenograph_host: %s", hipGetErrorString(e));
  rc = gficf_louvain_device(ctx, N, (const int64_t*)d_indptr, (const int32_t*)d_indices, (const double*)d_ax, h_cnt[1], resolution, algorithm,
                            n_start, n_iter, seed, (int32_t*)d_lab, n_clusters, modularity, d_lws, lws);
  if (rc) return rc;
  GFICF_HIP_CHECK(hipMemcpyAsync(labels, d_lab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
  return gficf_ctx_sync(ctx);
}
