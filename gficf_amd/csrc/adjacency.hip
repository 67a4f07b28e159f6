// adjacency.hip — the Jaccard edge list as the symmetric weighted adjacency matrix (second half of "next" row N1).
//
// What clustcells() builds from the kept edges before community detection,
//   g <- igraph::graph.data.frame(relations, directed = FALSE)                       (reference R/clustCells.R:69)
//   igraph::as_adjacency_matrix(g, attr = "weight", sparse = T)                       (reference R/clustCells.R:80,86)
// i.e. A[i,j] = A[j,i] = sum of the weights of all edges between i and j (an i -> j and a j -> i edge are two
// edges of the undirected multigraph, so a mutual pair carries 2w), as a sparse matrix with sorted indices — the
// input of RunModularityClustering (src/RModularityOptimizer.cpp).  igraph is third-party (unpinned); the
// restatement is "A = W + W^T over the directed edge list W, duplicates summed, a self edge counted once".
// Vertices are the cells 1..N in cell order (igraph orders vertices by first appearance in the edge list, which is
// the same whenever every cell keeps at least one edge).
//
// Device path: every kept edge emits the two entries (i,j) and (j,i) as 64-bit keys row << b | col (b = bits of N: 2 b
// significant bits — 34 at 100 k cells, five 8-bit radix passes; through round 5 the column sat in the low 32 bits: 32 + b
// bits, seven passes), the entries are ordered by one device-wide radix sort over just the bits in use (rocPRIM: a plain
// library sort, not the hot path), equal keys are summed in sorted order and the row pointer comes from a binary search per row.
// Round 6 tried the build without a sort (row buckets filled by atomics, rows ordered by a wave each): slower, because the
// transposed half needs a device-scope atomic per edge — profiles/r06_adjacency_row_buckets.txt; closed.
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr u64 ADJ_NONE = ~0ull;

__global__ __launch_bounds__(256) void k_adj_emit(const double* __restrict__ from, const double* __restrict__ to,
                                                  const double* __restrict__ w, int64_t cap, const int64_t* __restrict__ n_edges_p,
                                                  int64_t N, int b, u64* __restrict__ keys, double* __restrict__ vals,
                                                  uint32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cap) return;
  u64 k0 = ADJ_NONE, k1 = ADJ_NONE;
  double v = 0.0;
  const int64_t n_edges = n_edges_p ? *n_edges_p : cap;
  if (e < n_edges) {
    const double fi = from[e], fj = to[e];
    if (fi >= 1.0 && fi <= (double)N && fj >= 1.0 && fj <= (double)N && fi == trunc(fi) && fj == trunc(fj)) {
      const u64 i = (u64)fi - 1ull, j = (u64)fj - 1ull;
      v = w[e];
      k0 = (i << b) | j;
      if (i != j) k1 = (j << b) | i;                  // a self edge counts once
    } else {
      atomicOr(status, GFICF_ST_BAD_ID);
    }
  }
  keys[2 * e] = k0; vals[2 * e] = v;
  keys[2 * e + 1] = k1; vals[2 * e + 1] = v;
}

// flags[e] = 1 where a new (row, col) starts; flags[M] = 0 (becomes the number of entries after the scan)
__global__ __launch_bounds__(256) void k_adj_heads(const u64* __restrict__ keys, int64_t M, int64_t* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e > M) return;
  int64_t f = 0;
  if (e < M) {
    const u64 k = keys[e];
    f = (k != ADJ_NONE && (e == 0 || keys[e - 1] != k)) ? 1 : 0;
  }
  flags[e] = f;
}

// one thread per head: sum its run of equal keys (in sorted, i.e. emission, order) and write the entry
__global__ __launch_bounds__(256) void k_adj_write(const u64* __restrict__ keys, const double* __restrict__ vals, int64_t M, int b,
                                                   const int64_t* __restrict__ pos, int32_t* __restrict__ indices,
                                                   double* __restrict__ x, int32_t* __restrict__ urow) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= M) return;
  const u64 k = keys[e];
  if (k == ADJ_NONE || (e > 0 && keys[e - 1] == k)) return;
  double s = vals[e];
  for (int64_t t = e + 1; t < M && keys[t] == k; ++t) s += vals[t];
  const int64_t p = pos[e];
  indices[p] = (int32_t)(k & ((1ull << b) - 1ull));
  x[p] = s;
  urow[p] = (int32_t)(k >> b);
}

// indptr[r] = first entry whose row is >= r
__global__ __launch_bounds__(256) void k_adj_indptr(const int32_t* __restrict__ urow, const int64_t* __restrict__ nnz_p, int64_t N,
                                                    int64_t* __restrict__ indptr) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > N) return;
  const int64_t nnz = *nnz_p;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)urow[mid] < r) lo = mid + 1;
    else hi = mid;
  }
  indptr[r] = lo;
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

inline int id_bits(int64_t N) {                        // 2^b > N: an id in [0, N) never has all of its b bits set, so no real key is all ones
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) <= N) ++b;
  return b;
}
inline int key_bits(int64_t N) { return 2 * id_bits(N); }   // col in the low b bits, row above

size_t sort_temp_bytes(int64_t M, int bits) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (u64*)nullptr, (u64*)nullptr, (double*)nullptr, (double*)nullptr, (size_t)M, 0u,
                                  (unsigned)bits, (hipStream_t) nullptr);
  return tmp;
}

}  // namespace

extern "C" {

size_t gficf_adjacency_workspace_bytes(int64_t N, int64_t edge_capacity) {
  if (N <= 0 || edge_capacity <= 0) return 256;
  const size_t M = 2 * (size_t)edge_capacity;
  return 2 * align256(M * sizeof(u64)) + 2 * align256(M * sizeof(double)) + align256((M + 1) * sizeof(int64_t)) +
         align256(M * sizeof(int32_t)) + align256(sort_temp_bytes((int64_t)M, key_bits(N))) + 256;
}

int gficf_adjacency_device(gficf_ctx* ctx, int64_t N, int64_t edge_capacity, const int64_t* d_n_edges, const double* d_from,
                           const double* d_to, const double* d_weight, void* d_ws, size_t ws_bytes, int64_t* d_indptr,
                           int32_t* d_indices, double* d_x) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0 || edge_capacity < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  if (!d_indptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (N == 0 || edge_capacity == 0) {
    GFICF_HIP_CHECK(hipMemsetAsync(d_indptr, 0, sizeof(int64_t) * (size_t)(N + 1), ctx->stream));
    return GFICF_OK;
  }
  if (!d_from || !d_to || !d_weight || !d_ws || !d_indices || !d_x) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ws_bytes < gficf_adjacency_workspace_bytes(N, edge_capacity)) GFICF_FAIL(GFICF_ERR_CAPACITY, "adjacency workspace too small");
  const int64_t M = 2 * edge_capacity;
  const int bits = key_bits(N), idb = id_bits(N);
  char* p = (char*)d_ws;
  u64* k_in = (u64*)p;        p += align256((size_t)M * sizeof(u64));
  u64* k_out = (u64*)p;       p += align256((size_t)M * sizeof(u64));
  double* v_in = (double*)p;  p += align256((size_t)M * sizeof(double));
  double* v_out = (double*)p; p += align256((size_t)M * sizeof(double));
  int64_t* pos = (int64_t*)p; p += align256((size_t)(M + 1) * sizeof(int64_t));
  int32_t* urow = (int32_t*)p; p += align256((size_t)M * sizeof(int32_t));
  size_t tmp_bytes = sort_temp_bytes(M, bits);
  void* tmp = (void*)p;
  hipLaunchKernelGGL(k_adj_emit, dim3((unsigned)gficf_ceil_div(edge_capacity, 256)), dim3(256), 0, ctx->stream, d_from, d_to, d_weight,
                     edge_capacity, d_n_edges, N, idb, k_in, v_in, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  // unused slots carry the all-ones key; they only need to end up behind every real key: ids are below 2^b - 1, so among the `bits` bits
  // sorted the all-ones pattern is larger than any real key
  GFICF_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)M, 0u, (unsigned)bits, ctx->stream));
  hipLaunchKernelGGL(k_adj_heads, dim3((unsigned)gficf_ceil_div(M + 1, 256)), dim3(256), 0, ctx->stream, k_out, M, pos);
  GFICF_HIP_CHECK(hipGetLastError());
  int rc = gficf_exclusive_scan_i64(ctx, pos, M + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adj_write, dim3((unsigned)gficf_ceil_div(M, 256)), dim3(256), 0, ctx->stream, k_out, v_out, M, idb, pos, d_indices, d_x, urow);
  hipLaunchKernelGGL(k_adj_indptr, dim3((unsigned)gficf_ceil_div(N + 1, 256)), dim3(256), 0, ctx->stream, urow, pos + M, N, d_indptr);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------- host form (R glue)
struct gficf_adj_plan {
  int64_t N = 0, nnz = 0;
  int64_t* d_indptr = nullptr;
  int32_t* d_indices = nullptr;
  double* d_x = nullptr;
};

void gficf_adj_plan_free(gficf_ctx* ctx) {        // the buffers are pieces of pool slot 6: nothing to release
  delete ctx->adj_plan;
  ctx->adj_plan = nullptr;
}

extern "C" {

int gficf_adjacency_host_plan(gficf_ctx* ctx, int64_t N, int64_t n_edges, const double* from, const double* to,
                              const double* weight, int64_t* nnz) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0 || n_edges < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (!nnz) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "nnz is NULL");
  if (n_edges > 0 && (!from || !to || !weight)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  gficf_adj_plan_free(ctx);
  gficf_adj_plan* p = new gficf_adj_plan();
  ctx->adj_plan = p;
  p->N = N;
  *nnz = 0;
  const size_t eb = sizeof(double) * (size_t)(n_edges > 0 ? n_edges : 1), cap = (size_t)(n_edges > 0 ? 2 * n_edges : 1);
  const size_t wsb = gficf_adjacency_workspace_bytes(N, n_edges);
  void *d_e = nullptr, *d_ws = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, 3 * eb, &d_e);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, wsb, &d_ws);
  gficf_arena ar;
  const size_t o_ip = ar.take(sizeof(int64_t) * (size_t)(N + 1)), o_ii = ar.take(sizeof(int32_t) * cap), o_xx = ar.take(sizeof(double) * cap);
  if (e == hipSuccess) e = ar.bind(ctx, 6);
  p->d_indptr = ar.at<int64_t>(o_ip); p->d_indices = ar.at<int32_t>(o_ii); p->d_x = ar.at<double>(o_xx);
  double* d_from = (double*)d_e;
  double* d_to = d_from + (n_edges > 0 ? n_edges : 1);
  double* d_w = d_to + (n_edges > 0 ? n_edges : 1);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_from, from, eb, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_to, to, eb, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_w, weight, eb, hipMemcpyHostToDevice, ctx->stream);
  int rc = GFICF_OK;
  int64_t total = 0;
  if (e == hipSuccess) {
    rc = gficf_adjacency_device(ctx, N, n_edges, nullptr, d_from, d_to, d_w, d_ws, wsb, p->d_indptr, p->d_indices, p->d_x);
    if (!rc) e = hipMemcpyAsync(&total, p->d_indptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) { gficf_set_error("HIP failure in gficf_adjacency_host_plan: %s", hipGetErrorString(e)); rc = GFICF_ERR_HIP; }
  if (rc) { gficf_adj_plan_free(ctx); return rc; }
  p->nnz = total;
  *nnz = total;
  return GFICF_OK;
}

int gficf_adjacency_host_finish(gficf_ctx* ctx, void* indptr, int indptr_is_i64, int32_t* indices, double* x) {
  GFICF_CTX_ENTER(ctx);
  gficf_adj_plan* p = ctx->adj_plan;
  if (!p) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_adjacency_host_finish without a plan");
  if (!indptr || (p->nnz > 0 && (!indices || !x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  std::vector<int64_t> ip((size_t)p->N + 1);
  if (p->nnz > 0 && indices && x) {               // (fresh result vectors of the caller: huge pages + parallel first touch)
    gficf_prefault(indices, sizeof(int32_t) * (size_t)p->nnz);
    gficf_prefault(x, sizeof(double) * (size_t)p->nnz);
  }
  hipError_t e = hipMemcpyAsync(ip.data(), p->d_indptr, sizeof(int64_t) * ip.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && p->nnz > 0) e = hipMemcpyAsync(indices, p->d_indices, sizeof(int32_t) * (size_t)p->nnz, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && p->nnz > 0) e = hipMemcpyAsync(x, p->d_x, sizeof(double) * (size_t)p->nnz, hipMemcpyDeviceToHost, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) {
    if (indptr_is_i64) std::memcpy(indptr, ip.data(), sizeof(int64_t) * ip.size());
    else for (size_t c = 0; c < ip.size(); ++c) ((int32_t*)indptr)[c] = (int32_t)ip[c];
  }
  gficf_adj_plan_free(ctx);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_adjacency_host_finish: %s", hipGetErrorString(e));
  return GFICF_OK;
}

}  // extern "C"
