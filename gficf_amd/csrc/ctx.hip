// ctx.hip — library context, error strings, and the int64 exclusive scan helper.
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "common.h"

static thread_local char g_err[768] = "";

void gficf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

hipError_t gficf_pool_get(gficf_ctx* ctx, int slot, size_t bytes, void** out) {
  if (bytes > ctx->pool_bytes[slot]) {
    if (ctx->pool[slot]) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipFree(ctx->pool[slot]);
      ctx->pool[slot] = nullptr;
      ctx->pool_bytes[slot] = 0;
    }
    hipError_t e = hipMalloc(&ctx->pool[slot], bytes);
    if (e != hipSuccess) return e;
    ctx->pool_bytes[slot] = bytes;
  }
  *out = ctx->pool[slot];
  return hipSuccess;
}

void gficf_print(gficf_ctx* ctx, const char* line) {
  if (ctx && ctx->print_fn) { ctx->print_fn(line); return; }
  fputs(line, stdout);
  fflush(stdout);
}

hipError_t gficf_host_stage_get(gficf_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->h_stage_bytes) {
    if (ctx->h_stage) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipHostFree(ctx->h_stage);
      ctx->h_stage = nullptr;
      ctx->h_stage_bytes = 0;
    }
    hipError_t e = hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    ctx->h_stage_bytes = bytes;
  }
  *out = ctx->h_stage;
  return hipSuccess;
}

void gficf_advise_hugepages(void* p, size_t bytes) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
  constexpr uintptr_t HP = (uintptr_t)2 << 20;
  if (!p || bytes < 2 * HP || getenv("GFICF_HIP_NO_HUGEPAGE")) return;
  const uintptr_t a = ((uintptr_t)p + HP - 1) & ~(HP - 1), e = ((uintptr_t)p + bytes) & ~(HP - 1);
  if (e > a) (void)madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);      // (a failure changes nothing)
#else
  (void)p; (void)bytes;
#endif
}

void gficf_prefault(void* p, size_t bytes) {
  constexpr size_t PAGE = 4096, MIN_PER_THREAD = 8u << 20;
  if (!p || bytes < 2 * MIN_PER_THREAD) return;               // small buffers: not worth the threads
  gficf_advise_hugepages(p, bytes);
  unsigned hw = std::thread::hardware_concurrency();
  size_t nt = hw ? hw : 4;
  if (nt > 16) nt = 16;
  if (nt > bytes / MIN_PER_THREAD) nt = bytes / MIN_PER_THREAD;
  if (const char* e = getenv("GFICF_HIP_PREFAULT_THREADS")) {
    const int v = atoi(e);
    if (v <= 0) return;
    nt = (size_t)v;
  }
  volatile char* const base = (volatile char*)p;
  const size_t per = ((bytes / nt + PAGE - 1) / PAGE) * PAGE;
  std::vector<std::thread> th;
  try {                                            // (a thread that cannot be started: its pages are mapped by whoever writes them later)
    th.reserve(nt);
    for (size_t t = 0; t < nt; ++t) {
      const size_t b = t * per, e = b + per < bytes ? b + per : bytes;
      if (b >= e) break;
      th.emplace_back([=] {
        // the buffer is output-only (fully overwritten afterwards): writing is allowed and is what maps the page
        size_t q = b;
        const size_t mis = (size_t)((uintptr_t)(base + q) & (PAGE - 1));
        if (mis) { base[q] = 0; q += PAGE - mis; }
        for (; q < e; q += PAGE) base[q] = 0;
      });
    }
  } catch (...) {
  }
  for (auto& t : th) t.join();
}

extern "C" {

int gficf_hip_abi_version(void) { return GFICF_HIP_ABI_VERSION; }

int gficf_ctx_set_print(gficf_ctx* ctx, void (*fn)(const char*)) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  ctx->print_fn = fn;
  return GFICF_OK;
}

int gficf_ctx_trim(gficf_ctx* ctx) {
  GFICF_CTX_ENTER(ctx);
  GFICF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  gficf_host_plan_free(ctx);
  gficf_edge_plan_free(ctx);
  gficf_adj_plan_free(ctx);
  for (int s = 0; s < GFICF_POOL_SLOTS; ++s) {
    if (ctx->pool[s]) (void)hipFree(ctx->pool[s]);
    ctx->pool[s] = nullptr;
    ctx->pool_bytes[s] = 0;
  }
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  ctx->h_stage = nullptr;
  ctx->h_stage_bytes = 0;
  return GFICF_OK;
}

const char* gficf_last_error(void) { return g_err; }

int gficf_device_count(int* count) {
  if (!count) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    gficf_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return GFICF_ERR_NO_DEVICE;
  }
  *count = n;
  return GFICF_OK;
}

int gficf_ctx_create(int device, void* stream, gficf_ctx** out) {
  if (!out) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    GFICF_FAIL(GFICF_ERR_NO_DEVICE, "no HIP device visible (libgficf_hip needs an AMD GPU; there is no CPU fallback)");
  if (device < 0 || device >= n)
    GFICF_FAIL(GFICF_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
  GFICF_HIP_CHECK(hipSetDevice(device));
  gficf_ctx* c = new gficf_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  c->ws_bytes = 1u << 20;
  hipError_t e = hipMalloc((void**)&c->d_status, sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(c->d_status, 0, sizeof(uint32_t));
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_status, sizeof(uint32_t), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&c->d_ws, c->ws_bytes);
  if (e == hipSuccess) e = hipMemset(c->d_ws, 0, c->ws_bytes);     // scan ticket + descriptors (epoch 0 = never written)
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_flags, 4 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(c->d_flags, 0, 4 * sizeof(uint32_t));
  if (e != hipSuccess) {
    gficf_set_error("context allocation failed: %s", hipGetErrorString(e));
    gficf_ctx_destroy(c);
    return GFICF_ERR_HIP;
  }
  *c->h_status = 0;
  *out = c;
  return GFICF_OK;
}

void gficf_ctx_destroy(gficf_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  gficf_host_plan_free(ctx);
  gficf_edge_plan_free(ctx);
  gficf_adj_plan_free(ctx);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  if (ctx->d_ws) (void)hipFree(ctx->d_ws);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  for (void* q : ctx->pool)
    if (q) (void)hipFree(q);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->lv_host) (void)hipHostFree(ctx->lv_host);
  for (hipEvent_t e : ctx->lv_ev)
    if (e) (void)hipEventDestroy(e);
  delete ctx;
}

int gficf_ctx_set_stream(gficf_ctx* ctx, void* stream) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  ctx->stream = (hipStream_t)stream;
  return GFICF_OK;
}

int gficf_ctx_set_gficf_options(gficf_ctx* ctx, int icf_type, int norm) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  if (icf_type < 0 || icf_type > 2 || norm < 0 || norm > 1) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "icf_type must be 0..2 and norm 0..1");
  ctx->icf_type = icf_type;
  ctx->norm_l1 = norm;
  return GFICF_OK;
}

int gficf_ctx_set_louvain_options(gficf_ctx* ctx, int modularity_function) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  if (modularity_function != 1 && modularity_function != 2) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "Modularity parameter must be equal to 1 or 2.");
  ctx->lv_modularity_fn = modularity_function;
  return GFICF_OK;
}

int gficf_ctx_set_jaccard_options(gficf_ctx* ctx, int truncate_noninteger_ids) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  ctx->jaccard_trunc = truncate_noninteger_ids ? 1 : 0;
  return GFICF_OK;
}

int gficf_ctx_set_jaccard_distinct(gficf_ctx* ctx, int assume_distinct) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  ctx->jaccard_assume_distinct = assume_distinct ? 1 : 0;
  return GFICF_OK;
}

int gficf_ctx_set_jaccard_direct_max_edges(gficf_ctx* ctx, int64_t max_edges) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  ctx->jaccard_direct_max_edges = max_edges < 0 ? -1 : max_edges;
  return GFICF_OK;
}

int gficf_ctx_sync(gficf_ctx* ctx) {
  GFICF_CTX_ENTER(ctx);
  GFICF_HIP_CHECK(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  GFICF_HIP_CHECK(hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream));
  GFICF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  uint32_t st = *ctx->h_status;
  if (st & GFICF_ST_BAD_ID)
    GFICF_FAIL(GFICF_ERR_BAD_ID, "kNN index matrix holds an id outside [1, N] or a non-integer value");
  if (st & GFICF_ST_BAD_VALUE)
    GFICF_FAIL(GFICF_ERR_BAD_VALUE, "a non-finite kNN coordinate, or an edge weight that is negative / not finite");
  if (st & GFICF_ST_BAD_CSC)
    GFICF_FAIL(GFICF_ERR_BAD_CSC, "CSC matrix malformed: row index outside [0, G) or colptr not monotone");
  if (st & GFICF_ST_DUP_IDS)
    GFICF_FAIL(GFICF_ERR_DUPLICATE_IDS, "a row of the kNN index matrix names an id twice and the context was told rows hold distinct ids "
                                        "(gficf_ctx_set_jaccard_distinct): discard the edges and re-run ingest + edges with the option off");
  if (st & GFICF_ST_SET_OVERFLOW)
    GFICF_FAIL(GFICF_ERR_SET_OVERFLOW, "a row's ids overflowed the edge kernel's hash set (more than six found their bucket full: ids spread uniformly at k near "
                                       "256) and the context was told rows hold distinct ids: discard the edges and re-run with the option off, or with "
                                       "GFICF_JACCARD_SORTED_FROM=57 in the environment (sorted-row path)");
  if (st & GFICF_ST_EXPLICIT_ZERO)
    GFICF_FAIL(GFICF_ERR_EXPLICIT_ZEROS, "the CSC matrix stores explicit zeros, which gficf_csc_device's count of stored entries takes for "
                                         "non-zero cells (rowSums(M != 0), reference R/gficf.R:40,88): call gficf_csc_exact_device");
  if (st & GFICF_ST_HALO_OVERFLOW)
    GFICF_FAIL(GFICF_ERR_CAPACITY, "sharded Jaccard, halo exchange: the block names more rows of one owner than the request slots hold "
                                   "(ids without locality): use the all-gather exchange for this input");
  if (st & GFICF_ST_NOT_GROUPED)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_adjacency_device: the edge list was passed as grouped by source cell (grouped_by_source = 1) and is not");
  if (st & GFICF_ST_TOO_DENSE)
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "Louvain: one hash class of a vertex's neighbouring communities overflowed the 8192-slot table");
  return GFICF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ exclusive scan (int64)
namespace {

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline int64_t wave_inclusive_scan(int64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int64_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// Block-wide exclusive scan of one value per thread; returns the exclusive prefix and the
// block total through *total.  SCAN_THREADS threads.
__device__ inline int64_t block_exclusive_scan(int64_t v, int64_t* total) {
  __shared__ int64_t s_wave[SCAN_THREADS / 64];
  __shared__ int64_t s_total;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t inc = wave_inclusive_scan(v, lane);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int64_t w = lane < SCAN_THREADS / 64 ? s_wave[lane] : 0;
    int64_t winc = wave_inclusive_scan(w, lane);
    if (lane < SCAN_THREADS / 64) s_wave[lane] = winc - w;
    if (lane == SCAN_THREADS / 64 - 1) s_total = winc;
  }
  __syncthreads();
  int64_t r = inc - v + s_wave[wave];
  *total = s_total;
  __syncthreads();
  return r;
}

// Single-pass scan with decoupled look-back over tiles of SCAN_TILE elements (gficf_lookback_exclusive, common.h).
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_lookback(int64_t* __restrict__ d, int64_t n, unsigned long long* ws,
                                                                uint32_t epoch, uint32_t* __restrict__ status) {
  __shared__ unsigned long long s_tile;
  __shared__ int64_t s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(&ws[0], 1ull);
  __syncthreads();
  const int64_t tile = (int64_t)s_tile;
  const int64_t base = tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int t = 0; t < SCAN_ITEMS; ++t) {
    v[t] = base + t < n ? d[base + t] : 0;
    s += v[t];
  }
  int64_t total;
  int64_t ex = block_exclusive_scan(s, &total);
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0 && (total < 0 || (total >> GFICF_LB_VALUE_BITS) != 0)) atomicOr(status, GFICF_ST_BAD_CSC);    // counts out of range
    const int64_t run = gficf_lookback_exclusive(ws, tile, (int64_t)gridDim.x, epoch, total);
    if (threadIdx.x == 0) s_prefix = run;
  }
  __syncthreads();
  ex += s_prefix;
#pragma unroll
  for (int t = 0; t < SCAN_ITEMS; ++t) {
    if (base + t < n) d[base + t] = ex;
    ex += v[t];
  }
}

}  // namespace

int gficf_ws_next_epoch(gficf_ctx* ctx, int64_t n_tiles, uint32_t* epoch) {
  if ((size_t)(n_tiles + 1) * sizeof(unsigned long long) > ctx->ws_bytes)
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "%lld tiles exceed the look-back workspace", (long long)n_tiles);
  if (++ctx->scan_epoch >= (1u << 22)) {            // epoch wrap: start over from clean descriptors
    GFICF_HIP_CHECK(hipMemsetAsync(ctx->d_ws, 0, ctx->ws_bytes, ctx->stream));
    ctx->scan_epoch = 1;
  }
  *epoch = ctx->scan_epoch;
  return GFICF_OK;
}

int gficf_exclusive_scan_i64(gficf_ctx* ctx, int64_t* d_data, int64_t n) {
  if (n <= 0) return GFICF_OK;
  const int64_t nb = gficf_ceil_div(n, SCAN_TILE);
  uint32_t epoch = 0;
  const int rc = gficf_ws_next_epoch(ctx, nb, &epoch);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, d_data, n, (unsigned long long*)ctx->d_ws,
                     epoch, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}
