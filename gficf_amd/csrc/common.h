// common.h — context, error plumbing and small device helpers shared by the HIP sources.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#include "gficf_hip.h"

// Deferred device-side validation flags (OR-ed into gficf_ctx::d_status by kernels).
constexpr uint32_t GFICF_ST_BAD_ID = 1u;    // kNN id outside [1, N] or not an integer
constexpr uint32_t GFICF_ST_BAD_CSC = 2u;   // rowidx outside [0, G) / colptr not monotone
constexpr uint32_t GFICF_ST_BAD_VALUE = 4u; // non-finite coordinate handed to the kNN search / bad edge weight
constexpr uint32_t GFICF_ST_TOO_DENSE = 8u; // Louvain: a vertex touches more communities than its table holds
constexpr uint32_t GFICF_ST_EXPLICIT_ZERO = 16u; // gficf_csc_device met an explicitly stored zero (its fast count is then not exact)
constexpr uint32_t GFICF_ST_DUP_IDS = 64u;       // Jaccard, rows taken to hold distinct ids (gficf_ctx_set_jaccard_distinct): the edge kernel met one that does not
constexpr uint32_t GFICF_ST_SET_OVERFLOW = 128u;  // Jaccard, distinct-ids mode, general kernel: a row's ids overflowed its hash set beyond the list that is checked for repeats
constexpr uint32_t GFICF_ST_NOT_GROUPED = 256u;  // adjacency: the edge list was promised grouped by source cell and is not
constexpr uint32_t GFICF_ST_HALO_OVERFLOW = 32u; // sharded Jaccard, halo form: a block names more rows of one owner than the request slots hold

constexpr int GFICF_POOL_SLOTS = 9;

struct gficf_host_plan;  // gficf_csc.hip
struct gficf_edge_plan;  // jaccard.hip
struct gficf_adj_plan;   // adjacency.hip

struct gficf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;
  uint32_t* d_status = nullptr;   // device status word
  uint32_t* h_status = nullptr;   // pinned host mirror
  void* d_ws = nullptr;           // scan ticket + tile descriptors (fixed size, allocated and zeroed at create)
  uint32_t scan_epoch = 0;        // tag of the current scan launch's descriptors (22 bits, never 0)
  size_t ws_bytes = 0;
  // cur_zero: when set, the scaling pass ORs GFICF_ST_EXPLICIT_ZERO into it on meeting an explicitly stored zero
  // (gficf_csc_device points it at d_status: the violation surfaces at the next gficf_ctx_sync).
  uint32_t* d_flags = nullptr;
  uint32_t* cur_zero = nullptr;
  // options of the GF-ICF chain that the reference's internal helpers take (gficf() itself always uses the defaults):
  // getIdfW(type = classic / prob / smooth) R/gficf.R:89-91, l.norm(norm = l2 / l1) R/gficf.R:100
  int icf_type = 0;
  int norm_l1 = 0;
  // RunModularityClustering(modularity = 1 standard / 2 alternative), reference src/RModularityOptimizer.cpp:36,100 (clustcells()
  // always passes 1)
  int lv_modularity_fn = 1;
  // gficf_ctx_set_jaccard_options: non-integer double ids are truncated as the reference does (src/rcpp_parallel_jaccard_coeff.cpp:28)
  int jaccard_trunc = 0;
  // gficf_ctx_set_jaccard_distinct: the ingest does not look for duplicate ids inside a row; the edge kernel, which meets every
  // one while it builds the row's hash set, raises GFICF_ST_DUP_IDS instead (deferred; the caller re-runs with the option off)
  int jaccard_assume_distinct = 0;
  // gficf_ctx_set_jaccard_direct_max_edges: edges up to which gficf_jaccard_device takes the one-launch form (-1: the build's default)
  int64_t jaccard_direct_max_edges = -1;
  int quiet_rerun = 0;               // a host entry re-runs its sequence after GFICF_ERR_DUPLICATE_IDS: banners are not printed again
  gficf_host_plan* plan = nullptr;
  gficf_edge_plan* edge_plan = nullptr;
  gficf_adj_plan* adj_plan = nullptr;
  // grow-only device scratch of the host entry points (kept between calls, released at destroy or by
  // gficf_ctx_trim): slots 0-3 scratch of one-call entries, 4 GF-ICF host plan, 5 filtered-edge plan,
  // 6 adjacency plan, 7 outputs of the GF-ICF finish call, 8 per-workgroup partial histograms of the GF-ICF count pass
  // (the one scratch a device entry draws from the pool).  No host entry allocates device memory per call.
  void* pool[GFICF_POOL_SLOTS] = {};
  size_t pool_bytes[GFICF_POOL_SLOTS] = {};
  // grow-only PINNED host staging (the compact return of gficf_jaccard_host: uint16 counts land here before the host expands them)
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  // Louvain: the pinned block its control kernels report into and the two events of its one-iteration-ahead loop (louvain.hip; lazily made)
  void* lv_host = nullptr;
  hipEvent_t lv_ev[2] = {nullptr, nullptr};
  // print hook (R glue: Rprintf); NULL = stdout
  void (*print_fn)(const char*) = nullptr;
};

// Device scratch slot of at least `bytes` bytes (reallocated only when it has to grow).
hipError_t gficf_pool_get(gficf_ctx* ctx, int slot, size_t bytes, void** out);
// Pinned host staging of at least `bytes` bytes (same policy; released by gficf_ctx_trim / destroy).
hipError_t gficf_host_stage_get(gficf_ctx* ctx, size_t bytes, void** out);

// Sub-allocation of one pool slot: take() the pieces (256 B aligned), then bind() once.
struct gficf_arena {
  size_t off = 0;
  char* base = nullptr;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = (off + (bytes ? bytes : 1) + 255) & ~(size_t)255;
    return o;
  }
  hipError_t bind(gficf_ctx* ctx, int slot) { return gficf_pool_get(ctx, slot, off ? off : 256, (void**)&base); }
  template <typename T>
  T* at(size_t o) const { return reinterpret_cast<T*>(base + o); }
};

// banner lines of the host entries (the reference prints with Rprintf)
void gficf_print(gficf_ctx* ctx, const char* line);

// A freshly allocated result buffer of the caller's (R's allocator does not ask for huge pages; where transparent huge pages are in
// "madvise" mode — this image, most distributions — its first touch is one 4 KB page fault at a time): advise huge pages for the 2 MB-aligned
// interior before the first touch, 512 x fewer faults.  Harmless where unsupported; GFICF_HIP_NO_HUGEPAGE in the environment: off.
void gficf_advise_hugepages(void* p, size_t bytes);

// Touch every page of a freshly allocated host buffer from several threads (first-touch page faults of a large
// result buffer otherwise run on the one thread doing the device-to-host copy and dominate it).
void gficf_prefault(void* p, size_t bytes);

// share(t) for t in [0, nt) on nt host threads; a thread that cannot be started (std::system_error) has its share run on the calling thread
// instead: nothing is thrown across the C ABI and the result is the same
template <typename F>
inline void gficf_run_shares(int64_t nt, F&& share) {
  std::vector<std::thread> th;
  int64_t started = 0;
  try {
    th.reserve((size_t)(nt > 1 ? nt - 1 : 0));
    for (; started + 1 < nt; ++started) th.emplace_back(share, started);
  } catch (...) {
  }
  for (int64_t t = started; t < nt; ++t) share(t);            // the last share (and every share no thread could be had for): on this thread
  for (auto& x : th) x.join();
}

// releases the host-form GF-ICF plan held by the context, if any (gficf_csc.hip)
void gficf_host_plan_free(gficf_ctx* ctx);
// same for the host-form filtered edge build (jaccard.hip)
void gficf_edge_plan_free(gficf_ctx* ctx);
// gficf_jaccard_edges_filtered_device on a table built from RENUMBERED cells (row p of the table = original cell d_order[p], 0-based; ids inside
// the table in the new numbering, 1-based): both columns come out in the ORIGINAL ids, the edges in the order of the new numbering.  Internal
// since ABI 7 (its one caller is gficf_phenograph_host with GFICF_PHENOGRAPH_ORDER=1: cells renumbered by the search's pivot order; off by default).
int gficf_jaccard_edges_filtered_mapped(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin, int64_t cell_end, uint16_t* d_u_ws,
                                        int64_t* d_cell_ptr, double* d_from, double* d_to, double* d_weight, const int32_t* d_order);
// same for the host-form adjacency build (adjacency.hip)
void gficf_adj_plan_free(gficf_ctx* ctx);

// thread-local last error message
void gficf_set_error(const char* fmt, ...);

#define GFICF_FAIL(code, ...)        \
  do {                               \
    gficf_set_error(__VA_ARGS__);    \
    return (code);                   \
  } while (0)

#define GFICF_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gficf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                      __LINE__);                                                           \
      return GFICF_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

// Binds the calling thread to the context's device.
#define GFICF_CTX_ENTER(ctx)                                         \
  do {                                                               \
    if (!(ctx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");    \
    GFICF_HIP_CHECK(hipSetDevice((ctx)->device));                    \
  } while (0)

__host__ __device__ static inline int64_t gficf_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }


// ---------------------------------------------------------------- decoupled look-back over ticketed tiles (ctx->d_ws)
// ws[0] is the ticket word (tiles are claimed in ticket order, so a tile's predecessors always belong to running
// workgroups), ws[1 + t] tile t's descriptor:  bits 63..42 epoch | 41..40 status (1 = aggregate, 2 = inclusive prefix) |
// 39..0 value, one 8-byte granule written by ONE store and polled with agent-scope loads (the data is the flag, valid
// across XCDs).  The epoch (host counter, never 0; gficf_ws_next_epoch) tells a launch's descriptors from older ones, so
// nothing is zeroed between launches; the workgroup that draws the last ticket resets the ticket word.
constexpr int GFICF_LB_VALUE_BITS = 40;

__device__ inline unsigned long long gficf_lb_desc(uint32_t epoch, uint32_t status, int64_t value) {
  return ((unsigned long long)epoch << 42) | ((unsigned long long)status << GFICF_LB_VALUE_BITS) |
         ((unsigned long long)value & ((1ull << GFICF_LB_VALUE_BITS) - 1ull));
}

// Called by the FIRST WAVE of a workgroup (all 64 lanes) that holds ticket `tile` and the tile's own `total`: publishes the
// aggregate, sums the predecessors' (64 descriptors per round trip: lane l reads tile t0 - l; tiles before the first count
// as an inclusive prefix of 0), publishes the inclusive prefix and returns the exclusive one (same value in every lane).
__device__ inline int64_t gficf_lookback_exclusive(unsigned long long* ws, int64_t tile, int64_t n_tiles, uint32_t epoch, int64_t total) {
  const int lane = threadIdx.x & 63;
  unsigned long long* const desc = ws + 1;
  int64_t run = 0;
  if (tile > 0) {
    if (lane == 0) __hip_atomic_store(desc + tile, gficf_lb_desc(epoch, 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int64_t t0 = tile - 1;; t0 -= 64) {
      const int64_t t = t0 - lane;
      unsigned long long x = gficf_lb_desc(epoch, 2u, 0);
      if (t >= 0) {
        do {      // tickets are handed out in order: every earlier tile is running or done, its descriptor will appear
          x = __hip_atomic_load(desc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((uint32_t)(x >> 42) != epoch || ((x >> GFICF_LB_VALUE_BITS) & 3ull) == 0ull);
      }
      const unsigned long long incl = __ballot(((x >> GFICF_LB_VALUE_BITS) & 3ull) == 2ull);   // never 0 in the round that reaches tile 0
      const int stop = incl ? __builtin_ctzll(incl) : 63;                                      // nearest inclusive prefix
      int64_t v = lane <= stop ? (int64_t)(x & ((1ull << GFICF_LB_VALUE_BITS) - 1ull)) : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
      run += v;
      if (incl) break;
    }
  }
  if (lane == 0) {
    __hip_atomic_store(desc + tile, gficf_lb_desc(epoch, 2u, run + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile == n_tiles - 1) __hip_atomic_store(ws, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every ticket is out
  }
  return run;
}

// Next epoch of the context's look-back workspace (zeroes the workspace on wrap); checks that n_tiles descriptors fit.
int gficf_ws_next_epoch(gficf_ctx* ctx, int64_t n_tiles, uint32_t* epoch);

// In-place exclusive scan of n int64 values on the context's stream (scan.hip).
// One launch (decoupled look-back); uses ctx->d_ws.  Sums must stay below 2^40.
int gficf_exclusive_scan_i64(gficf_ctx* ctx, int64_t* d_data, int64_t n);
