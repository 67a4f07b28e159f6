// halo.hip — the sharded Jaccard build on LOCAL ids: a rank's own block of cells plus the remote rows it names ("halo").
//
// What is sharded: the cells of the reference's parallelFor(0, N, worker) (src/rcpp_parallel_jaccard_coeff.cpp:73); the edges of
// cell i need row i and the k rows it names (:28-36).  With the all-gather form every rank holds the whole table; when the ids
// have locality (cells in a spatial / cluster order) a block names few rows outside itself, and this form fetches only those:
//   plan    : mark the ids the block names outside itself in a bitmap over all cells, rank the set bits, and list them per
//             owner in fixed-capacity request slots (cap ids per owner: the all-to-alls that follow have equal, host-known
//             sizes — no count exchange, no host round trip; a block that names more than cap rows of one owner raises
//             GFICF_ERR_CAPACITY at the next sync and the caller switches to the all-gather form);
//   serve   : the owner copies the requested rows of ITS input block (raw global ids) into reply slots;
//   relabel : the extended index matrix of the rank's sub-problem — own cells then halo slots — in local ids
//             (own cell c -> c - b + 1; halo slot q -> n_local + q + 1; an id a halo row names outside own + halo -> 0: it
//             cannot be in any own row, so it never counts), and the local -> global map for the edge store.
// The sub-problem has n_local + P * cap rows: below 2^17 it takes the compact 64 B-row table and the fast edge kernel whatever
// N_total is (the all-gather form falls back to 128 B rows from 2^17 cells on).  Kernels here are O(n_local * k) elementwise /
// small scans; the edge build is jaccard.hip's, with the map applied where the edges are written.
// (The plan as ONE launch — every marking workgroup draws a ticket, the last one ranks the words and lists the bits alone — was
// built and measured at the end of round 3: bit-identical, 180-210 us with an agent-scope fence per workgroup (each one writes
// the XCD's L2 back), 44-78 us without fences (s_waitcnt + agent-scope loads of the bitmap) against the 14-20 us of the fill +
// three launches below: what the plan costs is its chain of dependent memory round trips, which one workgroup on one CU walks
// more slowly than three small grids do, not the launches.)
#include "common.h"
#include "halo_map.h"

namespace {

constexpr int HP_THREADS = 1024;

// K1: bits of the ids this block names outside itself.  idx: (k, ld) column-major block, global 1-based ids.  With locality the few outside ids of a wave fall into one or two bitmap words,
// named ~k times each across the block: the lanes of a wave that hit the same word OR their bits together and ONE lane issues
// the atomic (a first version issued one atomic per reference: a few thousand atomics on a dozen words = 30 us of L2
// serialisation at 100 k cells).
__global__ __launch_bounds__(256) void k_halo_mark(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                                   int64_t b, uint32_t* __restrict__ bitmap) {
  // grid: x over the cells (whole waves), y over the slots — the few waves that sit on a block seam have outside ids in every
  // slot; one slot per wave spreads their serial word-by-word loop below over k waves instead of one
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.y;
  const int64_t n_round = (n_local + 63) & ~(int64_t)63;         // whole waves stay together (the ballots below)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * 256) {
    const int64_t id = i < n_local ? (int64_t)idx[(int64_t)j * ld + i] : 0;
    bool pend = id >= 1 && id <= N_total && (id <= b || id > b + n_local);
    const uint32_t w = (uint32_t)((id - 1) >> 5), m = 1u << ((id - 1) & 31);
    unsigned long long pm = __ballot(pend);
    while (pm) {                                                 // wave-uniform; rarely entered
      const int leader = __builtin_ctzll(pm);
      const uint32_t wl = (uint32_t)__shfl((int)w, leader);
      const bool mine = pend && w == wl;
      uint32_t acc = mine ? m : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc |= (uint32_t)__shfl_xor((int)acc, d);
      if (lane == leader) atomicOr(bitmap + wl, acc);            // no return value: nothing waits for it (a test-before-set load here
                                                                 // put one memory latency into every turn of this loop: 50 us)
      if (mine) pend = false;
      pm = __ballot(pend);
    }
  }
}

// K2 (one workgroup): exclusive rank of every bitmap word and the first rank of every owner; clears the request slots.
// word_rank[w] = set bits in words < w; owner_start[r] = rank of the first bit of owner r (r = 0..P; [P] = all set bits).
// The bitmap is swept in super-tiles of 1024 threads x 32 words; a thread's 32 words are 8 independent 16 B loads (a first
// version walked its words one dependent load at a time: 78 us at 800 k cells against 4).  `words` is a multiple of 4 and the
// buffers are 16 B aligned (gficf_jaccard_halo_workspace_bytes).
constexpr int HP_PER = 32;                          // words per thread and super-tile

__global__ __launch_bounds__(HP_THREADS) void k_halo_compact(const uint32_t* __restrict__ bitmap, int64_t words, int64_t N_total, int P, int64_t rpr,
                                                             int cap, int32_t* __restrict__ word_rank, int32_t* __restrict__ owner_start,
                                                             int32_t* __restrict__ req_out, uint32_t* __restrict__ status) {
  __shared__ int s_wave[HP_THREADS / 64];
  __shared__ int s_run;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  typedef int v4i __attribute__((ext_vector_type(4)));
  if (tid == 0) s_run = 0;
  for (int64_t t = tid; t < (int64_t)P * cap; t += HP_THREADS) req_out[t] = 0;       // clear the request slots
  __syncthreads();
  for (int64_t st = 0; st < words; st += (int64_t)HP_THREADS * HP_PER) {
    const int64_t w0 = st + (int64_t)tid * HP_PER;
    v4u v[HP_PER / 4];
#pragma unroll
    for (int c = 0; c < HP_PER / 4; ++c) v[c] = (w0 + 4 * c < words) ? *reinterpret_cast<const v4u*>(bitmap + w0 + 4 * c) : v4u{0u, 0u, 0u, 0u};
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < HP_PER / 4; ++c) cnt += __popc(v[c].x) + __popc(v[c].y) + __popc(v[c].z) + __popc(v[c].w);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = s_run + incl - cnt;
#pragma unroll
    for (int w = 0; w < HP_THREADS / 64; ++w) run += w < wave ? s_wave[w] : 0;
#pragma unroll
    for (int c = 0; c < HP_PER / 4; ++c) {
      if (w0 + 4 * c < words) {
        v4i o;
        o.x = run; run += __popc(v[c].x);
        o.y = run; run += __popc(v[c].y);
        o.z = run; run += __popc(v[c].z);
        o.w = run; run += __popc(v[c].w);
        *reinterpret_cast<v4i*>(word_rank + w0 + 4 * c) = o;
      }
    }
    __syncthreads();
    if (tid == HP_THREADS - 1) s_run = run;          // (the last thread's running count is the tile's inclusive total)
    __syncthreads();
  }
  const int total = s_run;
  if (tid <= P) {                                   // word_rank is complete (same workgroup, behind the barriers above)
    int64_t bit = (int64_t)tid * rpr;               // first bit of owner tid
    if (bit > N_total) bit = N_total;
    int v = total;
    if (bit < N_total) {
      const int64_t w = bit >> 5;
      v = word_rank[w] + __popc(bitmap[w] & ((1u << (bit & 31)) - 1u));
    }
    owner_start[tid] = v;
  }
}

// K2b: every set bit into its owner's list — one thread per id (a first version let the thread that owns a bitmap word walk its
// bits: with locality the set bits sit in a dozen words, i.e. in three or four threads, 50 dependent steps each: 60-90 us).
__global__ __launch_bounds__(256) void k_halo_emit(const uint32_t* __restrict__ bitmap, int64_t N_total, int P, int64_t rpr, int cap,
                                                   const int32_t* __restrict__ word_rank, const int32_t* __restrict__ owner_start,
                                                   int32_t* __restrict__ req_out, uint32_t* __restrict__ status) {
  for (int64_t bit = (int64_t)blockIdx.x * 256 + threadIdx.x; bit < N_total; bit += (int64_t)gridDim.x * 256) {
    const int64_t w = bit >> 5;
    const uint32_t word = bitmap[w], m = 1u << (bit & 31);
    if ((word & m) == 0u) continue;
    const int owner = (int)(bit / rpr);
    const int pos = word_rank[w] + __popc(word & (m - 1u)) - owner_start[owner];
    if (pos < cap) req_out[(int64_t)owner * cap + pos] = (int32_t)(bit + 1);
    else atomicOr(status, GFICF_ST_HALO_OVERFLOW);
  }
}

// K3: the rows asked of this rank.  req_in: n_req ids (0 = empty slot: nothing is written, the requester reads a slot's row only
// where it asked for one), all inside this rank's block (b, b + n_local].  One thread per (slot, j).
__global__ __launch_bounds__(256) void k_halo_serve(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t b,
                                                    const int32_t* __restrict__ req_in, int64_t n_req, int32_t* __restrict__ rows_out,
                                                    uint32_t* __restrict__ status) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n_req; q += (int64_t)gridDim.x * 256) {
    const int64_t id = req_in[q];
    if (id == 0) continue;
    const int64_t row = id - 1 - b;
    const bool ok = row >= 0 && row < n_local;
    if (!ok) atomicOr(status, GFICF_ST_BAD_ID);    // a request for a row this rank does not own: the ranks disagree on the blocks
    for (int j = 0; j < k; ++j) rows_out[q * k + j] = ok ? idx[(int64_t)j * ld + row] : 0;
  }
}

// K4: extended index matrix (k, n_ext) in local ids + the local -> global map.  n_ext = n_local + P * cap.
__global__ __launch_bounds__(256) void k_halo_relabel(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                                      int64_t b, int P, int64_t rpr, int cap, const uint32_t* __restrict__ bitmap,
                                                      const int32_t* __restrict__ word_rank, const int32_t* __restrict__ owner_start,
                                                      const int32_t* __restrict__ req_out, const int32_t* __restrict__ rows_in,
                                                      int32_t* __restrict__ idx_ext, int32_t* __restrict__ l2g) {
  const int64_t n_ext = n_local + (int64_t)P * cap;
  const int64_t n = n_ext * (int64_t)k;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t j = t / n_ext, i = t - j * n_ext;
    int32_t v;
    if (i < n_local) {
      v = gficf_halo_local(idx[j * ld + i], N_total, b, n_local, rpr, cap, bitmap, word_rank, owner_start);
      if (j == 0) l2g[i] = (int32_t)(b + i + 1);
    } else {
      const int64_t q = i - n_local;
      const int32_t gid = req_out[q];
      v = 0;
      if (gid != 0) {
        v = gficf_halo_local(rows_in[q * k + j], N_total, b, n_local, rpr, cap, bitmap, word_rank, owner_start);
        if (v < 0) v = 0;                          // (its owner's ingest reports the bad id)
      }
      if (j == 0) l2g[i] = gid;
    }
    idx_ext[t] = v;
  }
}

}  // namespace

extern "C" {

size_t gficf_jaccard_halo_workspace_bytes(int64_t N_total, int P) {
  if (N_total < 0 || P < 1) return 0;
  const size_t words = ((size_t)((N_total + 31) / 32) + 4) & ~(size_t)3;
  // bitmap | word_rank | owner_start (P + 1)
  return ((words * 4 + 255) & ~(size_t)255) * 2 + (((size_t)P + 1) * 4 + 255 & ~(size_t)255);
}

int gficf_jaccard_halo_plan_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                   int64_t cell_begin, int P, int64_t rows_per_rank, int cap, void* d_ws, int32_t* d_req_out) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || N_total < 0 || cell_begin < 0 || cell_begin + n_local > N_total || P < 1 || rows_per_rank < 1 || cap < 1)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo plan: sizes out of range");
  if ((int64_t)P * rows_per_rank < N_total) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo plan: P * rows_per_rank < N_total");
  if (N_total > 0x7FFFFFFFll || P > HP_THREADS - 1) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "halo plan: N_total beyond int32 ids or more than %d ranks", HP_THREADS - 1);
  if (k > GFICF_JACCARD_MAX_K) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d neighbours per cell: this build handles at most GFICF_JACCARD_MAX_K = %d", k, GFICF_JACCARD_MAX_K);
  if (!d_ws || !d_req_out || (n_local > 0 && k > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (n_local > 0 && ld < n_local) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld < n_local");
  const int64_t words = ((N_total + 31) / 32 + 4) & ~(int64_t)3;
  const size_t seg = ((size_t)words * 4 + 255) & ~(size_t)255;
  uint32_t* const bitmap = (uint32_t*)d_ws;
  int32_t* const word_rank = (int32_t*)((char*)d_ws + seg);
  int32_t* const owner_start = (int32_t*)((char*)d_ws + 2 * seg);
  GFICF_HIP_CHECK(hipMemsetAsync(bitmap, 0, (size_t)words * 4, ctx->stream));
  if (n_local > 0 && k > 0) {
    int64_t blocks = gficf_ceil_div(n_local, 256 * 4);
    if (blocks > (int64_t)ctx->num_cus * 4) blocks = (int64_t)ctx->num_cus * 4;
    hipLaunchKernelGGL(k_halo_mark, dim3((unsigned)blocks, (unsigned)k), dim3(256), 0, ctx->stream, d_idx, n_local, k, ld, N_total, cell_begin, bitmap);
  }
  hipLaunchKernelGGL(k_halo_compact, dim3(1), dim3(HP_THREADS), 0, ctx->stream, bitmap, words, N_total, P, rows_per_rank, cap, word_rank,
                     owner_start, d_req_out, ctx->d_status);
  if (N_total > 0) {
    int64_t eb = gficf_ceil_div(N_total, 256);
    if (eb > (int64_t)ctx->num_cus * 16) eb = (int64_t)ctx->num_cus * 16;
    hipLaunchKernelGGL(k_halo_emit, dim3((unsigned)eb), dim3(256), 0, ctx->stream, bitmap, N_total, P, rows_per_rank, cap, word_rank, owner_start,
                       d_req_out, ctx->d_status);
  }
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_halo_serve_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t cell_begin,
                                    const int32_t* d_req_in, int64_t n_req, int32_t* d_rows_out) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || n_req < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo serve: negative size");
  if (n_req == 0 || k == 0) return GFICF_OK;
  if (!d_req_in || !d_rows_out || (n_local > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  int64_t blocks = gficf_ceil_div(n_req, 256);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_halo_serve, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_idx, n_local, k, ld, cell_begin, d_req_in, n_req,
                     d_rows_out, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_halo_relabel_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                      int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                      const int32_t* d_rows_in, int32_t* d_idx_ext, int32_t* d_l2g) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || N_total < 0 || P < 1 || cap < 1 || rows_per_rank < 1) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo relabel: sizes out of range");
  if (!d_ws || !d_req_out || !d_rows_in || !d_idx_ext || !d_l2g || (n_local > 0 && k > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  const int64_t n_ext = n_local + (int64_t)P * cap;
  if (n_ext * (int64_t)k == 0) return GFICF_OK;
  const int64_t words = ((N_total + 31) / 32 + 4) & ~(int64_t)3;
  const size_t seg = ((size_t)words * 4 + 255) & ~(size_t)255;
  const uint32_t* const bitmap = (const uint32_t*)d_ws;
  const int32_t* const word_rank = (const int32_t*)((const char*)d_ws + seg);
  const int32_t* const owner_start = (const int32_t*)((const char*)d_ws + 2 * seg);
  int64_t blocks = gficf_ceil_div(n_ext * (int64_t)k, 256 * 8);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_halo_relabel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_idx, n_local, k, ld, N_total, cell_begin, P,
                     rows_per_rank, cap, bitmap, word_rank, owner_start, d_req_out, d_rows_in, d_idx_ext, d_l2g);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

}  // extern "C"
