// halo.hip — the sharded Jaccard build on LOCAL ids: a rank's own block of cells plus the remote rows it names ("halo").
//
// What is sharded: the cells of the reference's parallelFor(0, N, worker) (src/rcpp_parallel_jaccard_coeff.cpp:73); the edges of
// cell i need row i and the k rows it names (:28-36).  With the all-gather form every rank holds the whole table; when the ids
// have locality (cells in a spatial / cluster order) a block names few rows outside itself, and this form fetches only those:
//   plan    : mark the ids the block names outside itself in an owner-aligned bitmap over all cells (k_halo_mark), then — one
//             workgroup per owner, one launch — rank each owner's set bits and list them in the owner's fixed-capacity request slots (cap ids per owner: the all-to-alls that follow have equal, host-known
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
constexpr int HP_LIST = 64;              // non-empty words of a sweep whose bits are listed by all threads together

// owner-aligned position of a global id's bit: word r * wpo + (local >> 5), bit local & 31 (halo_map.h)
__device__ inline void halo_bit_of(uint32_t id, uint32_t rpr, uint32_t wpo, uint32_t& w, uint32_t& m) {
  const uint32_t bit = id - 1u, owner = bit / rpr, local = bit - owner * rpr;
  w = owner * wpo + (local >> 5);
  m = 1u << (local & 31u);
}

// K1: bits of the ids this block names outside itself.  idx: (k, ld) column-major block, global 1-based ids.  With locality the few outside ids of a wave fall into one or two bitmap words,
// named ~k times each across the block: the lanes of a wave that hit the same word OR their bits together and ONE lane issues
// the atomic (a first version issued one atomic per reference: a few thousand atomics on a dozen words = 30 us of L2
// serialisation at 100 k cells).
__global__ __launch_bounds__(256) void k_halo_mark(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                                   int64_t b, uint32_t rpr, uint32_t wpo, uint32_t* __restrict__ bitmap) {
  // grid: x over the cells (whole waves), y over the slots — the few waves that sit on a block seam have outside ids in every
  // slot; one slot per wave spreads their serial word-by-word loop below over k waves instead of one
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.y;
  const int64_t n_round = (n_local + 63) & ~(int64_t)63;         // whole waves stay together (the ballots below)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * 256) {
    const int64_t id = i < n_local ? (int64_t)idx[(int64_t)j * ld + i] : 0;
    bool pend = id >= 1 && id <= N_total && (id <= b || id > b + n_local);
    unsigned long long pm = __ballot(pend);
    if (pm == 0ull) continue;                                    // (the common case: every id of the wave lies inside the block)
    uint32_t w = 0, m = 0;
    if (pend) halo_bit_of((uint32_t)id, rpr, wpo, w, m);
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

// K2 (one workgroup PER OWNER): rank of every bitmap word inside its owner, the owner's request slots, and the bitmap handed back
// all zero.  Round 3 ranked the whole bitmap in ONE workgroup and listed the bits in a third launch (plus a memset in front of
// the marking): 20 us of a 79 us chain at 8 ranks.  Owner-aligned words make the owners independent — a rank inside the owner IS
// the request slot — so P workgroups do rank + list side by side in one launch, and the word a thread has read it clears (the next
// step's marking finds zeros: no memset launch); what later kernels look up is winfo = {word, rank}, one 8 B load.
// A thread owns 4 consecutive words of a sweep (one 16 B load); the owner's wpo words are swept in super-tiles of 4096 words.
__global__ __launch_bounds__(HP_THREADS) void k_halo_rank_emit(uint32_t* __restrict__ bitmap, uint2* __restrict__ winfo, uint32_t wpo, int64_t rpr,
                                                               int cap, int32_t* __restrict__ req_out, uint32_t* __restrict__ status) {
  __shared__ int s_wave[2][HP_THREADS / 64];
  __shared__ int s_nlist[2];
  __shared__ uint4 s_list[2][HP_LIST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int owner = blockIdx.x;
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  uint32_t* const bm = bitmap + (size_t)owner * wpo;
  uint2* const wi = winfo + (size_t)owner * wpo;
  int32_t* const slots = req_out + (int64_t)owner * cap;
  bool over = false;
  int base = 0;                                                  // set bits of the owner in front of the sweep (every thread keeps its own copy)
  int par = 0;
  // One barrier per sweep, and one that waits for the LDS only: a __syncthreads() here also waits for the thread's global stores
  // (the cleared bitmap words, the winfo records, the slots) — four store round trips in a kernel whose whole job is a dozen
  // memory operations per thread: 10.9 us at 8 owners x 3125 words against the ~5 us a launch costs anyway.
  for (uint32_t st = 0; st < wpo; st += HP_THREADS * 4, par ^= 1) {
    const uint32_t w0 = st + (uint32_t)tid * 4u;
    v4u v = {0u, 0u, 0u, 0u};
    if (w0 < wpo) {                                              // (wpo is a multiple of 4)
      v = *reinterpret_cast<const v4u*>(bm + w0);
      if ((v.x | v.y | v.z | v.w) != 0u) *reinterpret_cast<v4u*>(bm + w0) = v4u{0u, 0u, 0u, 0u};   // hand the bitmap back clear
    }
    const int cnt = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[par][wave] = incl;                    // (two sets of slots: the next sweep's writes cannot pass this sweep's reads)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int run = base + incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < HP_THREADS / 64; ++w) {
      const int t = s_wave[par][w];
      run += w < wave ? t : 0;
      total += t;
    }
    base += total;
    // The set bits into their slots.  With locality they sit in a dozen words, i.e. in three or four threads: walking a word's
    // bits in its own thread is up to 128 dependent turns of one lane (7 of this kernel's 11 us at 190 rows in 8 owners).  So
    // the non-empty words go on a short list in LDS and ALL threads take one (word, bit) pair each; only a sweep with more
    // non-empty words than the list holds (ids without locality) is walked by the owning threads, in parallel across them.
    if (tid == 0) s_nlist[par] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
    int rr[4];
    if (w0 < wpo) {
      int r = run;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        rr[c] = r;
        wi[w0 + c] = make_uint2(wd[c], (uint32_t)r);
        r += __popc(wd[c]);
        if (wd[c] != 0u) {
          const int e = atomicAdd(&s_nlist[par], 1);
          if (e < HP_LIST) s_list[par][e] = make_uint4(w0 + c, wd[c], (uint32_t)rr[c], 0u);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int nl = s_nlist[par];
    if (nl <= HP_LIST) {
      for (int t = tid; t < nl * 32; t += HP_THREADS) {
        const uint4 en = s_list[par][t >> 5];
        const uint32_t m = 1u << (t & 31);
        if (en.y & m) {
          const int r = (int)en.z + __popc(en.y & (m - 1u));
          if (r < cap) slots[r] = (int32_t)((int64_t)owner * rpr + (int64_t)en.x * 32 + (t & 31) + 1);
          else over = true;
        }
      }
    } else if (w0 < wpo) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int r = rr[c];
        uint32_t rest = wd[c];
        while (rest) {
          const int bpos = __builtin_ctz(rest);
          rest &= rest - 1u;
          if (r < cap) slots[r] = (int32_t)((int64_t)owner * rpr + (int64_t)(w0 + c) * 32 + bpos + 1);
          else over = true;
          ++r;
        }
      }
    }
  }
  // the slots behind the ones in use are empty (the ones in use were written above: no thread clears what another one fills)
  for (int t = base + tid; t < cap; t += HP_THREADS) slots[t] = 0;
  if (over) atomicOr(status, GFICF_ST_HALO_OVERFLOW);
}

// K3: the rows asked of this rank (halo_map.h: gficf_halo_serve_rows) as a launch of its own; the fused form lets it ride in the
// launch that ingests the own cells (jaccard.hip: gficf_jaccard_halo_serve_ingest_device).
__global__ __launch_bounds__(256) void k_halo_serve(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t b,
                                                    const int32_t* __restrict__ req_in, int64_t n_req, int32_t* __restrict__ rows_out,
                                                    uint32_t* __restrict__ status) {
  gficf_halo_serve_rows(idx, n_local, k, ld, b, req_in, n_req, rows_out, status, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// K4: extended index matrix (k, n_ext) in local ids + the local -> global map.  n_ext = n_local + P * cap.
__global__ __launch_bounds__(256) void k_halo_relabel(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                                      int64_t b, int P, int64_t rpr, int cap, const uint2* __restrict__ winfo, uint32_t wpo,
                                                      const int32_t* __restrict__ req_out, const int32_t* __restrict__ rows_in,
                                                      int32_t* __restrict__ idx_ext, int32_t* __restrict__ l2g) {
  const int64_t n_ext = n_local + (int64_t)P * cap;
  const int64_t n = n_ext * (int64_t)k;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t j = t / n_ext, i = t - j * n_ext;
    int32_t v;
    if (i < n_local) {
      v = gficf_halo_local(idx[j * ld + i], N_total, b, n_local, rpr, cap, winfo, wpo);
      if (j == 0) l2g[i] = (int32_t)(b + i + 1);
    } else {
      const int64_t q = i - n_local;
      const int32_t gid = req_out[q];
      v = 0;
      if (gid != 0) {
        v = gficf_halo_local(rows_in[q * k + j], N_total, b, n_local, rpr, cap, winfo, wpo);
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
  const size_t wpo = (size_t)gficf_halo_wpo(gficf_ceil_div(N_total > 0 ? N_total : 1, P));
  // bitmap (P * wpo words) | winfo (P * wpo x 8 B)
  return (((size_t)P * wpo * 4 + 255) & ~(size_t)255) + (size_t)P * wpo * 8 + 256;
}

int gficf_jaccard_halo_plan_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                   int64_t cell_begin, int P, int64_t rows_per_rank, int cap, void* d_ws, int32_t* d_req_out) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || N_total < 0 || cell_begin < 0 || cell_begin + n_local > N_total || P < 1 || rows_per_rank < 1 || cap < 1)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo plan: sizes out of range");
  if (rows_per_rank != gficf_ceil_div(N_total > 0 ? N_total : 1, P))
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo plan: rows_per_rank = %lld, expected ceil(N_total / P) = %lld (equal-pitch blocks; the workspace is laid out for it)",
               (long long)rows_per_rank, (long long)gficf_ceil_div(N_total > 0 ? N_total : 1, P));
  if (N_total > 0x7FFFFFFFll || P > 65535) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "halo plan: N_total beyond int32 ids or more than 65535 ranks");
  if (k > GFICF_JACCARD_MAX_K) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d neighbours per cell: this build handles at most GFICF_JACCARD_MAX_K = %d", k, GFICF_JACCARD_MAX_K);
  if (!d_ws || !d_req_out || (n_local > 0 && k > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (n_local > 0 && ld < n_local) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld < n_local");
  const int64_t wpo = gficf_halo_wpo(rows_per_rank);
  uint32_t* const bitmap = (uint32_t*)d_ws;
  uint2* const winfo = (uint2*)((char*)d_ws + (((size_t)P * (size_t)wpo * 4 + 255) & ~(size_t)255));
  // (no memset: the bitmap is zero when the workspace is new — the caller zeroes it once — and k_halo_rank_emit hands it back zero)
  if (n_local > 0 && k > 0) {
    int64_t blocks = gficf_ceil_div(n_local, 256 * 4);
    if (blocks > (int64_t)ctx->num_cus * 4) blocks = (int64_t)ctx->num_cus * 4;
    hipLaunchKernelGGL(k_halo_mark, dim3((unsigned)blocks, (unsigned)k), dim3(256), 0, ctx->stream, d_idx, n_local, k, ld, N_total, cell_begin,
                       (uint32_t)rows_per_rank, (uint32_t)wpo, bitmap);
  }
  hipLaunchKernelGGL(k_halo_rank_emit, dim3((unsigned)P), dim3(HP_THREADS), 0, ctx->stream, bitmap, winfo, (uint32_t)wpo, rows_per_rank, cap, d_req_out,
                     ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_halo_serve_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t cell_begin,
                                    const int32_t* d_req_in, int64_t n_req, int32_t* d_rows_out) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || n_req < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo serve: negative size");
  if (n_req == 0 || k == 0) return GFICF_OK;
  if (!d_req_in || !d_rows_out || (n_local > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  int64_t blocks = gficf_ceil_div(gficf_halo_serve_items(n_req, k), 256);
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
  const int64_t wpo = gficf_halo_wpo(rows_per_rank);
  const uint2* const winfo = (const uint2*)((const char*)d_ws + (((size_t)P * (size_t)wpo * 4 + 255) & ~(size_t)255));
  int64_t blocks = gficf_ceil_div(n_ext * (int64_t)k, 256 * 8);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_halo_relabel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_idx, n_local, k, ld, N_total, cell_begin, P,
                     rows_per_rank, cap, winfo, (uint32_t)wpo, d_req_out, d_rows_in, d_idx_ext, d_l2g);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

}  // extern "C"
