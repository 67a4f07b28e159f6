// jaccard.hip — Phenograph kNN -> Jaccard edge build for gfx950 (MI355X).
//
// Replaces the reference hot loop JCoefficient::operator()
// (reference src/rcpp_parallel_jaccard_coeff.cpp:24-55): for every cell i and neighbour
// slot j,  u = |multiset(row i) ∩ multiset(row idx[i,j])|  and the edge row
// (i+1, idx[i,j], u/(2k-u)), zero when u == 0.
//
// This is integer set work bounded by memory, not a contraction: no MFMA.  Design:
//   * ingest  : the R matrix (column-major, int32 or double) is transposed once into a row-major table, one row per
//               cell, so that one neighbour row is one (or a few) contiguous 64..1024 B reads: wide rows (KPAD uint32
//               ids, KPAD in {16,32,64,128,256}) or, below 2^17 cells, compact rows of half the bytes (pre-hashed 16-bit
//               halves + a bitmap of bit 16; "table row formats" below).  Ids are validated here; one bit of a row
//               flags a row that holds duplicate ids (never the case for real kNN output).
//   * edges   : one wave64 per cell.  Row i is staged in LDS as a 2-slot-bucket hash set (one ds_read_b64 per
//               probe, no probing loop); 4..16 neighbour rows are gathered per wave-instruction (16 B per lane), every
//               lane probes the set with its ids, and the per-edge intersection count is a DPP sum over the row's
//               lanes.  Weights come from a per-block LDS table W[u] = u/(2k-u) computed in IEEE double, so they are
//               bit-identical to the reference's division.  For k <= 32 (k_jaccard_edges_pipe) a wave keeps two cells
//               in flight and writes its edges four consecutive cells at a time; k_jaccard_edges is the general form.
//   * rows with duplicate ids (multiset semantics): exact slow path (all-pairs with occurrence ranks), keys that
//               overflow a bucket: a short per-wave list.
// DESIGN.md section 3 has the measurements behind these choices (memory-only model, ablations, instruction rates).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"
#include "halo_map.h"

#include "jaccard_shared.h"
#include "jaccard_ingest.h"
#include "jaccard_edges_general.h"
#include "jaccard_edges_pipe.h"
#include "jaccard_edges_bits.h"
#include "jaccard_sorted.h"
#include "jaccard_direct.h"

namespace {


// edge_kernel_dup_status() reads the EdgeOut argument at EDGE_KERNARG_OUT: both kernels must take exactly EdgeKernArgs' list
static_assert(std::is_same<decltype(&k_jaccard_edges<32, false, false, OUT_RMAT, false>), EdgeKernFn>::value &&
              std::is_same<decltype(&k_jaccard_edges<64, false, true, OUT_U16, true>), EdgeKernFn>::value &&
              std::is_same<decltype(&k_jaccard_edges_pipe<32, false, true, OUT_RMAT, true, false, true>), EdgeKernFn>::value &&
              std::is_same<decltype(&k_jaccard_edges_pipe<16, true, false, OUT_RMAT_U, true, true, false>), EdgeKernFn>::value &&
              std::is_same<decltype(&k_jaccard_edges_bits<7, OUT_RMAT, false>), EdgeKernFn>::value,
              "the edge kernels' parameter list and EdgeKernArgs differ: edge_kernel_dup_status() would read a wrong slot");


// ------------------------------------------------------------------ edge filter (N1)
// The caller's next line, relations[relations[,3] > 0, ] (reference R/clustCells.R:66), on the
// device: per-cell count of edges with u > 0 -> exclusive scan -> ordered compacted write.
__global__ __launch_bounds__(256) void k_edge_kept_count(const uint16_t* __restrict__ u16, int64_t n_cells, int k,
                                                         int64_t* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c == 0) out[n_cells] = 0;
  if (c >= n_cells) return;
  int n = 0;
  for (int s = 0; s < k; ++s) n += u16[c * k + s] != 0;
  out[c] = n;
}

template <int KPAD, bool CMP>
__global__ __launch_bounds__(256) void k_edge_write(const uint32_t* __restrict__ table, const uint16_t* __restrict__ u16,
                                                    int k, int64_t cell_begin, int64_t n_cells,
                                                    const int64_t* __restrict__ ptr, double* __restrict__ from,
                                                    double* __restrict__ to, double* __restrict__ weight, int pitch,
                                                    const int32_t* __restrict__ order) {
  __shared__ double s_lut[GFICF_JACCARD_MAX_K + 1];
  for (int u = threadIdx.x; u <= k; u += 256) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);   // reference :51
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t c = w0; c < n_cells; c += nw) {
    int64_t pos = ptr[c];
    for (int s0 = 0; s0 < k; s0 += 64) {
      const int s = s0 + lane;
      int u = 0;
      uint32_t dst = 0;
      if (s < k) {
        u = u16[c * k + s];
        dst = row_slot_id(table + (cell_begin + c) * pitch, s, KPAD, CMP);
      }
      const bool kp = u > 0;
      const unsigned long long m = __ballot(kp);
      if (kp) {
        const int64_t d = pos + __popcll(m & lt_mask);
        // order != NULL: the table is in a renumbering of the cells (row p = original cell order[p]): both columns in original ids
        from[d] = order ? (double)(order[cell_begin + c] + 1) : (double)(uint32_t)(cell_begin + c + 1);
        to[d] = order ? (double)(order[dst - 1u] + 1) : (double)dst;
        weight[d] = s_lut[u];
      }
      pos += __popcll(m);
    }
  }
}

// ------------------------------------------------- non-integer double ids, the reference's way (gficf_ctx_set_jaccard_options)
// src/rcpp_parallel_jaccard_coeff.cpp:28-52 on the raw doubles: kk = (int)(mat(i,j) - 1) addresses the neighbour row, the two
// rows are intersected as multisets of doubles (std::sort + std::set_intersection), the edge row is (i+1, kk+1, u/(2k-u)).
// One wave per cell, rows staged in LDS, all-pairs with occurrence ranks (the multiset rule of slow_cell).  Values outside
// (0, N+1) — where the reference reads outside the matrix — and NaN raise GFICF_ST_BAD_ID; their edges are left zero.
__global__ __launch_bounds__(256) void k_jaccard_trunc_f64(const double* __restrict__ mat, int64_t N, int k, int64_t ld,
                                                           double* __restrict__ o_src, double* __restrict__ o_dst,
                                                           double* __restrict__ o_w, uint32_t* __restrict__ status) {
  __shared__ double sA[4][GFICF_JACCARD_MAX_K], sB[4][GFICF_JACCARD_MAX_K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* const A = sA[wave];
  double* const B = sB[wave];
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  const double hi = (double)N + 1.0;
  for (int64_t i = w0; i < N; i += nw) {
    bool bad = false;
    for (int e = lane; e < k; e += 64) {
      const double v = mat[(int64_t)e * ld + i];
      bad |= !(v > 0.0 && v < hi);
      A[e] = v;
    }
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    wave_lds_fence();
    for (int s = 0; s < k; ++s) {
      const double v = A[s];
      const int64_t r = i * (int64_t)k + s;
      int u = 0;
      int64_t kk = 0;
      if (v > 0.0 && v < hi) {                                      // wave-uniform (A[s] is one value)
        kk = (int64_t)(int)(v - 1.0);                               // reference :28, truncation toward zero
        for (int e = lane; e < k; e += 64) B[e] = mat[(int64_t)e * ld + kk];
        wave_lds_fence();
        int cnt = 0;
        for (int e = lane; e < k; e += 64) {
          const double b = B[e];
          int rank = 0, ca = 0;
          for (int t = 0; t < k; ++t) {
            ca += (A[t] == b);
            rank += (t < e && B[t] == b);
          }
          cnt += rank < ca;                                         // min multiplicity (std::set_intersection)
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        u = cnt;
        wave_lds_fence();
      }
      if (lane == 0) {
        o_src[r] = u > 0 ? (double)(i + 1) : 0.0;                   // :49
        o_dst[r] = u > 0 ? (double)(kk + 1) : 0.0;                  // :50
        o_w[r] = u > 0 ? (double)u / (2.0 * (double)k - (double)u) : 0.0;   // :51
      }
    }
    wave_lds_fence();
  }
}

// ------------------------------------------------- packed table rows (multi-GPU transport)
// The all-gather of table rows is what bounds the N > 1 path (xGMI), so rows travel bit-packed:
// k ids of ceil(log2(N+1)) bits each + 1 bit for the row's duplicate flag, rounded up to whole
// 32-bit words (k = 30, N = 800 k: 76 B instead of the 128 B row pitch).  One thread per row.
__host__ __device__ inline int id_bits(int64_t N) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) <= N) ++b;
  return b;
}
__host__ __device__ inline int packed_words(int64_t N, int k) { return (k * id_bits(N) + 1 + 31) / 32; }

// 64 rows per workgroup, staged through LDS so that both the table read and the packed write are
// contiguous runs; every thread assembles whole 32-bit output words.
constexpr int PACK_ROWS = 64;

__global__ __launch_bounds__(256) void k_pack_rows(const uint32_t* __restrict__ table, int64_t n_rows, int k, int kpad, int compact,
                                                   int bits, int wpr, uint32_t* __restrict__ packed, int pitch) {
  extern __shared__ uint32_t s_tb[];                        // PACK_ROWS * row_words words
  const int roww = compact ? kpad / 2 : kpad;               // words of the row that hold its ids (dual rows: the compact part; pitch = 64)
  const int fl = k * bits;                                  // bit position of the duplicate flag
  for (int64_t row0 = (int64_t)blockIdx.x * PACK_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * PACK_ROWS) {
    const int rows_here = (int)((n_rows - row0) < PACK_ROWS ? (n_rows - row0) : PACK_ROWS);
    for (int e = threadIdx.x; e < rows_here * roww; e += 256) s_tb[e] = table[(row0 + e / roww) * pitch + e % roww];
    __syncthreads();
    for (int e = threadIdx.x; e < rows_here * wpr; e += 256) {
      const int rr = e / wpr, w = e % wpr;
      const uint32_t* row = s_tb + rr * roww;
      const int lo = 32 * w, hi = lo + 32;
      uint32_t v = 0;
      for (int j = lo / bits; j < k && j * bits < hi; ++j) {
        const uint32_t id = row_slot_id(row, j, kpad, compact != 0);
        const int pos = j * bits - lo;
        v |= pos >= 0 ? id << pos : id >> (-pos);
      }
      if (fl >= lo && fl < hi) v |= (row_dup_flag(row, kpad, compact != 0) ? 1u : 0u) << (fl - lo);
      packed[row0 * wpr + e] = v;
    }
    __syncthreads();
  }
}

// 64 rows per workgroup: the packed words are staged in LDS with coalesced loads, every thread then
// assembles whole table words, so the table is written as contiguous runs.
constexpr int UNPACK_ROWS = 64;

__device__ inline uint32_t packed_id(const uint32_t* in, int j, int bits, int wpr, uint32_t mask) {
  const int off = j * bits, w = off >> 5, sh = off & 31;
  unsigned long long v = in[w];
  if (w + 1 < wpr) v |= (unsigned long long)in[w + 1] << 32;
  return (uint32_t)(v >> sh) & mask;
}

__global__ __launch_bounds__(256) void k_unpack_rows(const uint32_t* __restrict__ packed, int64_t n_rows, int k, int kpad, int compact,
                                                     int bits, int wpr, uint32_t* __restrict__ table, int pitch) {
  extern __shared__ uint32_t s_pk[];                        // UNPACK_ROWS * wpr words
  __shared__ uint32_t s_ids[4][64], s_prow[4][32];          // dual rows: a wave's row as ids / its planar part
  const uint32_t mask = (uint32_t)(((unsigned long long)1 << bits) - 1ull);
  const int fl_w = (k * bits) >> 5, fl_b = (k * bits) & 31; // position of the duplicate flag
  const int roww = compact ? kpad / 2 : kpad;
  const int hiw = kpad / 2 - kpad / 32;                     // compact rows: first word of high bits
  for (int64_t row0 = (int64_t)blockIdx.x * UNPACK_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * UNPACK_ROWS) {
    const int rows_here = (int)((n_rows - row0) < UNPACK_ROWS ? (n_rows - row0) : UNPACK_ROWS);
    for (int e = threadIdx.x; e < rows_here * wpr; e += 256) s_pk[e] = packed[row0 * wpr + e];
    __syncthreads();
    for (int e = threadIdx.x; e < rows_here * roww; e += 256) {
      const int rr = e / roww, j = e % roww;
      const uint32_t* in = s_pk + rr * wpr;
      const uint32_t dupf = ((in[fl_w] >> fl_b) & 1u) << 31;
      uint32_t v = 0;
      if (!compact) {
        if (j < k) v = packed_id(in, j, bits, wpr, mask);
        if (j == 0) v |= dupf;
      } else if (j < hiw) {
        if (2 * j < k) v = scramble16(packed_id(in, 2 * j, bits, wpr, mask) & 0xFFFFu);
        if (2 * j + 1 < k) v |= scramble16(packed_id(in, 2 * j + 1, bits, wpr, mask) & 0xFFFFu) << 16;
      } else {
        const int j0 = (j - hiw) * 32;
        for (int b = 0; b < 32 && j0 + b < k; ++b) v |= ((packed_id(in, j0 + b, bits, wpr, mask) >> 16) & 1u) << b;
        if (j == roww - 1) v |= dupf;
      }
      table[(row0 + rr) * pitch + j] = v;
    }
    if (pitch == DUAL_PITCH && compact && kpad == 64) {     // dual rows: the planar part of every row, a wave per row
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int rr = wave; rr < rows_here; rr += 4) {
        const uint32_t* in = s_pk + rr * wpr;
        s_ids[wave][lane] = lane < k ? packed_id(in, lane, bits, wpr, mask) : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        planar_row_build(s_ids[wave], k, ((in[fl_w] >> fl_b) & 1u) != 0u, s_prow[wave], lane);
        if (lane < 32) table[(row0 + rr) * pitch + 32 + lane] = s_prow[wave][lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    __syncthreads();
  }
}

template <typename T>
int launch_ingest(gficf_ctx* ctx, const T* d_idx, int64_t n_rows, int k, int64_t ld, int64_t N_total,
                  uint32_t* table, int zero_ok = 0) {
  if (sorted_fmt(k)) return launch_ingest_sorted<T>(ctx, d_idx, n_rows, k, ld, N_total, table, zero_ok);
  const TableFmt f = table_fmt(N_total, k);
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  const int64_t tiles2 = gficf_ceil_div(n_rows, INGEST2_ROWS);
  const unsigned grid2 = (unsigned)(tiles2 < cap ? tiles2 : cap);
  const int64_t tiles = gficf_ceil_div(n_rows, INGEST_ROWS);
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
  static const bool use_reg = getenv("GFICF_JACCARD_INGEST_REG") != nullptr;     // test hook: the one-thread-per-cell variant
  // rows taken to hold distinct ids (gficf_ctx_set_jaccard_distinct): no duplicate scan; not for sub-problems in local ids
  const bool scan = !(ctx->jaccard_assume_distinct && zero_ok == 0);
#define LAUNCH_INGEST_REG(KP, CM)                                                                                          \
  do {                                                                                                                     \
    if (use_reg)                                                                                                           \
      hipLaunchKernelGGL((k_ingest_reg<T, KP, CM>), dim3(grid2), dim3(INGEST2_ROWS), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, \
                         table, ctx->d_status, zero_ok);                                                                   \
    else if (scan)                                                                                                         \
      hipLaunchKernelGGL((k_ingest_tile<T, KP, CM>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total,  \
                         table, ctx->d_status, zero_ok, gficf_halo_map{});                                                 \
    else                                                                                                                   \
      hipLaunchKernelGGL((k_ingest_tile<T, KP, CM, false, false>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, \
                         table, ctx->d_status, zero_ok, gficf_halo_map{});                                                 \
  } while (0)
#define LAUNCH_INGEST(KP, CM)                                                                                   \
  hipLaunchKernelGGL((k_ingest<T, KP, CM>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, \
                     table, ctx->d_status, zero_ok, scan ? 1 : 0)
  switch (f.kpad) {
    case 16: LAUNCH_INGEST_REG(16, false); break;
    case 32: if (f.compact) LAUNCH_INGEST_REG(32, true); else LAUNCH_INGEST_REG(32, false); break;
    case 64:
      if (f.dual) {                                          // dual rows: the tile kernel writes both parts of every row
        if (scan)
          hipLaunchKernelGGL((k_ingest_tile<T, 64, true, false, true, true>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total,
                             table, ctx->d_status, zero_ok, gficf_halo_map{});
        else
          hipLaunchKernelGGL((k_ingest_tile<T, 64, true, false, false, true>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total,
                             table, ctx->d_status, zero_ok, gficf_halo_map{});
      } else if (f.compact) LAUNCH_INGEST_REG(64, true); else LAUNCH_INGEST_REG(64, false);
      break;
    case 128: if (f.compact) LAUNCH_INGEST(128, true); else LAUNCH_INGEST(128, false); break;
    default: if (f.compact) LAUNCH_INGEST(256, true); else LAUNCH_INGEST(256, false); break;
  }
#undef LAUNCH_INGEST
#undef LAUNCH_INGEST_REG
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

// workgroups per CU for an edge kernel: as many as fit, up to 8 (measured at 100 k x 30, tools/r02_sweep.sh: 4 per CU 44-47 us,
// 6: 43-46, 8: 41-42, 10-12: 40-42 — the row gathers are latency-bound, more waves keep more of them in flight).  Asked of the
// runtime for the kernel that is launched (the pipelined and the general form differ in registers), once per kernel.
template <typename K>
int edge_blocks_per_cu(K kernel, int threads, size_t lds_bytes, std::atomic<int>& cache, int* out) {
  int v = cache.load(std::memory_order_relaxed);
  if (v == 0) {
    int nb = 0;
    GFICF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds_bytes));
    v = nb > 8 ? 8 : nb > 0 ? nb : 1;
    if (const char* e = getenv("GFICF_JACCARD_BLOCKS_PER_CU")) {   // tuning knob
      const int t = atoi(e);
      if (t > 0) v = t;
    }
    cache.store(v, std::memory_order_relaxed);   // same value whoever computes it
  }
  *out = v;
  return GFICF_OK;
}

template <int KPAD, bool BIG, bool CMP, int OUT, bool MAP>
int launch_edges_m(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  using C = JCfg<KPAD, CMP>;
  // grid = what is resident at once (occupancy x CUs); waves stride over the cells
  static std::atomic<int> bpc_general{0}, bpc_pipe{0}, bpc_pipe_nob16{0};
  constexpr size_t lds_bytes = edges_lds_bytes<KPAD, CMP>();
  int blocks_per_cu = 1;
  static const bool no_pipe = getenv("GFICF_JACCARD_NO_PIPE") != nullptr;        // test hook: the one-cell-at-a-time kernel for every k
  if constexpr (C::EPL == 1 && C::SPQ <= 4) {
    if (!no_pipe) {
      const bool nob16 = CMP && N < 65536;     // no id carries bit 16: the kernel variant that does not look for it (configs 1-3 of BASELINE.json)
      int rc;
      if constexpr (!MAP) {
        if (o.dup_status != nullptr) {           // a table without row flags (gficf_ctx_set_jaccard_distinct)
          static std::atomic<int> bpc_nf{0}, bpc_nf_nob16{0};
          if (nob16) rc = edge_blocks_per_cu(k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, !CMP, false, true>, C::WAVES * 64, lds_bytes, bpc_nf_nob16, &blocks_per_cu);
          else rc = edge_blocks_per_cu(k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, true, false, true>, C::WAVES * 64, lds_bytes, bpc_nf, &blocks_per_cu);
          if (rc) return rc;
          const int64_t cap_n = (int64_t)ctx->num_cus * blocks_per_cu;
          const int64_t need_n = gficf_ceil_div(gficf_ceil_div(ce - cb, 4), C::WAVES);
          unsigned grid_n = (unsigned)(need_n < cap_n ? need_n : cap_n);
          if (o.xcd && grid_n > 8) grid_n = (grid_n + 7u) & ~7u;
          if (nob16)
            hipLaunchKernelGGL((k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, !CMP, false, true>), dim3(grid_n), dim3(C::WAVES * 64), lds_bytes, ctx->stream,
                               table, N, k, cb, ce, o);
          else
            hipLaunchKernelGGL((k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, true, false, true>), dim3(grid_n), dim3(C::WAVES * 64), lds_bytes, ctx->stream,
                               table, N, k, cb, ce, o);
          GFICF_HIP_CHECK(hipGetLastError());
          return GFICF_OK;
        }
      }
      if (nob16) rc = edge_blocks_per_cu(k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, !CMP, MAP>, C::WAVES * 64, lds_bytes, bpc_pipe_nob16, &blocks_per_cu);
      else rc = edge_blocks_per_cu(k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, true, MAP>, C::WAVES * 64, lds_bytes, bpc_pipe, &blocks_per_cu);
      if (rc) return rc;
      const int64_t cap = (int64_t)ctx->num_cus * blocks_per_cu;
      const int64_t need4 = gficf_ceil_div(gficf_ceil_div(ce - cb, 4), C::WAVES);       // a wave takes four cells at a time
      unsigned grid4 = (unsigned)(need4 < cap ? need4 : cap);
      if (o.xcd && grid4 > 8) grid4 = (grid4 + 7u) & ~7u;       // (surplus workgroups find no cell and leave)
      if (nob16)
        hipLaunchKernelGGL((k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, !CMP, MAP>), dim3(grid4), dim3(C::WAVES * 64), lds_bytes, ctx->stream,
                           table, N, k, cb, ce, o);
      else
        hipLaunchKernelGGL((k_jaccard_edges_pipe<KPAD, BIG, CMP, OUT, true, MAP>), dim3(grid4), dim3(C::WAVES * 64), lds_bytes, ctx->stream, table,
                           N, k, cb, ce, o);
      GFICF_HIP_CHECK(hipGetLastError());
      return GFICF_OK;
    }
  }
  const int rc = edge_blocks_per_cu(k_jaccard_edges<KPAD, BIG, CMP, OUT, MAP>, C::WAVES * 64, lds_bytes, bpc_general, &blocks_per_cu);
  if (rc) return rc;
  const int64_t blocks_needed = gficf_ceil_div(ce - cb, C::WAVES);
  const int64_t cap = (int64_t)ctx->num_cus * blocks_per_cu;
  unsigned grid = (unsigned)(blocks_needed < cap ? blocks_needed : cap);
  if (o.xcd && grid > 8) grid = (grid + 7u) & ~7u;
  hipLaunchKernelGGL((k_jaccard_edges<KPAD, BIG, CMP, OUT, MAP>), dim3(grid), dim3(C::WAVES * 64), lds_bytes, ctx->stream, table,
                     N, k, cb, ce, o);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int KPAD, bool BIG, bool CMP, int OUT>
int launch_edges_o(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if constexpr (OUT != OUT_U16) {              // (the counts-only output has no neighbour column)
    if (o.l2g) return launch_edges_m<KPAD, BIG, CMP, OUT, true>(ctx, table, N, k, cb, ce, o);
  }
  return launch_edges_m<KPAD, BIG, CMP, OUT, false>(ctx, table, N, k, cb, ce, o);
}

template <int KPAD, bool BIG, bool CMP>
int launch_edges_t(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if (o.u16) return launch_edges_o<KPAD, BIG, CMP, OUT_U16>(ctx, table, N, k, cb, ce, o);
  if (o.u) return launch_edges_o<KPAD, BIG, CMP, OUT_RMAT_U>(ctx, table, N, k, cb, ce, o);
  return launch_edges_o<KPAD, BIG, CMP, OUT_RMAT>(ctx, table, N, k, cb, ce, o);
}

// dual rows (32 < k <= 55, N <= 131070): the bit-set kernel
template <int NST, int OUT, bool MAP>
int launch_bits_m(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  static std::atomic<int> bpc{0};
  static std::atomic<bool> attr_set[64];
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_jaccard_edges_bits<NST, OUT, MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BITS_LDS_BYTES));
    attr_set[ctx->device & 63] = true;
  }
  int blocks_per_cu = 1;
  const int rc = edge_blocks_per_cu(k_jaccard_edges_bits<NST, OUT, MAP>, BITS_WAVES * 64, BITS_LDS_BYTES, bpc, &blocks_per_cu);
  if (rc) return rc;
  const int64_t need = gficf_ceil_div(ce - cb, BITS_WAVES);
  const int64_t cap = (int64_t)ctx->num_cus * blocks_per_cu;
  unsigned grid = (unsigned)(need < cap ? need : cap);
  if (o.xcd && grid > 8) grid = (grid + 7u) & ~7u;
  hipLaunchKernelGGL((k_jaccard_edges_bits<NST, OUT, MAP>), dim3(grid), dim3(BITS_WAVES * 64), BITS_LDS_BYTES, ctx->stream, table, N, k, cb, ce, o);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int NST>
int launch_bits(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if (o.u16) return launch_bits_m<NST, OUT_U16, false>(ctx, table, N, k, cb, ce, o);
  if (o.u) return o.l2g ? launch_bits_m<NST, OUT_RMAT_U, true>(ctx, table, N, k, cb, ce, o) : launch_bits_m<NST, OUT_RMAT_U, false>(ctx, table, N, k, cb, ce, o);
  return o.l2g ? launch_bits_m<NST, OUT_RMAT, true>(ctx, table, N, k, cb, ce, o) : launch_bits_m<NST, OUT_RMAT, false>(ctx, table, N, k, cb, ce, o);
}

template <int KPAD>
int launch_edges(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if constexpr (KPAD == 64) {
    if (table_fmt(N, k).dual) {
      const int nst = (k + 7) / 8;                           // gather steps of a cell, 8 neighbour rows each
      return nst <= 5 ? launch_bits<5>(ctx, table, N, k, cb, ce, o) : nst == 6 ? launch_bits<6>(ctx, table, N, k, cb, ce, o)
                                                                                : launch_bits<7>(ctx, table, N, k, cb, ce, o);
    }
  }
  // 32-bit byte offsets and the 24-bit hash multiply need table < 4 GiB and ids < 2^24
  static const bool force_big = getenv("GFICF_JACCARD_FORCE_BIG") != nullptr;   // test hook for the 64-bit variant
  const bool big = force_big || N >= (1ll << 24) || N * (int64_t)KPAD * 4 >= (1ll << 32);
  if (KPAD >= 32 && table_fmt(N, k).compact)          // compact rows: N < 2^17, never the 64-bit variant (the test hook switches them off)
    return launch_edges_t<KPAD, false, KPAD >= 32>(ctx, table, N, k, cb, ce, o);
  return big ? launch_edges_t<KPAD, true, false>(ctx, table, N, k, cb, ce, o)
             : launch_edges_t<KPAD, false, false>(ctx, table, N, k, cb, ce, o);
}

int launch_edges_k(gficf_ctx* ctx, const uint32_t* t, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if (sorted_fmt(k)) return launch_edges_sorted(ctx, t, N, k, cb, ce, o);      // exact for every row: no flags, no deferred report
  if (const char* e = getenv("GFICF_JACCARD_XCD")) o.xcd = atoi(e) != 0 ? 1u : 0u;     // A/B switch, read per call
  if (ctx->jaccard_assume_distinct) o.dup_status = ctx->d_status;                      // the table may carry no duplicate flags
  switch (kpad_for(k)) {
    case 16: return launch_edges<16>(ctx, t, N, k, cb, ce, o);
    case 32: return launch_edges<32>(ctx, t, N, k, cb, ce, o);
    case 64: return launch_edges<64>(ctx, t, N, k, cb, ce, o);
    case 128: return launch_edges<128>(ctx, t, N, k, cb, ce, o);
    default: return launch_edges<256>(ctx, t, N, k, cb, ce, o);
  }
}

// edges from which gficf_jaccard_host returns through uint16 counts + a host-side expansion instead of copying the 24 B/edge matrix
// back (GFICF_JACCARD_HOST_COMPACT_MIN_EDGES in the environment; 0 = always, a huge number = never)
int64_t host_compact_min_edges() {                              // (read per call: an A/B switch inside one process; a call is milliseconds)
  const char* e = getenv("GFICF_JACCARD_HOST_COMPACT_MIN_EDGES");
  return e ? (int64_t)atoll(e) : (int64_t)(1 << 20);          // measured crossover ~1 M edges (profiles/r05_host_compact_ab.txt)
}

int check_nk(int64_t N, int k) {
  if (N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N = %lld is negative", (long long)N);
  if (k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "k = %d is negative", k);
  if (k > SORTED_MAX_K)
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d neighbours per cell: this build handles at most %d (uint16 intersection counts; the reference has no limit)", k,
               SORTED_MAX_K);
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  return GFICF_OK;
}

}  // namespace

/* The host entries take rows to hold distinct ids (no duplicate scan in the ingest, gficf_ctx_set_jaccard_distinct) and re-run
 * the exact sequence by themselves when the edge kernel reports a row that does not: the reference's result for every input,
 * the scan's time saved for every input a kNN search produces.  GFICF_JACCARD_SCAN_DUPS in the environment: always scan. */
template <typename F>
static int distinct_first(gficf_ctx* ctx, F&& body) {
  if (!ctx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ctx is NULL");
  static const bool always_scan = getenv("GFICF_JACCARD_SCAN_DUPS") != nullptr;
  const int saved = ctx->jaccard_assume_distinct;
  ctx->jaccard_assume_distinct = always_scan ? 0 : 1;
  int rc = body();
  if (rc == GFICF_ERR_DUPLICATE_IDS || rc == GFICF_ERR_SET_OVERFLOW) {
    // a repeated id: the exact sequence (duplicate scan, multiset counts).  A hash set that overflowed (ids spread uniformly at k near 256):
    // the sorted-row path, exact for every input and ~13 x faster there than the exact hash-set sequence, whose cells with an overflowing set
    // fall to all-pairs (5 000 x 256: 1.4 ms against 22 ms, profiles/r05_bigk_time.txt); below k = 57, where there is no sorted path, the exact one.
    ctx->jaccard_assume_distinct = 0;
    g_force_sorted = rc == GFICF_ERR_SET_OVERFLOW ? 1 : 0;
    ctx->quiet_rerun = 1;                      // (the banner lines have been printed)
    rc = body();
    ctx->quiet_rerun = 0;
    g_force_sorted = 0;
  }
  ctx->jaccard_assume_distinct = saved;
  return rc;
}

extern "C" {

int gficf_jaccard_kpad(int k) { return (k < 0 || k > SORTED_MAX_K) ? -1 : sorted_fmt(k) ? 2 * sorted_kp(k) : kpad_for(k); }

int gficf_jaccard_row_words(int64_t N_total, int k) {
  if (N_total < 0 || N_total > 0x7FFFFFFFll || k < 0 || k > SORTED_MAX_K) return -1;
  return table_fmt(N_total, k).row_words;
}

int gficf_jaccard_ingest_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t n_rows, int k,
                                int64_t ld, int64_t N_total, int32_t* d_table_rows) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0 || n_rows > N_total) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_rows = %lld outside [0, N_total]", (long long)n_rows);
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_idx || !d_table_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ld < n_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < n_rows = %lld", (long long)ld, (long long)n_rows);
  if (idx_is_f64) return launch_ingest<double>(ctx, (const double*)d_idx, n_rows, k, ld, N_total, (uint32_t*)d_table_rows);
  return launch_ingest<int32_t>(ctx, (const int32_t*)d_idx, n_rows, k, ld, N_total, (uint32_t*)d_table_rows);
}

int gficf_jaccard_edges_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin,
                               int64_t cell_end, double* d_src, double* d_dst, double* d_w, int32_t* d_u) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (cell_begin < 0 || cell_end < cell_begin || cell_end > N)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "cell range [%lld, %lld) outside [0, %lld]", (long long)cell_begin, (long long)cell_end, (long long)N);
  if (cell_end == cell_begin || k == 0) return GFICF_OK;
  if (!d_table || !d_src || !d_dst || !d_w) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  EdgeOut o{d_src, d_dst, d_w, d_u, nullptr, 0};
  return launch_edges_k(ctx, (const uint32_t*)d_table, N, k, cell_begin, cell_end, o);
}

/* Rows of a sharded sub-problem in local ids (halo.hip): n_ext rows, ids in [1, n_ext], 0 = no id in the slot. */
int gficf_jaccard_ingest_local_device(gficf_ctx* ctx, const int32_t* d_idx_ext, int64_t n_ext, int k, int64_t ld, int32_t* d_table) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(n_ext, k);
  if (rc) return rc;
  if (n_ext == 0 || k == 0) return GFICF_OK;
  if (!d_idx_ext || !d_table) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ld < n_ext) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < n_ext = %lld", (long long)ld, (long long)n_ext);
  return launch_ingest<int32_t>(ctx, d_idx_ext, n_ext, k, ld, n_ext, (uint32_t*)d_table, 1);
}

/* relabel + ingest in one launch (k <= 64): the sub-problem's table straight from the block's global ids, the reply slots and
 * the plan's workspace (csrc/halo.hip); also writes d_l2g.  Returns GFICF_ERR_UNSUPPORTED for k > 64 (the caller then runs
 * gficf_jaccard_halo_relabel_device + gficf_jaccard_ingest_local_device).
 * Three forms of the one kernel: rows [0, n_ext) after both exchanges (the peer form, gficf_jaccard_halo_ingest_peer_device: nothing is exchanged there); the OWN cells' rows
 * together with the owner-side serve step, between the two exchanges (gficf_jaccard_halo_serve_ingest_device: the own rows need
 * the plan, not the replies); the halo slots' rows behind the second exchange (gficf_jaccard_halo_ingest_slots_device: a few
 * hundred rows — what is left on the critical path between the replies and the edge kernel). */
static int halo_ingest_launch(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total, int64_t cell_begin,
                              int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out, const int32_t* d_rows_in,
                              int32_t* d_table, int32_t* d_l2g, int64_t row_begin, int64_t row_end, const int32_t* d_req_in, int64_t n_req,
                              int32_t* d_rows_out, const int32_t* const* peer_idx = nullptr, const int64_t* peer_ld = nullptr) {
  GFICF_CTX_ENTER(ctx);
  if (n_local < 0 || k < 0 || N_total < 0 || P < 1 || cap < 1 || rows_per_rank < 1 || cell_begin < 0 || cell_begin + n_local > N_total)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo ingest: sizes out of range");
  const int64_t n_ext = n_local + (int64_t)P * cap;
  int rc = check_nk(n_ext, k);
  if (rc) return rc;
  if (k > 64) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "halo ingest: the fused form covers k <= 64");
  // GFICF_JACCARD_SORTED_FROM = 57 .. 64 moves k into the sorted-row format, which this launch does not write (it would fill hash-set rows
  // into a table the edge launcher reads as sorted ones): the caller takes relabel + gficf_jaccard_ingest_local_device, as for k > 64
  if (sorted_fmt(k)) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "halo ingest: k = %d is in the sorted-row format (GFICF_JACCARD_SORTED_FROM): the fused form does not cover it", k);
  if (n_ext == 0 || k == 0) return GFICF_OK;
  if (!d_ws || !d_req_out || (row_end > n_local && !d_rows_in && !peer_idx) || !d_table || !d_l2g || (n_local > 0 && !d_idx)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (peer_idx) {
    if (!peer_ld) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer (peer_ld)");
    if (P > GFICF_HALO_MAX_PEERS) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "halo ingest, peer form: %d owners, at most %d", P, GFICF_HALO_MAX_PEERS);
    for (int o = 0; o < P; ++o) {
      const int64_t ob = (int64_t)o * rows_per_rank, on = ob >= N_total ? 0 : (N_total - ob < rows_per_rank ? N_total - ob : rows_per_rank);
      if (on > 0 && (!peer_idx[o] || peer_ld[o] < on)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo ingest, peer form: block of owner %d missing or its ld below its %lld rows", o, (long long)on);
    }
  }
  if (n_req > 0 && (!d_req_in || !d_rows_out)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer (serve step)");
  if (n_local > 0 && ld < n_local) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld < n_local");
  const int64_t wpo = gficf_halo_wpo(rows_per_rank);                   // layout of gficf_jaccard_halo_workspace_bytes
  const int64_t gcap = (int64_t)ctx->num_cus * 8;
  const int64_t tiles = gficf_ceil_div(row_end - row_begin, INGEST_ROWS);
  const unsigned grid_i = (unsigned)(tiles < gcap ? tiles : gcap);
  int64_t sb = n_req > 0 ? gficf_ceil_div(gficf_halo_serve_items(n_req, k), 256) : 0;
  if (sb > (int64_t)ctx->num_cus * 2) sb = (int64_t)ctx->num_cus * 2;
  if (grid_i == 0 && sb == 0) return GFICF_OK;
  gficf_halo_map hm{(const uint2*)((const char*)d_ws + (((size_t)P * (size_t)wpo * 4 + 255) & ~(size_t)255)), d_req_out, d_rows_in, d_l2g, n_local,
                    N_total, cell_begin, rows_per_rank, cap, (uint32_t)wpo, row_begin, row_end, d_req_in, n_req, d_rows_out, (int)sb,
                    row_begin >= n_local ? 1 : 0};
  if (peer_idx) {
    hm.peer_n = P;
    hm.skip_empty = 1;                                       // (also when the launch covers the own cells too: a tile of slots nobody asked for is never referred to)
    for (int o = 0; o < P; ++o) { hm.peer_idx[o] = peer_idx[o]; hm.peer_ld[o] = peer_ld[o]; }
  }
  const TableFmt f = table_fmt(n_ext, k);
  const unsigned grid = grid_i + (unsigned)sb;
  // rows taken to hold distinct ids (gficf_ctx_set_jaccard_distinct): no duplicate scan here either — a rank's own rows are
  // inserted by its mapped edge kernel, which reports a repeated id; the halo rows are own rows of the ranks that sent them
  const bool scan = !ctx->jaccard_assume_distinct;
#define LAUNCH_HALO_INGEST(KP, CM)                                                                                                \
  do {                                                                                                                            \
    if (scan)                                                                                                                     \
      hipLaunchKernelGGL((k_ingest_tile<int32_t, KP, CM, true>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, row_end, k, ld, n_ext, \
                         (uint32_t*)d_table, ctx->d_status, 1, hm);                                                               \
    else                                                                                                                          \
      hipLaunchKernelGGL((k_ingest_tile<int32_t, KP, CM, true, false>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, row_end, k, ld, \
                         n_ext, (uint32_t*)d_table, ctx->d_status, 1, hm);                                                        \
  } while (0)
  switch (f.kpad) {
    case 16: LAUNCH_HALO_INGEST(16, false); break;
    case 32: if (f.compact) LAUNCH_HALO_INGEST(32, true); else LAUNCH_HALO_INGEST(32, false); break;
    default:
      if (f.dual) {
        if (scan)
          hipLaunchKernelGGL((k_ingest_tile<int32_t, 64, true, true, true, true>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, row_end, k, ld, n_ext,
                             (uint32_t*)d_table, ctx->d_status, 1, hm);
        else
          hipLaunchKernelGGL((k_ingest_tile<int32_t, 64, true, true, false, true>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, row_end, k, ld, n_ext,
                             (uint32_t*)d_table, ctx->d_status, 1, hm);
      } else if (f.compact) LAUNCH_HALO_INGEST(64, true); else LAUNCH_HALO_INGEST(64, false);
      break;
  }
#undef LAUNCH_HALO_INGEST
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_halo_serve_ingest_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                           int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                           const int32_t* d_req_in, int64_t n_req, int32_t* d_rows_out, int32_t* d_table, int32_t* d_l2g) {
  if (n_req < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "halo serve: negative size");
  return halo_ingest_launch(ctx, d_idx, n_local, k, ld, N_total, cell_begin, P, rows_per_rank, cap, d_ws, d_req_out, nullptr, d_table, d_l2g, 0, n_local,
                            d_req_in, n_req, d_rows_out);
}

int gficf_jaccard_halo_ingest_slots_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                           int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                           const int32_t* d_rows_in, int32_t* d_table, int32_t* d_l2g) {
  return halo_ingest_launch(ctx, d_idx, n_local, k, ld, N_total, cell_begin, P, rows_per_rank, cap, d_ws, d_req_out, d_rows_in, d_table, d_l2g, n_local,
                            n_local + (int64_t)P * cap, nullptr, 0, nullptr);
}

/* The whole table of a sub-problem with NO exchange, in one launch (one process, every device maps the others' memory:
 * gficf_multi_jaccard_halo_device): the own cells' rows from the block, and the row of every requested id read where it lies —
 * d_peer_idx[o] is owner o's block of global ids (a device pointer this device can dereference, column-major with leading
 * dimension peer_ld[o]; blocks of equal pitch rows_per_rank), P <= 16 owners.  Needs the plan only. */
int gficf_jaccard_halo_ingest_peer_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                          int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                          const int32_t* const* d_peer_idx, const int64_t* peer_ld, int32_t* d_table, int32_t* d_l2g) {
  if (!d_peer_idx || !peer_ld) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer (peer blocks)");
  return halo_ingest_launch(ctx, d_idx, n_local, k, ld, N_total, cell_begin, P, rows_per_rank, cap, d_ws, d_req_out, nullptr, d_table, d_l2g, 0,
                            n_local + (int64_t)P * cap, nullptr, 0, nullptr, d_peer_idx, peer_ld);
}

/* Edges of the first n_cells rows of such a table; column 1 = src_offset + cell + 1, column 2 = d_l2g[local id - 1]. */
int gficf_jaccard_edges_mapped_device(gficf_ctx* ctx, const int32_t* d_table, int64_t n_ext, int k, int64_t n_cells, int64_t src_offset,
                                      const int32_t* d_l2g, double* d_src, double* d_dst, double* d_w, int32_t* d_u) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(n_ext, k);
  if (rc) return rc;
  if (n_cells < 0 || n_cells > n_ext || src_offset < 0 || src_offset + n_cells > 0x7FFFFFFFll)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_cells = %lld / src_offset = %lld out of range", (long long)n_cells, (long long)src_offset);
  if (n_cells == 0 || k == 0) return GFICF_OK;
  if (!d_table || !d_l2g || !d_src || !d_dst || !d_w) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  EdgeOut o{d_src, d_dst, d_w, d_u, nullptr, 0, d_l2g, (uint32_t)src_offset};
  return launch_edges_k(ctx, (const uint32_t*)d_table, n_ext, k, 0, n_cells, o);
}

int gficf_jaccard_one_launch(gficf_ctx* ctx, int64_t N, int k) { return (ctx && direct_applies(ctx, N, k)) ? 1 : 0; }

int gficf_jaccard_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                         int32_t* d_table_ws, double* d_rmat, int32_t* d_u) {
  const int64_t E = N * (int64_t)k;
  if (ctx && direct_applies(ctx, N, k) && d_idx && d_rmat && ld >= N) {
    // a small problem whose rows are taken to hold distinct ids: one launch straight from the input, no table (d_table_ws unused)
    GFICF_CTX_ENTER(ctx);
    EdgeOut o{d_rmat, d_rmat + E, d_rmat + 2 * E, d_u, nullptr, 0};
    return launch_direct(ctx, d_idx, idx_is_f64, N, k, ld, o);
  }
  int rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, d_table_ws);
  if (rc) return rc;
  return gficf_jaccard_edges_device(ctx, d_table_ws, N, k, 0, N, d_rmat, d_rmat + E, d_rmat + 2 * E, d_u);
}

static int gficf_jaccard_counts_host_body(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, uint16_t* u);

static int gficf_jaccard_host_body(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                       double* rmat, int print_output) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (print_output && !ctx->quiet_rerun) gficf_print(ctx, "Running Parallell Jaccard Coefficient Estimation...\n");  // reference :63
  const int64_t E = N * (int64_t)k;
  bool trunc_path = false;
  if (E > 0 && idx && idx_is_f64 && ctx->jaccard_trunc && ld >= N) {
    // strict drop-in mode: does the matrix hold a non-integer value at all?  (host scan, only with the option on)
    const double* const m = (const double*)idx;
    for (int j = 0; j < k && !trunc_path; ++j)
      for (int64_t i = 0; i < N; ++i) {
        const double v = m[(int64_t)j * ld + i];
        if (v != std::trunc(v)) { trunc_path = true; break; }     // (NaN too: the kernel rejects it)
      }
  }
  if (trunc_path) {
    if (!rmat) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
    if (k > GFICF_JACCARD_MAX_K)
      GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "non-integer ids (GFICF_HIP_TRUNCATE_IDS) with k = %d: the truncating kernel covers k <= %d", k, GFICF_JACCARD_MAX_K);
    void* d_idx = nullptr;
    double* d_rmat = nullptr;
    hipError_t e = gficf_pool_get(ctx, 0, sizeof(double) * (size_t)ld * (size_t)k, &d_idx);
    if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(double) * 3 * (size_t)E, (void**)&d_rmat);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, sizeof(double) * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
    rc = GFICF_OK;
    if (e == hipSuccess) {
      int64_t blocks = gficf_ceil_div(N, 4);
      if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
      hipLaunchKernelGGL(k_jaccard_trunc_f64, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const double*)d_idx, N, k, ld, d_rmat, d_rmat + E,
                         d_rmat + 2 * E, ctx->d_status);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(rmat, d_rmat, sizeof(double) * 3 * (size_t)E, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) rc = gficf_ctx_sync(ctx);
      else (void)hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_host: %s", hipGetErrorString(e));
    if (rc) return rc;
  } else if (E > 0 && idx && rmat && ld >= N && E >= host_compact_min_edges()) {
    // ---- the compact return (round 5): the 24 B row of an edge is a function of (i, idx[i,j], u), and idx is on the host already.
    // So 2 B per edge cross PCIe instead of 24 (uint16 counts into pinned staging) and the host cores write the reference's matrix
    // (gficf_jaccard_expand_host, several threads; their first touch of a freshly allocated R matrix runs in parallel too):
    // Into a FRESH malloc'ed result matrix (what R hands over) 100 k x 30 takes 4.9 ms per call with the matrix copied back and 2.0 ms this
    // way, 1 M x 30 54 ms and 8.7 ms; below ~1 M edges the copy wins (profiles/r05_host_compact_ab.txt).  Same bits: the weights are the
    // reference's own division u / (2.0 * k - u) on the host.
    void* h_u = nullptr;
    const hipError_t e = gficf_host_stage_get(ctx, sizeof(uint16_t) * (size_t)E, &h_u);
    if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_host: %s", hipGetErrorString(e));
    rc = gficf_jaccard_counts_host_body(ctx, idx, idx_is_f64, N, k, ld, (uint16_t*)h_u);      // upload, kernels, counts into the pinned staging, sync
    if (rc) return rc;
    // threads of the expansion: one per ~250 k edges, at most 32 and at most the machine's (starting a thread costs tens of microseconds)
    const unsigned hw = std::thread::hardware_concurrency();
    int64_t nth = E / 250000;
    if (nth > 32) nth = 32;
    if (hw && nth > (int64_t)hw) nth = hw;
    if (nth < 2) nth = 2;
    rc = gficf_jaccard_expand_host(idx, idx_is_f64, N, k, ld, (const uint16_t*)h_u, rmat, (int)nth);
    if (rc) return rc;
  } else if (E > 0) {
    if (!idx || !rmat) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
    if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
    const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
    const int roww = table_fmt(N, k).row_words;
    void* d_idx = nullptr;
    int32_t* d_table = nullptr;
    double* d_rmat = nullptr;
    // device scratch comes from the context's grow-only pool: no hipMalloc/hipFree per call
    hipError_t e = gficf_pool_get(ctx, 0, esz * (size_t)ld * (size_t)k, &d_idx);
    if (e == hipSuccess) e = gficf_pool_get(ctx, 1, sizeof(int32_t) * (size_t)N * (size_t)roww, (void**)&d_table);
    if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(double) * 3 * (size_t)E, (void**)&d_rmat);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
    rc = GFICF_OK;
    if (e == hipSuccess) {
      rc = gficf_jaccard_device(ctx, d_idx, idx_is_f64, N, k, ld, d_table, d_rmat, nullptr);
      gficf_advise_hugepages(rmat, sizeof(double) * 3 * (size_t)E);  // (a fresh R matrix of 4 MB or more: huge pages for its first touch by the copy;
      gficf_prefault(rmat, sizeof(double) * 3 * (size_t)E);          //  >= 16 MB: also touched from several threads while the kernels run)
      if (rc == GFICF_OK) e = hipMemcpyAsync(rmat, d_rmat, sizeof(double) * 3 * (size_t)E, hipMemcpyDeviceToHost, ctx->stream);
      if (rc == GFICF_OK && e == hipSuccess) rc = gficf_ctx_sync(ctx);
      else (void)hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_host: %s", hipGetErrorString(e));
    if (rc) return rc;
  }
  if (print_output) gficf_print(ctx, "Done!!\n");  // reference :77
  return GFICF_OK;
}

/* Compact host return: the intersection counts alone, 2 B per edge instead of the reference's 24 B row (a row is a
 * function of (i, idx[i,j], u): gficf_jaccard_expand_host rebuilds the reference matrix from them on the host). */
static int gficf_jaccard_counts_host_body(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, uint16_t* u) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  const int64_t E = N * (int64_t)k;
  if (E == 0) return GFICF_OK;
  if (!idx || !u) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
  const int roww = table_fmt(N, k).row_words;
  void* d_idx = nullptr;
  int32_t* d_table = nullptr;
  uint16_t* d_u = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, esz * (size_t)ld * (size_t)k, &d_idx);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, sizeof(int32_t) * (size_t)N * (size_t)roww, (void**)&d_table);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(uint16_t) * (size_t)E, (void**)&d_u);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    EdgeOut o{nullptr, nullptr, nullptr, nullptr, d_u, 0};
    if (direct_applies(ctx, N, k)) rc = launch_direct(ctx, d_idx, idx_is_f64, N, k, ld, o);      // small problem: one launch, no table
    else {
      rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, d_table);
      if (!rc) rc = launch_edges_k(ctx, (const uint32_t*)d_table, N, k, 0, N, o);
    }
    if (!rc) e = hipMemcpyAsync(u, d_u, sizeof(uint16_t) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_counts_host: %s", hipGetErrorString(e));
  return rc;
}

/* Host-side expansion of the counts into the reference's (N*k) x 3 matrix (src/rcpp_parallel_jaccard_coeff.cpp:48-52):
 * row i*k+j = (i+1, idx[i,j], u/(2.0*k-u)) when u > 0, zeros otherwise.  Pure host code (no device, no context);
 * n_threads <= 0: one per hardware thread, at most 16.  Inputs are not validated again (the counts come from a call
 * that validated the ids). */
int gficf_jaccard_expand_host(const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, const uint16_t* u, double* rmat,
                              int n_threads) {
  if (N < 0 || k < 0 || k > SORTED_MAX_K) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N or k out of range");
  const int64_t E = N * (int64_t)k;
  if (E == 0) return GFICF_OK;
  if (!idx || !u || !rmat) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  gficf_advise_hugepages(rmat, sizeof(double) * 3 * (size_t)E);     // (a fresh R matrix: its first touch happens in the threads below)
  std::vector<double> lut((size_t)k + 1);
  for (int v = 0; v <= k; ++v) lut[v] = (double)v / (2.0 * (double)k - (double)v);     // reference :51
  unsigned hw = std::thread::hardware_concurrency();
  int64_t nt = n_threads > 0 ? n_threads : (hw ? (hw > 16 ? 16 : hw) : 4);
  if (nt > N) nt = N;
  if (E < (1 << 16)) nt = 1;
  const int32_t* const ii = (const int32_t*)idx;
  const double* const id = (const double*)idx;
  auto work = [&](int64_t c0, int64_t c1) {
    for (int64_t i = c0; i < c1; ++i) {
      for (int j = 0; j < k; ++j) {
        const int64_t r = i * k + j;
        const int v = u[r];
        const double dst = idx_is_f64 ? id[(int64_t)j * ld + i] : (double)ii[(int64_t)j * ld + i];
        rmat[r] = v > 0 ? (double)(i + 1) : 0.0;
        rmat[E + r] = v > 0 ? dst : 0.0;
        rmat[2 * E + r] = v > 0 ? lut[v <= k ? v : k] : 0.0;
      }
    }
  };
  if (nt <= 1) { work(0, N); return GFICF_OK; }
  const int64_t per = (N + nt - 1) / nt;
  gficf_run_shares(nt, [&](int64_t t) {
    const int64_t c0 = t * per, c1 = c0 + per < N ? c0 + per : N;
    if (c0 < c1) work(c0, c1);
  });
  return GFICF_OK;
}

int gficf_jaccard_packed_words(int64_t N_total, int k) {
  if (N_total < 0 || N_total > 0x7FFFFFFFll || k < 0 || k > SORTED_MAX_K) return -1;
  if (sorted_fmt(k)) return table_fmt(N_total, k).row_words;        // sorted rows travel as they are
  return packed_words(N_total, k);
}

int gficf_jaccard_pack_rows_device(gficf_ctx* ctx, const int32_t* d_table_rows, int64_t n_rows, int k, int64_t N_total,
                                   uint32_t* d_packed) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative n_rows");
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_table_rows || !d_packed) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (sorted_fmt(k)) {
    GFICF_HIP_CHECK(hipMemcpyAsync(d_packed, d_table_rows, sizeof(uint32_t) * (size_t)n_rows * (size_t)table_fmt(N_total, k).row_words,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return GFICF_OK;
  }
  int64_t blocks = gficf_ceil_div(n_rows, PACK_ROWS);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  const TableFmt f = table_fmt(N_total, k);
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), (size_t)PACK_ROWS * f.row_words * sizeof(uint32_t), ctx->stream,
                     (const uint32_t*)d_table_rows, n_rows, k, f.kpad, f.compact ? 1 : 0, id_bits(N_total), packed_words(N_total, k), d_packed,
                     f.row_words);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_unpack_rows_device(gficf_ctx* ctx, const uint32_t* d_packed, int64_t n_rows, int k, int64_t N_total,
                                     int32_t* d_table_rows) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative n_rows");
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_table_rows || !d_packed) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (sorted_fmt(k)) {
    GFICF_HIP_CHECK(hipMemcpyAsync(d_table_rows, d_packed, sizeof(uint32_t) * (size_t)n_rows * (size_t)table_fmt(N_total, k).row_words,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return GFICF_OK;
  }
  const int wpr = packed_words(N_total, k);
  int64_t blocks = gficf_ceil_div(n_rows, UNPACK_ROWS);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  const TableFmt f = table_fmt(N_total, k);
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)blocks), dim3(256), (size_t)UNPACK_ROWS * wpr * sizeof(uint32_t), ctx->stream,
                     d_packed, n_rows, k, f.kpad, f.compact ? 1 : 0, id_bits(N_total), wpr, (uint32_t*)d_table_rows, f.row_words);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

static int edges_filtered(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin, int64_t cell_end,
                          uint16_t* d_u_ws, int64_t* d_cell_ptr, double* d_from, double* d_to, double* d_weight, int set_mode,
                          const int32_t* d_order = nullptr) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (cell_begin < 0 || cell_end < cell_begin || cell_end > N)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "cell range [%lld, %lld) outside [0, %lld]", (long long)cell_begin, (long long)cell_end, (long long)N);
  if (!d_cell_ptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  const int64_t n_cells = cell_end - cell_begin;
  if (n_cells == 0 || k == 0) {
    GFICF_HIP_CHECK(hipMemsetAsync(d_cell_ptr, 0, sizeof(int64_t) * (size_t)(n_cells + 1), ctx->stream));
    return GFICF_OK;
  }
  if (!d_table || !d_u_ws || !d_from || !d_to || !d_weight) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  // 1. intersection counts only (no 24 B/edge matrix)
  EdgeOut o{nullptr, nullptr, nullptr, nullptr, d_u_ws, set_mode};
  const uint32_t* t = (const uint32_t*)d_table;
  rc = launch_edges_k(ctx, t, N, k, cell_begin, cell_end, o);
  if (rc) return rc;
  // 2. kept edges per cell -> offsets
  hipLaunchKernelGGL(k_edge_kept_count, dim3((unsigned)gficf_ceil_div(n_cells, 256)), dim3(256), 0, ctx->stream, d_u_ws,
                     n_cells, k, d_cell_ptr);
  GFICF_HIP_CHECK(hipGetLastError());
  rc = gficf_exclusive_scan_i64(ctx, d_cell_ptr, n_cells + 1);
  if (rc) return rc;
  // 3. ordered compacted write
  int64_t blocks = gficf_ceil_div(n_cells, 4);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  if (sorted_fmt(k)) {
    hipLaunchKernelGGL(k_edge_write_sorted, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, d_u_ws, k, cell_begin, n_cells, d_cell_ptr, d_from,
                       d_to, d_weight, sorted_kp(k), d_order);
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  }
#define LAUNCH_EW(KP, CM)                                                                                               \
  hipLaunchKernelGGL((k_edge_write<KP, CM>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, d_u_ws, k, cell_begin, \
                     n_cells, d_cell_ptr, d_from, d_to, d_weight, pitch, d_order)
  const bool cm = table_fmt(N, k).compact;
  const int pitch = table_fmt(N, k).row_words;
  switch (kpad_for(k)) {
    case 16: LAUNCH_EW(16, false); break;
    case 32: if (cm) LAUNCH_EW(32, true); else LAUNCH_EW(32, false); break;
    case 64: if (cm) LAUNCH_EW(64, true); else LAUNCH_EW(64, false); break;
    case 128: if (cm) LAUNCH_EW(128, true); else LAUNCH_EW(128, false); break;
    default: if (cm) LAUNCH_EW(256, true); else LAUNCH_EW(256, false); break;
  }
#undef LAUNCH_EW
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_edges_filtered_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin,
                                        int64_t cell_end, uint16_t* d_u_ws, int64_t* d_cell_ptr, double* d_from,
                                        double* d_to, double* d_weight) {
  return edges_filtered(ctx, d_table, N, k, cell_begin, cell_end, d_u_ws, d_cell_ptr, d_from, d_to, d_weight, 0);
}

}  // extern "C"

// (internal since ABI 7: common.h)
int gficf_jaccard_edges_filtered_mapped(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin, int64_t cell_end,
                                        uint16_t* d_u_ws, int64_t* d_cell_ptr, double* d_from, double* d_to, double* d_weight,
                                        const int32_t* d_order) {
  if (!d_order) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer (d_order)");
  return edges_filtered(ctx, d_table, N, k, cell_begin, cell_end, d_u_ws, d_cell_ptr, d_from, d_to, d_weight, 0, d_order);
}

extern "C" {

// host form of the filtered build: plan runs everything and returns the edge count, finish copies out.
// Device buffers are pieces of the context's pool slot 5 (kept between calls; nothing is allocated per call).
struct gficf_edge_plan {
  int64_t n_edges = 0;
  double* d_from = nullptr;
  double* d_to = nullptr;
  double* d_weight = nullptr;
  // compact form (round 5, from host_compact_min_edges() edges on): the uint16 counts came back instead of the kept edges; finish writes
  // from / to / weight on the host from (cell, idx[cell, slot], u) — the caller's idx must stay valid until finish (it is the argument
  // of the same `.Call` / Python call)
  bool compact = false;
  std::vector<uint16_t> u;
  std::vector<int64_t> first;        // first[t]: output position of thread t's first kept edge; first[nthreads] = n_edges
  const void* idx = nullptr;
  int idx_is_f64 = 0, k = 0;
  int64_t N = 0, ld = 0;
};

// cells [c0, c1) of thread t of nt
static inline void cell_share(int64_t N, int64_t nt, int64_t t, int64_t& c0, int64_t& c1) {
  const int64_t per = (N + nt - 1) / nt;
  c0 = t * per < N ? t * per : N;
  c1 = c0 + per < N ? c0 + per : N;
}

static void edge_plan_free(gficf_ctx* ctx) {
  delete ctx->edge_plan;
  ctx->edge_plan = nullptr;
}

}  // extern "C"

void gficf_edge_plan_free(gficf_ctx* ctx) { edge_plan_free(ctx); }      // (common.h: for gficf_ctx_trim / destroy)

extern "C" {

static int gficf_jaccard_filtered_host_plan_body(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                                     int64_t* n_edges) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (!n_edges) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_edges is NULL");
  edge_plan_free(ctx);
  *n_edges = 0;
  const int64_t E = N * (int64_t)k;
  gficf_edge_plan* p = new gficf_edge_plan();
  ctx->edge_plan = p;
  if (E == 0) return GFICF_OK;
  if (!idx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  if (E >= host_compact_min_edges()) {
    // the compact form: 2 B per edge over PCIe, the kept edges counted (here) and written (in finish) by the host cores
    p->compact = true;
    p->idx = idx; p->idx_is_f64 = idx_is_f64; p->N = N; p->k = k; p->ld = ld;
    try {
      p->u.resize((size_t)E);
    } catch (...) {
      edge_plan_free(ctx);
      GFICF_FAIL(GFICF_ERR_HIP, "out of host memory for %lld intersection counts", (long long)E);
    }
    rc = gficf_jaccard_counts_host_body(ctx, idx, idx_is_f64, N, k, ld, p->u.data());
    if (rc) { edge_plan_free(ctx); return rc; }
    const unsigned hw = std::thread::hardware_concurrency();
    int64_t nt = E / 250000;
    if (nt > 32) nt = 32;
    if (hw && nt > (int64_t)hw) nt = hw;
    if (nt < 1) nt = 1;
    p->first.assign((size_t)nt + 1, 0);
    const uint16_t* const u = p->u.data();
    int64_t* const first = p->first.data();
    gficf_run_shares(nt, [=](int64_t t) {
      int64_t c0, c1, n = 0;
      cell_share(N, nt, t, c0, c1);
      for (int64_t r = c0 * k; r < c1 * k; ++r) n += u[r] != 0;
      first[t + 1] = n;
    });
    for (int64_t t = 0; t < nt; ++t) first[t + 1] += first[t];
    p->n_edges = first[nt];
    *n_edges = p->n_edges;
    return GFICF_OK;
  }
  const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
  const int roww = table_fmt(N, k).row_words;
  gficf_arena ar;
  const size_t o_idx = ar.take(esz * (size_t)ld * (size_t)k), o_tab = ar.take(sizeof(int32_t) * (size_t)N * (size_t)roww);
  const size_t o_u = ar.take(sizeof(uint16_t) * (size_t)E), o_ptr = ar.take(sizeof(int64_t) * (size_t)(N + 1));
  const size_t o_f = ar.take(sizeof(double) * (size_t)E), o_t = ar.take(sizeof(double) * (size_t)E), o_w = ar.take(sizeof(double) * (size_t)E);
  hipError_t e = ar.bind(ctx, 5);
  void* d_idx = ar.at<void>(o_idx);
  int32_t* d_table = ar.at<int32_t>(o_tab);
  uint16_t* d_u = ar.at<uint16_t>(o_u);
  int64_t* d_ptr = ar.at<int64_t>(o_ptr);
  p->d_from = ar.at<double>(o_f); p->d_to = ar.at<double>(o_t); p->d_weight = ar.at<double>(o_w);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  rc = GFICF_OK;
  int64_t total = 0;
  if (e == hipSuccess) {
    rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, d_table);
    if (!rc) rc = gficf_jaccard_edges_filtered_device(ctx, d_table, N, k, 0, N, d_u, d_ptr, p->d_from, p->d_to, p->d_weight);
    if (!rc) e = hipMemcpyAsync(&total, d_ptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) { gficf_set_error("HIP failure in gficf_jaccard_filtered_host_plan: %s", hipGetErrorString(e)); rc = GFICF_ERR_HIP; }
  if (rc) { edge_plan_free(ctx); return rc; }
  p->n_edges = total;
  *n_edges = total;
  return GFICF_OK;
}

/* The package's second Jaccard entry, the serial jaccard_coeff(idx, printOutput) (reference src/jaccard_coeff.cpp:19-44,
 * .Call symbol _gficf_jaccard_coeff, src/RcppExports.cpp:36): same edges, but (a) the intersection is Rcpp::intersect,
 * i.e. of the two rows as SETS (it differs from the parallel entry only for rows with duplicate ids), and (b) the rows
 * with u > 0 are written one after the other from the top of the (N*k) x 3 matrix (`r++`, :36-41), the rest stays 0. */
static int gficf_jaccard_coeff_host_body(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, double* weights,
                             int print_output) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (print_output && !ctx->quiet_rerun) gficf_print(ctx, "Running Jaccard Coefficient Estimation...\n");  // reference :25
  const int64_t E = N * (int64_t)k;
  if (E == 0) return GFICF_OK;
  if (!idx || !weights) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
  const int roww = table_fmt(N, k).row_words;
  void *d_idx = nullptr, *d_table = nullptr, *d_out = nullptr, *d_aux = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, esz * (size_t)ld * (size_t)k, &d_idx);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, sizeof(int32_t) * (size_t)N * (size_t)roww, &d_table);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(double) * 3 * (size_t)E, &d_out);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 3, sizeof(uint16_t) * (size_t)E + 64 + sizeof(int64_t) * (size_t)(N + 1), &d_aux);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  int64_t total = 0;
  if (e == hipSuccess) {
    double* d_from = (double*)d_out;
    int64_t* d_ptr = (int64_t*)d_aux;
    uint16_t* d_u = (uint16_t*)(d_ptr + N + 1);
    rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, (int32_t*)d_table);
    if (!rc) rc = edges_filtered(ctx, (const int32_t*)d_table, N, k, 0, N, d_u, d_ptr, d_from, d_from + E, d_from + 2 * E, 1);
    if (!rc) e = hipMemcpyAsync(&total, d_ptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
    if (!rc && e == hipSuccess) {
      gficf_advise_hugepages(weights, sizeof(double) * 3 * (size_t)E);                         // (a fresh R matrix, first touched by the memset)
      std::memset(weights, 0, sizeof(double) * 3 * (size_t)E);                                  // NumericMatrix weights(nrow*ncol, 3), :21
      for (int c = 0; c < 3 && e == hipSuccess && total > 0; ++c)
        e = hipMemcpyAsync(weights + (size_t)c * (size_t)E, d_from + (size_t)c * (size_t)E, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);
    }
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_coeff_host: %s", hipGetErrorString(e));
  return rc;
}

int gficf_jaccard_filtered_host_finish(gficf_ctx* ctx, double* from, double* to, double* weight) {
  GFICF_CTX_ENTER(ctx);
  gficf_edge_plan* p = ctx->edge_plan;
  if (!p) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_jaccard_filtered_host_finish without a plan");
  hipError_t e = hipSuccess;
  if (p->compact && p->n_edges > 0) {
    if (!from || !to || !weight) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
    const size_t b = sizeof(double) * (size_t)p->n_edges;
    gficf_advise_hugepages(from, b);
    gficf_advise_hugepages(to, b);
    gficf_advise_hugepages(weight, b);
    const int64_t N = p->N, ld = p->ld, nt = (int64_t)p->first.size() - 1;
    const int k = p->k, f64 = p->idx_is_f64;
    const uint16_t* const u = p->u.data();
    const int32_t* const ii = (const int32_t*)p->idx;
    const double* const id = (const double*)p->idx;
    const int64_t* const first = p->first.data();
    const double twok = 2.0 * (double)k;
    gficf_run_shares(nt, [=](int64_t t) {
        int64_t c0, c1;
        cell_share(N, nt, t, c0, c1);
        int64_t d = first[t];
        for (int64_t i = c0; i < c1; ++i)
          for (int j = 0; j < k; ++j) {
            const int v = u[i * k + j];
            if (v > 0) {                                            // reference R/clustCells.R:66 on the rows of :48-52, in order
              from[d] = (double)(i + 1);
              to[d] = f64 ? id[(int64_t)j * ld + i] : (double)ii[(int64_t)j * ld + i];
              weight[d] = (double)v / (twok - (double)v);
              ++d;
            }
          }
      });
    edge_plan_free(ctx);
    return GFICF_OK;
  }
  if (p->n_edges > 0) {
    if (!from || !to || !weight) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
    const size_t b = sizeof(double) * (size_t)p->n_edges;
    // (freshly allocated by the caller as a rule: huge pages asked for and the pages mapped from several threads, not one fault at a
    // time under the copies)
    gficf_prefault(from, b);
    gficf_prefault(to, b);
    gficf_prefault(weight, b);
    e = hipMemcpyAsync(from, p->d_from, b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(to, p->d_to, b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(weight, p->d_weight, b, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
  }
  edge_plan_free(ctx);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_filtered_host_finish: %s", hipGetErrorString(e));
  return GFICF_OK;
}

int gficf_jaccard_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                       double* rmat, int print_output) {
  return distinct_first(ctx, [&]() { return gficf_jaccard_host_body(ctx, idx, idx_is_f64, N, k, ld, rmat, print_output); });
}

int gficf_jaccard_counts_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, uint16_t* u) {
  return distinct_first(ctx, [&]() { return gficf_jaccard_counts_host_body(ctx, idx, idx_is_f64, N, k, ld, u); });
}

int gficf_jaccard_filtered_host_plan(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                                     int64_t* n_edges) {
  return distinct_first(ctx, [&]() { return gficf_jaccard_filtered_host_plan_body(ctx, idx, idx_is_f64, N, k, ld, n_edges); });
}

int gficf_jaccard_coeff_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, double* weights,
                             int print_output) {
  return distinct_first(ctx, [&]() { return gficf_jaccard_coeff_host_body(ctx, idx, idx_is_f64, N, k, ld, weights, print_output); });
}

}  // extern "C"
