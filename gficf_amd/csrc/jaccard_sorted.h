// jaccard_sorted.h — the exact Jaccard path for k > GFICF_JACCARD_MAX_K neighbours per cell (round 5).
// Included by jaccard.hip INSIDE its anonymous namespace, behind EdgeOut / decode_id / wave_lds_fence.
//
// The reference loops over mat.ncol() with no limit (src/rcpp_parallel_jaccard_coeff.cpp:26-46: two copies, two std::sort,
// std::set_intersection over the sorted rows — multiset semantics).  The hash-set / bit-set kernels of jaccard.hip stop at 256
// slots; beyond that the table takes a third format and the edge build follows the reference's own plan, sort + merge, with the
// sort done ONCE per row at ingest instead of once per edge:
//   "sorted" rows (k > 256): KP = k rounded up to 64; a row is 2 KP words:
//       words [0, KP)    the ids in slot order (0 = no id; what the edge order and the neighbour column are read from),
//       words [KP, 2 KP) the same ids ascending (pads 0xFFFFFFFF behind the k-th).
//   ingest : a tiled transpose of the column-major input (coalesced both ways) into the slot-order halves, then one workgroup per
//            row sorts it — a bitonic network with every comparator pointing up, so that the positions past k act as +infinity
//            without being stored; in LDS for k <= 16384, in place in the row beyond;
//   edges  : one workgroup per cell, one wave per edge: element p of the cell's sorted row A (value v, the r-th of its run: r = p -
//            lower_bound(A, v)) is in the multiset intersection with the neighbour's sorted row B iff B holds more than r copies
//            of v, i.e. B[lower_bound(B, v) + r] == v — one binary search per element, no second sort, no merge.  A, the ranks r
//            and the wave's B are staged in LDS while six rows fit 64 KB (k <= 2688); beyond, the searches read the table.
//   The serial entry's set semantics (Rcpp::intersect, src/jaccard_coeff.cpp:33): only the first element of a run counts.
// Slow next to the fast kernels (k log k per edge instead of k probes) but exact for every input and every k up to 65535 (the
// uint16 counts of the filtered / compact returns); a data set with k > 256 neighbours per cell is not a Phenograph workload.
#pragma once

#include "jaccard_shared.h"

namespace {

constexpr int SORTED_MAX_K = 65535;
constexpr uint32_t SORTED_PAD = 0xFFFFFFFFu;
constexpr int SORTED_SORT_LDS_K = 16384;      // rows sorted in LDS up to this k (64 KB)
constexpr int SORTED_EDGE_LDS_KP = 2688;      // edge kernel: A, ranks and four B rows in LDS up to this KP (6 x 4 x 2688 = 63 KB)

// (sorted_kp, sorted_from_k and sorted_fmt are defined in jaccard.hip, in front of table_fmt, which needs them)

// Slot-order halves: 64 rows x 64 slots per workgroup step; blockIdx.y = the tile of slots.
template <typename T>
__global__ __launch_bounds__(256) void k_ingest_slots(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld, int64_t N_total,
                                                      uint32_t* __restrict__ table, uint32_t* __restrict__ status, int zero_ok, int kp) {
  __shared__ uint32_t tile[64][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = (int)blockIdx.y * 64;
  const int64_t pitch = 2 * (int64_t)kp;
  for (int64_t row0 = (int64_t)blockIdx.x * 64; row0 < n_rows; row0 += (int64_t)gridDim.x * 64) {
    const int64_t r = row0 + lane;
    bool bad = false;
    for (int m = wave; m < 64; m += 4) {
      const int j = j0 + m;
      uint32_t v = 0;
      if (j < k && r < n_rows) {
        bool ok;
        v = decode_id<T>(idx[(int64_t)j * ld + r], N_total, ok, zero_ok);
        bad |= !ok;
      }
      tile[lane][m] = v;
    }
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    __syncthreads();
    const int rows_here = (int)((n_rows - row0) < 64 ? (n_rows - row0) : 64);
    for (int e = tid; e < rows_here * 64; e += 256) {
      const int rr = e >> 6, m = e & 63;
      table[(row0 + rr) * pitch + j0 + m] = tile[rr][m];
    }
    __syncthreads();
  }
}

__device__ inline void sorted_cswap(uint32_t* buf, int a, int b) {
  const uint32_t x = buf[a], y = buf[b];
  if (x > y) { buf[a] = y; buf[b] = x; }
}

// Ascending halves: one workgroup per row.  Bitonic network in its all-ascending form (the first step of every merge compares
// mirrored positions, the rest at distance d): positions >= k stand for +infinity and a comparator that touches one is skipped.
__global__ __launch_bounds__(256) void k_sort_rows(uint32_t* __restrict__ table, int64_t n_rows, int k, int kp, int use_lds) {
  extern __shared__ uint32_t s_sort[];
  const int tid = threadIdx.x;
  int np2 = 2;
  while (np2 < k) np2 <<= 1;
  const int pairs = np2 >> 1;
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const uint32_t* const slots = table + r * 2 * (int64_t)kp;
    uint32_t* const out = table + r * 2 * (int64_t)kp + kp;
    uint32_t* const buf = use_lds ? s_sort : out;
    for (int e = tid; e < k; e += 256) buf[e] = slots[e];
    __syncthreads();
    for (int lsize = 1; (1 << lsize) <= np2; ++lsize) {     // merged runs of 2^lsize
      const int size = 1 << lsize, half = size >> 1;
      for (int t = tid; t < pairs; t += 256) {
        const int blk = t >> (lsize - 1), off = t & (half - 1);
        const int a = blk * size + off, b = blk * size + size - 1 - off;
        if (b < k) sorted_cswap(buf, a, b);
      }
      __syncthreads();
      for (int ld = lsize - 2; ld >= 0; --ld) {               // distance 2^ld
        const int d = 1 << ld;
        for (int t = tid; t < pairs; t += 256) {
          const int a = ((t >> ld) << (ld + 1)) + (t & (d - 1)), b = a + d;
          if (b < k) sorted_cswap(buf, a, b);
        }
        __syncthreads();
      }
    }
    for (int e = tid; e < kp; e += 256) {
      if (e >= k) out[e] = SORTED_PAD;
      else if (use_lds) out[e] = buf[e];
    }
    __syncthreads();
  }
}

// first position in the ascending run [0, n) whose value is >= v
__device__ inline int sorted_lower_bound(const uint32_t* a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <int OUT, bool MAP>
__global__ __launch_bounds__(256) void k_jaccard_edges_sorted(const uint32_t* __restrict__ table, int64_t N, int k, int64_t cell_begin,
                                                              int64_t cell_end, EdgeOut o, int kp, int use_lds) {
  extern __shared__ uint32_t s_rows[];                       // use_lds: A[kp] | rank[kp] | B[4][kp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t pitch = 2 * (int64_t)kp;
  const double twok = 2.0 * (double)k;
  uint32_t* const sA = s_rows;
  uint32_t* const sR = s_rows + kp;
  uint32_t* const sB = s_rows + 2 * kp + wave * kp;
  for (int64_t i = cell_begin + blockIdx.x; i < cell_end; i += gridDim.x) {
    const uint32_t* const slots = table + i * pitch;
    const uint32_t* A = slots + kp;
    if (use_lds) {
      for (int e = tid; e < k; e += 256) sA[e] = A[e];
      __syncthreads();
      for (int e = tid; e < k; e += 256) sR[e] = (uint32_t)(e - sorted_lower_bound(sA, k, sA[e]));
      __syncthreads();
      A = sA;
    }
    const int64_t out_base = (i - cell_begin) * (int64_t)k;
    for (int j = wave; j < k; j += 4) {
      const uint32_t dst = slots[j];                          // (wave-uniform)
      int cnt = 0;
      if (dst != 0) {
        const uint32_t* B = table + (int64_t)(dst - 1) * pitch + kp;
        if (use_lds) {
          for (int e = lane; e < k; e += 64) sB[e] = B[e];
          wave_lds_fence();
          B = sB;
        }
        for (int p = lane; p < k; p += 64) {
          const uint32_t v = A[p];
          const int r = use_lds ? (int)sR[p] : p - sorted_lower_bound(A, k, v);
          const int pos = sorted_lower_bound(B, k, v) + r;    // the (r+1)-th copy of v in B, if B holds that many
          cnt += (v != 0u && pos < k && B[pos] == v && (o.set_mode == 0 || r == 0)) ? 1 : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (use_lds) wave_lds_fence();
      }
      if (lane == 0) {
        const int64_t r = out_base + j;
        const bool pos = cnt > 0;
        if (OUT != OUT_U16) {
          uint32_t dcol = dst;
          if (MAP && dst != 0) dcol = (uint32_t)o.l2g[dst - 1];
          o.src[r] = pos ? (double)((uint32_t)(i + 1) + o.src_off) : 0.0;                      // reference :49
          o.dst[r] = pos ? (double)dcol : 0.0;                                                  // reference :50
          o.w[r] = pos ? (double)cnt / (twok - (double)cnt) : 0.0;                              // reference :51
        }
        if (OUT == OUT_RMAT_U) o.u[r] = cnt;
        if (OUT == OUT_U16) o.u16[r] = (uint16_t)cnt;
      }
    }
    __syncthreads();                                          // (the next cell overwrites A and the ranks)
  }
}

// ordered compacted write of the edge filter for sorted rows (k_edge_write's job; the neighbour column from the slot-order half)
__global__ __launch_bounds__(256) void k_edge_write_sorted(const uint32_t* __restrict__ table, const uint16_t* __restrict__ u16, int k,
                                                           int64_t cell_begin, int64_t n_cells, const int64_t* __restrict__ ptr,
                                                           double* __restrict__ from, double* __restrict__ to, double* __restrict__ weight, int kp,
                                                           const int32_t* __restrict__ order) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const double twok = 2.0 * (double)k;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t c = w0; c < n_cells; c += nw) {
    int64_t pos = ptr[c];
    for (int s0 = 0; s0 < k; s0 += 64) {
      const int s = s0 + lane;
      int u = 0;
      uint32_t dst = 0;
      if (s < k) {
        u = u16[c * k + s];
        dst = table[(cell_begin + c) * 2 * (int64_t)kp + s];
      }
      const bool keep = u > 0;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t d = pos + __popcll(m & lt_mask);
        from[d] = order ? (double)(order[cell_begin + c] + 1) : (double)(uint32_t)(cell_begin + c + 1);
        to[d] = order ? (double)(order[dst - 1u] + 1) : (double)dst;
        weight[d] = (double)u / (twok - (double)u);           // reference :51
      }
      pos += __popcll(m);
    }
  }
}

template <typename T>
int launch_ingest_sorted(gficf_ctx* ctx, const T* d_idx, int64_t n_rows, int k, int64_t ld, int64_t N_total, uint32_t* table, int zero_ok) {
  const int kp = sorted_kp(k);
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  const int64_t tiles = gficf_ceil_div(n_rows, 64);
  int64_t gx = gficf_ceil_div(cap, kp / 64);
  if (gx > tiles) gx = tiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((k_ingest_slots<T>), dim3((unsigned)gx, (unsigned)(kp / 64)), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, table,
                     ctx->d_status, zero_ok, kp);
  GFICF_HIP_CHECK(hipGetLastError());
  const int use_lds = k <= SORTED_SORT_LDS_K ? 1 : 0;
  const int64_t gs = n_rows < cap ? n_rows : cap;
  hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)gs), dim3(256), use_lds ? sizeof(uint32_t) * (size_t)k : 0, ctx->stream, table, n_rows, k, kp, use_lds);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int OUT>
int launch_edges_sorted_o(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  const int kp = sorted_kp(k);
  const int use_lds = kp <= SORTED_EDGE_LDS_KP ? 1 : 0;
  const size_t lds = use_lds ? sizeof(uint32_t) * 6 * (size_t)kp : 0;
  const int64_t cap = (int64_t)ctx->num_cus * (use_lds ? (lds > 32768 ? 1 : lds > 16384 ? 2 : 4) : 8);
  const int64_t cells = ce - cb;
  const unsigned grid = (unsigned)(cells < cap ? cells : cap);
  if constexpr (OUT != OUT_U16) {
    if (o.l2g) {
      hipLaunchKernelGGL((k_jaccard_edges_sorted<OUT, true>), dim3(grid), dim3(256), lds, ctx->stream, table, N, k, cb, ce, o, kp, use_lds);
      GFICF_HIP_CHECK(hipGetLastError());
      return GFICF_OK;
    }
  }
  hipLaunchKernelGGL((k_jaccard_edges_sorted<OUT, false>), dim3(grid), dim3(256), lds, ctx->stream, table, N, k, cb, ce, o, kp, use_lds);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

inline int launch_edges_sorted(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  if (o.u16) return launch_edges_sorted_o<OUT_U16>(ctx, table, N, k, cb, ce, o);
  if (o.u) return launch_edges_sorted_o<OUT_RMAT_U>(ctx, table, N, k, cb, ce, o);
  return launch_edges_sorted_o<OUT_RMAT>(ctx, table, N, k, cb, ce, o);
}

}  // namespace
