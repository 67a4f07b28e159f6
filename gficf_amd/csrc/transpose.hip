// transpose.hip — `t(data$gficf)`: the GF-ICF matrix (genes x cells, CSC) turned into cells x genes CSC, i.e. the CSR
// form of the same matrix.  "Next" row N3 (second half) of the scope table: the reference does this with Matrix::t at
// R/dimensinalityReduction.R:33 and :100 (`data$pca$cells = t(data$gficf)`), the input of its PCA.
//
// The result is a dgCMatrix again, so within every gene the cell indices must come out ascending: a STABLE counting
// sort of the stored entries by gene.  Three launches and the context's scan:
//   k_tr_count    one workgroup per block of consecutive cells: histogram of the block's entries over the genes in LDS
//                 (36 864 u32 counters = 144 KB of the CU's 160 KB), written out as row b of a blocks x G matrix;
//   k_tr_prefix   per gene the exclusive prefix over the blocks (in place) and the gene's total;
//   (scan)        totals -> out_ptr (G + 1 entries);
//   k_tr_scatter  the block's counters start at its prefix; its cells are taken IN ORDER, all threads sharing one cell
//                 (a valid column names a gene once, so a cell's entries never collide on a counter), a barrier
//                 between cells: position = out_ptr[gene] + counter++.  The next cell's loads are in flight while
//                 the current one is placed.
// HBM-bound integer/byte work: 4 B/entry for the count, 12 B read + 12 B written for the scatter (28 B/entry
// algorithmic); the scattered 4 + 8 B stores are what bounds it (partial lines until a gene's run in the block fills).
// Matrices with more than TR_GENES genes take ceil(G / TR_GENES) sweeps (gene ranges in grid.y).
#include <vector>

#include <atomic>

#include "common.h"

namespace {

constexpr int TR_THREADS = 1024;
constexpr int TR_GENES = 36864;       // LDS counters per workgroup
constexpr int TR_PAIR_GENES = 9000;   // up to here the pairing form fits (16 B of LDS per gene)
constexpr int TR_MAX_CPB = 1024;      // cells per block: their colptr slice sits in LDS next to the counters
constexpr int TR_PRE = 2;             // entries per thread fetched ahead for the next cell

struct TrShape {
  int64_t G, n_cells, nnz;
  int cpb, n_blocks, n_ranges, gcap, pair;
};

static TrShape tr_shape(int64_t G, int64_t N, int64_t nnz, bool may_pair = true) {
  TrShape s;
  s.G = G; s.n_cells = N; s.nnz = nnz;
  // Few genes (the filtered GF-ICF matrix): the pairing form, one block per CU — the fewer blocks write at the same time,
  // the more of a gene's 128-byte lines complete in cache (measured 54 k cells x 4.6 k genes: 768 blocks 0.90 ms, 256
  // blocks 0.76 ms).  Many genes: plain stores, about three blocks per CU (23 k genes: 2.0 ms against 2.4 ms).
  s.pair = may_pair && G <= TR_PAIR_GENES && nnz < ((int64_t)1 << 32);
  int64_t cpb = gficf_ceil_div(N > 0 ? N : 1, s.pair ? 256 : 768);           // MI355X: 256 CUs
  if (cpb < 16) cpb = 16;
  if (cpb > TR_MAX_CPB) cpb = TR_MAX_CPB;
  s.cpb = (int)cpb;
  s.n_blocks = (int)gficf_ceil_div(N > 0 ? N : 1, cpb);
  s.gcap = (int)(G < TR_GENES ? (G > 0 ? G : 1) : TR_GENES);
  s.n_ranges = (int)gficf_ceil_div(G > 0 ? G : 1, TR_GENES);
  return s;
}

// counters (+ the waiting entries of the pairing form) | begin of every cell of the block (8 B) | its length (4 B)
static size_t tr_lds_bytes(const TrShape& s) {
  return (size_t)((s.gcap + 1) & ~1) * (s.pair ? 16 : 4) + (size_t)(s.cpb + 1) * 12;
}

// a column's entry range, clamped into the arrays (a bad colptr is reported by k_tr_count and must not fault)
__device__ static inline int64_t tr_clamp(int64_t e, int64_t nnz) { return e < 0 ? 0 : e > nnz ? nnz : e; }

// Columns are given as begin / end pointers (the canonical CSC hands in colptr and colptr + 1, the pointerB / pointerE form of
// gficf_csc_scale_be_device its two arrays): a wave takes a cell at a time, its lanes the cell's entries.
__global__ __launch_bounds__(TR_THREADS) void k_tr_count(TrShape s, const int64_t* __restrict__ ptr_b, const int64_t* __restrict__ ptr_e,
                                                         const int32_t* __restrict__ rowidx, uint32_t* __restrict__ cnt,
                                                         uint32_t* __restrict__ status) {
  extern __shared__ uint32_t s_cnt[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g0 = (int64_t)blockIdx.y * s.gcap;
  const int gn = (int)(s.G - g0 < s.gcap ? s.G - g0 : s.gcap);
  for (int t = tid; t < gn; t += TR_THREADS) s_cnt[t] = 0u;
  __syncthreads();
  const int64_t c0 = (int64_t)b * s.cpb, c1 = c0 + s.cpb < s.n_cells ? c0 + s.cpb : s.n_cells;
  bool bad = false;
  for (int64_t c = c0 + wave; c < c1; c += TR_THREADS / 64) {
    const int64_t lo = ptr_b[c], hi = ptr_e[c];
    bad |= lo < 0 || hi < lo || hi > s.nnz;
    const int64_t e0 = tr_clamp(lo, s.nnz), e1 = tr_clamp(hi, s.nnz);
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int64_t g = rowidx[e];
      if (g < 0 || g >= s.G) { bad = true; continue; }
      const int64_t t = g - g0;
      if (t >= 0 && t < gn) atomicAdd(&s_cnt[t], 1u);
    }
  }
  if (bad) atomicOr(status, GFICF_ST_BAD_CSC);
  __syncthreads();
  uint32_t* row = cnt + (size_t)b * (size_t)s.G + g0;
  for (int t = tid; t < gn; t += TR_THREADS) row[t] = s_cnt[t];
}

// Per gene: exclusive prefix of its counts over the blocks (in place) and its total.  A workgroup takes 16 genes x 16
// slices of the blocks: slice sums, a 16-step scan through LDS, then the prefixes (the matrix is read twice, from L2).
__global__ __launch_bounds__(256) void k_tr_prefix(TrShape s, uint32_t* __restrict__ cnt, int64_t* __restrict__ total) {
  __shared__ uint32_t s_sum[16][17];
  const int gl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t g = (int64_t)blockIdx.x * 16 + gl;
  const int per = (s.n_blocks + 15) / 16;
  const int b0 = sl * per, b1 = b0 + per < s.n_blocks ? b0 + per : s.n_blocks;
  uint32_t sum = 0;
  if (g < s.G)
    for (int b = b0; b < b1; ++b) sum += cnt[(size_t)b * (size_t)s.G + g];
  s_sum[sl][gl] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int t = 0; t < sl; ++t) run += s_sum[t][gl];
  if (g < s.G) {
    for (int b = b0; b < b1; ++b) {
      uint32_t* p = cnt + (size_t)b * (size_t)s.G + g;
      const uint32_t v = *p;
      *p = run;
      run += v;
    }
    if (sl == 15) total[g] = (int64_t)run;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) total[s.G] = 0;
}

// MODE 0: the LDS counters hold offsets within the gene and every entry looks its gene's start up in out_ptr (nnz >= 2^32).
// MODE 1: the counters hold absolute output positions (nnz < 2^32, always so for a dgCMatrix).
// MODE 2: as 1, and entries leave in aligned pairs: one that lands on an even position waits in LDS (12 B per gene) for
//         its odd neighbour — the same gene in a later cell of the block — and both go out as one 8-byte and one 16-byte
//         store.  The scattered stores are what this kernel costs, and this halves them.
template <int MODE>
__global__ __launch_bounds__(TR_THREADS) void k_tr_scatter(TrShape s, const int64_t* __restrict__ ptr_b, const int64_t* __restrict__ ptr_e,
                                                           const int32_t* __restrict__ rowidx, const double* __restrict__ x,
                                                           const uint32_t* __restrict__ cnt, const int64_t* __restrict__ out_ptr,
                                                           int32_t* __restrict__ out_idx, double* __restrict__ out_x) {
  extern __shared__ uint32_t s_cnt[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t g0 = (int64_t)blockIdx.y * s.gcap;
  const int gn = (int)(s.G - g0 < s.gcap ? s.G - g0 : s.gcap);
  const int gpad = (s.gcap + 1) & ~1;
  int64_t* s_cp = (int64_t*)(s_cnt + gpad);                             // 8-byte aligned behind the counters: begin of every cell
  double* s_sx = (double*)(s_cp + s.cpb + 1);                           // MODE 2: the waiting entry of every gene
  int32_t* s_sc = (int32_t*)(s_sx + (MODE == 2 ? gpad : 0));            //         its cell, -1 = none
  uint32_t* s_len = (uint32_t*)(s_sc + (MODE == 2 ? gpad : 0));         // entries of every cell (a column holds fewer than 2^32)
  const int64_t c0 = (int64_t)b * s.cpb, c1 = c0 + s.cpb < s.n_cells ? c0 + s.cpb : s.n_cells;
  const int nc = (int)(c1 - c0);
  const uint32_t* row = cnt + (size_t)b * (size_t)s.G + g0;
  for (int t = tid; t < gn; t += TR_THREADS) {
    s_cnt[t] = row[t] + (MODE ? (uint32_t)out_ptr[g0 + t] : 0u);
    if (MODE == 2) s_sc[t] = -1;
  }
  for (int t = tid; t < nc; t += TR_THREADS) {
    const int64_t lo = tr_clamp(ptr_b[c0 + t], s.nnz), hi = tr_clamp(ptr_e[c0 + t], s.nnz);
    s_cp[t] = lo;
    s_len[t] = hi > lo ? (uint32_t)(hi - lo) : 0u;
  }
  __syncthreads();

  auto place = [&](int t, int64_t base, int32_t cell, double xv) {
    const int64_t pos = base + (int64_t)atomicAdd(&s_cnt[t], 1u);
    if (pos >= s.nnz) return;                                           // never for a valid matrix
    if (MODE == 2) {
      if ((pos & 1) == 0) {
        s_sc[t] = cell;
        s_sx[t] = xv;
        return;
      }
      const int32_t pc = s_sc[t];
      if (pc >= 0) {
        s_sc[t] = -1;
        *(int2*)(out_idx + pos - 1) = make_int2(pc, cell);
        *(double2*)(out_x + pos - 1) = make_double2(s_sx[t], xv);
        return;
      }
    }
    out_idx[pos] = cell;
    out_x[pos] = xv;
  };

  int pg[TR_PRE];            // gene - g0, or -1: nothing to place
  double px[TR_PRE];
  int64_t pb[TR_PRE];
  auto fetch = [&](int ci) {
    const int64_t lo = s_cp[ci], hi = lo + (int64_t)s_len[ci];
#pragma unroll
    for (int u = 0; u < TR_PRE; ++u) {
      const int64_t e = lo + tid + u * TR_THREADS;
      pg[u] = -1;
      if (e < hi) {
        const int64_t g = rowidx[e];
        const int64_t t = g - g0;
        if (g >= 0 && g < s.G && t >= 0 && t < gn) {
          pg[u] = (int)t;
          px[u] = x[e];
          pb[u] = MODE ? 0 : out_ptr[g];
        }
      }
    }
  };
  if (nc > 0) fetch(0);
  for (int ci = 0; ci < nc; ++ci) {
    int cg[TR_PRE];
    double cx[TR_PRE];
    int64_t cb[TR_PRE];
#pragma unroll
    for (int u = 0; u < TR_PRE; ++u) { cg[u] = pg[u]; cx[u] = px[u]; cb[u] = pb[u]; }
    if (ci + 1 < nc) fetch(ci + 1);
    const int32_t cell = (int32_t)(c0 + ci);
#pragma unroll
    for (int u = 0; u < TR_PRE; ++u)
      if (cg[u] >= 0) place(cg[u], cb[u], cell, cx[u]);
    // the part of a long column beyond the entries fetched ahead
    const int64_t hi = s_cp[ci] + (int64_t)s_len[ci];
    for (int64_t e = s_cp[ci] + tid + (int64_t)TR_PRE * TR_THREADS; e < hi; e += TR_THREADS) {
      const int64_t g = rowidx[e];
      const int64_t t = g - g0;
      if (g >= 0 && g < s.G && t >= 0 && t < gn) place((int)t, MODE ? 0 : out_ptr[g], cell, x[e]);
    }
    __syncthreads();
  }
  if (MODE == 2) {
    // entries still waiting: their odd neighbour belongs to the next block (or there is none)
    for (int t = tid; t < gn; t += TR_THREADS) {
      const int32_t pc = s_sc[t];
      if (pc >= 0) {
        const int64_t pos = (int64_t)s_cnt[t] - 1;
        out_idx[pos] = pc;
        out_x[pos] = s_sx[t];
      }
    }
  }
}

}  // namespace

extern "C" {

size_t gficf_csc_transpose_workspace_bytes(int64_t G, int64_t n_cells) {
  if (G < 0 || n_cells < 0) return 0;
  const TrShape s = tr_shape(G, n_cells, 0, false);          // the plain form has the most blocks
  return (size_t)s.n_blocks * (size_t)(G > 0 ? G : 1) * sizeof(uint32_t) + 256;
}

static int transpose_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int64_t* d_col_end, const int32_t* d_rowidx,
                            const double* d_x, int64_t nnz, int64_t* d_out_ptr, int32_t* d_out_idx, double* d_out_x, void* d_ws, size_t ws_bytes);

int gficf_csc_transpose_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int32_t* d_rowidx,
                               const double* d_x, int64_t nnz, int64_t* d_out_ptr, int32_t* d_out_idx, double* d_out_x, void* d_ws,
                               size_t ws_bytes) {
  return transpose_launch(ctx, G, n_cells, d_colptr, d_colptr ? d_colptr + 1 : nullptr, d_rowidx, d_x, nnz, d_out_ptr, d_out_idx, d_out_x, d_ws, ws_bytes);
}

/* t() of a matrix in the pointerB / pointerE form (gficf_csc_scale_be_device): cell c's entries are [d_col_begin[c], d_col_end[c])
 * of arrays with `capacity` entries; the result is an ordinary compact cells x genes CSC (d_out_ptr[G] = number of entries;
 * d_out_idx / d_out_x need room for them: `capacity` always suffices). */
int gficf_csc_transpose_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_col_begin, const int64_t* d_col_end,
                                  const int32_t* d_rowidx, const double* d_x, int64_t capacity, int64_t* d_out_ptr, int32_t* d_out_idx,
                                  double* d_out_x, void* d_ws, size_t ws_bytes) {
  if (G > 0 && n_cells > 0 && !d_col_end) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  return transpose_launch(ctx, G, n_cells, d_col_begin, d_col_end, d_rowidx, d_x, capacity, d_out_ptr, d_out_idx, d_out_x, d_ws, ws_bytes);
}

static int transpose_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int64_t* d_col_end, const int32_t* d_rowidx,
                            const double* d_x, int64_t nnz, int64_t* d_out_ptr, int32_t* d_out_idx, double* d_out_x, void* d_ws, size_t ws_bytes) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || n_cells < 0 || nnz < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (!d_out_ptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (G == 0 || n_cells == 0) {
    GFICF_HIP_CHECK(hipMemsetAsync(d_out_ptr, 0, sizeof(int64_t) * ((size_t)G + 1), ctx->stream));
    return GFICF_OK;
  }
  if (!d_colptr || !d_ws || (nnz > 0 && (!d_rowidx || !d_x || !d_out_idx || !d_out_x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ws_bytes < gficf_csc_transpose_workspace_bytes(G, n_cells))
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "workspace too small: %zu < %zu bytes", ws_bytes, gficf_csc_transpose_workspace_bytes(G, n_cells));
  if (n_cells > INT32_MAX) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "more than 2^31 - 1 cells");
  const TrShape s = tr_shape(G, n_cells, nnz);
  const size_t lds = tr_lds_bytes(s);
  static std::atomic<bool> attr_set[64];                 // per device: the attribute belongs to the device's copy of the kernel
  if (!attr_set[ctx->device & 63]) {
    const int mx = (int)((size_t)TR_GENES * 4 + (size_t)(TR_MAX_CPB + 1) * 12);  // 159 756 B of the CU's 160 KB
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_tr_count, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_tr_scatter<0>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_tr_scatter<1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_tr_scatter<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    attr_set[ctx->device & 63] = true;
  }
  uint32_t* cnt = (uint32_t*)d_ws;
  const dim3 grid((unsigned)s.n_blocks, (unsigned)s.n_ranges);
  hipLaunchKernelGGL(k_tr_count, grid, dim3(TR_THREADS), (size_t)((s.gcap + 1) & ~1) * 4, ctx->stream, s, d_colptr, d_col_end, d_rowidx, cnt, ctx->d_status);
  hipLaunchKernelGGL(k_tr_prefix, dim3((unsigned)gficf_ceil_div(G, 16)), dim3(256), 0, ctx->stream, s, cnt, d_out_ptr);
  GFICF_HIP_CHECK(hipGetLastError());
  const int rc = gficf_exclusive_scan_i64(ctx, d_out_ptr, G + 1);
  if (rc) return rc;
  auto* kern = nnz >= ((int64_t)1 << 32) ? k_tr_scatter<0> : s.pair ? k_tr_scatter<2> : k_tr_scatter<1>;
  hipLaunchKernelGGL(kern, grid, dim3(TR_THREADS), lds, ctx->stream, s, d_colptr, d_col_end, d_rowidx, d_x, cnt, d_out_ptr, d_out_idx, d_out_x);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_csc_transpose_host(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr, int colptr_is_i64, const int32_t* rowidx,
                             const double* x, int64_t* out_ptr, int32_t* out_idx, double* out_x) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (!colptr || !out_ptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  std::vector<int64_t> h_cp((size_t)N + 1);
  for (int64_t c = 0; c <= N; ++c) h_cp[(size_t)c] = colptr_is_i64 ? ((const int64_t*)colptr)[c] : (int64_t)((const int32_t*)colptr)[c];
  bool mono = h_cp[0] == 0;
  for (int64_t c = 0; c < N && mono; ++c) mono = h_cp[(size_t)c + 1] >= h_cp[(size_t)c];
  const int64_t nnz = h_cp[(size_t)N];
  if (!mono) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr does not start at 0 or is not monotone");
  if (nnz > 0 && (!rowidx || !x || !out_idx || !out_x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  const size_t nsz = (size_t)(nnz > 0 ? nnz : 1), wsb = gficf_csc_transpose_workspace_bytes(G, N);
  gficf_arena ar;                                   // pool slot 0: no allocation per call
  const size_t o_cp = ar.take(sizeof(int64_t) * ((size_t)N + 1)), o_op = ar.take(sizeof(int64_t) * ((size_t)G + 1));
  const size_t o_ri = ar.take(sizeof(int32_t) * nsz), o_oi = ar.take(sizeof(int32_t) * nsz);
  const size_t o_x = ar.take(sizeof(double) * nsz), o_ox = ar.take(sizeof(double) * nsz), o_ws = ar.take(wsb);
  hipError_t e = ar.bind(ctx, 0);
  int64_t* const d_cp = ar.at<int64_t>(o_cp); int64_t* const d_op = ar.at<int64_t>(o_op);
  int32_t* const d_ri = ar.at<int32_t>(o_ri); int32_t* const d_oi = ar.at<int32_t>(o_oi);
  double* const d_x = ar.at<double>(o_x); double* const d_ox = ar.at<double>(o_ox);
  void* const d_ws = ar.at<void>(o_ws);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cp, h_cp.data(), sizeof(int64_t) * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_ri, rowidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  int rc = GFICF_OK;
  if (e == hipSuccess) {
    rc = gficf_csc_transpose_device(ctx, G, N, d_cp, d_ri, d_x, nnz, d_op, d_oi, d_ox, d_ws, wsb);
    if (!rc && nnz > 0) {                          // map the caller's fresh result pages while the kernels run
      gficf_prefault(out_x, sizeof(double) * (size_t)nnz);
      gficf_prefault(out_idx, sizeof(int32_t) * (size_t)nnz);
    }
    // validate (status word) before the results are handed back
    if (!rc) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
    if (!rc) e = hipMemcpyAsync(out_ptr, d_op, sizeof(int64_t) * ((size_t)G + 1), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess && nnz > 0) e = hipMemcpyAsync(out_idx, d_oi, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess && nnz > 0) e = hipMemcpyAsync(out_x, d_ox, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_csc_transpose_host: %s", hipGetErrorString(e));
  return rc;
}

}  // extern "C"
