// gficf_csc.hip — GF-ICF normalisation of a CSC genes x cells matrix for gfx950 (MI355X).
//
// Replaces the R-level chain of gficf() (reference R/gficf.R:17-33, normalize = FALSE):
//   R/gficf.R:40-41   gene filter by cell frequency        -> k_gene_count + k_gene_table
//   R/gficf.R:59      GF:  x / S_c (per-cell L1)            \
//   R/gficf.R:88-89   ICF weights log((N+1)/(nt_g+1))        | k_gene_table (weights)
//   R/gficf.R:79      x * w_g                                | k_scale_cells (per cell)
//   R/gficf.R:100-103 per-cell L2, Inf -> 0                 /
// HBM-bound sparse scaling in f64, no MFMA.  The only global dependency is the per-gene
// cell count nt_g, so the matrix is read twice: once to count (pass A), once to scale
// (pass B); the row subset of R/gficf.R:41 is a stream compaction fused into pass B whose
// output offsets come from a per-cell kept-count pass + scan.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <thread>

#include "common.h"

namespace {

#include "gficf_count.h"
#include "gficf_scale.h"
#include "gficf_signatures.h"

}  // namespace

// ----------------------------------------------------------------------------- C ABI
extern "C" {

// sum_mode (LDS-histogram form): 0 = d_nt[g] += the row sum (the C ABI's count step), 1 = d_nt[g] = it (d_nt need not be
// zeroed), 2 = no row sum here: the caller runs k_nt_sum_table on *part_out / *rows_out.
static int launch_count(gficf_ctx* ctx, int64_t G, const int32_t* d_rowidx, const double* d_x, int64_t nnz, int64_t* d_nt,
                        int sum_mode = 0, uint32_t** part_out = nullptr, int* rows_out = nullptr) {
  const bool overwrite = sum_mode == 1;
  int64_t blocks = gficf_ceil_div(nnz, (int64_t)CNT_THREADS * 16);
  // 16 B vector loads need 16 B-aligned bases (slab starts are multiples of 16384 entries)
  const bool vec = (((uintptr_t)d_rowidx | (uintptr_t)d_x) & 15u) == 0;
  const bool lds_hist = G <= CNT_LDS_MAX_G;
  size_t lds = 0;
  const int64_t Gp = (G + 3) & ~(int64_t)3;
  uint32_t* d_part = nullptr;
  if (lds_hist) {
    // LDS histogram: G counters per workgroup; as many workgroups per CU as LDS allows
    lds = (size_t)Gp * sizeof(uint32_t);
    int per_cu = (int)((160 * 1024) / (lds + 256));
    per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;
    if (blocks > (int64_t)ctx->num_cus * per_cu) blocks = (int64_t)ctx->num_cus * per_cu;
    {
      // every workgroup flushes its G counters as a row of the partial table (written here, read by the row sum): on a small input
      // that traffic exceeds the input's own (config 2: 512 rows x 80 KB against 57 MB of row indices) — bounded to a quarter of the
      // bytes read: configs 1 / 2 pass 25.5 -> 24.0 / 115.5 -> 108.9 us, config 3 and up unchanged (tools/lab/count_flush_probe.py;
      // GFICF_COUNT_FLUSH_RATIO: the A/B hook, 0 = no bound; read per call)
      const char* const e = getenv("GFICF_COUNT_FLUSH_RATIO");
      const int64_t ratio = e ? atoll(e) : 4;
      if (ratio > 0) {
        int64_t cap = nnz / (ratio * Gp);
        if (cap < 32) cap = 32;
        if (blocks > cap) blocks = cap;
      }
    }
    static std::atomic<bool> attr_set[64];
    if (!attr_set[ctx->device & 63]) {
      const int mx = CNT_LDS_MAX_G * (int)sizeof(uint32_t);
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_gene_count<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_gene_count<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_gene_count<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_gene_count<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
      attr_set[ctx->device & 63] = true;
    }
    GFICF_HIP_CHECK(gficf_pool_get(ctx, 8, sizeof(uint32_t) * (size_t)Gp * (size_t)blocks, (void**)&d_part));
  } else if (blocks > (int64_t)ctx->num_cus * 2) {
    blocks = (int64_t)ctx->num_cus * 2;
  }
#define LAUNCH_CNT(L, V, X)                                                                                          \
  hipLaunchKernelGGL((k_gene_count<L, V, X>), dim3((unsigned)blocks), dim3(CNT_THREADS), lds, ctx->stream, d_rowidx, d_x, \
                     nnz, G, (unsigned long long*)d_nt, d_part, ctx->d_status)
  const int sel = (lds_hist ? 4 : 0) | (vec ? 2 : 0) | (d_x ? 1 : 0);
  switch (sel) {
    case 7: LAUNCH_CNT(true, true, true); break;
    case 6: LAUNCH_CNT(true, true, false); break;
    case 5: LAUNCH_CNT(true, false, true); break;
    case 4: LAUNCH_CNT(true, false, false); break;
    case 3: LAUNCH_CNT(false, true, true); break;
    case 2: LAUNCH_CNT(false, true, false); break;
    case 1: LAUNCH_CNT(false, false, true); break;
    default: LAUNCH_CNT(false, false, false); break;
  }
#undef LAUNCH_CNT
  if (part_out) { *part_out = d_part; *rows_out = (int)blocks; }
  if (lds_hist && sum_mode != 2) {
    if (overwrite)
      hipLaunchKernelGGL(k_nt_sum<false>, dim3((unsigned)gficf_ceil_div(G, 64)), dim3(NS_WAVES * 64), 0, ctx->stream, d_part, Gp,
                         (int)blocks, G, (unsigned long long*)d_nt);
    else
      hipLaunchKernelGGL(k_nt_sum<true>, dim3((unsigned)gficf_ceil_div(G, 64)), dim3(NS_WAVES * 64), 0, ctx->stream, d_part, Gp,
                         (int)blocks, G, (unsigned long long*)d_nt);
  }
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_csc_count_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                           const int32_t* d_rowidx, const double* d_x, int64_t nnz, int64_t* d_nt) {
  GFICF_CTX_ENTER(ctx);
  (void)d_colptr;
  if (G < 0 || n_cells < 0 || nnz < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (G > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "G = %lld exceeds int32 row indices", (long long)G);
  if (nnz == 0 || G == 0) return GFICF_OK;
  if (!d_rowidx || !d_x || !d_nt) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  return launch_count(ctx, G, d_rowidx, d_x, nnz, d_nt);
}

int gficf_csc_genes_device(gficf_ctx* ctx, int64_t G, int64_t N_total, const int64_t* d_nt, double prop_min,
                           double prop_max, const double* d_w_in, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w,
                           int64_t* d_gkept) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || N_total < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (G > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "G = %lld exceeds int32 row indices", (long long)G);
  if (!d_gkept || (G > 0 && (!d_nt || !d_keep || !d_genes || !d_w))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  const int64_t tiles = G > 0 ? gficf_ceil_div(G, GT_THREADS) : 1;
  hipLaunchKernelGGL(k_gene_table, dim3((unsigned)tiles), dim3(GT_THREADS), 0, ctx->stream, G, N_total, d_nt, prop_min, prop_max,
                     d_w_in, d_keep, d_genes, d_w, d_gkept, ctx->icf_type);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_csc_colptr_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                            const int32_t* d_rowidx, const uint8_t* d_keep, const int64_t* d_gkept,
                            int64_t* d_out_colptr) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || n_cells < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (!d_colptr || !d_out_colptr || !d_gkept || !d_keep) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  int64_t blocks = gficf_ceil_div(n_cells > 0 ? n_cells : 1, CC_THREADS / 64);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  const size_t lds = (size_t)((G + 31) / 32) * sizeof(uint32_t);     // G <= 2^31 -> at most 256 MiB: checked below
  if (lds > 64 * 1024) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "G = %lld too large for the LDS keep bitmask", (long long)G);
  hipLaunchKernelGGL(k_cell_kept_count, dim3((unsigned)blocks), dim3(CC_THREADS), lds, ctx->stream, G, n_cells, d_colptr,
                     d_rowidx, d_keep, d_gkept, d_out_colptr, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return gficf_exclusive_scan_i64(ctx, d_out_colptr, n_cells + 1);
}

static int scale_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int32_t* d_rowidx, const double* d_x,
                        int64_t nnz, const gficf_gene_entry* d_genes, const int64_t* d_gkept, const int64_t* d_out_colptr,
                        int32_t* d_out_rowidx, double* d_out_x, int64_t* d_out_end);

int gficf_csc_scale_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                           const int32_t* d_rowidx, const double* d_x, int64_t nnz, const gficf_gene_entry* d_genes,
                           const int64_t* d_gkept, const int64_t* d_out_colptr, int32_t* d_out_rowidx, double* d_out_x) {
  return scale_launch(ctx, G, n_cells, d_colptr, d_rowidx, d_x, nnz, d_genes, d_gkept, d_out_colptr, d_out_rowidx, d_out_x, nullptr);
}

/* The scaling pass in the pointerB / pointerE ("four-array") form of a compressed matrix: the kept entries of cell c are written
 * from position d_colptr[c] on — every cell compacts inside its own input range — and d_out_end[c] is the position behind its
 * last kept entry.  No global output positions are needed, so the kept-count pass and its scan (gficf_csc_colptr_device: a third
 * read of rowidx, 14 % of the pass at config 3) do not run.  For the device-resident chain: gficf_csc_transpose_be_device and
 * gficf_cluster_signatures_be_device read this form; the canonical compacted CSC stays what the host entries return. */
int gficf_csc_scale_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int32_t* d_rowidx,
                              const double* d_x, int64_t nnz, const gficf_gene_entry* d_genes, const int64_t* d_gkept,
                              int64_t* d_out_end, int32_t* d_out_rowidx, double* d_out_x) {
  if (!d_out_end && n_cells > 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (n_cells > 0 && nnz == 0) {                   // no stored entry: every cell ends where it begins (position 0)
    GFICF_CTX_ENTER(ctx);
    GFICF_HIP_CHECK(hipMemsetAsync(d_out_end, 0, sizeof(int64_t) * (size_t)n_cells, ctx->stream));
    return GFICF_OK;
  }
  return scale_launch(ctx, G, n_cells, d_colptr, d_rowidx, d_x, nnz, d_genes, d_gkept, d_colptr, d_out_rowidx, d_out_x, d_out_end);
}

static int scale_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int32_t* d_rowidx, const double* d_x,
                        int64_t nnz, const gficf_gene_entry* d_genes, const int64_t* d_gkept, const int64_t* d_out_colptr,
                        int32_t* d_out_rowidx, double* d_out_x, int64_t* d_out_end) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || n_cells < 0 || nnz < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (n_cells == 0 || nnz == 0) return GFICF_OK;
  if (!d_colptr || !d_rowidx || !d_x || !d_genes || !d_gkept || !d_out_colptr || !d_out_rowidx || !d_out_x)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  // Two variants, selected on the device by the number of kept genes (unknown to the host without a
  // sync): each kernel returns at once when the input is the other one's.
  static const bool force_global = getenv("GFICF_SCALE_FORCE_GLOBAL") != nullptr;   // test hook
  static const bool force_semi = getenv("GFICF_SCALE_FORCE_SEMI") != nullptr;       // test hook: LDS variant, weights from global memory
  const bool try_lds = sl_fits(G, 0) && !force_global;     // else not even the row ids fit LDS
  if (try_lds) {
    static std::atomic<bool> attr_set[64];
    if (!attr_set[ctx->device & 63]) {
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_scale_cells_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SL_LDS_BYTES));
      attr_set[ctx->device & 63] = true;
    }
    int64_t blocks = gficf_ceil_div(n_cells, SL_THREADS / 64);
    if (blocks > ctx->num_cus) blocks = ctx->num_cus;
    hipLaunchKernelGGL(k_scale_cells_lds, dim3((unsigned)blocks), dim3(SL_THREADS), SL_LDS_BYTES, ctx->stream, G, n_cells,
                       d_colptr, d_rowidx, d_x, d_genes, d_gkept, d_out_colptr, d_out_rowidx, d_out_x, ctx->norm_l1, ctx->cur_zero,
                       getenv("GFICF_SCALE_STATIC_CELLS") != nullptr ? 1 : 0,       // test hook, read per call (A/B inside one process)
                       force_semi ? 1 : 0, d_out_end);
  }
  // The LDS variant takes every input whose new row ids fit 16 bits (G_kept < 65535, decided on the device); with fewer
  // than 65535 genes that is every input, and the global-gather variant is not launched at all.
  if (try_lds && G < 0xFFFF) {
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  }
  int64_t blocks = n_cells;
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_scale_cells, dim3((unsigned)blocks), dim3(SC_THREADS), 0, ctx->stream, G, n_cells, d_colptr,
                     d_rowidx, d_x, d_genes, try_lds ? d_gkept : (const int64_t*)nullptr, d_out_colptr, d_out_rowidx, d_out_x,
                     ctx->norm_l1, ctx->cur_zero, d_out_end);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

static int signatures_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int64_t* d_col_end,
                             const int32_t* d_rowidx, const double* d_x, const int32_t* d_cluster, int32_t C, double* d_out);

int gficf_cluster_signatures_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                                    const int32_t* d_rowidx, const double* d_x, const int32_t* d_cluster, int32_t C,
                                    double* d_out) {
  return signatures_launch(ctx, G, n_cells, d_colptr, d_colptr ? d_colptr + 1 : nullptr, d_rowidx, d_x, d_cluster, C, d_out);
}

/* The same sums over a matrix in the pointerB / pointerE form (gficf_csc_scale_be_device): cell c's entries are
 * [d_col_begin[c], d_col_end[c]). */
int gficf_cluster_signatures_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_col_begin, const int64_t* d_col_end,
                                       const int32_t* d_rowidx, const double* d_x, const int32_t* d_cluster, int32_t C, double* d_out) {
  if (n_cells > 0 && G > 0 && C > 0 && !d_col_end) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  return signatures_launch(ctx, G, n_cells, d_col_begin, d_col_end, d_rowidx, d_x, d_cluster, C, d_out);
}

static int signatures_launch(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int64_t* d_col_end,
                             const int32_t* d_rowidx, const double* d_x, const int32_t* d_cluster, int32_t C, double* d_out) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || n_cells < 0 || C < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (G == 0 || C == 0) return GFICF_OK;
  if (!d_out) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  GFICF_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)G * (size_t)C, ctx->stream));
  if (n_cells == 0) return GFICF_OK;
  if (!d_colptr || !d_rowidx || !d_x || !d_cluster) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (G <= SIG_MAX_G && C <= 65535 && n_cells >= 1024) {
    // grouped form: scratch = start[C + 1] | cursor[C] | order[n_cells] from the context's pool (stream-ordered like
    // everything else the context enqueues)
    const size_t off_cur = ((size_t)(C + 1) * 8 + 255) & ~(size_t)255, off_ord = (off_cur + (size_t)C * 4 + 255) & ~(size_t)255;
    void* scratch = nullptr;
    GFICF_HIP_CHECK(gficf_pool_get(ctx, 3, off_ord + (size_t)n_cells * 4, &scratch));
    int64_t* const start = (int64_t*)scratch;
    uint32_t* const cursor = (uint32_t*)((char*)scratch + off_cur);
    int32_t* const order = (int32_t*)((char*)scratch + off_ord);
    GFICF_HIP_CHECK(hipMemsetAsync(scratch, 0, off_ord, ctx->stream));
    const unsigned cb = (unsigned)(gficf_ceil_div(n_cells, 256) < 512 ? gficf_ceil_div(n_cells, 256) : 512);
    hipLaunchKernelGGL(k_sig_count, dim3(cb), dim3(256), 0, ctx->stream, n_cells, d_cluster, C, start, ctx->d_status);
    const int rc = gficf_exclusive_scan_i64(ctx, start, (int64_t)C + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sig_fill, dim3((unsigned)gficf_ceil_div(n_cells, 256)), dim3(256), 0, ctx->stream, n_cells, d_cluster, C, start, cursor,
                       order);
    static std::atomic<bool> attr_set[64];
    if (!attr_set[ctx->device & 63]) {
      GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_sig_sum, hipFuncAttributeMaxDynamicSharedMemorySize, SIG_MAX_G * (int)sizeof(double)));
      attr_set[ctx->device & 63] = true;
    }
    // slices per cluster: enough workgroups to fill the chip a few times over, whatever the number of clusters
    int64_t slices = gficf_ceil_div((int64_t)ctx->num_cus * 4, (int64_t)C);
    if (slices < 1) slices = 1;
    if (slices > 1024) slices = 1024;
    hipLaunchKernelGGL(k_sig_sum, dim3((unsigned)slices, (unsigned)C), dim3(256), (size_t)G * sizeof(double), ctx->stream, G, d_colptr, d_col_end, d_rowidx, d_x,
                       start, order, d_out);
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  }
  int64_t blocks = gficf_ceil_div(n_cells, 4);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_cluster_signatures, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, G, n_cells, d_colptr, d_col_end, d_rowidx,
                     d_x, d_cluster, C, d_out, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_cluster_signatures_host(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr, int colptr_is_i64,
                                  const int32_t* rowidx, const double* x, const int32_t* cluster, int32_t C, double* out) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || N < 0 || C < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (G == 0 || C == 0) return GFICF_OK;
  if (!colptr || !out || (N > 0 && !cluster)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  std::vector<int64_t> cp((size_t)N + 1);
  for (int64_t c = 0; c <= N; ++c)
    cp[(size_t)c] = colptr_is_i64 ? ((const int64_t*)colptr)[c] : (int64_t)((const int32_t*)colptr)[c];
  for (int64_t c = 0; c < N; ++c)
    if (cp[(size_t)c + 1] < cp[(size_t)c] || cp[0] != 0) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr not monotone at cell %lld", (long long)c);
  const int64_t nnz = cp[(size_t)N];
  if (nnz > 0 && (!rowidx || !x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  const size_t nsz = (size_t)(nnz > 0 ? nnz : 1), csz = (size_t)(N > 0 ? N : 1);
  gficf_arena ar;                                   // pool slot 0 (the device form takes slot 3 for its own scratch)
  const size_t o_cp = ar.take(sizeof(int64_t) * ((size_t)N + 1)), o_ri = ar.take(sizeof(int32_t) * nsz), o_x = ar.take(sizeof(double) * nsz);
  const size_t o_cl = ar.take(sizeof(int32_t) * csz), o_out = ar.take(sizeof(double) * (size_t)G * (size_t)C);
  hipError_t e = ar.bind(ctx, 0);
  int64_t* const d_cp = ar.at<int64_t>(o_cp); int32_t* const d_ri = ar.at<int32_t>(o_ri); double* const d_x = ar.at<double>(o_x);
  int32_t* const d_cl = ar.at<int32_t>(o_cl); double* const d_out = ar.at<double>(o_out);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cp, cp.data(), sizeof(int64_t) * cp.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_ri, rowidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && N > 0) e = hipMemcpyAsync(d_cl, cluster, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, ctx->stream);
  int rc = GFICF_OK;
  if (e == hipSuccess) {
    rc = gficf_cluster_signatures_device(ctx, G, N, d_cp, d_ri, d_x, d_cl, C, d_out);
    if (!rc) e = hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)G * (size_t)C, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_cluster_signatures_host: %s", hipGetErrorString(e));
  return rc;
}

size_t gficf_csc_genes_bytes(int64_t G) {
  const size_t g = (size_t)(G > 0 ? G : 1);
  return g * sizeof(gficf_gene_entry) + g * sizeof(double) + ((g * sizeof(uint16_t) + 63) & ~(size_t)63) + 64;
}

static int csc_sequence(gficf_ctx* ctx, bool exact, int64_t G, int64_t N, const int64_t* d_colptr, const int32_t* d_rowidx,
                        const double* d_x, int64_t nnz, double prop_min, double prop_max, const double* d_w_in,
                        int64_t* d_nt, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept,
                        int64_t* d_out_colptr, int32_t* d_out_rowidx, double* d_out_x, bool be = false) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || N < 0 || nnz < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (G > 0 && !d_nt) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  // the LDS-histogram form of the count writes every counter itself (k_nt_sum): no fill launch in front of it
  const bool count_writes_all = nnz > 0 && G > 0 && G <= CNT_LDS_MAX_G;
  if (G > 0 && !count_writes_all) GFICF_HIP_CHECK(hipMemsetAsync(d_nt, 0, sizeof(int64_t) * (size_t)G, ctx->stream));
  int rc = GFICF_OK;
  bool table_done = false;
  if (nnz > 0 && G > 0) {
    if (!d_rowidx || !d_x || !d_nt) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
    if (count_writes_all) {
      // count, then row sum + gene table in ONE launch (k_nt_sum_table)
      if (!d_gkept || !d_keep || !d_genes || !d_w) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
      uint32_t* d_part = nullptr;
      int rows = 0;
      rc = launch_count(ctx, G, d_rowidx, exact ? d_x : (const double*)nullptr, nnz, d_nt, 2, &d_part, &rows);
      const int64_t tiles = gficf_ceil_div(G, 64);
      uint32_t epoch = 0;
      if (!rc) rc = gficf_ws_next_epoch(ctx, tiles, &epoch);
      if (!rc) {
        hipLaunchKernelGGL(k_nt_sum_table, dim3((unsigned)tiles), dim3(NS_WAVES * 64), 0, ctx->stream, d_part, (G + 3) & ~(int64_t)3, rows, G, N,
                           prop_min, prop_max, d_w_in, (unsigned long long*)d_nt, d_keep, d_genes, d_w, d_gkept, ctx->icf_type,
                           (unsigned long long*)ctx->d_ws, epoch);
        GFICF_HIP_CHECK(hipGetLastError());
        table_done = true;
      }
    } else {
      rc = launch_count(ctx, G, d_rowidx, exact ? d_x : (const double*)nullptr, nnz, d_nt, 0);
    }
  }
  if (!rc && !table_done) rc = gficf_csc_genes_device(ctx, G, N, d_nt, prop_min, prop_max, d_w_in, d_keep, d_genes, d_w, d_gkept);
  if (!rc && !be) rc = gficf_csc_colptr_device(ctx, G, N, d_colptr, d_rowidx, d_keep, d_gkept, d_out_colptr);
  ctx->cur_zero = exact ? nullptr : ctx->d_status;
  if (!rc && !be) rc = gficf_csc_scale_device(ctx, G, N, d_colptr, d_rowidx, d_x, nnz, d_genes, d_gkept, d_out_colptr, d_out_rowidx, d_out_x);
  if (!rc && be) rc = gficf_csc_scale_be_device(ctx, G, N, d_colptr, d_rowidx, d_x, nnz, d_genes, d_gkept, d_out_colptr, d_out_rowidx, d_out_x);
  ctx->cur_zero = nullptr;
  return rc;
}

/* gficf_csc_device / gficf_csc_exact_device with the output in the pointerB / pointerE form (gficf_csc_scale_be_device): three
 * launches — count, gene table, scale — instead of five; d_out_end[N] takes the place of d_out_colptr[N + 1]. */
int gficf_csc_be_device(gficf_ctx* ctx, int exact, int64_t G, int64_t N, const int64_t* d_colptr, const int32_t* d_rowidx,
                        const double* d_x, int64_t nnz, double prop_min, double prop_max, const double* d_w_in,
                        int64_t* d_nt, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept,
                        int64_t* d_out_end, int32_t* d_out_rowidx, double* d_out_x) {
  return csc_sequence(ctx, exact != 0, G, N, d_colptr, d_rowidx, d_x, nnz, prop_min, prop_max, d_w_in, d_nt, d_keep, d_genes, d_w, d_gkept,
                      d_out_end, d_out_rowidx, d_out_x, true);
}

/* Fast sequence: pass A counts stored entries without reading x (4 B/nnz instead of 12).  That equals
 * nt_g = #{x != 0} unless the matrix stores explicit zeros, which the scaling pass checks for free (it reads every x
 * anyway): it then raises a deferred status and the next gficf_ctx_sync() returns GFICF_ERR_EXPLICIT_ZEROS — the outputs
 * are to be discarded and gficf_csc_exact_device called instead.  (Round 1 enqueued the exact sequence behind every
 * call, gated on a device flag: eight launches that returned at once in the common case, 6 % of the pass.) */
int gficf_csc_device(gficf_ctx* ctx, int64_t G, int64_t N, const int64_t* d_colptr, const int32_t* d_rowidx,
                     const double* d_x, int64_t nnz, double prop_min, double prop_max, const double* d_w_in,
                     int64_t* d_nt, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept,
                     int64_t* d_out_colptr, int32_t* d_out_rowidx, double* d_out_x) {
  return csc_sequence(ctx, false, G, N, d_colptr, d_rowidx, d_x, nnz, prop_min, prop_max, d_w_in, d_nt, d_keep, d_genes, d_w, d_gkept,
                      d_out_colptr, d_out_rowidx, d_out_x);
}

/* Exact sequence: pass A reads x and counts the non-zero entries (12 B/nnz); explicit zeros are handled. */
int gficf_csc_exact_device(gficf_ctx* ctx, int64_t G, int64_t N, const int64_t* d_colptr, const int32_t* d_rowidx,
                           const double* d_x, int64_t nnz, double prop_min, double prop_max, const double* d_w_in,
                           int64_t* d_nt, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept,
                           int64_t* d_out_colptr, int32_t* d_out_rowidx, double* d_out_x) {
  return csc_sequence(ctx, true, G, N, d_colptr, d_rowidx, d_x, nnz, prop_min, prop_max, d_w_in, d_nt, d_keep, d_genes, d_w, d_gkept,
                      d_out_colptr, d_out_rowidx, d_out_x);
}

}  // extern "C"

// ------------------------------------------------------------------- host form (R glue)
// Device buffers are pieces of the context's pool (slot 4 for the plan, slot 7 for the outputs of the finish
// call): kept between calls, nothing is allocated or freed per call.
struct gficf_host_plan {
  int64_t G = 0, N = 0, nnz = 0, nnz_kept = 0, g_kept = 0;
  int colptr_is_i64 = 0;
  int64_t* d_colptr = nullptr;
  int32_t* d_rowidx = nullptr;
  double* d_x = nullptr;
  double* d_w_in = nullptr;
  int64_t* d_nt = nullptr;
  uint8_t* d_keep = nullptr;
  gficf_gene_entry* d_genes = nullptr;
  double* d_w = nullptr;
  int64_t* d_gkept = nullptr;
  int64_t* d_out_colptr = nullptr;
};

void gficf_host_plan_free(gficf_ctx* ctx) {
  delete ctx->plan;
  ctx->plan = nullptr;
}

// ------------------------------------------------------------ the values of M[keep, ]
// normCounts' row subsetting (reference R/gficf.R:40) is what gficf() stores as $rawCounts (:22).  The filtered matrix has the structure
// of the GF-ICF result itself (the same kept entries in the same order: new colptr, renumbered row ids), so only its VALUES are missing:
// the x of the kept entries.  They never leave the host — the caller's x is there already and the result must end there: host threads
// stream them from the caller's vectors (12 B read per stored entry, 8 B written per kept one) while the device scales and the results
// cross PCIe, instead of another 8 B per kept entry over PCIe.  (R's `M[keep, ]` on a 60 M-entry dgCMatrix, or scipy's, takes ~150 ms.)
struct ColPtr {
  const void* p;
  int is64;
  int64_t operator[](int64_t c) const { return is64 ? ((const int64_t*)p)[c] : (int64_t)((const int32_t*)p)[c]; }
};

// returns GFICF_OK or GFICF_ERR_BAD_CSC (some cell's kept entries do not fill [kept_colptr[c], kept_colptr[c+1]) exactly); sets no message:
// it may run on a helper thread
static int kept_values(int64_t G, int64_t N, ColPtr cp, const int32_t* rowidx, const double* x, const uint8_t* keep, ColPtr kcp,
                       int32_t* out_rowidx, double* out_x) {
  const int64_t nnz = N > 0 ? cp[N] : 0, nk = N > 0 ? kcp[N] : 0;
  if (nnz <= 0 || G <= 0) return nk == 0 ? GFICF_OK : GFICF_ERR_BAD_CSC;      // (no gene: nothing is kept, and keep[] has no element to read)
  std::vector<int32_t> remap;
  if (out_rowidx) {
    remap.resize((size_t)G);
    int32_t r = 0;
    for (int64_t g = 0; g < G; ++g) remap[(size_t)g] = keep[g] ? r++ : -1;
  }
  const int32_t* const rm = out_rowidx ? remap.data() : nullptr;
  gficf_advise_hugepages(out_x, sizeof(double) * (size_t)nk);
  if (out_rowidx) gficf_advise_hugepages(out_rowidx, sizeof(int32_t) * (size_t)nk);
  int64_t nt = nnz / 2000000;
  const int64_t hw = (int64_t)std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;
  if (hw > 0 && nt > hw) nt = hw;
  if (nt < 1) nt = 1;
  std::atomic<int> bad{0};
  // a share = the cells whose first entry lies in its share of the stored entries
  auto first_cell = [&](int64_t target) {
    int64_t lo = 0, hi = N;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (cp[mid] >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  // Per cell: count the kept entries first (the cell is a few KB, hot in L1 for the second sweep) and compare with the cell's slot — nothing
  // is written for a matrix that is not the plan's — then a sweep WITHOUT a branch on the keep flag (taken 70 / 30 at the 5 % filter): every
  // entry is stored at the running position and the position moves on only for a kept one.  The sweep ends with the cell's last kept entry
  // (d == dend: the entries behind it are all dropped), so every store lands inside the cell's own slot.
  gficf_run_shares(nt, [&](int64_t t) {
    const int64_t c0 = t == 0 ? 0 : first_cell(nnz / nt * t), c1 = t + 1 == nt ? N : first_cell(nnz / nt * (t + 1));
    const uint32_t Gu = (uint32_t)(G > 0xFFFFFFFFll ? 0xFFFFFFFFll : G);
    for (int64_t c = c0; c < c1; ++c) {
      int64_t d = kcp[c];
      const int64_t dend = kcp[c + 1], q0 = cp[c], q1 = cp[c + 1];
      int64_t cnt = 0;
      for (int64_t q = q0; q < q1; ++q) {
        const uint32_t g = (uint32_t)rowidx[q];
        cnt += (int64_t)((g < Gu) & (keep[g < Gu ? g : 0] != 0));
      }
      if (cnt != dend - d) { bad.store(1); return; }
      if (rm) {
        for (int64_t q = q0; q < q1 && d < dend; ++q) {
          const uint32_t g = (uint32_t)rowidx[q], gg = g < Gu ? g : 0;
          __builtin_nontemporal_store(x[q], out_x + d);
          __builtin_nontemporal_store(rm[gg], out_rowidx + d);
          d += (int64_t)((g < Gu) & (keep[gg] != 0));
        }
      } else {
        for (int64_t q = q0; q < q1 && d < dend; ++q) {
          const uint32_t g = (uint32_t)rowidx[q], gg = g < Gu ? g : 0;
          __builtin_nontemporal_store(x[q], out_x + d);
          d += (int64_t)((g < Gu) & (keep[gg] != 0));
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);      // the non-temporal stores are out before the share is reported done
  });
  return bad.load() ? GFICF_ERR_BAD_CSC : GFICF_OK;
}

#define PLAN_HIP(expr)                                                                              \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      gficf_set_error("%s failed: %s", #expr, hipGetErrorString(_e));                               \
      (void)hipStreamSynchronize(ctx->stream);                                                      \
      gficf_host_plan_free(ctx);                                                                    \
      return GFICF_ERR_HIP;                                                                         \
    }                                                                                               \
  } while (0)

extern "C" {

int gficf_normalize_csc_host_plan(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr, int colptr_is_i64,
                                  const int32_t* rowidx, const double* x, double prop_min, double prop_max,
                                  const double* w_in, int64_t* G_kept, int64_t* nnz_kept) {
  GFICF_CTX_ENTER(ctx);
  if (G < 0 || N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative dimension");
  if (!colptr || !G_kept || !nnz_kept) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  gficf_host_plan_free(ctx);
  std::vector<int64_t> cp((size_t)N + 1);
  for (int64_t c = 0; c <= N; ++c)
    cp[(size_t)c] = colptr_is_i64 ? ((const int64_t*)colptr)[c] : (int64_t)((const int32_t*)colptr)[c];
  if (cp[0] != 0) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr[0] = %lld, expected 0", (long long)cp[0]);
  for (int64_t c = 0; c < N; ++c)
    if (cp[(size_t)c + 1] < cp[(size_t)c]) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr not monotone at cell %lld", (long long)c);
  const int64_t nnz = cp[(size_t)N];
  if (nnz > 0 && (!rowidx || !x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  gficf_host_plan* p = new gficf_host_plan();
  ctx->plan = p;
  p->G = G; p->N = N; p->nnz = nnz; p->colptr_is_i64 = colptr_is_i64;
  const size_t gsz = (size_t)(G > 0 ? G : 1), nsz = (size_t)(nnz > 0 ? nnz : 1);
  gficf_arena ar;
  const size_t o_cp = ar.take(sizeof(int64_t) * ((size_t)N + 1)), o_ri = ar.take(sizeof(int32_t) * nsz), o_x = ar.take(sizeof(double) * nsz);
  const size_t o_nt = ar.take(sizeof(int64_t) * gsz), o_keep = ar.take(gsz), o_genes = ar.take(gficf_csc_genes_bytes(G));
  const size_t o_w = ar.take(sizeof(double) * gsz), o_gk = ar.take(sizeof(int64_t)), o_ocp = ar.take(sizeof(int64_t) * ((size_t)N + 1));
  const size_t o_win = ar.take(sizeof(double) * gsz);
  PLAN_HIP(ar.bind(ctx, 4));
  p->d_colptr = ar.at<int64_t>(o_cp); p->d_rowidx = ar.at<int32_t>(o_ri); p->d_x = ar.at<double>(o_x);
  p->d_nt = ar.at<int64_t>(o_nt); p->d_keep = ar.at<uint8_t>(o_keep); p->d_genes = ar.at<gficf_gene_entry>(o_genes);
  p->d_w = ar.at<double>(o_w); p->d_gkept = ar.at<int64_t>(o_gk); p->d_out_colptr = ar.at<int64_t>(o_ocp);
  PLAN_HIP(hipMemcpyAsync(p->d_colptr, cp.data(), sizeof(int64_t) * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream));
  if (nnz > 0) {
    PLAN_HIP(hipMemcpyAsync(p->d_rowidx, rowidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    PLAN_HIP(hipMemcpyAsync(p->d_x, x, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  if (w_in && G > 0) {
    p->d_w_in = ar.at<double>(o_win);
    PLAN_HIP(hipMemcpyAsync(p->d_w_in, w_in, sizeof(double) * (size_t)G, hipMemcpyHostToDevice, ctx->stream));
  }
  PLAN_HIP(hipMemsetAsync(p->d_nt, 0, sizeof(int64_t) * gsz, ctx->stream));
  int rc = gficf_csc_count_device(ctx, G, N, p->d_colptr, p->d_rowidx, p->d_x, nnz, p->d_nt);
  if (!rc) rc = gficf_csc_genes_device(ctx, G, N, p->d_nt, prop_min, prop_max, p->d_w_in, p->d_keep, p->d_genes, p->d_w, p->d_gkept);
  if (!rc) rc = gficf_csc_colptr_device(ctx, G, N, p->d_colptr, p->d_rowidx, p->d_keep, p->d_gkept, p->d_out_colptr);
  int64_t hk[2] = {0, 0};
  if (!rc) {
    PLAN_HIP(hipMemcpyAsync(&hk[0], p->d_gkept, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PLAN_HIP(hipMemcpyAsync(&hk[1], p->d_out_colptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    rc = gficf_ctx_sync(ctx);
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  if (rc) { gficf_host_plan_free(ctx); return rc; }
  p->g_kept = hk[0];
  p->nnz_kept = hk[1];
  *G_kept = hk[0];
  *nnz_kept = hk[1];
  return GFICF_OK;
}

// the finish call; raw_rowidx / raw_x: NULL, or the caller's @i / @x again for the values of M[keep, ] (gficf_normalize_csc_host_finish_raw)
static int host_finish(gficf_ctx* ctx, uint8_t* keep, int64_t* nt, double* w, void* out_colptr, int32_t* out_rowidx, double* out_x,
                       const int32_t* raw_rowidx, const double* raw_x, int32_t* out_raw_rowidx, double* out_raw_x) {
  GFICF_CTX_ENTER(ctx);
  gficf_host_plan* p = ctx->plan;
  if (!p) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_normalize_csc_host_finish without a plan");
  if (!out_colptr || (p->nnz_kept > 0 && (!out_rowidx || !out_x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  const bool want_raw = out_raw_x != nullptr && p->nnz_kept > 0;
  if (want_raw && (!raw_rowidx || !raw_x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "the raw values of the kept rows need the matrix's rowidx and x again");
  const size_t ksz = (size_t)(p->nnz_kept > 0 ? p->nnz_kept : 1);
  gficf_arena ar;
  const size_t o_ri = ar.take(sizeof(int32_t) * ksz), o_x = ar.take(sizeof(double) * ksz);
  PLAN_HIP(ar.bind(ctx, 7));
  int32_t* const d_ori = ar.at<int32_t>(o_ri);
  double* const d_ox = ar.at<double>(o_x);
  std::vector<int64_t> cp((size_t)p->N + 1), cp_in;
  std::vector<uint8_t> keep_h;
  std::thread raw_thread;
  int raw_rc = GFICF_OK;
  if (want_raw) {
    // the new column pointers and the keep flags are the plan's (the stream is idle: the plan ended in a sync); the raw values of the
    // kept rows are then gathered by host threads from the caller's own vectors WHILE the scaling pass runs and its results come back
    cp_in.resize((size_t)p->N + 1);
    keep_h.resize((size_t)(p->G > 0 ? p->G : 1));
    PLAN_HIP(hipMemcpyAsync(cp.data(), p->d_out_colptr, sizeof(int64_t) * cp.size(), hipMemcpyDeviceToHost, ctx->stream));
    PLAN_HIP(hipMemcpyAsync(cp_in.data(), p->d_colptr, sizeof(int64_t) * cp_in.size(), hipMemcpyDeviceToHost, ctx->stream));
    if (p->G > 0) PLAN_HIP(hipMemcpyAsync(keep_h.data(), p->d_keep, (size_t)p->G, hipMemcpyDeviceToHost, ctx->stream));
    PLAN_HIP(hipStreamSynchronize(ctx->stream));
  }
  int rc = gficf_csc_scale_device(ctx, p->G, p->N, p->d_colptr, p->d_rowidx, p->d_x, p->nnz, p->d_genes, p->d_gkept,
                                  p->d_out_colptr, d_ori, d_ox);
  if (!rc && want_raw) {
    const int64_t G = p->G, N = p->N;
    const int64_t *ci = cp_in.data(), *co = cp.data();
    const uint8_t* kh = keep_h.data();
    int* const rcp = &raw_rc;
    auto job = [=] { *rcp = kept_values(G, N, ColPtr{ci, 1}, raw_rowidx, raw_x, kh, ColPtr{co, 1}, out_raw_rowidx, out_raw_x); };
    try { raw_thread = std::thread(job); } catch (...) { job(); }
  }
  // the caller's result vectors are freshly allocated as a rule: map their pages from several threads instead of one
  // page fault at a time under the device-to-host copy — the row indices while the scaling pass runs, the values (twice
  // as many bytes) on a helper thread while the row indices are being copied (a pageable copy holds the calling thread)
  std::thread fault_x;
  if (!rc && p->nnz_kept > 0) {
    gficf_prefault(out_rowidx, sizeof(int32_t) * (size_t)p->nnz_kept);
    double* const ox = out_x;
    const size_t xb = sizeof(double) * (size_t)p->nnz_kept;
    try { fault_x = std::thread([ox, xb] { gficf_prefault(ox, xb); }); } catch (...) { }
  }
  hipError_t e = hipSuccess;
  if (!rc) {
    if (!want_raw) e = hipMemcpyAsync(cp.data(), p->d_out_colptr, sizeof(int64_t) * cp.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && p->nnz_kept > 0) e = hipMemcpyAsync(out_rowidx, d_ori, sizeof(int32_t) * (size_t)p->nnz_kept, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && keep && p->G > 0) e = hipMemcpyAsync(keep, p->d_keep, (size_t)p->G, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && nt && p->G > 0) e = hipMemcpyAsync(nt, p->d_nt, sizeof(int64_t) * (size_t)p->G, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && w && p->G > 0) e = hipMemcpyAsync(w, p->d_w, sizeof(double) * (size_t)p->G, hipMemcpyDeviceToHost, ctx->stream);
    if (fault_x.joinable()) fault_x.join();
    if (e == hipSuccess && p->nnz_kept > 0) e = hipMemcpyAsync(out_x, d_ox, sizeof(double) * (size_t)p->nnz_kept, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) rc = gficf_ctx_sync(ctx);
  }
  if (fault_x.joinable()) fault_x.join();
  if (raw_thread.joinable()) raw_thread.join();
  if (e != hipSuccess || rc) (void)hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { gficf_set_error("HIP failure in gficf_normalize_csc_host_finish: %s", hipGetErrorString(e)); rc = GFICF_ERR_HIP; }
  if (!rc && raw_rc) {
    gficf_set_error("the rowidx / x handed to the finish call are not the matrix of the plan (the kept entries of some cell do not match its count)");
    rc = raw_rc;
  }
  if (!rc) {
    if (p->colptr_is_i64) std::memcpy(out_colptr, cp.data(), sizeof(int64_t) * cp.size());
    else for (size_t c = 0; c < cp.size(); ++c) ((int32_t*)out_colptr)[c] = (int32_t)cp[c];
  }
  gficf_host_plan_free(ctx);
  return rc;
}

int gficf_normalize_csc_host_finish(gficf_ctx* ctx, uint8_t* keep, int64_t* nt, double* w, void* out_colptr,
                                    int32_t* out_rowidx, double* out_x) {
  return host_finish(ctx, keep, nt, w, out_colptr, out_rowidx, out_x, nullptr, nullptr, nullptr, nullptr);
}

int gficf_normalize_csc_host_finish_raw(gficf_ctx* ctx, uint8_t* keep, int64_t* nt, double* w, void* out_colptr,
                                        int32_t* out_rowidx, double* out_x, const int32_t* rowidx, const double* x,
                                        int32_t* out_raw_rowidx, double* out_raw_x) {
  return host_finish(ctx, keep, nt, w, out_colptr, out_rowidx, out_x, rowidx, x, out_raw_rowidx, out_raw_x);
}

int gficf_csc_kept_values_host(int64_t G, int64_t N, const void* colptr, int colptr_is_i64, const int32_t* rowidx, const double* x,
                               const uint8_t* keep, const void* kept_colptr, int32_t* out_rowidx, double* out_x) {
  if (G < 0 || N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative dimension");
  if (N == 0) return GFICF_OK;
  if (!colptr || !kept_colptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  const ColPtr ci{colptr, colptr_is_i64}, co{kept_colptr, colptr_is_i64};
  // (a public entry without a context: the same checks the plan call makes — a pointer that does not start at 0 would index in front of the arrays)
  if (ci[0] != 0 || co[0] != 0) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr[0] = %lld, kept_colptr[0] = %lld: both must be 0", (long long)ci[0], (long long)co[0]);
  if (ci[N] > 0 && (!rowidx || !x || !keep)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  if (co[N] > 0 && !out_x) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  for (int64_t c = 0; c < N; ++c)
    if (ci[c + 1] < ci[c] || co[c + 1] < co[c]) GFICF_FAIL(GFICF_ERR_BAD_CSC, "column pointers not monotone at cell %lld", (long long)c);
  const int rc = kept_values(G, N, ci, rowidx, x, keep, co, out_rowidx, out_x);
  if (rc) GFICF_FAIL(rc, "kept_colptr is not the column pointer of M[keep, ] (the kept entries of some cell do not match its count)");
  return GFICF_OK;
}

}  // extern "C"
