/* gficf_hip.h — C ABI of libgficf_hip.so: MI355X (gfx950) GF-ICF normalisation and
 * Phenograph kNN -> Jaccard edge build, with the steps either side of them in clustcells()
 * (neighbour search, edge filter, adjacency matrix, community detection, cluster signatures)
 * and t(gficf).
 *
 * This is the drop-in boundary for ONE hot path of the dibbelab/gficf R package.  Plain
 * pointers and sizes only; no R, Rcpp or torch types.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repository).  The R
 * `.Call` glue a maintainer would add on top of it is in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns a gficf_status (0 = OK); gficf_last_error() gives the
 *     message for the calling thread's last failure.  Nothing here aborts or throws.
 *   - "host" entry points take caller-owned host buffers and do H2D / compute / D2H
 *     themselves (this is what the R glue binds).  "device" entry points take caller-
 *     owned device buffers (HBM-resident) and only enqueue work on the context's stream;
 *     call gficf_ctx_sync() to wait and to collect deferred input-validation errors.
 *   - ids in kNN matrices are 1-based, as R hands them over.
 *   - matrices are column-major (R layout) unless stated otherwise.
 */
#ifndef GFICF_HIP_H
#define GFICF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFICF_HIP_ABI_VERSION 6

typedef enum gficf_status {
  GFICF_OK = 0,
  GFICF_ERR_INVALID_ARG = 1,   /* NULL pointer, negative size, k out of range ...          */
  GFICF_ERR_BAD_ID = 2,        /* a kNN id outside [1, N] or not an integer (the reference
                                  has undefined behaviour there:
                                  src/rcpp_parallel_jaccard_coeff.cpp:28,34)               */
  GFICF_ERR_BAD_CSC = 3,       /* rowidx outside [0, G), colptr not monotone               */
  GFICF_ERR_NO_DEVICE = 4,     /* no HIP device / device index out of range                */
  GFICF_ERR_HIP = 5,           /* a HIP runtime call failed; message has hipGetErrorString */
  GFICF_ERR_UNSUPPORTED = 6,   /* k > GFICF_JACCARD_MAX_K_EXACT, table too large for the kernel,
                                  edge weights too large for the Louvain fixed point ...   */
  GFICF_ERR_CAPACITY = 7,      /* caller-provided output buffer too small; halo request
                                  slots of the sharded Jaccard build too few (deferred)    */
  GFICF_ERR_BAD_VALUE = 8,     /* a non-finite coordinate in the kNN point matrix, a
                                  negative / non-finite edge weight (Louvain)             */
  GFICF_ERR_DUPLICATE_IDS = 10, /* Jaccard with gficf_ctx_set_jaccard_distinct on: a row of the index matrix names an id
                                  twice (deferred); every edge computed from that table is to be discarded and the
                                  sequence re-run with the option off                                              */
  GFICF_ERR_EXPLICIT_ZEROS = 9, /* gficf_csc_device met an explicitly stored zero: its count of
                                  stored entries is then not rowSums(M != 0); discard the
                                  outputs and call gficf_csc_exact_device                  */
  GFICF_ERR_SET_OVERFLOW = 11  /* ABI 7.  Jaccard with gficf_ctx_set_jaccard_distinct on, 56 < k <= 256: more than six ids of one row found
                                  both slots of their hash-set bucket taken (ids spread uniformly over many cells at k near 256; not what a
                                  kNN search returns).  NOT a repeated id: discard the edges and re-run with the option off, or on the
                                  sorted-row path (GFICF_JACCARD_SORTED_FROM=57).  The host entries re-run on the sorted-row path by
                                  themselves.  (Through ABI 6 this was reported as GFICF_ERR_DUPLICATE_IDS.)                        */
} gficf_status;

/* k <= GFICF_JACCARD_MAX_K neighbours per cell take the hash-set / bit-set edge kernels; the reference's loop has no limit
 * (src/rcpp_parallel_jaccard_coeff.cpp:26-36), so beyond it every Jaccard entry switches to an exact sort + search path on
 * "sorted" table rows (csrc/jaccard_sorted.h: rows sorted once at ingest, an edge is k binary searches; multiset / set
 * semantics as the reference's; slower per edge), up to GFICF_JACCARD_MAX_K_EXACT (the uint16 intersection counts). */
#define GFICF_JACCARD_MAX_K 256
#define GFICF_JACCARD_MAX_K_EXACT 65535

typedef struct gficf_ctx gficf_ctx;  /* one per (thread, device): stream, workspace, status */

/* ------------------------------------------------------------------ library / context */
int gficf_hip_abi_version(void);
/* Number of HIP devices (does not create a context on any of them). */
int gficf_device_count(int* count);
/* device: HIP device ordinal.  stream: a hipStream_t owned by the caller, or NULL for the
 * device's default (null) stream.  Work of every *_device call is enqueued there. */
int gficf_ctx_create(int device, void* stream, gficf_ctx** out);
void gficf_ctx_destroy(gficf_ctx* ctx);
/* Rebind the stream later work is enqueued on (e.g. torch's current stream). */
int gficf_ctx_set_stream(gficf_ctx* ctx, void* stream);
/* Wait for the stream, then report (and clear) deferred device-side validation errors
 * (GFICF_ERR_BAD_ID / GFICF_ERR_BAD_CSC / GFICF_ERR_BAD_VALUE / GFICF_ERR_UNSUPPORTED) of the
 * *_device calls enqueued since last sync. */
int gficf_ctx_sync(gficf_ctx* ctx);
const char* gficf_last_error(void);
/* Where the banner lines of the host entries go (the reference prints them with Rprintf,
 * src/rcpp_parallel_jaccard_coeff.cpp:61-64,75-78): NULL = stdout; the R glue passes a wrapper of Rprintf. */
int gficf_ctx_set_print(gficf_ctx* ctx, void (*fn)(const char* line));
/* Device memory of the host entry points comes from a grow-only pool kept by the context between calls (no
 * allocation per call); this releases it (and any unfinished plan). */
int gficf_ctx_trim(gficf_ctx* ctx);

/* ----------------------------------------------------------------------------- Jaccard
 * Replaces  SEXP _gficf_rcpp_parallel_jaccard_coef(SEXP mat, SEXP printOutput)
 *           (src/RcppExports.cpp:61-70, registered :89) ->
 *           rcpp_parallel_jaccard_coef()  (src/rcpp_parallel_jaccard_coeff.cpp:59-80) ->
 *           JCoefficient::operator()      (src/rcpp_parallel_jaccard_coeff.cpp:24-55).
 *
 * For every cell i in [0,N) and slot j in [0,k):  kk = idx(i,j);
 *   u = | multiset(row i) ∩ multiset(row kk) |          (std::set_intersection semantics)
 *   u > 0 : rmat row i*k+j = (i+1, kk, u / (2.0*k - u));   u == 0 : the row stays 0.
 */

/* Host form.  idx: N x k column-major with leading dimension ld (>= N), int32
 * (idx_is_f64 == 0; what uwot returns) or double (idx_is_f64 != 0; what Rcpp coerces to,
 * src/RcppExports.cpp:65).  rmat: caller-allocated (N*k) x 3 column-major doubles
 * (src/rcpp_parallel_jaccard_coeff.cpp:67), fully overwritten.
 * print_output mirrors the reference's printOutput banners (:61-64, :75-78) on stdout. */
int gficf_jaccard_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k,
                       int64_t ld, double* rmat, int print_output);

/* Strict drop-in mode for ids given as NON-INTEGER doubles (never the case for kNN output; default off: such a matrix is
 * rejected with GFICF_ERR_BAD_ID).  With truncate_noninteger_ids != 0 gficf_jaccard_host reproduces what the reference
 * does with them (src/rcpp_parallel_jaccard_coeff.cpp:28-46): the neighbour ROW is addressed by truncation,
 * kk = (int)(mat(i,j) - 1), while the two rows are intersected as the raw doubles they hold (3.2 and 3.7 both name row 3
 * but are different set elements), and column 2 of the result is kk + 1.  Accepted values: 0 < v < N + 1 (v in (0, 1)
 * truncates to row 0 like 1.0 does); anything else — the reference would read outside the matrix — is GFICF_ERR_BAD_ID.
 * A matrix whose doubles are all integer-valued takes the ordinary kernels; one with non-integer values an exact
 * all-pairs kernel on the doubles (one wave per cell; not tuned: the case is a curiosity).  Applies to
 * gficf_jaccard_host on this context (the entry the .Call binds); the R glue turns it on with GFICF_HIP_TRUNCATE_IDS=1.
 * The truncating kernel covers k <= GFICF_JACCARD_MAX_K (a matrix WITH non-integer values and more neighbours per cell is
 * GFICF_ERR_UNSUPPORTED); integer-valued ids have no such limit (see GFICF_JACCARD_MAX_K above). */
int gficf_ctx_set_jaccard_options(gficf_ctx* ctx, int truncate_noninteger_ids);

/* Rows of a kNN index matrix hold k DISTINCT ids — uwot / Annoy and the search of this library never return one twice — but
 * the reference does not require it: its std::set_intersection over the two sorted rows (src/rcpp_parallel_jaccard_coeff.cpp:
 * 38-46) treats them as multisets, and so does this library's exact path.  Knowing whether a row repeats an id took the ingest
 * an all-pairs scan per row (k^2 / 2 comparisons: 6 of its 10.7 us at 100 k x 30, 10 of 29 us at k = 50).  With assume_distinct != 0
 * the ingest (gficf_jaccard_ingest_device) skips that scan, and the edge kernels — which insert every row of their cell range
 * into a hash set anyway and see a repeated id there for free — raise a deferred GFICF_ERR_DUPLICATE_IDS at the next
 * gficf_ctx_sync instead.  The caller then DISCARDS every edge computed from that table and re-runs ingest + edges with the
 * option off (the same pattern as gficf_csc_device / GFICF_ERR_EXPLICIT_ZEROS).  The check is complete only if every row of
 * the table is the own row of some cell that is computed: one context over all cells (the single-device sequence), or cell
 * blocks over several contexts / ranks that between them cover every cell — the error then comes from the context that owns
 * the row, and EVERY context's edges are to be discarded.  The halo form follows the option too (gficf_jaccard_halo_ingest_device
 * and the mapped edge kernels); gficf_jaccard_ingest_local_device always scans.
 * The host entries of this header (gficf_jaccard_host, _counts_host, _filtered_host_plan, gficf_jaccard_coeff_host) run the
 * fast sequence and re-run the exact one by themselves when it is needed: their results are the reference's for every input.
 * Default: off. */
int gficf_ctx_set_jaccard_distinct(gficf_ctx* ctx, int assume_distinct);

/* Small problems in ONE launch (round 5; csrc/jaccard_direct.h).  With gficf_ctx_set_jaccard_distinct on, gficf_jaccard_device (and
 * with it gficf_jaccard_host / gficf_jaccard_counts_host, which turn the option on themselves) builds a problem of at most
 * max_edges = N * k edges, k <= 32, straight from the column-major input in one kernel, without a table (d_table_ws is not
 * touched): at BASELINE config 1 (3 000 x 15) a step is bound by its two launches, not by bytes (6.2 us against 8.0).  A row that
 * repeats an id raises the same deferred GFICF_ERR_DUPLICATE_IDS as the table path.  max_edges < 0: the build's default — k <= 16
 * and N * k <= 65 536, the measured crossover (GFICF_JACCARD_DIRECT_MAX_EDGES in the environment overrides the edge count);
 * 0: never. */
int gficf_ctx_set_jaccard_direct_max_edges(gficf_ctx* ctx, int64_t max_edges);
/* 1 if gficf_jaccard_device on this context, as it is set up now, builds an N x k problem in that one launch; 0 otherwise. */
int gficf_jaccard_one_launch(gficf_ctx* ctx, int64_t N, int k);

/* Compact host return (the reference's 24 B row is a function of (i, idx[i,j], u)): the intersection counts alone,
 * u[i*k + j], 2 B per edge across PCIe instead of 24.  gficf_jaccard_expand_host rebuilds the reference's (N*k) x 3
 * matrix from them on the host (pure host code, several threads; n_threads <= 0: one per hardware thread, at most 16):
 * row i*k+j = (i+1, idx[i,j], u/(2.0*k-u)) when u > 0, zeros otherwise (src/rcpp_parallel_jaccard_coeff.cpp:48-52). */
int gficf_jaccard_counts_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, uint16_t* u);
int gficf_jaccard_expand_host(const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, const uint16_t* u, double* rmat,
                              int n_threads);

/* The package's second Jaccard entry:  SEXP _gficf_jaccard_coeff(SEXP idx, SEXP printOutput)
 *   (src/RcppExports.cpp:36-45, registered :87) -> jaccard_coeff() (src/jaccard_coeff.cpp:19-44), the serial form
 *   (not called by clustcells(), but exported).  Same edges and weights u / (2k - u); two differences:
 *   the intersection is Rcpp::intersect (:33), i.e. of the two rows as sets — it only differs from the parallel entry
 *   for rows that hold an id twice —, and rows with u > 0 are written one after the other from the top of the
 *   (N*k) x 3 matrix (`r++`, :36-41), the rows below stay zero.  weights: caller-allocated (N*k) x 3 column-major. */
int gficf_jaccard_coeff_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k,
                             int64_t ld, double* weights, int print_output);

/* Device-resident pipeline, split where a multi-GPU caller needs the seam:
 *   1. ingest : column-major ids of a block of cells -> row-major padded int32 table rows
 *   2. (multi-GPU only) the caller all-gathers the table rows of all blocks (RCCL)
 *   3. edges  : table + a block of cells -> that block's rows of rmat
 */

/* Slots of a table row for a given k: 16, 32, 64, 128 or 256 (k rounded up); k > 256: 2 * (k rounded up to 64) — the row
 * pitch of the sorted format (slot-order ids + the same ids ascending).  -1 beyond GFICF_JACCARD_MAX_K_EXACT. */
int gficf_jaccard_kpad(int k);
/* Row pitch of the table in 32-bit words, a function of (N_total, k) alone (every rank of a sharded build computes the
 * same): gficf_jaccard_kpad(k) words of 32-bit ids, or — for data sets of fewer than 2^17 cells, whose ids fit 17 bits,
 * and k <= kpad - kpad/16 (k = 30: 16 words = 64 B) — half of that: 16-bit low halves + one bitmap of high bits; for
 * 32 < k <= 55 and N_total <= 131070 (round 4) 64 words: that compact row plus a second copy of the ids regrouped for the
 * gathers of the bit-set edge kernel ("dual" rows; GFICF_JACCARD_DUAL=0 in the environment, read per call, keeps the plain
 * compact rows and the general kernel: an A/B switch — every caller of one table must see the same setting); for
 * k > GFICF_JACCARD_MAX_K gficf_jaccard_kpad(k) words (sorted rows).  The row layout is private to the library; callers
 * only size and slice the table by this pitch.  A buffer of N * kpad words always suffices. */
int gficf_jaccard_row_words(int64_t N_total, int k);

/* d_idx: n_rows x k column-major (ld >= n_rows) device matrix holding the neighbour ids
 * of cells [row0, row0+n_rows) of an N_total-cell data set.  Writes d_table_rows
 * (n_rows x gficf_jaccard_row_words(N_total, k) int32, row-major; caller passes the address of the block's first row).
 * Ids are validated against [1, N_total]; failures are reported by gficf_ctx_sync(). */
int gficf_jaccard_ingest_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64,
                                int64_t n_rows, int k, int64_t ld, int64_t N_total,
                                int32_t* d_table_rows);

/* Transport form of table rows for step 2 (the all-gather is what bounds the N > 1 path): k ids of
 * ceil(log2(N_total+1)) bits each + the row's duplicate flag, bit-packed into
 * gficf_jaccard_packed_words(N_total, k) 32-bit words per row (76 B instead of 128 B at k = 30,
 * N_total = 800 k).  pack: table rows -> packed rows; unpack: the inverse (pads zeroed).  Sorted rows (k > 256) travel as
 * they are: packed_words == row_words, pack / unpack are copies. */
int gficf_jaccard_packed_words(int64_t N_total, int k);
int gficf_jaccard_pack_rows_device(gficf_ctx* ctx, const int32_t* d_table_rows, int64_t n_rows, int k,
                                   int64_t N_total, uint32_t* d_packed);
int gficf_jaccard_unpack_rows_device(gficf_ctx* ctx, const uint32_t* d_packed, int64_t n_rows, int k,
                                     int64_t N_total, int32_t* d_table_rows);

/* d_table: the FULL N x row_words table.  Computes edges of cells [cell_begin, cell_end).
 * d_src/d_dst/d_w: three arrays of (cell_end-cell_begin)*k doubles — the block's slice of
 * the three columns of rmat (pass rmat + cell_begin*k, rmat + E + cell_begin*k,
 * rmat + 2E + cell_begin*k to land directly in the reference layout).
 * d_u: optional (may be NULL) int32 intersection counts, same indexing. */
int gficf_jaccard_edges_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k,
                               int64_t cell_begin, int64_t cell_end, double* d_src,
                               double* d_dst, double* d_w, int32_t* d_u);

/* ---- The sharded build on LOCAL ids ("halo" exchange; csrc/halo.hip).  What is sharded: the cells of the reference's
 * parallelFor(0, N, worker) (src/rcpp_parallel_jaccard_coeff.cpp:73); a block of cells needs its own rows and the rows it names
 * (:28-36).  When the ids have locality (cells in a spatial / cluster order) those are few, and a rank fetches just them
 * instead of taking part in an all-gather of the whole table:
 *   plan    (every rank)  marks the ids its block names outside itself and lists them per owner in P x cap request slots
 *                         (0 = empty): d_req_out, to be exchanged by an all-to-all with EQUAL splits of cap ids (no counts
 *                         to exchange, no host round trip).  A block that names more than cap rows of one owner raises
 *                         GFICF_ERR_CAPACITY at the next gficf_ctx_sync: the caller then uses the all-gather form;
 *   serve   (every rank)  copies the rows asked of it — raw global ids from ITS input block — into P x cap x k reply slots
 *                         (second all-to-all, equal splits of cap * k ids);
 *   relabel (every rank)  writes the index matrix of the rank's sub-problem, (k, n_ext) column-major with
 *                         n_ext = n_local + P * cap: own cells first, then the halo slots, all in local ids (own cell c ->
 *                         c - cell_begin + 1, halo slot q -> n_local + q + 1, anything else a halo row names -> 0: it is in
 *                         no own row), and d_l2g[n_ext], the global id of every local row (0 for an empty slot);
 *   then gficf_jaccard_ingest_local_device (ids in [0, n_ext], 0 = no id) and gficf_jaccard_edges_mapped_device (the first
 *   n_local rows' edges; column 1 = src_offset + cell + 1, column 2 through d_l2g) give the block's rows of the
 *   reference's matrix, bit for bit what the all-gather form gives.  With n_ext < 2^17 the sub-problem takes the compact
 *   64 B-row table whatever N_total is.  d_idx: the block's (k, ld) column-major int32 ids (what the sharded path carries);
 *   d_ws: gficf_jaccard_halo_workspace_bytes(N_total, P) bytes, written by plan and read by relabel. */
/* Round 4: the workspace is OWNER-ALIGNED (owner r's rows are the bits of its own run of words), so the plan is TWO launches —
 * mark, then one workgroup per owner that ranks the owner's bits, lists them in its request slots and hands the bitmap back
 * zero (no memset per step: the CALLER ZEROES d_ws ONCE, before its first plan; after that the library keeps it consistent) —
 * and rows_per_rank must be ceil(N_total / P), the pitch of gficf_multi_cell_blocks.  For k <= 64 the rest of a step is two more
 * launches around the second all-to-all: gficf_jaccard_halo_serve_ingest_device (the rows asked of this rank -> d_rows_out, and
 * IN THE SAME LAUNCH the table rows of the own cells, which need the plan but not the replies) and
 * gficf_jaccard_halo_ingest_slots_device (the table rows of the halo slots in use, from the replies: what is left between the
 * second exchange and the edge kernel).  gficf_jaccard_halo_ingest_device (all rows in one launch, behind both exchanges) and the
 * unfused serve / relabel / ingest_local calls stay. */
size_t gficf_jaccard_halo_workspace_bytes(int64_t N_total, int P);
int gficf_jaccard_halo_plan_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                   int64_t cell_begin, int P, int64_t rows_per_rank, int cap, void* d_ws, int32_t* d_req_out);
int gficf_jaccard_halo_serve_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t cell_begin,
                                    const int32_t* d_req_in, int64_t n_req, int32_t* d_rows_out);
int gficf_jaccard_halo_relabel_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                      int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                      const int32_t* d_rows_in, int32_t* d_idx_ext, int32_t* d_l2g);
int gficf_jaccard_ingest_local_device(gficf_ctx* ctx, const int32_t* d_idx_ext, int64_t n_ext, int k, int64_t ld, int32_t* d_table);
/* relabel + ingest_local in one launch, for k <= 64 (GFICF_ERR_UNSUPPORTED beyond: run the two calls above): the table of the
 * sub-problem straight from the block's global ids, the reply slots and the plan's workspace; writes d_l2g as relabel does. */
int gficf_jaccard_halo_ingest_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                     int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                     const int32_t* d_rows_in, int32_t* d_table, int32_t* d_l2g);
int gficf_jaccard_halo_serve_ingest_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                           int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                           const int32_t* d_req_in, int64_t n_req, int32_t* d_rows_out, int32_t* d_table, int32_t* d_l2g);
int gficf_jaccard_halo_ingest_slots_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                           int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                           const int32_t* d_rows_in, int32_t* d_table, int32_t* d_l2g);
/* The whole table of the sub-problem with NO exchange, in one launch behind the plan (one process whose devices map each other's
 * memory, see gficf_multi_jaccard_halo_device): the own cells' rows, and the row of every requested id read where it lies —
 * d_peer_idx[o]: owner o's block of global ids, a device pointer THIS device can dereference ((k, peer_ld[o]) column-major;
 * blocks of equal pitch rows_per_rank; both arrays are host arrays of P <= 16 entries). */
int gficf_jaccard_halo_ingest_peer_device(gficf_ctx* ctx, const int32_t* d_idx, int64_t n_local, int k, int64_t ld, int64_t N_total,
                                          int64_t cell_begin, int P, int64_t rows_per_rank, int cap, const void* d_ws, const int32_t* d_req_out,
                                          const int32_t* const* d_peer_idx, const int64_t* peer_ld, int32_t* d_table, int32_t* d_l2g);
int gficf_jaccard_edges_mapped_device(gficf_ctx* ctx, const int32_t* d_table, int64_t n_ext, int k, int64_t n_cells, int64_t src_offset,
                                      const int32_t* d_l2g, double* d_src, double* d_dst, double* d_w, int32_t* d_u);

/* Single-GPU convenience: ingest + edges, d_rmat in the reference layout ((N*k) x 3
 * column-major).  d_table_ws: caller-provided workspace of N*row_words int32 (kept by the
 * caller so that no allocation happens per call), d_u optional. */
int gficf_jaccard_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t N, int k,
                         int64_t ld, int32_t* d_table_ws, double* d_rmat, int32_t* d_u);

/* Edge filter of clustcells() fused on the device ("next" row N1): the caller's next line,
 *   relations <- relations[relations[,3] > 0, ]            (R/clustCells.R:66)
 * Only the edges with u > 0 are written, in the reference's row order (cell-major, slot order).
 * d_u_ws: workspace of (cell_end-cell_begin)*k uint16.  d_cell_ptr[n_cells+1]: offsets of every
 * source cell's edges in the output (d_cell_ptr[n_cells] = number of edges written).
 * d_from/d_to/d_weight: capacity (cell_end-cell_begin)*k doubles each. */
int gficf_jaccard_edges_filtered_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k,
                                        int64_t cell_begin, int64_t cell_end, uint16_t* d_u_ws,
                                        int64_t* d_cell_ptr, double* d_from, double* d_to,
                                        double* d_weight);
/* The same on a table built from RENUMBERED cells (row p of the table = original cell d_order[p], 0-based; ids inside the table in
 * the new numbering, 1-based): both columns come out in the ORIGINAL ids — from = d_order[p] + 1, to = d_order[id - 1] + 1 —, the
 * edges in the order of the new numbering.  For callers that renumber the cells by locality in front of the edge build (the pivot
 * order of the device search, gficf_knn_pivot_order_device: gficf_phenograph_host does for N >= 2^17) and hand the edges to a
 * consumer that does not care about their order (the adjacency build sorts them). */
int gficf_jaccard_edges_filtered_mapped_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k,
                                               int64_t cell_begin, int64_t cell_end, uint16_t* d_u_ws,
                                               int64_t* d_cell_ptr, double* d_from, double* d_to,
                                               double* d_weight, const int32_t* d_order);
/* Host form (two calls so that the caller can allocate exactly n_edges rows): R/clustCells.R:65-66.  idx must stay valid and unchanged
 * until finish has returned: from 2^20 edges on only the uint16 counts cross PCIe and finish writes the kept rows on the host from
 * (cell, idx[cell, slot], count) with several threads (round 5; the same rows, bit for bit). */
int gficf_jaccard_filtered_host_plan(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k,
                                     int64_t ld, int64_t* n_edges);
int gficf_jaccard_filtered_host_finish(gficf_ctx* ctx, double* from, double* to, double* weight);

/* Second half of N1, the graph hand-off: the kept edges as the symmetric weighted adjacency matrix that
 *   g <- igraph::graph.data.frame(relations, directed = FALSE)                       (R/clustCells.R:69)
 *   igraph::as_adjacency_matrix(g, attr = "weight", sparse = T)                       (R/clustCells.R:80,86)
 * hand to RunModularityClustering: A[i,j] = A[j,i] = sum of the weights of all edges between i and j (an i -> j and
 * a j -> i edge are two edges: a mutual pair carries 2w; a self edge counts once), cells in cell order, indices
 * sorted within a column, 0-based (dgCMatrix @p / @i / @x).
 * d_from / d_to / d_weight: edge list with capacity edge_capacity (1-based ids as doubles, what the filtered build
 * writes); d_n_edges: device count of valid edges (e.g. d_cell_ptr + n_cells of the filtered build) or NULL = all.
 * d_indptr[N+1] (d_indptr[N] = nnz), d_indices / d_x: capacity 2 * edge_capacity.  d_ws: scratch of
 * gficf_adjacency_workspace_bytes(N, edge_capacity) bytes. */
size_t gficf_adjacency_workspace_bytes(int64_t N, int64_t edge_capacity);
int gficf_adjacency_device(gficf_ctx* ctx, int64_t N, int64_t edge_capacity, const int64_t* d_n_edges,
                           const double* d_from, const double* d_to, const double* d_weight, void* d_ws,
                           size_t ws_bytes, int64_t* d_indptr, int32_t* d_indices, double* d_x);
/* Host form, two calls so that the caller can allocate exactly nnz entries. */
int gficf_adjacency_host_plan(gficf_ctx* ctx, int64_t N, int64_t n_edges, const double* from, const double* to,
                              const double* weight, int64_t* nnz);
int gficf_adjacency_host_finish(gficf_ctx* ctx, void* indptr, int indptr_is_i64, int32_t* indices, double* x);

/* ------------------------------------------------------------------------------ GF-ICF
 * Replaces the R-level chain of gficf()  (R/gficf.R:17-33, normalize = FALSE):
 *   normCounts filter R/gficf.R:40-41, tf R/gficf.R:59, getIdfW("classic") R/gficf.R:88-89,
 *   idf R/gficf.R:79, l.norm("l2") R/gficf.R:100-103 (applied per cell, R/gficf.R:25);
 * and, with w_in != NULL, the same chain as used by embedNewCells() R/cellClassifier.R:50-53.
 * The reference has no native entry for this path; this is the new one.
 *
 * Input: CSC genes x cells (dgCMatrix slots): colptr = @p (N+1 entries, int32 when
 * colptr_is_i64 == 0, else int64), rowidx = @i (0-based int32, sorted within a column),
 * x = @x (double).  Genes are kept iff  nt_g > N*prop_min  &&  nt_g <= N*prop_max  where
 * nt_g = #{cells with a non-zero entry}.  Output value of a stored kept entry:
 *     ((x / S_c) * w_g) * (1 / sqrt(sum_g' ((x'/S_c) * w_g')^2)),  S_c = sum of kept x of cell c,
 *     w_g = log((N+1)/(nt_g+1))   (or w_in[g]).
 * Cells with S_c == 0 get 0.0 in every stored kept entry (R would produce NaN).
 */

/* Options of the reference's internal helpers (gficf() itself always runs with the defaults 0, 0):
 *   icf_type: getIdfW(type = ...) R/gficf.R:89-91 — 0 "classic" log((N+1)/(nt+1)), 1 "prob" log((N-nt)/nt),
 *             2 "smooth" log(1 + N/nt);
 *   norm    : l.norm(norm = ...) R/gficf.R:100 — 0 "l2" 1/sqrt(sum v^2), 1 "l1" 1/sum v  (Inf -> 0 in both).
 * They apply to every later GF-ICF call on this context. */
int gficf_ctx_set_gficf_options(gficf_ctx* ctx, int icf_type, int norm);

/* Host form, two calls so that the caller (R glue) can allocate exactly-sized outputs:
 *   plan   : uploads the matrix, counts, filters; returns G_kept and nnz_kept.
 *   finish : writes keep[G] (0/1), nt[G] (raw count of every gene, dropped ones included),
 *            w[G] (0 for dropped genes), out_colptr[N+1]
 *            (same integer width as the input colptr), out_rowidx[nnz_kept] (renumbered
 *            over kept genes), out_x[nnz_kept]; releases the plan.  Any of keep/nt/w may
 *            be NULL.
 * w_in: NULL, or G weights indexed by original gene (ICF weights supplied). */
int gficf_normalize_csc_host_plan(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr,
                                  int colptr_is_i64, const int32_t* rowidx, const double* x,
                                  double prop_min, double prop_max, const double* w_in,
                                  int64_t* G_kept, int64_t* nnz_kept);
int gficf_normalize_csc_host_finish(gficf_ctx* ctx, uint8_t* keep, int64_t* nt, double* w,
                                    void* out_colptr, int32_t* out_rowidx, double* out_x);

/* gficf(storeRaw = TRUE) keeps the filtered counts, `$rawCounts` = normCounts' `M[keep, ]` (reference R/gficf.R:40,22).  That matrix
 * has the structure of the GF-ICF result itself — the same kept entries in the same order: out_colptr, out_rowidx — so only its values
 * are missing: the x of the kept entries.  They never leave the host (the caller's x is there and the result must end there):
 *   gficf_normalize_csc_host_finish_raw = the finish call + out_raw_x[nnz_kept] (and, if out_raw_rowidx is not NULL, a copy of
 *       the renumbered row ids of its own for a caller whose two matrices must not share a vector), gathered by host threads from
 *       `rowidx` / `x` — the SAME vectors the plan call was given — while the device scales and the results cross PCIe;
 *       GFICF_ERR_BAD_CSC if they are not the plan's matrix (some cell's kept entries do not match its count).  out_raw_x == NULL:
 *       exactly gficf_normalize_csc_host_finish.
 *   gficf_csc_kept_values_host = the same gather by itself, no device and no context involved (after the multi-GPU finish call, or
 *       for any `M[keep, ]` whose column pointer is known): kept_colptr (same integer width as colptr) must be the column pointer of
 *       M[keep, ]; out_rowidx may be NULL.  keep: G flags (0 / 1).  Both pointers must start at 0 and be monotone (GFICF_ERR_BAD_CSC);
 *       the outputs hold kept_colptr[N] entries — a cell whose kept entries do not fill its range exactly is GFICF_ERR_BAD_CSC, nothing is
 *       written outside [0, kept_colptr[N]). */
int gficf_normalize_csc_host_finish_raw(gficf_ctx* ctx, uint8_t* keep, int64_t* nt, double* w, void* out_colptr,
                                        int32_t* out_rowidx, double* out_x, const int32_t* rowidx, const double* x,
                                        int32_t* out_raw_rowidx, double* out_raw_x);
int gficf_csc_kept_values_host(int64_t G, int64_t N, const void* colptr, int colptr_is_i64, const int32_t* rowidx,
                               const double* x, const uint8_t* keep, const void* kept_colptr, int32_t* out_rowidx, double* out_x);

/* Device-resident pipeline (all pointers are device memory; colptr is int64 here), split
 * where a multi-GPU caller needs the seam (cells sharded by column block):
 *   1. count   : d_nt[g] += #{local cells with non-zero entry of gene g}   (d_nt zeroed by caller; a counter stays below 2^32)
 *   2. (multi-GPU only) caller all-reduces d_nt (sum) over ranks
 *   3. genes   : d_nt, N_total -> d_keep[G] (uint8), d_w[G], d_genes (the per-gene tables the
 *                scaling pass looks up: gficf_csc_genes_bytes(G) bytes, opaque), d_gkept[1] (int64)
 *   4. colptr  : per-cell kept counts + exclusive scan -> d_out_colptr[n_cells+1] (int64)
 *   5. scale   : writes d_out_rowidx / d_out_x (capacity >= kept nnz; nnz always suffices)
 */
typedef struct gficf_gene_entry {
  double w;         /* ICF weight of the gene (0 when dropped)        */
  int32_t remap;    /* row id among the kept genes, or -1 when dropped */
  int32_t reserved;
} gficf_gene_entry;

/* Size in bytes of the d_genes buffer for G genes: G 16-byte records {w, remap}, followed by the
 * compact tables the LDS-resident variant of the scaling pass stages (weights of the kept genes,
 * 16-bit new row ids). */
size_t gficf_csc_genes_bytes(int64_t G);

int gficf_csc_count_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                           const int32_t* d_rowidx, const double* d_x, int64_t nnz,
                           int64_t* d_nt);
int gficf_csc_genes_device(gficf_ctx* ctx, int64_t G, int64_t N_total, const int64_t* d_nt,
                           double prop_min, double prop_max, const double* d_w_in,
                           uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept);
int gficf_csc_colptr_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                            const int32_t* d_rowidx, const uint8_t* d_keep,
                            const int64_t* d_gkept, int64_t* d_out_colptr);
int gficf_csc_scale_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                           const int32_t* d_rowidx, const double* d_x, int64_t nnz,
                           const gficf_gene_entry* d_genes, const int64_t* d_gkept,
                           const int64_t* d_out_colptr, int32_t* d_out_rowidx, double* d_out_x);

/* Single-GPU convenience: steps 1,3,4,5 back to back on the context's stream.  Output
 * buffers need capacity nnz (upper bound); *d_out_colptr[n_cells] holds the kept nnz.
 * gficf_csc_device counts STORED entries in step 1 without reading x (4 B/nnz instead of 12), which is nt_g unless the
 * matrix stores explicit zeros; the scaling pass notices one for free and the next gficf_ctx_sync() then returns
 * GFICF_ERR_EXPLICIT_ZEROS: discard the outputs and call gficf_csc_exact_device (same arguments; step 1 reads x).
 * The host and multi-GPU entries always count exactly. */
int gficf_csc_device(gficf_ctx* ctx, int64_t G, int64_t N, const int64_t* d_colptr,
                     const int32_t* d_rowidx, const double* d_x, int64_t nnz, double prop_min,
                     double prop_max, const double* d_w_in, int64_t* d_nt, uint8_t* d_keep,
                     gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept, int64_t* d_out_colptr,
                     int32_t* d_out_rowidx, double* d_out_x);
int gficf_csc_exact_device(gficf_ctx* ctx, int64_t G, int64_t N, const int64_t* d_colptr,
                           const int32_t* d_rowidx, const double* d_x, int64_t nnz, double prop_min,
                           double prop_max, const double* d_w_in, int64_t* d_nt, uint8_t* d_keep,
                           gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept, int64_t* d_out_colptr,
                           int32_t* d_out_rowidx, double* d_out_x);

/* The device-resident chain in the pointerB / pointerE ("four-array") form of a compressed matrix — what sparse BLAS libraries
 * call CSC with separate column-begin and column-end arrays.  The canonical output above needs GLOBAL positions (the compacted
 * column pointer), i.e. a pass of its own over rowidx between the gene table and the scaling pass (gficf_csc_colptr_device: 14 % of
 * the whole pass at 88 M entries).  Here every cell compacts INSIDE ITS OWN INPUT RANGE: the kept entries of cell c are written
 * from position d_colptr[c] on and d_out_end[c] is the position behind the last one; (d_colptr, d_out_end, d_out_rowidx, d_out_x)
 * is the matrix — same entries, same order, same values as the canonical form, with unused room between the columns (the arrays
 * keep the input's nnz entries of capacity).  The steps that follow gficf() on the device read this form directly:
 * gficf_csc_transpose_be_device (t(gficf), whose result is an ordinary compact CSC) and gficf_cluster_signatures_be_device.
 * The host entries and gficf_csc_device keep returning the canonical form (R/gficf.R:17-33: a dgCMatrix).
 *   gficf_csc_scale_be_device : step 5 of the device pipeline in this form (no step 4);
 *   gficf_csc_be_device       : gficf_csc_device (exact == 0) / gficf_csc_exact_device (exact != 0) in this form: three launches. */
int gficf_csc_scale_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr, const int32_t* d_rowidx,
                              const double* d_x, int64_t nnz, const gficf_gene_entry* d_genes, const int64_t* d_gkept,
                              int64_t* d_out_end, int32_t* d_out_rowidx, double* d_out_x);
int gficf_csc_be_device(gficf_ctx* ctx, int exact, int64_t G, int64_t N, const int64_t* d_colptr, const int32_t* d_rowidx,
                        const double* d_x, int64_t nnz, double prop_min, double prop_max, const double* d_w_in,
                        int64_t* d_nt, uint8_t* d_keep, gficf_gene_entry* d_genes, double* d_w, int64_t* d_gkept,
                        int64_t* d_out_end, int32_t* d_out_rowidx, double* d_out_x);

/* ------------------------------------------------------------------- several GPUs behind the host entries
 * The reference's call sites are one `.Call` each (R/clustCells.R:65; gficf() R/gficf.R:17-33), so a drop-in that
 * shards over the GPUs of a node does it underneath that one call: single process, one context and one stream per
 * device, the partitioning of the multi-process path (gficf_amd/dist.py) —
 *   Jaccard: contiguous equal-pitch cell blocks; every device uploads and ingests its block of the kNN matrix, the table
 *            rows are exchanged device to device (hipMemcpyPeerAsync over xGMI; without peer access every device ingests the
 *            whole matrix), every device builds its block's edges and copies them into its three column slices of rmat;
 *   GF-ICF : contiguous cell blocks balanced by stored entries; the per-gene counts nt_g are summed across the devices
 *            through the host (G x 8 B each way), everything else is per block.
 * Same kernels and arithmetic as the single-device entries: same bits.  devices: HIP ordinals (NULL = 0 .. ndev-1; the
 * same ordinal may appear more than once — one block each, which is how the N > 1 path is tested on a one-GPU box).
 * The R glue reads the list from the environment variable GFICF_HIP_DEVICES ("0,1,2,3"). */
typedef struct gficf_multi gficf_multi;
int gficf_multi_create(const int* devices, int ndev, gficf_multi** out);
void gficf_multi_destroy(gficf_multi* m);
int gficf_multi_device_count(const gficf_multi* m);
int gficf_multi_set_print(gficf_multi* m, void (*fn)(const char* line));
/* The block arithmetic (pure host code): bounds[r] .. bounds[r+1] = cells of device r, ndev + 1 entries. */
int gficf_multi_cell_blocks(int64_t N, int ndev, int64_t* bounds);
int gficf_multi_cell_blocks_by_nnz(int64_t N, const void* colptr, int colptr_is_i64, int ndev, int64_t* bounds);
/* gficf_jaccard_host over the devices of m (same arguments, same result). */
int gficf_jaccard_host_multi(gficf_multi* m, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, double* rmat,
                             int print_output);
/* gficf_normalize_csc_host_plan / _finish over the devices of m (same arguments, same results). */
int gficf_normalize_csc_host_multi_plan(gficf_multi* m, int64_t G, int64_t N, const void* colptr, int colptr_is_i64,
                                        const int32_t* rowidx, const double* x, double prop_min, double prop_max,
                                        const double* w_in, int64_t* G_kept, int64_t* nnz_kept);
int gficf_normalize_csc_host_multi_finish(gficf_multi* m, uint8_t* keep, int64_t* nt, double* w, void* out_colptr,
                                          int32_t* out_rowidx, double* out_x);

/* The same sharded Jaccard step with EVERYTHING RESIDENT IN HBM (no host buffers): block r of the kNN matrix already lies on
 * device r of m (d_idx[r]: its (k, ld[r]) column-major ids, global 1-based; ld NULL = the block's row count); every device
 * ingests its block into its slice of ITS copy of the table (d_table[r]: N x gficf_jaccard_row_words(N, k) int32 on device r),
 * pulls the other P - 1 slices straight from the other devices with hipMemcpyPeerAsync — each pull on a copy stream of its own,
 * all pairs at once, ordered by events behind the owners' ingests: no collective, no host round trip — and builds the edges of
 * its block into d_out[r] (3 x n_r*k doubles on device r: the block's slices of the three columns of rmat).  Blocks are those of
 * gficf_multi_cell_blocks.  The call only ENQUEUES (on the contexts' own streams): the caller's inputs must be complete before
 * it, results and deferred input errors are collected by gficf_multi_sync.  Consecutive calls may reuse the same buffers (a
 * step's ingest waits for the pulls of the step before).  What is sharded: the cells of the reference's parallelFor(0, N, worker)
 * (src/rcpp_parallel_jaccard_coeff.cpp:73) under the one call of R/clustCells.R:64-65.
 * gficf_multi_set_jaccard_distinct: gficf_ctx_set_jaccard_distinct on every device's context — the blocks cover every cell,
 * so the deferred duplicate check stays complete; the error comes from gficf_multi_sync. */
int gficf_multi_jaccard_device(gficf_multi* m, const void* const* d_idx, int idx_is_f64, const int64_t* ld, int64_t N, int k,
                               int32_t* const* d_table, double* const* d_out);
/* The device-resident step for blocks whose ids have LOCALITY (a block names few rows outside itself — ids in a spatial or pivot
 * order): NOTHING is exchanged.  Every device plans the rows its block names outside (gficf_jaccard_halo_plan_device), builds the
 * table of its own sub-problem — its own cells' rows from its block, the requested rows READ WHERE THEY LIE, in the owners' blocks
 * of ids, through the peer mapping (needs peer access between every pair of devices; at most 16 devices; int32 ids; k <= 64) —
 * and the edges of its block: four launches per device (mark, rank + list, table, edges), no collective, no copy, no event between
 * devices.  Per device r: d_ws[r] gficf_jaccard_halo_workspace_bytes(N, P) bytes ZEROED ONCE by the caller, d_req[r] P * cap
 * int32, d_table[r] (n_r + P * cap) x gficf_jaccard_row_words(n_r + P * cap, k) int32, d_l2g[r] n_r + P * cap int32, d_out[r] as
 * in gficf_multi_jaccard_device.  A block that names more than cap rows of one owner raises GFICF_ERR_CAPACITY from
 * gficf_multi_sync (ids without locality: use gficf_multi_jaccard_device).  The step is POSTED to the context's per-device host
 * threads and this call returns before anything is enqueued, so the ONLY completion point is gficf_multi_sync: the blocks of ids (which
 * the other devices read through the peer mapping) and every buffer of the step must stay unchanged until it has returned — an event
 * the caller records after this call may still precede the step's launches and orders nothing.  The caller's inputs must be complete
 * (their producing streams synchronised) before the call.  d_ws is trusted to be all-zero at the first step (the plan hands it back
 * zero after every step); a workspace with stale bits gives wrong request slots without an error.  Same rows of rmat as the other
 * forms, bit for bit. */
int gficf_multi_jaccard_halo_device(gficf_multi* m, const int32_t* const* d_idx, const int64_t* ld, int64_t N, int k, int cap,
                                    void* const* d_ws, int32_t* const* d_req, int32_t* const* d_table, int32_t* const* d_l2g, double* const* d_out);
int gficf_multi_sync(gficf_multi* m);
int gficf_multi_set_jaccard_distinct(gficf_multi* m, int assume_distinct);

/* ------------------------------------------------------------------- cluster signatures
 * "Next" row N3: data$cluster.gene.rnk of clustcells() (R/clustCells.R:121-123):
 *   out[g, c] = sum over the cells of cluster c of gficf[g, cell]        (G x C doubles, column-major).
 * cluster[cell] in [0, C): the caller numbers the labels in order of first appearance, as
 * base::unique() does (R/clustCells.R:122).  f64 atomic adds: summation order is not fixed. */
int gficf_cluster_signatures_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                                    const int32_t* d_rowidx, const double* d_x,
                                    const int32_t* d_cluster, int32_t C, double* d_out);
/* the same over a matrix in the pointerB / pointerE form (cell c = [d_col_begin[c], d_col_end[c])) */
int gficf_cluster_signatures_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_col_begin, const int64_t* d_col_end,
                                       const int32_t* d_rowidx, const double* d_x, const int32_t* d_cluster, int32_t C, double* d_out);
int gficf_cluster_signatures_host(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr,
                                  int colptr_is_i64, const int32_t* rowidx, const double* x,
                                  const int32_t* cluster, int32_t C, double* out);

/* ------------------------------------------------------------------- transpose of the GF-ICF matrix
 * "Next" row N3, second half: data$pca$cells = t(data$gficf) (R/dimensinalityReduction.R:33, :100;
 * Matrix::t, third-party) — the genes x cells CSC matrix as cells x genes CSC (= its CSR form):
 *   out_ptr[G + 1], out_idx[nnz] (cell of every entry, ascending within a gene, 0-based), out_x[nnz].
 * Every stored entry is kept (explicit zeros too), as Matrix::t does.  A stable counting sort by
 * gene; d_ws: gficf_csc_transpose_workspace_bytes(G, n_cells) bytes of device scratch.  A row index
 * outside [0, G) or a bad colptr is reported by gficf_ctx_sync() (GFICF_ERR_BAD_CSC); the columns
 * must not name a gene twice (a dgCMatrix never does). */
size_t gficf_csc_transpose_workspace_bytes(int64_t G, int64_t n_cells);
int gficf_csc_transpose_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_colptr,
                               const int32_t* d_rowidx, const double* d_x, int64_t nnz,
                               int64_t* d_out_ptr, int32_t* d_out_idx, double* d_out_x, void* d_ws,
                               size_t ws_bytes);
/* the same of a matrix in the pointerB / pointerE form; `capacity` = entries the input arrays hold (the clamp for bad pointers);
 * the result is compact: d_out_ptr[G] entries. */
int gficf_csc_transpose_be_device(gficf_ctx* ctx, int64_t G, int64_t n_cells, const int64_t* d_col_begin, const int64_t* d_col_end,
                                  const int32_t* d_rowidx, const double* d_x, int64_t capacity, int64_t* d_out_ptr, int32_t* d_out_idx,
                                  double* d_out_x, void* d_ws, size_t ws_bytes);
int gficf_csc_transpose_host(gficf_ctx* ctx, int64_t G, int64_t N, const void* colptr,
                             int colptr_is_i64, const int32_t* rowidx, const double* x,
                             int64_t* out_ptr, int32_t* out_idx, double* out_x);

/* ------------------------------------------------------------------- community detection (Louvain)
 * "Next" row N4: RunModularityClustering(adjacency, 1, resolution, 1, n.start, n.iter, seed, verbose)
 * (R/clustCells.R:80,86 -> src/RModularityOptimizer.cpp:25-181 -> src/ModularityOptimizer.cpp:594-612) on the
 * symmetric weighted adjacency matrix that gficf_adjacency_* produce (CSC == CSR; the diagonal is ignored, as the
 * reference driver does).  RELAXED CONTRACT: the reference moves vertices one at a time in a seeded random order;
 * this is a deterministic parallel Louvain on the same objective,
 *   Q = (1/2W) * [ sum_ij A_ij delta(c_i, c_j) - resolution * sum_c K_c^2 / 2W ]   (calcQualityFunction, :461-482),
 * whose modularity is tested to lie within a stated tolerance of the reference's own, not label-for-label parity.
 * labels[v] in [0, *n_clusters), clusters numbered by decreasing size, ties by first vertex order
 * (Clustering::orderClustersByNNodes, :132-158); *modularity = Q of the returned labels.  n_iter >= 1 is the
 * reference's nIterations (a further pass restarts from the labels found so far and stops when nothing moves);
 * algorithm 1 = Louvain, 2 = Louvain with multilevel refinement (one more local moving on every level on the way back
 * up, :629-649).  n_start >= 1 "random starts" (:108-142): every start begins from singletons and the best modularity is
 * kept; nothing is random here — a start varies the seed of the hash that splits the vertices into sub-round classes,
 * derived from (seed, start); start 0 with seed 0 is the plain run.  The call synchronises the stream once per level of the hierarchy
 * (the convergence of a level is decided on the device; the host enqueues one iteration ahead).
 * The matrix must be symmetric (an undirected graph's adjacency matrix is; the reference reads only its strict lower
 * triangle and mirrors it, which is the same thing then).  Edge weights must be finite and in [0, 2^20].  GFICF_ERR_UNSUPPORTED only if one hash class of a vertex's neighbouring
 * communities overflows the 8192-slot table (vertices of any degree are handled in several passes; not observed). */
/* The optimiser's other quality function (RunModularityClustering(modularity = 2), src/RModularityOptimizer.cpp:36,100,
 * src/ModularityOptimizer.cpp:799-805): every node weighs 1 instead of its degree and the resolution is not divided by 2W,
 *   Q = (1/2W) * [ sum_ij A_ij delta(c_i, c_j) - resolution * sum_c n_c^2 ],   n_c = nodes in c;   resolution <= 1.
 * clustcells() always passes 1, which is the default of a new context. */
int gficf_ctx_set_louvain_options(gficf_ctx* ctx, int modularity_function);
size_t gficf_louvain_workspace_bytes(int64_t N, int64_t nnz);
/* ABI 7.  The starts are independent problems on one graph: gficf_louvain_device runs as many of them TOGETHER (one launch set over the
 * disjoint union of the copies, at most 16) as the workspace it is given holds — with gficf_louvain_workspace_bytes() bytes one at a time, with
 * gficf_louvain_workspace_bytes_starts(N, nnz, n_start) bytes all of min(n_start, 16) (about 24 B per matrix entry and 150 B per vertex for
 * every start run together).  The result does not depend on how many ran together. */
size_t gficf_louvain_workspace_bytes_starts(int64_t N, int64_t nnz, int n_start);
int gficf_louvain_device(gficf_ctx* ctx, int64_t N, const int64_t* d_indptr, const int32_t* d_indices,
                         const double* d_x, int64_t nnz, double resolution, int algorithm, int n_start,
                         int n_iter, int seed, int32_t* d_labels, int64_t* n_clusters, double* modularity, void* d_ws,
                         size_t ws_bytes);
int gficf_louvain_host(gficf_ctx* ctx, int64_t N, const void* indptr, int indptr_is_i64,
                       const int32_t* indices, const double* x, double resolution, int algorithm,
                       int n_start, int n_iter, int seed, int32_t* labels, int64_t* n_clusters, double* modularity);

/* ------------------------------------------------------------------- clustcells() in one call
 * The graph build and the community detection of clustcells() (R/clustCells.R:57-86) chained on the device — search
 * (k + 1 nearest, the cell itself dropped), Jaccard edges, weight > 0 filter, adjacency matrix, Louvain — with one upload
 * of the points and one download of the labels.  X: N x d column-major doubles (data$pca$cells); metric as in
 * gficf_knn_host; resolution .. seed as in gficf_louvain_host.  labels: int32[N], 0-based, clusters by decreasing size;
 * n_edges (optional): the number of edges kept by the filter.  Every stage keeps its own contract. */
int gficf_phenograph_host(gficf_ctx* ctx, const double* X, int64_t N, int d, int64_t ld, int k, int metric,
                          double resolution, int algorithm, int n_start, int n_iter, int seed,
                          int32_t* labels, int64_t* n_clusters, double* modularity, int64_t* n_edges);

/* ------------------------------------------------------------------- exact kNN search
 * "Next" row N2: the caller's step in front of the Jaccard build,
 *   neigh = uwot:::find_nn(data$pca$cells, k = k+1, include_self = T, method = "annoy",
 *                          metric = dist.method)$idx                    (R/clustCells.R:57,60)
 * uwot / Annoy are third-party and approximate; this is an EXACT search in f32 (Annoy's own
 * precision): for every query the k smallest (distance, index) pairs over ALL N points, the
 * query itself included (include_self = T), ties broken by the smaller index.  Distances:
 * manhattan = sum |a-b| (the reference's default, R/clustCells.R:46), euclidean = sqrt(sum (a-b)^2),
 * cosine = 1 - cos (rows L2-normalised first), correlation = cosine of the mean-centred rows (uwot's "correlation");
 * all accumulated in dimension order.
 */
typedef enum gficf_knn_metric {
  GFICF_KNN_MANHATTAN = 0,
  GFICF_KNN_EUCLIDEAN = 1,
  GFICF_KNN_COSINE = 2,
  GFICF_KNN_CORRELATION = 3    /* 1 - Pearson correlation = cosine distance of the rows with their mean removed */
} gficf_knn_metric;

#define GFICF_KNN_MAX_K 128

/* Row pitch (floats) of the point table for d dimensions (d rounded up to 4), -1 if d > 128. */
int gficf_knn_dpad(int d);

/* Device pipeline, split where a multi-GPU caller needs the seam (queries shard by cell block):
 *   1. prepare : a block of rows of the R matrix (n_rows x d column-major, f64 or f32, ld >= n_rows)
 *                -> row-major f32 point rows (n_rows x dpad, zero padded; cosine: normalised);
 *                non-finite coordinates are reported by gficf_ctx_sync() (GFICF_ERR_BAD_VALUE)
 *   2. (multi-GPU only) the caller all-gathers the point rows of all blocks
 *   3. search  : queries [q_begin, q_end) against all N points -> d_idx (1-based ids, (q_end-q_begin) x k
 *                column-major with leading dimension ld_out: column 0 is the nearest — the query
 *                itself unless an identical point has a smaller index), d_dist optional, same layout.
 *                d_idx + ld_out (k-1 columns) is exactly what gficf_jaccard_ingest_device reads.
 * d_ws: scratch of gficf_knn_workspace_bytes(ctx, q_end-q_begin, N, k) bytes. */
int gficf_knn_prepare_device(gficf_ctx* ctx, const void* d_X, int x_is_f64, int64_t n_rows, int d,
                             int64_t ld, int metric, float* d_point_rows);
size_t gficf_knn_workspace_bytes(gficf_ctx* ctx, int64_t n_queries, int64_t N, int k);
int gficf_knn_search_device(gficf_ctx* ctx, const float* d_points, int64_t N, int d, int k, int metric,
                            int64_t q_begin, int64_t q_end, void* d_ws, size_t ws_bytes,
                            int32_t* d_idx, float* d_dist, int64_t ld_out);

/* The cell order of the pruned search, for callers that want neighbour ids with LOCALITY (the halo form of the sharded Jaccard
 * build, R/clustCells.R:57-65: find_nn -> neigh[,-1] -> rcpp_parallel_jaccard_coef): d_order[p] = 0-based row of the point at
 * position p when the points are sorted by their (coarse, fine) pivot.  A function of the prepared points alone (every rank of
 * a sharded job computes the same order from the all-gathered points); a search over the points re-laid in this order returns
 * ids in which a contiguous block of cells names few rows outside itself.  d_ws: gficf_knn_workspace_bytes(ctx, N, N, 1). */
int gficf_knn_pivot_order_device(gficf_ctx* ctx, const float* d_points, int64_t N, int d, int metric, void* d_ws, size_t ws_bytes,
                                 int32_t* d_order);
/* Host form: X = N x d column-major doubles (an R numeric matrix); idx: N x k column-major int32
 * (1-based), dist: N x k column-major doubles or NULL. */
int gficf_knn_host(gficf_ctx* ctx, const double* X, int64_t N, int d, int64_t ld, int k, int metric,
                   int32_t* idx, double* dist);

#ifdef __cplusplus
}
#endif
#endif /* GFICF_HIP_H */
