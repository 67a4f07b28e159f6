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

// ------------------------------------------------------------------ pass A: nt_g counts
// nt_g = #{cells c : x[g,c] != 0}  (explicitly stored zeros do not count, as in
// rowSums(M != 0)).  Per-workgroup histogram in LDS (G counters), written out as one row of a
// workgroups x G table of partial counts that k_nt_sum adds up (round 1 flushed with one global
// atomic per touched gene: 5.9 M memory-side atomics = 21 of the pass's 69 us at config 3);
// falls back to global atomics per entry when G does not fit LDS.
// Each workgroup sweeps one contiguous slab; a thread takes 4 consecutive entries per load
// (16 B of rowidx, 2 x 16 B of x) and keeps two such groups in flight.
constexpr int CNT_THREADS = 1024;
constexpr int CNT_LDS_MAX_G = 36 * 1024;   // 144 KiB of uint32 counters

template <bool USE_LDS>
__device__ inline void count_one(int32_t g, double v, int64_t G, uint32_t* hist, unsigned long long* nt, bool& bad) {   // v: 1.0 when x is not read
  if (g < 0 || g >= G) { bad = true; return; }
  if (v != 0.0) {
    if (USE_LDS) atomicAdd(&hist[g], 1u);
    else atomicAdd(&nt[g], 1ull);
  }
}

// HAS_X == false counts every stored entry (4 B/nnz): exact whenever the matrix stores no explicit
// zeros, which the scaling pass verifies for free (it reads x anyway) — see gficf_csc_device.
template <bool USE_LDS, bool VEC, bool HAS_X>
__global__ __launch_bounds__(CNT_THREADS) void k_gene_count(const int32_t* __restrict__ rowidx,
                                                            const double* __restrict__ x, int64_t nnz, int64_t G,
                                                            unsigned long long* __restrict__ nt,
                                                            uint32_t* __restrict__ part, uint32_t* __restrict__ status) {
  extern __shared__ uint32_t s_hist[];
  const int64_t Gp = (G + 3) & ~(int64_t)3;                  // row pitch of the partial table (16 B rows)
  if (USE_LDS) {
    for (int64_t g = threadIdx.x; g < Gp; g += CNT_THREADS) s_hist[g] = 0;
    __syncthreads();
  }
  bool bad = false;
  constexpr int GROUPS = 4;                                  // 16 B rowidx + 32 B x per group, all in flight (12 groups without x: no faster)
  constexpr int64_t STRIDE = (int64_t)CNT_THREADS * 4;       // entries per group sweep of the workgroup
  constexpr int64_t CHUNK = STRIDE * GROUPS;
  const int64_t per_block = gficf_ceil_div(gficf_ceil_div(nnz, (int64_t)gridDim.x), CHUNK) * CHUNK;
  const int64_t p0 = (int64_t)blockIdx.x * per_block;
  const int64_t p1 = p0 + per_block < nnz ? p0 + per_block : nnz;
  int64_t p = p0;
  if (VEC) {
    for (; p + CHUNK <= p1; p += CHUNK) {
      const int64_t q = p + (int64_t)threadIdx.x * 4;
      typedef int v4i __attribute__((ext_vector_type(4)));
      typedef double v2d __attribute__((ext_vector_type(2)));
      v4i g[GROUPS];
      v2d xa[GROUPS], xb[GROUPS];
#pragma unroll
      for (int t = 0; t < GROUPS; ++t) {                     // streamed once: non-temporal
        g[t] = *reinterpret_cast<const v4i*>(rowidx + q + t * STRIDE);      // kept in the Infinity Cache for the kept-count pass
        xa[t] = HAS_X ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(x + q + t * STRIDE)) : v2d{1.0, 1.0};
        xb[t] = HAS_X ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(x + q + t * STRIDE + 2)) : v2d{1.0, 1.0};
      }
#pragma unroll
      for (int t = 0; t < GROUPS; ++t) {
        count_one<USE_LDS>(g[t].x, xa[t].x, G, s_hist, nt, bad);
        count_one<USE_LDS>(g[t].y, xa[t].y, G, s_hist, nt, bad);
        count_one<USE_LDS>(g[t].z, xb[t].x, G, s_hist, nt, bad);
        count_one<USE_LDS>(g[t].w, xb[t].y, G, s_hist, nt, bad);
      }
    }
  }
  for (p += threadIdx.x; p < p1; p += CNT_THREADS) count_one<USE_LDS>(rowidx[p], HAS_X ? x[p] : 1.0, G, s_hist, nt, bad);
  if (bad) atomicOr(status, GFICF_ST_BAD_CSC);
  if (USE_LDS) {
    __syncthreads();
    // this workgroup's row of the partial table, 16 B per lane, plain stores (k_nt_sum reads it next)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u* const dst = reinterpret_cast<v4u*>(part + (int64_t)blockIdx.x * Gp);
    const v4u* const src = reinterpret_cast<const v4u*>(s_hist);
    for (int64_t t = threadIdx.x; t < Gp / 4; t += CNT_THREADS) dst[t] = src[t];
  }
}

// nt[g] += sum over the partial rows.  A workgroup takes 64 genes: wave w adds rows w, w + 16, ... (coalesced 256 B runs,
// all loads of a thread independent), the 16 partial sums meet in LDS.  One writer per gene: no atomics.
constexpr int NS_WAVES = 16;

template <bool ADD>      // ADD: nt[g] += (the C ABI's count step accumulates into the caller's zeroed counters); else nt[g] =
__global__ __launch_bounds__(NS_WAVES * 64) void k_nt_sum(const uint32_t* __restrict__ part, int64_t Gp, int rows, int64_t G,
                                                          unsigned long long* __restrict__ nt) {
  __shared__ uint32_t s_acc[NS_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g = (int64_t)blockIdx.x * 64 + lane;
  uint32_t acc = 0;
  if (g < G) {
#pragma unroll 8
    for (int r = wave; r < rows; r += NS_WAVES) acc += part[(int64_t)r * Gp + g];
  }
  s_acc[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && g < G) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < NS_WAVES; ++w) t += s_acc[w][lane];
    if (ADD) { if (t) nt[g] += t; }
    else nt[g] = t;
  }
}

// Layout of the opaque per-gene buffer: G records {w, remap} | G doubles (weights of kept genes,
// indexed by new row id) | G uint16 (new row id, 0xFFFF = dropped).
__host__ __device__ inline double* genes_wkept(gficf_gene_entry* genes, int64_t G) { return reinterpret_cast<double*>(genes + G); }
__host__ __device__ inline const double* genes_wkept(const gficf_gene_entry* genes, int64_t G) { return reinterpret_cast<const double*>(genes + G); }
__host__ __device__ inline uint16_t* genes_remap16(gficf_gene_entry* genes, int64_t G) { return reinterpret_cast<uint16_t*>(genes_wkept(genes, G) + G); }
__host__ __device__ inline const uint16_t* genes_remap16(const gficf_gene_entry* genes, int64_t G) { return reinterpret_cast<const uint16_t*>(genes_wkept(genes, G) + G); }

// --------------------------------------------------- gene table: keep / remap / weights
// keep_g = nt_g > N*min && nt_g <= N*max (double compare, R/gficf.R:41); remap = exclusive
// scan of keep (new row id of a kept gene); w_g = log((N+1)/(nt_g+1)) (R/gficf.R:89) or the
// supplied weight.  One workgroup per 1024 genes; a workgroup obtains the number of kept
// genes in front of its tile by counting over nt[0 .. tile) itself (G is a few 10^4, the
// counters sit in L2), so there is no cross-workgroup dependency.
constexpr int GT_THREADS = 1024;

__device__ inline int block_sum_i32(int v, int* s_red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < GT_THREADS / 64; ++w) t += s_red[w];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(GT_THREADS) void k_gene_table(int64_t G, int64_t N_total, const int64_t* __restrict__ nt,
                                                           double prop_min, double prop_max,
                                                           const double* __restrict__ w_in, uint8_t* __restrict__ keep,
                                                           gficf_gene_entry* __restrict__ genes, double* __restrict__ w,
                                                           int64_t* __restrict__ gkept, int icf_type) {
  __shared__ int s_red[GT_THREADS / 64];
  __shared__ int s_wave_excl[GT_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double lo = (double)N_total * prop_min, hi = (double)N_total * prop_max;
  const int64_t tile0 = (int64_t)blockIdx.x * GT_THREADS;
  // kept genes in front of this tile
  int before = 0;
#pragma unroll 8
  for (int64_t g = tid; g < tile0; g += GT_THREADS) {          // independent loads: all in flight (was one at a time, 10 us)
    const double c = (double)nt[g];
    before += (c > lo && c <= hi) ? 1 : 0;
  }
  before = block_sum_i32(before, s_red);
  const int64_t g = tile0 + tid;
  double c = 0.0;
  bool kp = false;
  if (g < G) {
    c = (double)nt[g];
    kp = c > lo && c <= hi;
  }
  const unsigned long long m = __ballot(kp);
  if (lane == 0) s_red[wave] = __popcll(m);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int wv = 0; wv < GT_THREADS / 64; ++wv) { s_wave_excl[wv] = run; run += s_red[wv]; }
    if (tile0 + GT_THREADS >= G) *gkept = (int64_t)before + run;     // last tile publishes the total
  }
  __syncthreads();
  if (g < G) {
    const int r = before + s_wave_excl[wave] + __popcll(m & ((1ull << lane) - 1ull));
    double wv = 0.0;
    if (kp) {
      if (w_in) wv = w_in[g];
      else if (icf_type == 1) wv = log(((double)N_total - c) / c);               // "prob"    R/gficf.R:90
      else if (icf_type == 2) wv = log(1.0 + (double)N_total / c);               // "smooth"  R/gficf.R:91
      else wv = log(((double)N_total + 1.0) / (c + 1.0));                        // "classic" R/gficf.R:89
    }
    keep[g] = kp ? 1 : 0;
    w[g] = wv;
    gficf_gene_entry e;
    e.w = wv;
    e.remap = kp ? r : -1;
    e.reserved = 0;
    genes[g] = e;
    // compact tables for the LDS-resident scaling variant
    if (kp) genes_wkept(genes, G)[r] = wv;
    genes_remap16(genes, G)[g] = (kp && r < 0xFFFF) ? (uint16_t)r : (uint16_t)0xFFFF;
  }
}

// Row sum and gene table in one launch (the fused sequence, gficf_csc_device): the workgroup that has summed a tile of
// 64 genes also knows how many of them are kept; the new row ids need the kept genes in front of the tile, which come from a
// look-back over the earlier tiles' counts (gficf_lookback_exclusive; tiles taken in ticket order) instead of a second
// launch that counts them again.  Writes nt, keep, w, the gene records and the compact tables, and the number of kept genes.
__global__ __launch_bounds__(NS_WAVES * 64) void k_nt_sum_table(const uint32_t* __restrict__ part, int64_t Gp, int rows, int64_t G,
                                                                int64_t N_total, double prop_min, double prop_max,
                                                                const double* __restrict__ w_in, unsigned long long* __restrict__ nt,
                                                                uint8_t* __restrict__ keep, gficf_gene_entry* __restrict__ genes,
                                                                double* __restrict__ w, int64_t* __restrict__ gkept, int icf_type,
                                                                unsigned long long* ws, uint32_t epoch) {
  __shared__ uint32_t s_acc[NS_WAVES][64];
  __shared__ unsigned long long s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(&ws[0], 1ull);
  __syncthreads();
  const int64_t tile = (int64_t)s_tile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t g = tile * 64 + lane;
  uint32_t acc = 0;
  if (g < G) {
#pragma unroll 8
    for (int r = wave; r < rows; r += NS_WAVES) acc += part[(int64_t)r * Gp + g];
  }
  s_acc[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  uint32_t t = 0;
#pragma unroll
  for (int wv = 0; wv < NS_WAVES; ++wv) t += s_acc[wv][lane];
  const double c = (double)t;
  const bool kp = g < G && c > (double)N_total * prop_min && c <= (double)N_total * prop_max;   // R/gficf.R:41, comparison in double
  const unsigned long long m = __ballot(kp);
  const int64_t before = gficf_lookback_exclusive(ws, tile, (int64_t)gridDim.x, epoch, (int64_t)__popcll(m));
  if (lane == 0 && tile == (int64_t)gridDim.x - 1) *gkept = before + __popcll(m);
  if (g < G) {
    const int64_t r = before + __popcll(m & ((1ull << lane) - 1ull));
    double wv = 0.0;
    if (kp) {
      if (w_in) wv = w_in[g];
      else if (icf_type == 1) wv = log(((double)N_total - c) / c);               // "prob"    R/gficf.R:90
      else if (icf_type == 2) wv = log(1.0 + (double)N_total / c);               // "smooth"  R/gficf.R:91
      else wv = log(((double)N_total + 1.0) / (c + 1.0));                        // "classic" R/gficf.R:89
    }
    nt[g] = t;
    keep[g] = kp ? 1 : 0;
    w[g] = wv;
    gficf_gene_entry e;
    e.w = wv;
    e.remap = kp ? (int32_t)r : -1;
    e.reserved = 0;
    genes[g] = e;
    if (kp) genes_wkept(genes, G)[r] = wv;
    genes_remap16(genes, G)[g] = (kp && r < 0xFFFF) ? (uint16_t)r : (uint16_t)0xFFFF;
  }
}

// The contiguous range of cells that holds share number `share` (of gridDim.x) of the stored entries: range[0] = the smallest
// cell c with colptr[c] >= nnz * share / shares, range[1] the same for share + 1 (the last share ends with the last cell).  Called
// by the first wave of the workgroup: a 32-way search, lanes 0..31 for the start, 32..63 for the end (3-4 dependent loads).
constexpr int64_t SMALL_CELLS = 16384;           // below: cells are split evenly by number
__device__ inline void cell_range_by_entries(const int64_t* __restrict__ colptr, int64_t n_cells, int64_t* range, int64_t share) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
  const int64_t nb = (int64_t)gridDim.x;
  if (n_cells < SMALL_CELLS) {                   // small inputs: the search's dependent loads cost more than balance gains
    if (l == 0) range[half] = n_cells / nb * (share + half) + (n_cells % nb) * (share + half) / nb;
    return;
  }
  const int64_t b = share + half, nnz_all = colptr[n_cells];
  const int64_t target = nnz_all / nb * b + (nnz_all % nb) * b / nb;
  int64_t lo = 0, hi = n_cells;                  // the answer lies in [lo, hi]; colptr[hi] >= target throughout
  if (b >= nb) lo = hi;
  while (__any(hi > lo)) {
    const bool active = hi > lo;
    const int64_t step = active ? (hi - lo + 31) / 32 : 1;
    int64_t p = lo + step * l;
    if (p > hi) p = hi;
    const bool ge = active ? colptr[p] >= target : true;
    const unsigned int m = (unsigned int)(__ballot(ge) >> (half * 32));
    if (active) {
      const int f = m ? __builtin_ctz(m) : 32;   // first probe at or past the target
      if (f == 0) hi = lo;
      else {
        int64_t below = lo + step * (f - 1), at = hi;
        if (f < 32) { at = lo + step * f; if (at > hi) at = hi; }
        lo = below + 1 < at ? below + 1 : at;
        hi = at;
      }
    }
  }
  if (l == 0) range[half] = lo;
}

// ------------------------------------------------------- pass B0: kept entries per cell
// One wave per cell; out[c] = #{entries of cell c whose gene is kept}; out[n_cells] = 0,
// turned into the new colptr by an exclusive scan.  The keep mask is staged as a bitmask in
// LDS (G bits).  When no gene is dropped the count is the old column length and rowidx is
// not read at all.
constexpr int CC_THREADS = 256;

__global__ __launch_bounds__(CC_THREADS) void k_cell_kept_count(int64_t G, int64_t n_cells,
                                                                const int64_t* __restrict__ colptr,
                                                                const int32_t* __restrict__ rowidx,
                                                                const uint8_t* __restrict__ keep,
                                                                const int64_t* __restrict__ gkept,
                                                                int64_t* __restrict__ out, uint32_t* __restrict__ status) {
  extern __shared__ uint32_t s_bits[];      // ceil(G/32) words
  __shared__ int64_t s_range[2];
  __shared__ unsigned int s_next;
  const int lane = threadIdx.x & 63;
  const bool all_kept = (*gkept == G);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n_cells] = 0;
  // the workgroup's cells: its share of the stored entries (cell_range_by_entries), dealt to its waves one by one
  if (threadIdx.x < 64) {
    cell_range_by_entries(colptr, n_cells, s_range, (int64_t)gridDim.x - 1 - (int64_t)blockIdx.x);   // first workgroups: last cells
    if (threadIdx.x == 0) s_next = 0u;
  }
  if (all_kept) __syncthreads();
  if (!all_kept) {
    const int64_t words = (G + 31) / 32;
    const bool aligned4 = ((uintptr_t)keep & 3u) == 0;
    for (int64_t wd = threadIdx.x; wd < words; wd += CC_THREADS) {
      uint32_t bits = 0;
      const int64_t g0 = wd * 32;
      if (aligned4 && g0 + 32 <= G) {            // 32 keep bytes (0/1) -> one word
        const uint32_t* k4 = reinterpret_cast<const uint32_t*>(keep + g0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const uint32_t v = k4[t];
          bits |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * t);
        }
      } else {
        for (int b = 0; b < 32; ++b)
          if (g0 + b < G && keep[g0 + b]) bits |= 1u << b;
      }
      s_bits[wd] = bits;
    }
    __syncthreads();
  }
  // Cells are swept from the last to the first: pass A has just streamed rowidx front to back, so its
  // tail is what the 256 MiB Infinity Cache still holds; reading backwards re-uses it before it ages out.
  // (the workgroups take the ranges from the last to the first, and each walks its own backwards)
  const int64_t cell_lo = s_range[0], cell_hi = s_range[1];
  auto grab = [&]() -> int64_t {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&s_next, 1u);
    return cell_hi - 1 - (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
  };
  int64_t c_next = grab();
  for (int64_t c = c_next; c >= cell_lo; c = c_next) {
    c_next = grab();
    const int64_t p0 = colptr[c], p1 = colptr[c + 1];
    if (p1 < p0) { if (lane == 0) { atomicOr(status, GFICF_ST_BAD_CSC); out[c] = 0; } continue; }
    int64_t cnt;
    if (all_kept) {
      cnt = p1 - p0;
    } else {
      int n = 0;
      int64_t p = p0 + lane;
      for (; p + 7 * 64 < p1; p += 8 * 64) {        // 8 independent loads in flight per lane
        int32_t g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = rowidx[p + t * 64];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (g[t] >= 0 && g[t] < G) n += (s_bits[g[t] >> 5] >> (g[t] & 31)) & 1u;
      }
      if (p < p1) {                                 // tail of up to 8 x 64 entries: all loads in flight as well
        int32_t g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = p + t * 64 < p1 ? rowidx[p + t * 64] : -1;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (g[t] >= 0 && g[t] < G) n += (s_bits[g[t] >> 5] >> (g[t] & 31)) & 1u;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
      cnt = n;
    }
    if (lane == 0) out[c] = cnt;
  }
}

constexpr size_t SL_LDS_BYTES = 156 * 1024;

__host__ __device__ inline size_t sl_lds_need(int64_t G, int64_t gkept) {
  return (((size_t)G * 2 + 15) & ~(size_t)15) + (size_t)gkept * 8;
}
__host__ __device__ inline bool sl_fits(int64_t G, int64_t gkept) {
  return gkept < 0xFFFF && sl_lds_need(G, gkept) <= SL_LDS_BYTES;
}

// ---------------------------------------------------------------- pass B: scale a cell
// One workgroup of SC_WAVES waves per cell.  Wave w owns a contiguous run of the cell's
// entries, so kept entries keep their order and every wave's output run is contiguous.
// A cell of up to SC_WAVES*64*SC_CH entries is read from HBM exactly once: every thread keeps
// its entries (x, weight, new row id) in registers across the two workgroup reductions
//   S_c = sum of kept x                      (R/gficf.R:59)
//   q_c = sum ((x / S_c) * w_g)^2            (R/gficf.R:59,79,100)
// and then writes  (1/sqrt(q_c), Inf -> 0) * ((x / S_c) * w_g)   (R/gficf.R:100-103) compacted
// and renumbered.  Longer cells take three sweeps (the re-reads hit L2).
constexpr int SC_WAVES = 4;
constexpr int SC_THREADS = SC_WAVES * 64;
constexpr int SC_CH = 8;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__global__ __launch_bounds__(SC_THREADS) void k_scale_cells(int64_t G, int64_t n_cells,
                                                            const int64_t* __restrict__ colptr,
                                                            const int32_t* __restrict__ rowidx,
                                                            const double* __restrict__ x,
                                                            const gficf_gene_entry* __restrict__ genes,
                                                            const int64_t* __restrict__ gkept_p,
                                                            const int64_t* __restrict__ out_colptr,
                                                            int32_t* __restrict__ out_rowidx,
                                                            double* __restrict__ out_x, int norm_l1,
                                                            uint32_t* zero_flag, int64_t* __restrict__ out_end) {
  __shared__ double s_sum[SC_WAVES];
  if (gkept_p && *gkept_p < 0xFFFF) return;       // the LDS-resident variant handles this input (launched whenever the row ids fit LDS)
  bool saw_zero = false;                          // an explicitly stored zero (see gficf_csc_device)
  __shared__ int32_t s_cnt[SC_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int4* const gtab = reinterpret_cast<const int4*>(genes);
  for (int64_t c = blockIdx.x; c < n_cells; c += gridDim.x) {
    const int64_t p0 = colptr[c], p1 = colptr[c + 1];
    const int64_t len = p1 - p0;
    if (len <= 0) {                                 // uniform over the workgroup
      if (out_end != nullptr && threadIdx.x == 0) out_end[c] = out_colptr[c];
      continue;
    }
    const int64_t seg = gficf_ceil_div(gficf_ceil_div(len, SC_WAVES), 64) * 64;
    const int64_t a0 = p0 + (int64_t)wave * seg < p1 ? p0 + (int64_t)wave * seg : p1;
    const int64_t a1 = a0 + seg < p1 ? a0 + seg : p1;
    const bool cached = seg <= 64 * SC_CH;          // uniform over the workgroup
    double S = 0.0;
    int kept = 0;
    double xv[SC_CH], wv[SC_CH];
    int32_t rv[SC_CH];
    if (cached) {
      int32_t gv[SC_CH];
#pragma unroll
      for (int m = 0; m < SC_CH; ++m) {
        const int64_t p = a0 + m * 64 + lane;
        gv[m] = -1;
        xv[m] = 0.0;
        if (p < a1) { gv[m] = rowidx[p]; xv[m] = x[p]; }
      }
      // all gene-record gathers are issued before any is consumed (ids outside [0, G) re-read
      // record 0 and are masked afterwards)
      int4 ge[SC_CH];
#pragma unroll
      for (int m = 0; m < SC_CH; ++m) ge[m] = gtab[(gv[m] >= 0 && gv[m] < G) ? gv[m] : 0];
#pragma unroll
      for (int m = 0; m < SC_CH; ++m) {
        const bool valid = gv[m] >= 0 && gv[m] < G;
        saw_zero |= valid && xv[m] == 0.0;
        rv[m] = valid ? ge[m].z : -1;
        wv[m] = __hiloint2double(ge[m].y, ge[m].x);
        if (rv[m] >= 0) { S += xv[m]; ++kept; }
      }
    } else {
      for (int64_t p = a0 + lane; p < a1; p += 64) {
        const int32_t g = rowidx[p];
        const double xp = x[p];
        saw_zero |= xp == 0.0;
        if (g >= 0 && g < G && gtab[g].z >= 0) { S += xp; ++kept; }
      }
    }
    S = wave_sum(S);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kept += __shfl_xor(kept, d);
    if (lane == 0) { s_sum[wave] = S; s_cnt[wave] = kept; }
    __syncthreads();
    double Sc = 0.0;
    int64_t opos = out_colptr[c];
#pragma unroll
    for (int t = 0; t < SC_WAVES; ++t) {
      Sc += s_sum[t];
      if (t < wave) opos += s_cnt[t];
    }
    __syncthreads();
    double q = 0.0;
    if (cached) {
#pragma unroll
      for (int m = 0; m < SC_CH; ++m) {
        double v = 0.0;
        if (rv[m] >= 0 && Sc != 0.0) v = (xv[m] / Sc) * wv[m];
        xv[m] = v;
        q += norm_l1 ? v : v * v;
      }
    } else if (Sc != 0.0) {
      for (int64_t p = a0 + lane; p < a1; p += 64) {
        const int32_t g = rowidx[p];
        if (g >= 0 && g < G) {
          const int4 e = gtab[g];
          if (e.z >= 0) { const double v = (x[p] / Sc) * __hiloint2double(e.y, e.x); q += norm_l1 ? v : v * v; }
        }
      }
    }
    q = wave_sum(q);
    if (lane == 0) s_sum[wave] = q;
    __syncthreads();
    double qc = 0.0;
#pragma unroll
    for (int t = 0; t < SC_WAVES; ++t) qc += s_sum[t];
    __syncthreads();
    double nv = 1.0 / (norm_l1 ? qc : sqrt(qc));      // l.norm: l1 = 1/rowSums(m), l2 = 1/sqrt(rowSums(m^2))  R/gficf.R:100
    if (isinf(nv)) nv = 0.0;                        // R/gficf.R:101
    if (cached) {
#pragma unroll
      for (int m = 0; m < SC_CH; ++m) {
        if (a0 + m * 64 < a1) {                     // uniform over the wave
          const bool kp = rv[m] >= 0;
          const unsigned long long mk = __ballot(kp);
          if (kp) {
            const int64_t dst = opos + __popcll(mk & lt_mask);
            out_rowidx[dst] = rv[m];
            out_x[dst] = nv * xv[m];
          }
          opos += __popcll(mk);
        }
      }
    } else {
      for (int64_t pb = a0; pb < a1; pb += 64) {
        const int64_t p = pb + lane;
        bool kp = false;
        int32_t r = -1;
        double v = 0.0;
        if (p < a1) {
          const int32_t g = rowidx[p];
          if (g >= 0 && g < G) {
            const int4 e = gtab[g];
            r = e.z;
            kp = r >= 0;
            if (kp && Sc != 0.0) v = nv * ((x[p] / Sc) * __hiloint2double(e.y, e.x));
          }
        }
        const unsigned long long mk = __ballot(kp);
        if (kp) {
          const int64_t dst = opos + __popcll(mk & lt_mask);
          out_rowidx[dst] = r;
          out_x[dst] = v;
        }
        opos += __popcll(mk);
      }
    }
    // pointerB / pointerE form: the position behind the cell's last kept entry (the last wave's running position)
    if (out_end != nullptr && wave == SC_WAVES - 1 && lane == 0) out_end[c] = opos;
  }
  if (zero_flag != nullptr && saw_zero) atomicOr(zero_flag, GFICF_ST_EXPLICIT_ZERO);
}

// ------------------------------------------------ pass B, LDS-resident gene tables
// When the 16-bit row ids of all genes (2 G bytes) and the weights of the kept genes (8 G_kept
// bytes) fit the CU's LDS, one persistent workgroup per CU stages both once and then every wave
// scales whole cells on its own: no barriers after the staging, and the per-entry lookups are LDS
// reads instead of L2 requests (the global-gather variant above issues one L2 request per
// entry).  A lane keeps x and the new row id of its entries in registers (3 VGPRs per entry), the
// weight is read from LDS when it is needed.  Same arithmetic and order of operations.
// 768 threads and 32 register chunks (2048 entries per wave without re-reading; config 5 caps a cell at 2147).  With the cells
// dealt round-robin 896 / 28 was the best of 1024 / 24, 768 / 36, 640 / 40, 512 / 48 (profiles/r02_gficf_scale_ab.txt: the whole
// pass 0.451 -> 0.427 ms); with the cells dealt by entries (below) 768 / 32 is ahead of 896 / 28, 832 / 30, 768 / 28, 768 / 36, 704 / 34,
// 640 / 36 by 1-4 % (runs in separate processes, each variant in both modes of the process-to-process spread).
#ifndef GFICF_SL_THREADS
#define GFICF_SL_THREADS 768
#endif
#ifndef GFICF_SL_CH
#define GFICF_SL_CH 32
#endif
constexpr int SL_THREADS = GFICF_SL_THREADS;     // (A/B of these two: profiles/r02_gficf_scale_ab.txt)
constexpr int SL_CH = GFICF_SL_CH;               // chunks of 64 entries a wave keeps in registers
constexpr int SL_LB = 8;                          // chunks per batch on the long-cell path
// W_LDS: the weights of the kept genes are staged in LDS too (they fit next to the row ids); otherwise they are read
// from the gene table in global memory (8 G_kept bytes, L2-resident) — same kernel, same launch: the host cannot know
// G_kept without a sync, and a second kernel that returns at once still costs 5 us plus a launch gap.
template <bool W_LDS>
__device__ inline void sl_body(int64_t G, int64_t n_cells, const int64_t* __restrict__ colptr, const int32_t* __restrict__ rowidx,
                               const double* __restrict__ x, const gficf_gene_entry* __restrict__ genes, int64_t gkept,
                               const int64_t* __restrict__ out_colptr, int32_t* __restrict__ out_rowidx,
                               double* __restrict__ out_x, int norm_l1, uint32_t* zero_flag, int static_cells,
                               int64_t* __restrict__ out_end) {
  extern __shared__ unsigned char s_raw[];
  bool saw_zero = false;                          // an explicitly stored zero (see gficf_csc_device)
  uint16_t* const s_remap = reinterpret_cast<uint16_t*>(s_raw);
  double* const s_w = reinterpret_cast<double*>(s_raw + (((size_t)G * 2 + 15) & ~(size_t)15));
  const double* const g_w = genes_wkept(genes, G);
  auto weight = [&](int32_t r) -> double { return W_LDS ? s_w[r] : g_w[r]; };
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(genes_remap16(genes, G));
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_remap);
    for (int64_t t = threadIdx.x; t < (G + 1) / 2; t += SL_THREADS) dst[t] = src[t];
    if (W_LDS)
      for (int64_t t = threadIdx.x; t < gkept; t += SL_THREADS) s_w[t] = g_w[t];
  }
  // Cells: the workgroup owns the contiguous range of cells that holds its share of the stored ENTRIES (cells differ in
  // length by a factor of several: a static deal of cells to waves leaves the last waves working alone for ~15 of the
  // kernel's 270 us), and its waves take that range's cells one by one from a counter in LDS.
  __shared__ int64_t s_range[2];
  __shared__ unsigned int s_next;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    cell_range_by_entries(colptr, n_cells, s_range, (int64_t)blockIdx.x);
    if (threadIdx.x == 0) s_next = 0u;
  }
  __syncthreads();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t cell_lo = static_cells ? 0 : s_range[0], cell_hi = static_cells ? n_cells : s_range[1];
  int64_t static_c = ((int64_t)blockIdx.x * SL_THREADS + threadIdx.x) >> 6;     // test hook: the round-robin deal of cells to waves
  const int64_t nwaves = ((int64_t)gridDim.x * SL_THREADS) >> 6;
  auto grab = [&]() -> int64_t {                   // next cell of the range (one LDS atomic per wave and cell)
    if (static_cells) { const int64_t c = static_c; static_c += nwaves; return c; }
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&s_next, 1u);
    return cell_lo + (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
  };
  int64_t c_next = grab();
  for (int64_t c = c_next; c < cell_hi; c = c_next) {
    c_next = grab();                               // asked for early: the round trip hides behind this cell's loads
    const int64_t p0 = colptr[c], p1 = colptr[c + 1];
    const int64_t len = p1 - p0;
    if (len <= 0) {                               // uniform over the wave
      if (out_end != nullptr && lane == 0) out_end[c] = out_colptr[c];
      continue;
    }
    int64_t opos = out_colptr[c];
    // A wave keeps the first 64 * SL_CH entries of its cell (the "head") in registers and reads them from HBM exactly once.
    // Entries beyond that (the "tail": cells of more than 2048 stored entries — a third of the cells at SURVEY.md 8d's
    // density, common in real droplet data) are swept first, in batches of SL_LB chunks, for their part of the two sums
    // only, and read a second time (from L2: they were just read) when they are written, behind the head.  For the tail's
    // part of the norm the two reductions run in ONE sweep: sum_tail ((x / S) w)^2 is taken as (sum_tail (x w)^2) / S^2
    // — the same number up to rounding (a few 1e-16 relative, the contract is 1e-6) — because S is only known once the head
    // is in; the head's part keeps the reference's order of operations.  (Rounds 1-2 swept a long cell three times.)
    const int64_t head_len = len < 64 * SL_CH ? len : 64 * SL_CH;
    const int64_t t0 = p0 + head_len;             // first tail entry
    double S = 0.0, Qt = 0.0;
    if (t0 < p1) {
      for (int64_t base = t0; base < p1; base += 64 * SL_LB) {
        int32_t gz[SL_LB];
        double xb[SL_LB];
#pragma unroll
        for (int m = 0; m < SL_LB; ++m) {
          const int64_t p = base + m * 64 + lane;
          gz[m] = -1; xb[m] = 0.0;
          if (p < p1) { gz[m] = rowidx[p]; xb[m] = x[p]; }
        }
#pragma unroll
        for (int m = 0; m < SL_LB; ++m) {
          const uint32_t g = (uint32_t)gz[m];
          saw_zero |= g < (uint32_t)G && xb[m] == 0.0;
          const uint32_t r = g < (uint32_t)G ? (uint32_t)s_remap[g] : 0xFFFFu;
          if (r != 0xFFFFu) {
            S += xb[m];
            const double xw = xb[m] * weight((int32_t)r);
            Qt += norm_l1 ? xw : xw * xw;
          }
        }
      }
    }
    {
      const int n_it = (int)((head_len + 63) >> 6);
      const int64_t h1 = p0 + head_len;
      double xv[SL_CH];
      int32_t rv[SL_CH];
      // entries: all loads of the head are issued before any is consumed
#pragma unroll
      for (int m = 0; m < SL_CH; ++m) {
        rv[m] = -1;
        xv[m] = 0.0;
        if (m < n_it) {
          const int64_t p = p0 + m * 64 + lane;
          if (p < h1) { rv[m] = __builtin_nontemporal_load(rowidx + p); xv[m] = __builtin_nontemporal_load(x + p); }
        }
      }
#pragma unroll
      for (int m = 0; m < SL_CH; ++m) {
        if (m < n_it) {
          const uint32_t g = (uint32_t)rv[m];
          saw_zero |= g < (uint32_t)G && xv[m] == 0.0;
          const uint32_t r = g < (uint32_t)G ? (uint32_t)s_remap[g] : 0xFFFFu;
          rv[m] = r == 0xFFFFu ? -1 : (int32_t)r;
          if (rv[m] >= 0) S += xv[m];
        }
      }
      const double Sc = wave_sum(S);
      double q = 0.0;
#pragma unroll
      for (int m = 0; m < SL_CH; ++m) {
        if (m < n_it) {
          double v = 0.0;
          if (rv[m] >= 0 && Sc != 0.0) v = (xv[m] / Sc) * weight(rv[m]);
          xv[m] = v;
          q += norm_l1 ? v : v * v;
        }
      }
      double qc = wave_sum(q);
      if (t0 < p1 && Sc != 0.0) {                 // uniform over the wave: the tail's share of the norm
        const double qt = wave_sum(Qt);
        qc += norm_l1 ? qt / Sc : qt / (Sc * Sc);
      }
      double nv = 1.0 / (norm_l1 ? qc : sqrt(qc));
      if (isinf(nv)) nv = 0.0;                    // R/gficf.R:101
#pragma unroll
      for (int m = 0; m < SL_CH; ++m) {
        if (m < n_it) {
          const bool kp = rv[m] >= 0;
          const unsigned long long mk = __ballot(kp);
          if (kp) {
            const int64_t dst = opos + __popcll(mk & lt_mask);
            __builtin_nontemporal_store(rv[m], out_rowidx + dst);
            __builtin_nontemporal_store(nv * xv[m], out_x + dst);
          }
          opos += __popcll(mk);
        }
      }
      // the tail again (from L2), written behind the head
      for (int64_t base = t0; base < p1; base += 64 * SL_LB) {
        int32_t gz[SL_LB];
        double xb[SL_LB];
#pragma unroll
        for (int m = 0; m < SL_LB; ++m) {
          const int64_t p = base + m * 64 + lane;
          gz[m] = -1; xb[m] = 0.0;
          if (p < p1) { gz[m] = rowidx[p]; xb[m] = x[p]; }
        }
#pragma unroll
        for (int m = 0; m < SL_LB; ++m) {
          if (base + m * 64 < p1) {               // uniform over the wave
            const uint32_t g = (uint32_t)gz[m];
            const uint32_t r = g < (uint32_t)G ? (uint32_t)s_remap[g] : 0xFFFFu;
            const bool kp = r != 0xFFFFu;
            const unsigned long long mk = __ballot(kp);
            if (kp) {
              const int64_t dst = opos + __popcll(mk & lt_mask);
              const double v = Sc != 0.0 ? nv * ((xb[m] / Sc) * weight((int32_t)r)) : 0.0;
              __builtin_nontemporal_store((int32_t)r, out_rowidx + dst);
              __builtin_nontemporal_store(v, out_x + dst);
            }
            opos += __popcll(mk);
          }
        }
      }
      // pointerB / pointerE form: the position behind the cell's last kept entry
      if (out_end != nullptr && lane == 0) out_end[c] = opos;
    }
  }
  if (zero_flag != nullptr && saw_zero) atomicOr(zero_flag, GFICF_ST_EXPLICIT_ZERO);
}

// mode: 0 = by the data (weights in LDS when they fit), 1 = test hook: weights from global memory whatever their size
__global__ __launch_bounds__(SL_THREADS) void k_scale_cells_lds(int64_t G, int64_t n_cells,
                                                                const int64_t* __restrict__ colptr,
                                                                const int32_t* __restrict__ rowidx,
                                                                const double* __restrict__ x,
                                                                const gficf_gene_entry* __restrict__ genes,
                                                                const int64_t* __restrict__ gkept_p,
                                                                const int64_t* __restrict__ out_colptr,
                                                                int32_t* __restrict__ out_rowidx,
                                                                double* __restrict__ out_x, int norm_l1,
                                                                uint32_t* zero_flag, int static_cells, int mode,
                                                                int64_t* __restrict__ out_end) {
  const int64_t gkept = *gkept_p;
  if (gkept >= 0xFFFF) return;                    // new row ids do not fit 16 bits: the global-gather variant handles this input
  if (mode == 0 && sl_fits(G, gkept))
    sl_body<true>(G, n_cells, colptr, rowidx, x, genes, gkept, out_colptr, out_rowidx, out_x, norm_l1, zero_flag, static_cells, out_end);
  else
    sl_body<false>(G, n_cells, colptr, rowidx, x, genes, gkept, out_colptr, out_rowidx, out_x, norm_l1, zero_flag, static_cells, out_end);
}

__global__ __launch_bounds__(256) void k_zero_i64(int64_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ------------------------------------------------------- cluster signatures (next row N3)
// data$cluster.gene.rnk = sapply(unique(cluster), function(x) rowSums(gficf[, cluster %in% x]))
// (reference R/clustCells.R:121-123): out[g, c] = sum over the cells of cluster c of gficf[g, cell].
// One wave per cell; f64 atomic adds into the dense G x C result (column-major).  The order of the
// additions is not fixed, so the last bits can differ from run to run (well inside 1e-6).
// (columns are given as begin / end pointers: the canonical CSC hands in colptr and colptr + 1, the pointerB / pointerE form of
// gficf_csc_scale_be_device its two arrays)
__global__ __launch_bounds__(256) void k_cluster_signatures(int64_t G, int64_t n_cells, const int64_t* __restrict__ ptr_b, const int64_t* __restrict__ ptr_e,
                                                            const int32_t* __restrict__ rowidx, const double* __restrict__ x,
                                                            const int32_t* __restrict__ cluster, int32_t C,
                                                            double* __restrict__ out, uint32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t c = w0; c < n_cells; c += nw) {
    const int32_t cl = cluster[c];
    if (cl < 0 || cl >= C) { if (lane == 0) atomicOr(status, GFICF_ST_BAD_CSC); continue; }
    double* const col = out + (int64_t)cl * G;
    const int64_t p0 = ptr_b[c], p1 = ptr_e[c];
    for (int64_t p = p0 + lane; p < p1; p += 64) {
      const int32_t g = rowidx[p];
      if (g >= 0 && g < G) unsafeAtomicAdd(col + g, x[p]);      // the hardware f64 add (atomicAdd compiles to a compare-and-swap loop)
    }
  }
}

// The same sums with the additions kept on the CU: the cells are grouped by cluster first (counting sort of the cell ids),
// a workgroup takes a slice of ONE cluster's cells and adds their entries into G doubles of LDS (ds_add_f64), then adds
// its G partial sums to the result — a few hundred global atomics per gene instead of one per stored entry (the plain
// kernel above sits at the L2's rate for contended f64 atomics, 43 G/s).  Needs G doubles of LDS: G <= SIG_MAX_G.
constexpr int SIG_MAX_G = 18432;          // 144 KiB
constexpr int SIG_BINS = 4096;            // clusters binned in LDS while grouping the cells

__global__ __launch_bounds__(256) void k_sig_count(int64_t n_cells, const int32_t* __restrict__ cluster, int32_t C, int64_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ status) {
  __shared__ uint32_t s_n[SIG_BINS];
  const bool binned = C <= SIG_BINS;
  if (binned) {
    for (int t = threadIdx.x; t < C; t += 256) s_n[t] = 0u;
    __syncthreads();
  }
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n_cells; c += (int64_t)gridDim.x * 256) {
    const int32_t cl = cluster[c];
    if (cl < 0 || cl >= C) { atomicOr(status, GFICF_ST_BAD_CSC); continue; }
    if (binned) atomicAdd(&s_n[cl], 1u);
    else atomicAdd((unsigned long long*)&cnt[cl], 1ull);
  }
  if (binned) {
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += 256)
      if (s_n[t]) atomicAdd((unsigned long long*)&cnt[t], (unsigned long long)s_n[t]);
  }
}

// order[start[cl] ..] = the cells of cluster cl (in no particular order: only the order of the additions depends on it)
__global__ __launch_bounds__(256) void k_sig_fill(int64_t n_cells, const int32_t* __restrict__ cluster, int32_t C, const int64_t* __restrict__ start,
                                                  uint32_t* __restrict__ cursor, int32_t* __restrict__ order) {
  __shared__ uint32_t s_n[SIG_BINS], s_base[SIG_BINS];
  const bool binned = C <= SIG_BINS;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t cl = c < n_cells ? cluster[c] : -1;
  const bool ok = cl >= 0 && cl < C;
  if (!binned) {
    if (ok) order[start[cl] + atomicAdd(&cursor[cl], 1u)] = (int32_t)c;
    return;
  }
  for (int t = threadIdx.x; t < C; t += 256) s_n[t] = 0u;
  __syncthreads();
  uint32_t mine = 0;
  if (ok) mine = atomicAdd(&s_n[cl], 1u);
  __syncthreads();
  for (int t = threadIdx.x; t < C; t += 256)
    if (s_n[t]) s_base[t] = atomicAdd(&cursor[t], s_n[t]);
  __syncthreads();
  if (ok) order[start[cl] + s_base[cl] + mine] = (int32_t)c;
}

// grid (slices, C): workgroup (b, cl) sums slice b of cluster cl's cells
__global__ __launch_bounds__(256) void k_sig_sum(int64_t G, const int64_t* __restrict__ ptr_b, const int64_t* __restrict__ ptr_e, const int32_t* __restrict__ rowidx,
                                                 const double* __restrict__ x, const int64_t* __restrict__ start,
                                                 const int32_t* __restrict__ order, double* __restrict__ out) {
  extern __shared__ double s_acc[];
  const int cl = blockIdx.y;
  const int64_t lo0 = start[cl], n = start[cl + 1] - lo0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = lo0 + (int64_t)blockIdx.x * per, hi = lo + per < lo0 + n ? lo + per : lo0 + n;
  if (lo >= hi) return;                                   // uniform over the workgroup
  for (int64_t g = threadIdx.x; g < G; g += 256) s_acc[g] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t t = lo + wave; t < hi; t += 4) {
    const int64_t c = order[t];
    const int64_t p0 = ptr_b[c], p1 = ptr_e[c];
    for (int64_t p = p0 + lane; p < p1; p += 64) {
      const int32_t g = rowidx[p];
      if (g >= 0 && g < G) unsafeAtomicAdd(&s_acc[g], x[p]);       // ds_add_f64
    }
  }
  __syncthreads();
  double* const col = out + (int64_t)cl * G;
  for (int64_t g = threadIdx.x; g < G; g += 256) {
    const double v = s_acc[g];
    if (v != 0.0) unsafeAtomicAdd(col + g, v);
  }
}

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
  if (nnz <= 0) return nk == 0 ? GFICF_OK : GFICF_ERR_BAD_CSC;
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
  gficf_run_shares(nt, [&](int64_t t) {
    const int64_t c0 = t == 0 ? 0 : first_cell(nnz / nt * t), c1 = t + 1 == nt ? N : first_cell(nnz / nt * (t + 1));
    for (int64_t c = c0; c < c1; ++c) {
      int64_t d = kcp[c];
      const int64_t dend = kcp[c + 1], p1 = cp[c + 1];
      for (int64_t q = cp[c]; q < p1; ++q) {
        const uint32_t g = (uint32_t)rowidx[q];
        if (g < (uint64_t)G && keep[g]) {
          if (d >= dend) { bad.store(1); return; }
          out_x[d] = x[q];
          if (rm) out_rowidx[d] = rm[g];
          ++d;
        }
      }
      if (d != dend) { bad.store(1); return; }
    }
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
  if (ci[N] > 0 && (!rowidx || !x || !keep)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  if (co[N] > 0 && !out_x) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  for (int64_t c = 0; c < N; ++c)
    if (ci[c + 1] < ci[c] || co[c + 1] < co[c]) GFICF_FAIL(GFICF_ERR_BAD_CSC, "column pointers not monotone at cell %lld", (long long)c);
  const int rc = kept_values(G, N, ci, rowidx, x, keep, co, out_rowidx, out_x);
  if (rc) GFICF_FAIL(rc, "kept_colptr is not the column pointer of M[keep, ] (the kept entries of some cell do not match its count)");
  return GFICF_OK;
}

}  // extern "C"
