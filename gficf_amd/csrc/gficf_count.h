// gficf_count.h — GF-ICF pass A (cells per gene) and the per-gene tables (keep flags, new row ids, ICF weights).  Included by gficf_csc.hip
// inside its anonymous namespace (one translation unit: the split is for reading, the code object is the same).

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

