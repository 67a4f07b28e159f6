// gficf_scale.h — GF-ICF pass B0 (kept entries per cell: the new column pointer) and pass B (GF, ICF, L2 per cell, compacted and renumbered).
// Included by gficf_csc.hip inside its anonymous namespace, after gficf_count.h (the per-gene tables' layout).

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

