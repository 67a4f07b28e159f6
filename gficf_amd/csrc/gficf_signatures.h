// gficf_signatures.h — cluster signatures (next row N3: data$cluster.gene.rnk, reference R/clustCells.R:121-123).  Included by gficf_csc.hip
// inside its anonymous namespace.

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

