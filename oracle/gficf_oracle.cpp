// oracle/gficf_oracle.cpp — CPU restatement of the reference's GF-ICF normalisation.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under gficf_amd/ may import, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference path is pure R on the (un-vendored, un-pinned) Matrix
// package (reference DESCRIPTION:26, NAMESPACE:23) and R is absent from this image, so
// the reference cannot be run; it holds no tests or golden vectors for this path.  This
// restatement is checked against hand-derived known answers and an
// independent numpy/scipy restatement (oracle/oracle_np.py).
//
// What it restates, in the reference's order of operations (citations relative to
// /root/reference):
//   R/gficf.R:40-41    normCounts: ix_g = #{c : M[g,c] != 0}; keep gene iff
//                      ix_g > N*min  &  ix_g <= N*max        (comparison in double)
//   R/gficf.R:59       tf: S_c = colSums(M) over the kept genes; M[g,c] / S_c
//   R/gficf.R:88-89    getIdfW("classic"): nt_g = #{c : tf[g,c] != 0};
//                      w_g = log((N+1)/(nt_g+1))              (natural log)
//   R/gficf.R:79       idf: tf[g,c] * w_g
//   R/gficf.R:100-103  l.norm "l2" on the transposed matrix (R/gficf.R:25):
//                      n_c = 1/sqrt(sum_g v^2); n_c = 0 if infinite; n_c * v
//   R/gficf.R:43-47    edgeR TMM/CPM branch is NOT restated: it is a per-cell scale
//                      that cancels in tf (x/colSum(x)); it only changes $rawCounts.
//   R/cellClassifier.R:50-53  second caller: weights supplied (w_in != NULL), not
//                      recomputed.
// Sums run sequentially in storage order in double, as Matrix's colSums/rowSums do on a
// dgCMatrix.
//
// Defined behaviour where R's is degenerate: a cell whose kept entries sum to 0
// (S_c == 0) yields 0/0 = NaN over the whole densified column in R; here every stored
// kept entry of such a cell is written as 0.0.  Inputs are
// expected to be non-negative counts.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

extern "C" {

// Input: CSC genes x cells.  colptr[N+1] (int64), rowidx[nnz] (0-based, int32), x[nnz].
// w_in: NULL -> compute ICF weights (gficf()); else length-G weights indexed by the
//       ORIGINAL gene index (embedNewCells()).
// Outputs (caller-allocated):
//   keep[G] (0/1), nt[G] (int64; nt of dropped genes = 0), w[G] (0 for dropped genes),
//   out_colptr[N+1], out_rowidx[<=nnz] (renumbered over kept genes), out_x[<=nnz],
//   *G_kept, *nnz_kept.
// Returns 0 on success, -1 on malformed structure.
// icf_type: getIdfW(type) R/gficf.R:89-91 (0 classic, 1 prob, 2 smooth); norm_l1: l.norm(norm = "l1") R/gficf.R:100.
// gficf() itself always runs (0, 0).
int oracle_gficf_csc_ex(int64_t G, int64_t N, const int64_t* colptr, const int32_t* rowidx,
                        const double* x, double prop_min, double prop_max, const double* w_in, int icf_type, int norm_l1,
                        uint8_t* keep, int64_t* nt, double* w, int64_t* out_colptr,
                        int32_t* out_rowidx, double* out_x, int64_t* G_kept, int64_t* nnz_kept) {
  if (G < 0 || N < 0 || colptr[0] != 0) return -1;
  const int64_t nnz = colptr[N];
  for (int64_t p = 0; p < nnz; ++p)
    if (rowidx[p] < 0 || rowidx[p] >= G) return -1;

  // R/gficf.R:40  ix = rowSums(M != 0)
  std::vector<int64_t> ix((size_t)G, 0);
  for (int64_t p = 0; p < nnz; ++p)
    if (x[p] != 0.0) ix[rowidx[p]]++;
  // R/gficf.R:41  keep = ix > ncol(M)*min & ix <= ncol(M)*max
  std::vector<int32_t> remap((size_t)G, -1);
  int64_t gk = 0;
  for (int64_t g = 0; g < G; ++g) {
    bool kp = ((double)ix[g] > (double)N * prop_min) && ((double)ix[g] <= (double)N * prop_max);
    keep[g] = kp ? 1 : 0;
    if (kp) remap[g] = (int32_t)gk++;
    nt[g] = 0;
    w[g] = 0.0;
  }
  *G_kept = gk;

  // subset rows (R/gficf.R:41) -> compacted CSC, then tf (R/gficf.R:59)
  int64_t q = 0;
  out_colptr[0] = 0;
  for (int64_t c = 0; c < N; ++c) {
    const int64_t q0 = q;
    double S = 0.0;
    for (int64_t p = colptr[c]; p < colptr[c + 1]; ++p) {
      int32_t g = rowidx[p];
      if (!keep[g]) continue;
      out_rowidx[q] = remap[g];
      out_x[q] = x[p];
      S += x[p];                                   // colSums, storage order
      ++q;
    }
    for (int64_t t = q0; t < q; ++t) out_x[t] = (S != 0.0) ? out_x[t] / S : 0.0;
    out_colptr[c + 1] = q;
  }
  *nnz_kept = q;

  // R/gficf.R:88-89  nt = rowSums(tf != 0); w = log((ncol+1)/(nt+1))
  std::vector<int64_t> ntk((size_t)gk, 0);
  for (int64_t t = 0; t < q; ++t)
    if (out_x[t] != 0.0) ntk[out_rowidx[t]]++;
  std::vector<double> wk((size_t)gk, 0.0);
  for (int64_t g = 0; g < G; ++g) {
    if (!keep[g]) continue;
    int32_t r = remap[g];
    nt[g] = ntk[r];
    const double c = (double)ntk[r];
    if (w_in) wk[r] = w_in[g];
    else if (icf_type == 1) wk[r] = std::log(((double)N - c) / c);            // "prob"    R/gficf.R:90
    else if (icf_type == 2) wk[r] = std::log(1.0 + (double)N / c);            // "smooth"  R/gficf.R:91
    else wk[r] = std::log(((double)N + 1.0) / (c + 1.0));                     // "classic" R/gficf.R:89
    w[g] = wk[r];
  }

  // R/gficf.R:79  M * w ;  R/gficf.R:100-103  l2 per cell
  for (int64_t c = 0; c < N; ++c) {
    double ss = 0.0;
    for (int64_t t = out_colptr[c]; t < out_colptr[c + 1]; ++t) {
      out_x[t] = out_x[t] * wk[out_rowidx[t]];
      ss += norm_l1 ? out_x[t] : out_x[t] * out_x[t];
    }
    double nv = 1.0 / (norm_l1 ? ss : std::sqrt(ss));   // l1: 1/rowSums(m), l2: 1/sqrt(rowSums(m^2))  R/gficf.R:100
    if (std::isinf(nv)) nv = 0.0;                  // R/gficf.R:101
    for (int64_t t = out_colptr[c]; t < out_colptr[c + 1]; ++t) out_x[t] = nv * out_x[t];
  }
  return 0;
}

// The same computation over several host threads (the CPU baseline SURVEY.md §8d asks for: R itself runs this path on one
// thread; a multi-threaded restatement is the stronger host-side competitor).  Cells are cut into contiguous ranges, one
// per thread; the per-gene counts are per-thread histograms added up afterwards (integers: order does not matter); every
// per-cell sum runs in storage order exactly as in oracle_gficf_csc_ex, so the outputs are bit-identical to it.
int oracle_gficf_csc_mt(int64_t G, int64_t N, const int64_t* colptr, const int32_t* rowidx,
                        const double* x, double prop_min, double prop_max, const double* w_in, int icf_type, int norm_l1,
                        int threads, uint8_t* keep, int64_t* nt, double* w, int64_t* out_colptr,
                        int32_t* out_rowidx, double* out_x, int64_t* G_kept, int64_t* nnz_kept) {
  if (G < 0 || N < 0 || colptr[0] != 0) return -1;
  const int64_t nnz = colptr[N];
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, std::max<int64_t>(N, 1)));
  // cell ranges with about equal numbers of stored entries
  std::vector<int64_t> cut((size_t)T + 1, N);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) cut[t] = std::lower_bound(colptr, colptr + N + 1, nnz * t / T) - colptr;
  for (int t = 1; t <= T; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
  cut[T] = N;
  auto run = [&](const std::function<void(int)>& fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto& h : th) h.join();
  };
  std::vector<int> bad((size_t)T, 0);
  std::vector<std::vector<int64_t>> part((size_t)T);
  run([&](int t) {                                   // R/gficf.R:40  ix = rowSums(M != 0)
    part[t].assign((size_t)G, 0);
    for (int64_t p = colptr[cut[t]]; p < colptr[cut[t + 1]]; ++p) {
      if (rowidx[p] < 0 || rowidx[p] >= G) { bad[t] = 1; return; }
      if (x[p] != 0.0) part[t][rowidx[p]]++;
    }
  });
  for (int t = 0; t < T; ++t)
    if (bad[t]) return -1;
  std::vector<int32_t> remap((size_t)G, -1);
  int64_t gk = 0;
  for (int64_t g = 0; g < G; ++g) {                  // R/gficf.R:41
    int64_t ix = 0;
    for (int t = 0; t < T; ++t) ix += part[t][g];
    const bool kp = ((double)ix > (double)N * prop_min) && ((double)ix <= (double)N * prop_max);
    keep[g] = kp ? 1 : 0;
    if (kp) remap[g] = (int32_t)gk++;
    nt[g] = 0;
    w[g] = 0.0;
  }
  *G_kept = gk;
  run([&](int t) {                                   // kept entries per cell
    for (int64_t c = cut[t]; c < cut[t + 1]; ++c) {
      int64_t n = 0;
      for (int64_t p = colptr[c]; p < colptr[c + 1]; ++p) n += keep[rowidx[p]];
      out_colptr[c + 1] = n;
    }
  });
  out_colptr[0] = 0;
  for (int64_t c = 0; c < N; ++c) out_colptr[c + 1] += out_colptr[c];
  *nnz_kept = out_colptr[N];
  run([&](int t) {                                   // subset + tf (R/gficf.R:41,59), nt of the kept genes (R/gficf.R:88)
    part[t].assign((size_t)gk, 0);
    for (int64_t c = cut[t]; c < cut[t + 1]; ++c) {
      int64_t q = out_colptr[c];
      double S = 0.0;
      for (int64_t p = colptr[c]; p < colptr[c + 1]; ++p) {
        const int32_t g = rowidx[p];
        if (!keep[g]) continue;
        out_rowidx[q] = remap[g];
        out_x[q] = x[p];
        S += x[p];
        ++q;
      }
      for (int64_t u = out_colptr[c]; u < q; ++u) {
        out_x[u] = (S != 0.0) ? out_x[u] / S : 0.0;
        if (out_x[u] != 0.0) part[t][out_rowidx[u]]++;
      }
    }
  });
  std::vector<double> wk((size_t)gk, 0.0);
  for (int64_t g = 0; g < G; ++g) {                  // R/gficf.R:89-91
    if (!keep[g]) continue;
    const int32_t r = remap[g];
    int64_t n = 0;
    for (int t = 0; t < T; ++t) n += part[t][r];
    nt[g] = n;
    const double c = (double)n;
    if (w_in) wk[r] = w_in[g];
    else if (icf_type == 1) wk[r] = std::log(((double)N - c) / c);
    else if (icf_type == 2) wk[r] = std::log(1.0 + (double)N / c);
    else wk[r] = std::log(((double)N + 1.0) / (c + 1.0));
    w[g] = wk[r];
  }
  run([&](int t) {                                   // R/gficf.R:79, :100-103
    for (int64_t c = cut[t]; c < cut[t + 1]; ++c) {
      double ss = 0.0;
      for (int64_t u = out_colptr[c]; u < out_colptr[c + 1]; ++u) {
        out_x[u] = out_x[u] * wk[out_rowidx[u]];
        ss += norm_l1 ? out_x[u] : out_x[u] * out_x[u];
      }
      double nv = 1.0 / (norm_l1 ? ss : std::sqrt(ss));
      if (std::isinf(nv)) nv = 0.0;
      for (int64_t u = out_colptr[c]; u < out_colptr[c + 1]; ++u) out_x[u] = nv * out_x[u];
    }
  });
  return 0;
}

int oracle_gficf_csc(int64_t G, int64_t N, const int64_t* colptr, const int32_t* rowidx,
                     const double* x, double prop_min, double prop_max, const double* w_in,
                     uint8_t* keep, int64_t* nt, double* w, int64_t* out_colptr,
                     int32_t* out_rowidx, double* out_x, int64_t* G_kept, int64_t* nnz_kept) {
  return oracle_gficf_csc_ex(G, N, colptr, rowidx, x, prop_min, prop_max, w_in, 0, 0, keep, nt, w, out_colptr, out_rowidx, out_x,
                             G_kept, nnz_kept);
}

}  // extern "C"
