// oracle/jaccard_oracle.cpp — CPU restatement of the reference's Phenograph Jaccard step.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under gficf_amd/ may import, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and only as the checker / the timed CPU baseline.
//
// PARITY UNPINNED: the reference holds no tests or golden vectors for this path
// (reference tests/testthat.R:4 is commented out) and the reference translation unit
// cannot be compiled here (it includes <Rcpp.h> and <RcppParallel.h>, neither of which
// exists in this image, and R itself is absent).  This restatement is therefore checked
// only against hand-derived known answers and an independent numpy restatement
// (oracle/oracle_np.py) — see tests/test_oracle.py.
//
// What it restates (all citations relative to /root/reference):
//   src/rcpp_parallel_jaccard_coeff.cpp:24-55   JCoefficient::operator()  — the hot loop
//   src/rcpp_parallel_jaccard_coeff.cpp:59-80   rcpp_parallel_jaccard_coef — driver
//   src/jaccard_coeff.cpp:19-44                 jaccard_coeff — the serial entry (set intersection, packed rows)
//
// Per edge (i, j) the reference
//   :28     kk = (int)(mat(i,j) - 1)                      (double -> int truncation)
//   :30-36  copies row i and row kk of the column-major double matrix (stride N)
//   :38-39  std::sort on both copies
//   :41-46  std::set_intersection (multiset semantics) -> u = size of the result
//   :48-52  if (u > 0) rmat(i*k + j, 0..2) = (i+1, kk+1, u / (2.0*k - u)); else the
//           row stays at its zero initialisation (:67 allocates a zero-filled matrix).
// The same per-edge work (two strided copies, two sorts, one merge into a growing
// vector<int>) is kept here on purpose: this file is also what bench.py times as the
// "RcppParallel CPU path" (cpu_baseline.kind = "port").
//
// Defined behaviour where the reference has none: a value v whose truncation (int)(v - 1) falls
// outside [0, N) is undefined behaviour in the reference (mat.row(kk) out of range, :34).  Here it is
// rejected up front with return code -1.  Non-integer values inside (0, N + 1) are what the reference
// handles by truncation (row addressed by (int)(v - 1), rows intersected as the doubles they are) and
// are restated as such.
//
// Threading: the reference uses RcppParallel::parallelFor(0, N, worker) (:73), i.e. TBB
// blocked ranges over cells with work stealing.  Here: std::thread workers pulling
// fixed-size cell chunks from an atomic counter.  Results do not depend on the split
// (each (i,j) owns its output row).

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <iterator>
#include <thread>
#include <vector>

namespace {

struct JaccardArgs {
  const double* mat;   // N x k, column-major, 1-based ids stored as doubles
  int64_t N;
  int k;
  double* rmat;        // (N*k) x 3, column-major, zero-initialised by the caller-facing entry
  int32_t* u_out;      // optional N*k intersection counts (row-major edge order i*k+j)
  int64_t col_stride;  // rows of the output matrix (N*k for the whole matrix; (end-begin)*k for a block of cells)
  int64_t row_base;    // output row of edge (i, j) = i*k + j - row_base
};

// reference :24-55, one cell range
void jaccard_range(const JaccardArgs& a, int64_t begin, int64_t end) {
  const int64_t N = a.N;
  const int k = a.k;
  const int64_t E = a.col_stride;
  std::vector<double> v1(k), v2(k);
  for (int64_t i = begin; i < end; ++i) {
    for (int j = 0; j < k; ++j) {
      int kk = (int)(a.mat[(int64_t)j * N + i] - 1);                       // :28
      for (int t = 0; t < k; ++t) v1[t] = a.mat[(int64_t)t * N + i];       // :30-32
      for (int t = 0; t < k; ++t) v2[t] = a.mat[(int64_t)t * N + kk];      // :34-36
      std::sort(v1.begin(), v1.end());                                     // :38
      std::sort(v2.begin(), v2.end());                                     // :39
      std::vector<int> v_intersection;                                     // :41
      std::set_intersection(v1.begin(), v1.end(), v2.begin(), v2.end(),
                            std::back_inserter(v_intersection));           // :43-45
      int u = (int)v_intersection.size();                                  // :46
      const int64_t r = i * (int64_t)k + j - a.row_base;
      if (a.u_out) a.u_out[r] = u;
      if (u > 0) {                                                         // :48
        a.rmat[r] = (double)(i + 1);                                       // :49
        a.rmat[E + r] = (double)(kk + 1);                                  // :50
        a.rmat[2 * E + r] = u / (2.0 * k - u);                             // :51
      }
    }
  }
}

// cells [begin, end) over std::thread workers pulling fixed-size chunks from an atomic counter
void run_cells(const JaccardArgs& a, int64_t begin, int64_t end, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads == 1 || end - begin < 64) {
    jaccard_range(a, begin, end);
    return;
  }
  std::atomic<int64_t> next(begin);
  const int64_t chunk = 64;
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) {
    pool.emplace_back([&]() {
      for (;;) {
        int64_t b = next.fetch_add(chunk);
        if (b >= end) break;
        jaccard_range(a, b, std::min(end, b + chunk));
      }
    });
  }
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// mat:   N x k column-major doubles (what Rcpp hands the reference after coercion,
//        reference src/RcppExports.cpp:65).
// rmat:  caller-allocated (N*k) x 3 column-major doubles; zero-filled here (:67).
// u_out: optional, N*k int32.
// Returns 0, or -1 if an id is outside [1, N] or not finite.
int oracle_jaccard_f64(const double* mat, int64_t N, int k, double* rmat, int32_t* u_out,
                       int nthreads) {
  if (N < 0 || k < 0) return -2;
  const int64_t E = N * (int64_t)k;
  for (int64_t p = 0; p < E; ++p) {
    double v = mat[p];
    if (!(v > 0.0) || !(v < (double)N + 1.0)) return -1;      // (int)(v - 1) in [0, N): what the reference can address (:28,:34)
  }
  std::memset(rmat, 0, sizeof(double) * 3 * (size_t)E);
  JaccardArgs a{mat, N, k, rmat, u_out, E, 0};
  run_cells(a, 0, N, nthreads);
  return 0;
}

// The rows of the reference's matrix that belong to the cells [begin, end) only (a bounded sample for checks at sizes
// where the whole matrix would take minutes on a few cores): rmat_rows is ((end-begin)*k) x 3 column-major, row
// (i-begin)*k + j = the reference's row i*k + j.  Same per-edge code as above.
int oracle_jaccard_cells_f64(const double* mat, int64_t N, int k, int64_t begin, int64_t end, double* rmat_rows,
                             int32_t* u_rows, int nthreads) {
  if (N < 0 || k < 0 || begin < 0 || end < begin || end > N) return -2;
  const int64_t E = N * (int64_t)k;
  for (int64_t p = 0; p < E; ++p) {
    double v = mat[p];
    if (!(v > 0.0) || !(v < (double)N + 1.0)) return -1;      // (int)(v - 1) in [0, N): what the reference can address (:28,:34)
  }
  const int64_t rows = (end - begin) * (int64_t)k;
  std::memset(rmat_rows, 0, sizeof(double) * 3 * (size_t)rows);
  JaccardArgs a{mat, N, k, rmat_rows, u_rows, rows, begin * (int64_t)k};
  run_cells(a, begin, end, nthreads);
  return 0;
}

// The package's serial entry jaccard_coeff(idx, printOutput) (reference src/jaccard_coeff.cpp:19-44):
//   :29-30  for every i, j:  k = idx(i,j) - 1
//   :31-33  u = intersect(idx(i,_), idx(k,_)).size()   — Rcpp sugar intersect: the rows as SETS (unique common values)
//   :34-39  if (u > 0) { weights(r,0) = i+1; weights(r,1) = k+1; weights(r,2) = u/(2.0*ncol - u); r++; }
// so the rows with u > 0 are packed from the top of the zero-initialised (N*k) x 3 matrix (:21), in (i, j) order.
int oracle_jaccard_coeff_f64(const double* mat, int64_t N, int k, double* weights) {
  if (N < 0 || k < 0) return -2;
  const int64_t E = N * (int64_t)k;
  for (int64_t p = 0; p < E; ++p) {
    double v = mat[p];
    if (!(v > 0.0) || !(v < (double)N + 1.0)) return -1;      // (int)(v - 1) in [0, N): what the reference can address (:28,:34)
  }
  std::memset(weights, 0, sizeof(double) * 3 * (size_t)E);
  std::vector<double> a(k), b(k), common;
  int64_t r = 0;
  for (int64_t i = 0; i < N; ++i) {
    for (int j = 0; j < k; ++j) {
      const int kk = (int)(mat[(int64_t)j * N + i] - 1);
      for (int t = 0; t < k; ++t) a[t] = mat[(int64_t)t * N + i];
      for (int t = 0; t < k; ++t) b[t] = mat[(int64_t)t * N + kk];
      std::sort(a.begin(), a.end());
      std::sort(b.begin(), b.end());
      const size_t na = std::unique(a.begin(), a.end()) - a.begin(), nb = std::unique(b.begin(), b.end()) - b.begin();
      common.clear();
      std::set_intersection(a.begin(), a.begin() + na, b.begin(), b.begin() + nb, std::back_inserter(common));
      const int u = (int)common.size();
      if (u > 0) {
        weights[r] = (double)(i + 1);
        weights[E + r] = (double)(kk + 1);
        weights[2 * E + r] = u / (2.0 * k - u);
        ++r;
      }
    }
  }
  return 0;
}

// Convenience for integer kNN matrices (uwot returns an INTSXP; Rcpp coerces it to
// REALSXP before the reference sees it, src/RcppExports.cpp:65).  Same arithmetic.
int oracle_jaccard_i32(const int32_t* mat, int64_t N, int k, double* rmat, int32_t* u_out,
                       int nthreads) {
  std::vector<double> d((size_t)(N * (int64_t)k));
  for (size_t p = 0; p < d.size(); ++p) d[p] = (double)mat[p];
  return oracle_jaccard_f64(d.data(), N, k, rmat, u_out, nthreads);
}

}  // extern "C"
