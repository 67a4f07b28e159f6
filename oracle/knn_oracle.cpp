// knn_oracle.cpp — CPU brute-force k-nearest-neighbour oracle.  TEST INFRASTRUCTURE ONLY (see
// oracle/__init__.py): never imported, linked or called by the product package.
//
// The step it restates is the caller's line in front of the Jaccard build,
//   neigh = uwot:::find_nn(data$pca$cells, k = k+1, include_self = T, method = "annoy",
//                          metric = dist.method)$idx            (reference R/clustCells.R:57,60)
// uwot 0.x / RcppAnnoy are third-party dependencies of the reference (DESCRIPTION: Imports uwot,
// unpinned) and absent from /root/reference; Annoy is an APPROXIMATE forest search whose published
// contract is "the k items with the smallest metric distance, as far as the forest finds them",
// computed in f32.  The oracle is the exact form of that contract: all N distances per query in
// f32, the k smallest (distance, index) pairs, ties broken by the smaller index.  PARITY UNPINNED:
// the reference holds no fixture for this step and Annoy's output is seed- and build-dependent.
//
// Arithmetic (f32, in dimension order, the same chain the HIP kernel evaluates):
//   manhattan  acc = acc + |a - b|                      (Annoy Manhattan::distance)
//   euclidean  df = a - b; acc = fma(df, df, acc); sqrt at the end   (Annoy Euclidean, sqrt in get_nns)
//   cosine     rows divided by their f32 L2 norm first; acc = fma(a, b, acc); dist = 1 - acc
//   correlation  the same on rows with their f32 mean removed first (uwot "correlation" = Annoy angular on centred rows)
//              (uwot converts Annoy's angular distance sqrt(2(1-cos)) to 1 - cos)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint32_t f32_sortable(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
inline float sortable_f32(uint32_t s) {
  const uint32_t u = s ^ (((s >> 31) - 1u) | 0x80000000u);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// The fma chains use the hardware instruction (bit-identical to libm's fmaf, only much faster); a host
// without it is refused (rc 2) rather than silently slow.
#if defined(__x86_64__)
#define KNN_TARGET __attribute__((target("fma")))
#else
#define KNN_TARGET
#endif

template <int METRIC>
KNN_TARGET void query_block(const float* P, int64_t N, int d, int64_t q0, int64_t q1, int k, int32_t* idx, double* dist, int64_t ld) {
  std::vector<uint64_t> keys((size_t)N);
  for (int64_t q = q0; q < q1; ++q) {
    const float* a = P + q * d;
    for (int64_t j = 0; j < N; ++j) {
      const float* b = P + j * d;
      float acc = 0.0f;
      for (int t = 0; t < d; ++t) {
        if (METRIC == 0) {
          acc = acc + std::fabs(a[t] - b[t]);
        } else if (METRIC == 1) {
          const float df = a[t] - b[t];
          acc = __builtin_fmaf(df, df, acc);
        } else {
          acc = __builtin_fmaf(a[t], b[t], acc);
        }
      }
      const float dv = METRIC == 2 ? 1.0f - acc : acc;
      keys[(size_t)j] = ((uint64_t)f32_sortable(dv) << 32) | (uint64_t)(uint32_t)j;
    }
    std::partial_sort(keys.begin(), keys.begin() + k, keys.end());
    for (int t = 0; t < k; ++t) {
      idx[(int64_t)t * ld + q] = (int32_t)(uint32_t)keys[(size_t)t] + 1;
      float dv = sortable_f32((uint32_t)(keys[(size_t)t] >> 32));
      if (METRIC == 1) dv = std::sqrt(dv);
      if (dist) dist[(int64_t)t * ld + q] = (double)dv;
    }
  }
}

}  // namespace

// X: N x d column-major doubles (ld >= N).  idx / dist: N x k column-major.  metric: 0 manhattan,
// 1 euclidean, 2 cosine.  Returns 0, 1 for invalid arguments, 2 for a host without fma.
KNN_TARGET static void prepare_rows(const double* X, int64_t N, int d, int64_t ld, int metric, float* P) {
  for (int64_t r = 0; r < N; ++r) {
    float nrm = 0.0f, mean = 0.0f;
    bool scale = false;
    if (metric == 3) {                               // correlation: centre the row (f32 sum in dimension order / d)
      float sm = 0.0f;
      for (int t = 0; t < d; ++t) sm = sm + (float)X[(int64_t)t * ld + r];
      mean = sm / (float)d;
    }
    if (metric >= 2) {
      float s = 0.0f;
      for (int t = 0; t < d; ++t) { const float v = (float)X[(int64_t)t * ld + r] - mean; s = __builtin_fmaf(v, v, s); }
      nrm = std::sqrt(s);
      scale = nrm > 0.0f;
    }
    for (int t = 0; t < d; ++t) {
      float v = (float)X[(int64_t)t * ld + r] - mean;
      if (scale) v = v / nrm;
      P[(size_t)r * d + t] = v;
    }
  }
}

// Queries [q_begin, q_end) only (a bounded sample of the same workload for the timed CPU baseline); idx / dist
// keep the full N x k column-major shape, rows outside the range are left untouched.
extern "C" int oracle_knn_block(const double* X, int64_t N, int d, int64_t ld, int k, int metric, int64_t q_begin, int64_t q_end,
                                int32_t* idx, double* dist, int nthreads) {
  if (N < 0 || d <= 0 || k < 0 || k > N || ld < N || metric < 0 || metric > 3 || q_begin < 0 || q_end < q_begin || q_end > N) return 1;
  if (N == 0 || k == 0 || q_end == q_begin) return 0;
#if defined(__x86_64__)
  if (!__builtin_cpu_supports("fma")) return 2;
#endif
  std::vector<float> P((size_t)N * (size_t)d);
  prepare_rows(X, N, d, ld, metric, P.data());
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> th;
  const int64_t per = (q_end - q_begin + nthreads - 1) / nthreads;
  for (int w = 0; w < nthreads; ++w) {
    const int64_t q0 = std::min<int64_t>(q_end, q_begin + w * per), q1 = std::min<int64_t>(q_end, q0 + per);
    if (q0 >= q1) continue;
    th.emplace_back([=, &P]() {
      const float* p = P.data();
      if (metric == 0) query_block<0>(p, N, d, q0, q1, k, idx, dist, N);
      else if (metric == 1) query_block<1>(p, N, d, q0, q1, k, idx, dist, N);
      else query_block<2>(p, N, d, q0, q1, k, idx, dist, N);      // cosine, and correlation on the centred rows
    });
  }
  for (auto& t : th) t.join();
  return 0;
}

extern "C" int oracle_knn(const double* X, int64_t N, int d, int64_t ld, int k, int metric, int32_t* idx, double* dist,
                          int nthreads) {
  return oracle_knn_block(X, N, d, ld, k, metric, 0, N, idx, dist, nthreads);
}
