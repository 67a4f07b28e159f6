// tsan_host_check.cpp — the library's threaded pure-host entries under ThreadSanitizer (no GPU): built and run by
// `make -C gficf_amd/csrc tsan-check` against build_tsan/libgficf_hip_tsan.so.  Random shapes through
//   gficf_csc_kept_values_host (shares of cells on host threads; the values of M[keep, ]) and
//   gficf_jaccard_expand_host  (shares of cells on host threads; compact return -> the (N k) x 3 matrix),
// each compared with a serial restatement.  A data race between two shares is a ThreadSanitizer report (exit code 66).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gficf_hip.h"

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 12;
  std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
  auto uni = [&](int64_t lo, int64_t hi) { return lo + (int64_t)(rng() % (uint64_t)(hi - lo + 1)); };
  int64_t kv = 0, ex = 0;
  for (int r = 0; r < rounds; ++r) {
    {  // ---- kept values: enough stored entries for several shares (one per 2 M entries)
      const int64_t G = uni(50, 3000), N = uni(2000, 12000);
      std::vector<int64_t> cp((size_t)N + 1, 0);
      std::vector<int32_t> ri;
      std::vector<double> x;
      std::vector<uint8_t> keep((size_t)G);
      for (auto& k : keep) k = (rng() % 10) < 7;
      for (int64_t c = 0; c < N; ++c) {
        const int64_t len = (c % 97 == 0 || c + 3 >= N) ? 0 : uni(0, 1500);     // empty cells, an empty tail
        int64_t g = uni(0, 3);
        for (int64_t e = 0; e < len && g < G; ++e) { ri.push_back((int32_t)g); x.push_back((double)uni(0, 9)); g += uni(1, 1 + G / 700); }
        cp[(size_t)c + 1] = (int64_t)ri.size();
      }
      std::vector<int64_t> kcp((size_t)N + 1, 0);
      std::vector<double> want;
      std::vector<int32_t> want_i, remap((size_t)G, -1);
      int32_t nr = 0;
      for (int64_t g = 0; g < G; ++g) if (keep[(size_t)g]) remap[(size_t)g] = nr++;
      for (int64_t c = 0; c < N; ++c) {
        for (int64_t q = cp[(size_t)c]; q < cp[(size_t)c + 1]; ++q)
          if (keep[(size_t)ri[(size_t)q]]) { want.push_back(x[(size_t)q]); want_i.push_back(remap[(size_t)ri[(size_t)q]]); }
        kcp[(size_t)c + 1] = (int64_t)want.size();
      }
      std::vector<double> ox(want.size() + 1, -1.0);
      std::vector<int32_t> oi(want.size() + 1, -1);
      const int rc = gficf_csc_kept_values_host(G, N, cp.data(), 1, ri.data(), x.data(), keep.data(), kcp.data(), (r & 1) ? oi.data() : nullptr, ox.data());
      if (rc != 0) { fprintf(stderr, "kept values: rc %d: %s\n", rc, gficf_last_error()); return 1; }
      for (size_t q = 0; q < want.size(); ++q)
        if (ox[q] != want[q] || ((r & 1) && oi[q] != want_i[q])) { fprintf(stderr, "kept values: entry %zu differs\n", q); return 1; }
      if (ox[want.size()] != -1.0) { fprintf(stderr, "kept values: a store past the vector\n"); return 1; }
      ++kv;
    }
    {  // ---- expansion of the compact return
      const int64_t N = uni(3000, 60000);
      const int k = (int)uni(1, 60);
      const int64_t ld = N + uni(0, 3), E = N * k;
      std::vector<int32_t> idx((size_t)(ld * k));
      for (auto& v : idx) v = (int32_t)uni(1, N);
      std::vector<uint16_t> u((size_t)E);
      for (auto& v : u) v = (uint16_t)uni(0, k);
      std::vector<double> out((size_t)(3 * E), -1.0);
      const int rc = gficf_jaccard_expand_host(idx.data(), 0, N, k, ld, u.data(), out.data(), (int)uni(2, 12));
      if (rc != 0) { fprintf(stderr, "expand: rc %d: %s\n", rc, gficf_last_error()); return 1; }
      for (int64_t i = 0; i < N; ++i)
        for (int j = 0; j < k; ++j) {
          const int64_t e = i * k + j;
          const double uu = (double)u[(size_t)e];
          const double w0 = uu > 0 ? (double)(i + 1) : 0.0, w1 = uu > 0 ? (double)idx[(size_t)(j * ld + i)] : 0.0, w2 = uu > 0 ? uu / (2.0 * k - uu) : 0.0;
          if (out[(size_t)e] != w0 || out[(size_t)(E + e)] != w1 || out[(size_t)(2 * E + e)] != w2) { fprintf(stderr, "expand: edge %lld differs\n", (long long)e); return 1; }
        }
      ++ex;
    }
  }
  printf("tools/tsan_host_check.cpp: %lld kept-values cases and %lld expansions under ThreadSanitizer: no report, every result equal to the serial restatement\n",
         (long long)kv, (long long)ex);
  return 0;
}
