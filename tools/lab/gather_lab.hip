// gather_lab.hip — what bounds the row gather of the Jaccard edge kernel?  (tools only, not product code)
//
// A stripped model of k_jaccard_edges' memory pattern (one wave per cell; own row; k random row gathers of
// 16 B per lane, all in flight together; 24 B/edge written as three coalesced runs), without the hash set,
// so that the memory side can be varied alone:
//   * row bytes 128 (k ids x 4 B, the product's table) or 64 (k <= 30 ids as 16-bit low halves + one word of high bits)
//   * load flavour: plain / nt (non-temporal) / sc1 (L1 bypass)
//   * ids: windowed kNN relabelled by a random permutation (the bench input), the same without the relabelling
//     (perfect locality), uniformly random
//   * with / without the output stores
// and, linked against libgficf_hip.so, the product kernel on the same inputs for reference.
//
// Build: hipcc -O3 --offload-arch=gfx950 -I include tools/lab/gather_lab.hip -L gficf_amd -lgficf_hip -Wl,-rpath,'$ORIGIN/../../gficf_amd' -o tools/lab/gather_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gficf_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int FLAV>
__device__ inline v4u load16(const char* p) {
  if (FLAV == 1) return __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
  if (FLAV == 2) {
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
  }
  return *reinterpret_cast<const v4u*>(p);
}

// ROWB: bytes per table row.  128: uint32 ids[32].  64: uint16 lo[30], uint32 hi (bit j = bit 16 of id j).
// STORES: 0 none, 1 non-temporal, 2 plain.  fold: gathered row index & fold (0xffffffff: the real table; a smaller
// power of two minus one keeps the gathers random but inside a table prefix that fits an L2)
template <int ROWB, int FLAV, int STORES>
__global__ __launch_bounds__(256) void k_model(const char* __restrict__ table, long N, int k, double* __restrict__ o_src,
                                               double* __restrict__ o_dst, double* __restrict__ o_w, uint32_t fold) {
  constexpr int LPR = ROWB / 16, RPS = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  const int grow = lane / LPR;
  const uint32_t gcol = (uint32_t)(lane % LPR) * 16u;
  auto own = [&](long i) -> uint32_t {
    if (lane >= k) return 0u;
    if (ROWB == 128) return reinterpret_cast<const uint32_t*>(table + i * ROWB)[lane];
    const uint32_t lo = reinterpret_cast<const uint16_t*>(table + i * ROWB)[lane];
    const uint32_t hi = reinterpret_cast<const uint32_t*>(table + i * ROWB)[15];
    return lo | (((hi >> lane) & 1u) << 16);
  };
  long i = w0;
  uint32_t a_next = i < N ? own(i) : 0u;
  for (; i < N; i += nw) {
    const uint32_t a = a_next;
    const uint32_t asafe = a != 0 ? a : (uint32_t)(i + 1);
    if (i + nw < N) a_next = own(i + nw);
    const int steps = (k + RPS - 1) / RPS;
    uint32_t acc[4] = {0, 0, 0, 0};
    v4u bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < steps) {
        const uint32_t dst = (uint32_t)__shfl((int)asafe, s * RPS + grow);
        bv[s] = load16<FLAV>(table + (size_t)((dst - 1) & fold) * ROWB + gcol);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < steps) acc[s] = bv[s].x ^ bv[s].y ^ bv[s].z ^ bv[s].w;
    // fold to one value per slot (stands for the intersection count)
    uint32_t u = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < steps) {
        uint32_t x = acc[s];
        for (int d = 1; d < LPR; d <<= 1) x ^= __shfl_xor((int)x, d);
        const uint32_t v = (uint32_t)__shfl((int)x, (lane % RPS) * LPR);
        u = (lane / RPS == s) ? v : u;
      }
    }
    if (STORES == 1 && lane < k) {
      const long r = i * k + lane;
      __builtin_nontemporal_store((double)(uint32_t)(i + 1), o_src + r);
      __builtin_nontemporal_store((double)a, o_dst + r);
      __builtin_nontemporal_store((double)(u & 0xffu), o_w + r);
    }
    if (STORES == 2 && lane < k) {
      const long r = i * k + lane;
      o_src[r] = (double)(uint32_t)(i + 1);
      o_dst[r] = (double)a;
      o_w[r] = (double)(u & 0xffu);
    }
    if (!STORES && u == 0xdeadbeefu) o_w[0] = 1.0;
  }
}


// XCD split inside ONE launch: the blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), so block b
// belongs to group (b % 8) * NS / 8; a group walks ALL cells but gathers only the neighbour rows in ITS 1/NS of the table
// (the rows one XCD's L2 sees then fit it) and writes only those slots' edges (scattered 8 B stores: the other groups write
// the neighbouring slots of the same lines).  Memory-side model only (no hash set): is the L2 residency worth the scattered
// writes and the NS-fold own-row reads?
template <int NS, int STORES>
__global__ __launch_bounds__(256) void k_model_xcd(const char* __restrict__ table, long N, int k, double* __restrict__ o_src,
                                                   double* __restrict__ o_dst, double* __restrict__ o_w, uint32_t fold) {
  constexpr int ROWB = 64, LPR = ROWB / 16, RPS = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int x = blockIdx.x % 8, g = x * NS / 8;
  const uint32_t lo = (uint32_t)((unsigned long long)N * g / NS), hi = (uint32_t)((unsigned long long)N * (g + 1) / NS);
  const long bg = (long)(blockIdx.x / 8) * (8 / NS) + x % (8 / NS), nbg = (long)gridDim.x / NS;
  const long w0 = (bg * 256 + threadIdx.x) >> 6, nw = (nbg * 256) >> 6;
  const int grow = lane / LPR;
  const uint32_t gcol = (uint32_t)(lane % LPR) * 16u;
  auto own = [&](long i) -> uint32_t {
    if (lane >= k) return 0u;
    const uint32_t lo16 = reinterpret_cast<const uint16_t*>(table + i * ROWB)[lane];
    const uint32_t hi16 = reinterpret_cast<const uint32_t*>(table + i * ROWB)[15];
    return lo16 | (((hi16 >> lane) & 1u) << 16);
  };
  long i = w0;
  uint32_t a_next = i < N ? own(i) : 0u;
  for (; i < N; i += nw) {
    const uint32_t a = a_next;
    const uint32_t asafe = a != 0 ? a : (uint32_t)(i + 1);
    if (i + nw < N) a_next = own(i + nw);
    const int steps = (k + RPS - 1) / RPS;
    uint32_t acc[4] = {0, 0, 0, 0};
    v4u bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bv[s] = v4u{0, 0, 0, 0};
      if (s < steps) {
        const uint32_t dst = (uint32_t)__shfl((int)asafe, s * RPS + grow) - 1u;
        if (dst >= lo && dst < hi) bv[s] = load16<0>(table + (size_t)(dst & fold) * ROWB + gcol);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < steps) acc[s] = bv[s].x ^ bv[s].y ^ bv[s].z ^ bv[s].w;
    uint32_t u = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < steps) {
        uint32_t xx = acc[s];
        for (int d = 1; d < LPR; d <<= 1) xx ^= __shfl_xor((int)xx, d);
        const uint32_t v = (uint32_t)__shfl((int)xx, (lane % RPS) * LPR);
        u = (lane / RPS == s) ? v : u;
      }
    }
    const bool mine = lane < k && a != 0 && (a - 1u) >= lo && (a - 1u) < hi;
    if (STORES == 1 && mine) {
      const long r = i * k + lane;
      __builtin_nontemporal_store((double)(uint32_t)(i + 1), o_src + r);
      __builtin_nontemporal_store((double)a, o_dst + r);
      __builtin_nontemporal_store((double)(u & 0xffu), o_w + r);
    }
    if (!STORES && u == 0xdeadbeefu) o_w[0] = 1.0;
  }
}

// Range passes: the same model, but one launch gathers only the neighbour rows whose id lies in [lo, hi) (the other lanes of
// the gather instruction are masked off and issue no request), so that the rows a launch touches fit an XCD's L2.  Partial
// per-slot results travel between the passes in a byte plane (32 B per cell); the last pass adds them and writes the edges.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void k_model_range(const char* __restrict__ table, long N, int k, double* __restrict__ o_src,
                                                     double* __restrict__ o_dst, double* __restrict__ o_w, uint8_t* __restrict__ plane,
                                                     uint32_t lo, uint32_t hi) {
  constexpr int ROWB = 64, LPR = ROWB / 16, RPS = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  const int grow = lane / LPR;
  const uint32_t gcol = (uint32_t)(lane % LPR) * 16u;
  auto own = [&](long i) -> uint32_t {
    if (lane >= k) return 0u;
    const uint32_t lo16 = reinterpret_cast<const uint16_t*>(table + i * ROWB)[lane];
    const uint32_t hi16 = reinterpret_cast<const uint32_t*>(table + i * ROWB)[15];
    return lo16 | (((hi16 >> lane) & 1u) << 16);
  };
  long i = w0;
  uint32_t a_next = i < N ? own(i) : 0u;
  for (; i < N; i += nw) {
    const uint32_t a = a_next;
    const uint32_t asafe = a != 0 ? a : (uint32_t)(i + 1);
    if (i + nw < N) a_next = own(i + nw);
    uint32_t part = 0;
    if (!FIRST && lane < k) part = plane[i * 32 + lane];
    const int steps = (k + RPS - 1) / RPS;
    uint32_t acc[4] = {0, 0, 0, 0};
    v4u bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bv[s] = v4u{0, 0, 0, 0};
      if (s < steps) {
        const uint32_t dst = (uint32_t)__shfl((int)asafe, s * RPS + grow) - 1u;
        if (dst >= lo && dst < hi) bv[s] = load16<0>(table + (size_t)dst * ROWB + gcol);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < steps) acc[s] = bv[s].x ^ bv[s].y ^ bv[s].z ^ bv[s].w;
    uint32_t u = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < steps) {
        uint32_t x = acc[s];
        for (int d = 1; d < LPR; d <<= 1) x ^= __shfl_xor((int)x, d);
        const uint32_t v = (uint32_t)__shfl((int)x, (lane % RPS) * LPR);
        u = (lane / RPS == s) ? v : u;
      }
    }
    u = (u + part) & 0xffu;
    if (!LAST && lane < k) plane[i * 32 + lane] = (uint8_t)u;
    if (LAST && lane < k) {
      const long r = i * k + lane;
      __builtin_nontemporal_store((double)(uint32_t)(i + 1), o_src + r);
      __builtin_nontemporal_store((double)a, o_dst + r);
      __builtin_nontemporal_store((double)u, o_w + r);
    }
  }
}

// Quad stores: a wave takes FOUR consecutive cells, parks (dst id, count) of each in LDS and writes the three output
// arrays once per quad as 60 lanes x 16 B (4 cells x 30 slots x 8 B = 960 B = 15 whole 64 B segments per array) instead
// of four times 30 lanes x 8 B (240 B runs that straddle segments).  k = 30 only.
__global__ __launch_bounds__(256) void k_model_quad(const char* __restrict__ table, long N, int k, double* __restrict__ o_src,
                                                    double* __restrict__ o_dst, double* __restrict__ o_w, uint32_t fold) {
  constexpr int ROWB = 64, LPR = ROWB / 16, RPS = 64 / LPR;
  __shared__ uint32_t stage[4][4][32][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  const int grow = lane / LPR;
  const uint32_t gcol = (uint32_t)(lane % LPR) * 16u;
  auto own = [&](long i) -> uint32_t {
    if (lane >= k || i >= N) return 0u;
    const uint32_t lo = reinterpret_cast<const uint16_t*>(table + i * ROWB)[lane];
    const uint32_t hi = reinterpret_cast<const uint32_t*>(table + i * ROWB)[15];
    return lo | (((hi >> lane) & 1u) << 16);
  };
  const long nq = (N + 3) / 4;
  for (long q = w0; q < nq; q += nw) {
    uint32_t a_next = own(4 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long i = 4 * q + c;
      const uint32_t a = a_next;
      const uint32_t asafe = a != 0 ? a : (uint32_t)((i < N ? i : 0) + 1);
      if (c < 3) a_next = own(i + 1);
      v4u bv[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t dst = (uint32_t)__shfl((int)asafe, s * RPS + grow);
        bv[s] = load16<0>(table + (size_t)((dst - 1) & fold) * ROWB + gcol);
      }
      uint32_t u = 0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t x = bv[s].x ^ bv[s].y ^ bv[s].z ^ bv[s].w;
        for (int d = 1; d < LPR; d <<= 1) x ^= __shfl_xor((int)x, d);
        const uint32_t v = (uint32_t)__shfl((int)x, (lane % RPS) * LPR);
        u = (lane / RPS == s) ? v : u;
      }
      if (lane < 32) { stage[wave][c][lane][0] = a; stage[wave][c][lane][1] = u & 0xffu; }
    }
    // 120 edges of the quad, two per lane
    if (lane < 60) {
      typedef double v2d __attribute__((ext_vector_type(2)));
      const int e0 = 2 * lane, c0 = e0 / 30, j0 = e0 % 30, e1 = e0 + 1, c1 = e1 / 30, j1 = e1 % 30;
      const uint32_t a0 = stage[wave][c0][j0][0], u0 = stage[wave][c0][j0][1];
      const uint32_t a1 = stage[wave][c1][j1][0], u1 = stage[wave][c1][j1][1];
      const long r = 4 * q * 30 + e0;
      if (4 * q + 3 < N) {
        __builtin_nontemporal_store(v2d{(double)(uint32_t)(4 * q + c0 + 1), (double)(uint32_t)(4 * q + c1 + 1)}, reinterpret_cast<v2d*>(o_src + r));
        __builtin_nontemporal_store(v2d{(double)a0, (double)a1}, reinterpret_cast<v2d*>(o_dst + r));
        __builtin_nontemporal_store(v2d{(double)u0, (double)u1}, reinterpret_cast<v2d*>(o_w + r));
      }
    }
  }
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}

// ids: N x k row-major, 1-based.  mode 0: windowed (W = 100) + random relabelling, 1: windowed, cells in order, 2: uniform
static std::vector<uint32_t> make_ids(long N, int k, int mode) {
  std::vector<uint32_t> ids((size_t)N * k);
  std::vector<uint32_t> pi(N);
  for (long i = 0; i < N; ++i) pi[i] = (uint32_t)i;
  if (mode == 0)
    for (long i = N - 1; i > 0; --i) std::swap(pi[i], pi[rnd() % (i + 1)]);
  const int W = 100;
  std::vector<int> cand(2 * W);
  for (long c = 0; c < N; ++c) {
    uint32_t* row = &ids[(size_t)pi[c] * k];
    if (mode == 2) {
      for (int t = 0; t < k; ++t) {
        for (;;) {
          uint32_t v = (uint32_t)(rnd() % N);
          bool ok = v != (uint32_t)c;
          for (int t2 = 0; t2 < t && ok; ++t2) ok = row[t2] != v + 1;
          if (ok) { row[t] = v + 1; break; }
        }
      }
      continue;
    }
    for (int t = 0; t < W; ++t) { cand[t] = t - W; cand[W + t] = t + 1; }
    for (int t = 0; t < k; ++t) {
      const int r = t + (int)(rnd() % (2 * W - t));
      std::swap(cand[t], cand[r]);
      const long nb = ((c + cand[t]) % N + N) % N;
      row[t] = pi[nb] + 1;
    }
  }
  return ids;
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 100000;
  const int k = argc > 2 ? atoi(argv[2]) : 30;
  const bool product_only = argc > 3 && atoi(argv[3]) != 0;
  const long E = N * k;
  if (k > 30 || N >= (1 << 17)) { printf("the 64 B row model needs k <= 30 and N < 2^17\n"); return 1; }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; N = %ld, k = %d, E = %ld\n", prop.name, cus, N, k, E);
  double *s, *d, *w;
  CK(hipMalloc(&s, E * 8)); CK(hipMalloc(&d, E * 8)); CK(hipMalloc(&w, E * 8));
  char *t128, *t64;
  CK(hipMalloc(&t128, N * 128)); CK(hipMalloc(&t64, N * 64));
  int32_t* d_idx_cm;
  CK(hipMalloc(&d_idx_cm, E * 4));
  int32_t* d_table;
  CK(hipMalloc(&d_table, N * 128));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gficf_ctx* ctx = nullptr;
  if (gficf_ctx_create(0, nullptr, &ctx) != 0) { printf("ctx: %s\n", gficf_last_error()); return 1; }

  const char* mode_name[3] = {"windowed+permuted", "windowed in order", "uniform random"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<uint32_t> ids = make_ids(N, k, mode);
    std::vector<uint32_t> h128((size_t)N * 32, 0);
    std::vector<uint16_t> h64((size_t)N * 32, 0);
    std::vector<int32_t> cm((size_t)E);
    for (long i = 0; i < N; ++i) {
      uint32_t hi = 0;
      for (int j = 0; j < k; ++j) {
        const uint32_t v = ids[(size_t)i * k + j];
        h128[(size_t)i * 32 + j] = v;
        h64[(size_t)i * 32 + j] = (uint16_t)(v & 0xffffu);
        hi |= ((v >> 16) & 1u) << j;
        cm[(size_t)j * N + i] = (int32_t)v;
      }
      h64[(size_t)i * 32 + 30] = (uint16_t)(hi & 0xffffu);
      h64[(size_t)i * 32 + 31] = (uint16_t)(hi >> 16);
    }
    CK(hipMemcpy(t128, h128.data(), N * 128, hipMemcpyHostToDevice));
    CK(hipMemcpy(t64, h64.data(), N * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx_cm, cm.data(), E * 4, hipMemcpyHostToDevice));
    printf("---- ids: %s\n", mode_name[mode]);
    auto run = [&](const char* name, auto kern, const char* table, int grid, uint32_t fold = 0xffffffffu) {
      float best = 1e9, sum = 0;
      const int reps = 20;
      for (int rep = 0; rep < reps + 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, table, N, k, s, d, w, fold);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 3) { best = std::min(best, ms); sum += ms; }
      }
      printf("  %-44s grid %5d: best %6.1f us  mean %6.1f us  %5.1f G edges/s\n", name, grid, best * 1e3, sum / reps * 1e3, E / (best * 1e-3) / 1e9);
    };
    for (int bpc : {3, 6, 8}) {
      if (product_only) break;
      const int grid = cus * bpc;
      run("128 B rows, plain loads, stores", k_model<128, 0, 1>, t128, grid);
      run("128 B rows, nt loads, stores", k_model<128, 1, 1>, t128, grid);
      run("128 B rows, plain loads, no stores", k_model<128, 0, 0>, t128, grid);
      run(" 64 B rows, plain loads, stores", k_model<64, 0, 1>, t64, grid);
      run(" 64 B rows, nt loads, stores", k_model<64, 1, 1>, t64, grid);
      run(" 64 B rows, plain loads, no stores", k_model<64, 0, 0>, t64, grid);
    }
    if (!product_only) {
      for (int bpc : {8, 16}) {
        const int grid = cus * bpc;
        run(" 64 B rows, XCD split 2, stores", k_model_xcd<2, 1>, t64, grid);
        run(" 64 B rows, XCD split 4, stores", k_model_xcd<4, 1>, t64, grid);
        run(" 64 B rows, XCD split 8, stores", k_model_xcd<8, 1>, t64, grid);
        run(" 64 B rows, XCD split 2, no stores", k_model_xcd<2, 0>, t64, grid);
        run(" 64 B rows, XCD split 8, no stores", k_model_xcd<8, 0>, t64, grid);
      }
    }
    if (!product_only) {
      const int grid = cus * 8;
      if (k == 30 && N % 4 == 0) {
        run(" 64 B rows, QUAD stores (60 lanes x 16 B)", k_model_quad, t64, grid);
        run(" 64 B rows, QUAD stores, grid x 6/8", k_model_quad, t64, cus * 6);
        run(" 64 B rows, QUAD stores, folded into 2 MB", k_model_quad, t64, grid, 0x7fffu);
      }
      run(" 64 B rows, plain loads, PLAIN stores", k_model<64, 0, 2>, t64, grid);
      run("128 B rows, plain loads, PLAIN stores", k_model<128, 0, 2>, t128, grid);
      run(" 64 B rows, gathers folded into 4 MB", k_model<64, 0, 1>, t64, grid, 0xffffu);
      run(" 64 B rows, gathers folded into 2 MB", k_model<64, 0, 1>, t64, grid, 0x7fffu);
      run(" 64 B rows, gathers folded into 0.5 MB", k_model<64, 0, 1>, t64, grid, 0x1fffu);
      run(" 64 B rows, folded into 2 MB, no stores", k_model<64, 0, 0>, t64, grid, 0x7fffu);
      run(" 64 B rows, folded into 0.5 MB, no stores", k_model<64, 0, 0>, t64, grid, 0x1fffu);
      run(" 64 B rows, folded into 16 KB, no stores", k_model<64, 0, 0>, t64, grid, 0xffu);
    }

    if (!product_only) {
      uint8_t* plane;
      CK(hipMalloc(&plane, N * 32));
      for (int P : {1, 2, 3, 4}) {
        const int grid = cus * 8;
        float best = 1e9, sum = 0;
        const int reps = 20;
        for (int rep = 0; rep < reps + 3; ++rep) {
          CK(hipEventRecord(e0));
          for (int ps = 0; ps < P; ++ps) {
            const uint32_t lo = (uint32_t)(N * ps / P), hi = (uint32_t)(N * (ps + 1) / P);
            if (P == 1) hipLaunchKernelGGL((k_model_range<true, true>), dim3(grid), dim3(256), 0, 0, t64, N, k, s, d, w, plane, lo, hi);
            else if (ps == 0) hipLaunchKernelGGL((k_model_range<true, false>), dim3(grid), dim3(256), 0, 0, t64, N, k, s, d, w, plane, lo, hi);
            else if (ps == P - 1) hipLaunchKernelGGL((k_model_range<false, true>), dim3(grid), dim3(256), 0, 0, t64, N, k, s, d, w, plane, lo, hi);
            else hipLaunchKernelGGL((k_model_range<false, false>), dim3(grid), dim3(256), 0, 0, t64, N, k, s, d, w, plane, lo, hi);
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 3) { best = std::min(best, ms); sum += ms; }
        }
        printf("   64 B rows, %d range pass(es), all launches   grid %5d: best %6.1f us  mean %6.1f us  %5.1f G edges/s\n", P, grid, best * 1e3, sum / reps * 1e3,
               E / (best * 1e-3) / 1e9);
      }
      CK(hipFree(plane));
    }
    if (!product_only) run("128 B rows, sc1 loads (waited one by one)", k_model<128, 2, 1>, t128, cus * 6);
    // the product kernel on the same ids
    if (gficf_jaccard_ingest_device(ctx, d_idx_cm, 0, N, k, N, N, d_table) != 0) { printf("ingest: %s\n", gficf_last_error()); return 1; }
    float best = 1e9, sum = 0, besti = 1e9;
    for (int rep = 0; rep < 23; ++rep) {
      CK(hipEventRecord(e0));
      gficf_jaccard_edges_device(ctx, d_table, N, k, 0, N, s, d, w, nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3) { best = std::min(best, ms); sum += ms; }
      CK(hipEventRecord(e0));
      gficf_jaccard_ingest_device(ctx, d_idx_cm, 0, N, k, N, N, d_table);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3) besti = std::min(besti, ms);
    }
    if (gficf_ctx_sync(ctx) != 0) { printf("sync: %s\n", gficf_last_error()); return 1; }
    {   // the same two kernels, 20 launches each between one pair of events (launch gaps amortised)
      float ms_i, ms_e;
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 20; ++rep) gficf_jaccard_ingest_device(ctx, d_idx_cm, 0, N, k, N, N, d_table);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_i, e0, e1));
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 20; ++rep) gficf_jaccard_edges_device(ctx, d_table, N, k, 0, N, s, d, w, nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_e, e0, e1));
      printf("  %-44s            : ingest %5.1f us  edges %5.1f us per launch\n", "PRODUCT, 20 launches back to back", ms_i / 20 * 1e3, ms_e / 20 * 1e3);
    }
    printf("  %-44s            : best %6.1f us  mean %6.1f us  %5.1f G edges/s   (ingest best %5.1f us)\n", "PRODUCT k_jaccard_edges", best * 1e3, sum / 20 * 1e3,
           E / (best * 1e-3) / 1e9, besti * 1e3);
  }
  gficf_ctx_destroy(ctx);
  return 0;
}
