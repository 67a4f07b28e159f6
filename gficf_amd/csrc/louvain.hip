// louvain.hip — community detection on the Jaccard graph ("next" row N4 of the scope table).
//
// What clustcells() runs on the adjacency matrix of the kNN -> Jaccard graph:
//   RunModularityClustering(igraph::as_adjacency_matrix(g, attr = "weight", sparse = T), 1, resolution, 1 | 2, n.start,
//                           n.iter, seed, verbose)                                         (reference R/clustCells.R:80,86)
// -> RunModularityClusteringCpp (src/RModularityOptimizer.cpp:25-181) -> VOSClusteringTechnique::runLouvainAlgorithm
// (src/ModularityOptimizer.cpp:594-612): local moving (:484-592), network reduction, recursion; quality function
// calcQualityFunction (:461-482) = standard modularity with a resolution parameter,
//   Q = (1 / 2W) * [ sum_ij A_ij delta(c_i, c_j)  -  resolution * sum_c K_c^2 / 2W ],   K_c = sum of the degrees in c,
// over the strict lower triangle of the matrix (the driver skips the diagonal, src/RModularityOptimizer.cpp:73-75).
//
// RELAXED PARITY CONTRACT.  The reference is a sequential, seed-exact algorithm: vertices move one at
// a time in the order of a java.util.Random permutation.  A device form cannot follow that order and stay parallel, so
// this is NOT label-for-label parity: the contract is the same objective (the Q above, same resolution semantics, same
// diagonal handling, clusters numbered by decreasing size as Clustering::orderClustersByNNodes :132-158 does), a
// modularity within a stated tolerance of the reference's own result on the same graph (tests/test_louvain_gpu.py
// compares against a build of the reference's ModularityOptimizer.cpp, oracle/_ref/), and bit-reproducible output.
//
// Device algorithm (deterministic parallel Louvain):
//   * weights in 2^-32 fixed point (u64): every sum — a vertex's weight towards a community, the community totals —
//     is an integer sum, so atomics commute and the result does not depend on scheduling;
//   * local moving, synchronous within a sub-round: every vertex reads the same snapshot (labels, community totals and
//     sizes), accumulates its edge weight per neighbouring community in an LDS hash table (one wave per vertex up to 128
//     neighbours, one workgroup with an 8192-slot table beyond), takes the community with the best gain
//       gain(v -> c) = w(v, c) - k_v * K_c(without v) * resolution / 2W         (reference :540, ties: smaller id :541)
//     if that beats staying; two singletons never swap (only the larger id moves).  The vertices are split into S
//     hash classes that move one after the other (S grows as the graph gets small, where simultaneous moves hurt most);
//     totals are applied between sub-rounds.  An iteration that lowers Q is undone and ends the level;
//   * reduction: communities renumbered by a scan, every inter-community edge keyed (c_u << 32 | c_v), one rocPRIM
//     radix sort, equal keys summed (integers again), CSR rebuilt; repeat until nothing merges;
//   * n_iter > 1 restarts from the finest graph with the labels found so far, as the reference's iterations do;
//   * algorithm 2 (runLouvainAlgorithmWithMultilevelRefinement, :629-649): after the descent, back up through the levels
//     with one more local moving on each, seeded with the labels found below it; a level's graph is rebuilt from the
//     finest one through its saved vertex map (one sort) rather than kept.
//   * a vertex whose every option has a negative gain leaves for an unused cluster (:546-550): its own id, if free.
//   * n_start "random starts": each start varies the seed of the class hash (the only arbitrary choice there is); the best
//     modularity wins, as in the reference.  Start 0 with seed 0 is the plain run.
//   * the alternative modularity function (2): unit node weights, the resolution as given — a context option.
// Not reproduced: algorithm 3 (SLM).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <vector>

#include <atomic>

#include "common.h"

namespace {

typedef unsigned long long u64;

constexpr double LV_SCALE = 4294967296.0;      // 2^32
constexpr int LV_SMALL_DEG = 128;              // up to here: one wave per vertex, 256-slot table
constexpr int LV_SMALL_SLOTS = 256;
constexpr int LV_MID_DEG = 1024, LV_MID_SLOTS = 2048;   // one workgroup per vertex, 24 KB table
constexpr int LV_BIG_SLOTS = 8192;             // beyond: 32 KB keys + 64 KB sums
constexpr int LV_MAX_ITERS = 64;
constexpr int LV_MAX_SAVED = 12;              // levels whose vertex map is kept for the refinement of algorithm 2

struct LvGraph {
  int64_t n, m;
  const int64_t* ptr;
  const int32_t* nbr;
  const u64* wt;
  const u64* kv;
};

__device__ __host__ static inline uint32_t lv_hash(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}

// ---- level 0: fixed-point weights, validation
__global__ __launch_bounds__(256) void k_lv_fix(int64_t N, int64_t nnz, const int32_t* __restrict__ nbr, const double* __restrict__ x,
                                                u64* __restrict__ wt, u64* __restrict__ max_wt, uint32_t* __restrict__ status) {
  __shared__ u64 s_max[4];
  u64 mx = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
    const double v = x[e];
    const int32_t u = nbr[e];
    const bool ok = u >= 0 && u < N && v >= 0.0 && v <= 1048576.0;      // NaN fails the comparisons
    if (!ok) atomicOr(status, u >= 0 && u < N ? GFICF_ST_BAD_VALUE : GFICF_ST_BAD_CSC);
    const u64 f = ok ? (u64)llrint(v * LV_SCALE) : 0ull;
    wt[e] = f;
    mx = f > mx ? f : mx;
  }
  for (int d = 32; d > 0; d >>= 1) { const u64 o = __shfl_down(mx, d); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {                                        // the largest weight bounds every later sum (checked by the host)
    for (int t = 1; t < 4; ++t) mx = s_max[t] > mx ? s_max[t] : mx;
    if (mx) atomicMax(max_wt, mx);
  }
}

__global__ __launch_bounds__(256) void k_lv_vertex_weight(LvGraph g, u64* __restrict__ kv, u64* __restrict__ two_w,
                                                         uint32_t* __restrict__ status) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  u64 s = 0;
  if (v < g.n) {
    int64_t lo = g.ptr[v], hi = g.ptr[v + 1];
    if (lo < 0 || hi < lo || hi > g.m) { atomicOr(status, GFICF_ST_BAD_CSC); lo = hi = 0; }
    for (int64_t e = lo; e < hi; ++e) {
      const int32_t u = g.nbr[e];
      if (u != v && u >= 0 && u < g.n) s += g.wt[e];
    }
    kv[v] = s;
  }
  // block sum -> one atomic
  __shared__ u64 s_sum[4];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(two_w, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

__global__ __launch_bounds__(256) void k_lv_fill_u64(int64_t n, u64 v, u64* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

// singletons: every vertex its own community
__global__ __launch_bounds__(256) void k_lv_init(int64_t n, const u64* __restrict__ kv, int32_t* __restrict__ comm, u64* __restrict__ K,
                                                 int32_t* __restrict__ size) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) { comm[v] = (int32_t)v; K[v] = kv[v]; size[v] = 1; }
}

// Per-label totals (K, may be NULL) and member counts of given labels in [0, C).  With few labels every vertex would
// hit the same few addresses: up to LV_ACC_BINS labels are binned in LDS first, one global atomic per label and block.
constexpr int LV_ACC_BINS = 4096;
__global__ __launch_bounds__(256) void k_lv_accum(int64_t n, int64_t C, const int32_t* __restrict__ lab, const u64* __restrict__ kv,
                                                  int32_t* __restrict__ comm, u64* __restrict__ K, int32_t* __restrict__ size) {
  __shared__ u64 s_k[LV_ACC_BINS];
  __shared__ int32_t s_n[LV_ACC_BINS];
  const bool binned = C <= LV_ACC_BINS;
  if (binned) {
    for (int t = threadIdx.x; t < C; t += 256) { s_k[t] = 0ull; s_n[t] = 0; }
    __syncthreads();
  }
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
    const int32_t c = lab[v];
    if (comm) comm[v] = c;
    if (binned) {
      if (K) atomicAdd(&s_k[c], kv[v]);
      atomicAdd(&s_n[c], 1);
    } else {
      if (K) atomicAdd(&K[c], kv[v]);
      atomicAdd(&size[c], 1);
    }
  }
  if (binned) {
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += 256) {
      if (s_n[t]) atomicAdd(&size[t], s_n[t]);
      if (K && s_k[t]) atomicAdd(&K[t], s_k[t]);
    }
  }
}

// ---- local moving
struct LvMove {
  double r;            // resolution / 2W (in fixed-point units of 2W)
  int s, S;            // this sub-round's hash class
  uint32_t seed;       // of the class hash: what a "random start" varies
};

__device__ static inline void lv_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ static inline bool lv_better(double g, int32_t c, double bg, int32_t bc) { return g > bg || (g == bg && c < bc); }

// One wave per vertex (degree <= LV_SMALL_DEG).
__global__ __launch_bounds__(256) void k_lv_move_small(LvGraph g, LvMove mv, const int32_t* __restrict__ comm, const u64* __restrict__ K,
                                                       const int32_t* __restrict__ size, int32_t* __restrict__ next,
                                                       unsigned* __restrict__ moved) {
  __shared__ int32_t s_key[4][LV_SMALL_SLOTS];
  __shared__ u64 s_val[4][LV_SMALL_SLOTS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + wave;
  int64_t lo = 0, hi = 0;
  bool active = v < g.n;
  if (active) {
    lo = g.ptr[v]; hi = g.ptr[v + 1];
    active = hi - lo <= LV_SMALL_DEG && (mv.S == 1 || (int)(lv_hash((uint32_t)v + mv.seed) % (uint32_t)mv.S) == mv.s);
  }
  int32_t* key = s_key[wave];
  u64* val = s_val[wave];
  // the table belongs to this wave alone and LDS serves a wave's operations in issue order: a wave-level fence (no
  // workgroup barrier) is all that separates clearing, filling and reading it
  if (!active) return;
  for (int t = lane; t < LV_SMALL_SLOTS; t += 64) { key[t] = -1; val[t] = 0ull; }
  lv_wave_sync();
  {
    for (int64_t e = lo + lane; e < hi; e += 64) {
      const int32_t u = g.nbr[e];
      if (u == v) continue;
      const int32_t c = comm[u];
      uint32_t h = lv_hash((uint32_t)c) & (LV_SMALL_SLOTS - 1);
      for (;;) {
        const int32_t old = atomicCAS(&key[h], -1, c);
        if (old == -1 || old == c) { atomicAdd(&val[h], g.wt[e]); break; }
        h = (h + 1) & (LV_SMALL_SLOTS - 1);
      }
    }
  }
  lv_wave_sync();
  const int32_t cv = comm[v];
  const double kvd = (double)g.kv[v];
  double bg = -INFINITY, stay_w = 0.0;
  int32_t bc = INT32_MAX;
  for (int t = lane; t < LV_SMALL_SLOTS; t += 64) {
    const int32_t c = key[t];
    if (c < 0) continue;
    const double w = (double)val[t];
    if (c == cv) { stay_w = w; continue; }
    const double gain = w - kvd * (double)K[c] * mv.r;
    if (lv_better(gain, c, bg, bc)) { bg = gain; bc = c; }
  }
  for (int d = 32; d > 0; d >>= 1) {
    const double og = __shfl_xor(bg, d), ow = __shfl_xor(stay_w, d);
    const int32_t oc = __shfl_xor(bc, d);
    if (lv_better(og, oc, bg, bc)) { bg = og; bc = oc; }
    stay_w = fmax(stay_w, ow);
  }
  if (lane == 0) {
    const double g_stay = stay_w - kvd * (double)(K[cv] - g.kv[v]) * mv.r;
    bool move = bc != INT32_MAX && bg > g_stay;
    if (move && size[cv] == 1 && size[bc] == 1 && bc > cv) move = false;      // two singletons never swap
    int32_t to = move ? bc : cv;
    // every option loses: alone is better (the reference's move into an unused cluster, :546-550).  The unused cluster
    // is the vertex's own id when nobody holds it — unique per vertex, so simultaneous escapes never meet.
    if ((move ? bg : g_stay) < 0.0 && size[cv] > 1 && size[v] == 0) { to = (int32_t)v; move = true; }
    next[v] = to;
    if (move) atomicAdd(moved, 1u);
  }
}

// One workgroup per listed vertex (degree > LV_SMALL_DEG): SLOTS = 2048 up to LV_MID_DEG neighbours, 8192 beyond.
template <int SLOTS>
__global__ __launch_bounds__(256) void k_lv_move_big(LvGraph g, LvMove mv, const int32_t* __restrict__ big, const int32_t* __restrict__ comm,
                                                     const u64* __restrict__ K, const int32_t* __restrict__ size,
                                                     int32_t* __restrict__ next, unsigned* __restrict__ moved,
                                                     uint32_t* __restrict__ status) {
  extern __shared__ unsigned char s_raw[];
  u64* val = (u64*)s_raw;
  int32_t* key = (int32_t*)(val + SLOTS);
  __shared__ double s_g[256], s_w[4];
  __shared__ int32_t s_c[256];
  const int tid = threadIdx.x;
  const int64_t v = big[blockIdx.x];
  if (mv.S != 1 && (int)(lv_hash((uint32_t)v + mv.seed) % (uint32_t)mv.S) != mv.s) return;     // uniform per workgroup
  const int64_t lo = g.ptr[v], hi = g.ptr[v + 1];
  const int32_t cv = comm[v];
  const double kvd = (double)g.kv[v];
  double bg = -INFINITY, stay_w = 0.0;
  int32_t bc = INT32_MAX;
  // A vertex with more neighbours than the table comfortably holds (every neighbour can be its own community) is done in
  // P passes over its edges, pass p taking the communities of hash class p: about deg / P <= SLOTS / 2 of them at a time.
  const uint32_t P = (uint32_t)((hi - lo + SLOTS / 2 - 1) / (SLOTS / 2));
  for (uint32_t p = 0; p < (P ? P : 1u); ++p) {
    for (int t = tid; t < SLOTS; t += 256) { key[t] = -1; val[t] = 0ull; }
    __syncthreads();
    for (int64_t e = lo + tid; e < hi; e += 256) {
      const int32_t u = g.nbr[e];
      if (u == v) continue;
      const int32_t c = comm[u];
      const uint32_t hc = lv_hash((uint32_t)c);
      if (P > 1 && (hc >> 13) % P != p) continue;
      uint32_t h = hc & (SLOTS - 1);
      int probes = 0;
      for (;;) {
        const int32_t old = atomicCAS(&key[h], -1, c);
        if (old == -1 || old == c) { atomicAdd(&val[h], g.wt[e]); break; }
        h = (h + 1) & (SLOTS - 1);
        if (++probes >= SLOTS) { atomicOr(status, GFICF_ST_TOO_DENSE); break; }   // a hash class that overflows the table
      }
    }
    __syncthreads();
    for (int t = tid; t < SLOTS; t += 256) {
      const int32_t c = key[t];
      if (c < 0) continue;
      const double w = (double)val[t];
      if (c == cv) { stay_w = w; continue; }
      const double gain = w - kvd * (double)K[c] * mv.r;
      if (lv_better(gain, c, bg, bc)) { bg = gain; bc = c; }
    }
    __syncthreads();
  }
  for (int d = 32; d > 0; d >>= 1) stay_w = fmax(stay_w, __shfl_xor(stay_w, d));
  if ((tid & 63) == 0) s_w[tid >> 6] = stay_w;
  s_g[tid] = bg; s_c[tid] = bc;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (tid < d && lv_better(s_g[tid + d], s_c[tid + d], s_g[tid], s_c[tid])) { s_g[tid] = s_g[tid + d]; s_c[tid] = s_c[tid + d]; }
    __syncthreads();
  }
  if (tid == 0) {
    bg = s_g[0]; bc = s_c[0];
    stay_w = fmax(fmax(s_w[0], s_w[1]), fmax(s_w[2], s_w[3]));
    const double g_stay = stay_w - kvd * (double)(K[cv] - g.kv[v]) * mv.r;
    bool move = bc != INT32_MAX && bg > g_stay;
    if (move && size[cv] == 1 && size[bc] == 1 && bc > cv) move = false;
    int32_t to = move ? bc : cv;
    if ((move ? bg : g_stay) < 0.0 && size[cv] > 1 && size[v] == 0) { to = (int32_t)v; move = true; }     // see k_lv_move_small
    next[v] = to;
    if (move) atomicAdd(moved, 1u);
  }
}

// the vertices of the workgroup path: middle degrees from the front of the list, large ones from its end
__global__ __launch_bounds__(256) void k_lv_list_big(LvGraph g, int32_t* __restrict__ big, unsigned* __restrict__ n_mid_large) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= g.n) return;
  const int64_t deg = g.ptr[v + 1] - g.ptr[v];
  if (deg > LV_MID_DEG) big[g.n - 1 - atomicAdd(n_mid_large + 1, 1u)] = (int32_t)v;
  else if (deg > LV_SMALL_DEG) big[atomicAdd(n_mid_large, 1u)] = (int32_t)v;
}

// applies the sub-round's moves to the labels, totals and sizes
__global__ __launch_bounds__(256) void k_lv_apply(int64_t n, LvMove mv, const u64* __restrict__ kv, int32_t* __restrict__ comm,
                                                  const int32_t* __restrict__ next, u64* __restrict__ K, int32_t* __restrict__ size) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  if (mv.S != 1 && (int)(lv_hash((uint32_t)v + mv.seed) % (uint32_t)mv.S) != mv.s) return;
  const int32_t a = comm[v], b = next[v];
  if (a == b) return;
  const u64 k = kv[v];
  atomicAdd(&K[a], 0ull - k);
  atomicAdd(&K[b], k);
  atomicSub(&size[a], 1);
  atomicAdd(&size[b], 1);
  comm[v] = b;
}

// ---- quality: internal weight (integer) and sum of squared totals (fixed summation order)
__global__ __launch_bounds__(256) void k_lv_internal(LvGraph g, const int32_t* __restrict__ comm, u64* __restrict__ in_w) {
  __shared__ u64 s_sum[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u64 s = 0;
  for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < g.n; v += (int64_t)gridDim.x * 4) {
    const int32_t cv = comm[v];
    for (int64_t e = g.ptr[v] + lane; e < g.ptr[v + 1]; e += 64) {
      const int32_t u = g.nbr[e];
      if (u != v && comm[u] == cv) s += g.wt[e];
    }
  }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
  if (lane == 0) s_sum[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0 && (s_sum[0] | s_sum[1] | s_sum[2] | s_sum[3])) atomicAdd(in_w, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// Sum of the squared community totals, in a fixed order: LV_SQ_BLOCKS slices of the communities, a fixed tree inside each,
// the slices added up by the host in slice order.
constexpr int LV_SQ_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_lv_sumsq(int64_t n, const u64* __restrict__ K, double* __restrict__ part) {
  __shared__ double s_p[256];
  const int64_t per = (n + LV_SQ_BLOCKS - 1) / LV_SQ_BLOCKS;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  for (int64_t c = lo + threadIdx.x; c < hi; c += 256) { const double k = (double)K[c]; s += k * k; }
  s_p[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_p[threadIdx.x] += s_p[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = s_p[0];
}

// ---- reduction of the graph
__global__ __launch_bounds__(256) void k_lv_used(int64_t n, const int32_t* __restrict__ size, int64_t* __restrict__ flag) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c <= n) flag[c] = c < n && size[c] > 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_lv_coarse_weights(int64_t n, const int32_t* __restrict__ size, const int64_t* __restrict__ newid,
                                                           const u64* __restrict__ K, u64* __restrict__ kv2) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < n && size[c] > 0) kv2[newid[c]] = K[c];
}

// lab[v] = new id of the community of the level's vertex that original vertex v maps to (src == NULL: v itself)
__global__ __launch_bounds__(256) void k_lv_relabel(int64_t n, const int32_t* src, const int32_t* __restrict__ comm,
                                                    const int64_t* __restrict__ newid, int32_t* lab) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) lab[v] = (int32_t)newid[comm[src ? src[v] : v]];        // src may be lab itself: one read, one write per thread
}

__global__ __launch_bounds__(256) void k_lv_iota(int64_t n, int64_t* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) out[v] = v;
}

// seed[x] = the label of the original vertices that make up level vertex x (they all carry the same one)
__global__ __launch_bounds__(256) void k_lv_seed(int64_t n, const int32_t* __restrict__ top, const int32_t* __restrict__ lab,
                                                 int32_t* __restrict__ seed) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) seed[top[v]] = lab[v];
}

// none = n2 << 32: the key of an entry that stays inside a community; it sorts behind every kept key
__global__ __launch_bounds__(256) void k_lv_emit(LvGraph g, const int32_t* __restrict__ comm, const int64_t* __restrict__ newid, u64 none,
                                                 u64* __restrict__ keys, u64* __restrict__ vals) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + wave;
  if (v >= g.n) return;
  const u64 cv = (u64)newid[comm[v]];
  for (int64_t e = g.ptr[v] + lane; e < g.ptr[v + 1]; e += 64) {
    const int32_t u = g.nbr[e];
    u64 k = none;
    if (u != v) {
      const u64 cu = (u64)newid[comm[u]];
      if (cu != cv) k = (cv << 32) | cu;
    }
    keys[e] = k;
    vals[e] = g.wt[e];
  }
}

// flag[e] = 1 where a new (row, col) starts; flag[m] = 0 (the scan turns it into the number of coarse entries)
__global__ __launch_bounds__(256) void k_lv_heads(const u64* __restrict__ keys, int64_t m, u64 none, int64_t* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e > m) return;
  int64_t f = 0;
  if (e < m) {
    const u64 k = keys[e];
    f = (k != none && (e == 0 || keys[e - 1] != k)) ? 1 : 0;
  }
  flag[e] = f;
}

// pos = exclusive scan of the head flags: entry e belongs to coarse entry pos[e + 1] - 1.  A workgroup takes LV_RED_CHUNK
// consecutive sorted entries, sums them per coarse entry in LDS (their positions span less than the chunk) and adds
// each sum once: a graph reduced to a handful of communities would otherwise put millions of atomics on a few addresses.
constexpr int LV_RED_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_lv_reduce(const u64* __restrict__ keys, const u64* __restrict__ vals, int64_t m, u64 none,
                                                   const int64_t* __restrict__ pos, int32_t* __restrict__ nbr2, u64* __restrict__ wt2,
                                                   int64_t* __restrict__ row_cnt) {
  __shared__ u64 s_w[LV_RED_CHUNK];
  const int64_t e0 = (int64_t)blockIdx.x * LV_RED_CHUNK, e1 = e0 + LV_RED_CHUNK < m ? e0 + LV_RED_CHUNK : m;
  if (keys[e0] == none) return;                  // the dropped entries sort last: nothing kept in this chunk
  for (int t = threadIdx.x; t < LV_RED_CHUNK; t += 256) s_w[t] = 0ull;
  __syncthreads();
  const int64_t p_lo = pos[e0 + 1] - 1;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
    const u64 k = keys[e];
    if (k == none) break;
    const int64_t p = pos[e + 1] - 1;
    atomicAdd(&s_w[p - p_lo], vals[e]);
    if (pos[e + 1] != pos[e]) {                   // a head
      nbr2[p] = (int32_t)(k & 0xffffffffull);
      atomicAdd((u64*)&row_cnt[k >> 32], 1ull);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < LV_RED_CHUNK; t += 256)
    if (s_w[t]) atomicAdd(&wt2[p_lo + t], s_w[t]);
}

// ---- final numbering: clusters by decreasing size, ties by id (Clustering::orderClustersByNNodes, reference :132-158)
__global__ __launch_bounds__(256) void k_lv_size_keys(int64_t C, int64_t n, const int32_t* __restrict__ cnt, u64* __restrict__ keys,
                                                      u64* __restrict__ ids) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < C) { keys[c] = ((u64)(n - cnt[c]) << 32) | (u64)c; ids[c] = (u64)c; }
}

__global__ __launch_bounds__(256) void k_lv_rank(int64_t C, const u64* __restrict__ sorted_ids, int32_t* __restrict__ rank) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < C) rank[sorted_ids[p]] = (int32_t)p;
}

__global__ __launch_bounds__(256) void k_lv_final(int64_t n, const int32_t* __restrict__ lab, const int32_t* __restrict__ rank,
                                                  int32_t* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) out[v] = rank[lab[v]];
}

// ---- host side
struct Bump {
  char* base; size_t off, cap;
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;     // base == NULL: sizing only
    off += count * sizeof(T);
    return p;
  }
};

static size_t lv_sort_tmp_bytes(int64_t m) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (size_t)(m > 0 ? m : 1), 0u, 64u,
                                  (hipStream_t) nullptr);
  return tmp;
}

struct LvLevel {          // a coarse graph's arrays
  int64_t* ptr; int32_t* nbr; u64* wt; u64* kv;
};

struct LvWs {
  u64* wt0; u64* kv0;
  LvLevel lvl[2];
  int32_t *comm, *next, *snap_comm, *size, *snap_size, *big, *lab, *cnt, *rank, *tops, *best;
  u64 *K, *snap_K;
  int64_t* flag;            // max(n, m) + 1 entries: scans
  u64 *keys_a, *vals_a, *keys_b, *vals_b;
  void* sort_tmp; size_t sort_tmp_bytes;
  double* sq_part;          // LV_SQ_BLOCKS partial sums of squared community totals
  u64* scalars;             // [0] 2W, [1] internal weight, [2] moved (unsigned), [3] unused, [4] n_mid | n_large, [5] largest weight
};

static size_t lv_carve(LvWs* w, void* base, int64_t N, int64_t nnz) {
  Bump b{(char*)base, 0, 0};
  const size_t n = (size_t)(N > 0 ? N : 1), m = (size_t)(nnz > 0 ? nnz : 1);
  LvWs d;
  d.wt0 = b.take<u64>(m); d.kv0 = b.take<u64>(n);
  for (int i = 0; i < 2; ++i) {
    d.lvl[i].ptr = b.take<int64_t>(n + 1); d.lvl[i].nbr = b.take<int32_t>(m); d.lvl[i].wt = b.take<u64>(m); d.lvl[i].kv = b.take<u64>(n);
  }
  d.comm = b.take<int32_t>(n); d.next = b.take<int32_t>(n); d.snap_comm = b.take<int32_t>(n);
  d.size = b.take<int32_t>(n); d.snap_size = b.take<int32_t>(n); d.big = b.take<int32_t>(n);
  d.lab = b.take<int32_t>(n); d.cnt = b.take<int32_t>(n); d.rank = b.take<int32_t>(n);
  d.tops = b.take<int32_t>(n * LV_MAX_SAVED);
  d.best = b.take<int32_t>(n);
  d.K = b.take<u64>(n); d.snap_K = b.take<u64>(n);
  d.flag = b.take<int64_t>((n > m ? n : m) + 1);
  d.keys_a = b.take<u64>(m > n ? m : n); d.vals_a = b.take<u64>(m > n ? m : n);
  d.keys_b = b.take<u64>(m > n ? m : n); d.vals_b = b.take<u64>(m > n ? m : n);
  d.sort_tmp_bytes = lv_sort_tmp_bytes((int64_t)(m > n ? m : n));
  d.sort_tmp = b.take<char>(d.sort_tmp_bytes);
  d.scalars = b.take<u64>(8);
  d.sq_part = b.take<double>(LV_SQ_BLOCKS);
  if (w) *w = d;
  return b.off + 256;
}

static inline unsigned lv_blocks(int64_t n, int per) { return (unsigned)gficf_ceil_div(n > 0 ? n : 1, per); }

static int lv_sub_rounds(int64_t n) {
  if (const char* e = getenv("GFICF_LOUVAIN_SUBROUNDS")) { const int s = atoi(e); if (s >= 1 && s <= 64) return s; }
  return n > 50000 ? 2 : n > 4000 ? 4 : n > 400 ? 8 : 16;
}

struct LvQ { double q; u64 in_w; };

// Q of the current labels on graph g (self = weight already folded into the vertices), deterministic.
static int lv_quality(gficf_ctx* ctx, const LvGraph& g, const LvWs& w, u64 self_w, double two_w, double q_coef, double* q_out, u64* in_out) {
  GFICF_HIP_CHECK(hipMemsetAsync(w.scalars + 1, 0, sizeof(u64), ctx->stream));
  hipLaunchKernelGGL(k_lv_internal, dim3(lv_blocks(g.n, 4) < 2048u ? lv_blocks(g.n, 4) : 2048u), dim3(256), 0, ctx->stream, g, w.comm, w.scalars + 1);
  hipLaunchKernelGGL(k_lv_sumsq, dim3(LV_SQ_BLOCKS), dim3(256), 0, ctx->stream, g.n, w.K, w.sq_part);
  u64 h[4];
  double hp[LV_SQ_BLOCKS];
  GFICF_HIP_CHECK(hipMemcpyAsync(h, w.scalars, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GFICF_HIP_CHECK(hipMemcpyAsync(hp, w.sq_part, sizeof(hp), hipMemcpyDeviceToHost, ctx->stream));
  GFICF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double sq = 0.0;
  for (int i = 0; i < LV_SQ_BLOCKS; ++i) sq += hp[i];
  *in_out = h[1];
  *q_out = ((double)(h[1] + self_w)) / two_w - q_coef * sq;          // sq = sum of squared community totals (fixed point)
  return GFICF_OK;
}

}  // namespace

extern "C" {

size_t gficf_louvain_workspace_bytes(int64_t N, int64_t nnz) {
  if (N < 0 || nnz < 0) return 0;
  return lv_carve(nullptr, nullptr, N, nnz);
}

int gficf_louvain_device(gficf_ctx* ctx, int64_t N, const int64_t* d_indptr, const int32_t* d_indices, const double* d_x, int64_t nnz,
                         double resolution, int algorithm, int n_start, int n_iter, int seed, int32_t* d_labels, int64_t* n_clusters, double* modularity, void* d_ws,
                         size_t ws_bytes) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0 || nnz < 0 || n_iter < 1 || n_start < 1 || !(resolution >= 0.0))
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size, n_start < 1, n_iter < 1 or a negative resolution");
  if (algorithm != 1 && algorithm != 2) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "algorithm must be 1 (Louvain) or 2 (Louvain with multilevel refinement)");
  if (!n_clusters) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_clusters is NULL");
  *n_clusters = 0;
  if (modularity) *modularity = 0.0;
  if (N == 0) return GFICF_OK;
  if (!d_indptr || !d_labels || !d_ws || (nnz > 0 && (!d_indices || !d_x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (N > INT32_MAX) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "more than 2^31 - 1 vertices");
  if (ws_bytes < gficf_louvain_workspace_bytes(N, nnz))
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "workspace too small: %zu < %zu bytes", ws_bytes, gficf_louvain_workspace_bytes(N, nnz));
  static std::atomic<bool> attr_set[64];                 // per device: the attribute belongs to the device's copy of the kernel
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_lv_move_big<LV_BIG_SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_BIG_SLOTS * 12));
    attr_set[ctx->device & 63] = true;
  }
  LvWs w;
  lv_carve(&w, d_ws, N, nnz);
  hipStream_t st = ctx->stream;

  // level 0: fixed-point weights, vertex weights, 2W
  GFICF_HIP_CHECK(hipMemsetAsync(w.scalars, 0, 8 * sizeof(u64), st));
  if (nnz > 0) hipLaunchKernelGGL(k_lv_fix, dim3(lv_blocks(nnz, 256) < 2048u ? lv_blocks(nnz, 256) : 2048u), dim3(256), 0, st, N, nnz, d_indices, d_x, w.wt0, w.scalars + 5, ctx->d_status);
  LvGraph g0{N, nnz, d_indptr, d_indices, w.wt0, w.kv0};
  hipLaunchKernelGGL(k_lv_vertex_weight, dim3(lv_blocks(N, 256)), dim3(256), 0, st, g0, w.kv0, w.scalars, ctx->d_status);
  u64 h_sc[6] = {0, 0, 0, 0, 0, 0};
  GFICF_HIP_CHECK(hipMemcpyAsync(h_sc, w.scalars, sizeof(h_sc), hipMemcpyDeviceToHost, st));
  int rc = gficf_ctx_sync(ctx);                    // also reports a malformed matrix before anything follows it
  if (rc) return rc;
  const u64 two_w_fix = h_sc[0];
  if ((double)h_sc[5] * (double)nnz >= 9.0e18)     // the u64 sums (2W, community totals) could wrap
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "edge weights too large for the 2^-32 fixed-point sums (largest weight x entries >= 2^31): scale the matrix");
  const double two_w = (double)two_w_fix;
  if (two_w_fix == 0) {                            // no edges: every vertex is its own cluster, Q = 0
    hipLaunchKernelGGL(k_lv_init, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, w.kv0, d_labels, w.K, w.size);
    GFICF_HIP_CHECK(hipStreamSynchronize(st));
    *n_clusters = N;
    return GFICF_OK;
  }
  // standard modularity: node weight = degree, gain coefficient resolution / 2W, Q's second term resolution * sum K^2 / (2W)^2;
  // alternative (modularity function 2): node weight = 1 (2^32 in fixed point), coefficient = the resolution itself
  const bool alt = ctx->lv_modularity_fn == 2;
  if (alt && resolution > 1.0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "error: resolution<1 for alternative modularity");
  if (alt) hipLaunchKernelGGL(k_lv_fill_u64, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, (u64)LV_SCALE, w.kv0);
  const double r = alt ? resolution / LV_SCALE : resolution / two_w;
  const double q_coef = alt ? resolution / (LV_SCALE * two_w) : resolution / (two_w * two_w);
  double q_final = 0.0;
  bool have_labels = false;
  int64_t n_labels = 0;                          // labels of the previous pass lie in [0, n_labels)

  // ---- the steps of one level (host side; every one ends synchronised or enqueues on the context's stream)
  double q_prev = 0.0;                           // Q of the labels the level holds
  u64 in_w = 0;                                  // their internal weight on the level's graph
  unsigned n_mid = 0, n_large = 0;
  uint32_t start_seed = 0;                       // of the sub-round class hash (what a "random start" varies)
  const auto grid_cap = [](unsigned b) { return b < 1024u ? b : 1024u; };
  const bool debug = getenv("GFICF_LOUVAIN_DEBUG") != nullptr;      // per-iteration trace on stderr
  // labels (seed == NULL: singletons), totals, sizes, the workgroup-path vertex lists, Q
  const auto start_level = [&](const LvGraph& g, const int32_t* seed, int64_t seed_labels, u64 self_w) -> int {
    if (seed) {
      GFICF_HIP_CHECK(hipMemsetAsync(w.K, 0, sizeof(u64) * (size_t)g.n, st));
      GFICF_HIP_CHECK(hipMemsetAsync(w.size, 0, sizeof(int32_t) * (size_t)g.n, st));
      hipLaunchKernelGGL(k_lv_accum, dim3(grid_cap(lv_blocks(g.n, 256))), dim3(256), 0, st, g.n, seed_labels, seed, g.kv, w.comm, w.K, w.size);
    } else {
      hipLaunchKernelGGL(k_lv_init, dim3(lv_blocks(g.n, 256)), dim3(256), 0, st, g.n, g.kv, w.comm, w.K, w.size);
    }
    GFICF_HIP_CHECK(hipMemsetAsync(w.scalars + 4, 0, sizeof(u64), st));
    hipLaunchKernelGGL(k_lv_list_big, dim3(lv_blocks(g.n, 256)), dim3(256), 0, st, g, w.big, (unsigned*)(w.scalars + 4));
    unsigned h_cnt[2] = {0, 0};                  // vertices of middle and of large degree
    GFICF_HIP_CHECK(hipMemcpyAsync(h_cnt, w.scalars + 4, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    const int rc2 = lv_quality(ctx, g, w, self_w, two_w, q_coef, &q_prev, &in_w);
    n_mid = h_cnt[0]; n_large = h_cnt[1];
    return rc2;
  };
  const auto local_moving = [&](const LvGraph& g, u64 self_w, bool* level_moved) -> int {
    const int S = lv_sub_rounds(g.n);
    *level_moved = false;
    for (int iter = 0; iter < LV_MAX_ITERS; ++iter) {
      GFICF_HIP_CHECK(hipMemcpyAsync(w.snap_comm, w.comm, sizeof(int32_t) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
      GFICF_HIP_CHECK(hipMemcpyAsync(w.snap_size, w.size, sizeof(int32_t) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
      GFICF_HIP_CHECK(hipMemcpyAsync(w.snap_K, w.K, sizeof(u64) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
      GFICF_HIP_CHECK(hipMemsetAsync(w.scalars + 2, 0, sizeof(unsigned), st));
      for (int s = 0; s < S; ++s) {
        const LvMove mv{r, s, S, start_seed};
        hipLaunchKernelGGL(k_lv_move_small, dim3(lv_blocks(g.n, 4)), dim3(256), 0, st, g, mv, w.comm, w.K, w.size, w.next, (unsigned*)(w.scalars + 2));
        if (n_mid)
          hipLaunchKernelGGL(k_lv_move_big<LV_MID_SLOTS>, dim3(n_mid), dim3(256), LV_MID_SLOTS * 12, st, g, mv, w.big, w.comm, w.K, w.size, w.next,
                             (unsigned*)(w.scalars + 2), ctx->d_status);
        if (n_large)
          hipLaunchKernelGGL(k_lv_move_big<LV_BIG_SLOTS>, dim3(n_large), dim3(256), LV_BIG_SLOTS * 12, st, g, mv, w.big + (g.n - n_large), w.comm,
                             w.K, w.size, w.next, (unsigned*)(w.scalars + 2), ctx->d_status);
        hipLaunchKernelGGL(k_lv_apply, dim3(lv_blocks(g.n, 256)), dim3(256), 0, st, g.n, mv, g.kv, w.comm, w.next, w.K, w.size);
      }
      unsigned moved = 0;
      GFICF_HIP_CHECK(hipMemcpyAsync(&moved, w.scalars + 2, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      double q; u64 in_now;
      const int rc2 = lv_quality(ctx, g, w, self_w, two_w, q_coef, &q, &in_now);
      if (rc2) return rc2;
      if (moved == 0) break;
      if (q < q_prev) {                          // simultaneous moves made it worse: undo the iteration, the level ends
        GFICF_HIP_CHECK(hipMemcpyAsync(w.comm, w.snap_comm, sizeof(int32_t) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
        GFICF_HIP_CHECK(hipMemcpyAsync(w.size, w.snap_size, sizeof(int32_t) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
        GFICF_HIP_CHECK(hipMemcpyAsync(w.K, w.snap_K, sizeof(u64) * (size_t)g.n, hipMemcpyDeviceToDevice, st));
        break;
      }
      *level_moved = true;
      in_w = in_now;
      const bool small_gain = q - q_prev < 1e-7;
      if (debug) fprintf(stderr, "[louvain]   n=%lld iter %d: moved %u, Q %.9f -> %.9f\n", (long long)g.n, iter, moved, q_prev, q);
      q_prev = q;
      if (small_gain) break;
    }
    return GFICF_OK;
  };
  // the communities in use numbered 0 .. n2-1 (w.flag = old id -> new id); lab[v] = new id of comm[src[v]] (src NULL: v)
  const auto renumber = [&](const LvGraph& g, const int32_t* src, int64_t* n2) -> int {
    hipLaunchKernelGGL(k_lv_used, dim3(lv_blocks(g.n + 1, 256)), dim3(256), 0, st, g.n, w.size, w.flag);
    const int rc2 = gficf_exclusive_scan_i64(ctx, w.flag, g.n + 1);
    if (rc2) return rc2;
    GFICF_HIP_CHECK(hipMemcpyAsync(n2, w.flag + g.n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_lv_relabel, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, src, w.comm, w.flag, w.lab);
    GFICF_HIP_CHECK(hipStreamSynchronize(st));
    return GFICF_OK;
  };
  // g reduced by the labels newid[comm[.]] (n2 of them) into the arrays of nl; kv of the result is NOT set here
  const auto reduce = [&](const LvGraph& g, const int32_t* comm, const int64_t* newid, int64_t n2, LvLevel& nl, LvGraph* out) -> int {
    int64_t m2 = 0;
    if (g.m > 0) {
      const u64 none = (u64)n2 << 32;
      hipLaunchKernelGGL(k_lv_emit, dim3(lv_blocks(g.n, 4)), dim3(256), 0, st, g, comm, newid, none, w.keys_a, w.vals_a);
      unsigned bits = 33;                        // the keys in use: (row < n2) << 32 | col, and none = n2 << 32
      while (bits < 64 && ((int64_t)1 << (bits - 32)) <= n2) ++bits;
      size_t tb = w.sort_tmp_bytes;
      GFICF_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (size_t)g.m, 0u, bits, st));
      hipLaunchKernelGGL(k_lv_heads, dim3(lv_blocks(g.m + 1, 256)), dim3(256), 0, st, w.keys_b, g.m, none, w.flag);
      int rc2 = gficf_exclusive_scan_i64(ctx, w.flag, g.m + 1);
      if (rc2) return rc2;
      GFICF_HIP_CHECK(hipMemcpyAsync(&m2, w.flag + g.m, sizeof(int64_t), hipMemcpyDeviceToHost, st));
      GFICF_HIP_CHECK(hipStreamSynchronize(st));
      GFICF_HIP_CHECK(hipMemsetAsync(nl.wt, 0, sizeof(u64) * (size_t)(m2 > 0 ? m2 : 1), st));
      GFICF_HIP_CHECK(hipMemsetAsync(nl.ptr, 0, sizeof(int64_t) * (size_t)(n2 + 1), st));
      hipLaunchKernelGGL(k_lv_reduce, dim3(lv_blocks(g.m, LV_RED_CHUNK)), dim3(256), 0, st, w.keys_b, w.vals_b, g.m, none, w.flag, nl.nbr, nl.wt, nl.ptr);
      rc2 = gficf_exclusive_scan_i64(ctx, nl.ptr, n2 + 1);
      if (rc2) return rc2;
    } else {
      GFICF_HIP_CHECK(hipMemsetAsync(nl.ptr, 0, sizeof(int64_t) * (size_t)(n2 + 1), st));
    }
    *out = LvGraph{n2, m2, nl.ptr, nl.nbr, nl.wt, nl.kv};
    return GFICF_OK;
  };

  // Random starts (reference src/RModularityOptimizer.cpp:108-142): every start begins from singletons, runs up to n_iter
  // passes and is kept if its modularity beats the best so far.  Nothing here is random; what a start varies is the seed
  // of the hash that splits the vertices into sub-round classes (start 0 with seed 0 is the plain deterministic run).
  double q_best = -INFINITY;
  int64_t n_best = 0;
  for (int start = 0; start < n_start; ++start) {
  start_seed = start == 0 && seed == 0 ? 0u : lv_hash((uint32_t)seed * 0x9E3779B1u + (uint32_t)start + 1u);
  have_labels = false;
  n_labels = 0;
  for (int pass = 0; pass < n_iter; ++pass) {
    LvGraph g = g0;
    u64 self_w = 0;
    bool any_move = false;
    int n_saved = 0;                             // levels 1 .. n_saved have their vertex map in w.tops
    int64_t saved_n[LV_MAX_SAVED + 1];
    for (int level = 0;; ++level) {
      const bool seeded = level == 0 && have_labels;
      if (debug) fprintf(stderr, "[louvain] pass %d level %d: n=%lld m=%lld\n", pass, level, (long long)g.n, (long long)g.m);
      rc = start_level(g, seeded ? w.lab : nullptr, n_labels, self_w);
      if (rc) return rc;
      bool level_moved = false;
      rc = local_moving(g, self_w, &level_moved);
      if (rc) return rc;
      any_move |= level_moved;
      q_final = q_prev;
      int64_t n2 = 0;
      rc = renumber(g, level == 0 ? nullptr : w.lab, &n2);
      if (rc) return rc;
      have_labels = true;
      *n_clusters = n_labels = n2;
      if (n2 == g.n || n2 <= 1 || (!level_moved && !(seeded && n2 < g.n))) break;      // nothing merged: the descent is done
      // ---- the reduced graph
      LvLevel& nl = w.lvl[level & 1];
      hipLaunchKernelGGL(k_lv_coarse_weights, dim3(lv_blocks(g.n, 256)), dim3(256), 0, st, g.n, w.size, w.flag, w.K, nl.kv);
      LvGraph g2;
      rc = reduce(g, w.comm, w.flag, n2, nl, &g2);
      if (rc) return rc;
      self_w += in_w;
      g = g2;
      if (algorithm == 2 && n_saved == level && level < LV_MAX_SAVED) {               // level + 1's vertices: lab as it is now
        GFICF_HIP_CHECK(hipMemcpyAsync(w.tops + (size_t)level * (size_t)N, w.lab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, st));
        saved_n[++n_saved] = n2;
      }
    }

    // ---- algorithm 2: back up through the levels, local moving on each with the labels found below it
    // (runLouvainAlgorithmWithMultilevelRefinement, reference :629-649).  The graph of a level is rebuilt from the finest
    // one by its saved vertex map (one sort) instead of being kept.
    if (algorithm == 2 && any_move) {
      for (int level = n_saved; level >= 0; --level) {
        const int32_t* top = level ? w.tops + (size_t)(level - 1) * (size_t)N : nullptr;
        LvGraph gl = g0;
        u64 self_l = 0;
        const int32_t* seed = w.lab;
        if (level) {
          const int64_t nl_n = saved_n[level];
          LvLevel& nl = w.lvl[0];
          hipLaunchKernelGGL(k_lv_iota, dim3(lv_blocks(nl_n, 256)), dim3(256), 0, st, nl_n, w.flag);
          GFICF_HIP_CHECK(hipMemsetAsync(nl.kv, 0, sizeof(u64) * (size_t)nl_n, st));
          GFICF_HIP_CHECK(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * (size_t)nl_n, st));
          hipLaunchKernelGGL(k_lv_accum, dim3(grid_cap(lv_blocks(N, 256))), dim3(256), 0, st, N, nl_n, top, w.kv0, (int32_t*)nullptr, nl.kv, w.cnt);
          GFICF_HIP_CHECK(hipMemsetAsync(w.scalars + 1, 0, sizeof(u64), st));
          hipLaunchKernelGGL(k_lv_internal, dim3(lv_blocks(N, 4) < 2048u ? lv_blocks(N, 4) : 2048u), dim3(256), 0, st, g0, top, w.scalars + 1);
          GFICF_HIP_CHECK(hipMemcpyAsync(&self_l, w.scalars + 1, sizeof(u64), hipMemcpyDeviceToHost, st));
          hipLaunchKernelGGL(k_lv_seed, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, top, w.lab, w.rank);
          rc = reduce(g0, top, w.flag, nl_n, nl, &gl);               // synchronises: self_l is there
          if (rc) return rc;
          seed = w.rank;
        }
        rc = start_level(gl, seed, n_labels, self_l);
        if (rc) return rc;
        bool level_moved = false;
        rc = local_moving(gl, self_l, &level_moved);
        if (rc) return rc;
        q_final = q_prev;
        int64_t n2 = 0;
        rc = renumber(gl, top, &n2);
        if (rc) return rc;
        *n_clusters = n_labels = n2;
      }
    }
    if (!any_move) break;
  }
  if (debug) fprintf(stderr, "[louvain] start %d: Q %.9f, %lld clusters\n", start, q_final, (long long)n_labels);
  if (q_final > q_best) {                        // strictly better, as the reference keeps the first of equals (:128)
    q_best = q_final;
    n_best = n_labels;
    if (n_start > 1) GFICF_HIP_CHECK(hipMemcpyAsync(w.best, w.lab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, st));
  }
  }
  if (n_start > 1) GFICF_HIP_CHECK(hipMemcpyAsync(w.lab, w.best, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, st));
  q_final = q_best;
  *n_clusters = n_best;

  // ---- clusters by decreasing size
  const int64_t C = *n_clusters;
  GFICF_HIP_CHECK(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * (size_t)C, st));
  hipLaunchKernelGGL(k_lv_accum, dim3(lv_blocks(N, 256) < 1024u ? lv_blocks(N, 256) : 1024u), dim3(256), 0, st, N, C, w.lab, (const u64*)nullptr,
                     (int32_t*)nullptr, (u64*)nullptr, w.cnt);
  hipLaunchKernelGGL(k_lv_size_keys, dim3(lv_blocks(C, 256)), dim3(256), 0, st, C, N, w.cnt, w.keys_a, w.vals_a);
  size_t tb = w.sort_tmp_bytes;
  GFICF_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (size_t)C, 0u, 64u, st));
  hipLaunchKernelGGL(k_lv_rank, dim3(lv_blocks(C, 256)), dim3(256), 0, st, C, w.vals_b, w.rank);
  hipLaunchKernelGGL(k_lv_final, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, w.lab, w.rank, d_labels);
  GFICF_HIP_CHECK(hipGetLastError());
  if (modularity) *modularity = q_final;
  return gficf_ctx_sync(ctx);
}

int gficf_louvain_host(gficf_ctx* ctx, int64_t N, const void* indptr, int indptr_is_i64, const int32_t* indices, const double* x,
                       double resolution, int algorithm, int n_start, int n_iter, int seed, int32_t* labels, int64_t* n_clusters, double* modularity) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (n_clusters) *n_clusters = 0;
  if (modularity) *modularity = 0.0;
  if (N == 0) return GFICF_OK;
  if (!indptr || !labels || !n_clusters) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  std::vector<int64_t> h_ptr((size_t)N + 1);
  for (int64_t c = 0; c <= N; ++c) h_ptr[(size_t)c] = indptr_is_i64 ? ((const int64_t*)indptr)[c] : (int64_t)((const int32_t*)indptr)[c];
  bool mono = h_ptr[0] == 0;
  for (int64_t c = 0; c < N && mono; ++c) mono = h_ptr[(size_t)c + 1] >= h_ptr[(size_t)c];
  const int64_t nnz = h_ptr[(size_t)N];
  if (!mono) GFICF_FAIL(GFICF_ERR_BAD_CSC, "indptr does not start at 0 or is not monotone");
  if (nnz > 0 && (!indices || !x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  const size_t nsz = (size_t)(nnz > 0 ? nnz : 1), wsb = gficf_louvain_workspace_bytes(N, nnz);
  gficf_arena ar;                                   // pool slot 0: no allocation per call
  const size_t o_ptr = ar.take(sizeof(int64_t) * ((size_t)N + 1)), o_idx = ar.take(sizeof(int32_t) * nsz), o_x = ar.take(sizeof(double) * nsz);
  const size_t o_lab = ar.take(sizeof(int32_t) * (size_t)N), o_ws = ar.take(wsb);
  hipError_t e = ar.bind(ctx, 0);
  int64_t* const d_ptr = ar.at<int64_t>(o_ptr); int32_t* const d_idx = ar.at<int32_t>(o_idx); double* const d_x = ar.at<double>(o_x);
  int32_t* const d_lab = ar.at<int32_t>(o_lab); void* const d_ws = ar.at<void>(o_ws);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ptr, h_ptr.data(), sizeof(int64_t) * ((size_t)N + 1), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_idx, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  int rc = GFICF_OK;
  if (e == hipSuccess) {
    rc = gficf_louvain_device(ctx, N, d_ptr, d_idx, d_x, nnz, resolution, algorithm, n_start, n_iter, seed, d_lab, n_clusters, modularity, d_ws, wsb);
    if (!rc) e = hipMemcpyAsync(labels, d_lab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_louvain_host: %s", hipGetErrorString(e));
  return rc;
}

}  // extern "C"
