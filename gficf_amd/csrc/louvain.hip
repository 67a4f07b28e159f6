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
// Device algorithm (deterministic parallel Louvain), per start:
//   * weights in 2^-32 fixed point (u64): every sum — a vertex's weight towards a community, the community totals —
//     is an integer sum, so atomics commute and the result does not depend on scheduling or on the order of a row's entries;
//   * local moving, synchronous within a sub-round: every vertex reads the same snapshot (labels, community totals and
//     sizes), accumulates its edge weight per neighbouring community in an LDS hash table (one wave per vertex up to 128
//     entries, one workgroup beyond), takes the community with the best gain
//       gain(v -> c) = w(v, c) - k_v * K_c(without v) * resolution / 2W         (reference :540, ties: smaller id :541)
//     if that beats staying; two singletons never swap (only the larger id moves).  The vertices are split into S
//     hash classes that move one after the other (S grows as the graph gets small, where simultaneous moves hurt most);
//     totals are applied between sub-rounds.  An iteration that lowers Q is undone and ends the level;
//   * reduction: communities renumbered by a scan; every vertex sums its entries per neighbouring community (the same LDS
//     table) and appends one entry per community to the row of its own community in the coarse graph.  The coarse graph is
//     a MULTIGRAPH in pointerB / pointerE form (a row may name a neighbour once per member vertex; every consumer sums
//     per community anyway); no sort anywhere;
//   * n_iter > 1 restarts from the finest graph with the labels found so far, as the reference's iterations do;
//   * algorithm 2 (runLouvainAlgorithmWithMultilevelRefinement, :629-649): after the descent, back up through the levels
//     with one more local moving on each, seeded with the labels found below it; a level's graph is rebuilt from the
//     finest one through its saved vertex map rather than kept.
//   * a vertex whose every option has a negative gain leaves for an unused cluster (:546-550): its own id, if free.
//   * n_start "random starts": each start varies the seed of the class hash (the only arbitrary choice there is); the best
//     modularity wins, as in the reference.  Start 0 with seed 0 is the plain run.
//   * the alternative modularity function (2): unit node weights, the resolution as given — a context option.
// Not reproduced: algorithm 3 (SLM).
//
// Round 6 — the starts in ONE launch set, the iterations without host round trips:
//   * the n_start starts are independent problems on one graph.  They run as ONE problem on the disjoint union of B copies
//     ("components"; B = as many starts as the workspace holds, at most 16): union vertex b * N + v, labels / totals / sizes
//     per union vertex, every quantity that decides something (Q, moved vertices, sub-round count, seed) per component.
//     Level 0 never materialises the copies: a wave loads a vertex's row ONCE and evaluates it for every component.  Coarse
//     levels are one graph whose vertex ids are the scan's new ids (contiguous per component).  A component whose descent
//     or whole run has ended idles (its kernels skip it) while the others go on; its results are those of running that start
//     alone (tests/test_louvain_gpu.py: batched == one by one);
//   * convergence and undo are decided on the device (k_lv_decide: one small block per iteration): the move kernel of an
//     iteration's first sub-round also sums the internal weight of the labels it READS, i.e. the previous iteration's
//     result, so the quality check costs no pass of its own; an iteration's kernels look at the component's action flag and
//     do nothing once its level has ended.  The host enqueues one iteration AHEAD and waits for the previous one's event:
//     the stream is never drained inside a level; per level there is one synchronisation (the sizes of the next level);
//   * counters that every moving vertex used to hit with one global atomic (a single address: 150-220 us of the 158-226 us
//     a level-0 sub-round took while most vertices still moved) are per-block partial sums in fixed slots, added up by
//     k_lv_decide; the vertex kernels are persistent grids striding over the vertices.
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
constexpr int LV_MID_DEG = 4096, LV_MID_SLOTS = 1024;   // up to here: still one wave per vertex, a 1024-slot table of its own, passes by hash class beyond 512 entries
constexpr int LV_EMIT_SLOTS = 2048;            // the reduction's workgroup-per-vertex table for the same vertices
constexpr int LV_BIG_SLOTS = 8192;             // beyond: one workgroup per vertex, 32 KB keys + 64 KB sums
constexpr int LV_MAX_ITERS = 64;
constexpr int LV_MAX_SAVED = 12;              // levels whose vertex map is kept for the refinement of algorithm 2
constexpr int LV_MAX_B = 16;                  // starts run together
constexpr int LV_GRID = 1536;                 // persistent grid of the wave-per-vertex kernels (workgroups of 4 waves): what is RESIDENT at once on 256 CUs
                                              // (k_lv_move_small is held to 80 VGPRs: six workgroups a CU) — 2048 ran as 1.6 rounds of five a CU (8.1 -> 7.7 ms at
                                              // config 3, 10 starts; six a CU: 7.35; eight spill: 8.5)
constexpr int LV_GRID_BIG = 1280;             // ... of the middle-degree (two waves, 28 KB of LDS: five a CU) and workgroup-per-vertex kernels
constexpr int LV_SQ_BLOCKS = 64;              // slices of a component's communities in the fixed-order sum of squares
constexpr int LV_ACC_BINS = 4096;

enum { LV_IDLE = 0, LV_CONTINUE = 1, LV_STOP_KEEP = 2, LV_STOP_UNDO = 3 };

// One level's union graph.  Level 0: `rep` copies of the caller's matrix, never materialised (union vertex b * nb + v has
// the row of v with every neighbour shifted by b * nb); coarse levels: rep == 1.  Rows are [beg[v], end[v]).
struct LvG {
  int64_t n;                 // union vertices = rep * nb
  int64_t nb;                // vertices of the stored graph
  int32_t rep;
  const int64_t* beg;
  const int64_t* end;
  const int32_t* nbr;
  const u64* wt;
  const u64* kv;             // [nb] vertex weights
  const uint8_t* vcomp;      // [n] component of a union vertex; NULL at level 0: the copy number (also when there is one copy only)
};

struct LvComp {              // one start ("component" of the union)
  int64_t v0, n;             // its vertices at the current level: union ids [v0, v0 + n)
  int64_t seed_base;         // seeded start: community of a vertex = seed_base + its seed label
  int64_t lab_base;          // first LDS bin of its seed labels (binned accumulation)
  int64_t n_labels;          // the labels of the last renumbering lie in [0, n_labels)
  int64_t newbase, n2;       // after the level's renumbering: first new id, communities in use
  u64 self_w, in_w;          // weight folded into the level's vertices; internal weight of the accepted labels on the level's graph
  u64 in_run;                // internal weight of the labels as they are (kept up to date by the changes the sub-round-0 kernels report)
  double q_prev, q_final;
  uint32_t seed;             // of the sub-round class hash: what a "random start" varies
  int32_t S;                 // sub-rounds of the level
  int32_t live;              // takes part in this level
  int32_t cont;              // goes on to the next level (set when the level is renumbered)
  int32_t level_done, action, iter, level_moved, any_move, finished;
  int32_t n_saved, pad;
  int64_t saved_n[LV_MAX_SAVED + 1];
};

struct LvCtl {               // device control block
  LvComp c[LV_MAX_B];
  int32_t B, pad;
  unsigned n_mid, n_large;   // vertices of the workgroup path (list counts) of the graph the lists were built for
  int64_t n_union2;          // union vertices after the renumbering
};

struct LvHostComp { int64_t n2; double q_prev; int32_t cont, any_move, level_moved, live; };
struct LvHost {              // pinned host memory the control kernels write (read after an event)
  int32_t n_cont[LV_MAX_ITERS + 2];
  int64_t n_union2;
  unsigned n_mid, n_large;
  LvHostComp c[LV_MAX_B];
  double q_iter[LV_MAX_B];   // (debug trace)
  unsigned mv_iter[LV_MAX_B];
};

struct LvParts {             // per-block partial sums of the move kernels, fixed slots (no atomics, no clearing)
  // internal weight: written by an iteration's sub-round-0 kernels, read by the same iteration's k_lv_decide.  Moved vertices: summed over
  // ALL sub-rounds of an iteration and read by the NEXT iteration's k_lv_decide, after that iteration's sub-round 0 has started a new
  // count — hence two sets, by iteration parity.
  u64* in_small;  unsigned* mv_small[2];     // [LV_GRID][LV_MAX_B]
  u64* in_mid;    unsigned* mv_mid[2];       // [LV_GRID_BIG][LV_MAX_B]
  u64* in_large;  unsigned* mv_large[2];
  double* sq;                              // [LV_MAX_B][LV_SQ_BLOCKS]
};

__device__ __host__ static inline uint32_t lv_hash(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}

__device__ __host__ static inline int lv_sub_rounds_of(int64_t n, int s_env) {
  if (s_env >= 1 && s_env <= 64) return s_env;
  // sub-rounds an iteration is cut into (vertices of one hash class move together).  One is too few (modularity 0.03 - 0.06 below the reference's
  // on weakly structured graphs: neighbours move at once), two to sixteen give the same quality within run-to-run differences of 0.003 on graphs of 300
  // to 60 000 vertices (tools/louvain_subrounds_probe.py, profiles/r06_louvain_subrounds.txt).  Through round 6's first form small levels took 8 / 16:
  // the coarse levels of a big problem are a few hundred vertices, and 16 sub-rounds x 3 launches an iteration were 1 ms of the 10.8 at config 3.
  return n > 50000 ? 2 : 4;
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

// vertex weights (the row sums without the diagonal) and 2W: a wave per vertex, the row read by consecutive lanes; one atomic per workgroup
__global__ __launch_bounds__(256) void k_lv_vertex_weight(int64_t n, int64_t m, const int64_t* __restrict__ ptr, const int32_t* __restrict__ nbr,
                                                         const u64* __restrict__ wt, u64* __restrict__ kv, u64* __restrict__ two_w,
                                                         uint32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 total = 0;                                       // lane 0: the sum over this wave's vertices
  for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < n; v += (int64_t)gridDim.x * 4) {
    int64_t lo = ptr[v], hi = ptr[v + 1];
    if (lo < 0 || hi < lo || hi > m) { if (lane == 0) atomicOr(status, GFICF_ST_BAD_CSC); lo = hi = 0; }
    u64 s = 0;
    for (int64_t e = lo + lane; e < hi; e += 64) {
      const int32_t u = nbr[e];
      if (u != v && u >= 0 && u < n) s += wt[e];
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane == 0) { kv[v] = s; total += s; }
  }
  __shared__ u64 s_sum[4];
  if (lane == 0) s_sum[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (t) atomicAdd(two_w, t);
  }
}

__global__ __launch_bounds__(256) void k_lv_fill_u64(int64_t n, u64 v, u64* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

// ---- control kernels (one small block; thread b = component b)
// A batch begins: the seeds of its starts, nothing found yet.
__global__ void k_lv_ctl_batch(LvCtl* ctl, int B, int first_start, int seed) {
  const int b = threadIdx.x;
  if (b == 0) { ctl->B = B; ctl->n_mid = ctl->n_large = 0; ctl->n_union2 = 0; }
  if (b >= LV_MAX_B) return;
  LvComp& C = ctl->c[b];
  const int start = first_start + b;
  C.seed = start == 0 && seed == 0 ? 0u : lv_hash((uint32_t)seed * 0x9E3779B1u + (uint32_t)start + 1u);
  C.finished = b < B ? 0 : 1;
  C.any_move = 1;
  C.n_labels = 0;
  C.q_prev = C.q_final = 0.0;
  C.self_w = C.in_w = 0;
  C.live = C.cont = 0;
  C.n_saved = 0;
  C.v0 = C.n = C.n2 = C.newbase = 0;
}

// A level begins.  mode 0: level 0 of a pass (v0 = b * N, n = N; pass > 0: seeded with the labels found so far, a start whose last
// pass moved nothing has finished).  mode 1: the level below was renumbered and reduced (v0 = newbase, n = n2 for the
// components that go on).  mode 2: a refinement level of algorithm 2 (the participants and sizes were set by k_lv_ctl_refine).
__global__ void k_lv_ctl_level(LvCtl* ctl, int mode, int pass, int64_t N, int s_env) {
  __shared__ int64_t s_nl[LV_MAX_B];
  const int b = threadIdx.x;
  const int B = ctl->B;
  if (b < LV_MAX_B) {
    LvComp& C = ctl->c[b];
    if (mode == 0) {
      if (pass > 0 && !C.any_move) C.finished = 1;
      C.live = b < B && !C.finished;
      C.v0 = b < B ? (int64_t)b * N : 0;                 // (a slot beyond the batch: an empty range, never an index past the union)
      C.n = b < B ? N : 0;
      C.self_w = 0;
      C.any_move = 0;
      C.n_saved = 0;
      C.seed_base = C.v0;
    } else if (mode == 1) {
      C.live = C.cont;
      C.v0 = C.newbase; C.n = C.n2;
      C.seed_base = C.v0;
    }
    C.cont = 0;
    C.S = lv_sub_rounds_of(C.n, s_env);
    C.level_done = C.live ? 0 : 1;
    C.action = LV_IDLE;
    C.iter = 0;
    C.level_moved = 0;
    C.in_run = 0;
    s_nl[b] = b < B && C.live ? C.n_labels : 0;
  }
  __syncthreads();
  if (b < LV_MAX_B) {
    int64_t base = 0;
    for (int t = 0; t < b; ++t) base += s_nl[t];
    ctl->c[b].lab_base = base;
  }
}

// Labels (seed == NULL: singletons), totals and sizes of the level.  A seeded start leaves totals and sizes at zero for
// k_lv_accum.  The vertices of a component that does not take part are dead: own label, size 0 (they vanish at the next renumbering).
__global__ __launch_bounds__(256) void k_lv_init(LvG g, const LvCtl* __restrict__ ctl, const int32_t* __restrict__ seed, int32_t* __restrict__ comm,
                                                 u64* __restrict__ K, int32_t* __restrict__ size, int32_t* __restrict__ mark_r, int32_t* __restrict__ mark_w,
                                                 u64* __restrict__ iw) {
  for (int64_t gv = (int64_t)blockIdx.x * 256 + threadIdx.x; gv < g.n; gv += (int64_t)gridDim.x * 256) {
    mark_r[gv] = 0; mark_w[gv] = 0; iw[gv] = 0ull;
    const int64_t b = g.rep > 1 ? gv / g.nb : 0;
    const int64_t v = gv - b * g.nb;
    const int comp = g.vcomp ? (int)g.vcomp[gv] : (int)b;
    const LvComp& C = ctl->c[comp];
    if (!C.live) { comm[gv] = (int32_t)gv; K[gv] = 0ull; size[gv] = 0; continue; }
    if (seed) { comm[gv] = (int32_t)(C.seed_base + seed[gv]); K[gv] = 0ull; size[gv] = 0; }
    else { comm[gv] = (int32_t)gv; K[gv] = g.kv[v]; size[gv] = 1; }
  }
}

// Totals and member counts of the communities comm[] names (K, size zeroed before).  With few labels every vertex would hit the
// same few addresses: when all components' labels fit LV_ACC_BINS bins (bin = lab_base + label - seed_base) they are summed in
// LDS first, one global atomic per label and block.
__global__ __launch_bounds__(256) void k_lv_accum(LvG g, const LvCtl* __restrict__ ctl, int binned, const int32_t* __restrict__ comm,
                                                  u64* __restrict__ K, int32_t* __restrict__ size) {
  __shared__ u64 s_k[LV_ACC_BINS];
  __shared__ int32_t s_n[LV_ACC_BINS];
  __shared__ int32_t s_c[LV_ACC_BINS];
  if (binned) {
    for (int t = threadIdx.x; t < LV_ACC_BINS; t += 256) { s_k[t] = 0ull; s_n[t] = 0; s_c[t] = -1; }
    __syncthreads();
  }
  for (int64_t gv = (int64_t)blockIdx.x * 256 + threadIdx.x; gv < g.n; gv += (int64_t)gridDim.x * 256) {
    const int64_t b = g.rep > 1 ? gv / g.nb : 0;
    const int comp = g.vcomp ? (int)g.vcomp[gv] : (int)b;
    const LvComp& C = ctl->c[comp];
    if (!C.live) continue;
    const int32_t c = comm[gv];
    const u64 k = g.kv[gv - b * g.nb];
    if (binned) {
      const int bin = (int)(C.lab_base + ((int64_t)c - C.seed_base));
      s_c[bin] = c;
      atomicAdd(&s_k[bin], k);
      atomicAdd(&s_n[bin], 1);
    } else {
      atomicAdd(&K[c], k);
      atomicAdd(&size[c], 1);
    }
  }
  if (binned) {
    __syncthreads();
    for (int t = threadIdx.x; t < LV_ACC_BINS; t += 256) {
      if (s_n[t]) { atomicAdd(&size[s_c[t]], s_n[t]); if (s_k[t]) atomicAdd(&K[s_c[t]], s_k[t]); }
    }
  }
}

// the vertices of the workgroup path: middle degrees from the front of the list, large ones from its end
__global__ __launch_bounds__(256) void k_lv_list_big(LvG g, const int64_t* __restrict__ n_dev, int32_t* __restrict__ big, unsigned* __restrict__ n_mid_large) {
  const int64_t nb = n_dev ? *n_dev : g.nb;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nb; v += (int64_t)gridDim.x * 256) {
    const int64_t deg = g.end[v] - g.beg[v];
    if (deg > LV_MID_DEG) big[nb - 1 - atomicAdd(n_mid_large + 1, 1u)] = (int32_t)v;
    else if (deg > LV_SMALL_DEG) big[atomicAdd(n_mid_large, 1u)] = (int32_t)v;
  }
}

// ---- local moving
struct LvMove {
  double r;            // resolution / 2W (in fixed-point units of 2W)
  int s;               // this sub-round's hash class
  // Pruning ("fast local moving"): a vertex is looked at again only when it or one of its neighbours has moved since it was last looked at.
  // Every sub-round kernel of a level has an epoch (iteration * sub-rounds + sub-round + 1); a vertex that decides to move stamps itself and
  // its neighbours with it (plain stores of one value: no clearing, no race); a vertex of this kernel's class was last looked at one
  // iteration ago, so it is looked at now iff its stamp >= thr = epoch - sub-rounds (stamps start at 0: everybody in iteration 0).
  int epoch, thr;
};

__device__ static inline void lv_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ static inline bool lv_better(double g, int32_t c, double bg, int32_t bc) { return g > bg || (g == bg && c < bc); }

// Wave reductions on DPP row operations (no LDS traffic; the ds_bpermute butterflies they replace were ~1.2 us of the ~4.5 us a vertex took).
// Every lane must be active.  quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror leave each row of 16 uniform; row_bcast15 (rows
// 1, 3) and row_bcast31 (rows 2, 3) carry the rows' results down: lane 63 holds the wave's, read back with v_readlane.
template <int CTRL, int RM>
__device__ static inline int lv_dpp(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, RM, 0xF, false); }

template <int CTRL, int RM>
__device__ static inline void lv_sum_step(u64& x) {
  const uint32_t lo = (uint32_t)lv_dpp<CTRL, RM>(0, (int)(uint32_t)x), hi = (uint32_t)lv_dpp<CTRL, RM>(0, (int)(uint32_t)(x >> 32));
  x += ((u64)hi << 32) | lo;
}
__device__ static inline u64 lv_wave_sum(u64 x) {
  lv_sum_step<0xB1, 0xF>(x); lv_sum_step<0x4E, 0xF>(x); lv_sum_step<0x141, 0xF>(x); lv_sum_step<0x140, 0xF>(x);
  lv_sum_step<0x142, 0xA>(x); lv_sum_step<0x143, 0xC>(x);
  return ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
}

struct LvBest { double g; int32_t c; };      // best gain and its community (ties: the smaller id)

template <int CTRL, int RM>
__device__ static inline double lv_dpp_f64(double x) {
  return __hiloint2double(lv_dpp<CTRL, RM>(__double2hiint(x), __double2hiint(x)), lv_dpp<CTRL, RM>(__double2loint(x), __double2loint(x)));
}
// The wave's best, in every lane: the largest gain (v_max_f64 over the DPP steps), then the smallest community among the lanes that hold it —
// the same order as lv_better, in about half the instructions of reducing the pair at once.
__device__ static inline LvBest lv_wave_best(LvBest x) {
  double m = x.g;
  m = fmax(m, lv_dpp_f64<0xB1, 0xF>(m)); m = fmax(m, lv_dpp_f64<0x4E, 0xF>(m)); m = fmax(m, lv_dpp_f64<0x141, 0xF>(m));
  m = fmax(m, lv_dpp_f64<0x140, 0xF>(m)); m = fmax(m, lv_dpp_f64<0x142, 0xA>(m)); m = fmax(m, lv_dpp_f64<0x143, 0xC>(m));
  m = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
  int k = x.g == m ? x.c : INT32_MAX;
  k = min(k, lv_dpp<0xB1, 0xF>(k, k)); k = min(k, lv_dpp<0x4E, 0xF>(k, k)); k = min(k, lv_dpp<0x141, 0xF>(k, k));
  k = min(k, lv_dpp<0x140, 0xF>(k, k)); k = min(k, lv_dpp<0x142, 0xA>(k, k)); k = min(k, lv_dpp<0x143, 0xC>(k, k));
  return LvBest{m, __builtin_amdgcn_readlane(k, 63)};
}

// does component C take part in the sub-round kernel (s, first)?  first = the kernel of sub-round 0, which runs BEFORE the iteration's
// decision: every component whose level has not ended
__device__ static inline bool lv_runs(const LvComp& C, int s, bool first) {
  return first ? !C.level_done : (C.action == LV_CONTINUE && s < C.S);
}

// the decision for vertex gv, identical in every lane (all inputs are wave-uniform).  Kcv, szcv, szgv: total and size of its community, size of
// the community that carries its own id — loaded by the caller before the table work, off the critical path
__device__ static inline int32_t lv_decide_vertex(int64_t gv, int32_t cv, double bg, int32_t bc, u64 stay_w, u64 kvv, double r, u64 Kcv, int32_t szcv,
                                                  int32_t szgv, const int32_t* __restrict__ size, bool* moved) {
  const double kvd = (double)kvv;
  const double g_stay = (double)stay_w - kvd * (double)(Kcv - kvv) * r;
  bool move = bc != INT32_MAX && bg > g_stay;
  if (move && szcv == 1 && size[bc] == 1 && bc > cv) move = false;      // two singletons never swap
  int32_t to = move ? bc : cv;
  // every option loses: alone is better (the reference's move into an unused cluster, :546-550).  The unused cluster
  // is the vertex's own id when nobody holds it — unique per vertex, so simultaneous escapes never meet.
  if ((move ? bg : g_stay) < 0.0 && szcv > 1 && szgv == 0) { to = (int32_t)gv; move = true; }
  *moved = move;
  return to;
}

// claims or finds community c's slot in a table of SLOTS and adds w; *own: this lane claimed it (it will evaluate the community and hand the
// slot back empty).  Returns -1 when the table is full (reported by the caller).
template <int SLOTS>
__device__ static inline int lv_insert(int32_t* key, u64* val, int32_t c, u64 w, bool* own) {
  uint32_t h = lv_hash((uint32_t)c) & (SLOTS - 1);
  *own = false;
  for (int probes = 0; probes < SLOTS; ++probes) {
    const int32_t old = atomicCAS(&key[h], -1, c);
    if (old == -1 || old == c) { *own = old == -1; atomicAdd(&val[h], w); return (int)h; }
    h = (h + 1) & (SLOTS - 1);
  }
  return -1;
}

// one copy's view of a vertex in a round of k_lv_move_small
struct LvSide {
  int mode;                  // 0: not in this round, 1: only its share of the internal weight (sub-round 0, other class), 2: evaluate
  int comp;
  int64_t base, gv;
  int32_t cv;
  u64 Kcv; int32_t szcv, szgv;
};

template <bool FIRST>
__device__ static inline LvSide lv_side(const LvG& g, const LvCtl* __restrict__ ctl, int s, int b, int64_t v, const int32_t* __restrict__ comm,
                                        const u64* __restrict__ K, const int32_t* __restrict__ size) {
  LvSide x;
  x.mode = 0; x.comp = 0; x.base = 0; x.gv = 0; x.cv = 0; x.Kcv = 0; x.szcv = 0; x.szgv = 0;
  if (b >= g.rep) return x;
  x.comp = g.vcomp ? (int)g.vcomp[v] : b;
  const LvComp& C = ctl->c[x.comp];
  if (!lv_runs(C, s, FIRST)) return x;
  x.base = (int64_t)b * g.nb; x.gv = x.base + v;
  const bool in_class = C.S == 1 || (int)(lv_hash((uint32_t)(x.gv - C.v0) + C.seed) % (uint32_t)C.S) == s;
  if (!FIRST && !in_class) return x;
  x.mode = in_class ? 2 : 1;
  x.cv = comm[x.gv];
  if (in_class) { x.Kcv = K[x.cv]; x.szcv = size[x.cv]; x.szgv = size[x.gv]; }
  return x;
}

// hash class of a vertex for S sub-rounds (S is a power of two unless GFICF_LOUVAIN_SUBROUNDS says otherwise)
__device__ static inline int lv_class(uint32_t x, int S) {
  const uint32_t h = lv_hash(x);
  return (S & (S - 1)) == 0 ? (int)(h & (uint32_t)(S - 1)) : (int)(h % (uint32_t)S);
}

// One wave per vertex (at most LV_SMALL_DEG entries, two per lane, loaded ONCE and evaluated for every copy of the graph).  The wave's
// table is cleared once: the lane that claims a slot ("owner") evaluates that community and hands the slot back empty.
// A vertex of at most 64 entries (one per lane) takes TWO copies per round: the entries go into the table once with copy b's communities and
// once with copy b + 1's — the community ids of two copies never meet — and the two evaluations overlap their latencies.
// The entries INSIDE the vertex's own community — in a settled partition most of the row — never go through the table: they are added up in
// an LDS cell of their own (one instruction; the kernel is bound by VALU issue, ~130 instructions per evaluation before this form, and the
// LDS pipe idles), which is the staying side of the decision and, FIRST (sub-round 0), the vertex's share of the internal weight of the
// labels this kernel reads, i.e. of the previous iteration's result (k_lv_decide turns it into Q).
// Lane l < 16 carries component l's state for the whole kernel (read with v_readlane: no control-block loads in the loop).
// Tried on one box, variants interleaved, and dropped (profiles/r06_louvain_ab.txt): the copies as the OUTER loop, two at a time over all
// vertices, so that what is gathered for them stays in L2 (the early iterations miss L2 35 % of the time, ~1 GB of fabric traffic a sub-round)
// — 11.1 ms against 10.4 at the config-3 shape with ten starts: re-reading the rows per pair costs more than the misses; the copies
// INTERLEAVED in the state arrays: more misses, not fewer; both entries probing in one wave-uniform loop instead of two per-lane loops: no change.
template <bool FIRST>
__global__ __launch_bounds__(256, 6) void k_lv_move_small(LvG g, const LvCtl* __restrict__ ctl, LvMove mv, const int32_t* __restrict__ comm,
                                                       const u64* __restrict__ K, const int32_t* __restrict__ size, int32_t* __restrict__ next,
                                                       const int32_t* __restrict__ mark_r, int32_t* __restrict__ mark_w, u64* __restrict__ iw,
                                                       u64* __restrict__ part_in, unsigned* __restrict__ part_mv) {
  __shared__ int32_t s_key[4][LV_SMALL_SLOTS];
  __shared__ u64 s_val[4][LV_SMALL_SLOTS];
  __shared__ u64 s_stay[4][2];
  __shared__ u64 s_acc[4][LV_MAX_B];
  __shared__ unsigned s_cnt[4][LV_MAX_B];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t* key = s_key[wave];
  u64* val = s_val[wave];
  u64* cell = s_stay[wave];
  // the table belongs to this wave alone and LDS serves a wave's operations in issue order: a wave-level fence (no
  // workgroup barrier) is all that separates clearing, filling and reading it
  for (int t = lane; t < LV_SMALL_SLOTS; t += 64) { key[t] = -1; val[t] = 0ull; }
  if (lane < 2) cell[lane] = 0ull;
  int st_run = 0, st_S = 1;
  uint32_t st_off = 0;       // seed - v0: the class of union vertex gv is that of hash(gv + st_off)
  if (lane < LV_MAX_B) {
    const LvComp& C = ctl->c[lane];
    st_run = lv_runs(C, mv.s, FIRST) ? 1 : 0;
    st_S = C.S;
    st_off = C.seed - (uint32_t)C.v0;
  }
  lv_wave_sync();
  u64 acc = 0;               // lane b: internal weight summed for component b
  unsigned cnt = 0;          // lane b: vertices of component b that move
  // Look-ahead (round 6, late): the lanes first look at the wave's next 64 / P vertices at once — lane (j, b) at copy b of the j-th of them, P = the
  // copies rounded up to a power of two — and the wave then works through the vertices that have a copy to look at.  Before, a wave found out
  // vertex by vertex (row loaded, then one stamp per pair of copies, each load behind the last): 31 us a launch at 10 x 54 k vertices with NOTHING
  // to do, which is what the last ten iterations of a level are (a few hundred vertices still moving) — 3 - 5 us now.
  const int P = g.rep <= 1 ? 1 : g.rep <= 2 ? 2 : g.rep <= 4 ? 4 : g.rep <= 8 ? 8 : 16;
  const int la_j = lane / P, la_b = lane % P;
  const int64_t v_stride = (int64_t)gridDim.x * 4;
  for (int64_t vb = (int64_t)blockIdx.x * 4 + wave; vb < g.nb; vb += v_stride * (64 / P)) {
   u64 la_mask;
   {
    const int64_t vj = vb + (int64_t)la_j * v_stride;
    const bool there = la_b < g.rep && vj < g.nb;
    const int comp = there ? (g.vcomp ? (int)g.vcomp[vj] : la_b) : 0;
    const int run = __shfl(st_run, comp), S = __shfl(st_S, comp);           // (every lane takes part: the state sits in lanes 0 .. 15)
    const uint32_t off = (uint32_t)__shfl((int)st_off, comp);
    bool act = false;
    if (there && run) {
      const uint32_t gv = (uint32_t)((int64_t)la_b * g.nb + vj);
      const bool in_class = S == 1 || lv_class(gv + off, S) == mv.s;
      act = (in_class || FIRST) && mark_r[gv] >= mv.thr;
    }
    la_mask = __ballot(act);
   }
   while (la_mask) {                               // uniform over the wave
    const int jj = __builtin_ctzll(la_mask) / P;
    const unsigned look = (unsigned)((la_mask >> (jj * P)) & ((1ull << P) - 1ull));      // bit b: copy b of this vertex is looked at
    la_mask &= ~(((1ull << P) - 1ull) << (jj * P));
    const int64_t v = vb + (int64_t)jj * v_stride;
    const int64_t lo = g.beg[v], hi = g.end[v];
    if (hi - lo > LV_SMALL_DEG) continue;
    const int64_t e0 = lo + lane, e1 = e0 + 64;
    int32_t u0 = -1, u1 = -1;
    u64 w0 = 0, w1 = 0;
    if (e0 < hi) { u0 = g.nbr[e0]; w0 = g.wt[e0]; if (u0 == v) u0 = -1; }
    if (e1 < hi) { u1 = g.nbr[e1]; w1 = g.wt[e1]; if (u1 == v) u1 = -1; }
    const u64 kvv = g.kv[v];
    const double kvd = (double)kvv;
    const int comp0 = g.vcomp ? (int)g.vcomp[v] : 0;
    const bool pair = g.rep > 1 && hi - lo <= 64;
    for (int b = 0; b < g.rep; b += pair ? 2 : 1) {
      // side A: copy b (a coarse level: the vertex's component); side B: copy b + 1 in a pair round
      const int compA = g.vcomp ? comp0 : b, compB = b + 1;
      const uint32_t gvA = (uint32_t)(b * g.nb + v), gvB = gvA + (uint32_t)g.nb;
      int modeA = 0, modeB = 0;      // 0: not in this round, 1: only its share of the internal weight (sub-round 0, other class), 2: evaluate
      if (__builtin_amdgcn_readlane(st_run, compA)) {
        const int S = __builtin_amdgcn_readlane(st_S, compA);
        const bool in_class = S == 1 || lv_class(gvA + (uint32_t)__builtin_amdgcn_readlane((int)st_off, compA), S) == mv.s;
        modeA = in_class ? 2 : FIRST ? 1 : 0;
      }
      if (pair && compB < g.rep && __builtin_amdgcn_readlane(st_run, compB)) {
        const int S = __builtin_amdgcn_readlane(st_S, compB);
        const bool in_class = S == 1 || lv_class(gvB + (uint32_t)__builtin_amdgcn_readlane((int)st_off, compB), S) == mv.s;
        modeB = in_class ? 2 : FIRST ? 1 : 0;
      }
      // looked at only when it or a neighbour has moved since it was last looked at (the stamps this kernel reads were merged before it began;
      // read by the look-ahead above)
      if (modeA && !((look >> b) & 1u)) modeA = 0;
      if (modeB && !((look >> (b + 1)) & 1u)) modeB = 0;
      if (!modeA && !modeB) continue;
      const uint32_t baseA = gvA - (uint32_t)v, baseB = gvB - (uint32_t)v;
      // this lane's entries: (cx, wx) belongs to side A, (cy, wy) to side B in a pair round and to side A otherwise
      int32_t cvA = 0, cvB = 0, cx = -1, cy = -1;
      if (modeA) { cvA = comm[gvA]; if (u0 >= 0) cx = comm[baseA + (uint32_t)u0]; }
      if (pair) { if (modeB) { cvB = comm[gvB]; if (u0 >= 0) cy = comm[baseB + (uint32_t)u0]; } }
      else if (modeA && u1 >= 0) cy = comm[baseA + (uint32_t)u1];
      const u64 wx = w0, wy = pair ? w0 : w1;
      const int32_t cvy = pair ? cvB : cvA;
      const bool evA = modeA == 2, evB = modeB == 2, evY = pair ? evB : evA;
      const bool inx = cx >= 0 && cx == cvA, iny = cy >= 0 && cy == cvy;
      // totals and sizes the decisions need, on their way while the table is filled
      u64 Kx = 0, Ky = 0, KcvA = 0, KcvB = 0;
      int32_t szA = 0, szgA = 0, szB = 0, szgB = 0;
      if (evA) { KcvA = K[cvA]; szA = size[cvA]; szgA = size[gvA]; if (cx >= 0 && !inx) Kx = K[cx]; }
      if (evB) { KcvB = K[cvB]; szB = size[cvB]; szgB = size[gvB]; }
      if (evY && cy >= 0 && !iny) Ky = K[cy];
      if (inx) atomicAdd(&cell[0], wx);
      if (iny) atomicAdd(&cell[pair ? 1 : 0], wy);
      // Both entries go into the table together: the first probes back to back (one LDS round trip for the two), and a loop that the WAVE
      // leaves as soon as no lane is still probing (a scalar branch on a ballot; the per-lane probe loops this replaces were unrolled
      // into ~10 exec-mask instructions per probe and waited for every round trip twice).  At most 128 entries in 256 slots: it ends.
      int slotx = -1, sloty = -1;
      bool ownx = false, owny = false;
      if (evA && cx >= 0 && !inx) slotx = lv_insert<LV_SMALL_SLOTS>(key, val, cx, wx, &ownx);
      if (evY && cy >= 0 && !iny) sloty = lv_insert<LV_SMALL_SLOTS>(key, val, cy, wy, &owny);
      lv_wave_sync();
      const u64 stayA = cell[0], stayB = cell[1];
      LvBest ba{-INFINITY, INT32_MAX}, bb{-INFINITY, INT32_MAX};
      if (ownx) {
        ba.g = (double)val[slotx] - kvd * (double)Kx * mv.r; ba.c = cx;
        key[slotx] = -1; val[slotx] = 0ull;
      }
      if (owny) {
        LvBest& t = pair ? bb : ba;
        const double gain = (double)val[sloty] - kvd * (double)Ky * mv.r;
        if (lv_better(gain, cy, t.g, t.c)) { t.g = gain; t.c = cy; }
        key[sloty] = -1; val[sloty] = 0ull;
      }
      if (evA) ba = lv_wave_best(ba);
      if (evB) bb = lv_wave_best(bb);
      lv_wave_sync();                            // every lane has read the cells and the slots: they are emptied before the next round fills them
      if (lane < 2) cell[lane] = 0ull;
      if (modeA) {
        if (FIRST) {                             // the change of its share of the internal weight since it was last looked at
          const u64 was = iw[gvA];
          if (lane == 0) iw[gvA] = stayA;
          if (lane == compA) acc += stayA - was;
        }
        if (evA) {
          bool moved;
          const int32_t to = lv_decide_vertex((int64_t)gvA, cvA, ba.g, ba.c, stayA, kvv, mv.r, KcvA, szA, szgA, size, &moved);
          if (lane == 0) next[gvA] = to;
          if (lane == compA) cnt += moved ? 1u : 0u;
          if (moved) {                           // it and its neighbours are looked at again
            if (lane == 0) mark_w[gvA] = mv.epoch;
            if (u0 >= 0) mark_w[baseA + (uint32_t)u0] = mv.epoch;
            if (!pair && u1 >= 0) mark_w[baseA + (uint32_t)u1] = mv.epoch;
          }
        }
      }
      if (modeB) {
        if (FIRST) {
          const u64 was = iw[gvB];
          if (lane == 0) iw[gvB] = stayB;
          if (lane == compB) acc += stayB - was;
        }
        if (evB) {
          bool moved;
          const int32_t to = lv_decide_vertex((int64_t)gvB, cvB, bb.g, bb.c, stayB, kvv, mv.r, KcvB, szB, szgB, size, &moved);
          if (lane == 0) next[gvB] = to;
          if (lane == compB) cnt += moved ? 1u : 0u;
          if (moved) {
            if (lane == 0) mark_w[gvB] = mv.epoch;
            if (u0 >= 0) mark_w[baseB + (uint32_t)u0] = mv.epoch;
          }
        }
      }
    }
   }
  }
  if (lane < LV_MAX_B) { s_acc[wave][lane] = acc; s_cnt[wave][lane] = cnt; }
  __syncthreads();
  if (threadIdx.x < LV_MAX_B) {
    const int t = threadIdx.x;
    const unsigned c = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
    const size_t at = (size_t)blockIdx.x * LV_MAX_B + t;
    if (FIRST) { part_in[at] = s_acc[0][t] + s_acc[1][t] + s_acc[2][t] + s_acc[3][t]; part_mv[at] = c; }
    else part_mv[at] += c;                      // the same grid in every sub-round of the iteration: the slot is this block's
  }
}

// One wave per listed vertex and copy (LV_SMALL_DEG < entries <= LV_MID_DEG; workgroups of two waves, a 1024-slot table and a list of the
// claimed slots each).  The entries stream through in chunks of 64; up to 512 of them in one pass, beyond that in P passes, pass p taking the
// communities of hash class p.  Nothing is cleared or scanned: the claimed slots are listed, evaluated from the list and handed back empty.
template <bool FIRST>
__global__ __launch_bounds__(128) void k_lv_move_mid(LvG g, const LvCtl* __restrict__ ctl, LvMove mv, int64_t n_list, const int32_t* __restrict__ list,
                                                     const int32_t* __restrict__ comm, const u64* __restrict__ K, const int32_t* __restrict__ size,
                                                     int32_t* __restrict__ next, const int32_t* __restrict__ mark_r, int32_t* __restrict__ mark_w,
                                                     u64* __restrict__ iw, u64* __restrict__ part_in, unsigned* __restrict__ part_mv,
                                                     uint32_t* __restrict__ status) {
  __shared__ int32_t s_key[2][LV_MID_SLOTS];
  __shared__ u64 s_val[2][LV_MID_SLOTS];
  __shared__ uint16_t s_list[2][LV_MID_SLOTS];
  __shared__ u64 s_acc[2][LV_MAX_B];
  __shared__ unsigned s_cnt[2][LV_MAX_B];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t* key = s_key[wave];
  u64* val = s_val[wave];
  uint16_t* claimed = s_list[wave];
  for (int t = lane; t < LV_MID_SLOTS; t += 64) { key[t] = -1; val[t] = 0ull; }
  lv_wave_sync();
  const u64 lt = (1ull << lane) - 1ull;
  u64 acc = 0;
  unsigned cnt = 0;
  for (int64_t item = (int64_t)blockIdx.x * 2 + wave; item < n_list * g.rep; item += (int64_t)gridDim.x * 2) {
    const int b = (int)(item / n_list);
    const int64_t v = list[item - (int64_t)b * n_list];
    const LvSide A = lv_side<FIRST>(g, ctl, mv.s, b, v, comm, K, size);
    if (!A.mode || mark_r[A.gv] < mv.thr) continue;          // (pruning: see LvMove)
    const int64_t lo = g.beg[v], hi = g.end[v];
    if (FIRST && A.mode == 1) {
      u64 t = 0;
      for (int64_t e = lo + lane; e < hi; e += 64) {
        const int32_t u = g.nbr[e];
        if (u != v && comm[A.base + u] == A.cv) t += g.wt[e];
      }
      t = lv_wave_sum(t);
      const u64 was = iw[A.gv];
      if (lane == 0) iw[A.gv] = t;
      if (lane == A.comp) acc += t - was;
      continue;
    }
    const u64 kvv = g.kv[v];
    const double kvd = (double)kvv;
    LvBest best{-INFINITY, INT32_MAX};
    u64 stay = 0;                                  // this lane's entries inside the vertex's own community (never through the table: see k_lv_move_small)
    const uint32_t P = (uint32_t)((hi - lo + LV_MID_SLOTS / 2 - 1) / (LV_MID_SLOTS / 2));
    for (uint32_t p = 0; p < (P ? P : 1u); ++p) {
      int n_claimed = 0;
      // four chunks of 64 entries a step: their neighbour ids, then their communities, are loaded together (a chunk at a time every step waited
      // for two dependent round trips, three to eight times a row); every lane stays in the loop: the ballots below are the whole wave's
      for (int64_t e = lo + lane; e - lane < hi; e += 256) {
        int32_t uu[4] = {-1, -1, -1, -1}, cc[4] = {-1, -1, -1, -1};
        u64 ww[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (e + 64 * h < hi) { uu[h] = g.nbr[e + 64 * h]; ww[h] = g.wt[e + 64 * h]; }
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (uu[h] >= 0 && uu[h] != v) cc[h] = comm[A.base + uu[h]];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (e - lane + 64 * h >= hi) break;                          // uniform: the row ended with an earlier chunk
          bool own = false;
          int slot = -1;
          const int32_t c = cc[h];
          if (c >= 0) {
            if (c == A.cv) { if (p == 0) stay += ww[h]; }
            else if (P <= 1 || (lv_hash((uint32_t)c) >> 13) % P == p) {
              slot = lv_insert<LV_MID_SLOTS>(key, val, c, ww[h], &own);
              if (slot < 0) atomicOr(status, GFICF_ST_TOO_DENSE);     // a hash class that overflows the table
            }
          }
          const u64 m = __ballot(own);
          if (own) claimed[n_claimed + __popcll(m & lt)] = (uint16_t)slot;
          n_claimed += __popcll(m);
        }
      }
      lv_wave_sync();
      for (int i = lane; i < n_claimed; i += 64) {
        const int slot = claimed[i];
        const int32_t c = key[slot];
        const double gain = (double)val[slot] - kvd * (double)K[c] * mv.r;
        key[slot] = -1; val[slot] = 0ull;
        if (lv_better(gain, c, best.g, best.c)) { best.g = gain; best.c = c; }
      }
      lv_wave_sync();
    }
    best = lv_wave_best(best);
    stay = lv_wave_sum(stay);
    bool moved;
    const int32_t to = lv_decide_vertex(A.gv, A.cv, best.g, best.c, stay, kvv, mv.r, A.Kcv, A.szcv, A.szgv, size, &moved);
    if (lane == 0) next[A.gv] = to;
    if (FIRST) {
      const u64 was = iw[A.gv];
      if (lane == 0) iw[A.gv] = stay;
      if (lane == A.comp) acc += stay - was;
    }
    if (lane == A.comp) cnt += moved ? 1u : 0u;
    if (moved) {
      if (lane == 0) mark_w[A.gv] = mv.epoch;
      for (int64_t e = lo + lane; e < hi; e += 64) mark_w[A.base + g.nbr[e]] = mv.epoch;
    }
  }
  if (lane < LV_MAX_B) { s_acc[wave][lane] = acc; s_cnt[wave][lane] = cnt; }
  __syncthreads();
  if (threadIdx.x < LV_MAX_B) {
    const int t = threadIdx.x;
    const size_t at = (size_t)blockIdx.x * LV_MAX_B + t;
    if (FIRST) { part_in[at] = s_acc[0][t] + s_acc[1][t]; part_mv[at] = s_cnt[0][t] + s_cnt[1][t]; }
    else part_mv[at] += s_cnt[0][t] + s_cnt[1][t];
  }
}

// One workgroup per listed vertex and copy (more than LV_MID_DEG entries): an 8192-slot table, P passes over the entries.
template <int SLOTS, bool FIRST>
__global__ __launch_bounds__(256) void k_lv_move_big(LvG g, const LvCtl* __restrict__ ctl, LvMove mv, int large, int64_t n_list, const int32_t* __restrict__ big,
                                                     const int32_t* __restrict__ comm, const u64* __restrict__ K, const int32_t* __restrict__ size,
                                                     int32_t* __restrict__ next, const int32_t* __restrict__ mark_r, int32_t* __restrict__ mark_w,
                                                     u64* __restrict__ iw, u64* __restrict__ part_in, unsigned* __restrict__ part_mv,
                                                     uint32_t* __restrict__ status) {
  extern __shared__ unsigned char s_raw[];
  u64* val = (u64*)s_raw;
  int32_t* key = (int32_t*)(val + SLOTS);
  __shared__ int s_moved;
  __shared__ double s_g[256];
  __shared__ u64 s_w[4], s_in[4];
  __shared__ int32_t s_c[256];
  __shared__ u64 s_acc[LV_MAX_B];
  __shared__ unsigned s_cnt[LV_MAX_B];
  const int tid = threadIdx.x;
  if (tid < LV_MAX_B) { s_acc[tid] = 0ull; s_cnt[tid] = 0u; }
  __syncthreads();
  const int32_t* list = large ? big + (g.nb - n_list) : big;
  for (int64_t item = blockIdx.x; item < n_list * g.rep; item += gridDim.x) {
    const int b = (int)(item / n_list);
    const int64_t v = list[item - (int64_t)b * n_list];
    const int comp = g.vcomp ? (int)g.vcomp[v] : b;
    const LvComp& C = ctl->c[comp];
    if (!lv_runs(C, mv.s, FIRST)) continue;                                       // uniform per workgroup
    const int64_t base = (int64_t)b * g.nb, gv = base + v;
    const bool in_class = C.S == 1 || (int)(lv_hash((uint32_t)(gv - C.v0) + C.seed) % (uint32_t)C.S) == mv.s;
    if (!FIRST && !in_class) continue;
    if (mark_r[gv] < mv.thr) continue;                                            // (pruning: see LvMove)
    const int64_t lo = g.beg[v], hi = g.end[v];
    const int32_t cv = comm[gv];
    const u64 kvv = g.kv[v];
    if (FIRST && !in_class) {
      u64 t = 0;
      for (int64_t e = lo + tid; e < hi; e += 256) {
        const int32_t u = g.nbr[e];
        if (u != v && comm[base + u] == cv) t += g.wt[e];
      }
      t = lv_wave_sum(t);
      if ((tid & 63) == 0) s_in[tid >> 6] = t;
      __syncthreads();
      if (tid == 0) {
        const u64 now = s_in[0] + s_in[1] + s_in[2] + s_in[3];
        s_acc[comp] += now - iw[gv];
        iw[gv] = now;
      }
      __syncthreads();
      continue;
    }
    const double kvd = (double)kvv;
    double bg = -INFINITY;
    u64 stay_w = 0;
    int32_t bc = INT32_MAX;
    // A vertex with more entries than the table comfortably holds (every entry can be its own community) is done in
    // P passes over its entries, pass p taking the communities of hash class p: about deg / P <= SLOTS / 2 of them at a time.
    const uint32_t P = (uint32_t)((hi - lo + SLOTS / 2 - 1) / (SLOTS / 2));
    for (uint32_t p = 0; p < (P ? P : 1u); ++p) {
      for (int t = tid; t < SLOTS; t += 256) { key[t] = -1; val[t] = 0ull; }
      __syncthreads();
      for (int64_t e = lo + tid; e < hi; e += 256) {
        const int32_t u = g.nbr[e];
        if (u == v) continue;
        const int32_t c = comm[base + u];
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
        const u64 w = val[t];
        if (c == cv) { stay_w = w; continue; }
        const double gain = (double)w - kvd * (double)K[c] * mv.r;
        if (lv_better(gain, c, bg, bc)) { bg = gain; bc = c; }
      }
      __syncthreads();
    }
    for (int d = 32; d > 0; d >>= 1) { const u64 o = __shfl_xor(stay_w, d); stay_w = o > stay_w ? o : stay_w; }
    if ((tid & 63) == 0) s_w[tid >> 6] = stay_w;
    s_g[tid] = bg; s_c[tid] = bc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
      if (tid < d && lv_better(s_g[tid + d], s_c[tid + d], s_g[tid], s_c[tid])) { s_g[tid] = s_g[tid + d]; s_c[tid] = s_c[tid + d]; }
      __syncthreads();
    }
    if (tid == 0) {
      bg = s_g[0]; bc = s_c[0];
      stay_w = s_w[0];
      for (int t = 1; t < 4; ++t) stay_w = s_w[t] > stay_w ? s_w[t] : stay_w;
      bool moved;
      next[gv] = lv_decide_vertex(gv, cv, bg, bc, stay_w, kvv, mv.r, K[cv], size[cv], size[gv], size, &moved);
      s_cnt[comp] += moved ? 1u : 0u;
      if (FIRST) { s_acc[comp] += stay_w - iw[gv]; iw[gv] = stay_w; }
      s_moved = moved ? 1 : 0;
      if (moved) mark_w[gv] = mv.epoch;
    }
    __syncthreads();
    if (s_moved)
      for (int64_t e = lo + tid; e < hi; e += 256) mark_w[base + g.nbr[e]] = mv.epoch;
    __syncthreads();
  }
  __syncthreads();
  if (tid < LV_MAX_B) {
    const size_t at = (size_t)blockIdx.x * LV_MAX_B + tid;
    if (FIRST) { part_in[at] = s_acc[tid]; part_mv[at] = s_cnt[tid]; }
    else part_mv[at] += s_cnt[tid];
  }
}

// Start of an iteration: the sum of the squared community totals of every running component, in a fixed order (LV_SQ_BLOCKS slices
// of its communities, a fixed tree inside each, the slices added up in slice order by k_lv_decide), and a copy of the totals and sizes
// as they are now (parity = iteration & 1): what an undo of the NEXT iteration's moves goes back to.
__global__ __launch_bounds__(256) void k_lv_pre(const LvCtl* __restrict__ ctl, int parity, const u64* __restrict__ K, const int32_t* __restrict__ size,
                                                u64* __restrict__ snapK, int32_t* __restrict__ snapS, double* __restrict__ part_sq) {
  __shared__ double s_p[256];
  const int b = blockIdx.y;
  const LvComp& C = ctl->c[b];
  if (C.level_done) return;
  (void)parity;
  const int64_t per = (C.n + LV_SQ_BLOCKS - 1) / LV_SQ_BLOCKS;
  const int64_t lo = C.v0 + (int64_t)blockIdx.x * per, hi = lo + per < C.v0 + C.n ? lo + per : C.v0 + C.n;
  double s = 0.0;
  for (int64_t c = lo + threadIdx.x; c < hi; c += 256) {
    const u64 kc = K[c];
    snapK[c] = kc; snapS[c] = size[c];
    const double k = (double)kc;
    s += k * k;
  }
  s_p[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_p[threadIdx.x] += s_p[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) part_sq[b * LV_SQ_BLOCKS + blockIdx.x] = s_p[0];
}

// The decision of iteration `it` for every component (one block).  The partial sums of the sub-round-0 kernels give the internal weight
// of the labels as the iteration found them, i.e. the result of iteration it - 1, and the number of vertices that moved in it:
//   it == 0: Q of the level's starting labels; go on.           moved == 0: the level has converged.
//   Q < Q before: simultaneous moves made it worse -> undo iteration it - 1, the level ends.
//   else accept; a gain below 1e-7 ends the level.              LV_MAX_ITERS iterations: the level ends.
__global__ __launch_bounds__(1024) void k_lv_decide(LvCtl* ctl, LvParts pt, int nb_small, int nb_mid, int nb_large, int it, double two_w, double q_coef,
                                                   LvHost* __restrict__ host) {
  __shared__ u64 s_in[16][LV_MAX_B];
  __shared__ unsigned s_mv[16][LV_MAX_B];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int B = ctl->B;
  const int pv = (it & 1) ^ 1;                   // the moved counts of the iteration before this one
  {
    // element e of a slab = (row e / 16, component e % 16): thread t takes elements t, t + 1024, ... — always component t % 16, coalesced
    u64 a = 0;
    unsigned m = 0;
    // (eight loads in flight per slab and thread: one load behind the other, the 0.5 MB of partial sums took 21 us of a 29-launch critical path)
    auto slab = [&](const u64* __restrict__ in, const unsigned* __restrict__ mvp, int n) {
      for (int e0 = tid; e0 < n; e0 += 8 * 1024) {
        u64 av[8];
        unsigned mv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = e0 + j * 1024;
          av[j] = e < n ? in[e] : 0ull;
          mv8[j] = (it && e < n) ? mvp[e] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { a += av[j]; m += mv8[j]; }
      }
    };
    slab(pt.in_small, pt.mv_small[pv], nb_small * LV_MAX_B);
    slab(pt.in_mid, pt.mv_mid[pv], nb_mid * LV_MAX_B);
    slab(pt.in_large, pt.mv_large[pv], nb_large * LV_MAX_B);
    for (int d = 32; d >= LV_MAX_B; d >>= 1) { a += __shfl_xor(a, d); m += __shfl_xor(m, d); }      // lanes l, l + 16, l + 32, l + 48 hold component l % 16
    if (lane < LV_MAX_B) { s_in[wave][lane] = a; s_mv[wave][lane] = m; }
  }
  __syncthreads();
  __shared__ int s_cont[LV_MAX_B];
  if (tid < LV_MAX_B) {
    int cont = 0;
    if (tid < B) {
      LvComp& C = ctl->c[tid];
      if (C.level_done) {
        C.action = LV_IDLE;
      } else {
        u64 in_now = C.in_run;                       // + what the sub-round-0 kernels found changed (differences: the sums wrap as they should)
        unsigned moved = 0;
        for (int w = 0; w < 16; ++w) { in_now += s_in[w][tid]; moved += s_mv[w][tid]; }
        C.in_run = in_now;
        double sq = 0.0;
        for (int j = 0; j < LV_SQ_BLOCKS; ++j) sq += pt.sq[tid * LV_SQ_BLOCKS + j];
        const double q = ((double)(in_now + C.self_w)) / two_w - q_coef * sq;          // sq = sum of squared community totals (fixed point)
        int action;
        if (C.iter == 0) { C.q_prev = q; C.in_w = in_now; action = LV_CONTINUE; }
        else if (moved == 0) action = LV_STOP_KEEP;
        else if (q < C.q_prev) action = LV_STOP_UNDO;
        else {
          C.level_moved = 1;
          C.in_w = in_now;
          const bool small_gain = q - C.q_prev < 1e-7;
          C.q_prev = q;
          action = small_gain ? LV_STOP_KEEP : LV_CONTINUE;
        }
        if (action == LV_CONTINUE && C.iter >= LV_MAX_ITERS) action = LV_STOP_KEEP;
        if (action != LV_CONTINUE) C.level_done = 1;
        C.action = action;
        C.iter += 1;
        cont = action == LV_CONTINUE;
        host->q_iter[tid] = q;
        host->mv_iter[tid] = moved;
      }
    }
    s_cont[tid] = cont;
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int b = 0; b < LV_MAX_B; ++b) n += s_cont[b];
    host->n_cont[it] = n;
    __threadfence_system();
  }
}

// Applies the sub-round's moves to the labels, totals and sizes.  Sub-round 0 also carries out the iteration's decision: go on -> keep a
// copy of the labels as they are (what an undo of this iteration's moves goes back to); undo -> the labels, totals and sizes of one
// iteration ago come back.
__global__ __launch_bounds__(256) void k_lv_apply(LvG g, const LvCtl* __restrict__ ctl, int s, int parity, int32_t* __restrict__ comm,
                                                  const int32_t* __restrict__ next, u64* __restrict__ K, int32_t* __restrict__ size,
                                                  int32_t* __restrict__ snapc0, int32_t* __restrict__ snapc1, const u64* __restrict__ snapK_prev,
                                                  const int32_t* __restrict__ snapS_prev, int32_t* __restrict__ mark_r, const int32_t* __restrict__ mark_w) {
  int32_t* const snap_now = parity ? snapc1 : snapc0;
  const int32_t* const snap_prev = parity ? snapc0 : snapc1;
  for (int64_t gv = (int64_t)blockIdx.x * 256 + threadIdx.x; gv < g.n; gv += (int64_t)gridDim.x * 256) {
    const int64_t b = g.rep > 1 ? gv / g.nb : 0;
    const int comp = g.vcomp ? (int)g.vcomp[gv] : (int)b;
    const LvComp& C = ctl->c[comp];
    const int action = C.action;
    if (C.level_done && action == LV_IDLE) continue;
    // the stamps the sub-round's move kernels wrote become readable: those kernels read mark_r and write mark_w only, so what a vertex
    // sees never depends on when a neighbour's store lands
    { const int32_t mw = mark_w[gv]; if (mw > mark_r[gv]) mark_r[gv] = mw; }
    if (s == 0) {
      if (action == LV_CONTINUE) snap_now[gv] = comm[gv];
      else if (action == LV_STOP_UNDO) { comm[gv] = snap_prev[gv]; K[gv] = snapK_prev[gv]; size[gv] = snapS_prev[gv]; }
    }
    if (action != LV_CONTINUE || s >= C.S) continue;
    if (C.S != 1 && (int)(lv_hash((uint32_t)(gv - C.v0) + C.seed) % (uint32_t)C.S) != s) continue;
    const int32_t a = comm[gv], to = next[gv];
    if (a == to) continue;
    const u64 k = g.kv[gv - b * g.nb];
    atomicAdd(&K[a], 0ull - k);
    atomicAdd(&K[to], k);
    atomicSub(&size[a], 1);
    atomicAdd(&size[to], 1);
    comm[gv] = to;
  }
}

// ---- renumbering
__global__ __launch_bounds__(256) void k_lv_used(int64_t n, const int32_t* __restrict__ size, int64_t* __restrict__ flag) {
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c <= n; c += (int64_t)gridDim.x * 256) flag[c] = c < n && size[c] > 0 ? 1 : 0;
}

// After the scan: every component's first new id and number of communities; does its descent go on?
//   (reference loop: runLouvainAlgorithm recurses while the reduced network is smaller, src/ModularityOptimizer.cpp:594-612)
// refine: a refinement level of algorithm 2 (no descent, the labels are renumbered only).
__global__ void k_lv_ctl_renumbered(LvCtl* ctl, const int64_t* __restrict__ newid, int64_t n_union, int seeded, int refine, int algorithm, int level,
                                    LvHost* __restrict__ host) {
  const int b = threadIdx.x;
  if (b == 0) { ctl->n_union2 = newid[n_union]; host->n_union2 = newid[n_union]; }
  if (b >= LV_MAX_B) return;
  LvComp& C = ctl->c[b];
  C.newbase = newid[C.v0];
  C.n2 = C.live ? newid[C.v0 + C.n] - newid[C.v0] : 0;
  if (C.live) {
    C.q_final = C.q_prev;
    C.n_labels = C.n2;
    if (!refine) {
      C.any_move |= C.level_moved;
      const bool done = C.n2 == C.n || C.n2 <= 1 || (!C.level_moved && !(seeded && C.n2 < C.n));      // nothing merged: the descent is done
      C.cont = done ? 0 : 1;
      if (C.cont) {
        C.self_w += C.in_w;
        if (algorithm == 2 && C.n_saved == level && level < LV_MAX_SAVED) C.saved_n[++C.n_saved] = C.n2;
      }
    }
  } else {
    C.cont = 0;
  }
  host->c[b].n2 = C.n2; host->c[b].q_prev = C.q_prev; host->c[b].cont = C.cont; host->c[b].any_move = C.any_move;
  host->c[b].level_moved = C.level_moved; host->c[b].live = C.live;
}

// lab[b][v] = new LOCAL id (new id - the component's first) of the community of the level's vertex that original vertex v of start b
// maps to: src == NULL -> v itself (level 0), else src[b][v] (a local vertex id of the level; may be lab itself: one read, one write per thread)
__global__ __launch_bounds__(256) void k_lv_relabel(int64_t N, const LvCtl* __restrict__ ctl, const int32_t* src, const int32_t* __restrict__ comm,
                                                    const int64_t* __restrict__ newid, int32_t* lab) {
  const int b = blockIdx.y;
  const LvComp& C = ctl->c[b];
  if (!C.live) return;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < N; v += (int64_t)gridDim.x * 256) {
    const size_t at = (size_t)b * (size_t)N + (size_t)v;
    const int64_t x = C.v0 + (src ? (int64_t)src[at] : v);
    lab[at] = (int32_t)(newid[comm[x]] - C.newbase);
  }
}

// ---- reduction: the level's graph by the labels newid[comm[.]] -> the next level's (a multigraph in pointerB / pointerE form)
// Row capacities (the entries of a community's members: no row can need more), vertex weights and components of the new vertices.
// cap must be zero.  Few communities: the capacities are summed in LDS first (see k_lv_accum).
__global__ __launch_bounds__(256) void k_lv_rowcap(LvG g, const LvCtl* __restrict__ ctl, const int32_t* __restrict__ comm, const int64_t* __restrict__ newid,
                                                   const u64* __restrict__ K, const int32_t* __restrict__ size, int64_t* __restrict__ cap,
                                                   u64* __restrict__ kv2, uint8_t* __restrict__ vcomp2) {
  __shared__ u64 s_cap[LV_ACC_BINS];
  const bool binned = ctl->n_union2 <= LV_ACC_BINS;
  if (binned) {
    for (int t = threadIdx.x; t < LV_ACC_BINS; t += 256) s_cap[t] = 0ull;
    __syncthreads();
  }
  for (int64_t gv = (int64_t)blockIdx.x * 256 + threadIdx.x; gv < g.n; gv += (int64_t)gridDim.x * 256) {
    const int64_t b = g.rep > 1 ? gv / g.nb : 0;
    const int64_t v = gv - b * g.nb;
    const int comp = g.vcomp ? (int)g.vcomp[gv] : (int)b;
    const LvComp& C = ctl->c[comp];
    if (!C.live) continue;                                // (not numbered; and its own label may coincide with a live community's id when the labels are a rebuilt level's ids)
    const int32_t c = comm[gv];
    if (size[c] <= 0) continue;
    // the new vertex its community becomes: weight and component written by every member alike (the members of a community belong to
    // one component; a community id need not lie in its component's vertex range: the rebuilt levels of algorithm 2)
    const int64_t c2 = newid[c];
    kv2[c2] = C.cont ? K[c] : 0ull;
    vcomp2[c2] = (uint8_t)comp;
    if (!C.cont) continue;
    const u64 deg = (u64)(g.end[v] - g.beg[v]);
    if (deg == 0) continue;
    if (binned) atomicAdd(&s_cap[c2], deg);
    else atomicAdd((u64*)&cap[c2], deg);
  }
  if (binned) {
    __syncthreads();
    for (int t = threadIdx.x; t < LV_ACC_BINS; t += 256)
      if (s_cap[t]) atomicAdd((u64*)&cap[t], s_cap[t]);
  }
}

// One wave per vertex and copy (at most LV_SMALL_DEG entries): the entries summed per neighbouring NEW community; one entry per
// community is appended to the row of the vertex's own new community (cur: the rows' fill counts).  Entries inside the community
// are dropped (their weight is carried as the level's internal weight).
__global__ __launch_bounds__(256) void k_lv_emit_small(LvG g, const LvCtl* __restrict__ ctl, const int32_t* __restrict__ comm, const int64_t* __restrict__ newid,
                                                       const int64_t* __restrict__ beg2, int32_t* __restrict__ cur, int32_t* __restrict__ nbr2,
                                                       u64* __restrict__ wt2) {
  __shared__ int32_t s_key[4][LV_SMALL_SLOTS];
  __shared__ u64 s_val[4][LV_SMALL_SLOTS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t* key = s_key[wave];
  u64* val = s_val[wave];
  for (int t = lane; t < LV_SMALL_SLOTS; t += 64) { key[t] = -1; val[t] = 0ull; }
  lv_wave_sync();
  const u64 lt = (1ull << lane) - 1ull;
  for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < g.nb; v += (int64_t)gridDim.x * 4) {
    const int64_t lo = g.beg[v], hi = g.end[v];
    if (hi - lo > LV_SMALL_DEG || hi == lo) continue;
    const int64_t e0 = lo + lane, e1 = e0 + 64;
    int32_t u0 = -1, u1 = -1;
    u64 w0 = 0, w1 = 0;
    if (e0 < hi) { u0 = g.nbr[e0]; w0 = g.wt[e0]; if (u0 == v) u0 = -1; }
    if (e1 < hi) { u1 = g.nbr[e1]; w1 = g.wt[e1]; if (u1 == v) u1 = -1; }
    for (int b = 0; b < g.rep; ++b) {
      const int comp = g.vcomp ? (int)g.vcomp[v] : b;
      if (!ctl->c[comp].cont) continue;
      const int64_t base = (int64_t)b * g.nb;
      const int32_t cv = (int32_t)newid[comm[base + v]];
      int32_t c0 = u0 >= 0 ? (int32_t)newid[comm[base + u0]] : -1, c1 = u1 >= 0 ? (int32_t)newid[comm[base + u1]] : -1;
      if (c0 == cv) c0 = -1;
      if (c1 == cv) c1 = -1;
      int slot0 = -1, slot1 = -1;
      bool own0 = false, own1 = false;
      if (c0 >= 0) {
        uint32_t h = lv_hash((uint32_t)c0) & (LV_SMALL_SLOTS - 1);
        for (;;) {
          const int32_t old = atomicCAS(&key[h], -1, c0);
          if (old == -1) { own0 = true; break; }
          if (old == c0) break;
          h = (h + 1) & (LV_SMALL_SLOTS - 1);
        }
        slot0 = (int)h;
        atomicAdd(&val[h], w0);
      }
      if (c1 >= 0) {
        uint32_t h = lv_hash((uint32_t)c1) & (LV_SMALL_SLOTS - 1);
        for (;;) {
          const int32_t old = atomicCAS(&key[h], -1, c1);
          if (old == -1) { own1 = true; break; }
          if (old == c1) break;
          h = (h + 1) & (LV_SMALL_SLOTS - 1);
        }
        slot1 = (int)h;
        atomicAdd(&val[h], w1);
      }
      lv_wave_sync();
      const u64 m0 = __ballot(own0), m1 = __ballot(own1);
      const int n0 = __popcll(m0), total = n0 + __popcll(m1);
      int32_t p = 0;
      if (lane == 0 && total) p = atomicAdd(&cur[cv], total);
      p = __shfl(p, 0);
      const int64_t row = beg2[cv];
      if (own0) {
        const int64_t at = row + p + __popcll(m0 & lt);
        nbr2[at] = c0; wt2[at] = val[slot0];
        key[slot0] = -1; val[slot0] = 0ull;
      }
      if (own1) {
        const int64_t at = row + p + n0 + __popcll(m1 & lt);
        nbr2[at] = c1; wt2[at] = val[slot1];
        key[slot1] = -1; val[slot1] = 0ull;
      }
      lv_wave_sync();
    }
  }
}

template <int SLOTS>
__global__ __launch_bounds__(256) void k_lv_emit_big(LvG g, const LvCtl* __restrict__ ctl, int large, int64_t n_list, const int32_t* __restrict__ big,
                                                     const int32_t* __restrict__ comm, const int64_t* __restrict__ newid, const int64_t* __restrict__ beg2,
                                                     int32_t* __restrict__ cur, int32_t* __restrict__ nbr2, u64* __restrict__ wt2,
                                                     uint32_t* __restrict__ status) {
  extern __shared__ unsigned char s_raw[];
  u64* val = (u64*)s_raw;
  int32_t* key = (int32_t*)(val + SLOTS);
  __shared__ int s_n;
  __shared__ int32_t s_base;
  const int tid = threadIdx.x;
  const int32_t* list = large ? big + (g.nb - n_list) : big;
  for (int64_t item = blockIdx.x; item < n_list * g.rep; item += gridDim.x) {
    const int b = (int)(item / n_list);
    const int64_t v = list[item - (int64_t)b * n_list];
    const int comp = g.vcomp ? (int)g.vcomp[v] : b;
    if (!ctl->c[comp].cont) continue;
    const int64_t base = (int64_t)b * g.nb;
    const int64_t lo = g.beg[v], hi = g.end[v];
    const int32_t cv = (int32_t)newid[comm[base + v]];
    const int64_t row = beg2[cv];
    const uint32_t P = (uint32_t)((hi - lo + SLOTS / 2 - 1) / (SLOTS / 2));
    for (uint32_t p = 0; p < (P ? P : 1u); ++p) {
      for (int t = tid; t < SLOTS; t += 256) { key[t] = -1; val[t] = 0ull; }
      if (tid == 0) s_n = 0;
      __syncthreads();
      for (int64_t e = lo + tid; e < hi; e += 256) {
        const int32_t u = g.nbr[e];
        if (u == v) continue;
        const int32_t c = (int32_t)newid[comm[base + u]];
        if (c == cv) continue;
        const uint32_t hc = lv_hash((uint32_t)c);
        if (P > 1 && (hc >> 13) % P != p) continue;
        uint32_t h = hc & (SLOTS - 1);
        int probes = 0;
        for (;;) {
          const int32_t old = atomicCAS(&key[h], -1, c);
          if (old == -1 || old == c) { atomicAdd(&val[h], g.wt[e]); break; }
          h = (h + 1) & (SLOTS - 1);
          if (++probes >= SLOTS) { atomicOr(status, GFICF_ST_TOO_DENSE); break; }
        }
      }
      __syncthreads();
      int mine = 0;
      for (int t = tid; t < SLOTS; t += 256) mine += key[t] >= 0 ? 1 : 0;
      const int rank0 = mine ? atomicAdd(&s_n, mine) : 0;
      __syncthreads();
      if (tid == 0) s_base = s_n ? atomicAdd(&cur[cv], s_n) : 0;
      __syncthreads();
      int64_t at = row + s_base + rank0;
      for (int t = tid; t < SLOTS; t += 256)
        if (key[t] >= 0) { nbr2[at] = key[t]; wt2[at] = val[t]; ++at; }
      __syncthreads();
    }
  }
}

// end2 = beg2 + fill count
__global__ __launch_bounds__(256) void k_lv_finish_rows(const int64_t* __restrict__ n_dev, const int64_t* __restrict__ beg2, const int32_t* __restrict__ cur,
                                                        int64_t* __restrict__ end2) {
  const int64_t n = *n_dev;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) end2[c] = beg2[c] + cur[c];
}

// the list counts of the next level's graph, where the host reads them after the level's one synchronisation
__global__ void k_lv_ctl_publish(const LvCtl* __restrict__ ctl, LvHost* __restrict__ host) {
  if (threadIdx.x == 0) { host->n_mid = ctl->n_mid; host->n_large = ctl->n_large; __threadfence_system(); }
}

// ---- algorithm 2: the levels on the way back up
// The participants of refinement level `level` (starts that moved in this pass and saved that level), their vertex ranges (prefix of
// the saved sizes), and — level >= 1 — the base their level vertices get as labels of the finest graph (seed_base).
__global__ void k_lv_ctl_refine(LvCtl* ctl, int level, int64_t N) {
  __shared__ int64_t s_n[LV_MAX_B];
  const int b = threadIdx.x;
  const int B = ctl->B;
  if (b < LV_MAX_B) {
    LvComp& C = ctl->c[b];
    C.live = b < B && !C.finished && C.any_move && C.n_saved >= level;
    s_n[b] = C.live ? (level ? C.saved_n[level] : N) : 0;
  }
  __syncthreads();
  if (b < LV_MAX_B) {
    LvComp& C = ctl->c[b];
    int64_t base = 0;
    for (int t = 0; t < b; ++t) base += s_n[t];
    if (level == 0) { C.v0 = b < B ? (int64_t)b * N : 0; C.n = b < B ? N : 0; }
    else { C.v0 = base; C.n = s_n[b]; }
    C.seed_base = C.v0;
    C.self_w = 0;                          // level 0: nothing is folded into the vertices; a rebuilt level: set by k_lv_ctl_self
    C.newbase = C.v0; C.n2 = C.n;          // the level's vertices ARE the new ids of the rebuilding reduction
    C.cont = C.live;                       // ... which every participant takes part in
    if (b == LV_MAX_B - 1) ctl->n_union2 = base + s_n[b];
  }
}

// internal weight of given labels on the level's graph, per component (per-block partial sums in the slots of the small path)
__global__ __launch_bounds__(256) void k_lv_internal(LvG g, const LvCtl* __restrict__ ctl, const int32_t* __restrict__ comm, u64* __restrict__ part_in) {
  __shared__ u64 s_acc[4][LV_MAX_B];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u64 acc = 0;
  for (int64_t v = (int64_t)blockIdx.x * 4 + wave; v < g.nb; v += (int64_t)gridDim.x * 4) {
    for (int b = 0; b < g.rep; ++b) {
      const int comp = g.vcomp ? (int)g.vcomp[v] : b;
      if (!ctl->c[comp].live) continue;
      const int64_t base = (int64_t)b * g.nb;
      const int32_t cv = comm[base + v];
      u64 s = 0;
      for (int64_t e = g.beg[v] + lane; e < g.end[v]; e += 64) {
        const int32_t u = g.nbr[e];
        if (u != v && comm[base + u] == cv) s += g.wt[e];
      }
      s = lv_wave_sum(s);
      if (lane == comp) acc += s;
    }
  }
  if (lane < LV_MAX_B) s_acc[wave][lane] = acc;
  __syncthreads();
  if (threadIdx.x < LV_MAX_B) part_in[(size_t)blockIdx.x * LV_MAX_B + threadIdx.x] = s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x];
}

// self_w of the refinement level = the sum of those partial sums
__global__ __launch_bounds__(256) void k_lv_ctl_self(LvCtl* ctl, const u64* __restrict__ part_in, int nb) {
  __shared__ u64 s_in[4][LV_MAX_B];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int b = 0; b < LV_MAX_B; ++b) {
    u64 a = 0;
    for (int r = tid; r < nb; r += 256) a += part_in[(size_t)r * LV_MAX_B + b];
    a = lv_wave_sum(a);
    if (lane == 0) s_in[wave][b] = a;
  }
  __syncthreads();
  if (tid < LV_MAX_B) ctl->c[tid].self_w = ctl->c[tid].live ? s_in[0][tid] + s_in[1][tid] + s_in[2][tid] + s_in[3][tid] : 0ull;
}

// seedl[x] = the label of the original vertices that make up level vertex x (they all carry the same one)
__global__ __launch_bounds__(256) void k_lv_seed(int64_t N, const LvCtl* __restrict__ ctl, const int32_t* __restrict__ top, const int32_t* __restrict__ lab,
                                                 int32_t* __restrict__ seedl) {
  const int b = blockIdx.y;
  const LvComp& C = ctl->c[b];
  if (!C.live) return;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < N; v += (int64_t)gridDim.x * 256) {
    const size_t at = (size_t)b * (size_t)N + (size_t)v;
    seedl[C.v0 + top[at]] = lab[at];
  }
}

__global__ __launch_bounds__(256) void k_lv_iota(int64_t n, int64_t* __restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v <= n; v += (int64_t)gridDim.x * 256) out[v] = v;
}

// ---- final numbering: clusters by decreasing size, ties by id (Clustering::orderClustersByNNodes, reference :132-158)
__global__ __launch_bounds__(256) void k_lv_count(int64_t n, int64_t C, const int32_t* __restrict__ lab, int32_t* __restrict__ cnt) {
  __shared__ int32_t s_n[LV_ACC_BINS];
  const bool binned = C <= LV_ACC_BINS;
  if (binned) {
    for (int t = threadIdx.x; t < C; t += 256) s_n[t] = 0;
    __syncthreads();
  }
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
    if (binned) atomicAdd(&s_n[lab[v]], 1);
    else atomicAdd(&cnt[lab[v]], 1);
  }
  if (binned) {
    __syncthreads();
    for (int t = threadIdx.x; t < C; t += 256)
      if (s_n[t]) atomicAdd(&cnt[t], s_n[t]);
  }
}

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

__global__ __launch_bounds__(256) void k_lv_iota32(int64_t n, int32_t* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v < n) out[v] = (int32_t)v;
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

struct LvLevel {          // a coarse graph's arrays (capacity: the union's entries)
  int64_t* beg; int64_t* end; int32_t* nbr; u64* wt; u64* kv; uint8_t* vcomp; int32_t* big;
};

struct LvWs {
  u64* wt0; u64* kv0; int32_t* big0;
  LvLevel lvl[2];
  int32_t *comm, *next, *snapc[2], *size, *snapS[2], *cur, *lab, *seedl, *tops, *best, *cnt, *rank, *mark_r, *mark_w;
  u64 *K, *snapK[2], *iw;
  int64_t* flag;            // n_union + 1 entries: the renumbering scan
  u64 *keys_a, *vals_a, *keys_b, *vals_b;      // final numbering (N entries)
  void* sort_tmp; size_t sort_tmp_bytes;
  LvParts pt;
  LvCtl* ctl;
  u64* scalars;             // [0] 2W, [5] largest weight
};

// the workspace of B starts run together
static size_t lv_carve(LvWs* w, void* base, int64_t N, int64_t nnz, int B) {
  Bump b{(char*)base, 0, 0};
  const size_t n = (size_t)(N > 0 ? N : 1), m = (size_t)(nnz > 0 ? nnz : 1);
  const size_t nu = n * (size_t)B, mu = m * (size_t)B;
  LvWs d;
  d.wt0 = b.take<u64>(m); d.kv0 = b.take<u64>(n); d.big0 = b.take<int32_t>(n);
  for (int i = 0; i < 2; ++i) {
    d.lvl[i].beg = b.take<int64_t>(nu + 1); d.lvl[i].end = b.take<int64_t>(nu + 1); d.lvl[i].nbr = b.take<int32_t>(mu); d.lvl[i].wt = b.take<u64>(mu);
    d.lvl[i].kv = b.take<u64>(nu); d.lvl[i].vcomp = b.take<uint8_t>(nu); d.lvl[i].big = b.take<int32_t>(nu);
  }
  d.comm = b.take<int32_t>(nu); d.next = b.take<int32_t>(nu); d.size = b.take<int32_t>(nu); d.cur = b.take<int32_t>(nu);
  d.lab = b.take<int32_t>(nu); d.seedl = b.take<int32_t>(nu);
  for (int i = 0; i < 2; ++i) { d.snapc[i] = b.take<int32_t>(nu); d.snapS[i] = b.take<int32_t>(nu); d.snapK[i] = b.take<u64>(nu); }
  d.tops = b.take<int32_t>(nu * LV_MAX_SAVED);
  d.best = b.take<int32_t>(n); d.cnt = b.take<int32_t>(n); d.rank = b.take<int32_t>(n);
  d.K = b.take<u64>(nu); d.iw = b.take<u64>(nu);
  d.mark_r = b.take<int32_t>(nu); d.mark_w = b.take<int32_t>(nu);
  d.flag = b.take<int64_t>(nu + 1);
  d.keys_a = b.take<u64>(n); d.vals_a = b.take<u64>(n); d.keys_b = b.take<u64>(n); d.vals_b = b.take<u64>(n);
  d.sort_tmp_bytes = lv_sort_tmp_bytes((int64_t)n);
  d.sort_tmp = b.take<char>(d.sort_tmp_bytes);
  d.pt.in_small = b.take<u64>((size_t)LV_GRID * LV_MAX_B); d.pt.in_mid = b.take<u64>((size_t)LV_GRID_BIG * LV_MAX_B);
  d.pt.in_large = b.take<u64>((size_t)LV_GRID_BIG * LV_MAX_B);
  for (int i = 0; i < 2; ++i) {
    d.pt.mv_small[i] = b.take<unsigned>((size_t)LV_GRID * LV_MAX_B); d.pt.mv_mid[i] = b.take<unsigned>((size_t)LV_GRID_BIG * LV_MAX_B);
    d.pt.mv_large[i] = b.take<unsigned>((size_t)LV_GRID_BIG * LV_MAX_B);
  }
  d.pt.sq = b.take<double>((size_t)LV_MAX_B * LV_SQ_BLOCKS);
  d.ctl = b.take<LvCtl>(1);
  d.scalars = b.take<u64>(8);
  if (w) *w = d;
  return b.off + 256;
}

static inline unsigned lv_blocks(int64_t n, int per) { return (unsigned)gficf_ceil_div(n > 0 ? n : 1, per); }
static inline unsigned lv_grid(int64_t n, int per, unsigned cap) { const unsigned b = lv_blocks(n, per); return b < cap ? b : cap; }

static int lv_env_sub_rounds() {
  if (const char* e = getenv("GFICF_LOUVAIN_SUBROUNDS")) { const int s = atoi(e); if (s >= 1 && s <= 64) return s; }
  return 0;
}

// starts run together: as many as the workspace holds (GFICF_LOUVAIN_BATCH in the environment: at most that many — 1 = one after the other)
static int lv_batch_limit() {
  if (const char* e = getenv("GFICF_LOUVAIN_BATCH")) { const int b = atoi(e); if (b >= 1 && b <= LV_MAX_B) return b; }
  return LV_MAX_B;
}

}  // namespace

// the pinned block the control kernels report into, and the two events of the one-iteration-ahead loop (ctx.hip releases them)
int gficf_lv_host_get(gficf_ctx* ctx, void** host, hipEvent_t* ev) {
  if (!ctx->lv_host) {
    GFICF_HIP_CHECK(hipHostMalloc(&ctx->lv_host, sizeof(LvHost), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->lv_host, 0, sizeof(LvHost));
    for (int i = 0; i < 2; ++i) GFICF_HIP_CHECK(hipEventCreateWithFlags(&ctx->lv_ev[i], hipEventDisableTiming));
  }
  *host = ctx->lv_host;
  ev[0] = ctx->lv_ev[0]; ev[1] = ctx->lv_ev[1];
  return GFICF_OK;
}

extern "C" {

size_t gficf_louvain_workspace_bytes(int64_t N, int64_t nnz, int n_start) {
  if (N < 0 || nnz < 0) return 0;
  int B = n_start < 1 ? 1 : n_start > LV_MAX_B ? LV_MAX_B : n_start;
  const int lim = lv_batch_limit();
  if (B > lim) B = lim;
  while (B > 1 && (int64_t)B * N > (int64_t)INT32_MAX) --B;      // union vertex ids are int32
  return lv_carve(nullptr, nullptr, N, nnz, B);
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
  if (ws_bytes < gficf_louvain_workspace_bytes(N, nnz, 1))
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "workspace too small: %zu < %zu bytes", ws_bytes, gficf_louvain_workspace_bytes(N, nnz, 1));
  // as many starts together as the caller's workspace holds (gficf_louvain_workspace_bytes(N, nnz, n_start) sizes it for all of them)
  int Bmax = n_start < LV_MAX_B ? n_start : LV_MAX_B;
  if (Bmax > lv_batch_limit()) Bmax = lv_batch_limit();
  while (Bmax > 1 && ((int64_t)Bmax * N > (int64_t)INT32_MAX || lv_carve(nullptr, nullptr, N, nnz, Bmax) > ws_bytes)) --Bmax;
  static std::atomic<bool> attr_set[64];                 // per device: the attribute belongs to the device's copy of the kernel
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_lv_move_big<LV_BIG_SLOTS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_BIG_SLOTS * 12));
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_lv_move_big<LV_BIG_SLOTS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_BIG_SLOTS * 12));
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_lv_emit_big<LV_BIG_SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, LV_BIG_SLOTS * 12));
    attr_set[ctx->device & 63] = true;
  }
  LvWs w;
  lv_carve(&w, d_ws, N, nnz, Bmax);
  hipStream_t st = ctx->stream;
  void* host_v = nullptr;
  hipEvent_t ev[2];
  int rc = gficf_lv_host_get(ctx, &host_v, ev);
  if (rc) return rc;
  LvHost* const host = (LvHost*)host_v;

  // level 0: fixed-point weights, vertex weights, 2W, the vertices of the workgroup path
  GFICF_HIP_CHECK(hipMemsetAsync(w.scalars, 0, 8 * sizeof(u64), st));
  GFICF_HIP_CHECK(hipMemsetAsync(w.ctl, 0, sizeof(LvCtl), st));
  if (nnz > 0) hipLaunchKernelGGL(k_lv_fix, dim3(lv_blocks(nnz, 256) < 2048u ? lv_blocks(nnz, 256) : 2048u), dim3(256), 0, st, N, nnz, d_indices, d_x, w.wt0, w.scalars + 5, ctx->d_status);
  hipLaunchKernelGGL(k_lv_vertex_weight, dim3(lv_blocks(N, 4) < 2048u ? lv_blocks(N, 4) : 2048u), dim3(256), 0, st, N, nnz, d_indptr, d_indices, w.wt0, w.kv0, w.scalars, ctx->d_status);
  u64 h_sc[6] = {0, 0, 0, 0, 0, 0};
  GFICF_HIP_CHECK(hipMemcpyAsync(h_sc, w.scalars, sizeof(h_sc), hipMemcpyDeviceToHost, st));
  rc = gficf_ctx_sync(ctx);                        // also reports a malformed matrix before anything follows it
  if (rc) return rc;
  const u64 two_w_fix = h_sc[0];
  if ((double)h_sc[5] * (double)nnz >= 9.0e18)     // the u64 sums (2W, community totals) could wrap
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "edge weights too large for the 2^-32 fixed-point sums (largest weight x entries >= 2^31): scale the matrix");
  const double two_w = (double)two_w_fix;
  if (two_w_fix == 0) {                            // no edges: every vertex is its own cluster, Q = 0
    hipLaunchKernelGGL(k_lv_iota32, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, d_labels);
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
  const int s_env = lv_env_sub_rounds();
  const bool debug = getenv("GFICF_LOUVAIN_DEBUG") != nullptr;      // per-level trace on stderr

  // the finest graph's workgroup-path vertices: the same for every start, level-0 pass and batch
  LvG g00{N, N, 1, d_indptr, d_indptr + 1, d_indices, w.wt0, w.kv0, nullptr};
  hipLaunchKernelGGL(k_lv_list_big, dim3(lv_grid(N, 256, 1024)), dim3(256), 0, st, g00, (const int64_t*)nullptr, w.big0, &w.ctl->n_mid);
  hipLaunchKernelGGL(k_lv_ctl_publish, dim3(1), dim3(64), 0, st, (const LvCtl*)w.ctl, host);
  GFICF_HIP_CHECK(hipStreamSynchronize(st));
  const unsigned n_mid0 = host->n_mid, n_large0 = host->n_large;

  // ---- one level's steps; g = the level's union graph, big = its workgroup-path lists, (n_mid, n_large) their sizes
  struct Lists { const int32_t* big; unsigned n_mid, n_large; };
  // the labels as given (seed == NULL: singletons), totals, sizes
  const auto start_level = [&](const LvG& g, const int32_t* seed, bool binned) {
    hipLaunchKernelGGL(k_lv_init, dim3(lv_grid(g.n, 256, 2048)), dim3(256), 0, st, g, (const LvCtl*)w.ctl, seed, w.comm, w.K, w.size, w.mark_r, w.mark_w, w.iw);
    if (seed) hipLaunchKernelGGL(k_lv_accum, dim3(lv_grid(g.n, 256, 1024)), dim3(256), 0, st, g, (const LvCtl*)w.ctl, binned ? 1 : 0, (const int32_t*)w.comm, w.K, w.size);
  };
  // local moving until every component's level has ended: one iteration ahead of the device's decisions, never draining the stream
  const auto local_moving = [&](const LvG& g, const Lists& L, int S_max) -> int {
    const unsigned gs = lv_grid(g.nb, 4, LV_GRID), ga = lv_grid(g.n, 256, 2048);
    const unsigned gm = L.n_mid ? (unsigned)(((int64_t)L.n_mid * g.rep + 1) / 2 < LV_GRID_BIG ? ((int64_t)L.n_mid * g.rep + 1) / 2 : LV_GRID_BIG) : 0u;      // two waves a workgroup
    const unsigned gl = L.n_large ? (unsigned)((int64_t)L.n_large * g.rep < LV_GRID_BIG ? (int64_t)L.n_large * g.rep : LV_GRID_BIG) : 0u;
    for (int it = 0; it <= LV_MAX_ITERS + 1; ++it) {
      const int par = it & 1;
      hipLaunchKernelGGL(k_lv_pre, dim3(LV_SQ_BLOCKS, (unsigned)Bmax), dim3(256), 0, st, (const LvCtl*)w.ctl, par, (const u64*)w.K, (const int32_t*)w.size,
                         w.snapK[par], w.snapS[par], w.pt.sq);
      for (int s = 0; s < S_max; ++s) {
        const LvMove mv{r, s, it * S_max + s + 1, it * S_max + s + 1 - S_max};
        if (s == 0) {
          hipLaunchKernelGGL(k_lv_move_small<true>, dim3(gs), dim3(256), 0, st, g, (const LvCtl*)w.ctl, mv, (const int32_t*)w.comm, (const u64*)w.K,
                             (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_small, w.pt.mv_small[par]);
          if (gm) hipLaunchKernelGGL(k_lv_move_mid<true>, dim3(gm), dim3(128), 0, st, g, (const LvCtl*)w.ctl, mv, (int64_t)L.n_mid, L.big,
                                     (const int32_t*)w.comm, (const u64*)w.K, (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_mid, w.pt.mv_mid[par], ctx->d_status);
          if (gl) hipLaunchKernelGGL((k_lv_move_big<LV_BIG_SLOTS, true>), dim3(gl), dim3(256), LV_BIG_SLOTS * 12, st, g, (const LvCtl*)w.ctl, mv, 1, (int64_t)L.n_large, L.big,
                                     (const int32_t*)w.comm, (const u64*)w.K, (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_large, w.pt.mv_large[par], ctx->d_status);
          hipLaunchKernelGGL(k_lv_decide, dim3(1), dim3(1024), 0, st, w.ctl, w.pt, (int)gs, (int)gm, (int)gl, it, two_w, q_coef, host);
        } else {
          hipLaunchKernelGGL(k_lv_move_small<false>, dim3(gs), dim3(256), 0, st, g, (const LvCtl*)w.ctl, mv, (const int32_t*)w.comm, (const u64*)w.K,
                             (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_small, w.pt.mv_small[par]);
          if (gm) hipLaunchKernelGGL(k_lv_move_mid<false>, dim3(gm), dim3(128), 0, st, g, (const LvCtl*)w.ctl, mv, (int64_t)L.n_mid, L.big,
                                     (const int32_t*)w.comm, (const u64*)w.K, (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_mid, w.pt.mv_mid[par], ctx->d_status);
          if (gl) hipLaunchKernelGGL((k_lv_move_big<LV_BIG_SLOTS, false>), dim3(gl), dim3(256), LV_BIG_SLOTS * 12, st, g, (const LvCtl*)w.ctl, mv, 1, (int64_t)L.n_large, L.big,
                                     (const int32_t*)w.comm, (const u64*)w.K, (const int32_t*)w.size, w.next, (const int32_t*)w.mark_r, w.mark_w, w.iw, w.pt.in_large, w.pt.mv_large[par], ctx->d_status);
        }
        hipLaunchKernelGGL(k_lv_apply, dim3(ga), dim3(256), 0, st, g, (const LvCtl*)w.ctl, s, par, w.comm, (const int32_t*)w.next, w.K, w.size, w.snapc[0],
                           w.snapc[1], (const u64*)w.snapK[par ^ 1], (const int32_t*)w.snapS[par ^ 1], w.mark_r, (const int32_t*)w.mark_w);
      }
      GFICF_HIP_CHECK(hipEventRecord(ev[par], st));
      if (it >= 1) {                                 // the decision of the iteration BEFORE the one just enqueued
        GFICF_HIP_CHECK(hipEventSynchronize(ev[par ^ 1]));
        if (debug) {
          fprintf(stderr, "[louvain]   n=%lld iter %d: %d component(s) go on;", (long long)g.n, it - 1, host->n_cont[it - 1]);
          for (int b = 0; b < Bmax; ++b) fprintf(stderr, " %.9f (%u moved)", host->q_iter[b], host->mv_iter[b]);
          fprintf(stderr, "\n");
        }
        if (host->n_cont[it - 1] == 0) break;        // every level has ended: the iteration just enqueued does nothing
      }
    }
    return GFICF_OK;
  };
  // the communities in use numbered 0 .. (w.flag = old id -> new id, contiguous per component); lab = new local id of comm[src] (src NULL: the vertex)
  const auto renumber = [&](const LvG& g, const int32_t* src, bool seeded, bool refine, int level) -> int {
    hipLaunchKernelGGL(k_lv_used, dim3(lv_grid(g.n + 1, 256, 2048)), dim3(256), 0, st, g.n, (const int32_t*)w.size, w.flag);
    const int rc2 = gficf_exclusive_scan_i64(ctx, w.flag, g.n + 1);
    if (rc2) return rc2;
    hipLaunchKernelGGL(k_lv_ctl_renumbered, dim3(1), dim3(64), 0, st, w.ctl, (const int64_t*)w.flag, g.n, seeded ? 1 : 0, refine ? 1 : 0, algorithm, level, host);
    hipLaunchKernelGGL(k_lv_relabel, dim3(lv_grid(N, 256, 512), (unsigned)Bmax), dim3(256), 0, st, N, (const LvCtl*)w.ctl, src, (const int32_t*)w.comm,
                       (const int64_t*)w.flag, w.lab);
    return GFICF_OK;
  };
  // g reduced by the labels newid[comm[.]] (the components with `cont` set) into the arrays of nl, and the workgroup-path lists of the result
  const auto reduce = [&](const LvG& g, const Lists& L, const int64_t* newid, LvLevel& nl) -> int {
    GFICF_HIP_CHECK(hipMemsetAsync(nl.beg, 0, sizeof(int64_t) * (size_t)(g.n + 1), st));
    GFICF_HIP_CHECK(hipMemsetAsync(w.cur, 0, sizeof(int32_t) * (size_t)g.n, st));
    hipLaunchKernelGGL(k_lv_rowcap, dim3(lv_grid(g.n, 256, 1024)), dim3(256), 0, st, g, (const LvCtl*)w.ctl, (const int32_t*)w.comm, newid, (const u64*)w.K,
                       (const int32_t*)w.size, nl.beg, nl.kv, nl.vcomp);
    const int rc2 = gficf_exclusive_scan_i64(ctx, nl.beg, g.n + 1);
    if (rc2) return rc2;
    hipLaunchKernelGGL(k_lv_emit_small, dim3(lv_grid(g.nb, 4, LV_GRID)), dim3(256), 0, st, g, (const LvCtl*)w.ctl, (const int32_t*)w.comm, newid,
                       (const int64_t*)nl.beg, w.cur, nl.nbr, nl.wt);
    if (L.n_mid) {
      const unsigned gm = (unsigned)((int64_t)L.n_mid * g.rep < LV_GRID_BIG ? (int64_t)L.n_mid * g.rep : LV_GRID_BIG);
      hipLaunchKernelGGL(k_lv_emit_big<LV_EMIT_SLOTS>, dim3(gm), dim3(256), LV_EMIT_SLOTS * 12, st, g, (const LvCtl*)w.ctl, 0, (int64_t)L.n_mid, L.big, (const int32_t*)w.comm, newid,
                         (const int64_t*)nl.beg, w.cur, nl.nbr, nl.wt, ctx->d_status);
    }
    if (L.n_large) {
      const unsigned gl = (unsigned)((int64_t)L.n_large * g.rep < LV_GRID_BIG ? (int64_t)L.n_large * g.rep : LV_GRID_BIG);
      hipLaunchKernelGGL(k_lv_emit_big<LV_BIG_SLOTS>, dim3(gl), dim3(256), LV_BIG_SLOTS * 12, st, g, (const LvCtl*)w.ctl, 1, (int64_t)L.n_large, L.big, (const int32_t*)w.comm, newid,
                         (const int64_t*)nl.beg, w.cur, nl.nbr, nl.wt, ctx->d_status);
    }
    hipLaunchKernelGGL(k_lv_finish_rows, dim3(lv_grid(g.n, 256, 1024)), dim3(256), 0, st, (const int64_t*)&w.ctl->n_union2, (const int64_t*)nl.beg,
                       (const int32_t*)w.cur, nl.end);
    // the next level's lists (its vertex count is on the device: n_union2)
    GFICF_HIP_CHECK(hipMemsetAsync(&w.ctl->n_mid, 0, 2 * sizeof(unsigned), st));
    LvG g2{g.n, g.n, 1, nl.beg, nl.end, nl.nbr, nl.wt, nl.kv, nl.vcomp};
    hipLaunchKernelGGL(k_lv_list_big, dim3(lv_grid(g.n, 256, 1024)), dim3(256), 0, st, g2, (const int64_t*)&w.ctl->n_union2, nl.big, &w.ctl->n_mid);
    hipLaunchKernelGGL(k_lv_ctl_publish, dim3(1), dim3(64), 0, st, (const LvCtl*)w.ctl, host);
    return GFICF_OK;
  };
  const auto sync_level = [&]() -> int {            // the level's ONE synchronisation
    GFICF_HIP_CHECK(hipGetLastError());
    GFICF_HIP_CHECK(hipStreamSynchronize(st));
    return GFICF_OK;
  };
  const auto s_max_of = [&](const int64_t* n_of, const bool* live) {
    int S = 1;
    for (int b = 0; b < Bmax; ++b)
      if (live[b]) { const int s = lv_sub_rounds_of(n_of[b], s_env); S = s > S ? s : S; }
    return S;
  };

  // Random starts (reference src/RModularityOptimizer.cpp:108-142): every start begins from singletons, runs up to n_iter
  // passes and is kept if its modularity beats the best so far.  Nothing here is random; what a start varies is the seed
  // of the hash that splits the vertices into sub-round classes (start 0 with seed 0 is the plain deterministic run).
  double q_best = -INFINITY;
  int64_t n_best = 0;
  for (int first = 0; first < n_start; first += Bmax) {
    const int B = n_start - first < Bmax ? n_start - first : Bmax;
    hipLaunchKernelGGL(k_lv_ctl_batch, dim3(1), dim3(64), 0, st, w.ctl, B, first, seed);
    const LvG g0{(int64_t)B * N, N, B, d_indptr, d_indptr + 1, d_indices, w.wt0, w.kv0, nullptr};
    const Lists L0{w.big0, n_mid0, n_large0};
    bool finished[LV_MAX_B], any_move[LV_MAX_B], live[LV_MAX_B];
    int64_t n_of[LV_MAX_B], n_labels[LV_MAX_B];
    double q_final[LV_MAX_B];
    int n_saved[LV_MAX_B];
    int64_t saved_n[LV_MAX_B][LV_MAX_SAVED + 1];
    for (int b = 0; b < LV_MAX_B; ++b) { finished[b] = b >= B; any_move[b] = true; n_labels[b] = 0; q_final[b] = 0.0; n_saved[b] = 0; }
    for (int pass = 0; pass < n_iter; ++pass) {
      int n_live = 0;
      for (int b = 0; b < B; ++b) {
        if (pass > 0 && !any_move[b]) finished[b] = true;
        live[b] = !finished[b];
        n_of[b] = N;
        any_move[b] = false;
        n_saved[b] = 0;
        n_live += live[b] ? 1 : 0;
      }
      if (!n_live) break;
      LvG g = g0;
      Lists L = L0;
      int64_t tot_labels = 0;
      for (int b = 0; b < B; ++b) tot_labels += live[b] ? n_labels[b] : 0;
      for (int level = 0;; ++level) {
        const bool seeded = level == 0 && pass > 0;
        if (debug) fprintf(stderr, "[louvain] starts %d..%d pass %d level %d: union n=%lld\n", first, first + B - 1, pass, level, (long long)g.n);
        hipLaunchKernelGGL(k_lv_ctl_level, dim3(1), dim3(64), 0, st, w.ctl, level == 0 ? 0 : 1, pass, N, s_env);
        start_level(g, seeded ? w.lab : nullptr, tot_labels <= LV_ACC_BINS);
        rc = local_moving(g, L, s_max_of(n_of, live));
        if (rc) return rc;
        rc = renumber(g, level == 0 ? nullptr : w.lab, seeded, false, level);
        if (rc) return rc;
        if (algorithm == 2 && level < LV_MAX_SAVED)       // the next level's vertices of every start that goes on: lab as it is now
          GFICF_HIP_CHECK(hipMemcpyAsync(w.tops + (size_t)level * (size_t)Bmax * (size_t)N, w.lab, sizeof(int32_t) * (size_t)B * (size_t)N, hipMemcpyDeviceToDevice, st));
        LvLevel& nl = w.lvl[level & 1];
        rc = reduce(g, L, w.flag, nl);                    // (enqueued before the host knows whether anybody goes on: a small graph by then)
        if (rc) return rc;
        rc = sync_level();
        if (rc) return rc;
        int n_cont = 0;
        for (int b = 0; b < B; ++b) {
          if (!live[b]) continue;
          const LvHostComp& H = host->c[b];
          n_labels[b] = H.n2;
          q_final[b] = H.q_prev;
          any_move[b] = H.any_move != 0;
          if (H.cont) {
            if (algorithm == 2 && n_saved[b] == level && level < LV_MAX_SAVED) saved_n[b][++n_saved[b]] = H.n2;
            n_of[b] = H.n2;
            ++n_cont;
          } else {
            live[b] = false;
          }
          if (debug) fprintf(stderr, "[louvain]   start %d: %lld communities, Q %.9f, %s\n", first + b, (long long)H.n2, H.q_prev, H.cont ? "goes on" : "descent done");
        }
        if (!n_cont) break;
        const int64_t n2u = host->n_union2;
        g = LvG{n2u, n2u, 1, nl.beg, nl.end, nl.nbr, nl.wt, nl.kv, nl.vcomp};
        L = Lists{nl.big, host->n_mid, host->n_large};
      }

      // ---- algorithm 2: back up through the levels, local moving on each with the labels found below it
      // (runLouvainAlgorithmWithMultilevelRefinement, reference :629-649).  The graph of a level is rebuilt from the finest
      // one by its saved vertex map instead of being kept.
      if (algorithm == 2) {
        int top_level = -1;
        for (int b = 0; b < B; ++b)
          if (!finished[b] && any_move[b]) top_level = n_saved[b] > top_level ? n_saved[b] : top_level;
        for (int level = top_level; level >= 0; --level) {
          bool part[LV_MAX_B];
          int64_t n_lv[LV_MAX_B], n_un = 0, tot_lab = 0;
          for (int b = 0; b < LV_MAX_B; ++b) {
            part[b] = b < B && !finished[b] && any_move[b] && n_saved[b] >= level;
            n_lv[b] = part[b] ? (level ? saved_n[b][level] : N) : 0;
            n_un += n_lv[b];
            tot_lab += part[b] ? n_labels[b] : 0;
          }
          const int32_t* top = level ? w.tops + (size_t)(level - 1) * (size_t)Bmax * (size_t)N : nullptr;
          hipLaunchKernelGGL(k_lv_ctl_refine, dim3(1), dim3(64), 0, st, w.ctl, level, N);
          LvG gl = g0;
          Lists Ll = L0;
          const int32_t* seedp = w.lab;
          if (level) {
            // the level's graph: the finest one reduced by the saved vertex map (labels seed_base + top, new ids = the labels themselves)
            LvLevel& nl = w.lvl[0];
            hipLaunchKernelGGL(k_lv_init, dim3(lv_grid(g0.n, 256, 2048)), dim3(256), 0, st, g0, (const LvCtl*)w.ctl, top, w.comm, w.K, w.size, w.mark_r, w.mark_w, w.iw);
            hipLaunchKernelGGL(k_lv_accum, dim3(lv_grid(g0.n, 256, 1024)), dim3(256), 0, st, g0, (const LvCtl*)w.ctl, 0, (const int32_t*)w.comm, w.K, w.size);
            hipLaunchKernelGGL(k_lv_internal, dim3(lv_grid(N, 4, LV_GRID)), dim3(256), 0, st, g0, (const LvCtl*)w.ctl, (const int32_t*)w.comm, w.pt.in_small);
            hipLaunchKernelGGL(k_lv_seed, dim3(lv_grid(N, 256, 512), (unsigned)Bmax), dim3(256), 0, st, N, (const LvCtl*)w.ctl, top, (const int32_t*)w.lab, w.seedl);
            hipLaunchKernelGGL(k_lv_iota, dim3(lv_grid(g0.n + 1, 256, 2048)), dim3(256), 0, st, g0.n, w.flag);
            // the vertex ranges of the level are fixed by k_lv_ctl_refine: v0 of the level, not of the finest graph — seed_base carried them
            // into the labels above; from here on v0 / n are the level's
            rc = reduce(g0, L0, w.flag, nl);
            if (rc) return rc;
            hipLaunchKernelGGL(k_lv_ctl_self, dim3(1), dim3(256), 0, st, w.ctl, (const u64*)w.pt.in_small, (int)lv_grid(N, 4, LV_GRID));
            rc = sync_level();                            // the sizes of the lists
            if (rc) return rc;
            gl = LvG{n_un, n_un, 1, nl.beg, nl.end, nl.nbr, nl.wt, nl.kv, nl.vcomp};
            Ll = Lists{nl.big, host->n_mid, host->n_large};
            seedp = w.seedl;
          }
          if (debug) fprintf(stderr, "[louvain] starts %d..%d pass %d refinement level %d: union n=%lld\n", first, first + B - 1, pass, level, (long long)gl.n);
          hipLaunchKernelGGL(k_lv_ctl_level, dim3(1), dim3(64), 0, st, w.ctl, 2, pass, N, s_env);
          start_level(gl, seedp, tot_lab <= LV_ACC_BINS);
          rc = local_moving(gl, Ll, s_max_of(n_lv, part));
          if (rc) return rc;
          rc = renumber(gl, top, false, true, level);
          if (rc) return rc;
          rc = sync_level();
          if (rc) return rc;
          for (int b = 0; b < B; ++b)
            if (part[b]) { n_labels[b] = host->c[b].n2; q_final[b] = host->c[b].q_prev; }
        }
      }
    }
    for (int b = 0; b < B; ++b) {
      if (debug) fprintf(stderr, "[louvain] start %d: Q %.9f, %lld clusters\n", first + b, q_final[b], (long long)n_labels[b]);
      if (q_final[b] > q_best) {                     // strictly better, as the reference keeps the first of equals (:128)
        q_best = q_final[b];
        n_best = n_labels[b];
        GFICF_HIP_CHECK(hipMemcpyAsync(w.best, w.lab + (size_t)b * (size_t)N, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, st));
      }
    }
  }
  *n_clusters = n_best;

  // ---- clusters by decreasing size
  const int64_t C = n_best;
  GFICF_HIP_CHECK(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * (size_t)C, st));
  hipLaunchKernelGGL(k_lv_count, dim3(lv_grid(N, 256, 1024)), dim3(256), 0, st, N, C, (const int32_t*)w.best, w.cnt);
  hipLaunchKernelGGL(k_lv_size_keys, dim3(lv_blocks(C, 256)), dim3(256), 0, st, C, N, (const int32_t*)w.cnt, w.keys_a, w.vals_a);
  size_t tb = w.sort_tmp_bytes;
  GFICF_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (size_t)C, 0u, 64u, st));
  hipLaunchKernelGGL(k_lv_rank, dim3(lv_blocks(C, 256)), dim3(256), 0, st, C, (const u64*)w.vals_b, w.rank);
  hipLaunchKernelGGL(k_lv_final, dim3(lv_blocks(N, 256)), dim3(256), 0, st, N, (const int32_t*)w.best, (const int32_t*)w.rank, d_labels);
  GFICF_HIP_CHECK(hipGetLastError());
  if (modularity) *modularity = q_best;
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
  const size_t nsz = (size_t)(nnz > 0 ? nnz : 1), wsb = gficf_louvain_workspace_bytes(N, nnz, n_start);
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
