// knn.hip — exact k-nearest-neighbour search for gfx950 (MI355X): "next" row N2 of the hot path.
//
// Stands in for the caller's step in front of the Jaccard build,
//   neigh = uwot:::find_nn(data$pca$cells, k = k+1, include_self = T, method = "annoy", metric = dist.method)$idx
// (reference R/clustCells.R:57,60; uwot/Annoy are third-party and approximate).  This is an EXACT
// search: every query is compared with every point in f32 (Annoy stores f32 too) and the k
// smallest (distance, index) pairs are kept, ties broken by the smaller index — so the result is
// unique and checkable bit for bit against a CPU brute force.
//
// Metrics: manhattan (the reference's default, R/clustCells.R:46), euclidean, cosine (1 - cos), correlation (cosine of
// the mean-centred rows).
// Manhattan is |a-b| accumulation — VALU work, not a contraction, so no MFMA; euclidean and cosine
// share the same register-tiled kernel with a packed-fma chain in dimension order (an MFMA formulation
// |x|^2+|y|^2-2xy would change the rounding and with it the order of near-ties, and in f32 the matrix
// cores peak at the same 157 TFLOP/s as v_pk_fma_f32).
//
// Layout: points row-major f32 [N][dpad] (dpad = d rounded up to 4, zero padded; cosine: rows
// L2-normalised by the prepare kernel).  One workgroup = a tile of 64 queries x a slice of the
// candidates: the query tile stays in LDS ([dim][query]), candidate tiles of 128 points stream
// through a double-buffered LDS chunk of 16 dims; every thread accumulates a 4 x 8 block of
// distances in registers.  Per query a sorted list of the k best 64-bit keys
// (sortable(distance) << 32 | index) lives in LDS; a thread inserts a candidate only when it beats
// the list's last key (rare after the first tiles).  A query's row of the tile belongs to the 16 lanes
// of one wave, which take turns (wave-uniform loop, one elected lane per row and round): no locks.  A merge kernel
// writes the 1-based index matrix column-major — the layout the Jaccard ingest reads.
//
// Two forms of the same kernel, same bits:
//   plain  : every query tile against every candidate tile, the candidate range split S ways to fill the chip;
//   pruned : (default from 30 k points) the points are reordered so that a tile holds neighbours, every
//            (query tile, candidate tile) pair gets a triangle-inequality lower bound, and a query tile visits the
//            candidate tiles best bound first until the bound passes its queries' k-th best — see "pruned search:
//            preparation" below.  When the bounds show no cluster structure, a device flag routes to the plain form.
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include <atomic>

#include "common.h"

namespace {

constexpr int KNN_TC = 128;        // candidates per tile
constexpr int KNN_DK = 16;         // dims per LDS chunk of the candidate tile
constexpr int KNN_THREADS = 256;
constexpr int KNN_MAX_D = 128;
constexpr int KNN_MAX_SPLIT = 16;

__host__ __device__ inline int knn_dpad(int d) { return (d + 3) & ~3; }

// order-preserving map float -> uint32 (and back)
__device__ inline uint32_t f32_sortable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ inline float sortable_f32(uint32_t s) {
  return __uint_as_float(s ^ (((s >> 31) - 1u) | 0x80000000u));
}

// ----------------------------------------------------------------------------- prepare
// R matrix (column-major, f64 or f32) -> row-major f32 rows of dpad floats.  One thread per row;
// cosine: the row is divided by its f32 L2 norm (fma chain in dimension order), zero rows stay zero.
template <typename T>
__global__ __launch_bounds__(256) void k_knn_prepare(const T* __restrict__ X, int64_t n_rows, int d, int dpad, int64_t ld,
                                                     int metric, float* __restrict__ out, uint32_t* __restrict__ status) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  bool bad = false;
  float inv = 1.0f;
  bool scale = false;
  float mean = 0.0f;
  if (metric == GFICF_KNN_CORRELATION) {         // Pearson: centre the row first (f32 sum in dimension order / d)
    float sm = 0.0f;
    for (int t = 0; t < d; ++t) sm = sm + (float)X[(int64_t)t * ld + r];
    mean = sm / (float)d;
  }
  if (metric == GFICF_KNN_COSINE || metric == GFICF_KNN_CORRELATION) {
    float s = 0.0f;
    for (int t = 0; t < d; ++t) { const float v = (float)X[(int64_t)t * ld + r] - mean; s = fmaf(v, v, s); }
    const float nrm = sqrtf(s);
    scale = nrm > 0.0f;
    inv = nrm;
  }
  float* o = out + r * dpad;
  for (int t = 0; t < dpad; ++t) {
    float v = t < d ? (float)X[(int64_t)t * ld + r] : 0.0f;
    bad |= !(fabsf(v) <= FLT_MAX);              // NaN or +-Inf (also a double too large for f32)
    if (t < d) v = v - mean;
    if (scale) v = v / inv;
    o[t] = v;
  }
  if (bad) atomicOr(status, GFICF_ST_BAD_VALUE);
}

// ------------------------------------------------------------------------------ search
typedef unsigned long long u64;

// Insertions of one tile row.  The 16 lanes of a 16-lane group share the row (= the list).  The wave loops
// (uniformly) while any lane holds a candidate; per round the lowest such lane of each group hands one key
// to its group, and the group's lanes rebuild the list together: lane t owns entries t, t+16, ... and writes
//   cur <= key ? cur : (prev <= key ? key : prev)
// — an insertion shift with one LDS read and one LDS write per entry, no serial walk, no lock (one wave,
// program order).  Out of line: it runs rarely once the lists have warmed up and must not cost the
// distance loop its registers.
typedef __attribute__((address_space(3))) volatile u64 knn_lds_u64;

template <int KL>
__device__ __noinline__ float knn_row_insert(uint32_t list_addr, int kk, float d0, float d1, float d2, float d3, float d4, float d5,
                                             float d6, float d7, float tau, bool live, uint32_t j0, int tid,
                                             const int32_t* __restrict__ perm) {
  knn_lds_u64* const list = (knn_lds_u64*)(size_t)list_addr;      // LDS byte address of the row's list
  const float dv[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
  const int tx = tid & 15;
  uint32_t pass = 0;
  if (live) {
#pragma unroll
    for (int s = 0; s < 8; ++s) pass |= (dv[s] <= tau) ? 1u << s : 0u;
  }
  for (;;) {
    const u64 m = __ballot(pass != 0);
    if (m == 0) break;
    const uint32_t gm = (uint32_t)(m >> (tid & 48)) & 0xFFFFu;
    const int leader = __ffs(gm) - 1;                   // -1: this group has no candidate this round
    uint32_t khi = 0, klo = 0;
    if (pass != 0 && tx == leader) {
      const int s = __ffs(pass) - 1;
      pass &= pass - 1;
      float h = dv[0];
#pragma unroll
      for (int t = 1; t < 8; ++t) h = s == t ? dv[t] : h;
      khi = f32_sortable(h);
      klo = j0 + (uint32_t)((s < 4 ? 0 : 64) + tx * 4 + (s & 3));
      if (perm) klo = (uint32_t)perm[klo];              // pruned search: the key carries the point's original id
    }
    const int src = ((tid & 48) | (leader & 15)) << 2;
    khi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)khi);
    klo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)klo);
    if (leader >= 0) {
      const u64 key = ((u64)khi << 32) | (u64)klo;
      u64 nw[KL / 16];
#pragma unroll
      for (int e = 0; e < KL / 16; ++e) {
        const int pos = e * 16 + tx;
        const u64 cur = list[pos];
        const u64 prev = pos > 0 ? list[pos - 1] : 0ull;
        nw[e] = cur <= key ? cur : (prev <= key ? key : prev);
      }
#pragma unroll
      for (int e = 0; e < KL / 16; ++e) {
        const int pos = e * 16 + tx;
        if (pos < kk) list[pos] = nw[e];
      }
    }
  }
  // this wave is the only writer of its rows' lists, so the caller's register copy of the k-th best stays exact
  const uint32_t tau_hi = (uint32_t)(list[kk - 1] >> 32);
  return tau_hi == 0xFFFFFFFFu ? INFINITY : sortable_f32(tau_hi);     // list not full yet: everything enters
}

// One dimension of the 8 x 8 register block.  Accumulators are float pairs (two neighbouring candidates), so
// that a - b is one packed subtract per pair (v_pk_add_f32 with the query value broadcast by op_sel), the
// euclidean / cosine updates are packed fmas, and manhattan adds |d| with the source modifier (two plain adds
// per pair: there is no packed abs).  Built with -fno-slp-vectorize: the SLP vectoriser would otherwise pack
// the two adds and pay for it with two v_and to clear the sign bits.
typedef float knn_f2 __attribute__((ext_vector_type(2)));

// RQ = query rows per thread: 4 (tile of 64 queries; measured faster than 8 = 128 queries, which the code still supports:
// 100 k x 50, 31 nearest, manhattan: 27.7 against 33.1 ms — fewer insertions per slice, lighter epilogue).
struct KnnOperands {            // one dimension's slice of the tiles: RQ query values, 8 candidate values
  float4 a0, a1, b0, b1;
};
template <int RQ>
__device__ inline void knn_read(KnnOperands& o, const float* __restrict__ pa, const float* __restrict__ pb) {
  o.a0 = *reinterpret_cast<const float4*>(pa);
  if (RQ == 8) o.a1 = *reinterpret_cast<const float4*>(pa + 64);
  o.b0 = *reinterpret_cast<const float4*>(pb);
  o.b1 = *reinterpret_cast<const float4*>(pb + 64);
}

template <int METRIC, int RQ>
__device__ inline void knn_dim(knn_f2 (&acc)[RQ][4], const KnnOperands& o) {
  const float a[8] = {o.a0.x, o.a0.y, o.a0.z, o.a0.w, o.a1.x, o.a1.y, o.a1.z, o.a1.w};
  const knn_f2 b[4] = {{o.b0.x, o.b0.y}, {o.b0.z, o.b0.w}, {o.b1.x, o.b1.y}, {o.b1.z, o.b1.w}};
#pragma unroll
  for (int r = 0; r < RQ; ++r) {
    const knn_f2 ar = {a[r], a[r]};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (METRIC == GFICF_KNN_MANHATTAN) {
        const knn_f2 df = ar - b[s];
        acc[r][s].x = acc[r][s].x + __builtin_fabsf(df.x);
        acc[r][s].y = acc[r][s].y + __builtin_fabsf(df.y);
      } else if (METRIC == GFICF_KNN_EUCLIDEAN) {
        const knn_f2 df = ar - b[s];
        acc[r][s] = __builtin_elementwise_fma(df, df, acc[r][s]);
      } else {
        acc[r][s] = __builtin_elementwise_fma(ar, b[s], acc[r][s]);
      }
    }
  }
}

// Lists of up to 64 entries live in registers: lane t of the row's 16-lane group holds entries EPL*t .. EPL*t+EPL-1
// (EPL = KL / 16).  An insertion is the same shift as in the LDS form, but the entry in front of a lane's first one
// comes from the lane below through one DPP row shift (the group's lane 0 reads 0, i.e. "nothing in front"), so a
// round is ~25 VALU instructions and no LDS round trip.  Every lane of the group calls it with the group's key.
template <int EPL>
__device__ __forceinline__ void knn_reg_insert(u64 (&lst)[EPL], u64 key) {
  const u64 last = lst[EPL - 1];
  const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)last, 0x111, 0xf, 0xf, true);          // row_shr:1, zero fill
  const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(last >> 32), 0x111, 0xf, 0xf, true);
  u64 prev = ((u64)phi << 32) | (u64)plo;
#pragma unroll
  for (int h = 0; h < EPL; ++h) {
    const u64 cur = lst[h];
    lst[h] = cur <= key ? cur : (prev <= key ? key : prev);
    prev = cur;
  }
}

// Arguments of the tile kernel.  Queries and candidates are separate row-major arrays (the plain search passes
// Q = X + q_begin rows; the pruned search passes its two reordered copies).
struct KnnTileArgs {
  const float* Q;          // n_q query rows
  int64_t n_q;
  const float* X;          // N candidate rows
  int64_t N;
  int d, dpad, kk, S;
  u64* part;               // [n_q][S][kk] partial lists
  const int32_t* perm_x;   // PRUNE: original 0-based id of candidate row j (the keys carry original ids)
  const float* lb;         // PRUNE: [n_qt][n_ct] lower bound of the distance between query tile and candidate tile
  const int32_t* tile_n;   // PRUNE: points in candidate tile t (cells are padded to whole tiles: a prefix of the tile is real)
  const int32_t* qtile_n;  // PRUNE: queries in query tile qt
  const uint32_t* gate;    // optional: run only if *gate == gate_want (the pruned / plain choice is made on the device)
  uint32_t gate_want;
};

// One workgroup = one tile of TQ = 16 * RQ queries x a sequence of candidate tiles.
//   PRUNE == false: the candidate tiles of slice sp of S, in order.
//   PRUNE == true : S = 1; the candidate tiles in the order of their lower bound lb[qt][*], best first, and the
//     sequence ends at the first tile whose bound exceeds the largest k-th best distance of the tile's queries —
//     no point of that tile or of any later one can enter a list (the bound is a triangle-inequality bound
//     with slack for f32 rounding, built by k_knn_lb).  The next tile is chosen by every wave for itself from
//     the same LDS data (bounds, visited marks and the waves' k-th best maxima published one tile earlier,
//     double-buffered), so the waves agree without an extra barrier.
template <int METRIC, int KL, int RQ, bool PRUNE>
__global__ __launch_bounds__(KNN_THREADS, 2) void k_knn_tiles(const KnnTileArgs A) {
  constexpr int TQ = 16 * RQ;                                                // queries per workgroup
  if (A.gate != nullptr && *A.gate != A.gate_want) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int d = A.d, dpad = A.dpad, kk = A.kk, S = A.S;
  const int64_t N = A.N;
  float* const sA = reinterpret_cast<float*>(smem);                          // [dpad][TQ]
  float* const sB = sA + (size_t)dpad * TQ;                                  // [2][DK][TC]
  constexpr bool REGL = KL <= 64;                                            // lists in registers (else in LDS)
  constexpr int EPL = KL / 16;                                               // list entries per lane of a row's 16-lane group
  u64* const sKey = reinterpret_cast<u64*>(sB + 2 * KNN_DK * KNN_TC);        // !REGL: [TQ][KL]
  float* const sLb = reinterpret_cast<float*>(sKey + (REGL ? 0 : TQ * KL));  // PRUNE: [n_ct] bounds, +inf once visited
  const uint32_t key_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)sKey;   // LDS byte address
  __shared__ float s_wtau[2][KNN_THREADS / 64];                              // PRUNE: per-wave max of the k-th best, by tile parity

  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x / S, sp = blockIdx.x % S;
  const int64_t q0 = (int64_t)qt * TQ;
  const int nq_live = (PRUNE && A.qtile_n) ? A.qtile_n[qt] : (A.n_q - q0 < TQ ? (int)(A.n_q - q0) : TQ);   // rows that are real queries
  if (nq_live <= 0) return;                          // padding tile (uniform over the workgroup, before any barrier)
  const int64_t n_ct = gficf_ceil_div(N, KNN_TC);
  const int64_t ct0 = PRUNE ? 0 : n_ct * sp / S, ct1 = PRUNE ? n_ct : n_ct * (sp + 1) / S;
  const int nq4 = dpad >> 2;                         // float4 per point row
  const int nch = (d + KNN_DK - 1) / KNN_DK;         // chunks per candidate tile (padded dims are skipped)
  const float4* const X4 = reinterpret_cast<const float4*>(A.X);
  const float4* const Q4 = reinterpret_cast<const float4*>(A.Q);

  if (!REGL)
    for (int e = tid; e < TQ * KL; e += KNN_THREADS) sKey[e] = ~0ull;
  u64 lst[RQ][REGL ? EPL : 1];                       // REGL: this lane's entries of its rows' lists
#pragma unroll
  for (int r = 0; r < RQ; ++r)
#pragma unroll
    for (int h = 0; h < (REGL ? EPL : 1); ++h) lst[r][h] = ~0ull;
  // query tile -> sA[dim][query]; consecutive lanes take consecutive queries (conflict-free LDS writes)
  for (int f = tid; f < TQ * nq4; f += KNN_THREADS) {
    const int row = f & (TQ - 1), quad = f / TQ;
    const int64_t q = q0 + row;
    const float4 v = q < A.n_q ? Q4[q * nq4 + quad] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* o = sA + (size_t)(quad * 4) * TQ + row;
    o[0] = v.x; o[TQ] = v.y; o[2 * TQ] = v.z; o[3 * TQ] = v.w;
  }
  if (PRUNE) {
    for (int64_t e = tid; e < n_ct; e += KNN_THREADS) sLb[e] = A.lb[(int64_t)qt * n_ct + e];
    if (tid < 2 * (KNN_THREADS / 64)) s_wtau[tid / (KNN_THREADS / 64)][tid % (KNN_THREADS / 64)] = INFINITY;
  }

  // Staging of a candidate tile's dim chunk: 128 points x 16 dims = 512 float4, two per thread (point row_l,
  // float4 columns quad0 and quad0 + 2 of the chunk); consecutive lanes take consecutive points, so the
  // transposing LDS writes are conflict-free.
  const int row_l = tid & (KNN_TC - 1), quad0 = tid >> 7;
  float* const st0 = sB + (size_t)(quad0 * 4) * KNN_TC + row_l;             // LDS destination inside a buffer
  auto load_chunk = [&](int64_t ct, int c, float4 (&v)[2]) {
    const int64_t j = ct * KNN_TC + row_l;
    const int q4 = c * (KNN_DK / 4) + quad0;
    const float4* src = X4 + j * nq4 + q4;
    v[0] = (j < N && q4 < nq4) ? src[0] : make_float4(0.f, 0.f, 0.f, 0.f);
    v[1] = (j < N && q4 + 2 < nq4) ? src[2] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_chunk = [&](int buf, const float4 (&v)[2]) {
    float* o = st0 + (size_t)buf * KNN_DK * KNN_TC;
    o[0] = v[0].x; o[KNN_TC] = v[0].y; o[2 * KNN_TC] = v[0].z; o[3 * KNN_TC] = v[0].w;
    o += 8 * KNN_TC;
    o[0] = v[1].x; o[KNN_TC] = v[1].y; o[2 * KNN_TC] = v[1].z; o[3 * KNN_TC] = v[1].w;
  };
  // PRUNE: the unvisited tile with the smallest bound, or -1 when that bound is beyond every query's k-th best.
  // `skip` is the tile in work (its visited mark may not be visible to every wave yet); `par` selects the
  // published k-th best maxima of the tile before it.
  auto select_tile = [&](int64_t skip, int par) -> int64_t {
    float best = INFINITY;
    int bi = -1;
    for (int e = lane; e < (int)n_ct; e += 64) {
      const float v = sLb[e];
      if (e != (int)skip && v < best) { best = v; bi = e; }      // ascending e: the lowest index wins a tie
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
      const float ov = __shfl_xor(best, w);
      const int oi = __shfl_xor(bi, w);
      if (ov < best || (ov == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ov; bi = oi; }
    }
    float tmax = s_wtau[par][0];
#pragma unroll
    for (int w = 1; w < KNN_THREADS / 64; ++w) tmax = fmaxf(tmax, s_wtau[par][w]);
    return (bi >= 0 && best <= tmax) ? (int64_t)bi : -1;
  };

  knn_f2 acc[RQ][4];
#pragma unroll
  for (int r = 0; r < RQ; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[r][s] = knn_f2{0.0f, 0.0f};

  float tau[RQ];                // current k-th best distance of this thread's query rows
#pragma unroll
  for (int r = 0; r < RQ; ++r) tau[r] = INFINITY;

  __syncthreads();              // sLb / s_wtau / sA are in place
  int64_t cur = PRUNE ? select_tile(-1, 0) : (ct0 < ct1 ? ct0 : -1);
  float4 pre[2];
  if (cur >= 0) { load_chunk(cur, 0, pre); store_chunk(0, pre); }
  __syncthreads();

  int buf = 0, tile_no = 0;
  while (cur >= 0) {
    if (PRUNE && tid == 0) sLb[cur] = INFINITY;              // visited; read by select_tile only after a later barrier
    int64_t nxt = -1;
    for (int c = 0; c < nch; ++c) {
      bool have_next;
      if (c + 1 < nch) {
        load_chunk(cur, c + 1, pre);
        have_next = true;
      } else {
        nxt = PRUNE ? select_tile(cur, (tile_no + 1) & 1) : (cur + 1 < ct1 ? cur + 1 : -1);
        have_next = nxt >= 0;
        if (have_next) load_chunk(nxt, 0, pre);
      }
      const float* const pa = sA + (size_t)(c * KNN_DK) * TQ + ty * 4;
      const float* const pb = sB + (size_t)buf * KNN_DK * KNN_TC + tx * 4;
      const int nd = d - c * KNN_DK < KNN_DK ? d - c * KNN_DK : KNN_DK;
      {
        // Two dims per round (the point rows are zero padded to a multiple of 4 dims, and a zero dim adds exactly 0
        // to every metric's accumulator, so an odd d costs one padded dim).  The operands of the next dim are read
        // from LDS while the current one is accumulated.  One rolled loop for full and short chunks alike: the
        // accumulators never change registers.
        const int np = (nd + 1) >> 1;
        KnnOperands oa, ob;
        knn_read<RQ>(oa, pa, pb);
#pragma unroll 1       // measured: unroll 2 duplicates the body with a remainder copy and runs 45 % slower
        for (int t = 0; t < np; ++t) {
          knn_read<RQ>(ob, pa + (2 * t + 1) * TQ, pb + (2 * t + 1) * KNN_TC);
          knn_dim<METRIC, RQ>(acc, oa);
          if (t + 1 < np) knn_read<RQ>(oa, pa + (2 * t + 2) * TQ, pb + (2 * t + 2) * KNN_TC);
          knn_dim<METRIC, RQ>(acc, ob);
        }
      }
      if (c == nch - 1) {
        // the tile's distances of this thread against the current k-th best of their queries
        const int64_t j0 = cur * KNN_TC;
        // Only the data set's last candidate tile can hold fewer than TC points; it gets its own copy of the
        // code below (RAG), so that all the other tiles do not pay for the masking.
        const int nvalid = (PRUNE && A.tile_n) ? A.tile_n[cur] : (N - j0 < KNN_TC ? (int)(N - j0) : KNN_TC);
        auto epilogue = [&](auto rag_tag) {
          constexpr bool RAG = decltype(rag_tag)::value;      // RAG: candidates at columns >= nvalid do not exist
#pragma unroll
          for (int r = 0; r < RQ; ++r) {
            const int row = (r < 4 ? 0 : 64) + ty * 4 + (r & 3);
            const bool live = row < nq_live;
            // common case after the first tiles: nothing in the whole wave beats its query's k-th best
            bool any = false;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              const float av = (s & 1) ? acc[r][s >> 1].y : acc[r][s >> 1].x;
              bool ok = (METRIC == GFICF_KNN_COSINE ? 1.0f - av : av) <= tau[r];
              if (RAG) ok = ok && (s < 4 ? 0 : 64) + tx * 4 + (s & 3) < nvalid;
              any |= ok;
            }
            if (__ballot(any && live) != 0) {
              float dv[8];
#pragma unroll
              for (int s = 0; s < 8; ++s) {
                const float av = (s & 1) ? acc[r][s >> 1].y : acc[r][s >> 1].x;
                dv[s] = METRIC == GFICF_KNN_COSINE ? 1.0f - av : av;
                if (RAG && (s < 4 ? 0 : 64) + tx * 4 + (s & 3) >= nvalid) dv[s] = NAN;       // never <= tau
              }
              if (REGL) {
                // Insertions (rare once the lists have warmed up).  The 16 lanes of a 16-lane group share this row
                // (= this list).  The wave loops (uniformly) while any lane holds a candidate; per round the lowest
                // such lane of each group hands one key to its group, whose lanes shift it into their registers.
                uint32_t pass = 0;
                if (live) {
#pragma unroll
                  for (int s = 0; s < 8; ++s) pass |= (dv[s] <= tau[r]) ? 1u << s : 0u;
                }
                for (;;) {
                  const u64 m = __ballot(pass != 0);
                  if (m == 0) break;
                  const uint32_t gm = (uint32_t)(m >> (tid & 48)) & 0xFFFFu;
                  const int leader = __ffs(gm) - 1;             // -1: this group has no candidate this round
                  uint32_t khi = 0, klo = 0;
                  if (pass != 0 && tx == leader) {
                    const int s = __ffs(pass) - 1;
                    pass &= pass - 1;
                    float h = dv[0];
#pragma unroll
                    for (int t = 1; t < 8; ++t) h = s == t ? dv[t] : h;
                    khi = f32_sortable(h);
                    klo = (uint32_t)j0 + (uint32_t)((s < 4 ? 0 : 64) + tx * 4 + (s & 3));
                    if (PRUNE) klo = (uint32_t)A.perm_x[klo];   // the key carries the point's original id
                  }
                  const int src = ((tid & 48) | (leader & 15)) << 2;
                  khi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)khi);
                  klo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)klo);
                  if (leader >= 0) {
                    knn_reg_insert<REGL ? EPL : 1>(lst[r], ((u64)khi << 32) | (u64)klo);
                    {
                    // the list's k-th best may just have dropped: candidates it no longer admits need no round of their own
                    // (near tiles — the pruned form's, the pivot assignment's — pass most of their candidates by the stale bound)
                    uint32_t tn = (uint32_t)(lst[r][0] >> 32);
#pragma unroll
                    for (int h = 1; h < (REGL ? EPL : 1); ++h) tn = ((kk - 1) % EPL) == h ? (uint32_t)(lst[r][h] >> 32) : tn;
                    tn = (uint32_t)__shfl((int)tn, (kk - 1) / EPL, 16);
                    if (tn != 0xFFFFFFFFu && pass != 0) {
                      const float tnew = sortable_f32(tn);
                      uint32_t keep = 0;
#pragma unroll
                      for (int s = 0; s < 8; ++s) keep |= (dv[s] <= tnew) ? 1u << s : 0u;
                      pass &= keep;
                    }
                    }
                  }
                }
                // the row's k-th best: entry kk - 1 sits in lane (kk - 1) / EPL of the group
                uint32_t th = (uint32_t)(lst[r][0] >> 32);
#pragma unroll
                for (int h = 1; h < (REGL ? EPL : 1); ++h) th = ((kk - 1) % EPL) == h ? (uint32_t)(lst[r][h] >> 32) : th;
                th = (uint32_t)__shfl((int)th, (kk - 1) / EPL, 16);
                tau[r] = th == 0xFFFFFFFFu ? INFINITY : sortable_f32(th);      // list not full yet: everything enters
              } else {
                tau[r] = knn_row_insert<KL>(key_addr + (uint32_t)(row * KL * 8), kk, dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6],
                                            dv[7], tau[r], live, (uint32_t)j0, tid, PRUNE ? A.perm_x : (const int32_t*)nullptr);
              }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[r][s] = knn_f2{0.0f, 0.0f};
          }
        };
        if (nvalid < KNN_TC) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (PRUNE) {
          // this wave's largest k-th best (rows without a query do not count; an unfilled list is +inf)
          float tm = 0.0f;
#pragma unroll
          for (int r = 0; r < RQ; ++r)
            if ((r < 4 ? 0 : 64) + ty * 4 + (r & 3) < nq_live) tm = fmaxf(tm, tau[r]);
          tm = fmaxf(tm, __shfl_xor(tm, 16));
          tm = fmaxf(tm, __shfl_xor(tm, 32));
          if (lane == 0) s_wtau[tile_no & 1][wave] = tm;
        }
      }
      if (have_next) store_chunk(buf ^ 1, pre);
      __syncthreads();
      buf ^= 1;
    }
    cur = nxt;
    ++tile_no;
  }

  // partial lists of this candidate slice
  if (REGL) {
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int64_t q = q0 + (r < 4 ? 0 : 64) + ty * 4 + (r & 3);
#pragma unroll
      for (int h = 0; h < (REGL ? EPL : 1); ++h) {
        const int e = EPL * tx + h;
        if (q < A.n_q && e < kk) A.part[(q * S + sp) * kk + e] = lst[r][h];
      }
    }
  } else {
    for (int e = tid; e < TQ * kk; e += KNN_THREADS) {
      const int row = e / kk, t = e % kk;
      const int64_t q = q0 + row;
      if (q < A.n_q) A.part[(q * S + sp) * kk + t] = sKey[row * KL + t];
    }
  }
}

// k best of the S partial lists of a query (each ascending) -> 1-based ids / distances, column-major.
// perm_q (pruned search): the query's position in the caller's block.
__global__ __launch_bounds__(256) void k_knn_merge(const u64* __restrict__ part, int64_t n_q, int S, int kk, int metric,
                                                   const int32_t* __restrict__ perm_q, int32_t* __restrict__ idx,
                                                   float* __restrict__ dist, int64_t ld_out, const uint32_t* gate, uint32_t gate_want) {
  if (gate != nullptr && *gate != gate_want) return;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_q) return;
  const int64_t oq = perm_q ? (int64_t)perm_q[q] : q;
  if (oq < 0) return;                                // padding row of the reordered query block
  int pos[KNN_MAX_SPLIT];
#pragma unroll
  for (int s = 0; s < KNN_MAX_SPLIT; ++s) pos[s] = 0;
  const u64* const base = part + q * S * kk;
  for (int t = 0; t < kk; ++t) {
    u64 best = ~0ull;
    int bs = 0;
#pragma unroll
    for (int s = 0; s < KNN_MAX_SPLIT; ++s) {
      if (s < S && pos[s] < kk) {
        const u64 v = base[s * kk + pos[s]];
        if (v < best) { best = v; bs = s; }
      }
    }
#pragma unroll
    for (int s = 0; s < KNN_MAX_SPLIT; ++s) pos[s] += (s == bs && best != ~0ull) ? 1 : 0;
    int32_t id = 0;
    float dv = INFINITY;
    if (best != ~0ull) {
      id = (int32_t)(uint32_t)best + 1;
      dv = sortable_f32((uint32_t)(best >> 32));
      if (metric == GFICF_KNN_EUCLIDEAN) dv = sqrtf(dv);
    }
    idx[(int64_t)t * ld_out + oq] = id;
    if (dist) dist[(int64_t)t * ld_out + oq] = dv;
  }
}

// ------------------------------------------------------------------ pruned search: preparation
// The exact search skips whole candidate tiles that provably cannot contribute: with a centre c_t and radius r_t
// per candidate tile (r_t = largest distance of a tile point to c_t), every point x of the tile is at least
//   min over the query tile's points q of dist(q, c_t)  -  r_t
// away from every query of the tile (triangle inequality: manhattan and euclidean directly; cosine through the
// euclidean distance e of the unit vectors, 1 - cos = e^2 / 2).  For the bound to bite, the points are reordered so
// that a tile holds neighbours: every point is assigned to its nearest of ~N/128 pivots (the tile kernel itself
// with the pivots as candidates and k = 1) and the points are sorted by pivot.
constexpr int KNN_RQ = 4;
constexpr int KNN_TQ = 16 * KNN_RQ;
constexpr float KNN_LB_SLACK = 1e-4f;      // relative slack on the bound: f32 rounding of 128-dim sums is ~1e-5

__global__ __launch_bounds__(256) void k_knn_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ rows, int64_t n,
                                                         int nq4, int64_t stride_or_zero, float* __restrict__ dst) {
  // dst row r = src row rows[r] (rows == nullptr: src row r * stride_or_zero — evenly spaced pivots)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * nq4) return;
  const int64_t r = e / nq4;
  const int q4 = (int)(e % nq4);
  const int64_t sr = rows ? (int64_t)rows[r] : r * stride_or_zero;
  reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[sr * nq4 + q4];
}

__global__ __launch_bounds__(256) void k_knn_iota(int32_t* __restrict__ v, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) v[e] = (int32_t)e;
}

// sort key of a point: (coarse pivot of its fine pivot) << 13 | fine pivot   (ids 1-based, at most 4096 fine pivots)
__global__ __launch_bounds__(256) void k_knn_sort_keys(const int32_t* __restrict__ pivot_of, const int32_t* __restrict__ coarse_of,
                                                       int64_t n, int32_t* __restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) {
    const int32_t f = pivot_of[e];
    keys[e] = (coarse_of[f - 1] << 13) | f;
  }
}

// distance used by the bounds: the metric itself for manhattan, the euclidean distance otherwise
template <int METRIC>
__device__ inline float knn_bound_dist(const float* __restrict__ a, const float* __restrict__ b, int d) {
  float acc = 0.0f;
  for (int t = 0; t < d; ++t) {
    const float df = a[t] - b[t];
    acc = METRIC == GFICF_KNN_MANHATTAN ? acc + fabsf(df) : fmaf(df, df, acc);
  }
  return METRIC == GFICF_KNN_MANHATTAN ? acc : sqrtf(acc);
}

// one workgroup of 128 threads per candidate tile: centre = mean of the tile's points, radius = largest distance to it
template <int METRIC>
__global__ __launch_bounds__(KNN_TC) void k_knn_tile_stats(const float* __restrict__ X, const int32_t* __restrict__ tile_n, int d, int dpad,
                                                           float* __restrict__ centers, float* __restrict__ radius) {
  __shared__ float s_c[KNN_MAX_D];
  __shared__ float s_r[KNN_TC / 64];
  const int64_t t = blockIdx.x, j0 = t * KNN_TC;
  const int n = tile_n[t];
  if (n <= 0) { if (threadIdx.x == 0) radius[t] = -1.0f; return; }      // padding tile: marked empty
  for (int dim = threadIdx.x; dim < dpad; dim += KNN_TC) {
    float sum = 0.0f;
    for (int p = 0; p < n; ++p) sum += X[(j0 + p) * dpad + dim];
    const float c = sum / (float)n;
    s_c[dim] = c;
    centers[(int64_t)dim * gridDim.x + t] = c;        // [dim][tile]: k_knn_lb reads a dim of consecutive tiles at a time
  }
  __syncthreads();
  float r = 0.0f;
  if ((int)threadIdx.x < n) r = knn_bound_dist<METRIC>(X + (j0 + threadIdx.x) * dpad, s_c, d);
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) r = fmaxf(r, __shfl_xor(r, w));
  if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) radius[t] = fmaxf(s_r[0], s_r[1]);
}

// one workgroup per query tile: lb[qt][ct] = the bound above, in the domain of the keys (manhattan: distance,
// euclidean: squared distance, cosine: 1 - cos), lowered by the slack; never negative.
// It also counts the pairs that are certain to be pruned, for the choice between the two forms of the search: any
// candidate tile with at least kk points bounds every query's k-th best from above by max_q dist(q, centre) + radius
// (all its points are that close); U = the smallest such bound over the tiles; a pair with lb > U is never visited.
template <int METRIC>
__global__ __launch_bounds__(256) void k_knn_lb(const float* __restrict__ Q, const int32_t* __restrict__ qtile_n, int d, int dpad,
                                                const float* __restrict__ centers, const float* __restrict__ radius,
                                                const int32_t* __restrict__ tile_n, int kk, int64_t n_ct, float* __restrict__ lb,
                                                unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) float s_q[];   // [dpad][KNN_TQ]: a dimension of the 64 queries is contiguous (rows beyond nq: zeros, never used)
  __shared__ float s_u[4];
  const int64_t qt = blockIdx.x, q0 = qt * KNN_TQ;
  const int nq = qtile_n[qt];
  if (nq <= 0) return;                               // padding tile: its workgroup of the search exits at once
  for (int e = threadIdx.x; e < KNN_TQ * dpad; e += 256) {
    const int row = e / dpad, dim = e % dpad;
    s_q[dim * KNN_TQ + row] = row < nq ? Q[(q0 + row) * dpad + dim] : 0.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // cosine / correlation keys are 1 - dot of unit vectors, a d-term f32 sum: its computed value is within d * 2^-24 of the
  // exact one (sum |a_i b_i| <= 1), plus a few roundings of the normalisation; the absolute slack is twice that bound
  const float key_abs_slack = (float)(d + 8) * 1.1920929e-7f;
  auto to_key = [key_abs_slack](float e, float slack_sign) {     // distance of the bound's metric -> key domain, pushed by the slack
    if (METRIC == GFICF_KNN_MANHATTAN) return e;
    if (METRIC == GFICF_KNN_EUCLIDEAN) return e * e * (1.0f + slack_sign * KNN_LB_SLACK);
    return 0.5f * e * e * (1.0f + slack_sign * KNN_LB_SLACK) + slack_sign * key_abs_slack;
  };
  // one thread per candidate tile: the distance of EVERY query of the tile to the candidate tile's centre c (round 6; through round 5
  // the bound went through the query tile's own centre, dist(cq, c) - rq - r: in 50 dimensions the queries all sit at nearly the same
  // distance from c, far above that — on the config-3 stand-in 97 % of the pairs are provably out of reach this way, 82 % before).
  // (Screening the pairs with the old bound first and refining only those it leaves in reach was tried: its own upper bound of the k-th
  // best is so loose in 50 dimensions that every pair is left in reach — slower by the screening.)
  // Every point x of the candidate tile is within r of c, so  dist(q, x) >= min_q dist(q, c) - r  and  <= max_q dist(q, c) + r.
  float u = INFINITY;                                // upper bound of the queries' k-th best (key domain)
  for (int64_t c = threadIdx.x; c < n_ct; c += 256) {
    const float r = radius[c];
    if (r < 0.0f) { lb[qt * n_ct + c] = INFINITY; continue; }      // empty candidate tile: never visited
    float mn = INFINITY, mx = 0.0f;
    for (int g0 = 0; g0 < nq; g0 += 16) {             // sixteen queries at a time (their coordinates: LDS broadcast reads)
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
      for (int t = 0; t < d; ++t) {
        const float cv = centers[(int64_t)t * n_ct + c];            // [dim][tile]: consecutive threads, consecutive tiles
        const float4* q4 = reinterpret_cast<const float4*>(s_q + t * KNN_TQ + g0);     // four 16 B broadcast reads
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 q = q4[j4];
          const float qa[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float df = qa[j] - cv;
            acc[4 * j4 + j] = METRIC == GFICF_KNN_MANHATTAN ? acc[4 * j4 + j] + fabsf(df) : fmaf(df, df, acc[4 * j4 + j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (g0 + j < nq) {
          const float e = METRIC == GFICF_KNN_MANHATTAN ? acc[j] : sqrtf(acc[j]);
          mn = fminf(mn, e); mx = fmaxf(mx, e);
        }
      }
    }
    float b = mn - r - KNN_LB_SLACK * (mn + r);
    b = b > 0.0f ? to_key(b, -1.0f) : 0.0f;
    lb[qt * n_ct + c] = b > 0.0f ? b : 0.0f;
    if (tile_n[c] >= kk) u = fminf(u, to_key((mx + r) * (1.0f + KNN_LB_SLACK), 1.0f));
  }
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) u = fminf(u, __shfl_xor(u, w));
  if (lane == 0) s_u[wave] = u;
  __syncthreads();
  const float U = fminf(fminf(s_u[0], s_u[1]), fminf(s_u[2], s_u[3]));
  int npr = 0, np = 0;                               // pairs certain to be pruned / all pairs with a real candidate tile
  for (int64_t c = threadIdx.x; c < n_ct; c += 256) {
    const float b = lb[qt * n_ct + c];               // this thread's own stores
    if (b < INFINITY) { ++np; npr += b > U ? 1 : 0; }
  }
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) { npr += __shfl_xor(npr, w); np += __shfl_xor(np, w); }
  if (lane == 0) { atomicAdd(counters, (unsigned long long)npr); atomicAdd(counters + 1, (unsigned long long)np); }
}

// flag = 1 (plain search) when fewer than half of the (query tile, candidate tile) pairs are certain to be pruned: the
// data has too little cluster structure, and the plain form with its split candidate range runs faster
__global__ void k_knn_choose(const unsigned long long* __restrict__ counters, uint32_t force, uint32_t* __restrict__ flag) {
  *flag = force ? 0u : (2ull * counters[0] < counters[1] ? 1u : 0u);
}

// ---- cells padded to whole tiles
// The points are sorted by cell key; every cell is moved to a position that is a multiple of the tile size T, so that
// no tile holds points of two cells (a straddling tile spans both and its radius makes it unprunable for everybody
// near either cell).  Rows in between are padding: zero coordinates, id -1, never a candidate, never a query.
__global__ __launch_bounds__(256) void k_knn_cell_heads(const int32_t* __restrict__ keys, int64_t n, int64_t* __restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p > n) return;
  flags[p] = (p < n && (p == 0 || keys[p] != keys[p - 1])) ? 1 : 0;
}
// rank[] = exclusive scan of the head flags (rank[n] = number of cells); cstart[c] = first sorted position of cell c
__global__ __launch_bounds__(256) void k_knn_cell_starts(const int32_t* __restrict__ keys, const int64_t* __restrict__ rank, int64_t n,
                                                         int32_t* __restrict__ cstart) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p > n) return;
  if (p == n) { cstart[rank[n]] = (int32_t)n; return; }
  if (p == 0 || keys[p] != keys[p - 1]) cstart[rank[p]] = (int32_t)p;
}
// one workgroup: pstart[c] = padded start of cell c (exclusive scan of the cells' sizes rounded up to T), pstart[n_cells] =
// padded total; at most 4097 cells
__global__ __launch_bounds__(1024) void k_knn_cell_offsets(const int32_t* __restrict__ cstart, const int64_t* __restrict__ n_cells_p, int T,
                                                           int32_t* __restrict__ pstart) {
  __shared__ int s_part[1024];
  __shared__ int s_total;
  const int n_cells = (int)*n_cells_p;
  const int per = (n_cells + 1023) / 1024;
  const int c0 = threadIdx.x * per;
  int sum = 0;
  for (int c = c0; c < c0 + per && c < n_cells; ++c) sum += (cstart[c + 1] - cstart[c] + T - 1) / T * T;
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 1024; ++t) { const int v = s_part[t]; s_part[t] = run; run += v; }
    s_total = run;
  }
  __syncthreads();
  int run = s_part[threadIdx.x];
  for (int c = c0; c < c0 + per && c < n_cells; ++c) { pstart[c] = run; run += (cstart[c + 1] - cstart[c] + T - 1) / T * T; }
  if (threadIdx.x == 0) pstart[n_cells] = s_total;
}
// padded position of every sorted point; copy of its row; its id
__global__ __launch_bounds__(256) void k_knn_pad_scatter(const int32_t* __restrict__ keys, const int64_t* __restrict__ rank,
                                                         const int32_t* __restrict__ cstart, const int32_t* __restrict__ pstart,
                                                         const int32_t* __restrict__ perm, int64_t n, int nq4,
                                                         const float* __restrict__ src, float* __restrict__ dst, int32_t* __restrict__ perm_pad) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * nq4) return;
  const int64_t p = e / nq4;
  const int q4 = (int)(e % nq4);
  const bool head = p == 0 || keys[p] != keys[p - 1];
  const int64_t c = rank[p] + (head ? 1 : 0) - 1;
  const int64_t pp = (int64_t)pstart[c] + (p - cstart[c]);
  const int32_t id = perm[p];
  reinterpret_cast<float4*>(dst)[pp * nq4 + q4] = reinterpret_cast<const float4*>(src)[(int64_t)id * nq4 + q4];
  if (q4 == 0) perm_pad[pp] = id;
}
// real rows per tile (they are a prefix of the tile)
__global__ __launch_bounds__(256) void k_knn_tile_counts(const int32_t* __restrict__ perm_pad, int64_t n_tiles, int T, int32_t* __restrict__ tile_n) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tiles) return;
  int n = 0;
  for (int r = 0; r < T; ++r) n += perm_pad[t * T + r] >= 0 ? 1 : 0;
  tile_n[t] = n;
}

int knn_split(const gficf_ctx* ctx, int64_t n_q, int64_t N) {
  const int64_t n_qt = gficf_ceil_div(n_q > 0 ? n_q : 1, KNN_TQ), n_ct = gficf_ceil_div(N > 0 ? N : 1, KNN_TC);
  // enough work items for ~8 per CU, but every candidate slice at least 8 tiles long
  int64_t S = gficf_ceil_div((int64_t)ctx->num_cus * 8, n_qt);
  if (S > n_ct / 8) S = n_ct / 8;
  if (S > KNN_MAX_SPLIT) S = KNN_MAX_SPLIT;
  if (S < 1) S = 1;
  if (const char* e = getenv("GFICF_KNN_SPLIT")) {   // tuning knob / test hook
    const int v = atoi(e);
    if (v >= 1 && v <= KNN_MAX_SPLIT) S = v;
  }
  return (int)S;
}

// The pruned search pays for its preparation (~1 ms at 100 k points) only on larger inputs; the bounds of a query
// tile have to fit LDS.  GFICF_KNN_PRUNE=0 / 1 forces it off / on (test hook).
constexpr int64_t KNN_PRUNE_MIN_N = 30000;
constexpr int64_t KNN_PRUNE_MAX_TILES = 8192;
bool knn_use_prune(int64_t N) {
  const int64_t n_ct = gficf_ceil_div(N > 0 ? N : 1, KNN_TC);
  if (n_ct < 2 || n_ct > KNN_PRUNE_MAX_TILES) return false;
  if (const char* e = getenv("GFICF_KNN_PRUNE")) return atoi(e) != 0;
  return N >= KNN_PRUNE_MIN_N;
}
// Two levels of pivots: C fine pivots (a cell of ~KNN_CELL points each) and C/16 coarse ones.  Points are ordered by
// (coarse pivot of their fine pivot, fine pivot), so that cells that follow each other in memory are neighbours in
// space and a 128-point tile that straddles two cells stays compact.
inline int64_t knn_coarse(int64_t C) { return C / 16 > 2 ? C / 16 : 2; }

int64_t knn_pivots(int64_t N) {
  // points per pivot.  Round 6 swept it (one box, GFICF_KNN_PIVOT_CELL): on 30 blobs in 50 dimensions larger cells win from 400 k points on
  // (400 k: 36.8 ms at 256, 28.0 at 1024; 1 M: 303 / 207 / 186 ms at 244 / 1024 / 4096 — cells are padded to whole tiles and the per-tile
  // bounds grow with the number of cells), but a cell has to stay finer than the data's clusters: 1 M points in 200 clusters take 221 ms at
  // 256 and 2 358 ms at 1 953 (nothing left to prune: the plain form).  The size that is safe for both stays; bounds per cell first and per
  // tile inside the cells in reach would lift the cost that grows with the cells (not built).
  int64_t per = 256;
  if (const char* e = getenv("GFICF_KNN_PIVOT_CELL")) per = atoi(e) > 0 ? atoi(e) : per;    // lab knob: points per pivot
  int64_t c = gficf_ceil_div(N, per);
  return c < 2 ? 2 : c > 4096 ? 4096 : c;
}

int knn_check(int64_t N, int d, int k, int metric) {
  if (N < 0 || d < 0 || k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  if (d > KNN_MAX_D) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "d = %d exceeds %d dimensions", d, KNN_MAX_D);
  if (k > GFICF_KNN_MAX_K) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d exceeds GFICF_KNN_MAX_K = %d", k, GFICF_KNN_MAX_K);
  if (k > N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "k = %d neighbours asked of N = %lld points", k, (long long)N);
  if (metric != GFICF_KNN_MANHATTAN && metric != GFICF_KNN_EUCLIDEAN && metric != GFICF_KNN_COSINE && metric != GFICF_KNN_CORRELATION)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "unknown metric %d", metric);
  return GFICF_OK;
}

template <int METRIC, int KL, bool PRUNE>
int knn_launch(gficf_ctx* ctx, const KnnTileArgs& a) {
  const int64_t n_ct = gficf_ceil_div(a.N, KNN_TC);
  const size_t lds = (size_t)a.dpad * KNN_TQ * 4 + 2 * KNN_DK * KNN_TC * 4 + (KL > 64 ? (size_t)KNN_TQ * KL * 8 : 0) + (PRUNE ? (size_t)n_ct * 4 : 0);
  static std::atomic<bool> attr_set[64];
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_knn_tiles<METRIC, KL, KNN_RQ, PRUNE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    attr_set[ctx->device & 63] = true;
  }
  const int64_t blocks = gficf_ceil_div(a.n_q, KNN_TQ) * a.S;
  hipLaunchKernelGGL((k_knn_tiles<METRIC, KL, KNN_RQ, PRUNE>), dim3((unsigned)blocks), dim3(KNN_THREADS), lds, ctx->stream, a);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int METRIC, bool PRUNE>
int knn_launch_k(gficf_ctx* ctx, const KnnTileArgs& a) {
  if (a.kk <= 32) return knn_launch<METRIC, 32, PRUNE>(ctx, a);
  if (a.kk <= 64) return knn_launch<METRIC, 64, PRUNE>(ctx, a);
  return knn_launch<METRIC, 128, PRUNE>(ctx, a);
}

template <bool PRUNE>
int knn_launch_m(gficf_ctx* ctx, int metric, const KnnTileArgs& a) {
  switch (metric) {
    case GFICF_KNN_MANHATTAN: return knn_launch_k<GFICF_KNN_MANHATTAN, PRUNE>(ctx, a);
    case GFICF_KNN_EUCLIDEAN: return knn_launch_k<GFICF_KNN_EUCLIDEAN, PRUNE>(ctx, a);
    default: return knn_launch_k<GFICF_KNN_COSINE, PRUNE>(ctx, a);
  }
}

inline size_t knn_align(size_t b) { return (b + 255) & ~(size_t)255; }

size_t knn_sort_temp_bytes(int64_t n) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u, 32u,
                                  (hipStream_t) nullptr);
  return tmp;
}

// workspace of the pruned search, carved in this order.  Rows / tiles are counted with the padding of the cell-aligned
// layout at its worst (every cell adds less than one tile).
struct KnnPruneWs {
  int64_t C, xrows, qrows, n_ct, n_qt;      // pivots; padded candidate / query rows; candidate / query tiles (upper bounds)
  float *pivots, *coarse, *xp, *qp, *centers, *radius, *lb;
  u64* apart;                       // assignment: N lists of one key (also used for the pivots' own assignment)
  int32_t *pivot_of, *coarse_of, *keys, *key_tmp, *iota, *perm_sorted, *perm_x, *perm_q, *cstart, *pstart, *tile_n, *qtile_n;
  int64_t* rank;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  u64* part;
  u64* part_plain;                  // the plain search's partial lists (taken when the data does not prune)
  unsigned long long* counters;     // [0] zero-bound pairs, [1] pairs, [2] (as uint32) the choice flag
  size_t total;
};

KnnPruneWs knn_prune_ws(char* base, int64_t n_q, int64_t N, int dpad, int k) {
  KnnPruneWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += knn_align(bytes); return (void*)p; };
  const int64_t C = knn_pivots(N);
  w.C = C;
  w.n_ct = gficf_ceil_div(N, KNN_TC) + (C < N ? C : N);
  w.n_qt = gficf_ceil_div(n_q, KNN_TQ) + (C < n_q ? C : n_q);
  w.xrows = w.n_ct * KNN_TC;
  w.qrows = w.n_qt * KNN_TQ;
  w.pivots = (float*)take((size_t)C * dpad * 4);
  w.coarse = (float*)take((size_t)knn_coarse(C) * dpad * 4);
  w.coarse_of = (int32_t*)take((size_t)C * 4);
  w.keys = (int32_t*)take((size_t)N * 4);
  w.xp = (float*)take((size_t)w.xrows * dpad * 4);
  w.qp = (float*)take((size_t)w.qrows * dpad * 4);
  w.centers = (float*)take((size_t)w.n_ct * dpad * 4);
  w.radius = (float*)take((size_t)w.n_ct * 4);
  w.lb = (float*)take((size_t)w.n_qt * w.n_ct * 4);
  w.apart = (u64*)take((size_t)N * 8);
  w.pivot_of = (int32_t*)take((size_t)N * 4);
  w.key_tmp = (int32_t*)take((size_t)N * 4);
  w.iota = (int32_t*)take((size_t)N * 4);
  w.perm_sorted = (int32_t*)take((size_t)N * 4);
  w.perm_x = (int32_t*)take((size_t)w.xrows * 4);
  w.perm_q = (int32_t*)take((size_t)w.qrows * 4);
  w.cstart = (int32_t*)take((size_t)(C + 2) * 4);
  w.pstart = (int32_t*)take((size_t)(C + 2) * 4);
  w.tile_n = (int32_t*)take((size_t)w.n_ct * 4);
  w.qtile_n = (int32_t*)take((size_t)w.n_qt * 4);
  w.rank = (int64_t*)take((size_t)(N + 1) * 8);
  w.sort_tmp_bytes = knn_sort_temp_bytes(N);
  w.sort_tmp = take(w.sort_tmp_bytes);
  w.part = (u64*)take((size_t)w.qrows * (size_t)k * 8);
  w.part_plain = (u64*)take((size_t)n_q * (size_t)KNN_MAX_SPLIT * (size_t)k * 8);
  w.counters = (unsigned long long*)take(32);
  w.total = off + 256;
  return w;
}

}  // namespace

extern "C" {

int gficf_knn_dpad(int d) { return (d < 0 || d > KNN_MAX_D) ? -1 : knn_dpad(d); }

int gficf_knn_prepare_device(gficf_ctx* ctx, const void* d_X, int x_is_f64, int64_t n_rows, int d, int64_t ld, int metric,
                             float* d_point_rows) {
  GFICF_CTX_ENTER(ctx);
  int rc = knn_check(n_rows, d, 0, metric);
  if (rc) return rc;
  if (n_rows == 0 || d == 0) return GFICF_OK;
  if (!d_X || !d_point_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ld < n_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < n_rows = %lld", (long long)ld, (long long)n_rows);
  const unsigned blocks = (unsigned)gficf_ceil_div(n_rows, 256);
  if (x_is_f64)
    hipLaunchKernelGGL(k_knn_prepare<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)d_X, n_rows, d, knn_dpad(d), ld, metric, d_point_rows, ctx->d_status);
  else
    hipLaunchKernelGGL(k_knn_prepare<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)d_X, n_rows, d, knn_dpad(d), ld, metric, d_point_rows, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

size_t gficf_knn_workspace_bytes(gficf_ctx* ctx, int64_t n_queries, int64_t N, int k) {
  if (!ctx || n_queries <= 0 || N <= 0 || k <= 0) return 256;
  // the larger of the two forms, so that the size does not depend on the tuning knobs
  const size_t plain = (size_t)n_queries * (size_t)KNN_MAX_SPLIT * (size_t)k * sizeof(u64) + 256;
  const size_t small = (size_t)n_queries * (size_t)knn_split(ctx, n_queries, N) * (size_t)k * sizeof(u64) + 256;
  const size_t pruned = gficf_ceil_div(N, KNN_TC) <= KNN_PRUNE_MAX_TILES ? knn_prune_ws(nullptr, n_queries, N, knn_dpad(KNN_MAX_D), k).total : 0;
  (void)plain;
  return small > pruned ? small : pruned;
}

int gficf_knn_search_device(gficf_ctx* ctx, const float* d_points, int64_t N, int d, int k, int metric, int64_t q_begin,
                            int64_t q_end, void* d_ws, size_t ws_bytes, int32_t* d_idx, float* d_dist, int64_t ld_out) {
  GFICF_CTX_ENTER(ctx);
  int rc = knn_check(N, d, k, metric);
  if (rc) return rc;
  if (q_begin < 0 || q_end < q_begin || q_end > N)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "query range [%lld, %lld) outside [0, %lld]", (long long)q_begin, (long long)q_end, (long long)N);
  const int64_t n_q = q_end - q_begin;
  if (n_q == 0 || k == 0) return GFICF_OK;
  if (d == 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "points have no dimensions");
  if (!d_points || !d_ws || !d_idx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ld_out < n_q) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld_out = %lld < number of queries %lld", (long long)ld_out, (long long)n_q);
  if (ws_bytes < gficf_knn_workspace_bytes(ctx, n_q, N, k)) GFICF_FAIL(GFICF_ERR_CAPACITY, "kNN workspace too small");
  if (metric == GFICF_KNN_CORRELATION) metric = GFICF_KNN_COSINE;      // the prepared rows are centred: same search
  const int dpad = knn_dpad(d);
  const unsigned mblocks = (unsigned)gficf_ceil_div(n_q, 256);
  KnnTileArgs a{};
  a.d = d; a.dpad = dpad; a.kk = k;

  if (!knn_use_prune(N)) {
    // plain search: every query tile against every candidate tile, the candidate range split S ways
    a.Q = d_points + q_begin * dpad; a.n_q = n_q; a.X = d_points; a.N = N;
    a.S = knn_split(ctx, n_q, N);
    a.part = (u64*)d_ws;
    rc = knn_launch_m<false>(ctx, metric, a);
    if (rc) return rc;
    hipLaunchKernelGGL(k_knn_merge, dim3(mblocks), dim3(256), 0, ctx->stream, a.part, n_q, a.S, k, metric, (const int32_t*)nullptr, d_idx, d_dist, ld_out, (const uint32_t*)nullptr, 0u);
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  }

  // pruned search (see "pruned search: preparation" above)
  const KnnPruneWs w = knn_prune_ws((char*)d_ws, n_q, N, dpad, k);
  const int nq4 = dpad >> 2;
  const int64_t C = w.C;
  auto blocks_for = [](int64_t n) { return dim3((unsigned)gficf_ceil_div(n, 256)); };
  // 1. pivots = C evenly spaced points; every point's nearest pivot (the tile kernel, k = 1)
  hipLaunchKernelGGL(k_knn_gather_rows, blocks_for(C * nq4), dim3(256), 0, ctx->stream, d_points, (const int32_t*)nullptr, C, nq4, N / C, w.pivots);
  KnnTileArgs as{};
  as.Q = d_points; as.n_q = N; as.X = w.pivots; as.N = C; as.d = d; as.dpad = dpad; as.kk = 1; as.S = 1; as.part = w.apart;
  rc = knn_launch_m<false>(ctx, metric, as);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, blocks_for(N), dim3(256), 0, ctx->stream, w.apart, N, 1, 1, metric, (const int32_t*)nullptr, w.pivot_of, (float*)nullptr, N, (const uint32_t*)nullptr, 0u);
  // the fine pivots' own nearest coarse pivot (coarse pivots = every (C / Cc)-th fine pivot)
  const int64_t Cc = knn_coarse(C);
  hipLaunchKernelGGL(k_knn_gather_rows, blocks_for(Cc * nq4), dim3(256), 0, ctx->stream, w.pivots, (const int32_t*)nullptr, Cc, nq4, C / Cc, w.coarse);
  KnnTileArgs ac{};
  ac.Q = w.pivots; ac.n_q = C; ac.X = w.coarse; ac.N = Cc; ac.d = d; ac.dpad = dpad; ac.kk = 1; ac.S = 1; ac.part = w.apart;
  rc = knn_launch_m<false>(ctx, metric, ac);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, blocks_for(C), dim3(256), 0, ctx->stream, w.apart, C, 1, 1, metric, (const int32_t*)nullptr, w.coarse_of, (float*)nullptr, C, (const uint32_t*)nullptr, 0u);
  // 2. candidates sorted by (coarse, fine) pivot (stable: ties keep index order) and laid out cell by cell, every cell
  //    starting on a tile boundary; the queries of this block likewise, with the query tile size
  hipLaunchKernelGGL(k_knn_sort_keys, blocks_for(N), dim3(256), 0, ctx->stream, w.pivot_of, w.coarse_of, N, w.keys);
  hipLaunchKernelGGL(k_knn_iota, blocks_for(N), dim3(256), 0, ctx->stream, w.iota, N);
  GFICF_HIP_CHECK(hipMemsetAsync(w.xp, 0, sizeof(float) * (size_t)w.xrows * dpad, ctx->stream));
  GFICF_HIP_CHECK(hipMemsetAsync(w.qp, 0, sizeof(float) * (size_t)w.qrows * dpad, ctx->stream));
  GFICF_HIP_CHECK(hipMemsetAsync(w.perm_x, 0xFF, sizeof(int32_t) * (size_t)w.xrows, ctx->stream));
  GFICF_HIP_CHECK(hipMemsetAsync(w.perm_q, 0xFF, sizeof(int32_t) * (size_t)w.qrows, ctx->stream));
  auto layout = [&](const int32_t* keys_in, int64_t n, int T, const float* src, float* dst, int32_t* perm_pad, int64_t n_tiles, int32_t* tile_n) -> int {
    size_t tb = w.sort_tmp_bytes;
    GFICF_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, keys_in, w.key_tmp, w.iota, w.perm_sorted, (size_t)n, 0u, 26u, ctx->stream));
    hipLaunchKernelGGL(k_knn_cell_heads, blocks_for(n + 1), dim3(256), 0, ctx->stream, w.key_tmp, n, w.rank);
    int r = gficf_exclusive_scan_i64(ctx, w.rank, n + 1);
    if (r) return r;
    hipLaunchKernelGGL(k_knn_cell_starts, blocks_for(n + 1), dim3(256), 0, ctx->stream, w.key_tmp, w.rank, n, w.cstart);
    hipLaunchKernelGGL(k_knn_cell_offsets, dim3(1), dim3(1024), 0, ctx->stream, w.cstart, w.rank + n, T, w.pstart);
    hipLaunchKernelGGL(k_knn_pad_scatter, blocks_for(n * nq4), dim3(256), 0, ctx->stream, w.key_tmp, w.rank, w.cstart, w.pstart, w.perm_sorted, n, nq4,
                       src, dst, perm_pad);
    hipLaunchKernelGGL(k_knn_tile_counts, blocks_for(n_tiles), dim3(256), 0, ctx->stream, perm_pad, n_tiles, T, tile_n);
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  };
  rc = layout(w.keys, N, KNN_TC, d_points, w.xp, w.perm_x, w.n_ct, w.tile_n);
  if (rc) return rc;
  rc = layout(w.keys + q_begin, n_q, KNN_TQ, d_points + q_begin * dpad, w.qp, w.perm_q, w.n_qt, w.qtile_n);
  if (rc) return rc;
  // 3. centre and radius of every candidate tile; bound of every (query tile, candidate tile) pair
  GFICF_HIP_CHECK(hipMemsetAsync(w.counters, 0, 32, ctx->stream));
  const size_t lds_lb = (size_t)KNN_TQ * dpad * sizeof(float);
  switch (metric) {
    case GFICF_KNN_MANHATTAN:
      hipLaunchKernelGGL(k_knn_tile_stats<GFICF_KNN_MANHATTAN>, dim3((unsigned)w.n_ct), dim3(KNN_TC), 0, ctx->stream, w.xp, w.tile_n, d, dpad, w.centers, w.radius);
      hipLaunchKernelGGL(k_knn_lb<GFICF_KNN_MANHATTAN>, dim3((unsigned)w.n_qt), dim3(256), lds_lb, ctx->stream, w.qp, w.qtile_n, d, dpad, w.centers, w.radius, w.tile_n, k, w.n_ct, w.lb, w.counters);
      break;
    case GFICF_KNN_EUCLIDEAN:
      hipLaunchKernelGGL(k_knn_tile_stats<GFICF_KNN_EUCLIDEAN>, dim3((unsigned)w.n_ct), dim3(KNN_TC), 0, ctx->stream, w.xp, w.tile_n, d, dpad, w.centers, w.radius);
      hipLaunchKernelGGL(k_knn_lb<GFICF_KNN_EUCLIDEAN>, dim3((unsigned)w.n_qt), dim3(256), lds_lb, ctx->stream, w.qp, w.qtile_n, d, dpad, w.centers, w.radius, w.tile_n, k, w.n_ct, w.lb, w.counters);
      break;
    default:
      hipLaunchKernelGGL(k_knn_tile_stats<GFICF_KNN_COSINE>, dim3((unsigned)w.n_ct), dim3(KNN_TC), 0, ctx->stream, w.xp, w.tile_n, d, dpad, w.centers, w.radius);
      hipLaunchKernelGGL(k_knn_lb<GFICF_KNN_COSINE>, dim3((unsigned)w.n_qt), dim3(256), lds_lb, ctx->stream, w.qp, w.qtile_n, d, dpad, w.centers, w.radius, w.tile_n, k, w.n_ct, w.lb, w.counters);
      break;
  }
  GFICF_HIP_CHECK(hipGetLastError());
  // 4. the search over the reordered copies, tiles in best-first order with early exit; 5. back to the caller's order
  // ... unless the bounds say there is nothing to prune: then the plain search runs instead (chosen on the device,
  // both forms are enqueued and the one not chosen returns at once)
  uint32_t* const flag = (uint32_t*)(w.counters + 2);
  const char* fe = getenv("GFICF_KNN_PRUNE");
  hipLaunchKernelGGL(k_knn_choose, dim3(1), dim3(1), 0, ctx->stream, w.counters, (uint32_t)(fe && atoi(fe) != 0), flag);
  KnnTileArgs ap{};
  ap.Q = d_points + q_begin * dpad; ap.n_q = n_q; ap.X = d_points; ap.N = N; ap.d = d; ap.dpad = dpad; ap.kk = k;
  ap.S = knn_split(ctx, n_q, N); ap.part = w.part_plain; ap.gate = flag; ap.gate_want = 1u;
  rc = knn_launch_m<false>(ctx, metric, ap);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, dim3(mblocks), dim3(256), 0, ctx->stream, w.part_plain, n_q, ap.S, k, metric, (const int32_t*)nullptr, d_idx, d_dist, ld_out, (const uint32_t*)flag, 1u);
  a.Q = w.qp; a.n_q = w.qrows; a.X = w.xp; a.N = w.xrows; a.S = 1; a.part = w.part; a.perm_x = w.perm_x; a.lb = w.lb;
  a.tile_n = w.tile_n; a.qtile_n = w.qtile_n; a.gate = flag; a.gate_want = 0u;
  rc = knn_launch_m<true>(ctx, metric, a);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, blocks_for(w.qrows), dim3(256), 0, ctx->stream, w.part, w.qrows, 1, k, metric, (const int32_t*)w.perm_q, d_idx, d_dist, ld_out, (const uint32_t*)flag, 0u);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

/* The cell order of the pruned search, for callers that want ids with LOCALITY (the sharded Jaccard build's halo form,
 * gficf_amd/dist.py): d_order[p] = 0-based row of the point at position p when the points are sorted by their (coarse, fine)
 * pivot — steps 1-2 of the pruned search above, nothing else.  Points whose nearest neighbours share their pivot cell end up
 * next to each other, so a contiguous block of the order names few rows outside itself.  A function of the points alone: every
 * rank of a sharded job computes the same order from the same (all-gathered) points.  Workspace as for a search with N queries. */
int gficf_knn_pivot_order_device(gficf_ctx* ctx, const float* d_points, int64_t N, int d, int metric, void* d_ws, size_t ws_bytes,
                                 int32_t* d_order) {
  GFICF_CTX_ENTER(ctx);
  int rc = knn_check(N, d, 1, metric);
  if (rc) return rc;
  if (N == 0) return GFICF_OK;
  if (d == 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "points have no dimensions");
  if (!d_points || !d_ws || !d_order) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ws_bytes < gficf_knn_workspace_bytes(ctx, N, N, 1)) GFICF_FAIL(GFICF_ERR_CAPACITY, "kNN workspace too small");
  if (metric == GFICF_KNN_CORRELATION) metric = GFICF_KNN_COSINE;
  auto blocks_for = [](int64_t n) { return dim3((unsigned)gficf_ceil_div(n, 256)); };
  if (gficf_ceil_div(N, KNN_TC) > KNN_PRUNE_MAX_TILES || knn_pivots(N) < 2) {      // no pivots at this size: the order as given
    hipLaunchKernelGGL(k_knn_iota, blocks_for(N), dim3(256), 0, ctx->stream, d_order, N);
    GFICF_HIP_CHECK(hipGetLastError());
    return GFICF_OK;
  }
  const int dpad = knn_dpad(d);
  const KnnPruneWs w = knn_prune_ws((char*)d_ws, N, N, dpad, 1);
  const int nq4 = dpad >> 2;
  const int64_t C = w.C;
  hipLaunchKernelGGL(k_knn_gather_rows, blocks_for(C * nq4), dim3(256), 0, ctx->stream, d_points, (const int32_t*)nullptr, C, nq4, N / C, w.pivots);
  KnnTileArgs as{};
  as.Q = d_points; as.n_q = N; as.X = w.pivots; as.N = C; as.d = d; as.dpad = dpad; as.kk = 1; as.S = 1; as.part = w.apart;
  rc = knn_launch_m<false>(ctx, metric, as);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, blocks_for(N), dim3(256), 0, ctx->stream, w.apart, N, 1, 1, metric, (const int32_t*)nullptr, w.pivot_of, (float*)nullptr, N, (const uint32_t*)nullptr, 0u);
  const int64_t Cc = knn_coarse(C);
  hipLaunchKernelGGL(k_knn_gather_rows, blocks_for(Cc * nq4), dim3(256), 0, ctx->stream, w.pivots, (const int32_t*)nullptr, Cc, nq4, C / Cc, w.coarse);
  KnnTileArgs ac{};
  ac.Q = w.pivots; ac.n_q = C; ac.X = w.coarse; ac.N = Cc; ac.d = d; ac.dpad = dpad; ac.kk = 1; ac.S = 1; ac.part = w.apart;
  rc = knn_launch_m<false>(ctx, metric, ac);
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, blocks_for(C), dim3(256), 0, ctx->stream, w.apart, C, 1, 1, metric, (const int32_t*)nullptr, w.coarse_of, (float*)nullptr, C, (const uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(k_knn_sort_keys, blocks_for(N), dim3(256), 0, ctx->stream, w.pivot_of, w.coarse_of, N, w.keys);
  hipLaunchKernelGGL(k_knn_iota, blocks_for(N), dim3(256), 0, ctx->stream, w.iota, N);
  size_t tb = w.sort_tmp_bytes;
  GFICF_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys, w.key_tmp, w.iota, d_order, (size_t)N, 0u, 26u, ctx->stream));
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_knn_host(gficf_ctx* ctx, const double* X, int64_t N, int d, int64_t ld, int k, int metric, int32_t* idx, double* dist) {
  GFICF_CTX_ENTER(ctx);
  int rc = knn_check(N, d, k, metric);
  if (rc) return rc;
  if (N == 0 || k == 0) return GFICF_OK;
  if (d == 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "points have no dimensions");
  if (!X || !idx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  const size_t xb = sizeof(double) * (size_t)ld * (size_t)d, pb = sizeof(float) * (size_t)N * (size_t)knn_dpad(d);
  const size_t wsb = gficf_knn_workspace_bytes(ctx, N, N, k), ob = (size_t)N * (size_t)k;
  void *d_X = nullptr, *d_P = nullptr, *d_ws = nullptr, *d_out = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, xb, &d_X);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, pb, &d_P);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 2, wsb, &d_ws);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 3, ob * (sizeof(int32_t) + sizeof(float)), &d_out);
  if (e == hipSuccess) e = hipMemcpyAsync(d_X, X, xb, hipMemcpyHostToDevice, ctx->stream);
  std::vector<float> hd;
  if (e == hipSuccess) {
    int32_t* d_idx = (int32_t*)d_out;
    float* d_dist = (float*)(d_idx + ob);
    rc = gficf_knn_prepare_device(ctx, d_X, 1, N, d, ld, metric, (float*)d_P);
    if (!rc) rc = gficf_knn_search_device(ctx, (const float*)d_P, N, d, k, metric, 0, N, d_ws, wsb, d_idx, dist ? d_dist : nullptr, N);
    gficf_advise_hugepages(idx, ob * sizeof(int32_t));             // (fresh R matrices: first touched by the copies below)
    if (dist) gficf_advise_hugepages(dist, ob * sizeof(double));
    if (!rc) e = hipMemcpyAsync(idx, d_idx, ob * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess && dist) {
      hd.resize(ob);
      e = hipMemcpyAsync(hd.data(), d_dist, ob * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_knn_host: %s", hipGetErrorString(e));
  if (rc) return rc;
  if (dist)
    for (size_t t = 0; t < ob; ++t) dist[t] = (double)hd[t];
  return GFICF_OK;
}

}  // extern "C"
