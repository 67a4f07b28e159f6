// knn.hip — exact k-nearest-neighbour search for gfx950 (MI355X): "next" row N2 of the hot path.
//
// Stands in for the caller's step in front of the Jaccard build,
//   neigh = uwot:::find_nn(data$pca$cells, k = k+1, include_self = T, method = "annoy", metric = dist.method)$idx
// (reference R/clustCells.R:57,60; uwot/Annoy are third-party and approximate).  This is an EXACT
// search: every query is compared with every point in f32 (Annoy stores f32 too) and the k
// smallest (distance, index) pairs are kept, ties broken by the smaller index — so the result is
// unique and checkable bit for bit against a CPU brute force.
//
// Metrics: manhattan (the reference's default, R/clustCells.R:46), euclidean, cosine (1 - cos).
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
// of one wave, which take turns (wave-uniform loop, one elected lane per row and round): no locks.  The candidate range
// is split S ways to fill the chip; a merge kernel picks the k best of the S partial lists and
// writes the 1-based index matrix column-major — the layout the Jaccard ingest reads.
#include <cfloat>
#include <cstdlib>
#include <type_traits>
#include <vector>

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
  if (metric == GFICF_KNN_COSINE) {
    float s = 0.0f;
    for (int t = 0; t < d; ++t) { const float v = (float)X[(int64_t)t * ld + r]; s = fmaf(v, v, s); }
    const float nrm = sqrtf(s);
    scale = nrm > 0.0f;
    inv = nrm;
  }
  float* o = out + r * dpad;
  for (int t = 0; t < dpad; ++t) {
    float v = t < d ? (float)X[(int64_t)t * ld + r] : 0.0f;
    bad |= !(fabsf(v) <= FLT_MAX);              // NaN or +-Inf (also a double too large for f32)
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
                                             float d6, float d7, float tau, bool live, uint32_t j0, int tid) {
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

// RQ = query rows per thread: 8 (tile of 128 queries: rows ty*4.. and 64+ty*4..) or 4 (tile of 64 queries).
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

template <int METRIC, int KL, int RQ>
__global__ __launch_bounds__(KNN_THREADS, 2) void k_knn_tiles(const float* __restrict__ X, int64_t N, int d, int dpad, int kk,
                                                           int64_t q_begin, int64_t q_end, int S, u64* __restrict__ part) {
  constexpr int TQ = 16 * RQ;                                                // queries per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const sA = reinterpret_cast<float*>(smem);                          // [dpad][TQ]
  float* const sB = sA + (size_t)dpad * TQ;                              // [2][DK][TC]
  u64* const sKey = reinterpret_cast<u64*>(sB + 2 * KNN_DK * KNN_TC);        // [TQ][KL]
  const uint32_t key_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)(unsigned char*)sKey;   // LDS byte address

  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int qt = blockIdx.x / S, sp = blockIdx.x % S;
  const int64_t q0 = q_begin + (int64_t)qt * TQ;
  const int nq_live = q_end - q0 < TQ ? (int)(q_end - q0) : TQ;      // rows of the tile that are real queries
  const int64_t n_ct = gficf_ceil_div(N, KNN_TC);
  const int64_t ct0 = n_ct * sp / S, ct1 = n_ct * (sp + 1) / S;
  const int nq4 = dpad >> 2;                         // float4 per point row
  const int nch = (d + KNN_DK - 1) / KNN_DK;         // chunks per candidate tile (padded dims are skipped)
  const float4* const X4 = reinterpret_cast<const float4*>(X);

  for (int e = tid; e < TQ * KL; e += KNN_THREADS) sKey[e] = ~0ull;
  // query tile -> sA[dim][query]; consecutive lanes take consecutive queries (conflict-free LDS writes)
  for (int f = tid; f < TQ * nq4; f += KNN_THREADS) {
    const int row = f & (TQ - 1), quad = f / TQ;
    const int64_t q = q0 + row;
    const float4 v = q < N ? X4[q * nq4 + quad] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* o = sA + (size_t)(quad * 4) * TQ + row;
    o[0] = v.x; o[TQ] = v.y; o[2 * TQ] = v.z; o[3 * TQ] = v.w;
  }

  // Staging of the candidate tiles: step g of the flattened (candidate tile, dim chunk) sequence moves
  // 128 points x 16 dims = 512 float4, two per thread (point row_l, float4 columns quad0 and quad0 + 2 of the
  // chunk).  The source pointer advances by a constant per step; consecutive lanes take consecutive points, so
  // the transposing LDS writes are conflict-free.
  const int64_t G = (ct1 - ct0) * nch;
  const int row_l = tid & (KNN_TC - 1), quad0 = tid >> 7;
  const float4* pn = X4 + (ct0 * KNN_TC + row_l) * nq4 + quad0;             // chunk of the NEXT load
  const int64_t tile_step = (int64_t)KNN_TC * nq4 - (int64_t)(nch - 1) * (KNN_DK / 4);
  float* const st0 = sB + (size_t)(quad0 * 4) * KNN_TC + row_l;             // LDS destination inside a buffer
  auto load_chunk = [&](int c, int nvalid, float4 (&v)[2]) {
    const bool rowok = row_l < nvalid;
    const int q4 = c * (KNN_DK / 4) + quad0;
    v[0] = (rowok && q4 < nq4) ? pn[0] : make_float4(0.f, 0.f, 0.f, 0.f);
    v[1] = (rowok && q4 + 2 < nq4) ? pn[2] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_chunk = [&](int buf, const float4 (&v)[2]) {
    float* o = st0 + (size_t)buf * KNN_DK * KNN_TC;
    o[0] = v[0].x; o[KNN_TC] = v[0].y; o[2 * KNN_TC] = v[0].z; o[3 * KNN_TC] = v[0].w;
    o += 8 * KNN_TC;
    o[0] = v[1].x; o[KNN_TC] = v[1].y; o[2 * KNN_TC] = v[1].z; o[3 * KNN_TC] = v[1].w;
  };
  auto tile_valid = [&](int64_t t) { return N - t * KNN_TC < KNN_TC ? (int)(N - t * KNN_TC) : KNN_TC; };

  knn_f2 acc[RQ][4];
#pragma unroll
  for (int r = 0; r < RQ; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[r][s] = knn_f2{0.0f, 0.0f};

  float tau[RQ];                // current k-th best distance of this thread's query rows
#pragma unroll
  for (int r = 0; r < RQ; ++r) tau[r] = INFINITY;

  float4 pre[2];
  if (G > 0) { load_chunk(0, tile_valid(ct0), pre); store_chunk(0, pre); }
  __syncthreads();

  // (ct, c): candidate tile and dim chunk of step g; (nct, nc): those of step g + 1 (no divisions in the loop)
  int64_t ct = ct0, nct = ct0;
  int c = 0, nc = 0;
  for (int64_t g = 0; g < G; ++g) {
    ct = nct; c = nc;
    if (++nc == nch) { nc = 0; ++nct; pn += tile_step; } else { pn += KNN_DK / 4; }
    if (g + 1 < G) load_chunk(nc, tile_valid(nct), pre);
    const float* const pa = sA + (size_t)(c * KNN_DK) * TQ + ty * 4;
    const float* const pb = sB + (size_t)(g & 1) * KNN_DK * KNN_TC + tx * 4;
    const int nd = d - c * KNN_DK < KNN_DK ? d - c * KNN_DK : KNN_DK;
    {
      // Two dims per round (the point rows are zero padded to a multiple of 4 dims, and a zero dim adds exactly 0 to
      // every metric's accumulator, so an odd d costs one padded dim).  The operands of the next dim are read from
      // LDS while the current one is accumulated.  One rolled loop for full and short chunks alike: the
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
      // the tile's 8 x 8 distances of this thread against the current k-th best of their queries
      const int64_t j0 = ct * KNN_TC;
      // Only the data set's last candidate tile can hold fewer than TC points; it gets its own copy of the
      // code below (RAG), so that all the other tiles do not pay for the masking.
      auto epilogue = [&](auto rag_tag) {
        constexpr bool RAG = decltype(rag_tag)::value;
        const int nvalid = (int)(N - j0);                      // RAG only: candidates at columns >= nvalid do not exist
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
            tau[r] = knn_row_insert<KL>(key_addr + (uint32_t)(row * KL * 8), kk, dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6],
                                        dv[7], tau[r], live, (uint32_t)j0, tid);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[r][s] = knn_f2{0.0f, 0.0f};
        }
      };
      if (j0 + KNN_TC > N) epilogue(std::true_type{});
      else epilogue(std::false_type{});
    }
    if (g + 1 < G) store_chunk((int)((g + 1) & 1), pre);
    __syncthreads();
  }

  // partial lists of this candidate slice
  for (int e = tid; e < TQ * kk; e += KNN_THREADS) {
    const int row = e / kk, t = e % kk;
    const int64_t q = q0 + row;
    if (q < q_end) part[((q - q_begin) * S + sp) * kk + t] = sKey[row * KL + t];
  }
}

// k best of the S partial lists of a query (each ascending) -> 1-based ids / distances, column-major.
__global__ __launch_bounds__(256) void k_knn_merge(const u64* __restrict__ part, int64_t n_q, int S, int kk, int metric,
                                                   int32_t* __restrict__ idx, float* __restrict__ dist, int64_t ld_out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_q) return;
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
    idx[(int64_t)t * ld_out + q] = id;
    if (dist) dist[(int64_t)t * ld_out + q] = dv;
  }
}

// Query rows per thread: 4 (64-query tiles) by default.  Measured at 100 k x 50, 31 nearest, manhattan: 64-query tiles
// 27.7 ms, 128-query tiles 33.1 ms — the smaller tile halves the candidate splits' insertions and the per-tile
// epilogue weighs less; LDS operand traffic per VALU instruction is higher but the LDS pipe has the room.
// GFICF_KNN_RQ=8 selects the 128-query tile for lists of at most 32 entries (tuning knob).
inline int knn_rq(int k) {
  static const int forced = getenv("GFICF_KNN_RQ") ? atoi(getenv("GFICF_KNN_RQ")) : 0;
  return (k <= 32 && forced == 8) ? 8 : 4;
}

int knn_split(const gficf_ctx* ctx, int64_t n_q, int64_t N, int k) {
  const int64_t n_qt = gficf_ceil_div(n_q > 0 ? n_q : 1, 16 * knn_rq(k)), n_ct = gficf_ceil_div(N > 0 ? N : 1, KNN_TC);
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

int knn_check(int64_t N, int d, int k, int metric) {
  if (N < 0 || d < 0 || k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  if (d > KNN_MAX_D) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "d = %d exceeds %d dimensions", d, KNN_MAX_D);
  if (k > GFICF_KNN_MAX_K) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d exceeds GFICF_KNN_MAX_K = %d", k, GFICF_KNN_MAX_K);
  if (k > N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "k = %d neighbours asked of N = %lld points", k, (long long)N);
  if (metric != GFICF_KNN_MANHATTAN && metric != GFICF_KNN_EUCLIDEAN && metric != GFICF_KNN_COSINE)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "unknown metric %d", metric);
  return GFICF_OK;
}

template <int METRIC, int KL, int RQ>
int knn_launch(gficf_ctx* ctx, const float* X, int64_t N, int d, int kk, int64_t qb, int64_t qe, int S, u64* part) {
  const int dpad = knn_dpad(d);
  constexpr int TQ = 16 * RQ;
  const size_t lds = (size_t)dpad * TQ * 4 + 2 * KNN_DK * KNN_TC * 4 + (size_t)TQ * KL * 8;
  static bool attr_set[64] = {};
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_knn_tiles<METRIC, KL, RQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[ctx->device & 63] = true;
  }
  const int64_t blocks = gficf_ceil_div(qe - qb, TQ) * S;
  hipLaunchKernelGGL((k_knn_tiles<METRIC, KL, RQ>), dim3((unsigned)blocks), dim3(KNN_THREADS), lds, ctx->stream, X, N, d, dpad, kk, qb, qe, S, part);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int METRIC>
int knn_launch_k(gficf_ctx* ctx, const float* X, int64_t N, int d, int kk, int64_t qb, int64_t qe, int S, u64* part) {
  if (kk <= 32) return knn_rq(kk) == 8 ? knn_launch<METRIC, 32, 8>(ctx, X, N, d, kk, qb, qe, S, part) : knn_launch<METRIC, 32, 4>(ctx, X, N, d, kk, qb, qe, S, part);
  if (kk <= 64) return knn_launch<METRIC, 64, 4>(ctx, X, N, d, kk, qb, qe, S, part);
  return knn_launch<METRIC, 128, 4>(ctx, X, N, d, kk, qb, qe, S, part);
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
  if (!ctx || n_queries <= 0 || N <= 0 || k <= 0) return 16;
  return (size_t)n_queries * (size_t)knn_split(ctx, n_queries, N, k) * (size_t)k * sizeof(u64) + 16;
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
  const int S = knn_split(ctx, n_q, N, k);
  u64* part = (u64*)d_ws;
  switch (metric) {
    case GFICF_KNN_MANHATTAN: rc = knn_launch_k<GFICF_KNN_MANHATTAN>(ctx, d_points, N, d, k, q_begin, q_end, S, part); break;
    case GFICF_KNN_EUCLIDEAN: rc = knn_launch_k<GFICF_KNN_EUCLIDEAN>(ctx, d_points, N, d, k, q_begin, q_end, S, part); break;
    default: rc = knn_launch_k<GFICF_KNN_COSINE>(ctx, d_points, N, d, k, q_begin, q_end, S, part); break;
  }
  if (rc) return rc;
  hipLaunchKernelGGL(k_knn_merge, dim3((unsigned)gficf_ceil_div(n_q, 256)), dim3(256), 0, ctx->stream, part, n_q, S, k, metric, d_idx, d_dist, ld_out);
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
