// jaccard_direct.h — ONE launch for small Jaccard problems (round 5).  Included by jaccard.hip (stands on its own: includes jaccard_shared.h).
//
// At the small BASELINE shapes (config 1: 3 000 x 15, config 2: 10 000 x 30) ingest + edge kernel are 3.6 + 5.7 / 8.5 us of kernels
// and a step is bound by the two launches and the boundary between them, not by bytes.  A grid barrier between an ingest phase and
// an edge phase of one launch would cost more than the kernel boundary it replaces (4-6 us against 1.5-1.9 us:
// MI355X_MICROARCH.md, rows barrier-xcd / boundary), so this kernel has NO table at all: the whole column-major input is a few
// hundred KB and stays in L2, and a wave builds its cell's edges straight from it:
//   lane j < k reads the cell's own id of slot j (one strided load per lane; validated here: every row is the own row of exactly
//   one cell);  a gather instruction covers R = 64 / KPAD neighbours: lane (r, s) reads slot s of neighbour r's row, again straight
//   from the input (one 4 B load per lane, all of a cell's k / R gathers in flight together);  membership is k compares per
//   gathered id against the own ids taken lane by lane into scalar registers (v_readlane; XOR + running minimum, VALU only) — at
//   k <= 32 cheaper than building a hash set for one or two cells per wave;  the hits of a neighbour's lanes are one ballot + a
//   population count;  the weight is the reference's division itself (src/rcpp_parallel_jaccard_coeff.cpp:51), one per lane.
// Per id gathered that is 16 x the L2 requests of the table path (4 B instead of a 64 B row piece), which is why it only pays
// while launches, not requests, bound the step: gficf_jaccard_device takes it for small k and few edges (direct_applies below).
// Rows are taken to hold DISTINCT ids (what gficf_ctx_set_jaccard_distinct promises and the host entries assume first): a row that
// names an id twice is seen by its own cell (its self-compare counts two), which raises the deferred GFICF_ST_DUP_IDS — the
// exact sequence is then re-run, as for the table path.  Without that promise the table path runs.
#pragma once

#include "jaccard_shared.h"
#include "jaccard_ingest.h"      // decode_id

// edges up to which gficf_jaccard_device builds a small problem in ONE launch, without a table (jaccard_direct.h; measured
// crossover against ingest + edge kernel: profiles/r05_direct_ab.txt)
#ifndef GFICF_JACCARD_DIRECT_DEFAULT_EDGES
#define GFICF_JACCARD_DIRECT_DEFAULT_EDGES 65536
#endif

namespace {

template <typename T, int KPAD, int OUT>
__global__ __launch_bounds__(256) void k_jaccard_direct(const T* __restrict__ idx, int64_t N, int k, int64_t ld, EdgeOut o,
                                                        uint32_t* __restrict__ status) {
  static_assert(KPAD == 16 || KPAD == 32, "k <= 32");
  constexpr int G = KPAD;                  // lanes of one neighbour row
  constexpr int R = 64 / G;                // neighbour rows per gather instruction
  constexpr int NG = KPAD / R;             // gather instructions per cell
  constexpr uint32_t NONE = 0xFFFFFFFFu;   // what a lane without an id holds (no own id equals it)
  const int lane = threadIdx.x & 63;
  const int s = lane & (G - 1), rr = lane / G;
  const double twok = 2.0 * (double)k;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t i = w0; i < N; i += nw) {
    uint32_t a = 0;
    if (lane < k) {
      bool ok;
      a = decode_id<T>(idx[(int64_t)lane * ld + i], N, ok);
      if (!ok) atomicOr(status, GFICF_ST_BAD_ID);
    }
    uint32_t b[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t nb = (uint32_t)__shfl((int)a, g * R + rr);
      b[g] = NONE;
      if (s < k && nb != 0u) {
        bool ok;
        const uint32_t v = decode_id<T>(idx[(int64_t)s * ld + (int64_t)(nb - 1u)], N, ok);      // (its own cell reports a bad id)
        if (ok) b[g] = v;
      }
    }
    // k compares per id: min over the own ids of (id XOR own id) is 0 iff the id is in the row
    uint32_t m[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) m[g] = NONE;
    const uint32_t amine = (lane < k && a != 0u) ? a : NONE;
    uint32_t self = 0;
#pragma unroll
    for (int j = 0; j < KPAD; ++j) {
      if (j < k) {                                              // (wave-uniform)
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)a, j);
        if (oj != 0u) {                                         // (an invalid id was taken out above)
          self += (amine == oj) ? 1u : 0u;
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const uint32_t x = b[g] ^ oj;
            m[g] = x < m[g] ? x : m[g];
          }
        }
      }
    }
    if (__ballot(self > 1u) != 0ull && lane == 0) atomicOr(status, GFICF_ST_DUP_IDS);   // the row names an id twice: exact re-run
    // hits of neighbour (g, r) = set bits of its G lanes; lane j takes the count of slot j
    int u = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const unsigned long long mask = __ballot(m[g] == 0u);
      const int cnt = __popcll((mask >> ((lane & (R - 1)) * G)) & (G == 32 ? 0xFFFFFFFFull : 0xFFFFull));
      if ((lane / R) == g) u = cnt;
    }
    if (lane < k) {
      const int64_t r = i * (int64_t)k + lane;
      const bool pos = u > 0;
      if (OUT != OUT_U16) {
        __builtin_nontemporal_store(pos ? (double)(uint32_t)(i + 1) : 0.0, o.src + r);                  // reference :49
        __builtin_nontemporal_store(pos ? (double)a : 0.0, o.dst + r);                                   // reference :50
        __builtin_nontemporal_store(pos ? (double)u / (twok - (double)u) : 0.0, o.w + r);                // reference :51
      }
      if (OUT == OUT_RMAT_U) __builtin_nontemporal_store(u, o.u + r);
      if (OUT == OUT_U16) o.u16[r] = (uint16_t)u;
    }
  }
}

// edges below which the one-launch form is taken (GFICF_JACCARD_DIRECT_MAX_EDGES in the environment overrides; 0 = never)
inline int64_t direct_max_edges() {
  static const int64_t v = [] {
    const char* e = getenv("GFICF_JACCARD_DIRECT_MAX_EDGES");
    return e ? (int64_t)atoll(e) : (int64_t)GFICF_JACCARD_DIRECT_DEFAULT_EDGES;
  }();
  return v;
}

// Default: k <= 16 and at most 65 536 edges — the measured crossover (profiles/r05_direct_ab.txt: 3 000 x 15 6.2 us against 8.0 for one
// call of ingest + edge kernel and 11.2 for two; 10 000 x 15 12.5 against 8.8; at k = 30 the 32-lane form loses everywhere: 10 000 x 30
// 49 us against 11.9 — sixteen gathers of 4 B per lane and 30 x 16 compares per cell against 64 B row pieces and a hash probe per id).
// A limit set on the context (gficf_ctx_set_jaccard_direct_max_edges) applies as given, for k <= 32.
inline bool direct_applies(const gficf_ctx* ctx, int64_t N, int k) {
  if (!ctx->jaccard_assume_distinct || k < 1 || N < 1 || N > 0x7FFFFFFFll) return false;
  if (ctx->jaccard_direct_max_edges >= 0) return k <= 32 && N * (int64_t)k <= ctx->jaccard_direct_max_edges;
  return k <= 16 && N * (int64_t)k <= direct_max_edges();
}

template <typename T, int KPAD>
int launch_direct_t(gficf_ctx* ctx, const T* d_idx, int64_t N, int k, int64_t ld, EdgeOut o) {
  const int64_t need = gficf_ceil_div(N, 4), cap = (int64_t)ctx->num_cus * 8;
  const unsigned grid = (unsigned)(need < cap ? need : cap);
  if (o.u16) hipLaunchKernelGGL((k_jaccard_direct<T, KPAD, OUT_U16>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, N, k, ld, o, ctx->d_status);
  else if (o.u) hipLaunchKernelGGL((k_jaccard_direct<T, KPAD, OUT_RMAT_U>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, N, k, ld, o, ctx->d_status);
  else hipLaunchKernelGGL((k_jaccard_direct<T, KPAD, OUT_RMAT>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, N, k, ld, o, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

inline int launch_direct(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t N, int k, int64_t ld, EdgeOut o) {
  if (idx_is_f64)
    return k <= 16 ? launch_direct_t<double, 16>(ctx, (const double*)d_idx, N, k, ld, o) : launch_direct_t<double, 32>(ctx, (const double*)d_idx, N, k, ld, o);
  return k <= 16 ? launch_direct_t<int32_t, 16>(ctx, (const int32_t*)d_idx, N, k, ld, o) : launch_direct_t<int32_t, 32>(ctx, (const int32_t*)d_idx, N, k, ld, o);
}

}  // namespace
