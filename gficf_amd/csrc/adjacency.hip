// adjacency.hip — the Jaccard edge list as the symmetric weighted adjacency matrix (second half of "next" row N1).
//
// What clustcells() builds from the kept edges before community detection,
//   g <- igraph::graph.data.frame(relations, directed = FALSE)                       (reference R/clustCells.R:69)
//   igraph::as_adjacency_matrix(g, attr = "weight", sparse = T)                       (reference R/clustCells.R:80,86)
// i.e. A[i,j] = A[j,i] = sum of the weights of all edges between i and j (an i -> j and a j -> i edge are two
// edges of the undirected multigraph, so a mutual pair carries 2w), as a sparse matrix with sorted indices — the
// input of RunModularityClustering (src/RModularityOptimizer.cpp).  igraph is third-party (unpinned); the
// restatement is "A = W + W^T over the directed edge list W, duplicates summed, a self edge counted once".
// Vertices are the cells 1..N in cell order (igraph orders vertices by first appearance in the edge list, which is
// the same whenever every cell keeps at least one edge).
//
// Device path (round 6, second form).  Row r of A = the edges r -> * (half W) and the edges * -> r (half W^T).  The edge kernel emits
// W grouped by source cell, so half W needs no ordering at all: the run of every source is found where it lies.  Only the transposed
// half is ordered, and only by its row: ONE stable radix sort of E 8-byte elements (destination << 32 | edge number; b = bits of N
// significant: 17 at 100 k cells, two passes of 9 and 8 bits) — through round 5 the build sorted 2 E 64-bit keys row << 32 | col with
// rocPRIM over seven passes, in this round's first form over five.  The passes are this file's own (k_rs_hist / k_rs_scatter: no library
// call is left in the build).  A wave then merges the two halves of a row: ranks its <= 128 entries by (column, emission position) in
// LDS, sums the runs of one column in that order (the order the full stable sort gave) and writes the row; longer rows take a wave of
// their own (<= 512 entries) or a workgroup.
//   1. k_adj_edges    validate, element = destination << 32 | e, (source, weight) packed for the gather of step 4, run of every source
//   2. radix passes   E elements over b bits, stable: within one destination the edges stay in emission order
//   3. k_adj_bounds   first sorted position of every destination (binary search), k_adj_caps + scan: room of every row, long rows listed
//   4. k_adj_rows(_mid, _big)  the merge above, rows written compacted at the start of their room
//   5. scan of the final lengths -> indptr, k_adj_compact: rows to their final place
// An edge list that is NOT grouped by source (any caller-made list is allowed) takes one more sort of the same kind, by source.
// Round 6's first attempt without any sort (row buckets filled by atomics) was slower: a device-scope atomic per edge for the
// transposed half — profiles/r06_adjacency_row_buckets.txt.
#include <atomic>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr uint32_t ADJ_NONE = 0xFFFFFFFFu;
constexpr int ADJ_WAVE_ROW = 128;            // entries a wave orders by ranking (two a lane)
constexpr int ADJ_MID_ROW = 512;             // entries a wave of its own orders by ranking (eight a lane)
constexpr int ADJ_WG_ROW = 4096;             // entries a workgroup orders in LDS

struct AdjPk { uint32_t src; uint32_t pad; double w; };      // what the transposed half gathers per entry: one 16 B read

__device__ inline bool adj_edge(const double* __restrict__ from, const double* __restrict__ to, int64_t e, int64_t N, int64_t* i, int64_t* j) {
  const double fi = from[e], fj = to[e];
  if (!(fi >= 1.0 && fi <= (double)N && fj >= 1.0 && fj <= (double)N && fi == trunc(fi) && fj == trunc(fj))) return false;
  *i = (int64_t)fi - 1; *j = (int64_t)fj - 1;
  return true;
}

// One thread per edge slot.  tkey = destination (ADJ_NONE: slot unused, bad ids, or a self edge — counted once, in half W), tval = e.
// The run of every source: its head writes rbeg, its tail rend; a source with two heads means the list is not grouped (*not_grouped).
__global__ __launch_bounds__(256) void k_adj_edges(const double* __restrict__ from, const double* __restrict__ to, const double* __restrict__ w,
                                                   int64_t cap, const int64_t* __restrict__ n_edges_p, int64_t N, u64* __restrict__ tkv,
                                                   AdjPk* __restrict__ pk, int32_t* __restrict__ rbeg,
                                                   int32_t* __restrict__ rend, int32_t* __restrict__ nruns, uint32_t* __restrict__ not_grouped,
                                                   int promised, uint32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cap) return;
  const int64_t n_edges = n_edges_p ? (*n_edges_p < cap ? *n_edges_p : cap) : cap;
  uint32_t kt = ADJ_NONE;
  if (e < n_edges) {
    int64_t i, j;
    if (adj_edge(from, to, e, N, &i, &j)) {
      if (i != j) kt = (uint32_t)j;
      AdjPk q; q.src = (uint32_t)i; q.pad = 0u; q.w = w[e];
      pk[e] = q;
    } else {
      atomicOr(status, GFICF_ST_BAD_ID);
    }
    const double f = from[e];                               // (the runs count every edge whose source is a cell, whatever its destination)
    if (f >= 1.0 && f <= (double)N && f == trunc(f)) {
      const int64_t s = (int64_t)f - 1;
      if (e > 0 && from[e - 1] > f) not_grouped[1] = 1u;    // sources not ascending
      if (e == 0 || from[e - 1] != f) {
        rbeg[s] = (int32_t)e;
        if (atomicAdd(&nruns[s], 1) > 0) {
          *not_grouped = 1u;
          if (promised) atomicOr(status, GFICF_ST_NOT_GROUPED);
        }
      }
      if (e == n_edges - 1 || from[e + 1] != f) rend[s] = (int32_t)(e + 1);
    }
  }
  tkv[e] = ((u64)kt << 32) | (u64)e;                       // key above, edge number below: one 8 B element through the sort
}

// the list is not grouped: key = source for the second sort
__global__ __launch_bounds__(256) void k_adj_source_keys(const double* __restrict__ from, const double* __restrict__ to, int64_t cap,
                                                         const int64_t* __restrict__ n_edges_p, int64_t N, u64* __restrict__ skv) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= cap) return;
  const int64_t n_edges = n_edges_p ? (*n_edges_p < cap ? *n_edges_p : cap) : cap;
  int64_t i, j;
  skv[e] = ((u64)((e < n_edges && adj_edge(from, to, e, N, &i, &j)) ? (uint32_t)i : ADJ_NONE) << 32) | (u64)e;
}

#define ADJ_WAVE_SYNC()                                                                                                                   \
  do {                                                                                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// ---- the sort: stable partition passes over (key << 32 | edge number), least significant digit first (round 6: rocPRIM's onesweep took
// three 8-bit passes for the 17 bits of 100 k cells; digits of up to 10 bits make it two).  A pass = per-workgroup digit counts
// (k_rs_hist) -> one scan of the [digit][workgroup] matrix -> k_rs_scatter.  Stable by construction: a workgroup owns RS_TILE consecutive
// elements, its wave w the w-th quarter, a wave walks its quarter 64 consecutive elements at a time; an element's place = the scanned
// count of its (digit, workgroup) + the earlier waves' elements of that digit + the wave's own earlier ones + the lower lanes' in its round.
constexpr int RS_TILE = 4096;
constexpr int RS_MAX_BITS = 10;
constexpr int RS_MAX_WGS = 768;           // workgroups of a pass (each walks ceil(tiles / RS_MAX_WGS) tiles): bounds the count matrix, and is what is resident at
                                          // once (52 KB of LDS a workgroup: three a CU) — a grid beyond that runs as one and a fraction rounds

__device__ inline unsigned rs_digit(u64 el, int shift, unsigned mask) { return (unsigned)(el >> (32 + shift)) & mask; }

__global__ __launch_bounds__(256) void k_rs_hist(const u64* __restrict__ in, int64_t M, int shift, int bits, int64_t tiles_per_wg,
                                                 int64_t* __restrict__ hist) {
  __shared__ unsigned cnt[1 << RS_MAX_BITS];
  const unsigned nb = 1u << bits, mask = nb - 1u;
  for (unsigned d = threadIdx.x; d < nb; d += 256) cnt[d] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * tiles_per_wg * RS_TILE;
  int64_t end = base + tiles_per_wg * RS_TILE;
  if (end > M) end = M;
  for (int64_t i0 = base + threadIdx.x; i0 < end; i0 += 256 * 8) {      // eight loads in flight
    u64 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int64_t i = i0 + 256 * j; v[j] = in[i < end ? i : end - 1]; }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + 256 * j < end) atomicAdd(&cnt[rs_digit(v[j], shift, mask)], 1u);
  }
  __syncthreads();
  for (unsigned d = threadIdx.x; d < nb; d += 256) hist[(int64_t)d * gridDim.x + blockIdx.x] = (int64_t)cnt[d];
}

// LAST: the pass writes the key and the edge number apart (what the row kernels read); otherwise the element as it is.
// A workgroup walks its tiles in order and carries every digit's next place in the output along (in the registers of the digit's
// thread).  A tile is put in order in LDS first and written out from there: consecutive threads then write consecutive places of one
// digit's run (a wave's store touches ~8 runs instead of ~50 scattered places).
template <bool LAST>
__global__ __launch_bounds__(256) void k_rs_scatter(const u64* __restrict__ in, int64_t M, int shift, int bits, int64_t tiles_per_wg,
                                                    const int64_t* __restrict__ hist, u64* __restrict__ out, uint32_t* __restrict__ okey,
                                                    uint32_t* __restrict__ oval) {
  __shared__ unsigned cnt[4][1 << RS_MAX_BITS];                // counts of (wave, digit), then the wave's next place inside the tile
  __shared__ unsigned delta[1 << RS_MAX_BITS];                 // a digit's first place inside the tile, then (its place in the output) - that
  __shared__ u64 stage[RS_TILE];
  const unsigned nb = 1u << bits, mask = nb - 1u;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned gpos[(1 << RS_MAX_BITS) / 256];                     // next output place of digit threadIdx.x + 256 j
#pragma unroll
  for (unsigned j = 0; j < (1u << RS_MAX_BITS) / 256u; ++j) {
    const unsigned d = threadIdx.x + 256u * j;
    gpos[j] = d < nb ? (unsigned)hist[(int64_t)d * gridDim.x + blockIdx.x] : 0u;
  }
  const int64_t first_tile = (int64_t)blockIdx.x * tiles_per_wg;
  for (int64_t tile = first_tile; tile < first_tile + tiles_per_wg && tile * RS_TILE < M; ++tile) {
    for (unsigned d = threadIdx.x; d < 4u * (1u << RS_MAX_BITS); d += 256) (&cnt[0][0])[d] = 0u;
    __syncthreads();
    const int64_t tbase = tile * RS_TILE, wbase = tbase + (int64_t)wave * (RS_TILE / 4);
    u64 el[RS_TILE / 256];
#pragma unroll
    for (int r = 0; r < RS_TILE / 256; ++r) {                  // (all the loads first, no branch around them: sixteen in flight)
      const int64_t idx = wbase + r * 64 + lane;
      el[r] = in[idx < M ? idx : M - 1];
    }
#pragma unroll
    for (int r = 0; r < RS_TILE / 256; ++r)
      if (wbase + r * 64 + lane < M) atomicAdd(&cnt[wave][rs_digit(el[r], shift, mask)], 1u);
    __syncthreads();
    if (wave == 0) {                                           // exclusive scan of the tile's digit counts: (nb + 63) / 64 digits a lane
      const unsigned per = (nb + 63u) / 64u;
      unsigned loc[(1 << RS_MAX_BITS) / 64];
      unsigned sum = 0;
#pragma unroll
      for (unsigned j = 0; j < (1u << RS_MAX_BITS) / 64u; ++j) {
        const unsigned d = lane * per + j;
        loc[j] = sum;
        if (j < per && d < nb) sum += cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
      }
      unsigned inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned v = __shfl_up(inc, off); if (lane >= off) inc += v; }
      const unsigned base = inc - sum;
#pragma unroll
      for (unsigned j = 0; j < (1u << RS_MAX_BITS) / 64u; ++j) {
        const unsigned d = lane * per + j;
        if (j < per && d < nb) delta[d] = base + loc[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < (1u << RS_MAX_BITS) / 256u; ++j) {     // counts -> first places inside the tile, wave by wave
      const unsigned d = threadIdx.x + 256u * j;
      if (d < nb) {
        const unsigned first = delta[d];
        unsigned run = first;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned c = cnt[w][d]; cnt[w][d] = run; run += c; }
        delta[d] = gpos[j] - first;
        gpos[j] += run - first;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_TILE / 256; ++r) {
      const int64_t idx = wbase + r * 64 + lane;
      const bool valid = idx < M;
      const unsigned d = rs_digit(el[r], shift, mask);
      u64 peers = __ballot(valid);                             // the lanes of this round holding the same digit
      for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const u64 m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const unsigned below = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
      const unsigned at = valid ? cnt[wave][d] + below : 0u;
      ADJ_WAVE_SYNC();
      if (valid && below == 0u) cnt[wave][d] += (unsigned)__popcll(peers);    // the lowest lane of the group moves the wave's place on
      if (valid) stage[at] = el[r];
      ADJ_WAVE_SYNC();
    }
    __syncthreads();
    const int tile_n = (int)(M - tbase < RS_TILE ? M - tbase : RS_TILE);
    for (int i = threadIdx.x; i < tile_n; i += 256) {
      const u64 e = stage[i];
      const unsigned to = (unsigned)i + delta[rs_digit(e, shift, mask)];
      if (LAST) { okey[to] = (uint32_t)(e >> 32); oval[to] = (uint32_t)e; }
      else out[to] = e;
    }
    __syncthreads();
  }
}

// L[r] = first position of the sorted keys holding a key >= r, r = 0 .. N (unused slots sort last: L[N] = number of real keys)
__global__ __launch_bounds__(256) void k_adj_bounds(const uint32_t* __restrict__ keys, int64_t M, int64_t N, int32_t* __restrict__ L) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > N) return;
  int64_t lo = 0, hi = M;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < r) lo = mid + 1;
    else hi = mid;
  }
  L[r] = (int32_t)lo;
}

// room of row r = its entries before equal columns are summed; rows too long for a wave of k_adj_rows are listed here (one device atomic
// per workgroup and list: 10 % of the rows of a kNN graph of real data at k = 50 are, and an atomic per row on one address costs more than the build)
__global__ __launch_bounds__(256) void k_adj_caps(int64_t N, const int32_t* __restrict__ rbeg, const int32_t* __restrict__ rend,
                                                  const int32_t* __restrict__ tL, const uint32_t* __restrict__ broken,
                                                  int64_t* __restrict__ start, int64_t* __restrict__ len2, int32_t* __restrict__ mid,
                                                  int32_t* __restrict__ big, unsigned* __restrict__ n_listed) {
  __shared__ unsigned s_n[2], s_base[2];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 2) s_n[threadIdx.x] = 0u;
  __syncthreads();
  int64_t c = 0;
  if (r < N && !(broken && *broken)) {                     // (a list promised grouped that is not: its runs mean nothing — empty rows, the status says why)
    const int nw = rend[r] - rbeg[r];
    c = (nw > 0 ? nw : 0) + (tL[r + 1] - tL[r]);
  }
  if (r <= N) start[r] = c;
  if (r == N) len2[N] = 0;                                 // (the final lengths are scanned in place: N + 1 values)
  const int cls = c > ADJ_MID_ROW ? 1 : c > ADJ_WAVE_ROW ? 0 : -1;
  unsigned my = 0;
  if (cls >= 0) my = atomicAdd(&s_n[cls], 1u);
  __syncthreads();
  if (threadIdx.x < 2 && s_n[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&n_listed[threadIdx.x], s_n[threadIdx.x]);
  __syncthreads();
  if (cls >= 0) (cls ? big : mid)[s_base[cls] + my] = (int32_t)r;
}

struct AdjIn {
  const double* to; const double* w; const AdjPk* pk;
  const int32_t* rbeg; const int32_t* rend;       // half W: positions [rbeg, rend) of row r — edge numbers, or positions in `se` when the list was sorted by source
  const uint32_t* se;                             // NULL: the list is grouped by source
  const int32_t* tL; const uint32_t* te;          // half W^T: positions [tL[r], tL[r + 1]) of the edge numbers ordered by destination
  const uint32_t* unsorted;                       // device flag: some source is smaller than the one before it (half W^T then is not ordered by column)
};

__device__ inline u64 adj_key(uint32_t col, uint32_t pos) { return ((u64)col << 32) | pos; }

// entry t of a row whose half W holds nw entries starting at rb and whose half W^T starts at tb: key = (column, emission position), weight
__device__ inline void adj_load(const AdjIn& I, int rb, int nw, int tb, int64_t t, u64* key, double* wv) {
  if (t < nw) {
    const uint32_t e = I.se ? I.se[rb + t] : (uint32_t)(rb + t);
    *key = adj_key((uint32_t)((int64_t)I.to[e] - 1), 2u * e);
    *wv = I.w[e];
  } else {
    const uint32_t e = I.te[tb + (t - nw)];
    const AdjPk q = I.pk[e];
    *key = adj_key(q.src, 2u * e + 1u);
    *wv = q.w;
  }
}


// lower bound in a sorted run of distinct keys
__device__ inline int adj_lower_bound(const u64* a, int n, u64 key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// One wave, one row of at most 64 * EPL entries (EPL a lane).  Every lane ranks its entries against the whole row read from LDS
// (broadcast reads; the keys are distinct: the position is), writes them at their rank, then the heads of the runs of one column add
// their run up in order and the row goes to the start of its room, compacted.  Returns the row's final length.
template <int EPL, bool MERGE>
__device__ inline int adj_wave_row(const AdjIn& I, int64_t row, int64_t lo, int n, int lane, u64* key, u64* skey, double* sw,
                                   int32_t* __restrict__ bcol, double* __restrict__ bw) {
  const int rb = I.rbeg[row], tb = I.tL[row];
  const int nw = n - (I.tL[row + 1] - tb);
  const bool tsorted = MERGE && I.se == nullptr && *I.unsorted == 0u;
  u64 k[EPL];
  double w[EPL];
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    k[s] = ~0ull; w[s] = 0.0;
    if (s * 64 + lane < n) adj_load(I, rb, nw, tb, s * 64 + lane, &k[s], &w[s]);
    key[s * 64 + lane] = k[s];
  }
  ADJ_WAVE_SYNC();
  int r[EPL];
#pragma unroll
  for (int s = 0; s < EPL; ++s) r[s] = 0;
  if (tsorted) {
    // half W^T arrived ordered (ascending sources): an entry's rank = its rank inside its own half + the entries of the other half below it.
    // Against half W (at most k entries, unordered): counted; against half W^T: its index there, or one binary search.
    for (int t = 0; t < nw; ++t) {
      const u64 kt = key[t];
#pragma unroll
      for (int s = 0; s < EPL; ++s) r[s] += kt < k[s] ? 1 : 0;
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
      const int t = s * 64 + lane;
      if (t < nw) r[s] += adj_lower_bound(key + nw, n - nw, k[s]);
      else r[s] += t - nw;
    }
  } else {
    for (int t = 0; t < n; ++t) {
      const u64 kt = key[t];
#pragma unroll
      for (int s = 0; s < EPL; ++s) r[s] += kt < k[s] ? 1 : 0;
    }
  }
#pragma unroll
  for (int s = 0; s < EPL; ++s)
    if (s * 64 + lane < n) { skey[r[s]] = k[s]; sw[r[s]] = w[s]; }
  ADJ_WAVE_SYNC();
  // heads of the runs of one column, their output positions (ranks among the heads), their sums
  int out[EPL];
  int total = 0;
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int e = s * 64 + lane;
    const bool head = e < n && (e == 0 || (skey[e] >> 32) != (skey[e - 1] >> 32));
    const u64 m = __ballot(head);
    out[s] = head ? total + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    total += __popcll(m);
  }
#pragma unroll
  for (int s = 0; s < EPL; ++s) {
    const int e = s * 64 + lane;
    if (out[s] < 0) continue;
    const u64 c = skey[e] >> 32;
    double sum = sw[e];
    for (int t = e + 1; t < n && (skey[t] >> 32) == c; ++t) sum += sw[t];
    bcol[lo + out[s]] = (int32_t)c; bw[lo + out[s]] = sum;
  }
  ADJ_WAVE_SYNC();                                            // (the next row's loads overwrite key / skey / sw)
  return total;
}

// One wave per row of at most ADJ_WAVE_ROW entries; longer rows: up to ADJ_MID_ROW k_adj_rows_mid, beyond the workgroup kernel.
// len2[row] = the row's final length.
__global__ __launch_bounds__(256) void k_adj_rows(const AdjIn I, int64_t N, const int64_t* __restrict__ start, int32_t* __restrict__ bcol,
                                                  double* __restrict__ bw, int64_t* __restrict__ len2) {
  __shared__ u64 s_key[4][ADJ_WAVE_ROW];
  __shared__ u64 s_skey[4][ADJ_WAVE_ROW];
  __shared__ double s_sw[4][ADJ_WAVE_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
    const int64_t lo = start[row];
    const int64_t n64 = start[row + 1] - lo;
    if (n64 > ADJ_WAVE_ROW) continue;                        // listed by k_adj_caps for k_adj_rows_mid / k_adj_rows_big
    const int n = (int)n64;
    const int total = n > 0 ? adj_wave_row<ADJ_WAVE_ROW / 64, true>(I, row, lo, n, lane, s_key[wave], s_skey[wave], s_sw[wave], bcol, bw) : 0;
    if (lane == 0) len2[row] = total;
  }
}

// One wave (a workgroup of its own: 12 KB of LDS) per listed row of ADJ_WAVE_ROW + 1 .. ADJ_MID_ROW entries — the tail of the in-degrees of
// a kNN graph of real data (10 % of the rows at k = 50, 40 % at k = 100): the same ranking, eight entries a lane.
__global__ __launch_bounds__(64) void k_adj_rows_mid(const AdjIn I, const int64_t* __restrict__ start, const int32_t* __restrict__ mid,
                                                     const unsigned* __restrict__ n_listed, int32_t* __restrict__ bcol, double* __restrict__ bw,
                                                     int64_t* __restrict__ len2) {
  __shared__ u64 s_key[ADJ_MID_ROW];
  __shared__ u64 s_skey[ADJ_MID_ROW];
  __shared__ double s_sw[ADJ_MID_ROW];
  const int lane = threadIdx.x;
  for (unsigned item = blockIdx.x; item < n_listed[0]; item += gridDim.x) {
    const int64_t row = mid[item], lo = start[row];
    const int n = (int)(start[row + 1] - lo);
    const int total = n <= 256 ? adj_wave_row<4, true>(I, row, lo, n, lane, s_key, s_skey, s_sw, bcol, bw)
                               : adj_wave_row<ADJ_MID_ROW / 64, true>(I, row, lo, n, lane, s_key, s_skey, s_sw, bcol, bw);
    if (lane == 0) len2[row] = total;
  }
}

// bitonic network over n2 (a power of two) elements held in arrays k / w
__device__ inline void adj_bitonic(u64* k, double* w, int64_t n2, int tid, int nthreads) {
  for (int64_t size = 2; size <= n2; size <<= 1)
    for (int64_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (int64_t t = tid; t < n2 / 2; t += nthreads) {
        const int64_t lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 a = k[lo], b = k[hi];
        if ((a > b) == up) { k[lo] = b; k[hi] = a; const double x = w[lo]; w[lo] = w[hi]; w[hi] = x; }
      }
      __syncthreads();
    }
}

// One workgroup per listed row (more than ADJ_WAVE_ROW entries; at k = 50 a kNN graph of real data has thousands: in-degrees have a tail).
// Sources in ascending order (what the edge build writes unless the cells were renumbered): the transposed half arrives ordered by
// column already — the stable sort kept the emission order —, so only half W (at most k entries) is ordered, by a bitonic network in LDS,
// and the two halves are MERGED: every entry finds its place by one binary search in the other half.  Otherwise, or beyond what LDS
// holds that way: the whole row through the bitonic network (LDS up to ADJ_WG_ROW entries, global scratch gk / gw beyond — a row of that
// length is a hub of a degenerate graph: correct, not fast).
__global__ __launch_bounds__(256) void k_adj_rows_big(const AdjIn I, const int64_t* __restrict__ start, const int32_t* __restrict__ big,
                                                      const unsigned* __restrict__ n_big, int32_t* __restrict__ bcol, double* __restrict__ bw,
                                                      int64_t* __restrict__ len2, u64* __restrict__ gk, double* __restrict__ gw) {
  extern __shared__ unsigned char s_raw[];
  u64* sk = (u64*)s_raw;
  double* sw = (double*)(sk + ADJ_WG_ROW);
  __shared__ int s_total;
  __shared__ int s_cnt[256];
  const int tid = threadIdx.x;
  const bool merge_ok = I.se == nullptr && *I.unsorted == 0u;
  for (unsigned item = blockIdx.x; item < *n_big; item += gridDim.x) {
    const int64_t row = big[item], lo = start[row], n = start[row + 1] - lo;
    const int rb = I.rbeg[row], tb = I.tL[row];
    const int nw = (int)(n - (I.tL[row + 1] - tb));
    u64* k;
    double* w;
    int64_t nw2 = 1;
    while (nw2 < nw) nw2 <<= 1;
    if (merge_ok && nw2 + (n - nw) + n <= ADJ_WG_ROW) {
      const int nt = (int)(n - nw);
      u64* const tk = sk + nw2; double* const tw = sw + nw2;       // LDS: [half W, padded to nw2][half W^T][the merged row]
      for (int t = tid; t < (int)nw2; t += 256) {
        u64 kk = ~0ull; double ww = 0.0;
        if (t < nw) adj_load(I, rb, nw, tb, t, &kk, &ww);
        sk[t] = kk; sw[t] = ww;
      }
      for (int t = tid; t < nt; t += 256) adj_load(I, rb, nw, tb, nw + t, &tk[t], &tw[t]);
      __syncthreads();
      adj_bitonic(sk, sw, nw2, tid, 256);
      k = tk + nt; w = tw + nt;
      for (int t = tid; t < nt; t += 256) { const int at = t + adj_lower_bound(sk, nw, tk[t]); k[at] = tk[t]; w[at] = tw[t]; }
      for (int t = tid; t < nw; t += 256) { const int at = t + adj_lower_bound(tk, nt, sk[t]); k[at] = sk[t]; w[at] = sw[t]; }
      __syncthreads();
    } else {
      int64_t n2 = 1;
      while (n2 < n) n2 <<= 1;
      // a row too long for LDS is ordered in global scratch: n2 < 2 n entries at twice its room's start (the scratch arrays are twice the rooms)
      k = n2 <= ADJ_WG_ROW ? sk : gk + 2 * lo;
      w = n2 <= ADJ_WG_ROW ? sw : gw + 2 * lo;
      for (int64_t t = tid; t < n2; t += 256) {
        u64 kk = ~0ull; double ww = 0.0;
        if (t < n) adj_load(I, rb, nw, tb, t, &kk, &ww);
        k[t] = kk; w[t] = ww;
      }
      __syncthreads();
      adj_bitonic(k, w, n2, tid, 256);
    }
    if (tid == 0) s_total = 0;
    __syncthreads();
    // the heads of the runs of one column: every thread takes a contiguous slice, the slices' bases by a serial pass over 256 counts
    const int64_t per = (n + 255) / 256, t0 = (int64_t)tid * per, t1 = t0 + per < n ? t0 + per : n;
    int mine = 0;
    for (int64_t t = t0; t < t1; ++t) mine += (t == 0 || (k[t] >> 32) != (k[t - 1] >> 32)) ? 1 : 0;
    s_cnt[tid] = mine;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int c = s_cnt[t]; s_cnt[t] = run; run += c; } s_total = run; }
    __syncthreads();
    int at = s_cnt[tid];
    for (int64_t t = t0; t < t1; ++t) {
      if (!(t == 0 || (k[t] >> 32) != (k[t - 1] >> 32))) continue;
      const u64 c = k[t] >> 32;
      double sum = w[t];
      for (int64_t q = t + 1; q < n && (k[q] >> 32) == c; ++q) sum += w[q];
      bcol[lo + at] = (int32_t)c; bw[lo + at] = sum;
      ++at;
    }
    if (tid == 0) len2[row] = s_total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_adj_compact(int64_t N, const int64_t* __restrict__ start, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ bcol, const double* __restrict__ bw,
                                                     int32_t* __restrict__ indices, double* __restrict__ x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
    const int64_t src = start[row], dst = indptr[row], n = indptr[row + 1] - dst;
    for (int64_t t = lane; t < n; t += 64) { indices[dst + t] = bcol[src + t]; x[dst + t] = bw[src + t]; }
  }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

inline int id_bits(int64_t N) {                        // 2^b > N: an id in [0, N) never has all of its b bits set, so ADJ_NONE sorts behind every real key
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) <= N) ++b;
  return b;
}

inline int rs_passes(int b) { return (b + RS_MAX_BITS - 1) / RS_MAX_BITS; }
inline int rs_bits(int b) { const int p = rs_passes(b); return (b + p - 1) / p; }             // digit width: the passes share the bits evenly

struct AdjWs {
  u64 *kv0, *kv1;                                       // cap each: the elements of the sort, ping and pong
  int64_t* hist;                                        // [digit][workgroup] counts of a pass
  uint32_t *tkey, *te, *skey, *se;                      // cap each
  AdjPk* pk;                                            // cap
  int32_t *rbeg, *rend, *nruns;                         // N
  int32_t *tL, *sL;                                     // N + 2
  uint32_t* flag;                                       // not_grouped, sources not ascending, rows listed as mid, as big
  int64_t* start;                                       // N + 1: start of every row's room
  int32_t* bcol; double* bw;                            // 2 cap each: the rooms
  int32_t *mid, *big;                                   // N each: the listed rows
  u64* gk; double* gw;                                  // 4 cap each: scratch of rows too long for LDS
};

size_t adj_carve(AdjWs* w, void* base, int64_t N, int64_t cap) {
  size_t off = 0;
  const size_t n = (size_t)(N > 0 ? N : 1), c = (size_t)(cap > 0 ? cap : 1), m = 2 * c;
  auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return base ? (char*)base + o : (char*)nullptr; };
  AdjWs d;
  d.kv0 = (u64*)take(c * 8); d.kv1 = (u64*)take(c * 8);
  d.hist = (int64_t*)take(((size_t)1 << RS_MAX_BITS) * (size_t)RS_MAX_WGS * sizeof(int64_t));
  d.tkey = (uint32_t*)take(c * 4); d.te = (uint32_t*)take(c * 4);
  d.skey = (uint32_t*)take(c * 4); d.se = (uint32_t*)take(c * 4);
  d.pk = (AdjPk*)take(c * sizeof(AdjPk));
  d.rbeg = (int32_t*)take((3 * n + 4) * 4); d.rend = d.rbeg + n; d.nruns = d.rend + n; d.flag = (uint32_t*)(d.nruns + n);   // one memset
  d.tL = (int32_t*)take((n + 2) * 4); d.sL = (int32_t*)take((n + 2) * 4);
  d.start = (int64_t*)take((n + 1) * sizeof(int64_t));
  d.bcol = (int32_t*)take(m * sizeof(int32_t));
  d.bw = (double*)take(m * sizeof(double));
  d.mid = (int32_t*)take(n * sizeof(int32_t));
  d.big = (int32_t*)take(n * sizeof(int32_t));
  d.gk = (u64*)take(2 * m * sizeof(u64));
  d.gw = (double*)take(2 * m * sizeof(double));
  if (w) *w = d;
  return off + 256;
}

// the elements of kv0 ordered by their key's low b bits (stable) -> okey / oval; kv0 and kv1 are both overwritten
int adj_sort(gficf_ctx* ctx, const AdjWs& w, int64_t M, int b, uint32_t* okey, uint32_t* oval) {
  const int P = rs_passes(b), bp = rs_bits(b);
  const int64_t ntiles = gficf_ceil_div(M, RS_TILE);
  const int64_t tpw = gficf_ceil_div(ntiles, RS_MAX_WGS);                  // tiles a workgroup walks
  const unsigned G = (unsigned)gficf_ceil_div(ntiles, tpw);
  u64 *in = w.kv0, *out = w.kv1;
  for (int p = 0; p < P; ++p) {
    const int shift = p * bp, bits = b - shift < bp ? b - shift : bp;
    hipLaunchKernelGGL(k_rs_hist, dim3(G), dim3(256), 0, ctx->stream, (const u64*)in, M, shift, bits, tpw, w.hist);
    GFICF_HIP_CHECK(hipGetLastError());
    const int rc = gficf_exclusive_scan_i64(ctx, w.hist, ((int64_t)1 << bits) * (int64_t)G);
    if (rc) return rc;
    if (p == P - 1)
      hipLaunchKernelGGL(k_rs_scatter<true>, dim3(G), dim3(256), 0, ctx->stream, (const u64*)in, M, shift, bits, tpw, (const int64_t*)w.hist,
                         (u64*)nullptr, okey, oval);
    else
      hipLaunchKernelGGL(k_rs_scatter<false>, dim3(G), dim3(256), 0, ctx->stream, (const u64*)in, M, shift, bits, tpw, (const int64_t*)w.hist,
                         out, (uint32_t*)nullptr, (uint32_t*)nullptr);
    GFICF_HIP_CHECK(hipGetLastError());
    u64* t = in; in = out; out = t;
  }
  return GFICF_OK;
}

}  // namespace

extern "C" {

size_t gficf_adjacency_workspace_bytes(int64_t N, int64_t edge_capacity) {
  if (N <= 0 || edge_capacity <= 0) return 256;
  return adj_carve(nullptr, nullptr, N, edge_capacity);
}

int gficf_adjacency_device(gficf_ctx* ctx, int64_t N, int64_t edge_capacity, const int64_t* d_n_edges, const double* d_from,
                           const double* d_to, const double* d_weight, int grouped_by_source, void* d_ws, size_t ws_bytes,
                           int64_t* d_indptr, int32_t* d_indices, double* d_x) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0 || edge_capacity < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  if (edge_capacity > 0x3FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "more than 2^30 edges");
  if (!d_indptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (N == 0 || edge_capacity == 0) {
    GFICF_HIP_CHECK(hipMemsetAsync(d_indptr, 0, sizeof(int64_t) * (size_t)(N + 1), ctx->stream));
    return GFICF_OK;
  }
  if (!d_from || !d_to || !d_weight || !d_ws || !d_indices || !d_x) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ws_bytes < gficf_adjacency_workspace_bytes(N, edge_capacity)) GFICF_FAIL(GFICF_ERR_CAPACITY, "adjacency workspace too small");
  static std::atomic<bool> attr_set[64];
  if (!attr_set[ctx->device & 63]) {
    GFICF_HIP_CHECK(hipFuncSetAttribute((const void*)k_adj_rows_big, hipFuncAttributeMaxDynamicSharedMemorySize, ADJ_WG_ROW * 16));
    attr_set[ctx->device & 63] = true;
  }
  AdjWs w;
  adj_carve(&w, d_ws, N, edge_capacity);
  hipStream_t st = ctx->stream;
  const int b = id_bits(N);
  const unsigned ge = (unsigned)gficf_ceil_div(edge_capacity, 256), gn = (unsigned)gficf_ceil_div(N + 1, 256);
  const unsigned gr = (unsigned)(gficf_ceil_div(N, 4) < 4096 ? gficf_ceil_div(N, 4) : 4096);
  GFICF_HIP_CHECK(hipMemsetAsync(w.rbeg, 0, sizeof(int32_t) * (3 * (size_t)N + 4), st));     // runs of the sources + the three flags behind them
  hipLaunchKernelGGL(k_adj_edges, dim3(ge), dim3(256), 0, st, d_from, d_to, d_weight, edge_capacity, d_n_edges, N, w.kv0, w.pk, w.rbeg,
                     w.rend, w.nruns, w.flag, grouped_by_source != 0 ? 1 : 0, ctx->d_status);
  GFICF_HIP_CHECK(hipGetLastError());
  int rc = adj_sort(ctx, w, edge_capacity, b, w.tkey, w.te);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adj_bounds, dim3(gn), dim3(256), 0, st, (const uint32_t*)w.tkey, edge_capacity, N, w.tL);
  AdjIn in;
  in.to = d_to; in.w = d_weight; in.pk = w.pk; in.rbeg = w.rbeg; in.rend = w.rend; in.se = nullptr; in.tL = w.tL; in.te = w.te; in.unsorted = w.flag + 1;
  bool grouped = grouped_by_source != 0;
  if (!grouped) {                                   // the caller does not know: ask the pass that just looked at every edge
    uint32_t ng = 0;
    GFICF_HIP_CHECK(hipMemcpyAsync(&ng, w.flag, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GFICF_HIP_CHECK(hipStreamSynchronize(st));
    grouped = ng == 0;
  }
  if (!grouped) {
    hipLaunchKernelGGL(k_adj_source_keys, dim3(ge), dim3(256), 0, st, d_from, d_to, edge_capacity, d_n_edges, N, w.kv0);
    GFICF_HIP_CHECK(hipGetLastError());
    rc = adj_sort(ctx, w, edge_capacity, b, w.skey, w.se);
    if (rc) return rc;
    hipLaunchKernelGGL(k_adj_bounds, dim3(gn), dim3(256), 0, st, (const uint32_t*)w.skey, edge_capacity, N, w.sL);
    in.rbeg = w.sL; in.rend = w.sL + 1; in.se = w.se;
  }
  hipLaunchKernelGGL(k_adj_caps, dim3(gn), dim3(256), 0, st, N, in.rbeg, in.rend, (const int32_t*)w.tL, in.se ? (const uint32_t*)nullptr : (const uint32_t*)w.flag,
                     w.start, d_indptr, w.mid, w.big, w.flag + 2);
  GFICF_HIP_CHECK(hipGetLastError());
  rc = gficf_exclusive_scan_i64(ctx, w.start, N + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adj_rows, dim3(gr), dim3(256), 0, st, in, N, (const int64_t*)w.start, w.bcol, w.bw, d_indptr);
  hipLaunchKernelGGL(k_adj_rows_mid, dim3(3328), dim3(64), 0, st, in, (const int64_t*)w.start, (const int32_t*)w.mid, (const unsigned*)(w.flag + 2),
                     w.bcol, w.bw, d_indptr);
  hipLaunchKernelGGL(k_adj_rows_big, dim3(1024), dim3(256), ADJ_WG_ROW * 16, st, in, (const int64_t*)w.start, (const int32_t*)w.big,
                     (const unsigned*)(w.flag + 3), w.bcol, w.bw, d_indptr, w.gk, w.gw);
  rc = gficf_exclusive_scan_i64(ctx, d_indptr, N + 1);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adj_compact, dim3(gr), dim3(256), 0, st, N, (const int64_t*)w.start, (const int64_t*)d_indptr, (const int32_t*)w.bcol,
                     (const double*)w.bw, d_indices, d_x);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------- host form (R glue)
struct gficf_adj_plan {
  int64_t N = 0, nnz = 0;
  int64_t* d_indptr = nullptr;
  int32_t* d_indices = nullptr;
  double* d_x = nullptr;
};

void gficf_adj_plan_free(gficf_ctx* ctx) {        // the buffers are pieces of pool slot 6: nothing to release
  delete ctx->adj_plan;
  ctx->adj_plan = nullptr;
}

extern "C" {

int gficf_adjacency_host_plan(gficf_ctx* ctx, int64_t N, int64_t n_edges, const double* from, const double* to,
                              const double* weight, int64_t* nnz) {
  GFICF_CTX_ENTER(ctx);
  if (N < 0 || n_edges < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative size");
  if (!nnz) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "nnz is NULL");
  if (n_edges > 0 && (!from || !to || !weight)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  gficf_adj_plan_free(ctx);
  gficf_adj_plan* p = new gficf_adj_plan();
  ctx->adj_plan = p;
  p->N = N;
  *nnz = 0;
  const size_t eb = sizeof(double) * (size_t)(n_edges > 0 ? n_edges : 1), cap = (size_t)(n_edges > 0 ? 2 * n_edges : 1);
  const size_t wsb = gficf_adjacency_workspace_bytes(N, n_edges);
  void *d_e = nullptr, *d_ws = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, 3 * eb, &d_e);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, wsb, &d_ws);
  gficf_arena ar;
  const size_t o_ip = ar.take(sizeof(int64_t) * (size_t)(N + 1)), o_ii = ar.take(sizeof(int32_t) * cap), o_xx = ar.take(sizeof(double) * cap);
  if (e == hipSuccess) e = ar.bind(ctx, 6);
  p->d_indptr = ar.at<int64_t>(o_ip); p->d_indices = ar.at<int32_t>(o_ii); p->d_x = ar.at<double>(o_xx);
  double* d_from = (double*)d_e;
  double* d_to = d_from + (n_edges > 0 ? n_edges : 1);
  double* d_w = d_to + (n_edges > 0 ? n_edges : 1);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_from, from, eb, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_to, to, eb, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n_edges > 0) e = hipMemcpyAsync(d_w, weight, eb, hipMemcpyHostToDevice, ctx->stream);
  int rc = GFICF_OK;
  int64_t total = 0;
  if (e == hipSuccess) {
    rc = gficf_adjacency_device(ctx, N, n_edges, nullptr, d_from, d_to, d_w, 0, d_ws, wsb, p->d_indptr, p->d_indices, p->d_x);
    if (!rc) e = hipMemcpyAsync(&total, p->d_indptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) { gficf_set_error("HIP failure in gficf_adjacency_host_plan: %s", hipGetErrorString(e)); rc = GFICF_ERR_HIP; }
  if (rc) { gficf_adj_plan_free(ctx); return rc; }
  p->nnz = total;
  *nnz = total;
  return GFICF_OK;
}

int gficf_adjacency_host_finish(gficf_ctx* ctx, void* indptr, int indptr_is_i64, int32_t* indices, double* x) {
  GFICF_CTX_ENTER(ctx);
  gficf_adj_plan* p = ctx->adj_plan;
  if (!p) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_adjacency_host_finish without a plan");
  if (!indptr || (p->nnz > 0 && (!indices || !x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  std::vector<int64_t> ip((size_t)p->N + 1);
  if (p->nnz > 0 && indices && x) {               // (fresh result vectors of the caller: huge pages + parallel first touch)
    gficf_prefault(indices, sizeof(int32_t) * (size_t)p->nnz);
    gficf_prefault(x, sizeof(double) * (size_t)p->nnz);
  }
  hipError_t e = hipMemcpyAsync(ip.data(), p->d_indptr, sizeof(int64_t) * ip.size(), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && p->nnz > 0) e = hipMemcpyAsync(indices, p->d_indices, sizeof(int32_t) * (size_t)p->nnz, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && p->nnz > 0) e = hipMemcpyAsync(x, p->d_x, sizeof(double) * (size_t)p->nnz, hipMemcpyDeviceToHost, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) {
    if (indptr_is_i64) std::memcpy(indptr, ip.data(), sizeof(int64_t) * ip.size());
    else for (size_t c = 0; c < ip.size(); ++c) ((int32_t*)indptr)[c] = (int32_t)ip[c];
  }
  gficf_adj_plan_free(ctx);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_adjacency_host_finish: %s", hipGetErrorString(e));
  return GFICF_OK;
}

}  // extern "C"
