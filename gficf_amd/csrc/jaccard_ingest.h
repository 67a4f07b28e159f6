// jaccard_ingest.h — the ingest kernels of the Jaccard path (column-major ids -> table rows).  Included by jaccard.hip inside its
// anonymous namespace, behind the table row formats; split out of jaccard.hip in round 5 (same translation unit, same code).

// ------------------------------------------------------------------------------ ingest
// zero_ok: 0 stands for "no id in this slot" (rows of a sharded sub-problem in local ids, halo.hip) instead of being an error
#pragma once

#include "jaccard_shared.h"

namespace {

template <typename T>
__device__ inline uint32_t decode_id(T raw, int64_t N, bool& ok, int zero_ok = 0);
template <>
__device__ inline uint32_t decode_id<int32_t>(int32_t raw, int64_t N, bool& ok, int zero_ok) {
  ok = (raw >= 1 && (int64_t)raw <= N) || (zero_ok && raw == 0);
  return ok ? (uint32_t)raw : 0u;
}
template <>
__device__ inline uint32_t decode_id<double>(double raw, int64_t N, bool& ok, int zero_ok) {
  // reference: int k = mat(i,j) - 1  (:28) — only integer-valued ids are meaningful.
  ok = (raw >= 1.0 && raw <= (double)N && raw == trunc(raw)) || (zero_ok && raw == 0.0);
  return ok ? (uint32_t)raw : 0u;
}

constexpr int INGEST_ROWS = 64;

// Tile transpose: 64 cells x KPAD slots per step.  Reads are coalesced along cells
// (column-major input), writes are one contiguous run of the table (64 rows).
template <typename T, int KPAD, bool CMP>
__global__ __launch_bounds__(256) void k_ingest(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld,
                                                int64_t N_total, uint32_t* __restrict__ table,
                                                uint32_t* __restrict__ status, int zero_ok, int scan) {
  __shared__ uint32_t tile[INGEST_ROWS][KPAD + 1];
  __shared__ uint32_t dup[INGEST_ROWS];
  constexpr int ROWW = CMP ? CFmt<KPAD>::ROWW : KPAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row0 = (int64_t)blockIdx.x * INGEST_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * INGEST_ROWS) {
    const int64_t r = row0 + lane;
    bool bad = false;
    for (int j = wave; j < KPAD; j += 4) {
      uint32_t v = 0;
      if (j < k && r < n_rows) {
        bool ok;
        v = decode_id<T>(idx[(int64_t)j * ld + r], N_total, ok, zero_ok);
        bad |= !ok;
      }
      tile[lane][j] = v;
    }
    if (tid < INGEST_ROWS) dup[tid] = 0;
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    __syncthreads();
    // duplicate ids inside a row (multiset case): thread (row = lane, part = wave)
    bool d = false;
    for (int j = wave; scan && j < k; j += 4) {
      const uint32_t a = tile[lane][j];
      if (a != 0)
        for (int j2 = 0; j2 < j; ++j2) d |= (tile[lane][j2] == a);
    }
    if (d) dup[lane] = 1;
    __syncthreads();
    const int64_t rows_here = (n_rows - row0) < INGEST_ROWS ? (n_rows - row0) : INGEST_ROWS;
    const int n_out = (int)rows_here * ROWW;
    for (int e = tid; e < n_out; e += 256) {
      const int rr = e / ROWW, j = e % ROWW;
      uint32_t v;
      if (!CMP) {
        v = tile[rr][j];
        if (j == 0 && dup[rr]) v |= ROW_DUP_FLAG;
      } else if (j < CFmt<KPAD>::HIW) {
        v = scramble16(tile[rr][2 * j] & 0xFFFFu) | (scramble16(tile[rr][2 * j + 1] & 0xFFFFu) << 16);
      } else {
        const int j0 = (j - CFmt<KPAD>::HIW) * 32;
        v = 0;
        for (int b = 0; b < 32 && j0 + b < CFmt<KPAD>::KC; ++b) v |= ((tile[rr][j0 + b] >> 16) & 1u) << b;
        if (j == ROWW - 1 && dup[rr]) v |= ROW_DUP_FLAG;
      }
      table[row0 * ROWW + e] = v;
    }
    __syncthreads();
  }
}

// Tile variant for KPAD <= 64 (the common sizes): 64 cells per workgroup of 256 threads.  Reads are coalesced along
// cells, every (cell, slot) element is one thread's; the tile goes through LDS, then thread (cell = lane, part = wave)
// holds the cell's row in registers and checks its quarter of the id pairs for duplicates — min over the pairs of
// a XOR b, VALU only (a compare per pair would funnel through the scalar unit: v_cmp -> s_or, a dependent chain that
// cost 13 us at 100 k x 30) —, then the rows are packed and leave as contiguous 16 B-per-lane runs.  (Tried instead: every id
// inserted into a small per-row hash table in LDS with ds_cmpst, one returning atomic per element in place of a compare per
// pair of elements — 30 us against 10 at 100 k x 30: returning LDS atomics are far slower than the 186 vector instructions
// per thread of the all-pairs scan.  Round 3, at 64 slots where the scan is 1 225 pairs per row and holds the row in 181
// registers: an open-addressing table of 128 words per row, every swap of a round in flight together, 94 registers — 55 us
// against 29 at 100 k x 50; and `dup |= a == b` again, now as v_cmp_eq_u32 + s_or_b64 straight: 38 us against 29, 12.0
// against 10.7 at 100 k x 30.  The XOR + v_min_u32 form stays.  And once the scan had left the default path (SCAN = false): the
// rows packed in registers and stored straight from them, no LDS tile — every lane then writes its row's 16 B pieces at a 64 /
// 128 B stride — 7.8 us against 7.0 at 100 k x 30, 54 against 39 at 1 M x 30: the tile stays for the scan-less form too.)
template <int KPAD, int W>
__device__ inline uint32_t dup_part(const uint32_t (&r)[KPAD], int k) {
  uint32_t m = 0xFFFFFFFFu;                 // min over this part's pairs (j, j2 < j), j = W, W + 4, ...
#pragma unroll
  for (int j = W; j < KPAD; j += 4) {
    if (KPAD < 64 || j < k) {               // wave-uniform: slots past k hold no id (k = 50 in 64 slots: 1225 of the 2016 pairs; at
                                            // 32 slots the branches cost more than the few pairs they save: +0.9 us at k = 30)
#pragma unroll
      for (int j2 = 0; j2 < j; ++j2) {
        const uint32_t x = r[j] ^ r[j2];
        m = x < m ? x : m;
      }
    }
  }
  return m;
}

// HALO (int32 ids only): the rows of a sharded sub-problem (csrc/halo.hip) read straight from the block's global ids — own cells
// from idx, halo slots from the reply slots — and mapped to local ids on the fly (the unfused form writes the mapped index
// matrix first: one more kernel and 2 x 16 MB of traffic per step at 100 k cells); also writes the local -> global map.
// SCAN = false (gficf_ctx_set_jaccard_distinct): rows are taken to hold distinct ids and no flag is written; the edge kernel
// finds a repeated id when it inserts the row into its hash set and raises a deferred error.
template <typename T, int KPAD, bool CMP, bool HALO = false, bool SCAN = true, bool DUAL = false>
// (Holding the scan-less 64-slot variants to 7 waves per SIMD — so that the 1563 tiles of 100 k cells are all resident, where 79 / 93
// vector registers give 6 / 5 workgroups per CU — spills 5 / 16 registers and is no faster: 13.1 / 26.5 us against 12.8 / 20.8; the
// dual variant at 6 waves per SIMD, -DGFICF_INGEST_DUAL_WAVES=6: 9 spills, 24.5 us.)
#ifndef GFICF_INGEST_DUAL_WAVES
#define GFICF_INGEST_DUAL_WAVES 1
#endif
__global__ __launch_bounds__(256, HALO ? 4 : (DUAL && !SCAN) ? GFICF_INGEST_DUAL_WAVES : 1) void k_ingest_tile(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld,
                                                     int64_t N_total, uint32_t* __restrict__ table,
                                                     uint32_t* __restrict__ status, int zero_ok, const gficf_halo_map hm) {
  constexpr int ROWS = 64;
  constexpr int ROWW = CMP ? CFmt<KPAD>::ROWW : KPAD;
  static_assert(!DUAL || (CMP && KPAD == 64), "dual rows are compact rows of 64 slots");
  constexpr int PITCH = DUAL ? DUAL_PITCH : ROWW;            // words from one table row to the next
  __shared__ uint32_t tile[ROWS][KPAD + 1];
  __shared__ uint32_t dup[ROWS];
  // dual rows: a wave's scratch for the planar part of ONE row (a whole tile of them would cost 8 KB of LDS: six workgroups per CU
  // instead of nine, 1536 resident ones for the 1563 tiles of 100 k cells — a second round for the last 27: 26 us against 13)
  __shared__ uint32_t prow[DUAL ? 16 : 1][DUAL ? 32 : 1];     // (four rows in flight per wave)
  // dual rows: the planar build holds every row's "id >= 65536" mask as a ballot — the two bitmap words of the compact part, which
  // the writers below would otherwise gather bit by bit from 60 slots (the longest chain of the kernel, on one thread in four)
  __shared__ uint32_t bmap[DUAL ? ROWS + 3 : 1][2];
  __shared__ const int32_t* s_peer_idx[HALO ? GFICF_HALO_MAX_PEERS : 1];
  __shared__ int64_t s_peer_ld[HALO ? GFICF_HALO_MAX_PEERS : 1];
  // (the wave number as a scalar: a slot index j = wave + 4 m is then uniform and the 64-bit products j * ld stay in scalar registers —
  // as a vector value they cost two registers per load in flight, 32 of the 64-slot variants' 96, and a whole workgroup per CU)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // HALO: the launch ingests rows [row_begin, row_end) of the sub-problem (n_rows = row_end); its LAST serve_blocks workgroups do the
  // owner-side serve step instead (the rows other ranks asked of this one: independent of the ingest, one launch saved per step)
  int64_t row_first = 0;
  unsigned ingest_blocks = gridDim.x;
  if constexpr (HALO) {
    ingest_blocks = gridDim.x - (unsigned)hm.serve_blocks;
    if (blockIdx.x >= ingest_blocks) {
      gficf_halo_serve_rows(reinterpret_cast<const int32_t*>(idx), hm.n_local, k, ld, hm.b, hm.req_in, hm.n_req, hm.rows_out, status,
                            (int64_t)(blockIdx.x - ingest_blocks) * 256 + tid, (int64_t)hm.serve_blocks * 256);
      return;
    }
    row_first = hm.row_begin;
    // peer form: the owners' blocks, indexed by a lane's own owner below (a by-value array indexed per lane would go through scratch)
    if (hm.peer_n > 0) {
#pragma unroll
      for (int o = 0; o < GFICF_HALO_MAX_PEERS; ++o)
        if (tid == o) { s_peer_idx[o] = hm.peer_idx[o]; s_peer_ld[o] = hm.peer_ld[o]; }
      __syncthreads();
    }
  }
  // (peer form: the tiles are taken from the LAST one down — the few tiles of halo slots in use read their rows through a chain of
  // dependent loads, request -> owner's pointer -> the owner's block, possibly over xGMI: started first, that latency lies under the
  // own cells' tiles instead of behind them)
  const int64_t n_tiles = (n_rows - row_first + ROWS - 1) / ROWS;
  bool from_last = false;
  if constexpr (HALO) from_last = hm.peer_n > 0;
  for (int64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += ingest_blocks) {
    const int64_t row0 = row_first + (from_last ? n_tiles - 1 - tile_i : tile_i) * ROWS;
    const int64_t r = row0 + lane;
    // all loads of the thread are issued before the first is looked at
    T raw[KPAD / 4];
    if constexpr (HALO) {
      const int64_t q = r - hm.n_local;                      // halo slot of this row (own cells: negative)
      const int32_t gid = (r < n_rows && q >= 0) ? hm.req_out[q] : 0;
      // a tile of halo slots nobody asked for (most of them: the slots in use sit at the front of every owner's cap): nothing refers to
      // its rows — skipped whole (every wave of the workgroup sees the same 64 slots: the decision is workgroup-uniform)
      if (hm.skip_empty && row0 >= hm.n_local && __ballot(gid != 0) == 0ull) continue;
      if (wave == 0 && r < n_rows) hm.l2g[r] = q < 0 ? (int32_t)(hm.b + r + 1) : gid;
      if (row0 + ROWS <= hm.n_local) {                       // a tile of own cells (workgroup-uniform): the plain loads, all in flight
#pragma unroll
        for (int m = 0; m < KPAD / 4; ++m) {
          const int j = wave + 4 * m;
          raw[m] = j < k ? idx[(int64_t)j * ld + r] : (T)0;
        }
      } else {                                               // the seam tile and the halo slots (most of them empty)
#pragma unroll
        for (int m = 0; m < KPAD / 4; ++m) {
          const int j = wave + 4 * m;
          int32_t g = 0;
          if (j < k && r < n_rows) {
            if (q < 0) g = (int32_t)idx[(int64_t)j * ld + r];
            else if (gid != 0) {
              if (hm.peer_n > 0) {                             // the row where it lies: its owner's block (the plan asks owner o only for ids of o's block)
                const uint32_t o = (uint32_t)q / (uint32_t)hm.cap;
                g = s_peer_idx[o][(int64_t)j * s_peer_ld[o] + ((int64_t)gid - 1 - (int64_t)o * hm.rpr)];
              } else g = hm.rows_in[q * k + j];
            }
          }
          raw[m] = (T)g;
        }
      }
#pragma unroll
      for (int m = 0; m < KPAD / 4; ++m) {                   // global -> local (own rows: an invalid id stays invalid; halo rows: 0)
        const int j = wave + 4 * m;
        if (j < k && r < n_rows) {
          int32_t v = 0;
          if (q < 0 || gid != 0) {
            v = gficf_halo_local((int64_t)raw[m], hm.N_total, hm.b, hm.n_local, hm.rpr, hm.cap, hm.winfo, hm.wpo);
            if (q >= 0 && v < 0) v = 0;
          }
          raw[m] = (T)v;
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < KPAD / 4; ++m) {
        const int j = wave + 4 * m;
        raw[m] = (j < k && r < n_rows) ? idx[(int64_t)j * ld + r] : (T)0;
      }
    }
    if (tid < ROWS) dup[tid] = 0;
    bool bad = false;
#pragma unroll
    for (int m = 0; m < KPAD / 4; ++m) {
      const int j = wave + 4 * m;
      uint32_t v = 0;
      if (j < k && r < n_rows) {
        bool ok;
        v = decode_id<T>(raw[m], N_total, ok, zero_ok);
        bad |= !ok;
      }
      tile[lane][j] = v;
    }
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    __syncthreads();
    if constexpr (SCAN) {
      uint32_t rr[KPAD];
#pragma unroll
      for (int j = 0; j < KPAD; ++j) {
        const uint32_t v = tile[lane][j];
        rr[j] = v != 0 ? v : (0x80000000u | (uint32_t)j);       // empty slots: values no id and no other slot has
      }
      uint32_t m;
      switch (wave) {
        case 0: m = dup_part<KPAD, 0>(rr, k); break;
        case 1: m = dup_part<KPAD, 1>(rr, k); break;
        case 2: m = dup_part<KPAD, 2>(rr, k); break;
        default: m = dup_part<KPAD, 3>(rr, k); break;
      }
      if (m == 0) dup[lane] = 1;
      __syncthreads();
    }
    const int64_t rows_here = (n_rows - row0) < ROWS ? (n_rows - row0) : ROWS;
    if constexpr (DUAL) {                                   // a wave builds the planar parts of 16 rows of the tile, one after the other,
      for (int r0 = wave * 16; r0 < (int)rows_here && r0 < wave * 16 + 16; r0 += 4) {   // rows 16 w .. 16 w + 15, four at a time,
        const int n = (int)rows_here - r0 < 4 ? (int)rows_here - r0 : 4;                 // each written as one 128 B run
        const uint32_t* const ids[4] = {&tile[r0][0], &tile[r0 + 1 < ROWS ? r0 + 1 : r0][0], &tile[r0 + 2 < ROWS ? r0 + 2 : r0][0],
                                        &tile[r0 + 3 < ROWS ? r0 + 3 : r0][0]};
        uint32_t dm = 0;
        if (SCAN)
          for (int r = 0; r < n; ++r) dm |= (dup[r0 + r] != 0u ? 1u : 0u) << r;
        planar_rows_build4(ids, n, k, dm, &prow[wave * 4], lane, &bmap[r0]);
        const int r = lane >> 4, w2 = (lane & 15) * 2;                                   // 16 lanes per row, 8 B each
        if (r < n) *reinterpret_cast<uint2*>(table + (row0 + r0 + r) * PITCH + ROWW + w2) = make_uint2(prow[wave * 4 + r][w2], prow[wave * 4 + r][w2 + 1]);
        wave_lds_fence_early();
      }
      __syncthreads();                                      // (the masks are read by whichever thread writes the row's last words)
    }
    const int n_out4 = (int)rows_here * (ROWW / 4);
    for (int e = tid; e < n_out4; e += 256) {
      const int rr = e / (ROWW / 4), j0 = (e % (ROWW / 4)) * 4;
      uint4* const dst4 = reinterpret_cast<uint4*>(table + (row0 + rr) * PITCH + j0);       // (PITCH == ROWW unless the rows are dual)
      uint32_t w4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        uint32_t x;
        if (!CMP) {
          x = tile[rr][j];
          if (j == 0 && dup[rr]) x |= ROW_DUP_FLAG;
        } else if (j < CFmt<KPAD>::HIW) {
          x = scramble16(tile[rr][2 * j] & 0xFFFFu) | (scramble16(tile[rr][2 * j + 1] & 0xFFFFu) << 16);
        } else if (DUAL) {
          x = bmap[rr][j - CFmt<KPAD>::HIW];
          if (j == ROWW - 1 && dup[rr]) x |= ROW_DUP_FLAG;
        } else {
          const int b0 = (j - CFmt<KPAD>::HIW) * 32;
          x = 0;
#pragma unroll
          for (int b = 0; b < 32; ++b)
            if (b0 + b < CFmt<KPAD>::KC) x |= ((tile[rr][b0 + b] >> 16) & 1u) << b;
          if (j == ROWW - 1 && dup[rr]) x |= ROW_DUP_FLAG;
        }
        w4[c] = x;
      }
      *dst4 = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    __syncthreads();
  }
}

// Register variant for KPAD <= 64: one thread per cell.  The k loads of a thread are independent
// (all in flight at once) and each is a coalesced 256 B run per wave; duplicate detection is an
// all-pairs compare in registers; the tile goes through LDS once so that the table is written as
// contiguous 16 B-per-lane runs.
constexpr int INGEST2_ROWS = 64;

template <typename T, int KPAD, bool CMP>
__global__ __launch_bounds__(INGEST2_ROWS) void k_ingest_reg(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld,
                                                             int64_t N_total, uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ status, int zero_ok) {
  constexpr int ROWW = CMP ? CFmt<KPAD>::ROWW : KPAD;
  __shared__ uint32_t tile[INGEST2_ROWS][ROWW + 1];
  const int tid = threadIdx.x;
  for (int64_t row0 = (int64_t)blockIdx.x * INGEST2_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * INGEST2_ROWS) {
    const int64_t r = row0 + tid;
    uint32_t v[KPAD];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < KPAD; ++j) {
      v[j] = 0;
      if (j < k && r < n_rows) {
        bool ok;
        v[j] = decode_id<T>(idx[(int64_t)j * ld + r], N_total, ok, zero_ok);
        bad |= !ok;
      }
    }
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    bool dup = false;
#pragma unroll
    for (int j = 1; j < KPAD; ++j) {
      bool dj = false;
#pragma unroll
      for (int j2 = 0; j2 < j; ++j2) dj |= (v[j] == v[j2]);
      dup |= dj && v[j] != 0;
    }
    if (!CMP) {
      if (dup) v[0] |= ROW_DUP_FLAG;
#pragma unroll
      for (int j = 0; j < KPAD; ++j) tile[tid][j] = v[j];
    } else {
      using F = CFmt<KPAD>;
#pragma unroll
      for (int w = 0; w < F::HIW; ++w) tile[tid][w] = scramble16(v[2 * w] & 0xFFFFu) | (scramble16(v[2 * w + 1] & 0xFFFFu) << 16);
#pragma unroll
      for (int h = 0; h < F::NW; ++h) {
        uint32_t hw = 0;
#pragma unroll
        for (int b = 0; b < 32; ++b)
          if (h * 32 + b < F::KC) hw |= ((v[h * 32 + b] >> 16) & 1u) << b;
        if (h == F::NW - 1 && dup) hw |= ROW_DUP_FLAG;
        tile[tid][F::HIW + h] = hw;
      }
    }
    __syncthreads();
    const int64_t rows_here = (n_rows - row0) < INGEST2_ROWS ? (n_rows - row0) : INGEST2_ROWS;
    const int n_out4 = (int)rows_here * (ROWW / 4);
    uint4* const out4 = reinterpret_cast<uint4*>(table + row0 * ROWW);
    for (int e = tid; e < n_out4; e += INGEST2_ROWS) {
      const int rr = e / (ROWW / 4), jj = (e % (ROWW / 4)) * 4;
      out4[e] = make_uint4(tile[rr][jj], tile[rr][jj + 1], tile[rr][jj + 2], tile[rr][jj + 3]);
    }
    __syncthreads();
  }
}

}  // namespace
