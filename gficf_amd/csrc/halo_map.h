// halo_map.h — global id -> local id of a rank's sub-problem (csrc/halo.hip builds the per-owner bitmap and its ranks; the ingest
// of jaccard.hip applies the map on the fly in its fused form), and the owner-side "serve" step that shares a launch with it.
#pragma once
#include "common.h"

// The plan's workspace (gficf_jaccard_halo_workspace_bytes): OWNER-ALIGNED — owner r's rows are bits [0, rpr) of the wpo words
// that start at word r * wpo (wpo = words per owner, a multiple of 4) — so that one workgroup ranks one owner's bits on its own
// and a rank is a rank WITHIN the owner (= the request slot): no global scan, no owner_start table.
//   bitmap : P * wpo words, all zero between steps (the ranking kernel clears what it has read: no memset launch per step)
//   winfo  : P * wpo x {word, set bits of the owner in front of the word}: what the lookups read (one 8 B load)
__host__ __device__ inline int64_t gficf_halo_wpo(int64_t rpr) { return (((rpr + 31) / 32) + 3) & ~(int64_t)3; }

constexpr int GFICF_HALO_MAX_PEERS = 16;       // owners of the peer form (the kernel argument carries their pointers)

struct gficf_halo_map {
  const uint2* winfo;            // P * wpo x {bitmap word, rank of its first bit inside the owner}
  const int32_t* req_out;        // P * cap requested ids (0 = empty slot)
  const int32_t* rows_in;        // P * cap x k reply slots (raw global ids)
  int32_t* l2g;                  // out: global id of every local row
  int64_t n_local, N_total, b, rpr;
  int cap;
  uint32_t wpo;
  // rows [row_begin, row_end) of the sub-problem are ingested by this launch (own cells: [0, n_local); halo slots: behind them)
  int64_t row_begin, row_end;
  // the owner-side serve step riding in the same launch (the LAST serve_blocks workgroups): copies the rows asked of this rank
  const int32_t* req_in;
  int64_t n_req;
  int32_t* rows_out;
  int serve_blocks;
  int skip_empty;                // tiles of halo slots nobody asked for are not written (the launch that ingests ONLY the slots)
  // peer form (gficf_multi_jaccard_halo_device: one process, every device maps the others' memory): nothing is exchanged — the row of a
  // requested id is read where it lies, in its owner's block of global ids, by the launch that ingests the slots (peer_n owners with
  // equal-pitch blocks: owner o holds rows [o * rpr, ...), column j of its row i at peer_idx[o][j * peer_ld[o] + i]; 0: the rows come from rows_in)
  int peer_n;
  const int32_t* peer_idx[GFICF_HALO_MAX_PEERS];
  int64_t peer_ld[GFICF_HALO_MAX_PEERS];
};

// local id of a global id (1-based both); 0: not part of this rank's sub-problem; -1: not an id at all.  32-bit arithmetic on the
// common path (an id inside the block: one subtraction and one unsigned compare) — written with int64 compares the map cost
// 7 us per 3 M ids, as much as the ingest it is fused into.  N_total <= 2^31 - 1.
// (the outside-the-block path is kept out of line: inlined eight times into the fused ingest it took the kernel from 103 to 181
// vector registers and halved its occupancy)
__device__ __noinline__ static int32_t gficf_halo_local_outside(uint32_t id, uint32_t N_total, uint32_t n_local, uint32_t rpr, int cap,
                                                                const uint2* __restrict__ winfo, uint32_t wpo) {
  const uint32_t bit = id - 1u;
  if (bit >= N_total) return -1;                                       // 0, negative or beyond N_total: the ingest reports it
  const uint32_t owner = bit / rpr, local = bit - owner * rpr;
  const uint2 wi = winfo[(size_t)owner * wpo + (local >> 5)];
  const uint32_t m = 1u << (local & 31u);
  if ((wi.x & m) == 0u) return 0;
  const int pos = (int)wi.y + __popc(wi.x & (m - 1u));
  return pos < cap ? (int32_t)(n_local + owner * (uint32_t)cap + (uint32_t)pos + 1u) : 0;
}

__device__ inline int32_t gficf_halo_local(int64_t id64, int64_t N_total, int64_t b, int64_t n_local, int64_t rpr, int cap,
                                           const uint2* __restrict__ winfo, uint32_t wpo) {
  const uint32_t id = (uint32_t)id64;                                  // (callers hand in int32 ids: a negative one wraps above N_total)
  const uint32_t rel = id - 1u - (uint32_t)b;
  if (rel < (uint32_t)n_local) return (int32_t)(rel + 1u);             // inside the block
  return gficf_halo_local_outside(id, (uint32_t)N_total, (uint32_t)n_local, (uint32_t)rpr, cap, winfo, wpo);
}

// The rows asked of this rank: req_in holds n_req ids (0 = empty slot: nothing is written, the requester reads a slot's row only
// where it asked for one), all inside this rank's block (b, b + n_local].  One thread per (slot, group of 8 slots of the row):
// its 8 loads are issued together, then its 8 stores (a first version gave a thread the whole row, one dependent load -> store
// after the other: 8 us for 190 rows of k = 30).  `first` / `stride` in threads; the work items are n_req * ceil(k / 8).
__host__ __device__ inline int64_t gficf_halo_serve_items(int64_t n_req, int k) { return n_req * (int64_t)((k + 7) / 8); }

__device__ inline void gficf_halo_serve_rows(const int32_t* __restrict__ idx, int64_t n_local, int k, int64_t ld, int64_t b,
                                             const int32_t* __restrict__ req_in, int64_t n_req, int32_t* __restrict__ rows_out,
                                             uint32_t* __restrict__ status, int64_t first, int64_t stride) {
  const int jg = (k + 7) / 8;
  const int64_t items = n_req * (int64_t)jg;
  for (int64_t t = first; t < items; t += stride) {
    const int64_t q = t / jg;
    const int j0 = (int)(t - q * jg) * 8;
    const int64_t id = req_in[q];
    if (id == 0) continue;
    const int64_t row = id - 1 - b;
    const bool ok = row >= 0 && row < n_local;
    if (!ok && j0 == 0) atomicOr(status, GFICF_ST_BAD_ID);    // a request for a row this rank does not own: the ranks disagree on the blocks
    int32_t v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (ok && j0 + c < k) ? idx[(int64_t)(j0 + c) * ld + row] : 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (j0 + c < k) rows_out[q * k + j0 + c] = v[c];
  }
}
