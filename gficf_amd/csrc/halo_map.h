// halo_map.h — global id -> local id of a rank's sub-problem (csrc/halo.hip builds the bitmap and the ranks; the ingest of
// jaccard.hip applies the map on the fly in its fused form).
#pragma once
#include "common.h"

struct gficf_halo_map {
  const uint32_t* bitmap;        // bit (id - 1): the block names id outside itself
  const int32_t* word_rank;      // set bits in the words before a word
  const int32_t* owner_start;    // rank of the first bit of every owner
  const int32_t* req_out;        // P * cap requested ids (0 = empty slot)
  const int32_t* rows_in;        // P * cap x k reply slots (raw global ids)
  int32_t* l2g;                  // out: global id of every local row
  int64_t n_local, N_total, b, rpr;
  int cap;
};

// local id of a global id (1-based both); 0: not part of this rank's sub-problem; -1: not an id at all.  32-bit arithmetic on the
// common path (an id inside the block: one subtraction and one unsigned compare) — written with int64 compares the map cost
// 7 us per 3 M ids, as much as the ingest it is fused into.  N_total <= 2^31 - 1.
// (the outside-the-block path is kept out of line: inlined eight times into the fused ingest it took the kernel from 103 to 181
// vector registers and halved its occupancy)
__device__ __noinline__ static int32_t gficf_halo_local_outside(uint32_t id, uint32_t N_total, uint32_t n_local, uint32_t rpr, int cap,
                                                                const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ word_rank,
                                                                const int32_t* __restrict__ owner_start) {
  const uint32_t bit = id - 1u;
  if (bit >= N_total) return -1;                                       // 0, negative or beyond N_total: the ingest reports it
  const uint32_t w = bit >> 5, word = bitmap[w], m = 1u << (bit & 31u);
  if ((word & m) == 0u) return 0;
  const int owner = (int)(bit / rpr);
  const int pos = word_rank[w] + __popc(word & (m - 1u)) - owner_start[owner];
  return pos < cap ? (int32_t)(n_local + (uint32_t)owner * (uint32_t)cap + (uint32_t)pos + 1u) : 0;
}

__device__ inline int32_t gficf_halo_local(int64_t id64, int64_t N_total, int64_t b, int64_t n_local, int64_t rpr, int cap,
                                           const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ word_rank,
                                           const int32_t* __restrict__ owner_start) {
  const uint32_t id = (uint32_t)id64;                                  // (callers hand in int32 ids: a negative one wraps above N_total)
  const uint32_t rel = id - 1u - (uint32_t)b;
  if (rel < (uint32_t)n_local) return (int32_t)(rel + 1u);             // inside the block
  return gficf_halo_local_outside(id, (uint32_t)N_total, (uint32_t)n_local, (uint32_t)rpr, cap, bitmap, word_rank, owner_start);
}
