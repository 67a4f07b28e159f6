// jaccard_edges_general.h — k_jaccard_edges, the general hash-set edge kernel (every k <= 256 the other kernels do not take).
// Included by jaccard.hip (stands on its own: includes jaccard_shared.h), behind the edge kernels' shared helpers (JCfg, EdgeOut, probes).

// One wave per cell, cells strided over all waves of the grid.  Per cell:
//   * row i (one id per lane) is inserted into the wave's LDS hash set; keys that find both
//     slots of their bucket taken go to a small per-wave overflow list;
//   * "steps": each lane loads 16 B of a neighbour row (4 ids wide / 8 ids compact), so ROWB/16 lanes cover
//     one row and a wave-instruction gathers RPS = 1024/ROWB rows; U steps are in flight together;
//   * every lane probes the set with its ids (one ds_read_b64 per id), the per-row
//     intersection count is a DPP sum over the row's lanes;
//   * counts are permuted back to one-slot-per-lane and stored as three coalesced runs.
// The load of the next cell's own row is issued ahead of the gathers and the stores of the
// previous cell's edges behind them, so neither sits on the wait for the gathers.
// MAP: the table holds the local ids of a sharded sub-problem; the neighbour column is written through o.l2g (loaded per cell
// right after the own row is decoded, long before the edges are stored: a load at the store would put the wait for the gathers
// in front of it).
#pragma once

#include "jaccard_shared.h"

namespace {

template <int KPAD, bool BIG, bool CMP, int OUT, bool MAP = false>
__global__ __launch_bounds__(jc_threads<KPAD>) void k_jaccard_edges(
    const uint32_t* __restrict__ table, int64_t N, int k, int64_t cell_begin, int64_t cell_end, EdgeOut o) {
  using C = JCfg<KPAD, CMP>;
  using F = CFmt<KPAD>;
  static_assert(!(BIG && CMP), "compact rows hold 17-bit ids");
  using off_t = typename std::conditional<BIG, uint64_t, uint32_t>::type;
  // LDS (dynamic, laid out here so that a wave's hash set starts at a multiple of its size and a probe
  // address is (hash & mask) | wave_base):  hash sets | overflow list / slow-path rows | weight table
  extern __shared__ unsigned char smem[];
  constexpr uint32_t HBYTES = C::NB * 8;                      // bytes of one hash set
  constexpr uint32_t SETS = 1;
  constexpr uint32_t WBYTES = SETS * HBYTES;                  // bytes of one wave's set(s)
  uint32_t(*const s_rows)[2][KPAD] = reinterpret_cast<uint32_t(*)[2][KPAD]>(smem + C::WAVES * WBYTES);
  double* const s_lut = reinterpret_cast<double*>(smem + C::WAVES * WBYTES + C::WAVES * 2 * KPAD * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  // the wave's number as a scalar: everything derived from it (the cell index, row and output addresses) then lives in
  // scalar registers and is computed on the scalar unit instead of per lane
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const hbase = smem + wave * WBYTES;          // this wave's hash set(s)
  // W[u] = u / (2.0*k - u): same IEEE-754 double division as reference :51
  constexpr uint32_t DUPF_OFF = edges_dupflag_off<KPAD, CMP>();
  for (int u = tid; u <= k; u += C::WAVES * 64) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);
  for (int b = lane; b < (int)(SETS * C::NB); b += 64) reinterpret_cast<uint2*>(hbase)[b] = make_uint2(EMPTY, EMPTY);
  if (lane == 0) *reinterpret_cast<uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u) = 0u;
  __syncthreads();

  // LDS byte address of this wave's hash set (a multiple of WBYTES: dynamic LDS starts at 0 here,
  // there is no static LDS in this kernel), OR-ed with a bucket offset per probe.  Kept in a vector register (derived
  // from the vector thread id) so that mask-and-base is ONE v_and_or_b32 per probe (a scalar base would take the
  // instruction's only scalar operand slot away from the mask).
  const uint32_t wave_off = lds_address(smem) + (uint32_t)(tid >> 6) * WBYTES;
  // compact rows: the bucket mask and bit 16 as vector registers (operands of v_bitop3_b32)
  uint32_t bmask_v = (uint32_t)(C::NB - 1) << 3, bit16_v = 0x10000u;
  asm volatile("" : "+v"(bmask_v), "+v"(bit16_v));
  uint32_t* const ovlist = s_rows[wave][0];
  const char* const tbytes = reinterpret_cast<const char*>(table);
  const int grow = lane / C::LPR;                           // which of the RPS rows of a step this lane reads
  const int gl = lane % C::LPR;                             // this lane's 16 B piece of that row
  const uint32_t gcol = (uint32_t)gl * 16u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t nwaves = (int64_t)gridDim.x * C::WAVES;
  constexpr uint32_t ROWB = C::ROWB;
  constexpr int ROWW = ROWB / 4;
  // compact rows: where this lane finds the high bits of its 8 ids — byte (gl & 3) of high word gl / 4, which
  // sits in component hi_c of the piece held by lane hi_l of the row's lane group
  const int hi_abs = F::HIW + (gl >> 2);
  const int hi_l = lane - gl + (hi_abs >> 2), hi_c = hi_abs & 3;
  const bool tail = gl >= F::KC / 8;                        // the lane(s) holding the high-bit words

  // Own row of a cell: slot s -> register s / 64, lane s % 64.  The loads are issued one cell ahead and their
  // results stay untouched in registers until the next iteration decodes them (any arithmetic on them here would
  // put the wait for the load in front of the gathers).
  struct OwnRaw {
    uint32_t v[C::EPL];      // wide: the id word; compact: the 16-bit low half
    uint32_t hw[C::EPL];     // compact: the word of high bits covering the slot
    uint32_t last;           // compact: the row's last word (duplicate flag)
  };
  auto load_own = [&](int64_t row, OwnRaw& r) {
    const uint32_t* const rw = table + row * ROWW;
    if (CMP) r.last = rw[ROWW - 1];
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) {
      const int s = q * 64 + lane;
      r.v[q] = 0;
      r.hw[q] = 0;
      if (s < C::NSLOT) {
        if (!CMP) {
          r.v[q] = rw[s];
        } else {
          r.v[q] = reinterpret_cast<const uint16_t*>(rw)[s];
          r.hw[q] = (KPAD == 32) ? 0u : rw[F::HIW + (s >> 5)];      // KPAD = 32: the only high word is the last word
        }
      }
    }
  };
  // out: id | bit 31 = the row's duplicate flag; key: the form the hash set holds (wide: the id; compact: the stored,
  // pre-hashed half | bit 16 of the id)
  auto decode_own = [&](const OwnRaw& r, uint32_t (&out)[C::EPL], uint32_t (&key)[C::EPL]) {
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) {
      if (!CMP) {
        out[q] = r.v[q];
        key[q] = r.v[q] & ID_MASK;
      } else {
        const int s = q * 64 + lane;
        const uint32_t hw = (KPAD == 32) ? r.last : r.hw[q];
        const uint32_t hbit = ((hw >> (s & 31)) & 1u) << 16;
        const bool ok = s < C::NSLOT;
        key[q] = ok ? (r.v[q] | hbit) : 0u;
        out[q] = ok ? (unscramble16(r.v[q]) | hbit | (r.last & ROW_DUP_FLAG)) : 0u;
      }
    }
  };

  int64_t i = cell_begin + (int64_t)xcd_block(blockIdx.x, gridDim.x, o.xcd) * C::WAVES + wave;
  OwnRaw raw;
  raw.last = 0;
#pragma unroll
  for (int q = 0; q < C::EPL; ++q) { raw.v[q] = 0; raw.hw[q] = 0; }
  if (i < cell_end) load_own(i, raw);
  // edges of the previous cell, stored while the current cell's gathers are in flight
  bool have_prev = false;
  int64_t prev_i = 0;
  uint32_t prev_a[C::EPL];
  int prev_u[C::EPL];

  auto store_prev = [&]() {
    const int64_t pb = (prev_i - cell_begin) * (int64_t)k;
#pragma unroll
    for (int qq = 0; qq < C::EPL; ++qq) {
      const int slot = qq * 64 + lane;
      if (slot < C::NSLOT && slot < k) store_edge<OUT>(o, pb + slot, prev_i, prev_a[qq], prev_u[qq], s_lut);
    }
    have_prev = false;
  };

  // ids of a gathered piece (bv) -> id[]; returns the word that may carry the row's duplicate flag
  auto piece_ids = [&](const uint4& bv, uint32_t (&id)[C::IPL]) -> uint32_t {
    if (!CMP) {
      id[0] = bv.x & ID_MASK;                // only a row's first id can carry the duplicate flag
      id[1] = bv.y;
      id[2] = bv.z;
      id[3] = bv.w;
      return bv.x;
    }
    // the high-bit word of this lane's ids, from the lane that holds it
    uint32_t hw;
    if (KPAD == 32) {
      hw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bv.w, 0xFF, 0xf, 0xf, false);   // quad_perm [3,3,3,3]
    } else {
      hw = 0;
      if (F::NW >= 4) { const uint32_t v = (uint32_t)__shfl((int)bv.x, hi_l); hw = hi_c == 0 ? v : hw; }
      if (F::NW >= 4) { const uint32_t v = (uint32_t)__shfl((int)bv.y, hi_l); hw = hi_c == 1 ? v : hw; }
      { const uint32_t v = (uint32_t)__shfl((int)bv.z, hi_l); hw = hi_c == 2 ? v : hw; }
      { const uint32_t v = (uint32_t)__shfl((int)bv.w, hi_l); hw = hi_c == 3 ? v : hw; }
    }
    uint32_t hb = (hw >> ((gl & 3) * 8)) & 0xFFu;
    uint32_t wd[4] = {bv.x, bv.y, bv.z, bv.w};
    if (tail) {                              // high-bit words are not ids
      hb &= (1u << (F::KC % 8)) - 1u;
#pragma unroll
      for (int c = (F::KC % 8) / 2; c < 4; ++c) wd[c] = 0u;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint32_t lo = (t & 1) ? (wd[t >> 1] >> 16) : (wd[t >> 1] & 0xFFFFu);
      id[t] = lo | ((hb << (16 - t)) & 0x10000u);
    }
    return (gl == C::LPR - 1) ? bv.w : 0u;   // the row's last word holds the flag
  };

  // compact rows: the piece's four words with the high-bit words zeroed (wd), and the byte of high bits of this
  // lane's 8 ids (hb); returns the word that may carry the row's duplicate flag
  auto piece_words = [&](const uint4& bv, uint32_t (&wd)[4], uint32_t& hb) -> uint32_t {
    uint32_t hw;
    if (KPAD == 32) {
      hw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bv.w, 0xFF, 0xf, 0xf, false);   // quad_perm [3,3,3,3]
    } else {
      hw = 0;
      if (F::NW >= 4) { const uint32_t v = (uint32_t)__shfl((int)bv.x, hi_l); hw = hi_c == 0 ? v : hw; }
      if (F::NW >= 4) { const uint32_t v = (uint32_t)__shfl((int)bv.y, hi_l); hw = hi_c == 1 ? v : hw; }
      { const uint32_t v = (uint32_t)__shfl((int)bv.z, hi_l); hw = hi_c == 2 ? v : hw; }
      { const uint32_t v = (uint32_t)__shfl((int)bv.w, hi_l); hw = hi_c == 3 ? v : hw; }
    }
    hb = (hw >> ((gl & 3) * 8)) & 0xFFu;
    wd[0] = bv.x; wd[1] = bv.y; wd[2] = bv.z; wd[3] = bv.w;
    if (tail) {                              // high-bit words are not ids
      hb &= (1u << (F::KC % 8)) - 1u;
#pragma unroll
      for (int c = (F::KC % 8) / 2; c < 4; ++c) wd[c] = 0u;
    }
    return (gl == C::LPR - 1) ? bv.w : 0u;   // the row's last word holds the flag
  };

  for (; i < cell_end; i += nwaves) {
    uint32_t araw[C::EPL], a[C::EPL], akey[C::EPL], asafe[C::EPL];
    decode_own(raw, araw, akey);
    uint32_t flags = 0;
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) {
      flags |= araw[q];
      a[q] = araw[q] & ID_MASK;
      // a slot without a usable id (padding, rejected id) gathers the cell's own row instead; its
      // count is discarded at the store
      asafe[q] = a[q] != 0 ? a[q] : (uint32_t)(i + 1);
    }
    uint32_t ag[C::EPL];                         // what column 2 shows for the slot
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) ag[q] = MAP ? (uint32_t)o.l2g[asafe[q] - 1] : a[q];
    bool slow = __ballot((flags & ROW_DUP_FLAG) != 0) != 0ull;
    // next cell's own row: ahead of the gathers, so that it has landed by the next iteration
    const int64_t i_next = i + nwaves;
    if (i_next < cell_end) load_own(i_next, raw);

    int myu[C::EPL];
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) myu[q] = 0;
    int myslot[C::EPL];
    int nov = 0;
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) myslot[q] = -1;
    bool prev_stored = false;

    if (!slow) {
      uint32_t dupflags = 0;
      bool inserted = false, dup_here = false, own_dup = false;
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) {        // q: which register of row i holds the slots of these steps
        for (int t0 = 0; t0 < C::SPQ && (q * 64 + t0 * C::RPS) < k; t0 += C::U) {
          uint4 bv[C::U];
          // issue the gathers of U steps (U*RPS neighbour rows) before anything else
#pragma unroll
          for (int uu = 0; uu < C::U; ++uu) {
            const uint32_t dst = (uint32_t)__shfl((int)asafe[q], (t0 + uu) * C::RPS + grow);
            const off_t off = (off_t)(dst - 1) * ROWB + gcol;
            bv[uu] = *reinterpret_cast<const uint4*>(tbytes + off);
          }
          if (!prev_stored) {
            // the previous cell's edges ride behind the gathers (younger in vmcnt order, and of a count the compiler
            // knows, so the wait for the gathers does not wait for them)
            prev_stored = true;
            if (have_prev) store_prev();
          }
          if (!inserted) {
            // row i into the hash set, under the latency of the first gathers
            inserted = true;
#pragma unroll
            for (int qi = 0; qi < C::EPL; ++qi) {
              bool over = false;
              if (a[qi] != 0) {
                // wide rows: keyed by the id through the multiplicative hash; compact rows: keyed by the stored form,
                // whose bits 3.. ARE the hash
                const uint32_t key = akey[qi];
                const uint32_t bo = (CMP ? (key & ((uint32_t)(C::NB - 1) << 3)) : bucket_off<KPAD, BIG>(key)) + (uint32_t)wave * WBYTES;   // byte offset of the bucket in smem
                uint32_t old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo), EMPTY, key);
                if (old == EMPTY) {
                  myslot[qi] = (int)bo;
                } else {
                  dup_here |= old == key;        // an id twice in the row: the later one meets the earlier in one of the two slots ...
                  old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo + 4), EMPTY, key);
                  if (old == EMPTY) myslot[qi] = (int)bo + 4;
                  else { dup_here |= old == key; over = true; }
                }
              }
              const unsigned long long om = __ballot(over);
              if (om) {
                if (over) ovlist[nov + __popcll(om & lt_mask)] = akey[qi];
                nov += __popcll(om);
              }
            }
            wave_lds_fence();
            // a list longer than the six entries that are compared is reported as what it is (bit 1: GFICF_ST_SET_OVERFLOW) — not as a
            // repeated id (through round 5 it was: a spurious GFICF_ERR_DUPLICATE_IDS on uniformly spread ids at k near 256)
            // With the duplicate scan done at the ingest (exact mode) a long list is nothing to report: a row that repeats an id carries its
            // flag and never comes here, and the probes below walk the whole list — slower, exact.  (Through round 5 such a cell fell to the
            // all-pairs path: 22 ms instead of ~2 at 5 000 x 256 on uniformly spread ids.)
            const bool over_here = nov > 6 && edge_kernel_dup_status() != nullptr;
            if (nov > 1 && nov <= 6) dup_here |= ovlist_repeats(ovlist, nov);      // ... or both overflowed (rare)
            if (dup_here | over_here) {            // reported at the kernel's end (every lane that writes ORs its own bits into what it read)
              uint32_t* const fw = reinterpret_cast<uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u);
              atomicOr(fw, (dup_here ? 1u : 0u) | (over_here ? 2u : 0u));
            }
            own_dup = __ballot(dup_here | over_here) != 0ull;
          }
          int cnt[C::U];
#pragma unroll
          for (int uu = 0; uu < C::U; ++uu) {
            uint32_t miss = 0;
            int c;
            if (!CMP) {
              uint32_t id[C::IPL];
              dupflags |= piece_ids(bv[uu], id);
              // all probes of the piece are issued before the first is compared
              uint2 h[C::IPL];
#pragma unroll
              for (int t = 0; t < C::IPL; ++t) h[t] = lds_read_b64(bucket_off<KPAD, BIG>(id[t]) | wave_off);
              // misses, counted on the vector ALU alone: min(slot0 ^ id, slot1 ^ id, 1) is 0 on a hit and 1 on a miss (a compare
              // per slot would go v_cmp -> s_or -> v_addc through the scalar unit and its wait states for every probe)
#pragma unroll
              for (int t = 0; t < C::IPL; t += 2) {
                const uint32_t m0 = min3u_one(h[t].x ^ id[t], h[t].y ^ id[t]);
                const uint32_t m1 = min3u_one(h[t + 1].x ^ id[t + 1], h[t + 1].y ^ id[t + 1]);
                miss += m0 + m1;                 // one v_add3_u32
              }
              c = C::IPL - (int)miss;
              if (nov) {                          // wave-uniform, rare: ids that overflowed the set
                for (int t = 0; t < nov; ++t) {
                  const uint32_t ov = ovlist[t];
#pragma unroll
                  for (int tt = 0; tt < C::IPL; ++tt) c += (id[tt] == ov);
                }
              }
            } else {
              uint32_t wd[4], hb;
              dupflags |= piece_words(bv[uu], wd, hb);
              c = probe_compact_piece(wd, hb, bmask_v, bit16_v, wave_off);
              if (nov) {                          // wave-uniform, rare: ids that overflowed the set (kept in their stored form)
                for (int t = 0; t < nov; ++t) {
                  const uint32_t ov = ovlist[t];
#pragma unroll
                  for (int tt = 0; tt < 8; ++tt) c += (piece_key(wd, hb, tt) == ov);
                }
              }
            }
            cnt[uu] = c;
          }
#pragma unroll
          for (int uu = 0; uu < C::U; ++uu) {
            const int rowcnt = group_sum<C::LPR>(cnt[uu]);
            // slot s = (t0+uu)*RPS + r lives in lane s of myu[q]; its count sits in lanes r*LPR..
            const int v = __shfl(rowcnt, (lane % C::RPS) * C::LPR);
            myu[q] = (lane / C::RPS == t0 + uu) ? v : myu[q];
          }
        }
      }
      // a neighbour row with duplicate ids (or the own row, found at the insert): redo this cell exactly
      slow = own_dup || __ballot((dupflags & ROW_DUP_FLAG) != 0) != 0ull;
    }
    if (!prev_stored && have_prev) store_prev();    // own row with duplicates (or k == 0): the gather loop was skipped
    // ---- clear this cell's keys from the set
#pragma unroll
    for (int q = 0; q < C::EPL; ++q)
      if (myslot[q] >= 0) *reinterpret_cast<uint32_t*>(smem + myslot[q]) = EMPTY;
    wave_lds_fence();
    if (slow) {
      slow_cell<KPAD, CMP, OUT>(table, i, k, (i - cell_begin) * (int64_t)k, s_rows[wave][0], s_rows[wave][1], lane, o.src, o.dst, o.w,
                                o.u, o.u16, o.set_mode, s_lut, o.l2g, o.src_off);
    } else {
      have_prev = true;
      prev_i = i;
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) {
        prev_a[q] = ag[q];
        prev_u[q] = a[q] != 0 ? myu[q] : 0;      // rejected id: zero row
      }
    }
  }
  if (have_prev) store_prev();
  // a row of this wave's cells named an id twice: the deferred report of the "distinct ids" mode (no flags in the table)
  wave_lds_fence();
  const uint32_t fbits = *reinterpret_cast<const uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u);
  if (fbits != 0u && lane == 0) {
    uint32_t* const st = edge_kernel_dup_status();
    if (st != nullptr) atomicOr(st, ((fbits & 1u) ? GFICF_ST_DUP_IDS : 0u) | ((fbits & 2u) ? GFICF_ST_SET_OVERFLOW : 0u));
  }
}

}  // namespace
