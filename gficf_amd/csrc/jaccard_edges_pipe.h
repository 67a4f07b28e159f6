// jaccard_edges_pipe.h — k_jaccard_edges_pipe, the software-pipelined edge kernel for k <= 32 (the north-star shape).
// Included by jaccard.hip (stands on its own: includes jaccard_shared.h), behind the edge kernels' shared helpers.

// ------------------------------------------------------------------ edge kernel, software-pipelined (k <= 32)
// The kernel above is bound by neither its arithmetic nor its LDS probes (tools/lab: taking ALL probes out leaves its time
// unchanged, 40 fewer vector instructions per cell likewise) but by the latency of a cell's row gathers, which nothing in
// the wave overlaps: a wave issues the gathers of cell i and waits for them before it can do anything else.  This variant,
// for the row sizes whose gathers all fit in registers at once (k <= 32: one batch per cell), keeps TWO cells in
// flight per wave: the gathers of cell i+1 (and the own row of cell i+2) are issued before cell i's pieces are probed, so
// the memory system always has the wave's next requests while the wave computes.  For the wait on cell i's pieces to
// leave the younger requests alone the compiler must know how many there are: every load and store between two
// waits is unconditional (indices are clamped instead of branched on, the first cell is peeled instead of guarded), and
// cells that need the exact multiset path (rows with duplicate ids) are only flagged here and redone after the loop.
// Output: a wave takes its cells four consecutive ones at a time, parks (neighbour id, count) of each in LDS and writes
// the quad's 4k edges of every array with ONE store of 16 B per lane (k = 30: 960 B = 15 whole 64 B segments) instead of
// four runs of k x 8 B that straddle segments: 17 % fewer write requests, none of them partial (memory-only model,
// tools/lab/gather_lab.hip: 39.8 -> 35.5 us at 100 k x 30).
// B16 = false (compact rows only): N < 2^16, no id has bit 16 — the bitmap words of the rows are zero and are not looked at.
// NOFLAG: the table was ingested without the duplicate scan (gficf_ctx_set_jaccard_distinct) and carries no row flags: the
// kernel does not look for them (own row, every gathered piece: ~8 of its ~200 vector instructions per cell); a repeated id
// is found at the own row's insert, as in every variant.
#pragma once

#include "jaccard_shared.h"

namespace {

template <int KPAD, bool BIG, bool CMP, int OUT, bool B16 = true, bool MAP = false, bool NOFLAG = false>
__global__ __launch_bounds__(jc_threads<KPAD>) void k_jaccard_edges_pipe(
    const uint32_t* __restrict__ table, int64_t N, int k, int64_t cell_begin, int64_t cell_end, EdgeOut o) {
  using C = JCfg<KPAD, CMP>;
  using F = CFmt<KPAD>;
  // (Extended to 32 < k <= 64 — eight gather steps per cell, edges leaving a pair of cells at a time — the kernel needs 177
  // vector registers: two waves per SIMD, 164 us against 129 us of the one-cell-at-a-time kernel at 100 k x 50.  Not kept.)
  static_assert(C::EPL == 1 && C::SPQ <= 4, "one batch of gathers per cell");
  static_assert(!(BIG && CMP), "compact rows hold 17-bit ids");
  using off_t = typename std::conditional<BIG, uint64_t, uint32_t>::type;
  constexpr int NST = C::SPQ;                                 // gather steps of a cell, all in flight together
  extern __shared__ unsigned char smem[];
  constexpr uint32_t HBYTES = C::NB * 8;
  constexpr uint32_t SETS = 1;
  constexpr uint32_t WBYTES = SETS * HBYTES;
  uint32_t(*const s_rows)[2][KPAD] = reinterpret_cast<uint32_t(*)[2][KPAD]>(smem + C::WAVES * WBYTES);
  double* const s_lut = reinterpret_cast<double*>(smem + C::WAVES * WBYTES + C::WAVES * 2 * KPAD * 4);
  constexpr uint32_t STAGE_OFF = C::WAVES * WBYTES + C::WAVES * 2 * KPAD * 4 + (GFICF_JACCARD_MAX_K + 1) * 8;   // behind the weight table
  constexpr uint32_t STAGE_WAVE = 4 * 64 * 8;                 // 4 cells x 64 lanes x {id, count}
  constexpr uint32_t DUPF_OFF = edges_dupflag_off<KPAD, CMP>();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const hbase = smem + wave * WBYTES;
  for (int u = tid; u <= k; u += C::WAVES * 64) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);   // reference :51
  for (int b = lane; b < (int)(SETS * C::NB); b += 64) reinterpret_cast<uint2*>(hbase)[b] = make_uint2(EMPTY, EMPTY);
  for (int t = lane; t < 4 * 64; t += 64) reinterpret_cast<uint2*>(smem + STAGE_OFF + (uint32_t)wave * STAGE_WAVE)[t] = make_uint2(0u, 0u);
  if (lane == 0) *reinterpret_cast<uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u) = 0u;
  __syncthreads();

  const uint32_t wave_off = lds_address(smem) + (uint32_t)(tid >> 6) * WBYTES;
  // compact rows: the bucket mask and bit 16 as vector registers (operands of v_bitop3_b32)
  uint32_t bmask_v = (uint32_t)(C::NB - 1) << 3, bit16_v = 0x10000u;
  asm volatile("" : "+v"(bmask_v), "+v"(bit16_v));
  uint32_t* const ovlist = s_rows[wave][0];
  const char* const tbytes = reinterpret_cast<const char*>(table);
  const int grow = lane / C::LPR, gl = lane % C::LPR;
  const uint32_t gcol = (uint32_t)gl * 16u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t nwaves = (int64_t)gridDim.x * C::WAVES;
  constexpr uint32_t ROWB = C::ROWB;
  constexpr int ROWW = ROWB / 4;
  const int hi_abs = F::HIW + (gl >> 2);
  const int hi_l = lane - gl + (hi_abs >> 2), hi_c = hi_abs & 3;
  const bool tail = gl >= F::KC / 8;
  const int slot_c = lane < C::NSLOT ? lane : C::NSLOT - 1;  // lanes beyond the row's slots load a valid slot and are masked at the decode
  const bool slot_ok = lane < C::NSLOT;

  // the wave's cells: quads of four consecutive cells, quad q0 + m * nwaves for m = 0, 1, ...
  const int64_t first = cell_begin + 4 * ((int64_t)xcd_block(blockIdx.x, gridDim.x, o.xcd) * C::WAVES + wave);
  if (first >= cell_end) return;                              // (after the barrier; wave-uniform)
  const int64_t last_cell = cell_end - 1;
  const int64_t quad_step = 4 * nwaves - 3;                   // from the last cell of a quad to the first of the wave's next
  // quad store: lane L holds edges 2L and 2L + 1 of the quad's 4k; (cell in quad, slot) of both, as LDS addresses
  const uint32_t stage_w = lds_address(smem) + STAGE_OFF + (uint32_t)(tid >> 6) * STAGE_WAVE;
  int qc0, qc1;
  uint32_t qra0, qra1;
  {
    const int e0 = 2 * lane, e1 = e0 + 1;
    qc0 = (e0 >= k) + (e0 >= 2 * k) + (e0 >= 3 * k);
    qc1 = (e1 >= k) + (e1 >= 2 * k) + (e1 >= 3 * k);
    int j0 = e0 - qc0 * k, j1 = e1 - qc1 * k;               // lanes past the quad's edges: clamped (their stores fall outside the descriptor)
    j0 = j0 < 63 ? j0 : 63;
    j1 = j1 < 63 ? j1 : 63;
    qra0 = stage_w + (uint32_t)(qc0 * 64 + j0) * 8u;
    qra1 = stage_w + (uint32_t)(qc1 * 64 + j1) * 8u;
  }

  struct OwnRaw { uint32_t v, hw, last; };
  // own row of a cell: loads only (unconditional), decoded one iteration later
  auto load_own = [&](int64_t row, OwnRaw& r) {
    const uint32_t* const rw = table + row * ROWW;
    if (!CMP) {
      r.v = rw[slot_c];
      r.hw = 0; r.last = 0;
    } else {
      r.last = rw[ROWW - 1];
      r.v = reinterpret_cast<const uint16_t*>(rw)[slot_c];
      r.hw = (KPAD == 32) ? 0u : rw[F::HIW + (slot_c >> 5)];
    }
  };
  // The own row in the form the hash set holds (wide: id | bit 31 = the row's duplicate flag; compact: stored, pre-hashed
  // half | bit 16 of the id | bit 31 = the flag); 0 for lanes without a slot.  true_id() gives the id itself.
  auto decode_own = [&](const OwnRaw& r) -> uint32_t {
    uint32_t x;
    if (!CMP) x = r.v;
    else if (B16) x = r.v | ((((KPAD == 32 ? r.last : r.hw) >> (lane & 31)) & 1u) << 16) | (NOFLAG ? 0u : (r.last & ROW_DUP_FLAG));
    else x = r.v | (NOFLAG ? 0u : (r.last & ROW_DUP_FLAG));
    return slot_ok ? x : 0u;
  };
  auto true_id = [&](uint32_t keyraw) -> uint32_t {
    const uint32_t x = keyraw & ID_MASK;
    return CMP ? (unscramble16(x & 0xFFFFu) | (x & 0x10000u)) : x;
  };
  auto issue_gathers = [&](uint32_t asafe, uint4 (&bv)[NST]) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const uint32_t dst = (uint32_t)__shfl((int)asafe, st * C::RPS + grow);
      const off_t off = (off_t)(dst - 1) * ROWB + gcol;
      bv[st] = *reinterpret_cast<const uint4*>(tbytes + off);
    }
  };
  auto piece_words = [&](const uint4& bv, uint32_t (&wd)[4], uint32_t& hb) -> uint32_t {
    hb = 0;
    if (B16) {
      uint32_t hw;
      if (KPAD == 32) hw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bv.w, 0xFF, 0xf, 0xf, false);   // quad_perm [3,3,3,3]
      else {
        hw = 0;
        { const uint32_t v = (uint32_t)__shfl((int)bv.z, hi_l); hw = hi_c == 2 ? v : hw; }
        { const uint32_t v = (uint32_t)__shfl((int)bv.w, hi_l); hw = hi_c == 3 ? v : hw; }
      }
      hb = (hw >> ((gl & 3) * 8)) & 0xFFu;
    }
    wd[0] = bv.x; wd[1] = bv.y; wd[2] = bv.z; wd[3] = bv.w;
    if (tail) {
      hb &= (1u << (F::KC % 8)) - 1u;
#pragma unroll
      for (int c = (F::KC % 8) / 2; c < 4; ++c) wd[c] = 0u;
    }
    return (gl == C::LPR - 1) ? bv.w : 0u;
  };

  // counts of cell `a`'s slots from its gathered pieces (fast path); returns whether the cell needs the exact path
  auto process = [&](uint32_t araw, const uint4 (&bv)[NST], int& u_out) -> bool {
    const uint32_t a = araw & ID_MASK;            // the hash set's form of the id (see decode_own)
    bool slow = NOFLAG ? false : __ballot((araw & ROW_DUP_FLAG) != 0) != 0ull;
    // row i into the hash set
    int myslot = -1, nov = 0;
    bool dup_here = false;
    {
      bool over = false;
      if (a != 0) {
        // wide rows: keyed by the id through the multiplicative hash; compact rows: keyed by the stored form, whose bits 3.. ARE the hash
        const uint32_t key = a;
        const uint32_t bo = (CMP ? (key & ((uint32_t)(C::NB - 1) << 3)) : bucket_off<KPAD, BIG>(key)) + (uint32_t)wave * WBYTES;
        uint32_t old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo), EMPTY, key);
        if (old == EMPTY) myslot = (int)bo;
        else {
          dup_here |= old == key;               // an id twice in the row (a scanned row's flag says so too)
          old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo + 4), EMPTY, key);
          if (old == EMPTY) myslot = (int)bo + 4;
          else { dup_here |= old == key; over = true; }
        }
      }
      // (two equal ids walk the same two slots: the later one meets the earlier in one of them, or both overflow)
      const unsigned long long om = __ballot(over);
      if (om) {
        if (over) ovlist[nov + __popcll(om & lt_mask)] = a;
        nov += __popcll(om);
        if (nov > 1) {                           // (one cell in 300) the same id twice among the overflowed ones?
          wave_lds_fence();
          dup_here |= ovlist_repeats(ovlist, nov);
        }
      }
    }
    wave_lds_fence();
    uint32_t dupflags = 0;
    int myu = 0;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      int c;
      if (!CMP) {
        uint32_t miss = 0;
        uint32_t id[4] = {bv[st].x & ID_MASK, bv[st].y, bv[st].z, bv[st].w};
        if (!NOFLAG) dupflags |= bv[st].x;
        uint2 h[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = lds_read_b64(bucket_off<KPAD, BIG>(id[t]) | wave_off);
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const uint32_t m0 = min3u_one(h[t].x ^ id[t], h[t].y ^ id[t]);
          const uint32_t m1 = min3u_one(h[t + 1].x ^ id[t + 1], h[t + 1].y ^ id[t + 1]);
          miss += m0 + m1;
        }
        c = 4 - (int)miss;
        if (nov) {
          for (int t = 0; t < nov; ++t) {
            const uint32_t ov = ovlist[t];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) c += (id[tt] == ov);
          }
        }
      } else {
        uint32_t wd[4], hb;
        const uint32_t fw = piece_words(bv[st], wd, hb);
        if (!NOFLAG) dupflags |= fw;
        c = probe_compact_piece<B16>(wd, hb, bmask_v, bit16_v, wave_off);
        if (nov) {                                  // wave-uniform, rare: ids that overflowed the set (kept in their stored form)
          for (int t = 0; t < nov; ++t) {
            const uint32_t ov = ovlist[t];
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) c += (piece_key(wd, hb, tt) == ov);
          }
        }
      }
      const int rowcnt = group_sum<C::LPR>(c);
      const int v = __shfl(rowcnt, (lane % C::RPS) * C::LPR);
      myu = (lane / C::RPS == st) ? v : myu;
    }
    if (dup_here) *reinterpret_cast<uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u) = 1u;   // reported at the kernel's end
    slow |= __ballot(dup_here || (!NOFLAG && (dupflags & ROW_DUP_FLAG) != 0)) != 0ull;      // wave-uniform
    if (myslot >= 0) *reinterpret_cast<uint32_t*>(smem + myslot) = EMPTY;
    wave_lds_fence();
    u_out = a != 0 ? myu : 0;                   // rejected id: zero row
    return slow;
  };

  // ---- prologue: own row and gathers of the first cell, own row of the second
  OwnRaw raw;
  load_own(first, raw);
  uint32_t araw_cur = decode_own(raw);
  uint4 bv_cur[NST];
  uint32_t id_cur = true_id(araw_cur);
  // MAP: what column 2 shows for the slot, o.l2g[id - 1] — one more unconditional load per cell, issued with the cell's
  // gathers and first looked at when the cell's edges are parked, an iteration later
  uint32_t gid_cur = MAP ? (uint32_t)o.l2g[(id_cur != 0 ? id_cur : (uint32_t)(first + 1)) - 1] : 0u;
  issue_gathers(id_cur != 0 ? id_cur : (uint32_t)(first + 1), bv_cur);
  {
    const int64_t i1 = first + 1;
    load_own(i1 < cell_end ? i1 : last_cell, raw);
  }
  bool any_slow = false;
  int64_t prev_i = first;
  uint32_t prev_a = 0;
  int prev_u = 0;

  // Edges leave a quad of cells at a time.  park_prev: (neighbour id, count) of the cell just counted into the wave's
  // staging rows (all 64 lanes write: no predicate, no branch).  store_quad: 4k edges of each array through a buffer
  // descriptor that covers exactly them — lanes past 2k fall outside its range and the hardware drops their stores, so
  // there is no lane predicate and no branch around the stores (the compiler guards a predicated block with a branch
  // that skips it when no lane is active, which would make the number of memory operations between two waits unknown
  // to it).  Non-temporal (aux = 2): written once, never re-read here.
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef GFICF_EDGE_STORE_AUX
#define GFICF_EDGE_STORE_AUX 2
#endif
  constexpr int EDGE_STORE_AUX = GFICF_EDGE_STORE_AUX;
  auto park_prev = [&](int c) {                                       // c: the cell's place in its quad
    reinterpret_cast<uint2*>(smem + STAGE_OFF + (uint32_t)wave * STAGE_WAVE)[c * 64 + lane] = make_uint2(prev_a, (uint32_t)prev_u);
  };
  // ncells < 4: the wave's last, shorter quad.  Its edges may end in the middle of a lane's pair (k odd): a raw buffer
  // access is range-checked dword by dword, so the first half of such a lane is written and the second dropped.
  auto store_quad = [&](int64_t qfirst, int ncells) {
    wave_lds_fence();
    const uint2 p0 = lds_read_b64(qra0), p1 = lds_read_b64(qra1);     // {id, count} of the lane's two edges
    const int64_t pb = (qfirst - cell_begin) * (int64_t)k;            // scalar: first entry of the quad
    const int nedges = ncells * k;
    if (OUT != OUT_U16) {
      const uint32_t c1st = (uint32_t)(qfirst + 1) + o.src_off;
      const bool pos0 = p0.y > 0, pos1 = p1.y > 0;
      const v2d vs = {pos0 ? (double)(c1st + (uint32_t)qc0) : 0.0, pos1 ? (double)(c1st + (uint32_t)qc1) : 0.0};   // reference :49
      const v2d vd = {pos0 ? (double)p0.x : 0.0, pos1 ? (double)p1.x : 0.0};                                     // reference :50
      const v2d vw = {s_lut[p0.y], s_lut[p1.y]};                                                                 // reference :51 (lut[0] = 0.0: the zero row)
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(o.src + pb, 0, nedges * 8, 0x00020000);
      const auto rd = __builtin_amdgcn_make_buffer_rsrc(o.dst + pb, 0, nedges * 8, 0x00020000);
      const auto rw = __builtin_amdgcn_make_buffer_rsrc(o.w + pb, 0, nedges * 8, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, vs), rs, lane * 16, 0, EDGE_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, vd), rd, lane * 16, 0, EDGE_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, vw), rw, lane * 16, 0, EDGE_STORE_AUX);
    }
    if (OUT == OUT_RMAT_U) {
      const auto ru = __builtin_amdgcn_make_buffer_rsrc(o.u + pb, 0, nedges * 4, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b64(v2u{p0.y, p1.y}, ru, lane * 8, 0, 2);
    }
    if (OUT == OUT_U16) {        // 2 B per edge: a dword holds a lane's pair, and the range is checked per dword — the odd last edge goes out on its own
      const auto ru = __builtin_amdgcn_make_buffer_rsrc(o.u16 + pb, 0, nedges * 2, 0x00020000);
      const auto ru_even = __builtin_amdgcn_make_buffer_rsrc(o.u16 + pb, 0, (nedges & ~1) * 2, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(p0.y | (p1.y << 16), ru_even, lane * 4, 0, 0);
      if (nedges & 1) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)p0.y, ru, lane * 4, 0, 0);   // wave-uniform condition (rewrites the even edges with the same values)
    }
  };

  // one cell (place CQ in its quad): prefetch the next one's requests into `nxt`, [park the previous cell's edges; behind
  // the fourth of a quad: store the quad,] count this one's intersections from `cur`.  The two piece buffers swap roles
  // from cell to cell (the loop is unrolled by four, an even number): copying one into the other would need the data,
  // i.e. wait for the very gathers that are meant to stay in flight.
  auto body = [&](int64_t i, const uint4 (&cur)[NST], uint4 (&nxt)[NST], auto cq_tag, auto park_tag) {
    constexpr int CQ = decltype(cq_tag)::value;
    constexpr bool PARK = decltype(park_tag)::value;
    const int64_t i1 = i + (CQ == 3 ? quad_step : 1), i2 = i1 + (CQ == 2 ? quad_step : 1);
    const bool valid1 = i1 < cell_end;
    // next cell: its own row was requested an iteration ago
    const uint32_t araw_next = valid1 ? decode_own(raw) : 0u;
    const uint32_t a1 = true_id(araw_next);
    // the own row of the cell after next FIRST: next iteration's wait for it then leaves the gathers issued behind it in flight
    load_own(i2 < cell_end ? i2 : last_cell, raw);
    __builtin_amdgcn_sched_barrier(0);                                 // (the scheduler would hoist the gathers above the own-row load)
    const uint32_t gid_next = MAP ? (uint32_t)o.l2g[(a1 != 0 ? a1 : (uint32_t)(i + 1)) - 1] : 0u;
    issue_gathers(a1 != 0 ? a1 : (uint32_t)(i + 1), nxt);              // no next cell: every lane reads row i (one line)
    __builtin_amdgcn_sched_barrier(0);
    if (PARK) park_prev((CQ + 3) & 3);
    if (PARK && CQ == 0) store_quad(i - 4 * nwaves, 4);   // the quad before this one is complete
    __builtin_amdgcn_sched_barrier(0);
    int u;
    const bool slow = process(araw_cur, cur, u);
    any_slow |= slow;
    prev_i = i;
    prev_a = MAP ? gid_cur : id_cur;
    prev_u = u;
    araw_cur = araw_next;
    id_cur = a1;
    gid_cur = gid_next;
  };

  uint4 bv_b[NST];
  using T_ = std::true_type;
  int64_t i = first;
  int cq_last = 0;                                                     // place in its quad of the last cell counted
  body(i, bv_cur, bv_b, std::integral_constant<int, 0>{}, std::false_type{});
  for (;;) {
    if (i + 1 >= cell_end) break;
    i += 1; cq_last = 1;
    body(i, bv_b, bv_cur, std::integral_constant<int, 1>{}, T_{});
    if (i + 1 >= cell_end) break;
    i += 1; cq_last = 2;
    body(i, bv_cur, bv_b, std::integral_constant<int, 2>{}, T_{});
    if (i + 1 >= cell_end) break;
    i += 1; cq_last = 3;
    body(i, bv_b, bv_cur, std::integral_constant<int, 3>{}, T_{});
    if (i + quad_step >= cell_end) break;
    i += quad_step; cq_last = 0;
    body(i, bv_cur, bv_b, std::integral_constant<int, 0>{}, T_{});
  }
  park_prev(cq_last);
  store_quad(prev_i - cq_last, cq_last + 1);
  // ---- a row of this wave's cells named an id twice: the deferred report of the "distinct ids" mode (no flags in the table)
  if (any_slow) {
    wave_lds_fence();
    if (*reinterpret_cast<const uint32_t*>(smem + DUPF_OFF + (uint32_t)wave * 4u) != 0u && lane == 0) {
      uint32_t* const st = edge_kernel_dup_status();
      if (st != nullptr) atomicOr(st, GFICF_ST_DUP_IDS);
    }
  }
  // ---- cells with duplicate ids in their own row or in a neighbour row (never the case for real kNN output): the exact
  // multiset path, after the loop; their fast-path rows written above are overwritten (same wave, program order)
  if (any_slow) {
    __builtin_amdgcn_s_waitcnt(0);
    for (int64_t n = 0;; ++n) {                                       // the wave's cells again, in the same order
      const int64_t c = first + (n >> 2) * 4 * nwaves + (n & 3);
      if (c >= cell_end) break;
      const uint32_t* const rw = table + c * ROWW;
      const uint32_t a = lane < k ? row_slot_id(rw, lane, KPAD, CMP) : 0u;
      bool f = row_dup_flag(rw, KPAD, CMP);
      if (a != 0) f |= row_dup_flag(table + (int64_t)(a - 1) * ROWW, KPAD, CMP);
      if (__ballot(f) != 0ull)
        slow_cell<KPAD, CMP, OUT>(table, c, k, (c - cell_begin) * (int64_t)k, s_rows[wave][0], s_rows[wave][1], lane, o.src, o.dst, o.w,
                                  o.u, o.u16, o.set_mode, s_lut, o.l2g, o.src_off);
    }
  }
}

}  // namespace
