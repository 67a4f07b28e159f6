// jaccard_edges_bits.h — k_jaccard_edges_bits, the direct-address bit-set edge kernel on dual rows (32 < k <= 55, N <= 131 070).
// Included by jaccard.hip (stands on its own: includes jaccard_shared.h), behind the edge kernels' shared helpers.

// ------------------------------------------------------------------ edge kernel on dual rows: a direct-address bit set (32 < k <= 55)
// One wave per cell, two cells in flight per wave (the pipelined kernel's scheme: cell i+1's gathers and cell i+2's own row are
// requested before cell i's pieces are probed; every load and store between two waits is unconditional, so the wait counts stay
// exact).  Row i goes into the wave's BIT SET — 2^17 bits = 16 KiB of LDS, plane 0 = ids below 2^16, plane 1 = the rest, the wave's
// region 16 KiB-aligned so that a probe address is (bits of the half) | base, one v_bitop3 —: ds_or with return (an id already
// there = the row repeats it: the deferred duplicate report).  A lane gathers 16 B = 8 halves of the PLANAR part of a neighbour
// row; all eight lie in one plane, which the lane knows from the row's header (one ds_bpermute per piece).  Per id: word address
// (shift, bitop3), ds_read_b32, shift by the id's low five bits (v_lshrrev takes them straight from the packed word), and 1,
// add: 5-6 issue slots against ~7.5 of the hash-set probe, no overflow list, no set clearing beyond the k words touched, and the
// LDS reads are 4 B wide instead of 8.  LDS bounds the residency (3 waves of 16 KiB per workgroup, 3 workgroups per CU), which a
// kernel limited by its vector instructions tolerates; NST = gather steps of a cell (8 rows each), a template parameter so that
// the number of requests between two waits is a constant.
// waves per workgroup x cells in flight per wave, measured at 100 k x 50 on permuted ids (tools/bits_ab.sh, profiles/r04_bits_kernel.txt;
// the general kernel: 121 us): 2 x 2: 99 us, 4 x 2: 99, 2 x 3: 102, 3 x 2: 113, 3 x 3: 114, 1 x 2: 118 — what matters is that the
// waves a CU holds (LDS: 16 KiB each) divide evenly over its four SIMDs: 8 per CU (2 or 4 per workgroup), not 9.
#pragma once

#include "jaccard_shared.h"

namespace {

#ifndef GFICF_BITS_WAVES
#define GFICF_BITS_WAVES 2
#endif
#ifndef GFICF_BITS_DEPTH
#define GFICF_BITS_DEPTH 2
#endif
constexpr int BITS_WAVES = GFICF_BITS_WAVES;
constexpr int BITS_DEPTH = GFICF_BITS_DEPTH;                  // cells in flight per wave (2..4)
#ifdef GFICF_BITS_WHATIF_HALF_SET
// LAB ONLY (tools/lab/build_bits_variants.sh): a bit set of half the size — ids alias, the counts are WRONG — to see what the kernel
// would gain from sixteen resident waves per CU instead of eight (profiles/r05_bits_kernel.txt).  Never in the product build.
constexpr uint32_t BITS_WB = 8192u;
#else
constexpr uint32_t BITS_WB = 16384u;                          // LDS bytes of one wave's bit set
#endif
constexpr uint32_t BITS_LUT_OFF = BITS_WAVES * BITS_WB;
constexpr uint32_t BITS_DUPF_OFF = BITS_LUT_OFF + 64u * 8u;   // weight table: k + 1 <= 56 doubles
constexpr size_t BITS_LDS_BYTES = BITS_DUPF_OFF + BITS_WAVES * 4u;

__device__ inline uint32_t lds_read_b32(uint32_t addr) { return *(__attribute__((address_space(3))) const uint32_t*)(size_t)addr; }
__device__ inline void lds_write_b32(uint32_t addr, uint32_t v) { *(__attribute__((address_space(3))) uint32_t*)(size_t)addr = v; }
__device__ inline uint32_t lds_or_rtn_b32(uint32_t addr, uint32_t v) {
  return __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)(size_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NST, int OUT, bool MAP>
__global__ __launch_bounds__(BITS_WAVES * 64) void k_jaccard_edges_bits(
    const uint32_t* __restrict__ table, int64_t N, int k, int64_t cell_begin, int64_t cell_end, EdgeOut o) {
  using F = CFmt<64>;
  extern __shared__ unsigned char smem[];
  double* const s_lut = reinterpret_cast<double*>(smem + BITS_LUT_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int u = tid; u <= k; u += BITS_WAVES * 64) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);   // reference :51
  {
    uint4* const z = reinterpret_cast<uint4*>(smem + (uint32_t)wave * BITS_WB);
    for (int t = lane; t < (int)(BITS_WB / 16); t += 64) z[t] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (lane == 0) *reinterpret_cast<uint32_t*>(smem + BITS_DUPF_OFF + (uint32_t)wave * 4u) = 0u;
  __syncthreads();

  const uint32_t wbase = lds_address(smem) + (uint32_t)(tid >> 6) * BITS_WB;   // a multiple of 16 KiB (dynamic LDS starts at 0: no static LDS here)
#ifdef GFICF_BITS_WHATIF_HALF_SET
  uint32_t mask_v = 0x0FFCu;
#else
  uint32_t mask_v = 0x1FFCu;                                   // word offset inside a plane, as a vector register (operand of v_bitop3_b32)
#endif
  asm volatile("" : "+v"(mask_v));
  const char* const tbytes = reinterpret_cast<const char*>(table);
  const int grow = lane >> 3, gl = lane & 7;
  const uint32_t gcol = 128u + (uint32_t)gl * 16u;            // this lane's piece of the planar part of a row
  const int64_t nwaves = (int64_t)gridDim.x * BITS_WAVES;
  // Lane (row group g = lane / 8, position gl = lane % 8) OWNS slot gl * 8 + g of the cell's row: gather step st serves slots
  // st * 8 .. st * 8 + 7, row group g of the step gathers the row named by slot st * 8 + g — which is held by lane st OF THE SAME
  // GROUP.  So the neighbour id a group needs is a broadcast inside 8 lanes (two DPP moves), the group's count for the step lands
  // in the lane that owns the slot by a select, and the row header (in the group's eighth lane) is a DPP broadcast too: no
  // ds_bpermute anywhere.  The kernel is bound by its LDS pipe (random ds_read_b32 probes replay on bank conflicts); with
  // cross-lane traffic through LDS as well — id, header and count of every step — a cell cost 81 LDS instructions, now 58.
  const int slot = gl * 8 + grow;
  const int slot_c = slot < F::KC ? slot : F::KC - 1;
  const bool slot_ok = slot < F::KC;
  const int64_t first = cell_begin + (int64_t)xcd_block(blockIdx.x, gridDim.x, o.xcd) * BITS_WAVES + wave;
  if (first >= cell_end) return;                               // (after the barrier; wave-uniform)
  const int64_t last_cell = cell_end - 1;
  // lane P (compile-time) of every group of 8 lanes, broadcast to the group's 8 lanes
  auto bcast8 = [](uint32_t v, auto p_tag) -> uint32_t {
    constexpr int P = decltype(p_tag)::value;
    const int x = __builtin_amdgcn_update_dpp(0, (int)v, (P & 3) * 0x55, 0xf, 0xf, false);        // quad_perm [P%4 x 4]: each quad its own lane P%4
    // the quad that holds lane P hands its value to the other quad of the group: row_shr:4 into lanes 4-7 (banks 1, 3), row_shl:4 into 0-3
    return (uint32_t)(P < 4 ? __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xa, false) : __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false));
  };

  struct OwnRaw { uint32_t v, hw, last; };
  auto load_own = [&](int64_t row, OwnRaw& r) {                // the compact part of the row: loads only, decoded one iteration later
    const uint32_t* const rw = table + row * DUAL_PITCH;
    r.last = rw[F::ROWW - 1];
    r.v = reinterpret_cast<const uint16_t*>(rw)[slot_c];
    r.hw = rw[F::HIW + (slot_c >> 5)];
  };
  // id of the lane's slot | bit 31 = the row's duplicate flag; 0 for a lane without a slot or a slot without an id
  auto decode_own = [&](const OwnRaw& r) -> uint32_t {
    const uint32_t half = unscramble16(r.v);
    const uint32_t x = (half | (((r.hw >> (slot & 31)) & 1u) << 16));
    return (slot_ok && x != 0u) ? (x | (r.last & ROW_DUP_FLAG)) : (slot_ok ? (r.last & ROW_DUP_FLAG) : 0u);
  };
  auto gather_step = [&](uint32_t asafe, uint4& piece, auto st_tag) {
    const uint32_t dst = bcast8(asafe, st_tag);                // slot st * 8 + g: lane st of group g
    piece = *reinterpret_cast<const uint4*>(tbytes + (dst - 1u) * (uint32_t)(DUAL_PITCH * 4) + gcol);
  };
  auto issue_gathers = [&](uint32_t asafe, uint4 (&bv)[NST]) {
    gather_step(asafe, bv[0], std::integral_constant<int, 0>{});
    gather_step(asafe, bv[1], std::integral_constant<int, 1>{});
    gather_step(asafe, bv[2], std::integral_constant<int, 2>{});
    gather_step(asafe, bv[3], std::integral_constant<int, 3>{});
    gather_step(asafe, bv[4], std::integral_constant<int, 4>{});
    if constexpr (NST > 5) gather_step(asafe, bv[5], std::integral_constant<int, 5>{});
    if constexpr (NST > 6) gather_step(asafe, bv[6], std::integral_constant<int, 6>{});
  };
  // hits of the two halves of a packed word in the plane at `base`
  // v_lshrrev_b32 takes its shift from the low five bits of the operand: the packed word itself serves for the low half (written
  // in C the compiler masks the operand first and then extracts the bit with the half-rate v_bfe_u32: 6-7 issue slots per id
  // instead of 5-6)
  auto shr5 = [](uint32_t v, uint32_t by) -> uint32_t {
    uint32_t r;
    asm("v_lshrrev_b32 %0, %1, %2" : "=v"(r) : "v"(by), "v"(v));
    return r;
  };
  // hits of the 8 halves of a piece (four packed words) in the plane at `base`: all eight reads are issued before the first is used
  auto probe_piece = [&](const uint32_t (&w)[4], uint32_t base) -> int {
    uint32_t ad[8], bw[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ad[2 * c] = bitop3<0xEA>(w[c] >> 3, mask_v, base);         // ((half >> 5) << 2) | base, the low half
      ad[2 * c + 1] = bitop3<0xEA>(w[c] >> 19, mask_v, base);    // ... the high half
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) bw[t] = lds_read_b32(ad[t]);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t cnt = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) cnt += (shr5(bw[2 * c], w[c]) & 1u) + (shr5(bw[2 * c + 1], w[c] >> 16) & 1u);
    return (int)cnt;
  };

  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  // the edges of a cell: buffer stores through descriptors that cover exactly its k edges (lanes >= k fall outside and are
  // dropped by the hardware: no predicate, no branch); valid == false: a range of zero (nothing is written)
  auto store_cell = [&](int64_t cell, uint32_t dstid, int u, bool valid) {
    const int64_t pb = (cell - cell_begin) * (int64_t)k;
    const int n = valid ? k : 0;
    const bool pos = u > 0;
    if (OUT != OUT_U16) {
      const double vs = pos ? (double)((uint32_t)(cell + 1) + o.src_off) : 0.0;     // reference :49
      const double vd = pos ? (double)dstid : 0.0;                                    // reference :50
      const double vw = s_lut[u];                                                      // reference :51 (lut[0] = 0.0: the zero row)
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(o.src + pb, 0, n * 8, 0x00020000);
      const auto rd = __builtin_amdgcn_make_buffer_rsrc(o.dst + pb, 0, n * 8, 0x00020000);
      const auto rw = __builtin_amdgcn_make_buffer_rsrc(o.w + pb, 0, n * 8, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, vs), rs, slot * 8, 0, 2);      // (a lane's edge is its SLOT's)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, vd), rd, slot * 8, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, vw), rw, slot * 8, 0, 2);
    }
    if (OUT == OUT_RMAT_U) {
      const auto ru = __builtin_amdgcn_make_buffer_rsrc(o.u + pb, 0, n * 4, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32((uint32_t)u, ru, slot * 4, 0, 2);
    }
    if (OUT == OUT_U16) {
      const auto ru = __builtin_amdgcn_make_buffer_rsrc(o.u16 + pb, 0, n * 2, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b16((uint16_t)u, ru, slot * 2, 0, 0);
    }
  };

  // counts of the cell's slots from its gathered pieces; returns whether the cell needs the exact path
  auto process = [&](uint32_t araw, const uint4 (&bv)[NST], int& u_out) -> bool {
    const uint32_t a = araw & 0x1FFFFu;
    const bool has = a != 0u;
    bool dup_here = false;
    uint32_t my_addr = wbase;
    if (has) {                                                  // row i into the bit set
#ifdef GFICF_BITS_WHATIF_HALF_SET
      my_addr = wbase + ((a >> 16) << 12) + ((((a & 0xFFFFu) >> 5) << 2) & 0x0FFCu);
      const uint32_t m = 1u << (a & 31u);
      (void)lds_or_rtn_b32(my_addr, m);
#else
      my_addr = wbase + ((a >> 16) << 13) + (((a & 0xFFFFu) >> 5) << 2);
      const uint32_t m = 1u << (a & 31u);
      dup_here = (lds_or_rtn_b32(my_addr, m) & m) != 0u;        // already there: the row names the id twice
#endif
    }
    wave_lds_fence();
    uint32_t hflags = araw;
    int myu = 0;
#ifdef GFICF_BITS_PREFETCH
    // LAB VARIANT (profiles/r05_bits_kernel.txt): the eight set reads of step st + 1 are issued before the counts of step st are taken
    {
      uint32_t w4[2][4], bw[2][8];
      auto issue = [&](int st, uint32_t (&w)[4], uint32_t (&b)[8]) {
        const uint32_t hdr = bcast8(bv[st].w, std::integral_constant<int, 7>{});
        hflags |= hdr;
        const bool p1 = (uint32_t)gl >= (hdr & 15u);
        const uint32_t base = wbase | (p1 ? (BITS_WB >> 1) : 0u);
        w[0] = bv[st].x; w[1] = bv[st].y; w[2] = bv[st].z; w[3] = gl == 7 ? (p1 ? 0xFFFFFFFFu : 0u) : bv[st].w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          b[2 * c] = lds_read_b32(bitop3<0xEA>(w[c] >> 3, mask_v, base));
          b[2 * c + 1] = lds_read_b32(bitop3<0xEA>(w[c] >> 19, mask_v, base));
        }
      };
      issue(0, w4[0], bw[0]);
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        if (st + 1 < NST) issue(st + 1, w4[(st + 1) & 1], bw[(st + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t cnt = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) cnt += (shr5(bw[st & 1][2 * c], w4[st & 1][c]) & 1u) + (shr5(bw[st & 1][2 * c + 1], w4[st & 1][c] >> 16) & 1u);
        const int rowcnt = group_sum<8>((int)cnt);
        myu = gl == st ? rowcnt : myu;
      }
    }
#else
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const uint32_t hdr = bcast8(bv[st].w, std::integral_constant<int, 7>{});   // the row's header word sits in its eighth lane
      hflags |= hdr;
      const bool p1 = (uint32_t)gl >= (hdr & 15u);                               // this lane's 8 ids: first or second plane
#ifdef GFICF_BITS_WHATIF_HALF_SET
      const uint32_t base = wbase | (p1 ? 0x1000u : 0u);
#else
      const uint32_t base = wbase | (p1 ? 0x2000u : 0u);
#endif
      const uint32_t w4[4] = {bv[st].x, bv[st].y, bv[st].z, gl == 7 ? (p1 ? 0xFFFFFFFFu : 0u) : bv[st].w};   // (the header is not an id: the plane's pad instead)
      const int c = probe_piece(w4, base);
      const int rowcnt = group_sum<8>(c);                                        // every lane of the group: the row's count
      myu = gl == st ? rowcnt : myu;                                             // ... kept by the lane that owns slot st * 8 + g
    }
#endif
    if (has) lds_write_b32(my_addr, 0u);                        // the set is empty again (lanes sharing a word write the same zero)
    if (dup_here) *reinterpret_cast<uint32_t*>(smem + BITS_DUPF_OFF + (uint32_t)wave * 4u) = 1u;   // reported at the kernel's end
    wave_lds_fence();
    u_out = has ? myu : 0;                                      // rejected id / no slot: zero row
    return __ballot(dup_here || (hflags & ROW_DUP_FLAG) != 0u) != 0ull;
  };

  // ---- BITS_DEPTH cells in flight per wave: while cell m is counted, the gathers of cells m+1 .. m+DEPTH-1 and the own row of cell
  // m+DEPTH are outstanding.  Ring slots are compile-time indices (the loop is unrolled by DEPTH).  Two — the pipelined kernel's
  // depth — is enough: three or four changed nothing (the kernel is not waiting for its gathers; profiles/r04_bits_kernel.txt).
  constexpr int DEPTH = BITS_DEPTH;
  OwnRaw raw;
  uint4 bv[DEPTH][NST];
  uint32_t araw_r[DEPTH], gid_r[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) { araw_r[j] = 0u; gid_r[j] = 0u; }
  load_own(first, raw);
#pragma unroll
  for (int j = 0; j < DEPTH - 1; ++j) {                        // cells 0 .. DEPTH-2 of the wave: own row, gathers
    const int64_t c = first + (int64_t)j * nwaves;
    const uint32_t ar = c < cell_end ? decode_own(raw) : 0u;
    const uint32_t a = ar & 0x1FFFFu;
    const int64_t cn = c + nwaves;
    load_own(cn < cell_end ? cn : last_cell, raw);
    const uint32_t fb = (uint32_t)((c < cell_end ? c : last_cell) + 1);
    araw_r[j] = ar;
    gid_r[j] = MAP ? (uint32_t)o.l2g[(a != 0u ? a : fb) - 1u] : 0u;
    issue_gathers(a != 0u ? a : fb, bv[j]);
  }
  bool any_slow = false, have_prev = false;
  int64_t prev_i = first;
  uint32_t prev_a = 0;
  int prev_u = 0;

  auto body = [&](int64_t i, auto j_tag) {
    constexpr int J = decltype(j_tag)::value, JN = (J + DEPTH - 1) % DEPTH;
    const int64_t ig = i + (int64_t)(DEPTH - 1) * nwaves, io = ig + nwaves;
    const uint32_t araw_n = ig < cell_end ? decode_own(raw) : 0u;              // its own row was requested an iteration ago
    const uint32_t an = araw_n & 0x1FFFFu;
    load_own(io < cell_end ? io : last_cell, raw);                            // the own row of the cell after that one FIRST
    __builtin_amdgcn_sched_barrier(0);
    gid_r[JN] = MAP ? (uint32_t)o.l2g[(an != 0u ? an : (uint32_t)(i + 1)) - 1u] : 0u;
    issue_gathers(an != 0u ? an : (uint32_t)(i + 1), bv[JN]);                 // no such cell / no id: the lane reads row i (one line)
    araw_r[JN] = araw_n;
    __builtin_amdgcn_sched_barrier(0);
    store_cell(prev_i, prev_a, prev_u, have_prev);                            // the cell before this one: behind the gathers
    __builtin_amdgcn_sched_barrier(0);
    int u;
    any_slow |= process(araw_r[J], bv[J], u);
    prev_i = i;
    prev_a = MAP ? gid_r[J] : (araw_r[J] & 0x1FFFFu);
    prev_u = u;
    have_prev = true;
  };

  for (int64_t i = first;;) {
    body(i, std::integral_constant<int, 0>{});
    if (i + nwaves >= cell_end) break;
    i += nwaves;
    body(i, std::integral_constant<int, 1 % DEPTH>{});
    if (i + nwaves >= cell_end) break;
    i += nwaves;
    if constexpr (DEPTH >= 3) {
      body(i, std::integral_constant<int, 2 % DEPTH>{});
      if (i + nwaves >= cell_end) break;
      i += nwaves;
    }
    if constexpr (DEPTH >= 4) {
      body(i, std::integral_constant<int, 3 % DEPTH>{});
      if (i + nwaves >= cell_end) break;
      i += nwaves;
    }
  }
  store_cell(prev_i, prev_a, prev_u, true);
  // ---- the deferred report of the "distinct ids" mode, and the exact path for cells whose own row or a neighbour row holds an id
  // twice (never the case for real kNN output): after the loop; their fast-path rows written above are overwritten
  if (any_slow) {
    __builtin_amdgcn_s_waitcnt(0);
    wave_lds_fence();
    if (*reinterpret_cast<const uint32_t*>(smem + BITS_DUPF_OFF + (uint32_t)wave * 4u) != 0u && lane == 0) {
      uint32_t* const st = edge_kernel_dup_status();
      if (st != nullptr) atomicOr(st, GFICF_ST_DUP_IDS);
    }
    uint32_t* const sA = reinterpret_cast<uint32_t*>(smem + (uint32_t)wave * BITS_WB);      // (the wave's bit set is no longer needed)
    uint32_t* const sB = sA + 64;
    for (int64_t c = first; c < cell_end; c += nwaves) {
      const uint32_t* const rw = table + c * DUAL_PITCH;
      const uint32_t a = lane < k ? row_slot_id(rw, lane, 64, true) : 0u;
      bool f = row_dup_flag(rw, 64, true);
      if (a != 0u) f |= row_dup_flag(table + (int64_t)(a - 1u) * DUAL_PITCH, 64, true);
      if (__ballot(f) != 0ull)
        slow_cell<64, true, OUT>(table, c, k, (c - cell_begin) * (int64_t)k, sA, sB, lane, o.src, o.dst, o.w, o.u, o.u16, o.set_mode, s_lut, o.l2g,
                                 o.src_off, DUAL_PITCH);
    }
  }
}

}  // namespace
