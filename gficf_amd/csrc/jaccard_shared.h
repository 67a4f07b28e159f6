// jaccard_shared.h — what the Jaccard kernels share: the table formats (one row per cell; a function of (N, k) and the GFICF_JACCARD_* switches),
// the edge kernels' configuration (JCfg), output descriptor (EdgeOut), set probes and the exact per-cell path (slow_cell).  Until round 6
// this text sat in jaccard.hip between the #include lines of the kernel headers, which therefore compiled only inside that file, in that
// order; now every kernel header includes this one and stands on its own (`hipcc -fsyntax-only -x hip <header>`: tests/test_abi.py).
// Device code unchanged: the same text in the same order in the one translation unit that uses it (jaccard.hip).
#pragma once

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"
#include "halo_map.h"

namespace {


constexpr uint32_t ROW_DUP_FLAG = 0x80000000u;
constexpr uint32_t ID_MASK = 0x7FFFFFFFu;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

__host__ __device__ inline int kpad_for(int k) {
  return k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : k <= 128 ? 128 : 256;
}

// ------------------------------------------------------------------------ table row formats
// wide    : KPAD x uint32 ids (zero padded); bit 31 of word 0 = "row holds duplicate ids".
// compact : for data sets of fewer than 2^17 cells (ids fit 17 bits) a row of KPAD slots takes half the
//           bytes: KC = KPAD - KPAD/16 ids as uint16 halves, then NW = KPAD/32 words of high bits
//           (bit j of the bitmap = bit 16 of id j); bit 31 of the row's last word = the duplicate flag.
//           k = 30 -> 64 B instead of 128 B per row.  The edge kernel is bound by the row gathers (one
//           L1 miss per edge, served by L2 / Infinity Cache): half the table means twice the L2 hit rate.
//           The 16-bit halves are stored PRE-HASHED: half = rotr16((id & 0xFFFF) * A mod 2^16, 5), a bijection of
//           the low 16 id bits whose bits 3..10 are the top byte of the multiplicative hash — the byte offset of the
//           id's bucket in the edge kernel's hash set is then one AND of the stored half, and membership is tested
//           on the stored form itself (stored halves are equal iff the ids' low halves are).  On gfx950 most integer
//           vector instructions (shifts left, 24-bit multiplies, three-operand and/or, min3, every SDWA / DPP / packed
//           form) issue at half the rate of and / or / xor / add / shift-right / v_bitop3 (tools/lab/valu_lab.hip,
//           profiles/r02_valu_rates.txt), and the edge kernel spends its time in exactly those per probed id.
// The format is a function of (N_total, k) alone, so every rank of a sharded build agrees on it.
template <int KPAD>
struct CFmt {
  static constexpr int KC = KPAD - KPAD / 16;   // usable slots
  static constexpr int NW = KPAD / 32;          // words of high bits
  static constexpr int ROWW = KPAD / 2;         // row pitch in 32-bit words
  static constexpr int HIW = ROWW - NW;         // word index of the first high-bit word
};

constexpr uint32_t SCR_A = 0x9E37u, SCR_AINV = 0x7787u;     // A * AINV = 1 (mod 2^16)
__host__ __device__ inline uint32_t scramble16(uint32_t lo) {
  const uint32_t s = (lo * SCR_A) & 0xFFFFu;
  return ((s >> 5) | (s << 11)) & 0xFFFFu;
}
__host__ __device__ inline uint32_t unscramble16(uint32_t half) {
  const uint32_t s = ((half << 5) | (half >> 11)) & 0xFFFFu;
  return (s * SCR_AINV) & 0xFFFFu;
}

// ---- sorted rows (csrc/jaccard_sorted.h): which k take them
__host__ __device__ inline int sorted_kp(int k) { return (k + 63) & ~63; }
// The smallest k that takes this path.  Beyond GFICF_JACCARD_MAX_K it is the only one; below, the general hash-set kernel
// (k_jaccard_edges, 64 < k <= 256) competes with it, and loses from SORTED_FROM_DEFAULT on (profiles/r05_sorted_vs_general.txt).
// GFICF_JACCARD_SORTED_FROM in the environment moves the switch (57 .. 257; every rank of a sharded build must see the same value:
// dist.assert_same_format compares the GFICF_JACCARD_* environment).
#ifndef GFICF_JACCARD_SORTED_FROM_DEFAULT
#define GFICF_JACCARD_SORTED_FROM_DEFAULT 257
#endif
inline int sorted_from_k() {
  static const int v = [] {
    const char* e = getenv("GFICF_JACCARD_SORTED_FROM");
    const int t = e ? atoi(e) : GFICF_JACCARD_SORTED_FROM_DEFAULT;
    return t < 57 ? 57 : t > GFICF_JACCARD_MAX_K + 1 ? GFICF_JACCARD_MAX_K + 1 : t;
  }();
  return v;
}
// A host entry re-running its call after GFICF_ERR_SET_OVERFLOW takes the sorted-row path from k = 57 on, whatever the switch says
// (thread-local: the format is otherwise a function of (N, k) and the environment alone, and stays one for every device entry).
thread_local int g_force_sorted = 0;
inline bool sorted_fmt(int k) { return k >= sorted_from_k() || (g_force_sorted && k >= 57); }

struct TableFmt {
  int kpad;
  bool compact;
  bool dual;         // compact rows + a second, "planar" copy of the ids for the gathers of k_jaccard_edges_bits (below)
  int row_words;     // row PITCH of the table in 32-bit words
  bool sorted;       // k > GFICF_JACCARD_MAX_K: slot-order ids + the same ids ascending (jaccard_sorted.h)
};

// ---- dual rows (round 4): 32 < k <= 55 and N <= 131070.  The general edge kernel is bound by its probe arithmetic at these row
// sizes (100 k x 50: 6.6e7 vector instructions, 47 % of its LDS time conflict replays — profiles/r03_pmc_summary_c4.txt), and two of
// the ~7.5 issue slots of a probed id go into pulling the id's bit 16 out of the row's bitmap.  A direct-address BIT SET in LDS
// (2^17 bits = 16 KiB per wave) needs no hash, no key compare and no overflow list — word address, one ds_read_b32, shift, and,
// add — IF the plane (bit 16) of an id costs nothing per id.  So a table row becomes 64 words:
//   words  0..31  the compact row as before: what a cell's OWN row is read from (slot order = the reference's edge order), what
//                 every other kernel (exact path, edge filter, transport) reads;
//   words 32..63  the same ids once more, regrouped for the GATHERS: the ids below 2^16 first, in groups of 8 halves (one 16 B
//                 lane piece each) padded with 0x0000, then the ids from 2^16 on as (id - 2^16), padded with 0xFFFF — so every
//                 lane's 8 ids lie in ONE plane and the plane is a property of the lane; plain halves, not pre-hashed.  Word 63
//                 is a header: bits 0..3 = number of groups of the first plane, bit 31 = the row's duplicate flag.  62 slots hold
//                 any split of k <= 55 ids into two padded runs; id 0 ("none") pads the first plane, id 131071 the second:
//                 neither is ever in a set (ids are 1..N, N <= 131070).
// The counts do not depend on the order of a GATHERED row's ids, only the own row needs its slots in order: hence two copies.
// The gathered 128 B are one line, as before; the table doubles (25.6 MB at 100 k cells) but the lines the gathers touch do not.
constexpr int DUAL_PITCH = 64;
constexpr uint32_t DUAL_PAD1 = 0xFFFFu;
inline bool dual_enabled() {
  const char* e = getenv("GFICF_JACCARD_DUAL");            // A/B switch, read per call: 0 = the general kernel on plain compact rows
  return !(e && atoi(e) == 0);
}

inline bool compact_enabled() {
  static const bool on = [] {
    const char* e = getenv("GFICF_JACCARD_COMPACT");       // test hook: 0 keeps every table in the wide format
    return !(e && atoi(e) == 0);
  }();
  return on;
}

inline TableFmt table_fmt(int64_t N_total, int k) {
  TableFmt f;
  f.sorted = sorted_fmt(k);
  if (f.sorted) {                                            // (a function of k alone)
    f.kpad = 2 * ((k + 63) & ~63);
    f.compact = f.dual = false;
    f.row_words = f.kpad;
    return f;
  }
  f.kpad = kpad_for(k);
  static const bool force_big = getenv("GFICF_JACCARD_FORCE_BIG") != nullptr;     // test hook of the 64-bit kernel variant: wide rows
  f.compact = compact_enabled() && !force_big && N_total < (1ll << 17) && f.kpad >= 32 && k <= f.kpad - f.kpad / 16;
  f.dual = f.compact && f.kpad == 64 && k <= 55 && N_total <= 131070 && dual_enabled();
  f.row_words = f.dual ? DUAL_PITCH : f.compact ? f.kpad / 2 : f.kpad;
  return f;
}

// id of slot j (0 = none) / duplicate flag of a row given as 32-bit words (global memory or LDS)
__device__ inline uint32_t row_slot_id(const uint32_t* roww, int j, int kpad, bool compact) {
  if (!compact) return roww[j] & ID_MASK;
  const int kc = kpad - kpad / 16;
  if (j >= kc) return 0u;
  const uint32_t lo = unscramble16((roww[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
  const uint32_t hw = roww[kpad / 2 - kpad / 32 + (j >> 5)];
  return lo | (((hw >> (j & 31)) & 1u) << 16);
}
__device__ inline bool row_dup_flag(const uint32_t* roww, int kpad, bool compact) {
  return ((compact ? roww[kpad / 2 - 1] : roww[0]) & ROW_DUP_FLAG) != 0;
}

__device__ inline void wave_lds_fence_early() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// The planar part of a dual row (32 words, built in LDS by one wave): ids = the row's slot ids (0 = none), lane = slot.
__device__ inline void planar_row_build(const uint32_t* ids, int k, bool dupflag, uint32_t* prow, int lane) {
  const bool valid = lane < k;
  const uint32_t id = valid ? ids[lane] : 0u;
  const bool p1 = valid && id >= 65536u, p0 = valid && !p1;
  const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
  const int g0 = (__popcll(m0) + 7) >> 3;                      // groups of 8 halves of the first plane
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint16_t* const ph = reinterpret_cast<uint16_t*>(prow);
  ph[lane] = lane < 8 * g0 ? (uint16_t)0u : (uint16_t)DUAL_PAD1;   // 64 halves: the pads of both planes (the last two become the header)
  wave_lds_fence_early();
  if (p0) ph[__popcll(m0 & lt)] = (uint16_t)id;
  if (p1) ph[8 * g0 + __popcll(m1 & lt)] = (uint16_t)(id & 0xFFFFu);
  wave_lds_fence_early();
  if (lane == 0) prow[31] = (uint32_t)g0 | (dupflag ? ROW_DUP_FLAG : 0u);
  wave_lds_fence_early();
}

// Four rows at a time (the LDS round trips of the three phases are shared by the four): ids[r] = row r's slot ids, prow[r] its
// 32-word scratch, n = rows that exist (1..4); dup bit r of dupmask = row r's duplicate flag.
// bm (may be null): bm[r] = the 64-bit mask of row r's slots whose id is >= 65536 — the bit-16 bitmap of the row's compact part.
__device__ inline void planar_rows_build4(const uint32_t* const (&ids)[4], int n, int k, uint32_t dupmask, uint32_t (*prow)[32], int lane,
                                          uint32_t (*bm)[2] = nullptr) {
  const bool valid = lane < k;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t id[4];
  unsigned long long m0[4], m1[4];
  int g0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    id[r] = (valid && r < n) ? ids[r][lane] : 0u;
    const bool p1 = valid && id[r] >= 65536u, p0 = valid && !p1;
    m0[r] = __ballot(p0);
    m1[r] = __ballot(p1);
    g0[r] = (__popcll(m0[r]) + 7) >> 3;
    reinterpret_cast<uint16_t*>(prow[r])[lane] = lane < 8 * g0[r] ? (uint16_t)0u : (uint16_t)DUAL_PAD1;
  }
  wave_lds_fence_early();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint16_t* const ph = reinterpret_cast<uint16_t*>(prow[r]);
    const bool p1 = valid && id[r] >= 65536u;
    if (valid) ph[p1 ? 8 * g0[r] + __popcll(m1[r] & lt) : __popcll(m0[r] & lt)] = (uint16_t)(id[r] & 0xFFFFu);
  }
  wave_lds_fence_early();
  {
    const int gsel = lane == 0 ? g0[0] : lane == 1 ? g0[1] : lane == 2 ? g0[2] : g0[3];
    if (lane < 4) prow[lane][31] = (uint32_t)gsel | (((dupmask >> lane) & 1u) ? ROW_DUP_FLAG : 0u);
    if (bm != nullptr && lane < 4) {
      const unsigned long long msel = lane == 0 ? m1[0] : lane == 1 ? m1[1] : lane == 2 ? m1[2] : m1[3];
      bm[lane][0] = (uint32_t)msel;
      bm[lane][1] = (uint32_t)(msel >> 32);
    }
  }
  wave_lds_fence_early();
}


// (the ingest kernels use the part above only; the part below belongs to the edge kernels)

// ------------------------------------------------------------------------------- edges
template <int KPAD>
constexpr int jc_threads = (KPAD <= 128 ? 4 : 2) * 64;        // threads of an edge-kernel workgroup

template <int KPAD, bool CMP>
struct JCfg {
  static constexpr int ROWB = CMP ? KPAD * 2 : KPAD * 4;      // bytes per table row
  static constexpr int LPR = ROWB / 16;                       // lanes per neighbour row (16 B per lane)
  static constexpr int RPS = 64 / LPR;                        // neighbour rows per wave-instruction ("step")
  static constexpr int IPL = CMP ? 8 : 4;                     // ids a lane holds of a gathered row
  static constexpr int NSLOT = CMP ? CFmt<KPAD>::KC : KPAD;   // usable slots of a row
  static constexpr int EPL = KPAD > 64 ? KPAD / 64 : 1;       // registers holding row i (slot s -> reg s/64, lane s%64)
  static constexpr int SPQ = (KPAD < 64 ? KPAD : 64) / RPS;   // steps per register of row i
  static constexpr int NB = 8 * KPAD;                         // 2-slot buckets in the hash set
  static constexpr int WAVES = KPAD <= 128 ? 4 : 2;           // waves per workgroup
#ifndef GFICF_JACCARD_U
#define GFICF_JACCARD_U 8
#endif
  // steps whose gathers are in flight together in the one-cell-at-a-time kernel: 8 = the whole cell at 32 < k <= 64 (32 registers
  // of pieces; 95 registers in all, five waves per SIMD as before): 145 -> 139.5 us at 100 k x 50 against two batches of 4
  static constexpr int U = SPQ < GFICF_JACCARD_U ? SPQ : GFICF_JACCARD_U;
  static constexpr int LOG2NB = KPAD == 16 ? 7 : KPAD == 32 ? 8 : KPAD == 64 ? 9 : KPAD == 128 ? 10 : 11;
};

// Byte offset of an id's bucket inside a wave's hash set: bits [3, 3+LOG2NB) of id*K, i.e. a
// multiplicative hash of the id's low 3+LOG2NB bits.  BIG == false: ids < 2^24, full-rate 24-bit
// multiply (bound to the intrinsic by name: written as a plain product the masked multiply is
// canonicalised to the quarter-rate v_mul_lo_u32).
// LDS of an edge kernel: hash sets | own rows and overflow lists | weight table | (pipelined kernel) staging rows of the quad
// stores | one word per wave: "a row of this wave's cells names an id twice" (gficf_ctx_set_jaccard_distinct)
template <int KPAD, bool CMP>
constexpr uint32_t edges_dupflag_off() {
  using C = JCfg<KPAD, CMP>;
  return (uint32_t)(C::WAVES * C::NB * 8 + C::WAVES * 2 * KPAD * 4 + (GFICF_JACCARD_MAX_K + 1) * (int)sizeof(double) +
                    ((C::EPL == 1 && C::SPQ <= 4) ? C::WAVES * 4 * 64 * 8 : 0));
}
template <int KPAD, bool CMP>
constexpr size_t edges_lds_bytes() { return (size_t)edges_dupflag_off<KPAD, CMP>() + (size_t)JCfg<KPAD, CMP>::WAVES * 4; }

// The status word of the deferred duplicate report, read from the kernel's argument block only where it is needed (at the
// kernel's end, by a wave that met a repeated id): referenced as `o.dup_status` it would be loaded with the other arguments
// at the kernel's start and live in scalar registers for the whole kernel — the edge kernels sit at 91-95 vector registers
// with the scalar file full, and every pair kept alive there spills into vector registers and costs a wave per SIMD.
// Arguments: table (8) N (8) k (4 + 4) cell_begin (8) cell_end (8) EdgeOut.
struct EdgeOut;
__device__ inline uint32_t* edge_kernel_dup_status();

extern "C" __device__ uint32_t gficf_mul_u24(uint32_t a, uint32_t b) __asm("llvm.amdgcn.mul.u24.i32");

template <int KPAD, bool BIG>
__device__ inline uint32_t bucket_off(uint32_t id) {
  constexpr uint32_t HMASK = (uint32_t)(JCfg<KPAD, false>::NB - 1) << 3;
  return (BIG ? id * 0x9E3779B1u : gficf_mul_u24(id, 0x9E3779u)) & HMASK;
}

// LDS accessed at an integer byte address (base | offset folds into one v_and_or_b32 per probe).
typedef uint32_t gficf_v2u __attribute__((ext_vector_type(2)));
__device__ inline uint32_t lds_address(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ inline uint2 lds_read_b64(uint32_t addr) {
  const gficf_v2u v = *(__attribute__((address_space(3))) const gficf_v2u*)(size_t)addr;
  return make_uint2(v.x, v.y);
}

// min(a, b, 1): 0 iff a == 0 or b == 0.  Bound to v_min3_u32 by hand: written as C the compiler turns the "min with 1"
// back into a compare + conditional add through VCC.
__device__ inline uint32_t min3u_one(uint32_t a, uint32_t b) {
  uint32_t m;
  asm("v_min3_u32 %0, %1, %2, 1" : "=v"(m) : "v"(a), "v"(b));
  return m;
}

// v_bitop3_b32: any bitwise function of three inputs at the FULL vector rate (v_and_or_b32, v_or3_b32, v_xor3 forms issue at
// half of it on gfx950).  TT: truth table, bit (a << 2 | b << 1 | c) = f(a, b, c).  0xEA = (a & b) | c, 0x96 = a ^ b ^ c.
template <int TT>
__device__ inline uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(TT));
  return r;
}

// c + (this lane's bit of the 64-bit lane mask m): one v_addc with the mask as carry-in.
__device__ inline int add_lane_bit(int c, unsigned long long m) {
  int r;
  asm("v_addc_co_u32_e64 %0, vcc, %1, 0, %2" : "=v"(r) : "v"(c), "s"(m) : "vcc");
  return r;
}

// v_writelane_b32: drop a wave-uniform value into one lane of a VGPR (clang exposes no
// builtin for it; bind the LLVM intrinsic by name).
extern "C" __device__ int gficf_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

__device__ inline void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Does the overflow list of a cell's hash-set build (ids that found both slots of their bucket taken) hold an id twice?  Up to
// six entries are compared pair by pair — uniform LDS reads, fifteen compares, no loop: a loop over the list cost the kernels 16
// scalar and 5 vector registers and with them a wave per SIMD —; a longer list (one row in 10^11 at k = 30) is REPORTED as a
// repeat, which only costs the caller the exact re-run.
__device__ inline bool ovlist_repeats(const uint32_t* ovlist, int nov) {
  uint32_t v[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) v[t] = t < nov ? ovlist[t] : 0xFFFFFFF0u + (uint32_t)t;     // (distinct values no key takes)
  bool r = nov > 6;
#pragma unroll
  for (int a = 1; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < a; ++b) r |= v[a] == v[b];
  return r;
}

struct EdgeOut {
  double* src;      // the three columns of the reference's edge matrix (all NULL: counts only)
  double* dst;
  double* w;
  int32_t* u;       // optional intersection counts
  uint16_t* u16;    // optional intersection counts, compact (input of the edge filter)
  int set_mode;     // rows with duplicate ids: 0 = multiset intersection (std::set_intersection of the parallel entry),
                    // 1 = set intersection (Rcpp::intersect of the serial jaccard_coeff entry)
  // sharded sub-problem in local ids (halo.hip): column 1 is src_off + cell + 1, column 2 l2g[local id - 1] (NULL: the id itself)
  const int32_t* l2g = nullptr;
  uint32_t src_off = 0;
  // gficf_ctx_set_jaccard_distinct: the table was ingested without the duplicate scan; a cell whose own row names an id twice
  // (seen while the row goes into the hash set) ORs GFICF_ST_DUP_IDS here.  NULL: rows carry their duplicate flag (the scan ran).
  uint32_t* dup_status = nullptr;
  uint32_t xcd = 1;   // workgroups renumbered so that each XCD (workgroup index mod 8) works on one contiguous run of cells
};

// The parameter list of BOTH edge kernels as the kernel argument block lays it out (every argument at its natural alignment):
// the byte offset of the EdgeOut argument is derived from it, and the kernels' signatures are checked against it right behind
// their definitions (static_assert on the function types) — a parameter added or moved without this struct following fails to compile.
struct EdgeKernArgs { const uint32_t* table; int64_t N; int k; int64_t cell_begin; int64_t cell_end; EdgeOut o; };
using EdgeKernFn = void (*)(const uint32_t*, int64_t, int, int64_t, int64_t, EdgeOut);
constexpr int EDGE_KERNARG_OUT = (int)offsetof(EdgeKernArgs, o);
static_assert(EDGE_KERNARG_OUT == 40 && offsetof(EdgeKernArgs, cell_end) == 32, "kernel argument block of the edge kernels");
__device__ inline uint32_t* edge_kernel_dup_status() {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const char __attribute__((address_space(4))) * kptr;                // the kernel argument block lives in constant memory
  typedef uint32_t* const __attribute__((address_space(4))) * kslot;
  const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  return *(kslot)(ka + EDGE_KERNARG_OUT + offsetof(EdgeOut, dup_status));
#else
  return nullptr;
#endif
}

// Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2.  Renumbered, the workgroups of XCD x are
// x*nb/8 ... (x+1)*nb/8 - 1: at any time an XCD then counts one contiguous run of cells, and table rows that cells next to
// each other share (ids with locality) are found in that XCD's L2 instead of being fetched once per XCD.  Ids in order:
// 1 M x 30 +4 %, 100 k x 30 +1.4 %, 100 k x 50 +2 %; scrambled ids: no difference (profiles/r03_xcd_renumbering.txt).
__device__ inline uint32_t xcd_block(uint32_t b, uint32_t nb, uint32_t on) {
  return (on && (nb & 7u) == 0) ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
}

// Output modes of the edge kernel (a template parameter, so that the number of stores per cell is known to the
// compiler: it can then wait for the row gathers alone, leaving the stores issued behind them in flight)
constexpr int OUT_RMAT = 0;       // the three columns of the reference's edge matrix
constexpr int OUT_RMAT_U = 1;     // the same + int32 intersection counts
constexpr int OUT_U16 = 2;        // uint16 intersection counts only (edge filter, compact host return)

template <int OUT>
__device__ inline void store_edge(const EdgeOut o, int64_t r, int64_t cell, uint32_t dst, int u,
                                  const double* lut) {
  const bool pos = u > 0;
  // written once, never re-read by this kernel: non-temporal, so the table rows keep the L2
  if (OUT != OUT_U16) {
    __builtin_nontemporal_store(pos ? (double)((uint32_t)(cell + 1) + o.src_off) : 0.0, o.src + r);   // reference :49 (cell + 1 <= 2^31)
    __builtin_nontemporal_store(pos ? (double)dst : 0.0, o.dst + r);                    // reference :50
    __builtin_nontemporal_store(lut[u], o.w + r);                                       // reference :51 (lut[0] = 0/(2k) = 0.0: the zero row)
  }
  if (OUT == OUT_RMAT_U) __builtin_nontemporal_store(u, o.u + r);
  if (OUT == OUT_U16) o.u16[r] = (uint16_t)u;
}

// Exact multiset path for one cell whose own row or one of whose neighbour rows holds
// duplicate ids (never the case for real kNN output).
// u = sum over distinct values of min(multiplicity in A, multiplicity in B), evaluated as
// "element e of B counts iff its occurrence rank within B is below the value's multiplicity in A".
template <int KPAD, bool CMP, int OUT>
__device__ __noinline__ void slow_cell(const uint32_t* __restrict__ table, int64_t i, int k, int64_t out_base,
                                       uint32_t* sA, uint32_t* sB, int lane, double* o_src, double* o_dst,
                                       double* o_w, int32_t* o_u, uint16_t* o_u16, int set_mode, const double* lut,
                                       const int32_t* l2g, uint32_t src_off, int pitch = 0) {
  const EdgeOut o{o_src, o_dst, o_w, o_u, o_u16, set_mode, nullptr, src_off};
  const int ROWW = pitch ? pitch : (CMP ? CFmt<KPAD>::ROWW : KPAD);     // row pitch (dual rows: 64 words, the compact part in front)
  for (int e = lane; e < KPAD; e += 64) sA[e] = row_slot_id(table + i * ROWW, e, KPAD, CMP);
  wave_lds_fence();
  for (int s = 0; s < k; ++s) {
    const uint32_t dst = sA[s];
    int u = 0;
    if (dst != 0) {
      for (int e = lane; e < KPAD; e += 64) sB[e] = row_slot_id(table + (int64_t)(dst - 1) * ROWW, e, KPAD, CMP);
      wave_lds_fence();
      int cnt = 0;
      for (int e = lane; e < k; e += 64) {
        const uint32_t b = sB[e];
        if (b != 0) {
          int rank = 0, ca = 0;
          for (int t = 0; t < k; ++t) {
            ca += (sA[t] == b);
            rank += (t < e && sB[t] == b);
          }
          cnt += set_mode ? (rank == 0 && ca > 0) : (rank < ca);      // first occurrence of a shared value / min multiplicity
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
      u = cnt;
      wave_lds_fence();
    }
    if (lane == 0) store_edge<OUT>(o, out_base + s, i, (l2g && dst) ? (uint32_t)l2g[dst - 1] : dst, u, lut);
  }
}

// Hits among the 8 ids of a gathered piece of a compact row against the wave's hash set, on the stored form of the ids
// (pre-hashed half | bit 16): wd = the piece's four words (words of high bits zeroed), hb = the byte of bit-16 values of
// the 8 ids, bmask / bit16 = (NB-1) << 3 and 0x10000 in vector registers, base = LDS byte address of the set.  Per id: half
// (and / shift right), bucket address (one v_bitop3), its bit 16 (shift right + and), two three-way XORs against the
// bucket's slots (v_bitop3), a min3 and an add — 18 issue cycles against 29 for the assembled-id form.  The (rare) pass over
// the overflow list rebuilds the stored form with piece_key().
#ifndef GFICF_PROBE_BATCH
#define GFICF_PROBE_BATCH 8
#endif
// stored form (half | bit 16) of id t of a piece, as the hash set holds it
__device__ inline uint32_t piece_key(const uint32_t (&wd)[4], uint32_t hb, int t) {
  const uint32_t half = (t & 1) ? (wd[t >> 1] >> 16) : (wd[t >> 1] & 0xFFFFu);
  return half | (((hb >> t) & 1u) << 16);
}
// B16 = false: every id of the data set is below 2^16 (no bit 16 anywhere): the compare is a two-way XOR on the halves.
template <bool B16 = true>
__device__ inline int probe_compact_piece(const uint32_t (&wd)[4], uint32_t hb, uint32_t bmask, uint32_t bit16, uint32_t base) {
  const uint32_t H = hb << 16;
  uint32_t miss = 0;
#pragma unroll
  for (int b = 0; b < 8; b += GFICF_PROBE_BATCH) {
    uint2 h[GFICF_PROBE_BATCH];
    uint32_t key[GFICF_PROBE_BATCH], hs[GFICF_PROBE_BATCH];
#pragma unroll
    for (int t = 0; t < GFICF_PROBE_BATCH; ++t) {
      const int tt = b + t;
      key[t] = (tt & 1) ? (wd[tt >> 1] >> 16) : (wd[tt >> 1] & 0xFFFFu);
      h[t] = lds_read_b64(bitop3<0xEA>(key[t], bmask, base));          // (half & mask) | base
      hs[t] = B16 ? ((tt ? (H >> tt) : H) & bit16) : 0u;
    }
#pragma unroll
    for (int t = 0; t < GFICF_PROBE_BATCH; t += 2) {
      uint32_t m0, m1;
      if (B16) {
        m0 = min3u_one(bitop3<0x96>(h[t].x, key[t], hs[t]), bitop3<0x96>(h[t].y, key[t], hs[t]));
        m1 = min3u_one(bitop3<0x96>(h[t + 1].x, key[t + 1], hs[t + 1]), bitop3<0x96>(h[t + 1].y, key[t + 1], hs[t + 1]));
      } else {
        m0 = min3u_one(h[t].x ^ key[t], h[t].y ^ key[t]);
        m1 = min3u_one(h[t + 1].x ^ key[t + 1], h[t + 1].y ^ key[t + 1]);
      }
      miss += m0 + m1;
    }
#if GFICF_PROBE_BATCH < 8
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  return 8 - (int)miss;
}

// Sum over the LPR consecutive lanes that share one neighbour row; every lane of the group
// gets the sum.  DPP inside a 16-lane row, shuffles above.
template <int LPR>
__device__ inline int group_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);                  // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);                  // quad_perm [2,3,0,1]
  if (LPR >= 8) x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);   // row_half_mirror
  if (LPR >= 16) x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);  // row_mirror
  if (LPR >= 32) x += __shfl_xor(x, 16);
  if (LPR >= 64) x += __shfl_xor(x, 32);
  return x;
}


}  // namespace
