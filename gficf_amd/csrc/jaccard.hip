// jaccard.hip — Phenograph kNN -> Jaccard edge build for gfx950 (MI355X).
//
// Replaces the reference hot loop JCoefficient::operator()
// (reference src/rcpp_parallel_jaccard_coeff.cpp:24-55): for every cell i and neighbour
// slot j,  u = |multiset(row i) ∩ multiset(row idx[i,j])|  and the edge row
// (i+1, idx[i,j], u/(2k-u)), zero when u == 0.
//
// This is integer set work bounded by memory, not a contraction: no MFMA.  Design:
//   * ingest  : the R matrix (column-major, int32 or double) is transposed once into a
//               row-major int32 table with rows padded to KPAD in {16,32,64,128,256}
//               entries, so that one neighbour row is one (or a few) contiguous 64..1024 B
//               reads.  Ids are validated here; bit 31 of a row's first entry flags a
//               row that holds duplicate ids (never the case for real kNN output).
//   * edges   : one wave64 per cell.  Row i is staged in LDS as a 2-slot-bucket hash set
//               (one ds_read_b64 per probe, no probing loop); 256/KPAD neighbour rows are
//               gathered per wave-instruction (16 B per lane), every lane probes the set with
//               its ids, and the per-edge intersection count is a DPP sum over the row's lanes.
//               Weights come from a per-block LDS table W[u] = u/(2k-u) computed in IEEE
//               double, so they are bit-identical to the reference's division.
//   * rows with duplicate ids (multiset semantics), hash overflow: exact slow path in the
//               same kernel (all-pairs with occurrence ranks).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace {

constexpr uint32_t ROW_DUP_FLAG = 0x80000000u;
constexpr uint32_t ID_MASK = 0x7FFFFFFFu;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

__host__ __device__ inline int kpad_for(int k) {
  return k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : k <= 128 ? 128 : 256;
}

// ------------------------------------------------------------------------------ ingest
template <typename T>
__device__ inline uint32_t decode_id(T raw, int64_t N, bool& ok);
template <>
__device__ inline uint32_t decode_id<int32_t>(int32_t raw, int64_t N, bool& ok) {
  ok = raw >= 1 && (int64_t)raw <= N;
  return ok ? (uint32_t)raw : 0u;
}
template <>
__device__ inline uint32_t decode_id<double>(double raw, int64_t N, bool& ok) {
  // reference: int k = mat(i,j) - 1  (:28) — only integer-valued ids are meaningful.
  ok = raw >= 1.0 && raw <= (double)N && raw == trunc(raw);
  return ok ? (uint32_t)raw : 0u;
}

constexpr int INGEST_ROWS = 64;

// Tile transpose: 64 cells x KPAD slots per step.  Reads are coalesced along cells
// (column-major input), writes are one contiguous 64*KPAD*4 B run of the table.
template <typename T, int KPAD>
__global__ __launch_bounds__(256) void k_ingest(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld,
                                                int64_t N_total, uint32_t* __restrict__ table,
                                                uint32_t* __restrict__ status) {
  __shared__ uint32_t tile[INGEST_ROWS][KPAD + 1];
  __shared__ uint32_t dup[INGEST_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row0 = (int64_t)blockIdx.x * INGEST_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * INGEST_ROWS) {
    const int64_t r = row0 + lane;
    bool bad = false;
    for (int j = wave; j < KPAD; j += 4) {
      uint32_t v = 0;
      if (j < k && r < n_rows) {
        bool ok;
        v = decode_id<T>(idx[(int64_t)j * ld + r], N_total, ok);
        bad |= !ok;
      }
      tile[lane][j] = v;
    }
    if (tid < INGEST_ROWS) dup[tid] = 0;
    if (bad) atomicOr(status, GFICF_ST_BAD_ID);
    __syncthreads();
    // duplicate ids inside a row (multiset case): thread (row = lane, part = wave)
    bool d = false;
    for (int j = wave; j < k; j += 4) {
      const uint32_t a = tile[lane][j];
      if (a != 0)
        for (int j2 = 0; j2 < j; ++j2) d |= (tile[lane][j2] == a);
    }
    if (d) dup[lane] = 1;
    __syncthreads();
    const int64_t rows_here = (n_rows - row0) < INGEST_ROWS ? (n_rows - row0) : INGEST_ROWS;
    const int n_out = (int)rows_here * KPAD;
    for (int e = tid; e < n_out; e += 256) {
      const int rr = e / KPAD, j = e % KPAD;
      uint32_t v = tile[rr][j];
      if (j == 0 && dup[rr]) v |= ROW_DUP_FLAG;
      table[row0 * KPAD + e] = v;
    }
    __syncthreads();
  }
}

// Register variant for KPAD <= 64: one thread per cell.  The k loads of a thread are independent
// (all in flight at once) and each is a coalesced 256 B run per wave; duplicate detection is an
// all-pairs compare in registers; the tile goes through LDS once so that the table is written as
// contiguous 16 B-per-lane runs.
constexpr int INGEST2_ROWS = 64;

template <typename T, int KPAD>
__global__ __launch_bounds__(INGEST2_ROWS) void k_ingest_reg(const T* __restrict__ idx, int64_t n_rows, int k, int64_t ld,
                                                             int64_t N_total, uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ status) {
  __shared__ uint32_t tile[INGEST2_ROWS][KPAD + 1];
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
        v[j] = decode_id<T>(idx[(int64_t)j * ld + r], N_total, ok);
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
    if (dup) v[0] |= ROW_DUP_FLAG;
#pragma unroll
    for (int j = 0; j < KPAD; ++j) tile[tid][j] = v[j];
    __syncthreads();
    const int64_t rows_here = (n_rows - row0) < INGEST2_ROWS ? (n_rows - row0) : INGEST2_ROWS;
    const int n_out4 = (int)rows_here * (KPAD / 4);
    uint4* const out4 = reinterpret_cast<uint4*>(table + row0 * KPAD);
    for (int e = tid; e < n_out4; e += INGEST2_ROWS) {
      const int rr = e / (KPAD / 4), jj = (e % (KPAD / 4)) * 4;
      out4[e] = make_uint4(tile[rr][jj], tile[rr][jj + 1], tile[rr][jj + 2], tile[rr][jj + 3]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------- edges
template <int KPAD>
struct JCfg {
  static constexpr int LPR = KPAD / 4;                        // lanes per neighbour row (16 B per lane)
  static constexpr int RPS = 64 / LPR;                        // neighbour rows per wave-instruction ("step")
  static constexpr int EPL = KPAD > 64 ? KPAD / 64 : 1;       // registers holding row i (slot s -> reg s/64, lane s%64)
  static constexpr int SPQ = (KPAD < 64 ? KPAD : 64) / RPS;   // steps per register of row i
  static constexpr int NB = 8 * KPAD;                         // 2-slot buckets in the hash set
  static constexpr int WAVES = KPAD <= 128 ? 4 : 2;           // waves per workgroup
  static constexpr int U = SPQ < 4 ? SPQ : 4;                 // steps whose gathers are in flight together
  static constexpr int LOG2NB = KPAD == 16 ? 7 : KPAD == 32 ? 8 : KPAD == 64 ? 9 : KPAD == 128 ? 10 : 11;
};

// Byte offset of an id's bucket inside a wave's hash set: bits [3, 3+LOG2NB) of id*K, i.e. a
// multiplicative hash of the id's low 3+LOG2NB bits.  BIG == false: ids < 2^24, full-rate 24-bit
// multiply (bound to the intrinsic by name: written as a plain product the masked multiply is
// canonicalised to the quarter-rate v_mul_lo_u32).
extern "C" __device__ uint32_t gficf_mul_u24(uint32_t a, uint32_t b) __asm("llvm.amdgcn.mul.u24.i32");

template <int KPAD, bool BIG>
__device__ inline uint32_t bucket_off(uint32_t id) {
  constexpr uint32_t HMASK = (uint32_t)(JCfg<KPAD>::NB - 1) << 3;
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

struct EdgeOut {
  double* src;      // the three columns of the reference's edge matrix (all NULL: counts only)
  double* dst;
  double* w;
  int32_t* u;       // optional intersection counts
  uint16_t* u16;    // optional intersection counts, compact (input of the edge filter)
  int set_mode;     // rows with duplicate ids: 0 = multiset intersection (std::set_intersection of the parallel entry),
                    // 1 = set intersection (Rcpp::intersect of the serial jaccard_coeff entry)
};

__device__ inline void store_edge(const EdgeOut o, int64_t r, int64_t cell, uint32_t dst, int u,
                                  const double* lut) {
  const bool pos = u > 0;
  // written once, never re-read by this kernel: non-temporal, so the table rows keep the L2
  if (o.src) {
    __builtin_nontemporal_store(pos ? (double)(uint32_t)(cell + 1) : 0.0, o.src + r);   // reference :49 (cell + 1 <= 2^31)
    __builtin_nontemporal_store(pos ? (double)dst : 0.0, o.dst + r);                    // reference :50
    __builtin_nontemporal_store(pos ? lut[u] : 0.0, o.w + r);                           // reference :51
  }
  if (o.u) __builtin_nontemporal_store(u, o.u + r);
  if (o.u16) o.u16[r] = (uint16_t)u;
}

// Exact multiset path for one cell whose own row or one of whose neighbour rows holds
// duplicate ids (never the case for real kNN output).
// u = sum over distinct values of min(multiplicity in A, multiplicity in B), evaluated as
// "element e of B counts iff its occurrence rank within B is below the value's multiplicity in A".
template <int KPAD>
__device__ __noinline__ void slow_cell(const uint32_t* __restrict__ table, int64_t i, int k, int64_t out_base,
                                       uint32_t* sA, uint32_t* sB, int lane, double* o_src, double* o_dst,
                                       double* o_w, int32_t* o_u, uint16_t* o_u16, int set_mode, const double* lut) {
  const EdgeOut o{o_src, o_dst, o_w, o_u, o_u16, set_mode};
  for (int e = lane; e < KPAD; e += 64) sA[e] = table[i * KPAD + e] & ID_MASK;
  wave_lds_fence();
  for (int s = 0; s < k; ++s) {
    const uint32_t dst = sA[s];
    int u = 0;
    if (dst != 0) {
      for (int e = lane; e < KPAD; e += 64) sB[e] = table[(int64_t)(dst - 1) * KPAD + e] & ID_MASK;
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
    if (lane == 0) store_edge(o, out_base + s, i, dst, u, lut);
  }
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

// One wave per cell, cells strided over all waves of the grid.  Per cell:
//   * row i (one id per lane) is inserted into the wave's LDS hash set; keys that find both
//     slots of their bucket taken go to a small per-wave overflow list;
//   * "steps": each lane loads 16 B (4 ids) of a neighbour row, so KPAD/4 lanes cover one row
//     and a wave-instruction gathers RPS = 256/KPAD rows; U steps are in flight together;
//   * every lane probes the set with its 4 ids (one ds_read_b64 per id), the per-row
//     intersection count is a DPP sum over the row's lanes;
//   * counts are permuted back to one-slot-per-lane and stored as three coalesced runs.
// The load of the next cell's own row is issued ahead of the gathers and the stores of the
// previous cell's edges behind them, so neither sits on the wait for the gathers.
template <int KPAD, bool BIG>
__global__ __launch_bounds__(JCfg<KPAD>::WAVES * 64) void k_jaccard_edges(
    const uint32_t* __restrict__ table, int64_t N, int k, int64_t cell_begin, int64_t cell_end, EdgeOut o) {
  using C = JCfg<KPAD>;
  using off_t = typename std::conditional<BIG, uint64_t, uint32_t>::type;
  // LDS (dynamic, laid out here so that a wave's hash set starts at a multiple of its size and a probe
  // address is (hash & mask) | wave_base):  hash sets | overflow list / slow-path rows | weight table
  extern __shared__ unsigned char smem[];
  constexpr uint32_t HBYTES = C::NB * 8;                      // bytes of one wave's hash set
  uint32_t(*const s_rows)[2][KPAD] = reinterpret_cast<uint32_t(*)[2][KPAD]>(smem + C::WAVES * HBYTES);
  double* const s_lut = reinterpret_cast<double*>(smem + C::WAVES * HBYTES + C::WAVES * 2 * KPAD * 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* const hbase = smem + wave * HBYTES;          // this wave's hash set
  // W[u] = u / (2.0*k - u): same IEEE-754 double division as reference :51
  for (int u = tid; u <= k; u += C::WAVES * 64) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);
  for (int b = lane; b < C::NB; b += 64) reinterpret_cast<uint2*>(hbase)[b] = make_uint2(EMPTY, EMPTY);
  __syncthreads();

  // LDS byte address of this wave's hash set (a multiple of HBYTES: dynamic LDS starts at 0 here,
  // there is no static LDS in this kernel), OR-ed with a bucket offset per probe
  const uint32_t wave_off = lds_address(smem) + (uint32_t)wave * HBYTES;
  uint32_t* const ovlist = s_rows[wave][0];
  const char* const tbytes = reinterpret_cast<const char*>(table);
  const bool arow_lane = KPAD >= 64 || lane < KPAD;         // lanes that hold an id of row i
  const int grow = lane / C::LPR;                           // which of the RPS rows of a step this lane reads
  const uint32_t gcol = (uint32_t)(lane % C::LPR) * 16u;    // byte offset of this lane's 4 ids inside a row
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t nwaves = (int64_t)gridDim.x * C::WAVES;
  constexpr uint32_t ROWB = KPAD * 4;

  int64_t i = cell_begin + (int64_t)blockIdx.x * C::WAVES + wave;
  uint32_t araw[C::EPL];
#pragma unroll
  for (int q = 0; q < C::EPL; ++q) araw[q] = 0;
  if (i < cell_end && arow_lane) {
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) araw[q] = table[i * KPAD + q * 64 + lane];
  }
  // edges of the previous cell, stored while the current cell's gathers are in flight
  bool have_prev = false;
  int64_t prev_i = 0;
  uint32_t prev_a[C::EPL];
  int prev_u[C::EPL];

  for (; i < cell_end; i += nwaves) {
    uint32_t a[C::EPL], asafe[C::EPL];
    uint32_t flags = 0;
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) {
      flags |= araw[q];
      a[q] = araw[q] & ID_MASK;
      // a slot without a usable id (padding, rejected id) gathers the cell's own row instead; its
      // count is discarded at the store
      asafe[q] = a[q] != 0 ? a[q] : (uint32_t)(i + 1);
    }
    bool slow = __ballot((flags & ROW_DUP_FLAG) != 0) != 0ull;
    // next cell's own row: ahead of the gathers, so that it has landed by the next iteration
    const int64_t i_next = i + nwaves;
    if (i_next < cell_end && arow_lane) {
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) araw[q] = table[i_next * KPAD + q * 64 + lane];
    }
    int myslot[C::EPL];
    int nov = 0;
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) myslot[q] = -1;
    if (!slow) {
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) {
        bool over = false;
        if (a[q] != 0) {
          const uint32_t bo = bucket_off<KPAD, BIG>(a[q]) + (uint32_t)wave * HBYTES;   // byte offset of the bucket in smem
          uint32_t old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo), EMPTY, a[q]);
          if (old == EMPTY) {
            myslot[q] = (int)bo;
          } else {
            old = atomicCAS(reinterpret_cast<uint32_t*>(smem + bo + 4), EMPTY, a[q]);
            if (old == EMPTY) myslot[q] = (int)bo + 4;
            else over = true;
          }
        }
        const unsigned long long om = __ballot(over);
        if (om) {
          if (over) ovlist[nov + __popcll(om & lt_mask)] = a[q];
          nov += __popcll(om);
        }
      }
      wave_lds_fence();
    }

    int myu[C::EPL];
#pragma unroll
    for (int q = 0; q < C::EPL; ++q) myu[q] = 0;
    bool prev_stored = false;

    if (!slow) {
      uint32_t dupflags = 0;
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) {        // q: which register of row i holds the slots of these steps
        for (int t0 = 0; t0 < C::SPQ && (q * 64 + t0 * C::RPS) < k; t0 += C::U) {
          uint4 bv[C::U];
          // issue the gathers of U steps (U*RPS neighbour rows) before consuming any
#pragma unroll
          for (int uu = 0; uu < C::U; ++uu) {
            const uint32_t dst = (uint32_t)__shfl((int)asafe[q], (t0 + uu) * C::RPS + grow);
            const off_t off = (off_t)(dst - 1) * ROWB + gcol;
            bv[uu] = *reinterpret_cast<const uint4*>(tbytes + off);
          }
          if (!prev_stored) {
            // the previous cell's edges ride behind the gathers (younger in vmcnt order, so the
            // wait for the gathers does not wait for them)
            prev_stored = true;
            if (have_prev) {
              const int64_t pb = (prev_i - cell_begin) * (int64_t)k;
#pragma unroll
              for (int qq = 0; qq < C::EPL; ++qq) {
                const int slot = qq * 64 + lane;
                if (arow_lane && slot < k) store_edge(o, pb + slot, prev_i, prev_a[qq], prev_u[qq], s_lut);
              }
              have_prev = false;
            }
          }
          int cnt[C::U];
#pragma unroll
          for (int uu = 0; uu < C::U; ++uu) {
            dupflags |= bv[uu].x;
            bv[uu].x &= ID_MASK;             // only a row's first id can carry the duplicate flag
            const uint2 h0 = lds_read_b64(bucket_off<KPAD, BIG>(bv[uu].x) | wave_off);
            const uint2 h1 = lds_read_b64(bucket_off<KPAD, BIG>(bv[uu].y) | wave_off);
            const uint2 h2 = lds_read_b64(bucket_off<KPAD, BIG>(bv[uu].z) | wave_off);
            const uint2 h3 = lds_read_b64(bucket_off<KPAD, BIG>(bv[uu].w) | wave_off);
            int c = 0;
            c = add_lane_bit(c, __ballot(h0.x == bv[uu].x) | __ballot(h0.y == bv[uu].x));
            c = add_lane_bit(c, __ballot(h1.x == bv[uu].y) | __ballot(h1.y == bv[uu].y));
            c = add_lane_bit(c, __ballot(h2.x == bv[uu].z) | __ballot(h2.y == bv[uu].z));
            c = add_lane_bit(c, __ballot(h3.x == bv[uu].w) | __ballot(h3.y == bv[uu].w));
            cnt[uu] = c;
          }
          if (nov) {                          // wave-uniform, rare: ids that overflowed the set
            for (int t = 0; t < nov; ++t) {
              const uint32_t ov = ovlist[t];
#pragma unroll
              for (int uu = 0; uu < C::U; ++uu)
                cnt[uu] += (bv[uu].x == ov) + (bv[uu].y == ov) + (bv[uu].z == ov) + (bv[uu].w == ov);
            }
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
      // a neighbour row with duplicate ids: redo this cell exactly
      slow = __ballot((dupflags & ROW_DUP_FLAG) != 0) != 0ull;
    }
    if (!prev_stored && have_prev) {    // own row with duplicates: the gather loop was skipped
      const int64_t pb = (prev_i - cell_begin) * (int64_t)k;
#pragma unroll
      for (int qq = 0; qq < C::EPL; ++qq) {
        const int slot = qq * 64 + lane;
        if (arow_lane && slot < k) store_edge(o, pb + slot, prev_i, prev_a[qq], prev_u[qq], s_lut);
      }
      have_prev = false;
    }
    // ---- clear this cell's keys from the set
#pragma unroll
    for (int q = 0; q < C::EPL; ++q)
      if (myslot[q] >= 0) *reinterpret_cast<uint32_t*>(smem + myslot[q]) = EMPTY;
    wave_lds_fence();
    if (slow) {
      slow_cell<KPAD>(table, i, k, (i - cell_begin) * (int64_t)k, s_rows[wave][0], s_rows[wave][1], lane, o.src, o.dst, o.w,
                      o.u, o.u16, o.set_mode, s_lut);
    } else {
      have_prev = true;
      prev_i = i;
#pragma unroll
      for (int q = 0; q < C::EPL; ++q) {
        prev_a[q] = a[q];
        prev_u[q] = a[q] != 0 ? myu[q] : 0;      // rejected id: zero row
      }
    }
  }
  if (have_prev) {
    const int64_t pb = (prev_i - cell_begin) * (int64_t)k;
#pragma unroll
    for (int qq = 0; qq < C::EPL; ++qq) {
      const int slot = qq * 64 + lane;
      if (arow_lane && slot < k) store_edge(o, pb + slot, prev_i, prev_a[qq], prev_u[qq], s_lut);
    }
  }
}

// ------------------------------------------------------------------ edge filter (N1)
// The caller's next line, relations[relations[,3] > 0, ] (reference R/clustCells.R:66), on the
// device: per-cell count of edges with u > 0 -> exclusive scan -> ordered compacted write.
__global__ __launch_bounds__(256) void k_edge_kept_count(const uint16_t* __restrict__ u16, int64_t n_cells, int k,
                                                         int64_t* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c == 0) out[n_cells] = 0;
  if (c >= n_cells) return;
  int n = 0;
  for (int s = 0; s < k; ++s) n += u16[c * k + s] != 0;
  out[c] = n;
}

template <int KPAD>
__global__ __launch_bounds__(256) void k_edge_write(const uint32_t* __restrict__ table, const uint16_t* __restrict__ u16,
                                                    int k, int64_t cell_begin, int64_t n_cells,
                                                    const int64_t* __restrict__ ptr, double* __restrict__ from,
                                                    double* __restrict__ to, double* __restrict__ weight) {
  __shared__ double s_lut[GFICF_JACCARD_MAX_K + 1];
  for (int u = threadIdx.x; u <= k; u += 256) s_lut[u] = (double)u / (2.0 * (double)k - (double)u);   // reference :51
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t c = w0; c < n_cells; c += nw) {
    int64_t pos = ptr[c];
    for (int s0 = 0; s0 < k; s0 += 64) {
      const int s = s0 + lane;
      int u = 0;
      uint32_t dst = 0;
      if (s < k) {
        u = u16[c * k + s];
        dst = table[(cell_begin + c) * KPAD + s] & ID_MASK;
      }
      const bool kp = u > 0;
      const unsigned long long m = __ballot(kp);
      if (kp) {
        const int64_t d = pos + __popcll(m & lt_mask);
        from[d] = (double)(uint32_t)(cell_begin + c + 1);
        to[d] = (double)dst;
        weight[d] = s_lut[u];
      }
      pos += __popcll(m);
    }
  }
}

// ------------------------------------------------- packed table rows (multi-GPU transport)
// The all-gather of table rows is what bounds the N > 1 path (xGMI), so rows travel bit-packed:
// k ids of ceil(log2(N+1)) bits each + 1 bit for the row's duplicate flag, rounded up to whole
// 32-bit words (k = 30, N = 800 k: 76 B instead of the 128 B row pitch).  One thread per row.
__host__ __device__ inline int id_bits(int64_t N) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) <= N) ++b;
  return b;
}
__host__ __device__ inline int packed_words(int64_t N, int k) { return (k * id_bits(N) + 1 + 31) / 32; }

// 64 rows per workgroup, staged through LDS so that both the table read and the packed write are
// contiguous runs; every thread assembles whole 32-bit output words.
constexpr int PACK_ROWS = 64;

__global__ __launch_bounds__(256) void k_pack_rows(const uint32_t* __restrict__ table, int64_t n_rows, int k, int kpad,
                                                   int bits, int wpr, uint32_t* __restrict__ packed) {
  extern __shared__ uint32_t s_tb[];                        // PACK_ROWS * kpad words
  const int fl = k * bits;                                  // bit position of the duplicate flag
  for (int64_t row0 = (int64_t)blockIdx.x * PACK_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * PACK_ROWS) {
    const int rows_here = (int)((n_rows - row0) < PACK_ROWS ? (n_rows - row0) : PACK_ROWS);
    for (int e = threadIdx.x; e < rows_here * kpad; e += 256) s_tb[e] = table[row0 * kpad + e];
    __syncthreads();
    for (int e = threadIdx.x; e < rows_here * wpr; e += 256) {
      const int rr = e / wpr, w = e % wpr;
      const uint32_t* row = s_tb + rr * kpad;
      const int lo = 32 * w, hi = lo + 32;
      uint32_t v = 0;
      for (int j = lo / bits; j < k && j * bits < hi; ++j) {
        const uint32_t id = row[j] & ID_MASK;
        const int pos = j * bits - lo;
        v |= pos >= 0 ? id << pos : id >> (-pos);
      }
      if (fl >= lo && fl < hi) v |= (row[0] >> 31) << (fl - lo);
      packed[row0 * wpr + e] = v;
    }
    __syncthreads();
  }
}

// 64 rows per workgroup: the packed words are staged in LDS with coalesced loads, every thread then
// extracts ids for consecutive slots, so the table is written as contiguous runs.
constexpr int UNPACK_ROWS = 64;

__global__ __launch_bounds__(256) void k_unpack_rows(const uint32_t* __restrict__ packed, int64_t n_rows, int k, int kpad,
                                                     int bits, int wpr, uint32_t* __restrict__ table) {
  extern __shared__ uint32_t s_pk[];                        // UNPACK_ROWS * wpr words
  const uint32_t mask = (uint32_t)(((unsigned long long)1 << bits) - 1ull);
  const int fl_w = (k * bits) >> 5, fl_b = (k * bits) & 31; // position of the duplicate flag
  for (int64_t row0 = (int64_t)blockIdx.x * UNPACK_ROWS; row0 < n_rows; row0 += (int64_t)gridDim.x * UNPACK_ROWS) {
    const int rows_here = (int)((n_rows - row0) < UNPACK_ROWS ? (n_rows - row0) : UNPACK_ROWS);
    for (int e = threadIdx.x; e < rows_here * wpr; e += 256) s_pk[e] = packed[row0 * wpr + e];
    __syncthreads();
    for (int e = threadIdx.x; e < rows_here * kpad; e += 256) {
      const int rr = e / kpad, j = e % kpad;
      uint32_t id = 0;
      if (j < k) {
        const uint32_t* in = s_pk + rr * wpr;
        const int off = j * bits, w = off >> 5, sh = off & 31;
        unsigned long long v = in[w];
        if (w + 1 < wpr) v |= (unsigned long long)in[w + 1] << 32;
        id = (uint32_t)(v >> sh) & mask;
        if (j == 0) id |= ((in[fl_w] >> fl_b) & 1u) << 31;
      }
      table[row0 * kpad + e] = id;
    }
    __syncthreads();
  }
}

template <typename T>
int launch_ingest(gficf_ctx* ctx, const T* d_idx, int64_t n_rows, int k, int64_t ld, int64_t N_total,
                  uint32_t* table) {
  const int kpad = kpad_for(k);
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  const int64_t tiles2 = gficf_ceil_div(n_rows, INGEST2_ROWS);
  const unsigned grid2 = (unsigned)(tiles2 < cap ? tiles2 : cap);
  const int64_t tiles = gficf_ceil_div(n_rows, INGEST_ROWS);
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
#define LAUNCH_INGEST_REG(KP)                                                                                          \
  hipLaunchKernelGGL((k_ingest_reg<T, KP>), dim3(grid2), dim3(INGEST2_ROWS), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, \
                     table, ctx->d_status)
#define LAUNCH_INGEST(KP)                                                                                   \
  hipLaunchKernelGGL((k_ingest<T, KP>), dim3(grid), dim3(256), 0, ctx->stream, d_idx, n_rows, k, ld, N_total, \
                     table, ctx->d_status)
  switch (kpad) {
    case 16: LAUNCH_INGEST_REG(16); break;
    case 32: LAUNCH_INGEST_REG(32); break;
    case 64: LAUNCH_INGEST_REG(64); break;
    case 128: LAUNCH_INGEST(128); break;
    default: LAUNCH_INGEST(256); break;
  }
#undef LAUNCH_INGEST
#undef LAUNCH_INGEST_REG
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int KPAD>
constexpr size_t edges_lds_bytes() {
  using C = JCfg<KPAD>;
  return (size_t)C::WAVES * C::NB * 8 + (size_t)C::WAVES * 2 * KPAD * 4 + (GFICF_JACCARD_MAX_K + 1) * sizeof(double);
}

template <int KPAD, bool BIG>
int launch_edges_t(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  using C = JCfg<KPAD>;
  // grid = what is resident at once (occupancy x CUs); waves stride over the cells
  static int blocks_per_cu = 0;
  if (blocks_per_cu == 0) {
    int nb = 0;
    GFICF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_jaccard_edges<KPAD, BIG>, C::WAVES * 64, edges_lds_bytes<KPAD>()));
    // 3 workgroups per CU already saturate the L2-miss path that bounds this kernel (measured: 3..16
    // per CU run at the same speed, 2 is 20 % slower); not taking every wave slot leaves room for the
    // neighbouring step's edge kernel and the next batch's ingest / all-gather kernels, which the
    // pipelined caller runs concurrently on other streams (measured best there: 3)
    blocks_per_cu = nb > 3 ? 3 : nb > 0 ? nb : 1;
    if (const char* e = getenv("GFICF_JACCARD_BLOCKS_PER_CU")) {   // tuning knob
      const int v = atoi(e);
      if (v > 0) blocks_per_cu = v;
    }
  }
  const int64_t blocks_needed = gficf_ceil_div(ce - cb, C::WAVES);
  const int64_t cap = (int64_t)ctx->num_cus * blocks_per_cu;
  const unsigned grid = (unsigned)(blocks_needed < cap ? blocks_needed : cap);
  hipLaunchKernelGGL((k_jaccard_edges<KPAD, BIG>), dim3(grid), dim3(C::WAVES * 64), edges_lds_bytes<KPAD>(), ctx->stream, table,
                     N, k, cb, ce, o);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

template <int KPAD>
int launch_edges(gficf_ctx* ctx, const uint32_t* table, int64_t N, int k, int64_t cb, int64_t ce, EdgeOut o) {
  // 32-bit byte offsets and the 24-bit hash multiply need table < 4 GiB and ids < 2^24
  static const bool force_big = getenv("GFICF_JACCARD_FORCE_BIG") != nullptr;   // test hook for the 64-bit variant
  const bool big = force_big || N >= (1ll << 24) || N * (int64_t)KPAD * 4 >= (1ll << 32);
  return big ? launch_edges_t<KPAD, true>(ctx, table, N, k, cb, ce, o)
             : launch_edges_t<KPAD, false>(ctx, table, N, k, cb, ce, o);
}

int check_nk(int64_t N, int k) {
  if (N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N = %lld is negative", (long long)N);
  if (k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "k = %d is negative", k);
  if (k > GFICF_JACCARD_MAX_K)
    GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d exceeds GFICF_JACCARD_MAX_K = %d", k, GFICF_JACCARD_MAX_K);
  if (N > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "N = %lld exceeds int32 ids", (long long)N);
  return GFICF_OK;
}

}  // namespace

extern "C" {

int gficf_jaccard_kpad(int k) { return (k < 0 || k > GFICF_JACCARD_MAX_K) ? -1 : kpad_for(k); }

int gficf_jaccard_ingest_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t n_rows, int k,
                                int64_t ld, int64_t N_total, int32_t* d_table_rows) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0 || n_rows > N_total) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_rows = %lld outside [0, N_total]", (long long)n_rows);
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_idx || !d_table_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  if (ld < n_rows) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < n_rows = %lld", (long long)ld, (long long)n_rows);
  if (idx_is_f64) return launch_ingest<double>(ctx, (const double*)d_idx, n_rows, k, ld, N_total, (uint32_t*)d_table_rows);
  return launch_ingest<int32_t>(ctx, (const int32_t*)d_idx, n_rows, k, ld, N_total, (uint32_t*)d_table_rows);
}

int gficf_jaccard_edges_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin,
                               int64_t cell_end, double* d_src, double* d_dst, double* d_w, int32_t* d_u) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (cell_begin < 0 || cell_end < cell_begin || cell_end > N)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "cell range [%lld, %lld) outside [0, %lld]", (long long)cell_begin, (long long)cell_end, (long long)N);
  if (cell_end == cell_begin || k == 0) return GFICF_OK;
  if (!d_table || !d_src || !d_dst || !d_w) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  EdgeOut o{d_src, d_dst, d_w, d_u, nullptr, 0};
  const uint32_t* t = (const uint32_t*)d_table;
  switch (kpad_for(k)) {
    case 16: return launch_edges<16>(ctx, t, N, k, cell_begin, cell_end, o);
    case 32: return launch_edges<32>(ctx, t, N, k, cell_begin, cell_end, o);
    case 64: return launch_edges<64>(ctx, t, N, k, cell_begin, cell_end, o);
    case 128: return launch_edges<128>(ctx, t, N, k, cell_begin, cell_end, o);
    default: return launch_edges<256>(ctx, t, N, k, cell_begin, cell_end, o);
  }
}

int gficf_jaccard_device(gficf_ctx* ctx, const void* d_idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                         int32_t* d_table_ws, double* d_rmat, int32_t* d_u) {
  int rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, d_table_ws);
  if (rc) return rc;
  const int64_t E = N * (int64_t)k;
  return gficf_jaccard_edges_device(ctx, d_table_ws, N, k, 0, N, d_rmat, d_rmat + E, d_rmat + 2 * E, d_u);
}

int gficf_jaccard_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                       double* rmat, int print_output) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (print_output) { printf("Running Parallell Jaccard Coefficient Estimation...\n"); fflush(stdout); }  // reference :63
  const int64_t E = N * (int64_t)k;
  if (E > 0) {
    if (!idx || !rmat) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
    if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
    const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
    const int kpad = kpad_for(k);
    void* d_idx = nullptr;
    int32_t* d_table = nullptr;
    double* d_rmat = nullptr;
    // device scratch comes from the context's grow-only pool: no hipMalloc/hipFree per call
    hipError_t e = gficf_pool_get(ctx, 0, esz * (size_t)ld * (size_t)k, &d_idx);
    if (e == hipSuccess) e = gficf_pool_get(ctx, 1, sizeof(int32_t) * (size_t)N * (size_t)kpad, (void**)&d_table);
    if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(double) * 3 * (size_t)E, (void**)&d_rmat);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
    rc = GFICF_OK;
    if (e == hipSuccess) {
      rc = gficf_jaccard_device(ctx, d_idx, idx_is_f64, N, k, ld, d_table, d_rmat, nullptr);
      if (rc == GFICF_OK) e = hipMemcpyAsync(rmat, d_rmat, sizeof(double) * 3 * (size_t)E, hipMemcpyDeviceToHost, ctx->stream);
      if (rc == GFICF_OK && e == hipSuccess) rc = gficf_ctx_sync(ctx);
      else (void)hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_host: %s", hipGetErrorString(e));
    if (rc) return rc;
  }
  if (print_output) { printf("Done!!\n"); fflush(stdout); }  // reference :77
  return GFICF_OK;
}

int gficf_jaccard_packed_words(int64_t N_total, int k) {
  if (N_total < 0 || N_total > 0x7FFFFFFFll || k < 0 || k > GFICF_JACCARD_MAX_K) return -1;
  return packed_words(N_total, k);
}

int gficf_jaccard_pack_rows_device(gficf_ctx* ctx, const int32_t* d_table_rows, int64_t n_rows, int k, int64_t N_total,
                                   uint32_t* d_packed) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative n_rows");
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_table_rows || !d_packed) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  int64_t blocks = gficf_ceil_div(n_rows, PACK_ROWS);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), (size_t)PACK_ROWS * kpad_for(k) * sizeof(uint32_t), ctx->stream,
                     (const uint32_t*)d_table_rows, n_rows, k, kpad_for(k), id_bits(N_total), packed_words(N_total, k), d_packed);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_unpack_rows_device(gficf_ctx* ctx, const uint32_t* d_packed, int64_t n_rows, int k, int64_t N_total,
                                     int32_t* d_table_rows) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N_total, k);
  if (rc) return rc;
  if (n_rows < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative n_rows");
  if (n_rows == 0 || k == 0) return GFICF_OK;
  if (!d_table_rows || !d_packed) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  const int wpr = packed_words(N_total, k);
  int64_t blocks = gficf_ceil_div(n_rows, UNPACK_ROWS);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)blocks), dim3(256), (size_t)UNPACK_ROWS * wpr * sizeof(uint32_t), ctx->stream,
                     d_packed, n_rows, k, kpad_for(k), id_bits(N_total), wpr, (uint32_t*)d_table_rows);
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

static int edges_filtered(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin, int64_t cell_end,
                          uint16_t* d_u_ws, int64_t* d_cell_ptr, double* d_from, double* d_to, double* d_weight, int set_mode) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (cell_begin < 0 || cell_end < cell_begin || cell_end > N)
    GFICF_FAIL(GFICF_ERR_INVALID_ARG, "cell range [%lld, %lld) outside [0, %lld]", (long long)cell_begin, (long long)cell_end, (long long)N);
  if (!d_cell_ptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  const int64_t n_cells = cell_end - cell_begin;
  if (n_cells == 0 || k == 0) {
    GFICF_HIP_CHECK(hipMemsetAsync(d_cell_ptr, 0, sizeof(int64_t) * (size_t)(n_cells + 1), ctx->stream));
    return GFICF_OK;
  }
  if (!d_table || !d_u_ws || !d_from || !d_to || !d_weight) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer");
  // 1. intersection counts only (no 24 B/edge matrix)
  EdgeOut o{nullptr, nullptr, nullptr, nullptr, d_u_ws, set_mode};
  const uint32_t* t = (const uint32_t*)d_table;
  switch (kpad_for(k)) {
    case 16: rc = launch_edges<16>(ctx, t, N, k, cell_begin, cell_end, o); break;
    case 32: rc = launch_edges<32>(ctx, t, N, k, cell_begin, cell_end, o); break;
    case 64: rc = launch_edges<64>(ctx, t, N, k, cell_begin, cell_end, o); break;
    case 128: rc = launch_edges<128>(ctx, t, N, k, cell_begin, cell_end, o); break;
    default: rc = launch_edges<256>(ctx, t, N, k, cell_begin, cell_end, o); break;
  }
  if (rc) return rc;
  // 2. kept edges per cell -> offsets
  hipLaunchKernelGGL(k_edge_kept_count, dim3((unsigned)gficf_ceil_div(n_cells, 256)), dim3(256), 0, ctx->stream, d_u_ws,
                     n_cells, k, d_cell_ptr);
  GFICF_HIP_CHECK(hipGetLastError());
  rc = gficf_exclusive_scan_i64(ctx, d_cell_ptr, n_cells + 1);
  if (rc) return rc;
  // 3. ordered compacted write
  int64_t blocks = gficf_ceil_div(n_cells, 4);
  if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
#define LAUNCH_EW(KP)                                                                                               \
  hipLaunchKernelGGL((k_edge_write<KP>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, d_u_ws, k, cell_begin, \
                     n_cells, d_cell_ptr, d_from, d_to, d_weight)
  switch (kpad_for(k)) {
    case 16: LAUNCH_EW(16); break;
    case 32: LAUNCH_EW(32); break;
    case 64: LAUNCH_EW(64); break;
    case 128: LAUNCH_EW(128); break;
    default: LAUNCH_EW(256); break;
  }
#undef LAUNCH_EW
  GFICF_HIP_CHECK(hipGetLastError());
  return GFICF_OK;
}

int gficf_jaccard_edges_filtered_device(gficf_ctx* ctx, const int32_t* d_table, int64_t N, int k, int64_t cell_begin,
                                        int64_t cell_end, uint16_t* d_u_ws, int64_t* d_cell_ptr, double* d_from,
                                        double* d_to, double* d_weight) {
  return edges_filtered(ctx, d_table, N, k, cell_begin, cell_end, d_u_ws, d_cell_ptr, d_from, d_to, d_weight, 0);
}

// host form of the filtered build: plan runs everything and returns the edge count, finish copies out
struct gficf_edge_plan {
  int64_t n_edges = 0;
  double* d_from = nullptr;
  double* d_to = nullptr;
  double* d_weight = nullptr;
};

static void edge_plan_free(gficf_ctx* ctx) {
  gficf_edge_plan* p = ctx->edge_plan;
  if (!p) return;
  if (p->d_from) (void)hipFree(p->d_from);
  if (p->d_to) (void)hipFree(p->d_to);
  if (p->d_weight) (void)hipFree(p->d_weight);
  delete p;
  ctx->edge_plan = nullptr;
}

void gficf_edge_plan_free(gficf_ctx* ctx) { edge_plan_free(ctx); }

int gficf_jaccard_filtered_host_plan(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld,
                                     int64_t* n_edges) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (!n_edges) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "n_edges is NULL");
  edge_plan_free(ctx);
  *n_edges = 0;
  const int64_t E = N * (int64_t)k;
  gficf_edge_plan* p = new gficf_edge_plan();
  ctx->edge_plan = p;
  if (E == 0) return GFICF_OK;
  if (!idx) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
  const int kpad = kpad_for(k);
  void* d_idx = nullptr;
  int32_t* d_table = nullptr;
  uint16_t* d_u = nullptr;
  int64_t* d_ptr = nullptr;
  hipError_t e = hipMalloc(&d_idx, esz * (size_t)ld * (size_t)k);
  if (e == hipSuccess) e = hipMalloc((void**)&d_table, sizeof(int32_t) * (size_t)N * (size_t)kpad);
  if (e == hipSuccess) e = hipMalloc((void**)&d_u, sizeof(uint16_t) * (size_t)E);
  if (e == hipSuccess) e = hipMalloc((void**)&d_ptr, sizeof(int64_t) * (size_t)(N + 1));
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_from, sizeof(double) * (size_t)E);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_to, sizeof(double) * (size_t)E);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_weight, sizeof(double) * (size_t)E);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  rc = GFICF_OK;
  int64_t total = 0;
  if (e == hipSuccess) {
    rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, d_table);
    if (!rc) rc = gficf_jaccard_edges_filtered_device(ctx, d_table, N, k, 0, N, d_u, d_ptr, p->d_from, p->d_to, p->d_weight);
    if (!rc) e = hipMemcpyAsync(&total, d_ptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
  }
  if (d_idx) (void)hipFree(d_idx);
  if (d_table) (void)hipFree(d_table);
  if (d_u) (void)hipFree(d_u);
  if (d_ptr) (void)hipFree(d_ptr);
  if (e != hipSuccess) { gficf_set_error("HIP failure in gficf_jaccard_filtered_host_plan: %s", hipGetErrorString(e)); rc = GFICF_ERR_HIP; }
  if (rc) { edge_plan_free(ctx); return rc; }
  p->n_edges = total;
  *n_edges = total;
  return GFICF_OK;
}

/* The package's second Jaccard entry, the serial jaccard_coeff(idx, printOutput) (reference src/jaccard_coeff.cpp:19-44,
 * .Call symbol _gficf_jaccard_coeff, src/RcppExports.cpp:36): same edges, but (a) the intersection is Rcpp::intersect,
 * i.e. of the two rows as SETS (it differs from the parallel entry only for rows with duplicate ids), and (b) the rows
 * with u > 0 are written one after the other from the top of the (N*k) x 3 matrix (`r++`, :36-41), the rest stays 0. */
int gficf_jaccard_coeff_host(gficf_ctx* ctx, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, double* weights,
                             int print_output) {
  GFICF_CTX_ENTER(ctx);
  int rc = check_nk(N, k);
  if (rc) return rc;
  if (print_output) { printf("Running Jaccard Coefficient Estimation...\n"); fflush(stdout); }  // reference :25
  const int64_t E = N * (int64_t)k;
  if (E == 0) return GFICF_OK;
  if (!idx || !weights) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
  if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
  const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
  const int kpad = kpad_for(k);
  void *d_idx = nullptr, *d_table = nullptr, *d_out = nullptr, *d_aux = nullptr;
  hipError_t e = gficf_pool_get(ctx, 0, esz * (size_t)ld * (size_t)k, &d_idx);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 1, sizeof(int32_t) * (size_t)N * (size_t)kpad, &d_table);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 2, sizeof(double) * 3 * (size_t)E, &d_out);
  if (e == hipSuccess) e = gficf_pool_get(ctx, 3, sizeof(uint16_t) * (size_t)E + 64 + sizeof(int64_t) * (size_t)(N + 1), &d_aux);
  if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, esz * (size_t)ld * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  int64_t total = 0;
  if (e == hipSuccess) {
    double* d_from = (double*)d_out;
    int64_t* d_ptr = (int64_t*)d_aux;
    uint16_t* d_u = (uint16_t*)(d_ptr + N + 1);
    rc = gficf_jaccard_ingest_device(ctx, d_idx, idx_is_f64, N, k, ld, N, (int32_t*)d_table);
    if (!rc) rc = edges_filtered(ctx, (const int32_t*)d_table, N, k, 0, N, d_u, d_ptr, d_from, d_from + E, d_from + 2 * E, 1);
    if (!rc) e = hipMemcpyAsync(&total, d_ptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) rc = gficf_ctx_sync(ctx);
    else (void)hipStreamSynchronize(ctx->stream);
    if (!rc && e == hipSuccess) {
      std::memset(weights, 0, sizeof(double) * 3 * (size_t)E);                                  // NumericMatrix weights(nrow*ncol, 3), :21
      for (int c = 0; c < 3 && e == hipSuccess && total > 0; ++c)
        e = hipMemcpyAsync(weights + (size_t)c * (size_t)E, d_from + (size_t)c * (size_t)E, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);
    }
  }
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_coeff_host: %s", hipGetErrorString(e));
  return rc;
}

int gficf_jaccard_filtered_host_finish(gficf_ctx* ctx, double* from, double* to, double* weight) {
  GFICF_CTX_ENTER(ctx);
  gficf_edge_plan* p = ctx->edge_plan;
  if (!p) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_jaccard_filtered_host_finish without a plan");
  hipError_t e = hipSuccess;
  if (p->n_edges > 0) {
    if (!from || !to || !weight) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
    const size_t b = sizeof(double) * (size_t)p->n_edges;
    e = hipMemcpyAsync(from, p->d_from, b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(to, p->d_to, b, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(weight, p->d_weight, b, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
  }
  edge_plan_free(ctx);
  if (e != hipSuccess) GFICF_FAIL(GFICF_ERR_HIP, "HIP failure in gficf_jaccard_filtered_host_finish: %s", hipGetErrorString(e));
  return GFICF_OK;
}

}  // extern "C"
