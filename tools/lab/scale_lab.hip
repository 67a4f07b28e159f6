// scale_lab.hip — what costs the GF-ICF scaling pass its bandwidth?  (tools only)
// Same structure as k_scale_cells_lds (persistent 1024-thread workgroup per CU, gene tables in LDS, wave per cell,
// cell in registers, two wave reductions, compacted write), with parts switched off by template flags:
//   DIV    : (x / S) * w   vs  x * (1/S) * w
//   LOOKUP : remap / weight lookups in LDS  vs  entry kept iff (g & 1), weight 1
//   COMPACT: stream-compacted output  vs  every entry written in place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int CH = 24;
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
template <bool DIV, bool LOOKUP, bool COMPACT, bool NT, bool NTS = true, int THREADS = 1024, bool BITS = false, int CHN = CH>
__global__ __launch_bounds__(THREADS) void k_scale(long G, long n_cells, const long* __restrict__ colptr, const int* __restrict__ rowidx,
                                                const double* __restrict__ x, const unsigned short* __restrict__ remap16,
                                                const double* __restrict__ wk, long gkept, const long* __restrict__ ocp,
                                                int* __restrict__ ori, double* __restrict__ ox) {
  extern __shared__ unsigned char s_raw[];
  unsigned short* const s_remap = (unsigned short*)s_raw;
  double* const s_w = (double*)(s_raw + (((size_t)G * 2 + 15) & ~(size_t)15));
  uint2* const s_bits = (uint2*)s_raw;                  // BITS: per 32 genes {keep bits, kept genes before}
  double* const s_w2 = (double*)(s_raw + ((((size_t)(G + 31) / 32) * 8 + 15) & ~(size_t)15));
  if (LOOKUP && !BITS) {
  for (long t = threadIdx.x; t < (G + 1) / 2; t += THREADS) ((unsigned*)s_remap)[t] = ((const unsigned*)remap16)[t];
  for (long t = threadIdx.x; t < gkept; t += THREADS) s_w[t] = wk[t];
  }
  if (LOOKUP && BITS) {
    for (long t = threadIdx.x; t < (G + 31) / 32; t += THREADS) {
      unsigned b = 0, first = 0xFFFFFFFFu;
      for (int i = 0; i < 32; ++i) { const long g = t * 32 + i; if (g < G && remap16[g] != 0xFFFF) { b |= 1u << i; if (first == 0xFFFFFFFFu) first = remap16[g]; } }
      // kept genes before this word = remap of its first kept gene (or anything when the word is empty)
      s_bits[t] = make_uint2(b, first == 0xFFFFFFFFu ? 0u : first);
    }
    for (long t = threadIdx.x; t < gkept; t += THREADS) s_w2[t] = wk[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const long w0 = ((long)blockIdx.x * THREADS + threadIdx.x) >> 6, nw = ((long)gridDim.x * THREADS) >> 6;
  for (long c = w0; c < n_cells; c += nw) {
    const long p0 = colptr[c], p1 = colptr[c + 1];
    const int n_it = (int)((p1 - p0 + 63) >> 6);
    long opos = COMPACT ? ocp[c] : p0;
    double xv[CHN]; int rv[CHN];
#pragma unroll
    for (int m = 0; m < CHN; ++m) {
      rv[m] = -1; xv[m] = 0.0;
      if (m < n_it) { const long p = p0 + m * 64 + lane; if (p < p1) { rv[m] = NT ? __builtin_nontemporal_load(rowidx + p) : rowidx[p]; xv[m] = NT ? __builtin_nontemporal_load(x + p) : x[p]; } }
    }
    double S = 0.0;
#pragma unroll
    for (int m = 0; m < CHN; ++m) if (m < n_it) {
      const unsigned g = (unsigned)rv[m];
      unsigned r;
      if (LOOKUP && BITS) { const uint2 e = s_bits[(g < (unsigned)G ? g : 0u) >> 5]; const unsigned bit = 1u << (g & 31); r = (g < (unsigned)G && (e.x & bit)) ? e.y + __popc(e.x & (bit - 1u)) : 0xFFFFu; } else if (LOOKUP) r = g < (unsigned)G ? (unsigned)s_remap[g] : 0xFFFFu; else r = (g < (unsigned)G && (g & 1u)) ? (g >> 1) : 0xFFFFu;
      rv[m] = r == 0xFFFFu ? -1 : (int)r;
      if (rv[m] >= 0) S += xv[m];
    }
    const double Sc = wave_sum(S);
    const double rS = 1.0 / Sc;
    double q = 0.0;
#pragma unroll
    for (int m = 0; m < CHN; ++m) if (m < n_it) {
      double v = 0.0;
      if (rv[m] >= 0 && Sc != 0.0) v = (DIV ? xv[m] / Sc : xv[m] * rS) * (LOOKUP ? (BITS ? s_w2[rv[m]] : s_w[rv[m]]) : 1.0);
      xv[m] = v; q += v * v;
    }
    const double qc = wave_sum(q);
    double nv = 1.0 / sqrt(qc); if (isinf(nv)) nv = 0.0;
#pragma unroll
    for (int m = 0; m < CHN; ++m) if (m < n_it) {
      if (COMPACT) {
        const bool kp = rv[m] >= 0;
        const unsigned long long mk = __ballot(kp);
        if (kp) { const long dst = opos + __popcll(mk & lt); if (NTS) { __builtin_nontemporal_store(rv[m], ori + dst); __builtin_nontemporal_store(nv * xv[m], ox + dst); } else { ori[dst] = rv[m]; ox[dst] = nv * xv[m]; } }
        opos += __popcll(mk);
      } else {
        const long p = p0 + m * 64 + lane;
        if (p < p1) { if (NTS) { __builtin_nontemporal_store(rv[m], ori + p); __builtin_nontemporal_store(nv * xv[m], ox + p); } else { ori[p] = rv[m]; ox[p] = nv * xv[m]; } }
      }
    }
  }
}
int main() {
  const long ncol = 54000, G = 23000; const int len = 1107; const long n = ncol * len;
  std::vector<long> cp(ncol + 1), ocp(ncol + 1); std::vector<int> ri(n); std::vector<unsigned short> rm(G + 8, 0xFFFF); std::vector<double> w(G, 0.5);
  long gk = 0; for (long g = 0; g < G; ++g) if (g % 5 == 0) rm[g] = (unsigned short)gk++;        // 20 % of the genes kept ...
  unsigned s = 12345; long kept = 0;
  for (long c = 0; c <= ncol; ++c) cp[c] = c * len;
  for (long c = 0; c < ncol; ++c) { ocp[c] = kept; for (int e = 0; e < len; ++e) { s = s * 1664525u + 1013904223u; int g = (s >> 8) % 100 < 63 ? (int)(((s >> 9) % (G / 5)) * 5) : (int)((s >> 9) % G); ri[c * len + e] = g; kept += rm[g] != 0xFFFF; } }   // ... holding ~63 % of the entries
  ocp[ncol] = kept;
  long *d_cp, *d_ocp; int *d_ri, *d_ori; double *d_x, *d_ox, *d_w; unsigned short* d_rm;
  CK(hipMalloc(&d_cp, (ncol + 1) * 8)); CK(hipMalloc(&d_ocp, (ncol + 1) * 8)); CK(hipMalloc(&d_ri, n * 4)); CK(hipMalloc(&d_ori, n * 4));
  CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_ox, n * 8)); CK(hipMalloc(&d_w, G * 8)); CK(hipMalloc(&d_rm, (G + 8) * 2));
  CK(hipMemcpy(d_cp, cp.data(), (ncol + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ocp, ocp.data(), (ncol + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ri, ri.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rm, rm.data(), (G + 8) * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, w.data(), G * 8, hipMemcpyHostToDevice));
  { std::vector<double> hx(n, 1.0); CK(hipMemcpy(d_x, hx.data(), n * 8, hipMemcpyHostToDevice)); }
  const size_t lds = 156 * 1024;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("kept fraction %.3f\n", (double)kept / n);
  auto run = [&](const char* name, auto kern, double out_frac, int grid = 256, int threads = 1024, size_t l = 156 * 1024) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), l, 0, G, ncol, d_cp, d_ri, d_x, d_rm, d_w, gk, d_ocp, d_ori, d_ox);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-52s %7.1f us  %.2f TB/s\n", name, best * 1e3, (n * 12.0 + n * 12.0 * out_frac) / (best * 1e-3) / 1e12);
  };
  const double kf = (double)kept / n;
  run("full (div, lookup, compact, nt loads)", k_scale<true, true, true, true>, kf);
  run("plain loads", k_scale<true, true, true, false>, kf);
  run("no division", k_scale<false, true, true, true>, kf);
  run("no LDS lookups", k_scale<true, false, true, true>, kf);
  run("no compaction (writes everything)", k_scale<true, true, false, true>, 1.0);
  run("none of the three", k_scale<false, false, false, true>, 1.0);
  run("none, plain stores", (k_scale<false, false, false, true, false>), 1.0);
  run("none, plain loads+stores", (k_scale<false, false, false, false, false>), 1.0);
  run("none, plain ld/st, 256 thr x 2048 blocks, no LDS", (k_scale<false, false, false, false, false, 256>), 1.0, 2048, 256, 0);
  run("none, nt ld/st, 256 thr x 2048 blocks, no LDS", (k_scale<false, false, false, true, true, 256>), 1.0, 2048, 256, 0);
  run("full, plain stores", (k_scale<true, true, true, true, false>), kf);
  run("no div, plain stores", (k_scale<false, true, true, true, false>), kf);
  run("full but 2 WG x 512 thr (78 KB LDS each)", (k_scale<true, true, true, true, true, 512>), kf, 512, 512, 78 * 1024);
  run("bitmask+rank lookup, 1 WG x 1024", (k_scale<true, true, true, true, true, 1024, true>), kf, 256, 1024, 156 * 1024);
  run("bitmask+rank, 3 WG x 512 thr (52 KB)", (k_scale<true, true, true, true, true, 512, true>), kf, 768, 512, 52 * 1024);
  run("bitmask+rank, 3 WG x 512 thr (52 KB), 20 chunks", (k_scale<true, true, true, true, true, 512, true, 20>), kf, 768, 512, 52 * 1024);
  run("bitmask+rank, 6 WG x 256 thr (26 KB?)", (k_scale<true, true, true, true, true, 256, true, 20>), kf, 1536, 256, 48 * 1024);
  run("bitmask+rank, 2 WG x 1024 thr (78 KB), 20 chunks", (k_scale<true, true, true, true, true, 1024, true, 20>), kf, 512, 1024, 78 * 1024);
  return 0;
}
