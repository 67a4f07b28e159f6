// bw_lab.hip — access-pattern micro-benchmarks for the GF-ICF scale pass (tools only).
// Columns of LEN (rowidx int32, x double) entries; measures read (and read+write) GB/s for:
//   A: wave per column, 4 B + 8 B scalar loads per lane, CH chunks in flight
//   B: wave per column, 16 B vector loads per lane (int4 + 2 x double2)
//   C: plain grid-stride streaming with 16 B loads (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CH, bool WRITE>
__global__ __launch_bounds__(256) void k_wave_scalar(const int* __restrict__ ri, const double* __restrict__ x, long ncol, int len,
                                                     int* __restrict__ ori, double* __restrict__ ox, double* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  double acc = 0;
  for (long c = w0; c < ncol; c += nw) {
    const long p0 = c * len, p1 = p0 + len;
    const int nit = (len + 63) >> 6;
    int g[CH]; double v[CH];
#pragma unroll
    for (int m = 0; m < CH; ++m) { g[m] = 0; v[m] = 0; if (m < nit) { long p = p0 + m * 64 + lane; if (p < p1) { g[m] = ri[p]; v[m] = x[p]; } } }
    double s = 0;
#pragma unroll
    for (int m = 0; m < CH; ++m) s += v[m] + g[m];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (WRITE) {
#pragma unroll
      for (int m = 0; m < CH; ++m) if (m < nit) { long p = p0 + m * 64 + lane; if (p < p1) { ori[p] = g[m]; ox[p] = v[m] * s; } }
    }
    acc += s;
  }
  if (acc == 1.2345) sink[0] = acc;
}

template <int CHV, bool WRITE>
__global__ __launch_bounds__(256) void k_wave_vec(const int* __restrict__ ri, const double* __restrict__ x, long ncol, int len,
                                                  int* __restrict__ ori, double* __restrict__ ox, double* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  double acc = 0;
  for (long c = w0; c < ncol; c += nw) {
    const long p0 = c * len, p1 = p0 + len;
    const long pa = p0 & ~3l;
    const int nit = (int)((p1 - pa + 255) >> 8);
    int4 g[CHV]; double2 va[CHV], vb[CHV];
#pragma unroll
    for (int m = 0; m < CHV; ++m) {
      g[m] = make_int4(0, 0, 0, 0); va[m] = make_double2(0, 0); vb[m] = va[m];
      if (m < nit) { long p = pa + (long)(m * 64 + lane) * 4; if (p < p1) { g[m] = *(const int4*)(ri + p); va[m] = *(const double2*)(x + p); vb[m] = *(const double2*)(x + p + 2); } }
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < CHV; ++m) s += va[m].x + va[m].y + vb[m].x + vb[m].y + g[m].x + g[m].w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (WRITE) {
#pragma unroll
      for (int m = 0; m < CHV; ++m) if (m < nit) { long p = pa + (long)(m * 64 + lane) * 4; if (p < p1) { *(int4*)(ori + p) = g[m]; *(double2*)(ox + p) = make_double2(va[m].x * s, va[m].y * s); *(double2*)(ox + p + 2) = make_double2(vb[m].x * s, vb[m].y * s); } }
    }
    acc += s;
  }
  if (acc == 1.2345) sink[0] = acc;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_stream(const int* __restrict__ ri, const double* __restrict__ x, long n, int* __restrict__ ori,
                                                double* __restrict__ ox, double* __restrict__ sink) {
  double acc = 0;
  for (long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4; p < n; p += (long)gridDim.x * 256 * 4) {
    int4 g = *(const int4*)(ri + p); double2 a = *(const double2*)(x + p), b = *(const double2*)(x + p + 2);
    if (WRITE) { *(int4*)(ori + p) = g; *(double2*)(ox + p) = a; *(double2*)(ox + p + 2) = b; }
    acc += a.x + b.y + g.x;
  }
  if (acc == 1.2345) sink[0] = acc;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
  const long ncol = 54000; const int len = argc > 1 ? atoi(argv[1]) : 1107; const long n = ncol * len;
  int *ri, *ori; double *x, *ox, *sink;
  CK(hipMalloc(&ri, n * 4 + 64)); CK(hipMalloc(&ori, n * 4 + 64)); CK(hipMalloc(&x, n * 8 + 64)); CK(hipMalloc(&ox, n * 8 + 64)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(ri, 0, n * 4)); CK(hipMemset(x, 0, n * 8));
  const double rb = n * 12.0 / 1e6, wb = n * 12.0 / 1e6;   // MB
  for (int bpc : {2, 4, 8}) {
    const int grid = 256 * bpc;
    float t;
#define RUN(name, kern, wr) t = timeit([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, ri, x, ncol, len, ori, ox, sink); }, 10); \
    printf("%-28s blocks/CU=%d  %.1f us  %.2f TB/s\n", name, bpc, t * 1e3, (rb + (wr ? wb : 0)) / t / 1e6);
    RUN("wave scalar CH=24 read", (k_wave_scalar<24, false>), 0)
    RUN("wave scalar CH=24 r+w", (k_wave_scalar<24, true>), 1)
    RUN("wave vec CHV=6 read", (k_wave_vec<6, false>), 0)
    RUN("wave vec CHV=6 r+w", (k_wave_vec<6, true>), 1)
    t = timeit([&] { hipLaunchKernelGGL(k_stream<false>, dim3(grid), dim3(256), 0, 0, ri, x, n, ori, ox, sink); }, 10);
    printf("%-28s blocks/CU=%d  %.1f us  %.2f TB/s\n", "stream read", bpc, t * 1e3, rb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(k_stream<true>, dim3(grid), dim3(256), 0, 0, ri, x, n, ori, ox, sink); }, 10);
    printf("%-28s blocks/CU=%d  %.1f us  %.2f TB/s\n", "stream r+w", bpc, t * 1e3, (rb + wb) / t / 1e6);
  }
  return 0;
}
