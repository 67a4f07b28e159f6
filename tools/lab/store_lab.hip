// store_lab.hip — does the shape of the edge kernel's output stores matter?  (tools only)
//  A: one wave per cell, three arrays, 30 lanes x 8 B at byte offset cell*240 (what k_jaccard_edges does)
//  B: one wave per 8 cells, three arrays, 1920 B = 15 whole 128 B lines, 16 B per lane (LDS-staged form)
//  C: as A but with plain (not nontemporal) stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_a(double* s, double* d, double* w, long ncell, int k) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  for (long c = w0; c < ncell; c += nw) {
    if (lane < k) {
      const long r = c * k + lane;
      if (NT) { __builtin_nontemporal_store((double)c, s + r); __builtin_nontemporal_store((double)lane, d + r); __builtin_nontemporal_store(0.5, w + r); }
      else { s[r] = (double)c; d[r] = (double)lane; w[r] = 0.5; }
    }
  }
}
__global__ __launch_bounds__(256) void k_b(double* s, double* d, double* w, long ncell, int k) {
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  const long ngroups = ncell / 8;
  typedef double v2d __attribute__((ext_vector_type(2)));
  for (long g = w0; g < ngroups; g += nw) {
    const long base = g * 8 * k;           // doubles
    for (int t = lane; t < 4 * k; t += 64) {     // 8*k doubles = 4*k pairs
      __builtin_nontemporal_store(v2d{(double)g, (double)t}, (v2d*)(s + base) + t);
      __builtin_nontemporal_store(v2d{(double)g, (double)t}, (v2d*)(d + base) + t);
      __builtin_nontemporal_store(v2d{0.5, 0.25}, (v2d*)(w + base) + t);
    }
  }
}
int main() {
  const long ncell = 100000; const int k = 30; const long E = ncell * k;
  double *s, *d, *w; CK(hipMalloc(&s, E * 8)); CK(hipMalloc(&d, E * 8)); CK(hipMalloc(&w, E * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int grid) {
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, s, d, w, ncell, k);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-40s grid %5d: %7.1f us  %.2f TB/s\n", name, grid, best * 1e3, 3.0 * E * 8 / (best * 1e-3) / 1e12);
  };
  for (int grid : {768, 2048}) {
    run("A nt  8 B/lane, 240 B per cell", k_a<true>, grid);
    run("C     8 B/lane, 240 B per cell", k_a<false>, grid);
    run("B nt 16 B/lane, 1920 B per 8 cells", k_b, grid);
  }
  return 0;
}
