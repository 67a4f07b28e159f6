// ticket_lab.hip — how fast can every wave of a persistent grid claim work items from ONE global
// ticket counter (returning atomic add, lane 0), with and without streaming loads in between?
// (tools only; decides the work distribution of the fused GF-ICF scaling pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every wave: claim BATCH tickets per atomic until `total` are gone; per ticket stream `len` entries
// of (int, double) from a column-major pool (like one cell of the CSC matrix)
template <int BATCH>
__global__ __launch_bounds__(1024) void k_ticket(unsigned long long* ticket, long total, const int* __restrict__ ri,
                                                 const double* __restrict__ x, int len, long pool_cols, double* sink,
                                                 unsigned long long* desc) {
  const int lane = threadIdx.x & 63;
  double acc = 0;
  for (;;) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(ticket, (unsigned long long)BATCH);
    t = __shfl(t, 0);
    if ((long)t >= total) break;
    for (int b = 0; b < BATCH && (long)t + b < total; ++b) {
      const long c = ((long)t + b) % pool_cols;
      const long p0 = c * len;
      double s = 0;
      for (int m = lane; m < len; m += 64) s += x[p0 + m] + ri[p0 + m];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
      acc += s;
      // publish a descriptor (write-through store), as the look-back would
      if (desc && lane == 0) __hip_atomic_store(desc + t + b, (2ull << 62) | (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 1.2345) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long cols = 54000;
  const int len = 1107;
  int* ri; double* x; double* sink; unsigned long long* ticket; unsigned long long* desc;
  CK(hipMalloc(&ri, cols * len * 4)); CK(hipMalloc(&x, cols * len * 8)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&ticket, 8));
  CK(hipMalloc(&desc, 8 * 600000));
  CK(hipMemset(ri, 0, cols * len * 4)); CK(hipMemset(x, 0, cols * len * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, long total, int l, bool with_desc) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(ticket, 0, 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, ticket, total, ri, x, l, cols, sink, with_desc ? desc : nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s tickets %7ld len %5d desc %d : %8.1f us  (%.1f ns per ticket, %.2f TB/s)\n", name, total, l, (int)with_desc, best * 1e3,
           best * 1e6 / total, (double)total * l * 12 / (best * 1e-3) / 1e12);
  };
  run("batch1 no work", k_ticket<1>, 54000, 0, false);
  run("batch1 no work", k_ticket<1>, 540000, 0, false);
  run("batch4 no work", k_ticket<4>, 540000, 0, false);
  run("batch1 stream", k_ticket<1>, 54000, len, false);
  run("batch1 stream+desc", k_ticket<1>, 54000, len, true);
  run("batch4 stream", k_ticket<4>, 54000, len, false);
  run("batch16 stream", k_ticket<16>, 54000, len, false);
  run("batch1 stream short", k_ticket<1>, 540000, 100, false);
  return 0;
}
