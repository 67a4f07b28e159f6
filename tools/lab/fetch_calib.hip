// fetch_calib.hip — what do FETCH_SIZE / TCC_EA0_RDREQ count for the access shapes of this code?  (tools only)
//
// MI355X_MICROARCH.md §HBM prescribes FETCH_SIZE x 2 on gfx950 "128 B requests tallied at 64 B".  That holds for wide
// coalesced streaming reads; the Jaccard edge kernel's reads are 64 B row gathers (4 lanes x 16 B).  Three kernels with
// KNOWN bytes, on a table far larger than the 256 MiB Infinity Cache and every row touched exactly once (no reuse, no
// cache hit possible), to be run under rocprofv3 --pmc:
//   k_calib_gather64   rows of  64 B gathered in a random order (a permutation), 4 lanes x 16 B per row
//   k_calib_gather128  rows of 128 B, 8 lanes x 16 B per row
//   k_calib_stream     the same bytes read front to back, 16 B per lane, coalesced
// Prints the known bytes per launch; tools/make_traffic.py divides the counters by them.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/fetch_calib.hip -o tools/lab/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int ROWB>
__global__ __launch_bounds__(256) void k_calib_gather(const char* __restrict__ table, const uint32_t* __restrict__ perm, long n_rows,
                                                      uint32_t* __restrict__ sink) {
  constexpr int LPR = ROWB / 16, RPS = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const long w0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  uint32_t acc = 0;
  for (long r0 = w0 * RPS; r0 < n_rows; r0 += nw * RPS) {
    const long r = r0 + lane / LPR;
    if (r < n_rows) {
      const v4u v = *reinterpret_cast<const v4u*>(table + (long)perm[r] * ROWB + (lane % LPR) * 16);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;      // keeps the loads alive
}
template __global__ void k_calib_gather<64>(const char*, const uint32_t*, long, uint32_t*);
template __global__ void k_calib_gather<128>(const char*, const uint32_t*, long, uint32_t*);

__global__ __launch_bounds__(256) void k_calib_stream(const char* __restrict__ table, long bytes, uint32_t* __restrict__ sink) {
  const long t0 = (long)blockIdx.x * 256 + threadIdx.x, nt = (long)gridDim.x * 256;
  uint32_t acc = 0;
  for (long p = t0 * 16; p < bytes; p += nt * 16) {
    const v4u v = *reinterpret_cast<const v4u*>(table + p);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long table_bytes = (argc > 1 ? atol(argv[1]) : 2048) * (1l << 20);   // MiB
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  char* table;
  uint32_t *perm64, *perm128, *sink;
  CK(hipMalloc(&table, table_bytes));
  CK(hipMemset(table, 1, table_bytes));
  const long n64 = table_bytes / 64, n128 = table_bytes / 128;
  // a permutation of the rows: i -> (a * i + b) mod n with a odd (n is a power of two times something: use a coprime multiplier)
  auto make_perm = [&](long n, uint32_t** d) {
    std::vector<uint32_t> h((size_t)n);
    uint64_t a = 2654435761ull;
    while (true) { uint64_t x = a, y = (uint64_t)n; while (y) { uint64_t t = x % y; x = y; y = t; } if (x == 1) break; a += 2; }
    for (long i = 0; i < n; ++i) h[(size_t)i] = (uint32_t)((a * (uint64_t)i + 12345ull) % (uint64_t)n);
    CK(hipMalloc(d, sizeof(uint32_t) * (size_t)n));
    CK(hipMemcpy(*d, h.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
  };
  make_perm(n64, &perm64);
  make_perm(n128, &perm128);
  CK(hipMalloc(&sink, 64));
  const int grid = 256 * 8;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k_calib_gather<64>, dim3(grid), dim3(256), 0, 0, table, perm64, n64, sink);
    hipLaunchKernelGGL(k_calib_gather<128>, dim3(grid), dim3(256), 0, 0, table, perm128, n128, sink);
    hipLaunchKernelGGL(k_calib_stream, dim3(grid), dim3(256), 0, 0, table, table_bytes, sink);
  }
  CK(hipDeviceSynchronize());
  printf("calib known_bytes_per_launch k_calib_gather<64> rows=%ld row_bytes=64 table_bytes=%ld index_bytes=%ld\n", n64, table_bytes, n64 * 4);
  printf("calib known_bytes_per_launch k_calib_gather<128> rows=%ld row_bytes=128 table_bytes=%ld index_bytes=%ld\n", n128, table_bytes, n128 * 4);
  printf("calib known_bytes_per_launch k_calib_stream table_bytes=%ld\n", table_bytes);
  return 0;
}
