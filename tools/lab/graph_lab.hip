// graph_lab: cost of a chain of dependent, immediately-returning kernels — plain stream launches against one hipGraph launch.
// hipcc -O3 --offload-arch=gfx950 tools/lab/graph_lab.hip -o tools/lab/graph_lab && tools/lab/graph_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_gated(const unsigned* gate, unsigned* out) { if (*gate == 0u) return; out[threadIdx.x] = 1u; }
int main() {
  unsigned *gate, *out;
  CK(hipMalloc(&gate, 4)); CK(hipMalloc(&out, 4096)); CK(hipMemset(gate, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int CH = 8, REP = 200;
  for (int grid : {1, 32, 256}) {
    auto chain = [&]() { for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_gated, dim3(grid), dim3(256), 0, st, gate, out); };
    chain(); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) chain();
    CK(hipStreamSynchronize(st));
    double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
    printf("grid %3d: chain of %d gated kernels: stream %.1f us, graph %.1f us\n", grid, CH, us_stream, us_graph);
  }
  return 0;
}
