// probe_launch.hip -- cost of a dependent kernel boundary: back-to-back stream launches vs HIP-graph replay,
// for (almost) empty kernels of 256 / 512 / 2048 workgroups and chains of 10 kernels (the hot path's step).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/probes/probe_launch tools/probes/probe_launch.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) tiny(float *p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] += 1.0f;
}

int main() {
  float *buf;
  CK(hipMalloc(&buf, 2048 * 256 * 4));
  CK(hipMemset(buf, 0, 2048 * 256 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int chain = 10, reps = 300;
  for (int wgs : {256, 512, 2048}) {
    float ms_stream = 0, ms_graph = 0;
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < chain; ++k) tiny<<<wgs, 256, 0, st>>>(buf, wgs * 256);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_stream, e0, e1));
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < chain; ++k) tiny<<<wgs, 256, 0, st>>>(buf, wgs * 256);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_graph, e0, e1));
    }
    printf("%4d workgroups: chain of %d dependent kernels: stream launches %.2f us per kernel, graph replay %.2f us per kernel\n",
           wgs, chain, ms_stream * 1e3 / (reps * chain), ms_graph * 1e3 / (reps * chain));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
