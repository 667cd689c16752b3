// Probe: effective shader clock (s_memtime ticks per s_memrealtime tick, the latter is 100 MHz) and
// the issue cost of a dependent v_fma chain, idle and with every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, int iters) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) a = fmaf(a, 1.0000001f, 1e-9f);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = c1 - c0;
    out[blockIdx.x * 4 + 1] = r1 - r0;
    out[blockIdx.x * 4 + 2] = (unsigned long long)(a * 1e6f);
  }
}
int main() {
  unsigned long long *d, h[4 * 4096];
  (void)hipMalloc(&d, sizeof(h));
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int grid = cfg == 0 ? 1 : (cfg == 1 ? 256 : 4096), block = cfg == 0 ? 64 : 1024;
    const int iters = 4000;
    k<<<grid, block>>>(d, iters);
    (void)hipDeviceSynchronize();
    k<<<grid, block>>>(d, iters);
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * 4 * grid, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[i * 4]; rt += h[i * 4 + 1]; }
    cyc /= grid; rt /= grid;
    printf("grid %4d x %4d: memtime ticks %.0f  realtime ticks %.0f (%.1f us)  -> memtime clock %.1f MHz; %.2f memtime ticks per dependent fma; %.2f ns per fma\n",
           grid, block, cyc, rt, rt / 100.0, cyc / rt * 100.0, cyc / (iters * 64.0), rt * 10.0 / (iters * 64.0));
  }
  return 0;
}
