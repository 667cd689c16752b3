// Probe: throughput of LDS atomics on gfx950 -- ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs plain
// ds_read/ds_write (conflict-free, consecutive lanes -> consecutive addresses, 16 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
  __shared__ float lf[8192];
  __shared__ unsigned long long l64[4096];
  unsigned *lu = reinterpret_cast<unsigned *>(lf);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lf[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) l64[i] = 0;
  __syncthreads();
  const int base = (threadIdx.x * 7) & 4095;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int a = (base + j * 64 + it) & 4095;
      if (MODE == 0) atomicAdd(&lf[a], 1.0f);
      if (MODE == 1) atomicAdd(&lu[a], 1u);
      if (MODE == 2) atomicAdd(&l64[a], 1ull);
      if (MODE == 3) acc += lf[a];
      if (MODE == 4) lf[a] = acc + j;
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lf[threadIdx.x] + (float)l64[threadIdx.x & 4095];
}
template <int MODE>
float run(float *d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 1024>>>(d, 8);
  (void)hipEventRecord(e0);
  k<MODE><<<256, 1024>>>(d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float *d; (void)hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 2000;
  const double ops = 256.0 * 1024 * iters * 16;   // lane-ops
  const char *names[5] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_read_b32", "ds_write_b32"};
  float ms[5] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters)};
  for (int m = 0; m < 5; ++m)
    printf("%-12s %8.3f ms  %7.2f lane-ops/clk/CU (at 2.4 GHz)\n", names[m], ms[m],
           ops / (ms[m] * 1e-3) / 256 / 2.4e9);
  return 0;
}
