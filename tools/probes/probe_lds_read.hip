// Probe: ds_read_b128 throughput per CU for the gather's access pattern: a [cells][64 floats] image,
// lane l reads 16 B at cell(l / 16) * 256 + (l % 16) * 16 -- 4 different cells per wave instruction,
// bank == channel.  Variants: cells fixed per lane group vs data-dependent, 25 independent reads per
// iteration (as one gather step) with the loaded values consumed by FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int NREAD>
__global__ void __launch_bounds__(1024) k(float *out, const int *cells, int iters) {
  extern __shared__ float4 img[];
  const int ncell = 289;
  for (int i = threadIdx.x; i < ncell * 16; i += blockDim.x) img[i] = make_float4(i, 1, 2, 3);
  __syncthreads();
  // MODE 2: scattered, with the pixel of a lane chosen so that each HARDWARE lane group of ds_read_b128
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) reads ONE cell: group = parity of the lane's quad index
  const int lane = threadIdx.x & 63, cq = lane & 15;
  const int sub = MODE == 2 ? 2 * (lane >> 5) + (__popc((lane >> 2) & 7) & 1) : lane >> 4;
  float4 acc = make_float4(0, 0, 0, 0);
  int cell = (threadIdx.x * 7 + sub * 13) % (ncell - 32);
  const char *base = reinterpret_cast<const char *>(img) + cq * 16;
  for (int it = 0; it < iters; ++it) {
    int o[NREAD];
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      int c = cell + (r * 5) % 31;
      // scattered: every 16-lane pixel group reads a different pseudo-random cell each time (2 VALU ops)
      if (MODE >= 1) c = (cell * 37 + r * 53 + it * 11 + sub * 101) & 255;
      o[r] = c * 256;
    }
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      const float4 v = *reinterpret_cast<const float4 *>(base + o[r]);
      acc.x = fmaf(v.x, 1.0001f, acc.x); acc.y = fmaf(v.y, 1.0001f, acc.y);
      acc.z = fmaf(v.z, 1.0001f, acc.z); acc.w = fmaf(v.w, 1.0001f, acc.w);
    }
    cell = (cell + 3) % (ncell - 32);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE, int NREAD>
float run(float *d, const int *cells, int wgs_per_cu, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = 289 * 256;
  (void)hipFuncSetAttribute((const void *)k<MODE, NREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<MODE, NREAD><<<256 * wgs_per_cu, threads, lds>>>(d, cells, 8);
  (void)hipEventRecord(e0);
  k<MODE, NREAD><<<256 * wgs_per_cu, threads, lds>>>(d, cells, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float *d; int *cells;
  (void)hipMalloc(&d, 512 * 1024 * 4); (void)hipMalloc(&cells, 256);
  int h[64]; for (int i = 0; i < 64; ++i) h[i] = (i * 37) % 97;
  (void)hipMemcpy(cells, h, 256, hipMemcpyHostToDevice);
  const int iters = 2000;
  auto rep = [&](const char *name, float ms, int wgs, int threads, int nread) {
    const double reads = 256.0 * wgs * (threads / 64) * iters * nread;   // wave-instructions
    const double us = ms * 1e3;
    printf("%-44s %8.1f us  %5.2f ns per ds_read_b128 per CU  (%.1f clk at 2.4 GHz, %.0f B/clk/CU)\n", name, us,
           us * 1e3 / (reads / 256), us * 1e3 / (reads / 256) * 2.4, 1024.0 / (us * 1e3 / (reads / 256) * 2.4));
  };
  rep("25 reads/iter, 2 WG x 512 thr, static cells", run<0, 25>(d, cells, 2, 512, iters), 2, 512, 25);
  rep("25 reads/iter, 2 WG x 512 thr, scattered cells", run<1, 25>(d, cells, 2, 512, iters), 2, 512, 25);
  rep("25 reads/iter, 2 WG x 512 thr, scattered, hw-group lanes", run<2, 25>(d, cells, 2, 512, iters), 2, 512, 25);
  rep("25 reads/iter, 2 WG x 1024 thr, scattered cells", run<1, 25>(d, cells, 2, 1024, iters), 2, 1024, 25);
  rep("25 reads/iter, 2 WG x 1024 thr, scattered, hw-group lanes", run<2, 25>(d, cells, 2, 1024, iters), 2, 1024, 25);
  rep("25 reads/iter, 1 WG x 512 thr, static cells", run<0, 25>(d, cells, 1, 512, iters), 1, 512, 25);
  rep("25 reads/iter, 1 WG x 256 thr, static cells", run<0, 25>(d, cells, 1, 256, iters), 1, 256, 25);
  rep("8 reads/iter, 2 WG x 512 thr, static cells", run<0, 8>(d, cells, 2, 512, iters), 2, 512, 8);
  rep("4 reads/iter, 2 WG x 512 thr, scattered cells", run<1, 4>(d, cells, 2, 512, iters), 2, 512, 4);
  rep("25 reads/iter, 1 WG x 256 thr, scattered cells", run<1, 25>(d, cells, 1, 256, iters), 1, 256, 25);
  return 0;
}
