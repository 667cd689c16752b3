// Probe: the stage-0 gather writes d as 256-byte pieces (64 channels of one pixel) at a 4-KB stride (row-major [M][1024]);
// a K-blocked d [C/64][M][64] would make an item's 256 pixels one contiguous 64-KB block.  Store-only kernels, 1024 items
// of 256 x 256 B, 1024-thread workgroups walking 4 items each (as dw0p_kernel), 16 B per lane per store.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LAYOUT>
__global__ void __launch_bounds__(1024) k(float *d, int M, int C, int nitems) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = lane & 15, sub = lane >> 4;
  const int nchunk = C / 64;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int n = item / nchunk, chunk = item % nchunk;
    for (int j = 0; j < 4; ++j) {
      const long m = (long)n * 256 + wave * 16 + j * 4 + sub;
      float *p = LAYOUT == 0 ? d + m * C + chunk * 64 + cq * 4 : d + (long)chunk * M * 64 + m * 64 + cq * 4;
      *reinterpret_cast<float4 *>(p) = make_float4(item, j, lane, wave);
    }
  }
}
int main() {
  const int N = 64, C = 1024, M = N * 256;
  float *d, *junk;
  (void)hipMalloc(&d, (size_t)M * C * 4); (void)hipMalloc(&junk, 512u << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int layout = 0; layout < 2; ++layout) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipMemsetAsync(junk, rep, 512u << 20);
      (void)hipEventRecord(e0);
      if (layout == 0) k<0><<<256, 1024>>>(d, M, C, N * (C / 64)); else k<1><<<256, 1024>>>(d, M, C, N * (C / 64));
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("store layout %d: %.1f us (%.2f TB/s)\n", layout, best * 1e3, 67.1 / (best * 1e3));
  }
  return 0;
}
