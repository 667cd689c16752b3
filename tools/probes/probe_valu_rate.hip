// Probe: issue cost of v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32 (wave64), N independent accumulator chains, at 1, 2
// and 4 waves per SIMD.  Prints cycles per instruction per SIMD (s_memtime ticks = shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
using f2 = __attribute__((ext_vector_type(2))) float;
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int iters, float a, float b) {
  float r[16];
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i; p[i] = f2{(float)threadIdx.x + i, (float)i}; }
  const f2 a2 = f2{a, a + 1.f}, b2 = f2{b, b + 1.f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(a2), "v"(b2));
      if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(a2));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(a2), "v"(b2));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, float *d, long long *c, int threads) {
  const int iters = 2000;
  k<MODE><<<256, threads>>>(d, c, iters, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  long long h[256];
  (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += h[i];
  m /= 256;
  const int waves_per_simd = threads / 256;
  printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction per SIMD\n", name, waves_per_simd,
         m / ((double)iters * 16 * waves_per_simd));
}
int main() {
  float *d; long long *c;
  (void)hipMalloc(&d, 256 * 1024 * 4); (void)hipMalloc(&c, 256 * 8);
  for (int threads : {256, 512, 1024}) {
    run<0>("v_fma_f32", d, c, threads);
    run<1>("v_pk_fma_f32", d, c, threads);
    run<3>("v_pk_fma_f32 op_sel_hi (bcast a.x)", d, c, threads);
    run<2>("v_pk_mul_f32", d, c, threads);
  }
  return 0;
}
