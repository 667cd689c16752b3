// Probe: where the stage-0 scale kernel's time goes (x [64][1024][256] fp32, 8 rotating buffers).
// Variants add one ingredient at a time to the bare read pattern of probe_stream's k_seg.
#include "../../codenet_amd/csrc/cdn_common.h"
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>
constexpr int N = 64, C = 1024, HW = 256;
// MODE 0: loads only; 1: + weights/fma; 2: + LDS reduce + store s; 3: + block_minmax_finish
template <int MODE>
__global__ void __launch_bounds__(1024) k(const float *__restrict__ x, const float *__restrict__ w,
                                          float *__restrict__ s, float2 *mm, cdn::QUpdate qu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane, n = blockIdx.y;
  const float *xp = x + (long)n * C * HW + p;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int c = wave; c < C; c += 64) {
    const float v0 = xp[(long)c * HW], v1 = xp[(long)(c + 16) * HW], v2 = xp[(long)(c + 32) * HW], v3 = xp[(long)(c + 48) * HW];
    if (MODE >= 1) {
      a0 = fmaf(w[c], v0, a0); a1 = fmaf(w[c + 16], v1, a1); a2 = fmaf(w[c + 32], v2, a2); a3 = fmaf(w[c + 48], v3, a3);
    } else { a0 += v0; a1 += v1; a2 += v2; a3 += v3; }
  }
  float a = (a0 + a1) + (a2 + a3);
  __shared__ float red[16][64];
  float mn = INFINITY, mx = -INFINITY;
  if (MODE >= 2) {
    red[wave][lane] = a;
    __syncthreads();
    if (wave == 0) {
      float v = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) v += red[i][lane];
      v = fminf(fmaxf(v, -7.f), 8.f);
      s[(long)n * HW + p] = v;
      mn = mx = v;
    }
  } else if (a == 12345.f) s[0] = a;
  if (MODE >= 3) cdn::block_minmax_finish(mn, mx, mm, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, qu, &red[0][0]);
}
template <typename F>
float timeit(F launch, int iters = 40) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 8; ++i) launch(i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch(i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}
int main() {
  const long n = (long)N * C * HW;
  std::vector<float *> bufs(8);
  std::vector<float> hx(n);
  const bool rnd = getenv("PROBE_RANDOM") != nullptr;
  unsigned u = 12345u;
  for (auto &v : hx) { u = u * 1664525u + 1013904223u; v = rnd ? (float)(u >> 8) * (1.0f / 16777216.0f) : 0.f; }
  for (auto &b : bufs) { (void)hipMalloc(&b, n * 4); (void)hipMemcpy(b, hx.data(), n * 4, hipMemcpyHostToDevice); }
  float *w, *s, *xmin, *xmax; float2 *mm; unsigned *state, *counters;
  (void)hipMalloc(&w, C * 4); (void)hipMemset(w, 0, C * 4);
  (void)hipMalloc(&s, N * HW * 4); (void)hipMalloc(&mm, 16384 * 8);
  (void)hipMalloc(&xmin, 4); (void)hipMalloc(&xmax, 4); (void)hipMalloc(&state, 64); (void)hipMalloc(&counters, cdn::kArriveWords * 4);
  (void)hipMemset(xmin, 0, 4); (void)hipMemset(xmax, 0, 4); (void)hipMemset(state, 0, 64); (void)hipMemset(counters, 0, cdn::kArriveWords * 4);
  cdn::QUpdate qu{xmin, xmax, state, counters, -0.01f, 0.01f, 8, 1};
  const double mb = n * 4 / 1e6;
  auto rep = [&](const char *name, float us) { printf("%-34s %7.1f us  %6.2f TB/s\n", name, us, mb / us / 1e6); };
  dim3 g(4, N);
  rep("0 loads only", timeit([&](int i) { k<0><<<g, 1024>>>(bufs[i & 7], w, s, mm, qu); }));
  rep("1 + weights/fma", timeit([&](int i) { k<1><<<g, 1024>>>(bufs[i & 7], w, s, mm, qu); }));
  rep("2 + LDS reduce + store", timeit([&](int i) { k<2><<<g, 1024>>>(bufs[i & 7], w, s, mm, qu); }));
  rep("3 + minmax finish", timeit([&](int i) { k<3><<<g, 1024>>>(bufs[i & 7], w, s, mm, qu); }));
  rep("3 same buffer", timeit([&](int i) { k<3><<<g, 1024>>>(bufs[0], w, s, mm, qu); }));
  return 0;
}
