// Probe: phase timeline of the gather kernels from in-kernel stamps (thread 0 of each workgroup:
// 0 start, 1 staging issued+stored, 2 barrier passed, 3 gather done (thread 0's wave), 4 finish done).
// Includes the product source so the kernels are exactly the shipped ones.
#define CDN_STAMPS 1
#include "../../codenet_amd/csrc/codenet_fused.hip"
#include "../../codenet_amd/csrc/cdn_common.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

static void report(const char *name, int nwg, float us_kernel) {
  std::vector<unsigned long long> st(nwg * 8);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(cdn_stamps), st.size() * 8);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, st[i * 8]); t1 = std::max(t1, st[i * 8 + 4]); }
  double ph[4] = {0, 0, 0, 0};
  std::vector<double> starts;
  for (int i = 0; i < nwg; ++i) {
    for (int k = 0; k < 4; ++k) ph[k] += (double)(st[i * 8 + k + 1] - st[i * 8 + k]) / 100.0;
    starts.push_back((double)(st[i * 8] - t0) / 100.0);
  }
  std::sort(starts.begin(), starts.end());
  printf("%-28s kernel %6.1f us (events)  span %6.1f us  per-WG mean: stage %5.1f  barrier %5.1f  gather %5.1f  finish %5.1f us;"
         " WG starts: p25 %.1f p50 %.1f p75 %.1f max %.1f\n", name, us_kernel, (double)(t1 - t0) / 100.0,
         ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, starts[nwg / 4], starts[nwg / 2], starts[3 * nwg / 4], starts.back());
}

template <typename F>
static float time_us(F launch, int iters = 20) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}

int main() {
  const int N = 64;
  float *x, *s_raw, *wd, *d, *xmin, *xmax; float2 *mm; unsigned *state, *sstate, *xstate, *counters;
  (void)hipMalloc(&x, (size_t)N * 1024 * 256 * 4);
  {
    std::vector<float> hx((size_t)N * 1024 * 256);
    unsigned u = 12345u;
    for (auto &v : hx) { u = u * 1664525u + 1013904223u; v = (float)(u >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f; }
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  }
  (void)hipMalloc(&d, (size_t)N * 128 * 4096 * 4);
  (void)hipMalloc(&s_raw, (size_t)N * 4096 * 4);
  std::vector<float> ones((size_t)N * 4096, 1.3f);
  if (getenv("PROBE_RANDOM_S")) {   // s spread like a trained conv_scale output: N(1, 1.5) clamped to [-7, 8]
    unsigned u = 777u;
    for (auto &v : ones) {
      float a = 0.f;
      for (int i = 0; i < 4; ++i) { u = u * 1664525u + 1013904223u; a += (float)(u >> 8) * (1.0f / 16777216.0f); }
      v = fminf(fmaxf(1.0f + (a - 2.0f) * 2.6f, -7.f), 8.f);
    }
  }
  (void)hipMemcpy(s_raw, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&wd, 1024 * 9 * 4); (void)hipMemset(wd, 0, 1024 * 9 * 4);
  (void)hipMalloc(&mm, 16384 * 8);
  (void)hipMalloc(&xmin, 4); (void)hipMalloc(&xmax, 4); (void)hipMalloc(&state, 64); (void)hipMalloc(&sstate, 64); (void)hipMalloc(&xstate, 64);
  (void)hipMalloc(&counters, cdn::kArriveWords * 4);
  (void)hipMemset(xmin, 0, 4); (void)hipMemset(xmax, 0, 4); (void)hipMemset(state, 0, 64); (void)hipMemset(counters, 0, cdn::kArriveWords * 4);
  const float q[8] = {0, 0, 25.5f, 128.f, 0, 0, 0, 0};
  (void)hipMemcpy(sstate, q, 32, hipMemcpyHostToDevice); (void)hipMemcpy(xstate, q, 32, hipMemcpyHostToDevice);
  cdn::QUpdate qu{xmin, xmax, state, counters, -0.01f, 0.01f, 8, 1};
  // stage 0: C=1024, 16x16, NCHW, up=0
  {
    auto run = [&] { launch_dw2<64>(false, x, nullptr, s_raw, sstate, wd, d, mm, qu, N, 1024, 16, 16, 0, nullptr); };
    float us = time_us(run); run(); (void)hipDeviceSynchronize();
    report("stage0 dw2<64> 16x16", 16 * N, us);
  }
  // stage 1: C=256, 32x32 from 16x16 NHWC, up=1  -> dw2u<64>
  {
    auto run = [&] { launch_dw2<64>(true, x, xstate, s_raw, sstate, wd, d, mm, qu, N, 256, 32, 32, 1, nullptr); };
    float us = time_us(run); run(); (void)hipDeviceSynchronize();
    report("stage1 dw2u<64> 32x32", 4 * N, us);
  }
  // stage 2: C=128, 64x64 from 32x32 NHWC, up=1 -> dw2u<32>
  {
    auto run = [&] { launch_dw2<32>(true, x, xstate, s_raw, sstate, wd, d, mm, qu, N, 128, 64, 64, 1, nullptr); };
    float us = time_us(run); run(); (void)hipDeviceSynchronize();
    report("stage2 dw2u<32> 64x64", 4 * N, us);
  }
  return 0;
}
