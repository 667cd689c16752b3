// Probe (round 5): read rate of the stage-0 pointwise's A operand -- d [M = 16384][K = 1024] fp32 (67 MB), rows of 4 KB --
// when every WAVE streams its own 32 rows x k-range through a wave-private LDS-DMA ring (pws_kernel's structure):
// how many bytes must be in flight per CU before a launch this short gets past the ~2.7 TB/s of the tile-shaped
// pwi8_kernel?  Variants: ring depth R, waves per workgroup WPB (LDS = WPB * R * 4 KB decides the workgroups per CU),
// KS = waves splitting K.  Rotating buffers (HBM) and one buffer (Infinity-Cache resident, as d is in the pipeline).
// Build: hipcc -O3 --offload-arch=gfx950 probe_rowstream.hip -o probe_rowstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
constexpr int M = 16384, K = 1024;

__device__ __forceinline__ void glds16(const void *gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned long long b = (unsigned long long)gbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(bu), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

template <int R, int KS>
__global__ void __launch_bounds__(1024) k_rows(const float *A, float *out) {
  extern __shared__ float4 lds4[];
  char *lds = reinterpret_cast<char *>(lds4);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int gw = blockIdx.x * wpb + w;                 // global wave: (row block, k slice)
  const int rb = gw / KS, ks = gw % KS;
  const int Kw = K / KS, nit = Kw / 32;
  const float *abase = A + (long)rb * 32 * K + ks * Kw;
  const int dr = lane >> 3, dc = lane & 7;
  unsigned aoff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) aoff[u] = (unsigned)(8 * u + dr) * K * 4u + dc * 16;
  const unsigned ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)lds + w * R * 4096;
  const char *wring = lds + w * R * 4096;
  auto issue = [&](int t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) glds16(abase, aoff[u] + t * 128, ring + (t % R) * 4096 + u * 1024);
  };
  float4 acc = make_float4(0, 0, 0, 0);
  int issued = 0;
  for (; issued < R - 1 && issued < nit; ++issued) issue(issued);
  for (int t = 0; t < nit; ++t) {
    if (issued < nit) issue(issued++);
    const int pend = issued - 1 - t;                    // windows behind this one still in flight (4 DMAs each)
    if (pend == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (pend == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pend == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (pend == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (pend == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (pend == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    const float4 v = *reinterpret_cast<const float4 *>(wring + (t % R) * 4096 + lane * 64);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
// the tile-shaped pattern of pwi8_kernel: 256 threads own 64 rows, one 32-k tile (128 B per row) in flight
__global__ void __launch_bounds__(256) k_tiles(const float *A, float *out) {
  const int tid = threadIdx.x, lr = tid >> 3, lk = (tid & 7) * 4;
  const float *p0 = A + ((long)blockIdx.x * 64 + lr) * K + lk, *p1 = p0 + 32L * K;
  float4 a = make_float4(0, 0, 0, 0);
  float4 n0 = *reinterpret_cast<const float4 *>(p0), n1 = *reinterpret_cast<const float4 *>(p1);
  for (int k0 = 0; k0 < K; k0 += 32) {
    const float4 v0 = n0, v1 = n1;
    if (k0 + 32 < K) { n0 = *reinterpret_cast<const float4 *>(p0 + k0 + 32); n1 = *reinterpret_cast<const float4 *>(p1 + k0 + 32); }
    a.x += v0.x + v1.x; a.y += v0.y + v1.y; a.z += v0.z + v1.z; a.w += v0.w + v1.w;
    __syncthreads();
  }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = a.x;
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
template <int R, int KS>
void run(std::vector<float *> &bufs, float *out, int wpb) {
  const size_t lds = (size_t)wpb * R * 4096;
  if (lds > 160 * 1024) return;
  auto kern = k_rows<R, KS>;
  (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int waves = M / 32 * KS, grid = waves / wpb;
  const double mb = (double)M * K * 4 / 1e6;
  const float us_h = timeit([&](int i) { kern<<<grid, wpb * 64, lds, 0>>>(bufs[i & 7], out); });
  const float us_c = timeit([&](int i) { kern<<<grid, wpb * 64, lds, 0>>>(bufs[0], out); });
  printf("rows  R=%d KS=%d waves/WG=%2d LDS/WG=%3zu KB grid=%4d : rotating %6.1f us %5.2f TB/s | one buffer %6.1f us %5.2f TB/s\n", R, KS,
         wpb, lds / 1024, grid, us_h, mb / us_h / 1e6 * 1e6 / 1e6, us_c, mb / us_c / 1e6 * 1e6 / 1e6);
}
int main() {
  const long n = (long)M * K;
  std::vector<float *> bufs(8);
  for (auto &b : bufs) { (void)hipMalloc(&b, n * 4); (void)hipMemset(b, 0, n * 4); }
  float *out; (void)hipMalloc(&out, 4);
  const double mb = n * 4 / 1e6;
  {
    const float us_h = timeit([&](int i) { k_tiles<<<M / 64, 256>>>(bufs[i & 7], out); });
    const float us_c = timeit([&](int i) { k_tiles<<<M / 64, 256>>>(bufs[0], out); });
    printf("tiles (pwi8 pattern, 256 WGs of 64 rows, 1 k tile ahead): rotating %6.1f us %5.2f TB/s | one buffer %6.1f us %5.2f TB/s\n",
           us_h, mb / us_h, us_c, mb / us_c);
  }
  for (int wpb : {4, 8, 16}) {
    run<2, 4>(bufs, out, wpb); run<3, 4>(bufs, out, wpb); run<4, 4>(bufs, out, wpb); run<6, 4>(bufs, out, wpb); run<8, 4>(bufs, out, wpb);
  }
  for (int wpb : {4, 8}) { run<3, 1>(bufs, out, wpb); run<6, 1>(bufs, out, wpb); run<4, 8>(bufs, out, wpb); run<8, 8>(bufs, out, wpb); }
  return 0;
}
