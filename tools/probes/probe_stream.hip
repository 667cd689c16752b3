// Probe: achievable HBM read rate of the stage-0 scale reduction (x [64][1024][256] fp32, 67 MB)
// under different access patterns; 8 rotating buffers (536 MB) so the 256 MB Infinity Cache cannot
// serve the reads.  Build: hipcc -O3 --offload-arch=gfx950 probe_stream.hip -o probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 64, C = 1024, HW = 256;

// plain stream: float4 per lane, grid-stride
__global__ void __launch_bounds__(1024) k_stream(const float4 *x, float *out, long n4) {
  float4 a = make_float4(0, 0, 0, 0);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
    a.x += v0.x + v1.x + v2.x + v3.x; a.y += v0.y + v1.y + v2.y + v3.y;
    a.z += v0.z + v1.z + v2.z + v3.z; a.w += v0.w + v1.w + v2.w + v3.w;
  }
  for (; i < n4; i += stride) { const float4 v = x[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = a.x;
}
// current scale_nchw pattern: WG = (64-pixel segment, n); wave v: channels v + 16 j, 4 in flight
__global__ void __launch_bounds__(1024) k_seg(const float *x, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *xp = x + (long)blockIdx.y * C * HW + blockIdx.x * 64 + lane;
  float a = 0;
  for (int c = wave; c < C; c += 64) {
    const float v0 = xp[(long)c * HW], v1 = xp[(long)(c + 16) * HW], v2 = xp[(long)(c + 32) * HW], v3 = xp[(long)(c + 48) * HW];
    a += v0 + v1 + v2 + v3;
  }
  if (a == 12345.f) out[0] = a;
}
// plane pattern: a wave instruction reads one whole 1 KB channel plane (lane <-> 4 pixels);
// WG = (channel group g of G, n); wave v: channels g*C/G + v + 16 j, U in flight
template <int U>
__global__ void __launch_bounds__(1024) k_plane(const float *x, float *out, int G) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = C / G;
  const float4 *xp = reinterpret_cast<const float4 *>(x + ((long)blockIdx.y * C + (long)blockIdx.x * cg) * HW) + lane;
  float4 a = make_float4(0, 0, 0, 0);
  for (int c = wave; c < cg; c += 16 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = xp[(long)(c + 16 * u) * (HW / 4)];
#pragma unroll
    for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = a.x;
}
// contiguous-chunk pattern: WG (256 threads) reads a contiguous 64 KB block (what dw2 staging does)
__global__ void __launch_bounds__(256) k_chunk(const float4 *x, float *out) {
  const float4 *xp = x + (long)blockIdx.x * 4096 + threadIdx.x;
  float4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = xp[u * 256];
  float a = 0;
#pragma unroll
  for (int u = 0; u < 16; ++u) a += v[u].x + v[u].y + v[u].z + v[u].w;
  if (a == 12345.f) out[0] = a;
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
  return ms / iters * 1e3f;   // us per launch
}
int main() {
  const long n = (long)N * C * HW;
  std::vector<float *> bufs(8);
  for (auto &b : bufs) { (void)hipMalloc(&b, n * 4); (void)hipMemset(b, 0, n * 4); }
  float *out; (void)hipMalloc(&out, 4);
  const double mb = n * 4 / 1e6;
  auto rep = [&](const char *name, float us) { printf("%-34s %7.1f us  %6.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6); };
  for (int g : {256, 512, 1024, 2048})
    rep((std::string("stream grid=") + std::to_string(g) + "x1024").c_str(),
        timeit([&](int i) { k_stream<<<g, 1024>>>(reinterpret_cast<const float4 *>(bufs[i & 7]), out, n / 4); }));
  rep("stream grid=4096x256", timeit([&](int i) { k_stream<<<4096, 256>>>(reinterpret_cast<const float4 *>(bufs[i & 7]), out, n / 4); }));
  rep("seg (scale_nchw pattern)", timeit([&](int i) { k_seg<<<dim3(4, N), 1024>>>(bufs[i & 7], out); }));
  for (int G : {1, 2, 4, 8, 16}) {
    rep((std::string("plane U=4 G=") + std::to_string(G)).c_str(), timeit([&](int i) { k_plane<4><<<dim3(G, N), 1024>>>(bufs[i & 7], out, G); }));
  }
  for (int G : {4, 8}) rep((std::string("plane U=8 G=") + std::to_string(G)).c_str(), timeit([&](int i) { k_plane<8><<<dim3(G, N), 1024>>>(bufs[i & 7], out, G); }));
  rep("chunk 64KB/WG 256thr", timeit([&](int i) { k_chunk<<<1024, 256>>>(reinterpret_cast<const float4 *>(bufs[i & 7]), out); }));
  rep("same buffer: stream 2048x1024", timeit([&](int i) { k_stream<<<2048, 1024>>>(reinterpret_cast<const float4 *>(bufs[0]), out, n / 4); }));
  rep("same buffer: seg", timeit([&](int i) { k_seg<<<dim3(4, N), 1024>>>(bufs[0], out); }));
  return 0;
}
