// Probe: what bounds the LOAD SIDE of the stage-0 int8 pointwise (pwi8_kernel<64,128,2>: A[16384][1024] fp32 read as k-tiles of
// BM rows x 128 B, one barrier per k-tile, a second column-tile workgroup re-reading the same pieces)?  No arithmetic.
//   LAYOUT 0  row-major [M][K] (row stride 4 KB: what the gather writes today)
//   LAYOUT 1  K-blocked [K/64][M][64] (a k-tile pair of a workgroup = one contiguous block)
//   DEPTH     k-tiles of global loads in flight per thread (the kernel: 1)
//   BM        rows per workgroup (64: two workgroups per CU at M = 16384; 32: four)
// Build: hipcc -O3 --offload-arch=gfx950.  Results: DESIGN.md section 8.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LAYOUT, int DEPTH, int BM, bool BAR = true>
__global__ void __launch_bounds__(256) k(const float *A, float *out, int M, int K) {
  constexpr int AI = BM / 32;
  const int m0 = blockIdx.x * BM, tid = threadIdx.x;
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  float4 ring[DEPTH][AI];
  float acc = 0.f;
  auto addr = [&](int i, int t) -> const float4 * {
    const long m = m0 + lr + 32 * i;
    if (LAYOUT == 0) return reinterpret_cast<const float4 *>(A + m * K + 32 * t + lk);
    return reinterpret_cast<const float4 *>(A + (long)(t >> 1) * M * 64 + m * 64 + 32 * (t & 1) + lk);
  };
  const int nk = K / 32;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < AI; ++i) ring[d][i] = *addr(i, d);
  for (int t0 = 0; t0 < nk; t0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const float4 v = ring[d][i];
        acc += v.x + v.y + v.z + v.w;
        if (t + DEPTH < nk) ring[d][i] = *addr(i, t + DEPTH);
      }
      if (BAR) __syncthreads();
    }
  }
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}
template <int LAYOUT, int DEPTH, int BM, bool BAR = true, int NY = 2>
void run(const float *A, float *out, float *junk, int M, int K) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipMemsetAsync(junk, rep, 512u << 20);          // evict A from the Infinity Cache
    (void)hipEventRecord(e0);
    k<LAYOUT, DEPTH, BM, BAR><<<dim3(M / BM, NY), 256>>>(A, out, M, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("layout %d depth %d BM %3d barrier %d column tiles %d: %6.1f us  (%.2f TB/s of the 67 MB read once)\n", LAYOUT, DEPTH, BM, (int)BAR, NY, best * 1e3, 67.1 / (best * 1e3));
}
int main() {
  const int M = 16384, K = 1024;
  float *A, *out, *junk;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&junk, 512u << 20);
  (void)hipMemset(A, 0, (size_t)M * K * 4);
  run<0, 1, 64>(A, out, junk, M, K); run<0, 2, 64>(A, out, junk, M, K); run<0, 4, 64>(A, out, junk, M, K); run<0, 8, 64>(A, out, junk, M, K);
  run<0, 1, 32>(A, out, junk, M, K); run<0, 2, 32>(A, out, junk, M, K); run<0, 4, 32>(A, out, junk, M, K); run<0, 8, 32>(A, out, junk, M, K);
  run<1, 1, 64>(A, out, junk, M, K); run<1, 2, 64>(A, out, junk, M, K); run<1, 4, 64>(A, out, junk, M, K); run<1, 8, 64>(A, out, junk, M, K);
  run<1, 1, 32>(A, out, junk, M, K); run<1, 2, 32>(A, out, junk, M, K); run<1, 4, 32>(A, out, junk, M, K); run<1, 8, 32>(A, out, junk, M, K);
  // without the per-tile barrier / without the second column-tile workgroup re-reading the same pieces
  run<0, 2, 64, false>(A, out, junk, M, K); run<1, 2, 64, false>(A, out, junk, M, K);
  run<0, 2, 64, true, 1>(A, out, junk, M, K); run<1, 2, 64, true, 1>(A, out, junk, M, K);
  run<0, 4, 32, false, 1>(A, out, junk, M, K); run<1, 4, 32, false, 1>(A, out, junk, M, K);
  run<1, 8, 32, false, 1>(A, out, junk, M, K);
  return 0;
}
