// probe_gridbar.hip -- what does a device-wide barrier INSIDE a kernel cost on MI355X, next to a kernel boundary?
// (a per-unit persistent kernel would replace two of a ShuffleNetV2 unit's three kernel
// boundaries -- each a global QuantAct-range dependency -- by such barriers.)
//   fused:    phase A (y = 2x on the workgroup's slice) -> release fence, ticket; the last arriver bumps an epoch;
//             everybody else polls the epoch (BOUNDED spin: a scheduling problem shows up as wrong results, not
//             as a hang) -> acquire fence -> phase B (z = y + 1 on ANOTHER workgroup's slice, half the grid away,
//             i.e. usually written on another XCD)
//   separate: kernel A, kernel B back to back on one stream.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/probes/probe_gridbar tools/probes/probe_gridbar.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void phase_a(const float4 *x, float4 *y, long n4, int wg, int nwg) {
  const long per = (n4 + nwg - 1) / nwg, lo = (long)wg * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float4 v = x[i];
    v.x *= 2.f; v.y *= 2.f; v.z *= 2.f; v.w *= 2.f;
    y[i] = v;
  }
}
__device__ __forceinline__ void phase_b(const float4 *y, float4 *z, long n4, int wg, int nwg) {
  const int src = (wg + nwg / 2) % nwg;                       // somebody else's slice
  const long per = (n4 + nwg - 1) / nwg, lo = (long)src * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float4 v = y[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    z[i] = v;
  }
}

__global__ void __launch_bounds__(256) kern_a(const float4 *x, float4 *y, long n4) { phase_a(x, y, n4, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) kern_b(const float4 *y, float4 *z, long n4) { phase_b(y, z, n4, blockIdx.x, gridDim.x); }

__global__ void __launch_bounds__(256)
kern_fused(const float4 *x, float4 *y, float4 *z, long n4, unsigned *sync /* [0] tickets, [16] epoch */, unsigned *timeouts) {
  __shared__ unsigned e0s;
  if (threadIdx.x == 0) e0s = __hip_atomic_load(sync + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  phase_a(x, y, n4, blockIdx.x, gridDim.x);
  __threadfence();                                            // release: this workgroup's writes, device wide
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sync + 16, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(sync + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e0s) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000) { atomicAdd(timeouts, 1u); break; }   // bounded: never hang the GPU
      }
    }
  }
  __syncthreads();
  __threadfence();                                            // acquire
  phase_b(y, z, n4, blockIdx.x, gridDim.x);
}

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 512;
  for (long mb : {4L, 32L, 128L}) {
    const long n4 = mb * 1024 * 1024 / 16;
    float4 *x, *y, *z;
    unsigned *sync, *timeouts;
    CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&z, n4 * 16));
    CK(hipMalloc(&sync, 256)); CK(hipMalloc(&timeouts, 4));
    CK(hipMemset(sync, 0, 256)); CK(hipMemset(timeouts, 0, 4));
    std::vector<float> h(n4 * 4);
    for (long i = 0; i < n4 * 4; ++i) h[i] = (float)(i % 1000);
    CK(hipMemcpy(x, h.data(), n4 * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    float ms_sep = 0, ms_fused = 0;
    for (int pass = 0; pass < 2; ++pass) {                    // first pass warms up
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        kern_a<<<nwg, 256>>>(x, y, n4);
        kern_b<<<nwg, 256>>>(y, z, n4);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_sep, e0, e1));
      CK(hipMemset(z, 0, n4 * 16));
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) kern_fused<<<nwg, 256>>>(x, y, z, n4, sync, timeouts);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_fused, e0, e1));
    }
    CK(hipMemcpy(h.data(), z, n4 * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < n4 * 4; ++i) bad += h[i] != (float)(i % 1000) * 2.f + 1.f;
    unsigned to = 0;
    CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("%4ld MB x3, %d workgroups: two kernels %.1f us/iter, one kernel + grid barrier %.1f us/iter (wrong values %ld, spin timeouts %u)\n",
           mb, nwg, ms_sep * 1e3 / reps, ms_fused * 1e3 / reps, bad, to);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(z)); CK(hipFree(sync)); CK(hipFree(timeouts));
  }
  return 0;
}
