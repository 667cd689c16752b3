// probe_gridbar2.hip -- the device-wide barrier of probe_gridbar.hip redone the way MI355X wants it
// (/opt/skills/guides/MI355X_MICROARCH.md, price list rows barrier-counter / barrier-xcd; VERDICT r3 "next" #1a).
// probe_gridbar.hip had EVERY thread of EVERY workgroup execute __threadfence() and all workgroups arrive on ONE
// word: 87-162 us per barrier.  Here:
//   flat : lane 0 only: release fence -> ticket on one counter -> relaxed poll + s_sleep -> acquire fence
//   xcd  : XCD-hierarchical.  Lane 0 arrives on its XCC's counter (relaxed, no fence: its workgroup's stores are
//          drained to the XCD's L2 by s_waitcnt vmcnt(0) + __syncthreads()); the LAST arriver of an XCC executes ONE
//          agent-scope release fence (one L2 write-back per XCD instead of one per workgroup), arrives on the top
//          counter; the last XCC bumps the eight per-XCC generation words; everybody polls its own XCC's word
//          relaxed, then ONE acquire fence per workgroup.  The XCC of a workgroup is READ (HW_REG_XCC_ID), the
//          population of each XCC is counted by a census at kernel entry (placement is undefined by contract:
//          nothing here depends on blockIdx % 8).
//   wt   : phase stores are write-through (global_store_dwordx4 ... sc1): no release fence at all, two-level
//          ticket (sharded by XCC for speed only), poll, acquire.
// Every spin is bounded (a scheduling problem shows up in the timeout word, not as a hang).
// Work: R rounds of { y = 2x on the workgroup's own slice | barrier | z = y + 1 on the slice of the workgroup half
// a grid away (written on another XCD) | barrier } in ONE launch, against 2R kernels back to back.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/probes/probe_gridbar2 tools/probes/probe_gridbar2.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar {                      // every word on its own 128-byte line; zeroed by a memset before every launch
  unsigned xcnt[8 * 32];          // arrivals per XCC (monotonic within a launch)
  unsigned xgen[8 * 32];          // generation per XCC
  unsigned top[32];               // XCC leaders (xcd) / workgroups (flat) arrived
  unsigned census[8 * 32];        // workgroups per XCC
  unsigned census_total[32];
  unsigned timeouts[32];
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

#define SPIN_LIMIT 2000000

struct BarCtx { unsigned xcc, n_xcc, n_active, k; };

// kernel entry, lane 0 of the workgroup: join the census
__device__ __forceinline__ void bar_enter(Bar *b, BarCtx &c) {
  c.xcc = xcc_id();
  c.k = 0; c.n_xcc = 0; c.n_active = 0;
  __hip_atomic_fetch_add(&b->census[c.xcc * 32], 1u, RLX);
  __hip_atomic_fetch_add(&b->census_total[0], 1u, RLX);
}
// before the first barrier, lane 0: the census is complete once every workgroup of the grid has entered
__device__ __forceinline__ void bar_census(Bar *b, BarCtx &c) {
  long spins = 0;
  while (__hip_atomic_load(&b->census_total[0], RLX) < gridDim.x) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_LIMIT) { atomicAdd(&b->timeouts[0], 1u); break; }
  }
  unsigned act = 0;
  for (int x = 0; x < 8; ++x) {
    const unsigned n = __hip_atomic_load(&b->census[x * 32], RLX);
    act += n != 0;
    if (x == (int)c.xcc) c.n_xcc = n;
  }
  c.n_active = act;
}

template <int MODE>  // 0 flat, 1 xcd, 2 wt
__device__ __forceinline__ void grid_barrier(Bar *b, BarCtx &c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned k = ++c.k;
    long spins = 0;
    if (MODE == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&b->top[0], 1u, RLX);
      while (__hip_atomic_load(&b->top[0], RLX) < k * gridDim.x) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { atomicAdd(&b->timeouts[0], 1u); break; }
      }
    } else {
      const unsigned t = __hip_atomic_fetch_add(&b->xcnt[c.xcc * 32], 1u, RLX);
      if (t == k * c.n_xcc - 1) {                     // last arriver of this XCC
        if (MODE == 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned tt = __hip_atomic_fetch_add(&b->top[0], 1u, RLX);
        if (tt == k * c.n_active - 1)
          for (int x = 0; x < 8; ++x) __hip_atomic_store(&b->xgen[x * 32], k, RLX);
      }
      while (__hip_atomic_load(&b->xgen[c.xcc * 32], RLX) < k) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { atomicAdd(&b->timeouts[0], 1u); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <bool WT>
__device__ __forceinline__ void store4(float4 *p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  if (WT) {
    f4v w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
  } else *p = v;
}

template <bool WT>
__device__ __forceinline__ void phase_a(const float4 *x, float4 *y, long n4, int wg, int nwg, float rr) {
  const long per = (n4 + nwg - 1) / nwg, lo = (long)wg * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float4 v = x[i];
    v.x = v.x * 2.f + rr; v.y = v.y * 2.f + rr; v.z = v.z * 2.f + rr; v.w = v.w * 2.f + rr;   // differs per round: a stale y shows
    store4<WT>(y + i, v);
  }
}
template <bool WT>
__device__ __forceinline__ void phase_b(const float4 *y, float4 *z, long n4, int wg, int nwg) {
  const int src = (wg + nwg / 2 + 3) % nwg;                   // somebody else's slice, another XCD
  const long per = (n4 + nwg - 1) / nwg, lo = (long)src * per, hi = min(n4, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float4 v = y[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    store4<WT>(z + i, v);
  }
}

__global__ void __launch_bounds__(256) kern_a(const float4 *x, float4 *y, long n4, float rr) { phase_a<false>(x, y, n4, blockIdx.x, gridDim.x, rr); }
__global__ void __launch_bounds__(256) kern_b(const float4 *y, float4 *z, long n4) { phase_b<false>(y, z, n4, blockIdx.x, gridDim.x); }

template <int MODE>
__global__ void __launch_bounds__(256)
kern_fused(const float4 *x, float4 *y, float4 *z, long n4, int rounds, Bar *bar) {
  BarCtx c;
  if (threadIdx.x == 0) bar_enter(bar, c);
  for (int r = 0; r < rounds; ++r) {
    phase_a<MODE == 2>(x, y, n4, blockIdx.x, gridDim.x, (float)r);
    if (r == 0 && threadIdx.x == 0) bar_census(bar, c);
    grid_barrier<MODE>(bar, c);
    phase_b<MODE == 2>(y, z, n4, blockIdx.x, gridDim.x);
    if (r + 1 < rounds) grid_barrier<MODE>(bar, c);
  }
}

// barrier alone: nothing published between barriers
template <int MODE>
__global__ void __launch_bounds__(256) kern_bar_only(int nbar, Bar *bar) {
  BarCtx c;
  if (threadIdx.x == 0) { bar_enter(bar, c); bar_census(bar, c); }
  for (int r = 0; r < nbar; ++r) grid_barrier<MODE>(bar, c);
}

int main(int argc, char **argv) {
  const int rounds = 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Bar *bar;
  CK(hipMalloc(&bar, sizeof(Bar)));
  Bar hb;
  for (int nwg : {256, 512, 1024}) {
    // (1) the barrier alone: 64 barriers per launch
    for (int mode = 0; mode < 3; ++mode) {
      const int nbar = 64, reps = 20;
      float ms = 0;
      for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
          CK(hipMemsetAsync(bar, 0, sizeof(Bar)));
          if (mode == 0) kern_bar_only<0><<<nwg, 256>>>(nbar, bar);
          if (mode == 1) kern_bar_only<1><<<nwg, 256>>>(nbar, bar);
          if (mode == 2) kern_bar_only<2><<<nwg, 256>>>(nbar, bar);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
      printf("barrier alone, %4d workgroups, %-4s: %6.2f us per barrier (launch + memset included over %d barriers; timeouts %u; census",
             nwg, mode == 0 ? "flat" : mode == 1 ? "xcd" : "wt", ms * 1e3 / reps / nbar, nbar, hb.timeouts[0]);
      for (int x = 0; x < 8; ++x) printf(" %u", hb.census[x * 32]);
      printf(")\n");
    }
    // (2) with data
    for (long mb : {4L, 32L, 128L}) {
      const long n4 = mb * 1024 * 1024 / 16;
      float4 *x, *y, *z;
      CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&z, n4 * 16));
      std::vector<float> h(n4 * 4);
      for (long i = 0; i < n4 * 4; ++i) h[i] = (float)(i % 1000);
      CK(hipMemcpy(x, h.data(), n4 * 16, hipMemcpyHostToDevice));
      const int reps = 50;
      float ms_sep = 0;
      for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r)
          for (int q = 0; q < rounds; ++q) {
            kern_a<<<nwg, 256>>>(x, y, n4, (float)q);
            kern_b<<<nwg, 256>>>(y, z, n4);
          }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_sep, e0, e1));
      }
      printf("%4ld MB x3, %4d workgroups: %d kernels %.1f us per phase pair |", mb, nwg, 2 * rounds, ms_sep * 1e3 / reps / rounds);
      for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        CK(hipMemset(z, 0, n4 * 16));
        CK(hipMemset(y, 0, n4 * 16));
        for (int pass = 0; pass < 2; ++pass) {
          CK(hipEventRecord(e0));
          for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(bar, 0, sizeof(Bar)));
            if (mode == 0) kern_fused<0><<<nwg, 256>>>(x, y, z, n4, rounds, bar);
            if (mode == 1) kern_fused<1><<<nwg, 256>>>(x, y, z, n4, rounds, bar);
            if (mode == 2) kern_fused<2><<<nwg, 256>>>(x, y, z, n4, rounds, bar);
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        CK(hipMemcpy(h.data(), z, n4 * 16, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < n4 * 4; ++i) bad += h[i] != (float)(i % 1000) * 2.f + (float)(rounds - 1) + 1.f;
        CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        printf(" %s %.1f us (wrong %ld, timeouts %u)", mode == 0 ? "flat" : mode == 1 ? "xcd" : "wt", ms * 1e3 / reps / rounds, bad, hb.timeouts[0]);
      }
      printf("\n");
      CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(z));
    }
  }
  return 0;
}
