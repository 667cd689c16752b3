// cdn_common.h -- shared host-side helpers for libcodenet_dcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/codenet_dcn.h"

namespace cdn {

// Thread-local last-error message (reference: AT_CHECK/AT_ERROR throw; launch errors are only
// printf-ed, _kernel.cu:271-275 -- here they are returned).
char *err_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CDN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return CDN_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kCUs = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr int kWave = 64;

// Geometry of one generic (modulated) deformable convolution call.
struct Geom {
  int N, C, H, W, Co, kH, kW, sH, sW, pH, pW, dH, dW, G, DG, Ho, Wo;
};

// Mirrors shape_check (dcn_deform_conv_cuda.cpp:61-149) for the parts that do not need the
// tensors themselves (the Python shim checks tensor shapes against these numbers).
int make_geom(Geom *g, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Co, int kH, int kW,
              int sH, int sW, int pH, int pW, int dH, int dW, int group, int dg);

// ---- optional per-kernel HIP-event profiler (bench.py's live roofline numbers) -----------------
// Off by default (zero overhead beyond one thread-local bool test per launch).  When enabled on
// the calling thread, every ProfScope records a start/end event pair on the launch stream.
enum ProfKernel { kProfScale = 0, kProfDw = 1, kProfPointwise = 2, kProfUnpack = 3, kProfUpdate = 4 };
struct ProfScope {
  ProfScope(int kernel_id, int tag, hipStream_t st);
  ~ProfScope();
  int slot_;
  hipStream_t st_;
};

// ---- QuantAct device state (codenet_quant.hip) --------------------------------------------------
// 8 x 4-byte words: [0] ordered-uint batch min, [1] ordered-uint batch max, [2] scale (f32),
// [3] zero-point (f32), [4] batch min (f32), [5] batch max (f32), [6..7] reserved.
constexpr int kQStateWords = 8;

// Reset the running batch min/max of up to three states (null entries skipped): 1 launch.
void launch_minmax_init(unsigned *s0, unsigned *s1, unsigned *s2, hipStream_t st);
// Range tracking + scale / zero-point derivation, see quantact_update_kernel.  Batch statistics
// come from (in this order of precedence) ext_min/ext_max device scalars, `partials`
// ([n_partials] {min,max} pairs written by producer workgroups, reduced here -- no atomics), or
// the ordered-uint words state[0..1].
void launch_quantact_update(float *x_min, float *x_max, unsigned *state, const float *ext_min,
                            const float *ext_max, const float2 *partials, int n_partials, int bits,
                            double momentum, int running, hipStream_t st);

}  // namespace cdn

#ifdef __HIPCC__
namespace cdn {
// Order-preserving float <-> uint map so min/max can use integer atomics.
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
// q = round(scale*x - zp) (half-even), no FMA contraction: quant_utils.py:33-41
__device__ __forceinline__ float quant_code(float x, float scale, float zp) {
  return rintf(__fsub_rn(__fmul_rn(scale, x), zp));
}
// (q + zp) / scale, true division: quant_utils.py:44-52
__device__ __forceinline__ float fake_quant(float x, float scale, float zp) {
  return __fdiv_rn(__fadd_rn(quant_code(x, scale, zp), zp), scale);
}
// Workgroup-level min/max -> ONE {min,max} pair stored at out[0] (plain store, no atomics: a
// single contended word sustains only ~88 atomics/us on MI355X).  Every thread of the workgroup
// must call it; `red` is >= 2*nwaves floats of LDS that nobody else is using.
__device__ __forceinline__ void block_minmax_store(float mn, float mx, float2 *out, float *red) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, m, 64));
    mx = fmaxf(mx, __shfl_xor(mx, m, 64));
  }
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[2 * wave] = mn;
    red[2 * wave + 1] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) {
      mn = fminf(mn, red[2 * i]);
      mx = fmaxf(mx, red[2 * i + 1]);
    }
    *out = make_float2(mn, mx);
  }
}
}  // namespace cdn
#endif

#define CDN_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return cdn::fail(code, __VA_ARGS__); \
  } while (0)
