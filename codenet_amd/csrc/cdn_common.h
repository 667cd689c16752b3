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

}  // namespace cdn

#define CDN_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return cdn::fail(code, __VA_ARGS__); \
  } while (0)
