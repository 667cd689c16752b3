// codenet_fused.hip -- one CoDeNet up-sampling stage as a fused kernel schedule (gfx950).
//
// What the reference runs per stage (W4A8: quant_modules.py:668-671 + quantize_model.py:79-81;
// fp32: modules/dcn_deform_conv.py:323-330 + shufflenetv2_dcn.py:303-308):
//     conv_scale -> Hardtanh -> [QuantAct] -> deform depthwise 3x3 -> [QuantAct] ->
//     conv_channel (+BN) -> ReLU -> [QuantAct] -> Upsample x2
// as ~12 framework ops with every tensor round-tripping through memory.  Here a stage is
//     scale (+min/max)  ->  [range update]  ->  gather/depthwise (+min/max)  ->  [range update]
//     ->  pointwise MFMA (+bias/BN, ReLU, min/max)  ->  [range update]
// with * QuantAct min/max reductions folded into the producing kernel's epilogue,
//      * fake-quantisation applied by the CONSUMER while it loads (same fp32 expression, so the
//        values are bit-identical to materialising them),
//      * the nearest x2 up-sampling folded into the consumer's addressing (the stage input is
//        kept at half resolution; scale prediction runs at half resolution: 4x less work),
//      * channels-last intermediates so every wave access is a contiguous row.
//
// gather/depthwise kernel (dw2): lane <-> 4 channels, 16 (or 8) lanes <-> one output pixel.
// The low-resolution input plane of CCH channels lives in LDS as [cell][CCH] rows with a zero
// border; a tap corner is ONE ds_read_b128 per lane.  With CCH = 64 a row is exactly the 64 LDS
// banks, so bank == channel and the read is conflict-free for ANY data-dependent cell -- the
// per-lane-column conflicts of the NCHW kernel (codenet_stage.hip) cannot occur.
#include "cdn_common.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

// In-kernel phase stamps for tools/probes/probe_dw.hip (which includes this file with
// -DCDN_STAMPS): thread 0 of every workgroup records s_memrealtime (100 MHz) at phase boundaries.
#ifdef CDN_STAMPS
// regions: 0 scale, 1 gather, 2 pointwise; 2048 workgroups x 8 stamps each
__device__ unsigned long long cdn_stamps[3 * 2048 * 8 + 64];   // + per-wave gather end times of workgroup 0
#define CDN_STAMPR(R, I)                                                                 \
  do {                                                                                   \
    if (threadIdx.x == 0)                                                                \
      cdn_stamps[(R) * 16384 + (((blockIdx.y * gridDim.x + blockIdx.x) & 2047) * 8 + (I))] = \
          __builtin_amdgcn_s_memrealtime();                                              \
  } while (0)
extern "C" int cdn_debug_read_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(cdn_stamps), sizeof(unsigned long long) * (3 * 2048 * 8 + 64));
}
extern "C" int cdn_debug_clear_stamps(void) {
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(cdn_stamps)) != hipSuccess) return 1;
  return (int)hipMemset(p, 0, sizeof(unsigned long long) * (3 * 2048 * 8 + 64));
}
#else
#define CDN_STAMPR(R, I) do { } while (0)
#endif
#ifdef CDN_STAMPS
#define CDN_STAMP_WAVE()                                                                   \
  do {                                                                                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.y == 0)                     \
      cdn_stamps[3 * 16384 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();       \
  } while (0)
#else
#define CDN_STAMP_WAVE() do { } while (0)
#endif
#define CDN_STAMP(I) CDN_STAMPR(1, I)

namespace {

using cdn::fake_quant;

struct Axis {
  int i0;
  float w0, w1;
  bool ok;
};

// Same fp32 position arithmetic as the reference pipeline `h_in + i*dil + anchor*(s-1)`
// (_kernel.cu:226 with modules/dcn_deform_conv.py:325); see codenet_stage.hip.
__device__ __forceinline__ Axis make_axis(int base, float off, int size) {
  Axis a;
  const float pos = (float)base + off;
  a.ok = pos > -1.0f && pos < (float)size;
  const float fl = floorf(pos);
  a.i0 = (int)fl;
  const float l = pos - fl;
  a.w1 = l;
  a.w0 = 1.0f - l;
  if (!a.ok) {
    a.i0 = 0;
    a.w0 = 0.0f;
    a.w1 = 0.0f;
  }
  return a;
}

// ---- int8 activation codes in HBM (frozen-range schedule, codenet_frozen.hip) -------------------------
// A frozen QuantAct (running_stat = False, quant_modules.py:203-219 skipped) has a fixed (scale, zp), so its
// output can cross HBM as ONE byte per element: the code q = round(scale*x - zp) itself (quant_utils.py:33-41,
// 60-75), which lies in [-128,127] for every x inside the frozen range; the value every consumer sees is
// (q + zp) / scale, exactly what the fp32 schedule materialises (the level L = q + zp is NOT byte sized: it runs
// from round(scale*x_min) to round(scale*x_min) + 255).  The reference does not clamp q (quant_utils.py:193-200);
// a byte must: a code outside
// [-128,127] is saturated AND reported through a device flag -- the caller then recomputes that batch on the
// fp32 schedule.  Rounding: the 1.5*2^23 trick, as in pwi8_kernel::ucode (round-half-even of the reference's
// two-rounding expression); values too large for the trick land far outside int8 and are flagged as well.
struct Code8 {
  float qs, qz;
};
using BadMask = int;     // per-lane: non-zero when some code of this lane saturated
__device__ __forceinline__ Code8 make_code8(const unsigned *state, BadMask &bad) {
  Code8 c;
  c.qs = reinterpret_cast<const float *>(state)[2];
  c.qz = reinterpret_cast<const float *>(state)[3];
  if (!(fabsf(c.qz) < 4.0e6f)) bad = 1;          // degenerate range (zp must stay an exactly representable integer)
  return c;
}
__device__ __forceinline__ int act_code8(float v, const Code8 &c, BadMask &bad) {
#pragma clang fp contract(off)
  const float y_p = c.qs * v;      // (plain operators under fp contract(off): two roundings, cdn_common.h)
  const float y = (y_p - c.qz) + 12582912.0f;
  const int a = (int)__float_as_uint(y) - 0x4B400000;      // rint(scale*v - zp)
  const int s = min(max(a, -128), 127);
  bad |= a ^ s;          // (non-zero iff the clamp changed the code: one xor + one or, no compare)
  return s;
}
__device__ __forceinline__ unsigned pack_code8(const float4 &v, const Code8 &c, BadMask &bad) {
  return (unsigned)(act_code8(v.x, c, bad) & 0xff) | ((unsigned)(act_code8(v.y, c, bad) & 0xff) << 8) |
         ((unsigned)(act_code8(v.z, c, bad) & 0xff) << 16) | ((unsigned)act_code8(v.w, c, bad) << 24);
}
// Workgroups are dealt to the 8 XCDs round-robin in launch order, and each XCD has its own L2.  With byte
// tensors a 32- or 64-channel chunk is 32 / 64 bytes of every 128-byte line, so the chunk workgroups of ONE image
// must share an L2 or every line crosses the fabric once per chunk (PMC: 33 MB read for the 8 MB stage-2 input,
// and partial-line writes of d from different XCDs).  Remap (chunk, image) so that XCD k walks the images
// [k*N/8, (k+1)*N/8) chunk by chunk; identity when the grid is not a multiple of 8.  Scalar arithmetic only.
__device__ __forceinline__ void xcd_remap(int &chunk, int &n) {
  const int nch = gridDim.x, total = gridDim.x * gridDim.y;
  if (total & 7) return;
  const int id = blockIdx.y * nch + blockIdx.x;
  const int item = (id & 7) * (total >> 3) + (id >> 3);
  n = item / nch;
  chunk = item - n * nch;
}

// running extremes of a float4 of accumulators: two three-operand instructions per side.  (fminf / fmaxf make the
// compiler quiet each operand first -- v_max_f32 x, x -- which was 16 of the 40 min/max instructions per gather step;
// the operands here are fma results, never signalling NaNs.)
__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// (No NaN flag here: the gather kernels are VALU-bound and a compare per output pair cost 15 % of them -- measured.  A NaN
// of d comes from a NaN of x, the weights or s: the gather kernels test what they STAGE instead -- memory-bound phases --
// and a tensor that is NaN throughout, the usual state behind a poisoned range, reduces to the empty pair (+inf, -inf),
// which the range update reads as NaN: cdn_common.h.)
__device__ __forceinline__ void track4(const float4 &a, float &mn, float &mn2, float &mx, float &mx2) {
  mn = min3f(mn, a.x, a.y);
  mn2 = min3f(mn2, a.z, a.w);
  mx = max3f(mx, a.x, a.y);
  mx2 = max3f(mx2, a.z, a.w);
}

// four stored codes -> the fake-quantised values (q + zp) / scale (Markstein division, bit-identical to
// cdn::fake_quant_r of the pre-quantisation value that produced the code)
__device__ __forceinline__ float4 unpack_code8(unsigned u, float scale, float zp, float r) {
  float4 t;
  const float l0 = __fadd_rn((float)(int)(signed char)(u & 0xff), zp), l1 = __fadd_rn((float)(int)(signed char)((u >> 8) & 0xff), zp);
  const float l2 = __fadd_rn((float)(int)(signed char)((u >> 16) & 0xff), zp), l3 = __fadd_rn((float)(int)(signed char)(u >> 24), zp);
  float q0 = __fmul_rn(l0, r); t.x = fmaf(fmaf(-q0, scale, l0), r, q0);
  q0 = __fmul_rn(l1, r); t.y = fmaf(fmaf(-q0, scale, l1), r, q0);
  q0 = __fmul_rn(l2, r); t.z = fmaf(fmaf(-q0, scale, l2), r, q0);
  q0 = __fmul_rn(l3, r); t.w = fmaf(fmaf(-q0, scale, l3), r, q0);
  return t;
}

// ------------------------------------------------------------------------------------------
// scale, NCHW input (stage 0 of the model: x comes from the PyTorch backbone).
// s[n,p] = clamp(b + sum_c w[c]*x[n,c,p]); block min/max -> state.
// ------------------------------------------------------------------------------------------
constexpr int kScaleWaves = 16;
__global__ void __launch_bounds__(kScaleWaves * 64)
scale_nchw_kernel(const float *__restrict__ x, const float *__restrict__ w,
                  const float *__restrict__ b, float *__restrict__ s, float2 *mm, cdn::QUpdate qu,
                  int C, int HW, float lo, float hi) {
  CDN_STAMPR(0, 0);
  // 16 waves x 64 pixels: wave v reduces channels v, v+16, ... with 4 loads in flight per lane
  // (~16 KB of 256-byte rows in flight per workgroup) -- the HBM-bound C -> 1 reduction.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const int n = blockIdx.y;
  const bool live = p < HW;
  const float *xp = x + (long)n * C * HW + (live ? p : 0);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int c = wave;
  constexpr int S = kScaleWaves;
  // 16 loads in flight per lane (one 16-wave workgroup per CU: 8 in flight left the kernel at 2.7 TB/s); the
  // accumulation order is that of the 4-wide loop below (a0 takes c, c + 4S, ...), so the sums are bit-identical
  // to it whatever the unroll
  for (; c + 15 * S < C; c += 16 * S) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = xp[(long)(c + u * S) * HW];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      a0 = fmaf(w[c + u * S], v[u], a0);
      a1 = fmaf(w[c + (u + 1) * S], v[u + 1], a1);
      a2 = fmaf(w[c + (u + 2) * S], v[u + 2], a2);
      a3 = fmaf(w[c + (u + 3) * S], v[u + 3], a3);
    }
  }
  for (; c + 3 * S < C; c += 4 * S) {
    const float v0 = xp[(long)c * HW], v1 = xp[(long)(c + S) * HW];
    const float v2 = xp[(long)(c + 2 * S) * HW], v3 = xp[(long)(c + 3 * S) * HW];
    a0 = fmaf(w[c], v0, a0);
    a1 = fmaf(w[c + S], v1, a1);
    a2 = fmaf(w[c + 2 * S], v2, a2);
    a3 = fmaf(w[c + 3 * S], v3, a3);
  }
  for (; c < C; c += S) a0 = fmaf(w[c], xp[(long)c * HW], a0);
  __shared__ float red[kScaleWaves][64];
  red[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  if (wave == 0) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kScaleWaves; ++i) v += red[i][lane];
    v += b ? b[0] : 0.0f;
    v = cdn::clamp_keep_nan(v, lo, hi);
    if (live) {
      s[(long)n * HW + p] = v;
      mn = mx = v;
      has_nan = (v != v);
    }
  }
  CDN_STAMPR(0, 2);
  if (mm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), mm, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, &red[0][0]);
  CDN_STAMPR(0, 3);
}

// ------------------------------------------------------------------------------------------
// scale, channels-last input x[pix][C] (stored resolution), optional fake-quant on load.
// One wave per pixel step; lane covers channels 4*lane + 256*j (float4 loads: C % 4 == 0).
// ------------------------------------------------------------------------------------------
template <bool XQ, bool X8 = false>      // X8: x is a byte tensor of codes of the quantiser xq (see Code8)
__global__ void __launch_bounds__(256)
scale_nhwc_kernel(const float *__restrict__ x, const unsigned *__restrict__ xq,
                  const float *__restrict__ w, const float *__restrict__ b, float *__restrict__ s,
                  float2 *mm, cdn::QUpdate qu, int C, long npix, float lo, float hi) {
  __shared__ float red[12];
  CDN_STAMPR(0, 0);
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float qs = 1.f, qz = 0.f, qr_ = 1.f;
  if (XQ) {
    qs = reinterpret_cast<const float *>(xq)[2];
    qr_ = __fdiv_rn(1.0f, qs);   // Markstein division in fake_quant_r
    qz = reinterpret_cast<const float *>(xq)[3];
  }
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  // one pixel's dot product (a wave: 64 lanes x 4 channels per round); `take` loads, `dot` consumes -- two pixels'
  // loads are issued before either is consumed (round 5: at stage 1 every wave has exactly two pixels, and the
  // second load used to wait behind the first one's reduction: one memory latency instead of two per launch)
  auto take = [&](long p, int c) -> float4 {
    if (X8)
      return unpack_code8(*reinterpret_cast<const unsigned *>(reinterpret_cast<const signed char *>(x) + p * C + c), qs, qz,
                          qr_);
    return *reinterpret_cast<const float4 *>(x + p * C + c);
  };
  auto dot = [&](float4 v, int c, float acc) -> float {
    const float4 ww = *reinterpret_cast<const float4 *>(w + c);
    if (XQ && !X8) {
      v.x = cdn::fake_quant_r(v.x, qs, qz, qr_);
      v.y = cdn::fake_quant_r(v.y, qs, qz, qr_);
      v.z = cdn::fake_quant_r(v.z, qs, qz, qr_);
      v.w = cdn::fake_quant_r(v.w, qs, qz, qr_);
    }
    acc = fmaf(ww.x, v.x, acc);
    acc = fmaf(ww.y, v.y, acc);
    acc = fmaf(ww.z, v.z, acc);
    return fmaf(ww.w, v.w, acc);
  };
  auto finish = [&](float acc, long p) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    float v = acc + (b ? b[0] : 0.0f);
    v = cdn::clamp_keep_nan(v, lo, hi);
    if (lane == 0) s[p] = v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    has_nan |= (v != v);
  };
  long p = wave;
  for (; p + nwaves < npix; p += 2 * nwaves) {
    const long p2 = p + nwaves;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const float4 v0 = take(p, c), v1 = take(p2, c);
      a0 = dot(v0, c, a0);
      a1 = dot(v1, c, a1);
    }
    finish(a0, p);
    finish(a1, p2);
  }
  if (p < npix) {
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) acc = dot(take(p, c), c, acc);
    finish(acc, p);
  }
  CDN_STAMPR(0, 2);
  if (mm) cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), mm, blockIdx.x, gridDim.x, qu, red);
  CDN_STAMPR(0, 3);
}

// ------------------------------------------------------------------------------------------
// scale, channels-last input, C <= 256: one workgroup = 64 consecutive pixels.  Phase 1 streams the
// tile's 64*C floats as fully coalesced 16-byte loads, ALL of a thread's loads (C/16 <= 16) in flight
// before the first use; each is fake-quantised, dotted with its 4 weights and parked in LDS
// part[pixel][C/4].  Phase 2: 4 threads per pixel sum the partials (conflict-free: row stride
// = 4 mod 32 banks).  Used for large planes only (see the launch site).
// ------------------------------------------------------------------------------------------
constexpr int kScaleTilePix = 64;
template <bool XQ, bool X8 = false>
__global__ void __launch_bounds__(256)
scale_nhwc_tile_kernel(const float *__restrict__ x, const unsigned *__restrict__ xq,
                       const float *__restrict__ w, const float *__restrict__ b,
                       float *__restrict__ s, float2 *mm, cdn::QUpdate qu, int C, long npix,
                       float lo, float hi) {
  extern __shared__ float part[];     // [64][LD], then 16 floats of reduction scratch
  CDN_STAMPR(0, 0);
  const int CQ = C >> 2;
  const int LD = ((CQ + 31) & ~31) + 4;
  float *red = part + kScaleTilePix * LD;
  const long pix0 = (long)blockIdx.x * kScaleTilePix;
  const int tile_pix = (int)min((long)kScaleTilePix, npix - pix0);
  const int total = tile_pix * CQ;    // float4 items of this tile
  float qs = 1.f, qz = 0.f, qr_ = 1.f;
  if (XQ) {
    qs = reinterpret_cast<const float *>(xq)[2];
    qr_ = __fdiv_rn(1.0f, qs);   // Markstein division in fake_quant_r
    qz = reinterpret_cast<const float *>(xq)[3];
  }
  const float4 *xt = reinterpret_cast<const float4 *>(x + (X8 ? 0 : pix0 * C));
  const unsigned *xt8 = reinterpret_cast<const unsigned *>(reinterpret_cast<const signed char *>(x) + pix0 * C);
  const float4 *w4 = reinterpret_cast<const float4 *>(w);
  constexpr int U = 16;               // 64 * 64 / 256: every load of a C = 256 tile in flight
  float4 v[U];
  unsigned v8[X8 ? U : 1];
#pragma unroll
  for (int i = 0; i < U; ++i) {
    const int q = threadIdx.x + 256 * i;
    if (X8) v8[X8 ? i : 0] = q < total ? xt8[q] : 0u;
    else v[i] = q < total ? xt[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < U; ++i) {
    const int q = threadIdx.x + 256 * i;
    if (q < total) {
      const int pix = q / CQ, cq = q - pix * CQ;
      const float4 ww = w4[cq];
      float4 t = X8 ? unpack_code8(v8[X8 ? i : 0], qs, qz, qr_) : v[i];
      if (XQ && !X8) {
        t.x = cdn::fake_quant_r(t.x, qs, qz, qr_);
        t.y = cdn::fake_quant_r(t.y, qs, qz, qr_);
        t.z = cdn::fake_quant_r(t.z, qs, qz, qr_);
        t.w = cdn::fake_quant_r(t.w, qs, qz, qr_);
      }
      part[pix * LD + cq] = fmaf(ww.w, t.w, fmaf(ww.z, t.z, fmaf(ww.y, t.y, ww.x * t.x)));
    }
  }
  __syncthreads();
  const int pix = threadIdx.x >> 2, k = threadIdx.x & 3;
  float acc = 0.f;
  if (pix < tile_pix)
    for (int i = k; i < CQ; i += 4) acc += part[pix * LD + i];
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  if (pix < tile_pix) {
    float r = acc + (b ? b[0] : 0.0f);
    r = cdn::clamp_keep_nan(r, lo, hi);
    if (k == 0) s[pix0 + pix] = r;
    mn = mx = r;
    has_nan = (r != r);
  }
  CDN_STAMPR(0, 2);
  if (mm) cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), mm, blockIdx.x, gridDim.x, qu, red);
  CDN_STAMPR(0, 3);
}

// ------------------------------------------------------------------------------------------
// dw2: gather + depthwise 3x3, lanes <-> channels.  See file header.
//   x        stage input at STORED resolution (Hl x Wl = (H>>up) x (W>>up)):
//            NHWC_IN ? [n][Hl*Wl][C] : [n][C][Hl*Wl]
//   s_raw    [n][Hl*Wl]  (scale at stored resolution; up-sampling replicates it)
//   d        [n][H*W][C] channels-last output at stage resolution
// ------------------------------------------------------------------------------------------
// (row_fetch / fetch_record / owner_item: the DPP record fetch lives in cdn_common.h, shared with the gather backward)
using cdn::fetch_record;
using cdn::owner_item;
#ifndef CDN_DPP16
#define CDN_DPP16 1      // DPP fetch when an item spans a whole 16-lane row (CCH = 64)
#endif
#ifndef CDN_DPP8
#define CDN_DPP8 1       // DPP fetch for 8-lane items (CCH = 32): two movs per value (1.8 us per step ahead of
                         // ds_bpermute once the tap math was factored; it made no difference before)
#endif
template <int LPP>
constexpr bool use_dpp() { return (LPP == 16 && CDN_DPP16) || (LPP == 8 && CDN_DPP8); }

constexpr int kDw2MaxThreads = 1024;   // workgroup size is chosen per launch (512 or 1024)

// The gather + depthwise of one staged plane:
// img = [(Hl+1)*(Wl+1)][CCH] cells with the zero row / column, wl = [CCH][9] weights, sl = scale plane.
// OUT8: d is a byte tensor of codes (see Code8) instead of fp32; mn / mx are not tracked.
// The input is at the resolution of the output (up-sampled inputs go to dw2u_kernel): img has a zero cell in front
// of it, so the corner columns i0, i0 + 1 of a tap are adjacent cells for every i0 in [-1, W - 1].
template <int CCH, bool OUT8>
__device__ __forceinline__ void dw2_gather(const float4 *img, const float *wl, const float *sl,
                                           float *__restrict__ d, int n, int c0, int C, int H, int W,
                                           int kWaves, float &mn, float &mx, bool &has_nan,
                                           const Code8 *c8 = nullptr, BadMask *bad = nullptr) {
  constexpr bool ADJ = true;
  constexpr int up = 0;
  constexpr int LPP = CCH / 4;     // lanes per pixel
  constexpr int PPW = 64 / LPP;    // pixels per wave step
  const int tid = threadIdx.x;
  const int Hl = H >> up, Wl = W >> up;
  const int HW = H * W;
  const int Wc = Wl + 1;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lane = tid & 63, wave = tid >> 6;
  const int cq = lane % LPP, sub = lane / LPP;
  float wk[9][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k][e] = wl[(cq * 4 + e) * 9 + k];

  // byte offsets: row part + column part; out-of-image coordinates select the zero row / column
  const int rstride = Wc * LPP * 16;
  auto row_off = [&](int yy) { return (((unsigned)yy < (unsigned)H) ? (yy >> up) : Hl) * rstride; };
  constexpr int kCell = LPP * 16;
  auto col_off = [&](int xx) {
    if (ADJ) return xx * kCell;                 // xx in [-1, W - 1]: as it is
    return (((unsigned)xx < (unsigned)W) ? (xx >> up) : Wl) * kCell;
  };
  const char *imgb = reinterpret_cast<const char *>(img) + cq * 16;

  const bool vec_store = ((C & 3) == 0);
  // The kernel is VALU-issue bound and the tap geometry is the bulk of the VALU work, so it is
  // computed ONCE per pixel: in the geometry phase lane i owns pixel pb+i (64 different pixels
  // per wave instruction instead of 16 lanes repeating the same pixel); in the gather phase the
  // LPP lanes of an output pixel fetch that pixel's 16-word record from its owner lane with DPP row broadcasts
  // (fetch_record; no LDS storage, no LDS pipeline) and only add offsets, mix and accumulate.
  // every wave owns a contiguous pixel range (a multiple of PPW), so all waves gather even when the
  // plane has fewer than 64 pixels per wave (stage 0: 256 pixels over 8 waves)
  const int ppw = ((HW + kWaves - 1) / kWaves + PPW - 1) / PPW * PPW;
  const int p_begin = wave * ppw, p_end = min(HW, p_begin + ppw);
  for (int pb = p_begin; pb < p_end; pb += 64) {
    // ---- geometry phase ------------------------------------------------------------------
    int g_r[5], g_c[5];
    float g_w[8];
    {
      const int p = min(pb + (use_dpp<LPP>() ? owner_item<LPP>(lane) : lane), HW - 1);
      const int h = p / W, w = p - h * W;
      const float t = sl[(h >> up) * Wl + (w >> up)] - 1.0f;
      const Axis ya = make_axis(h - 1, -t, H), yb = make_axis(h + 1, t, H);
      const Axis xa = make_axis(w - 1, -t, W), xb = make_axis(w + 1, t, W);
      g_r[0] = row_off(ya.i0); g_r[1] = row_off(ya.i0 + 1);
      g_r[2] = row_off(yb.i0); g_r[3] = row_off(yb.i0 + 1);
      g_r[4] = row_off(h);
      g_c[0] = col_off(xa.i0); g_c[1] = ADJ ? 0 : col_off(xa.i0 + 1);
      g_c[2] = col_off(xb.i0); g_c[3] = ADJ ? 0 : col_off(xb.i0 + 1);
      g_c[4] = col_off(w);
      g_w[0] = ya.w0; g_w[1] = ya.w1; g_w[2] = yb.w0; g_w[3] = yb.w1;
      g_w[4] = xa.w0; g_w[5] = xa.w1; g_w[6] = xb.w0; g_w[7] = xb.w1;
    }
    // ---- gather phase: PPW pixels per step ---------------------------------------------------
#pragma unroll 1
    for (int j = 0; j < 64 / PPW; ++j) {
      const int src = j * PPW + sub;          // owner lane of this lane's pixel
      const int p = pb + src;
      int r[5], c[5];
      float wt[8];
#if defined(CDN_DIAG) && CDN_DIAG == 1   // diagnostic build: no cross-lane fetch (wrong results)
#pragma unroll
      for (int q = 0; q < 5; ++q) { r[q] = g_r[q]; c[q] = g_c[q]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) wt[q] = g_w[q];
      (void)src;
#else
      if (use_dpp<LPP>() && ADJ) {
        int gi[8] = {g_r[0], g_r[1], g_r[2], g_r[3], g_r[4], g_c[0], g_c[2], g_c[4]}, oi[8];
        fetch_record<LPP == 8>(j, gi, g_w, oi, wt);
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = oi[q];
        c[0] = oi[5]; c[2] = oi[6]; c[4] = oi[7];
      } else if (use_dpp<LPP>()) {
        int gi[10], oi[10];
#pragma unroll
        for (int q = 0; q < 5; ++q) { gi[q] = g_r[q]; gi[5 + q] = g_c[q]; }
        fetch_record<LPP == 8>(j, gi, g_w, oi, wt);
#pragma unroll
        for (int q = 0; q < 5; ++q) { r[q] = oi[q]; c[q] = oi[5 + q]; }
      } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          r[q] = __shfl(g_r[q], src, 64);
          if (!ADJ || !(q & 1)) c[q] = __shfl(g_c[q], src, 64);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) wt[q] = __shfl(g_w[q], src, 64);
      }
      if (ADJ) { c[1] = c[0] + kCell; c[3] = c[2] + kCell; }   // (folded into the reads' immediate offsets)
#endif
      if (pb + j * PPW >= p_end) break;       // wave-uniform: whole step beyond this wave's range
      float4 acc = z4;
#if defined(CDN_DIAG) && CDN_DIAG == 2   // diagnostic build: no LDS cell reads (wrong results)
#define CDN_RD(O) make_float4(__int_as_float(O), 1.f, 2.f, 3.f)
#else
#define CDN_RD(O) (*reinterpret_cast<const float4 *>(imgb + (O)))
#endif
#define CDN_WACC(K, TV)                     \
  acc.x = fmaf(wk[K][0], TV.x, acc.x);      \
  acc.y = fmaf(wk[K][1], TV.y, acc.y);      \
  acc.z = fmaf(wk[K][2], TV.z, acc.z);      \
  acc.w = fmaf(wk[K][3], TV.w, acc.w);
      // the 25 cell offsets of the step: taps 0..8 in order, corner taps (r0c0, r0c1, r1c0, r1c1)
      const int o[25] = {r[0] + c[0], r[0] + c[1], r[1] + c[0], r[1] + c[1],      // tap 0
                         r[0] + c[4], r[1] + c[4],                                // tap 1
                         r[0] + c[2], r[0] + c[3], r[1] + c[2], r[1] + c[3],      // tap 2
                         r[4] + c[0], r[4] + c[1],                                // tap 3
                         r[4] + c[4],                                             // tap 4
                         r[4] + c[2], r[4] + c[3],                                // tap 5
                         r[2] + c[0], r[2] + c[1], r[3] + c[0], r[3] + c[1],      // tap 6
                         r[2] + c[4], r[3] + c[4],                                // tap 7
                         r[2] + c[2], r[2] + c[3], r[3] + c[2], r[3] + c[3]};     // tap 8
      // Reads are issued at their use: under the 128-VGPR cap of the 16-waves-per-CU launch the compiler keeps 2-4 in
      // flight.  (Tried and removed: all 25 reads of a step in flight at <= 256 threads / 256 VGPRs, and a rolling
      // window of 13 -- both slower in the pipeline)
#define CDN_VAL(I) CDN_RD(o[I])
      // corner taps: cells I..I+3, axis weights (Y0,Y1) x (X0,X1)
#define CDN_TAP4(I, Y0, Y1, X0, X1, K)                                    \
  {                                                                       \
    const float4 v00 = CDN_VAL(I), v01 = CDN_VAL(I + 1);                  \
    const float4 v10 = CDN_VAL(I + 2), v11 = CDN_VAL(I + 3);              \
    const float w00 = Y0 * X0, w01 = Y0 * X1, w10 = Y1 * X0, w11 = Y1 * X1; \
    float4 tv;                                                            \
    tv.x = ((w00 * v00.x + w01 * v01.x) + w10 * v10.x) + w11 * v11.x;     \
    tv.y = ((w00 * v00.y + w01 * v01.y) + w10 * v10.y) + w11 * v11.y;     \
    tv.z = ((w00 * v00.z + w01 * v01.z) + w10 * v10.z) + w11 * v11.z;     \
    tv.w = ((w00 * v00.w + w01 * v01.w) + w10 * v10.w) + w11 * v11.w;     \
    CDN_WACC(K, tv)                                                       \
  }
      // edge taps: one axis exact, cells I, I+1 with weights (A0, A1)
#define CDN_TAP2(I, A0, A1, K)                      \
  {                                                 \
    const float4 v0 = CDN_VAL(I), v1 = CDN_VAL(I + 1); \
    float4 tv;                                      \
    tv.x = A0 * v0.x + A1 * v1.x;                   \
    tv.y = A0 * v0.y + A1 * v1.y;                   \
    tv.z = A0 * v0.z + A1 * v1.z;                   \
    tv.w = A0 * v0.w + A1 * v1.w;                   \
    CDN_WACC(K, tv)                                 \
  }
      CDN_TAP4(0, wt[0], wt[1], wt[4], wt[5], 0)
      CDN_TAP2(4, wt[0], wt[1], 1)
      CDN_TAP4(6, wt[0], wt[1], wt[6], wt[7], 2)
      CDN_TAP2(10, wt[4], wt[5], 3)
      {
        const float4 vc = CDN_VAL(12);
        CDN_WACC(4, vc)
      }
      CDN_TAP2(13, wt[6], wt[7], 5)
      CDN_TAP4(15, wt[2], wt[3], wt[4], wt[5], 6)
      CDN_TAP2(19, wt[2], wt[3], 7)
      CDN_TAP4(21, wt[2], wt[3], wt[6], wt[7], 8)
#undef CDN_VAL
#undef CDN_TAP4
#undef CDN_TAP2
#undef CDN_WACC
#undef CDN_RD
      if (OUT8) {
        if (p < p_end && c0 + cq * 4 + 3 < C)       // (C % 4 == 0 is checked by the host)
          reinterpret_cast<unsigned *>(reinterpret_cast<signed char *>(d) + ((long)n * HW + p) * C + c0)[cq] =
              pack_code8(acc, *c8, *bad);
      } else if (p < p_end) {
        float *dp = d + ((long)n * HW + p) * C + c0 + cq * 4;
        const int cbase = c0 + cq * 4;
        if (vec_store && cbase + 3 < C) {
#if defined(CDN_DIAG) && CDN_DIAG == 5   // diagnostic build: no global store of d (wrong results)
          if (acc.x == 1234.5f)
#endif
          *reinterpret_cast<float4 *>(dp) = acc;
          track4(acc, mn, mn, mx, mx);      // one pair: the kernel sits at its 128-VGPR cap
        } else {
          const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cbase + e < C) {
              dp[e] = av[e];
              mn = fminf(mn, av[e]);
              mx = fmaxf(mx, av[e]);
              has_nan |= (av[e] != av[e]);
            }
        }
      }
    }
  }
}

// X8 (with NHWC_IN and XQ): x is a byte tensor of codes of the quantiser xq.  OUT8: d is a byte tensor of codes
// of the quantiser qu.state (frozen: no range tracking), dmm is reinterpreted as the overflow flag word.
template <int CCH, bool NHWC_IN, bool XQ, bool SQ, int MAXT, bool X8 = false, bool OUT8 = false>
__global__ void __launch_bounds__(MAXT)
dw2_kernel(const float *__restrict__ x, const unsigned *__restrict__ xq,
           const float *__restrict__ s_raw, const unsigned *__restrict__ sq,
           const float *__restrict__ wd, float *__restrict__ d, float2 *dmm, cdn::QUpdate qu,
           int C, int H, int W, int up) {
  static_assert(!X8 || (NHWC_IN && XQ), "codes come channels-last with their quantiser state");
  // LDS: [(Hl+1)*(Wl+1)][CCH] image cells; row Hl and column Wl are ZERO and every out-of-image
  // corner coordinate maps there (per-corner zeroing of the reference, _kernel.cu:97-108) -- a
  // cell address is just row offset + column offset, no bounds test and no clamp per corner.
  // Then the chunk's depthwise weights [CCH][9], the scale plane [Hl*Wl], reduction scratch.
  // In front of the image: ONE zero cell, so that cell (r, -1) -- the cell in front of row r -- is always zero
  // (the zero column of row r - 1, or that leading cell): with up == 0 the two corner columns i0, i0 + 1 of a tap
  // are adjacent cells for every i0 in [-1, W - 1] and the gather fetches one column offset per tap class.
  extern __shared__ float4 img_lds[];
  CDN_STAMP(0);
  constexpr int LPP = CCH / 4;     // lanes per pixel
  float4 *img = img_lds + LPP;
  const int kDw2Threads = blockDim.x, kWaves = kDw2Threads / 64;
  const int Hl = H >> up, Wl = W >> up;
  const int HWl = Hl * Wl;
  int n = blockIdx.y, chunk = blockIdx.x;
  if (X8 || OUT8) xcd_remap(chunk, n);
  const int c0 = chunk * CCH;
  const int tid = threadIdx.x;
  const int Wc = Wl + 1;                       // cells per LDS row
  const int cells = (Hl + 1) * Wc;
  float *wl = reinterpret_cast<float *>(img + (size_t)cells * LPP);
  float *sl = wl + CCH * 9;
  float *red = sl + HWl;
  float xs = 1.f, xz = 0.f, ss = 1.f, sz = 0.f, xr_ = 1.f;
  if (XQ) {
    xs = reinterpret_cast<const float *>(xq)[2];
    xr_ = __fdiv_rn(1.0f, xs);   // Markstein division in fake_quant_r
    xz = reinterpret_cast<const float *>(xq)[3];
  }
  if (SQ) {
    ss = reinterpret_cast<const float *>(sq)[2];
    sz = reinterpret_cast<const float *>(sq)[3];
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = tid; q < (Wc + Hl) * LPP; q += kDw2Threads) {   // zero row, then zero column
    const int i = q / LPP;
    const int cell = i < Wc ? Hl * Wc + i : (i - Wc) * Wc + Wl;
    img[cell * LPP + (q % LPP)] = z4;
  }
  if (tid < LPP) img_lds[tid] = z4;                            // the leading zero cell
  // the chunk's depthwise weights and the scale plane are requested BEFORE the image (one element per thread;
  // stored to LDS after it), so their round trips overlap the image's instead of following it
  const float w_pre = (tid < CCH * 9 && c0 + tid / 9 < C) ? wd[(long)c0 * 9 + tid] : 0.0f;
  const float s_pre = tid < HWl ? s_raw[(long)n * HWl + tid] : 0.0f;
  // ---- stage the image -------------------------------------------------------------------
  // All of a thread's global loads of a batch are issued before the first one is used (kStageU in
  // flight per thread); a plain loop leaves ONE dependent load per thread in flight and the
  // staging then costs a full HBM round trip per iteration (measured: 21 of 58 us at stage 0).
  constexpr int kStageU = 8;
  // a NaN of d comes from a NaN of what is staged here (x, the weights): tested in this memory-bound phase, not per output
  bool has_nan = (w_pre != w_pre);
  if (X8) {
    const signed char *xg = reinterpret_cast<const signed char *>(x) + (long)n * HWl * C + c0;
    const int total = HWl * LPP;
    for (int base = 0; base < total; base += kDw2Threads * kStageU) {
      unsigned v[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * kDw2Threads + tid;
        const int pix = q / LPP, cq = q % LPP;
        v[u] = 0u;
        if (q < total && c0 + cq * 4 + 3 < C) v[u] = *reinterpret_cast<const unsigned *>(xg + (long)pix * C + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * kDw2Threads + tid;
        const int pix = q / LPP, cq = q % LPP;
        if (q < total)
          img[((pix / Wl) * Wc + (pix % Wl)) * LPP + cq] =
              (c0 + cq * 4 + 3 < C) ? unpack_code8(v[u], xs, xz, xr_) : z4;
      }
    }
  } else if (NHWC_IN) {
    const float *xg = x + (long)n * HWl * C + c0;
    const int total = HWl * LPP;
    for (int base = 0; base < total; base += kDw2Threads * kStageU) {
      float4 v[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * kDw2Threads + tid;
        const int pix = q / LPP, cq = q % LPP;
        v[u] = z4;
        if (q < total && c0 + cq * 4 + 3 < C)
          v[u] = *reinterpret_cast<const float4 *>(xg + (long)pix * C + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * kDw2Threads + tid;
        const int pix = q / LPP, cq = q % LPP;
        if (q < total) {
          float4 t = v[u];
          if (XQ) {
            t.x = cdn::fake_quant_r(t.x, xs, xz, xr_);
            t.y = cdn::fake_quant_r(t.y, xs, xz, xr_);
            t.z = cdn::fake_quant_r(t.z, xs, xz, xr_);
            t.w = cdn::fake_quant_r(t.w, xs, xz, xr_);
          }
          has_nan |= __builtin_isunordered(t.x, t.y) | __builtin_isunordered(t.z, t.w);
          img[((pix / Wl) * Wc + (pix % Wl)) * LPP + cq] = t;
        }
      }
    }
  } else {
    // lane <-> channel so the four scalar LDS stores of a wave hit consecutive banks
    float *imgf = reinterpret_cast<float *>(img);
    const int quads = (HWl + 3) >> 2;
    const int total = quads * CCH;
    if ((HWl & 3) == 0) {
      for (int base = 0; base < total; base += kDw2Threads * kStageU) {
        float4 v[kStageU];
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
          const int q = base + u * kDw2Threads + tid;
          const int cl = q % CCH, j = q / CCH;
          const int c = min(c0 + cl, C - 1), jj = min(j, quads - 1);   // clamped: always a valid load
          v[u] = *reinterpret_cast<const float4 *>(x + ((long)n * C + c) * HWl + jj * 4);
        }
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
          const int q = base + u * kDw2Threads + tid;
          const int cl = q % CCH, j = q / CCH;
          if (q < total) {
            const bool live = c0 + cl < C;
            const float e4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int pix = j * 4 + e;
              const float t = XQ ? cdn::fake_quant_r(e4[e], xs, xz, xr_) : e4[e];
              has_nan |= live && (t != t);
              imgf[((pix / Wl) * Wc + (pix % Wl)) * CCH + cl] = live ? t : 0.0f;
            }
          }
        }
      }
    } else {
      for (int q = tid; q < total; q += kDw2Threads) {
        const int cl = q % CCH, j = q / CCH;
        const int c = c0 + cl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
          const float *xp = x + ((long)n * C + c) * HWl + j * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j * 4 + e < HWl) v[e] = xp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pix = j * 4 + e;
          if (pix < HWl) {
            const float t = XQ ? cdn::fake_quant_r(v[e], xs, xz, xr_) : v[e];
            has_nan |= (t != t);
            imgf[((pix / Wl) * Wc + (pix % Wl)) * CCH + cl] = t;
          }
        }
      }
    }
  }
  // ---- chunk weights (coalesced) and the (fake-quantised) scale plane: first element per thread prefetched ----
  if (tid < CCH * 9) wl[tid] = w_pre;
  for (int q = tid + kDw2Threads; q < CCH * 9; q += kDw2Threads)
    wl[q] = (c0 + q / 9 < C) ? wd[(long)c0 * 9 + q] : 0.0f;
  if (tid < HWl) sl[tid] = SQ ? fake_quant(s_pre, ss, sz) : s_pre;
  for (int q = tid + kDw2Threads; q < HWl; q += kDw2Threads) {
    float sv = s_raw[(long)n * HWl + q];
    if (SQ) sv = fake_quant(sv, ss, sz);
    sl[q] = sv;
  }
  CDN_STAMP(1);
  __syncthreads();
  CDN_STAMP(2);
  float mn = INFINITY, mx = -INFINITY;
  if (OUT8) {
    BadMask bad = 0;
    const Code8 c8 = make_code8(qu.state, bad);
    dw2_gather<CCH, true>(img, wl, sl, d, n, c0, C, H, W, kWaves, mn, mx, has_nan, &c8, &bad);
    if (bad) atomicOr(reinterpret_cast<unsigned *>(dmm), 1u);
    return;
  }
  dw2_gather<CCH, false>(img, wl, sl, d, n, c0, C, H, W, kWaves, mn, mx, has_nan);
  CDN_STAMP_WAVE();
  CDN_STAMP(3);
  if (dmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), dmm, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, red);
  CDN_STAMP(4);
}

// ------------------------------------------------------------------------------------------
// dw0p: PERSISTENT form of dw2_kernel<64> for an NCHW fp32 input at output resolution (stage 0 of the model),
// with the next item's input in flight by LDS-DMA while the current item is gathered.
//
// dw2_kernel runs stage 0 as two rounds of two 512-thread workgroups per CU that stage (HBM) and gather (LDS / VALU)
// in lock-step: HBM idles during the gather and the SIMDs idle during staging (in-kernel stamps).
// A register-staged double buffer cannot be built under the 128-VGPR cap of 16 waves per CU.  Here ONE 1024-thread
// workgroup per CU walks over its items (image n, 64-channel chunk):
//
//   raw   [64][HW] floats  <- global_load_lds_dwordx4 (no VGPRs): an item's 64 channel planes are ONE contiguous
//                             64*HW*4-byte block of the NCHW tensor; 1-KB pieces, each lane's 16 bytes = 4 pixels of a
//                             channel, stored at slot (quad ^ (channel & 15)) of the channel's row -- the XOR is
//                             applied to the per-lane SOURCE address (the DMA's LDS side is lane-linear), so that
//   img   [cell][64]       <- the transposing pass (lane <-> channel: ds_read_b128 of raw, four ds_write_b32 into the
//                             [cell][channel] image the gather wants) reads 16 different 16-byte bank groups per
//                             hardware lane group and writes 32 consecutive banks: conflict-free both ways;
//   wl, sl (two sets)      <- global_load_lds_dword: the chunk's depthwise weights and the image's scale plane.
//
//   per item:  wait for this item's DMA | barrier | raw -> img (+ fake-quant of s) | barrier |
//              issue the NEXT item's DMA into raw / the other (wl, sl) set | gather this item (dw2_gather, unchanged).
//
// Same arithmetic as dw2_kernel (same staged values, same gather routine): results are bit-identical to it.
// Needs W % 4 == 0, H*W a power of two >= 64 (whole 1-KB pieces per channel group, shifts instead of divisions in the
// per-step DMA issue), C % 64 == 0.
// ------------------------------------------------------------------------------------------
// LDS-DMA as inline asm, on purpose: with __builtin_amdgcn_global_load_lds the compiler orders every later LDS
// read of the wave behind the pending DMA (it cannot prove that the gather's image reads do not alias the raw
// buffer being filled) and emits s_waitcnt vmcnt(0) right after the issue -- the next item's transfer was then
// fully exposed (in-kernel stamps: 9.0 / 11.6 / 13.8 us per item with the DMA "in flight" against 4.5 us for the last
// item without one).  The asm form is invisible to that bookkeeping; completion is waited for explicitly
// (s_waitcnt vmcnt(0) + barrier at the top of every item).  M0 = wave-uniform LDS byte address of the piece, the
// lane's 16 (4) bytes land at M0 + lane * 16 (4); M0 is saved and restored (compiler-reserved register).
// (glds16 / glds4 / lds_addr_uniform: cdn_common.h, shared with the training path's pointwise kernel)
using cdn::glds16;
using cdn::glds4;
using cdn::lds_addr_uniform;

template <bool SQ, bool OUT8>
__global__ void __launch_bounds__(1024)
dw0p_kernel(const float *__restrict__ x, const float *__restrict__ s_raw, const unsigned *__restrict__ sq,
            const float *__restrict__ wd, float *__restrict__ d, float2 *dmm, cdn::QUpdate qu,
            int C, int H, int W, int nitems, int Cx) {
  // Cx = channels of x / wd; C = row length of d, a multiple of 64 >= Cx.  Cx < C (round 4: CoDeNet2x, 2153 = 33 * 64 +
  // 41): the last chunk is RAGGED -- its missing planes and weights are fetched from the chunk's LAST REAL channel
  // (clamped source addresses: no read outside the tensors, no predicated DMA), so the pad channels of d are exact
  // duplicates of channel Cx - 1: the {min, max} of d are unchanged and the pointwise kernel multiplies them by its
  // zero-padded weight codes.
  extern __shared__ float4 img_lds[];
  CDN_STAMP(0);
  constexpr int CCH = 64, LPP = 16, kWaves = 16;
  const int HW = H * W, Wc = W + 1, cells = (H + 1) * Wc;
  const int Q = HW >> 2;                      // 16-byte quads per channel plane
  const int nchunk = C / CCH;
  float4 *img = img_lds + LPP;
  float *raw = reinterpret_cast<float *>(img + (size_t)cells * LPP);
  float *wl0 = raw + CCH * HW;                // two sets of [64][9]
  float *sl0 = wl0 + 2 * CCH * 9;             // two sets of [HW]
  float *red = sl0 + 2 * HW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float ss = 1.f, sz = 0.f;
  if (SQ) {
    ss = reinterpret_cast<const float *>(sq)[2];
    sz = reinterpret_cast<const float *>(sq)[3];
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = tid; q < (Wc + H) * LPP; q += 1024) {   // zero row, then zero column: written once
    const int i = q / LPP;
    const int cell = i < Wc ? H * Wc + i : (i - Wc) * Wc + W;
    img[cell * LPP + (q % LPP)] = z4;
  }
  if (tid < LPP) img_lds[tid] = z4;                    // the leading zero cell

  const int npiece_w = CCH * 9 / 64, npiece_s = HW / 64;
  const unsigned raw_a = lds_addr_uniform(raw), wl_a = lds_addr_uniform(wl0), sl_a = lds_addr_uniform(sl0);
  // DMA of one item = Q 1-KB pieces of the raw buffer (wave w: pieces w, w + 16, ...) + the small pieces (weights,
  // scale plane).  issue_step(item, set, k): the wave's k-th raw piece, and with k == 0 its small piece.
  const int qsh = 31 - __builtin_clz(Q);                           // Q is a power of two (host check)
  auto issue_step = [&](int item, int set, int k) {
    const int n = item / nchunk, chunk = item - n * nchunk;
    const int piece = __builtin_amdgcn_readfirstlane(wave + kWaves * k);
    if (piece < Q) {                                               // Q pieces of 1 KB = 64 * HW * 4 bytes
      const int idx = piece * 64 + lane;                           // 16-byte slot of the raw buffer
      const int c = idx >> qsh, slot = idx & (Q - 1);
      const int cs = min(c, Cx - 1 - chunk * CCH);                 // ragged last chunk: the last real channel again
#if defined(CDN_DIAG) && CDN_DIAG == 8   // diagnostic build: every item reads the planes of item 0 (wrong results)
      glds16(x, (unsigned)(((cs << qsh) + (slot ^ (c & 15))) << 4), raw_a + piece * 1024);
#else
      glds16(x + ((long)n * Cx + (long)chunk * CCH) * HW, (unsigned)(((cs << qsh) + (slot ^ (c & 15))) << 4),
             raw_a + piece * 1024);
#endif
    }
    if (k == 0)
      for (int sp = wave; sp < npiece_w + npiece_s; sp += kWaves) {
        if (sp < npiece_w) {
          // (ragged last chunk: weight 9 c + k of a pad channel c comes from the last real channel)
          const int wi = sp * 64 + lane, wc = wi / 9, wlast = Cx - 1 - chunk * CCH;
          const int wsrc = wc <= wlast ? wi : wlast * 9 + (wi - wc * 9);
          glds4(wd + (long)chunk * CCH * 9, (unsigned)wsrc * 4, wl_a + (set * CCH * 9 + sp * 64) * 4);
        }
        else
          glds4(s_raw + (long)n * HW + (sp - npiece_w) * 64, lane * 4, sl_a + (set * HW + (sp - npiece_w) * 64) * 4);
      }
  };
  const int npieces = (Q + kWaves - 1) / kWaves;                   // raw pieces per wave
  auto issue = [&](int item, int set, int k0) {
    for (int k = k0; k < max(npieces, 1); ++k) issue_step(item, set, k);
  };
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  BadMask bad = 0;
  Code8 c8 = {1.f, 0.f};
  if (OUT8) c8 = make_code8(qu.state, bad);
  // (tried: s_setprio(1) for the second-dispatched half of the waves -- 0.2514 vs 0.2508 ms per step, nothing)
  int item = blockIdx.x, set = 0;
  if (item < nitems) issue(item, 0, 0);
  for (; item < nitems; item += gridDim.x, set ^= 1) {
    const int n = item / nchunk, chunk = item - n * nchunk;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the item's DMA have landed ...
    __syncthreads();       // ... everybody's have, and everybody is done with img and the other (wl, sl) set
    CDN_STAMP(1);
    if (item == (int)blockIdx.x) CDN_STAMP(5);
    // ---- raw -> img: lane <-> channel, wave w takes quads w, w + 16, ... -----------------------------------
    float *imgf = reinterpret_cast<float *>(img);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    for (int qd = wave; qd < Q; qd += kWaves) {
      const float4 v = *reinterpret_cast<const float4 *>(raw + (ln * Q + (qd ^ (ln & 15))) * 4);
      const int p = qd * 4, row = p / W, col = p - row * W;         // W % 4 == 0: one row
      float *dst = imgf + (row * Wc + col) * CCH + ln;
      dst[0] = v.x; dst[CCH] = v.y; dst[2 * CCH] = v.z; dst[3 * CCH] = v.w;
      has_nan |= __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);   // (see track4: x is tested where it is staged)
    }
    float *sl = sl0 + set * HW;
    if (SQ) for (int q = tid; q < HW; q += 1024) sl[q] = fake_quant(sl[q], ss, sz);
    __syncthreads();
    CDN_STAMP(2);
    // the NEXT item's transfer: in flight during the gather below.  (Issuing one raw piece per gather step instead of
    // the 64-KB burst was built and measured: no difference, 0.2521 vs 0.2524 ms per step, and the hook cost the
    // byte-code instantiation 20 spilled VGPRs.)
    const int nxt = item + gridDim.x;
    if (nxt < nitems) issue(nxt, set ^ 1, 0);
    if (OUT8) dw2_gather<CCH, true>(img, wl0 + set * CCH * 9, sl, d, n, chunk * CCH, C, H, W, kWaves, mn, mx, has_nan, &c8, &bad);
    else dw2_gather<CCH, false>(img, wl0 + set * CCH * 9, sl, d, n, chunk * CCH, C, H, W, kWaves, mn, mx, has_nan);
    CDN_STAMP(3);
    if (item == (int)blockIdx.x) CDN_STAMP(6);
    if (item == (int)(blockIdx.x + gridDim.x)) CDN_STAMP(7);
  }
  if (OUT8) {
    if (bad) atomicOr(reinterpret_cast<unsigned *>(dmm), 1u);
    return;
  }
  if (dmm) cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), dmm, blockIdx.x, gridDim.x, qu, red);
  CDN_STAMP(4);
}

// ------------------------------------------------------------------------------------------
// dw2u: the gather + depthwise kernel for an UP-SAMPLED input (up = 1, channels-last, stages >= 1),
// processing one 2x2 block of output pixels -- one stored pixel (Y,X) -- per lane group.
//
// The four output pixels of a block share the scale s (it is predicted at the stored resolution),
// so their tap positions are translates of each other by one output pixel = half a stored cell.
// Along one axis the 2+2 bilinear corners of the two pixels fall into at most TWO consecutive
// stored cells {cb, cb+1}: a corner tap needs 2x2 = 4 cell reads for the WHOLE block instead of 16,
// an edge tap 2 instead of 8, the centre tap 1 instead of 4 -- 25 ds_read_b128 per 16 outputs/lane
// instead of 100, and one geometry record per block instead of four.  Each pixel keeps its own
// fp32 corner weights (positions bit-identical to the reference pipeline); they are folded onto the
// two cells per axis, W[slot] = sum of the corner weights that land in that cell.  (If fp32 rounding
// makes a pixel's far corner land one cell further -- its weight is then ~1 ulp -- it is folded
// into slot 1; the effect is below 1e-6 of the value.)
// ------------------------------------------------------------------------------------------
struct AxisRaw {
  int fl;      // floor(pos), NOT parked
  float w0, w1;  // corner weights, zero when the sample is out of range on this axis
};

__device__ __forceinline__ AxisRaw make_axis_raw(int base, float off, int size) {
  AxisRaw a;
  const float pos = (float)base + off;
  const bool ok = pos > -1.0f && pos < (float)size;
  const float fl = floorf(pos);
  // clamp the index so far-out-of-range positions (|s| is not bounded by this kernel) stay sane
  a.fl = (int)fminf(fmaxf(fl, -4.0f), (float)size + 4.0f);
  const float l = pos - fl;
  a.w1 = ok ? l : 0.0f;
  a.w0 = ok ? 1.0f - l : 0.0f;
  return a;
}

// Fold the two pixels (a: lower coordinate, b = a + 1) of one axis class onto cells {cb, cb+1}.
// out[0..1] = pixel a's slot weights, out[2..3] = pixel b's.  Returns cb (stored-cell index).
__device__ __forceinline__ int fold_axis(const AxisRaw &a, const AxisRaw &b, float *out) {
  const int cb = a.fl >> 1;
  const int sa1 = min(max(((a.fl + 1) >> 1) - cb, 0), 1);
  const int sb0 = min(max((b.fl >> 1) - cb, 0), 1);
  const int sb1 = min(max(((b.fl + 1) >> 1) - cb, 0), 1);
  out[0] = a.w0 + (sa1 == 0 ? a.w1 : 0.0f);
  out[1] = (sa1 == 1 ? a.w1 : 0.0f);
  out[2] = (sb0 == 0 ? b.w0 : 0.0f) + (sb1 == 0 ? b.w1 : 0.0f);
  out[3] = (sb0 == 1 ? b.w0 : 0.0f) + (sb1 == 1 ? b.w1 : 0.0f);
  return cb;
}

#ifndef CDN_UNEVEN_SPLIT
#define CDN_UNEVEN_SPLIT 1
#endif
constexpr bool kUnevenSplit = CDN_UNEVEN_SPLIT != 0;

template <int CCH, bool XQ, bool SQ, bool X8 = false, bool OUT8 = false>   // X8 / OUT8: see dw2_kernel
__global__ void __launch_bounds__(512)   // ~250 VGPRs: 2 waves/SIMD (168 spills and is 2.5x slower)
dw2u_kernel(const float *__restrict__ x, const unsigned *__restrict__ xq,
            const float *__restrict__ s_raw, const unsigned *__restrict__ sq,
            const float *__restrict__ wd, float *__restrict__ d, float2 *dmm, cdn::QUpdate qu,
            int C, int H, int W, cdn::ScaleFromSums si) {
  static_assert(!X8 || XQ, "codes come with their quantiser state");
  // LDS: one ZERO cell, then the image as in dw2_kernel.  Cell (r, -1) is the cell in front of row r: the zero
  // column of row r - 1, or the leading zero cell for r = 0 -- so column -1 needs no parking and the two cells
  // {cb, cb + 1} of a column class are ALWAYS adjacent: one offset per class, the second read is the first plus
  // an immediate.
  extern __shared__ float4 img_lds[];
  CDN_STAMP(0);
  constexpr int LPP = CCH / 4;     // lanes per block (one float4 of channels each)
  float4 *img = img_lds + LPP;
  constexpr int PPW = 64 / LPP;    // blocks per wave step
  const int nthreads = blockDim.x, nwaves = nthreads / 64;
  const int Hl = H >> 1, Wl = W >> 1;
  const int HWl = Hl * Wl, HW = H * W;
  int n = blockIdx.y, chunk = blockIdx.x;
  if (X8 || OUT8) xcd_remap(chunk, n);
  const int c0 = chunk * CCH;
  const int tid = threadIdx.x;
  const int Wc = Wl + 1;
  const int cells = (Hl + 1) * Wc;
  float *wl = reinterpret_cast<float *>(img + (size_t)cells * LPP);
  float *sl = wl + CCH * 9;
  float *red = sl + HWl;
  float xs = 1.f, xz = 0.f, ss = 1.f, sz = 0.f, xr_ = 1.f;
  if (XQ) {
    xs = reinterpret_cast<const float *>(xq)[2];
    xr_ = __fdiv_rn(1.0f, xs);   // Markstein division in fake_quant_r
    xz = reinterpret_cast<const float *>(xq)[3];
  }
  if (SQ) {
    ss = reinterpret_cast<const float *>(sq)[2];
    sz = reinterpret_cast<const float *>(sq)[3];
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = tid; q < (Wc + Hl) * LPP; q += nthreads) {   // zero row, then zero column
    const int i = q / LPP;
    const int cell = i < Wc ? Hl * Wc + i : (i - Wc) * Wc + Wl;
    img[cell * LPP + (q % LPP)] = z4;
  }
  if (tid < LPP) img_lds[tid] = z4;                          // the leading zero cell
  // weights and scale plane requested before the image (see dw2_kernel)
  const float w_pre = (tid < CCH * 9 && c0 + tid / 9 < C) ? wd[(long)c0 * 9 + tid] : 0.0f;
  // scale plane: from the scale kernel, or (frozen chained stages) from the producer's exact integer sums --
  // s_raw = clamp(bias + sums / (sw * sc_x)): ONE rounding of the exact sum instead of the fp32 sum of C products
  float s_inv = 0.f, s_b = 0.f;
  if (XQ && si.sums) {
    s_inv = __fdiv_rn(1.0f, __fmul_rn(si.sw[0], xs));
    s_b = si.bias ? si.bias[0] : 0.0f;
  }
  const float f_b = (si.fparts && si.bias) ? si.bias[0] : 0.0f;
  const long f_stride = (long)gridDim.y * HWl;               // one plane of partial sums: every stored pixel of the batch
  auto s_at = [&](int q) -> float {
    if (XQ && si.sums) return fminf(fmaxf(fmaf((float)si.sums[(long)n * HWl + q], s_inv, s_b), si.lo), si.hi);
    if (!XQ && !SQ && si.fparts) {      // fp32 chained stages: the producer's per-column-tile partial sums, in plane order
      const float *fp = si.fparts + (long)n * HWl + q;
      float a = fp[0];
      for (int j = 1; j < si.nparts; ++j) a += fp[j * f_stride];
      return cdn::clamp_keep_nan(a + f_b, si.lo, si.hi);
    }
    return s_raw[(long)n * HWl + q];
  };
  const float s_pre = tid < HWl ? s_at(tid) : 0.0f;
  if (X8) {
    constexpr int kStageU = 8;
    const signed char *xg = reinterpret_cast<const signed char *>(x) + (long)n * HWl * C + c0;
    const int total = HWl * LPP;
    for (int base = 0; base < total; base += nthreads * kStageU) {
      unsigned v[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * nthreads + tid;
        const int pix = q / LPP, cq4 = q % LPP;
        v[u] = 0u;
        if (q < total && c0 + cq4 * 4 + 3 < C) v[u] = *reinterpret_cast<const unsigned *>(xg + (long)pix * C + cq4 * 4);
      }
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * nthreads + tid;
        const int pix = q / LPP, cq4 = q % LPP;
        if (q < total)
          img[((pix / Wl) * Wc + (pix % Wl)) * LPP + cq4] =
              (c0 + cq4 * 4 + 3 < C) ? unpack_code8(v[u], xs, xz, xr_) : z4;
      }
    }
  } else {
    // batched: kStageU loads per thread in flight (see dw2_kernel)
    constexpr int kStageU = 8;
    const float *xg = x + (long)n * HWl * C + c0;
    const int total = HWl * LPP;
    for (int base = 0; base < total; base += nthreads * kStageU) {
      float4 v[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * nthreads + tid;
        const int pix = q / LPP, cq4 = q % LPP;
        v[u] = z4;
        if (q < total && c0 + cq4 * 4 + 3 < C)
          v[u] = *reinterpret_cast<const float4 *>(xg + (long)pix * C + cq4 * 4);
      }
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const int q = base + u * nthreads + tid;
        const int pix = q / LPP, cq4 = q % LPP;
        if (q < total) {
          float4 t = v[u];
          if (XQ) {
            t.x = cdn::fake_quant_r(t.x, xs, xz, xr_);
            t.y = cdn::fake_quant_r(t.y, xs, xz, xr_);
            t.z = cdn::fake_quant_r(t.z, xs, xz, xr_);
            t.w = cdn::fake_quant_r(t.w, xs, xz, xr_);
          }
          img[((pix / Wl) * Wc + (pix % Wl)) * LPP + cq4] = t;
        }
      }
    }
  }
  if (tid < CCH * 9) wl[tid] = w_pre;
  for (int q = tid + nthreads; q < CCH * 9; q += nthreads)
    wl[q] = (c0 + q / 9 < C) ? wd[(long)c0 * 9 + q] : 0.0f;
  if (tid < HWl) sl[tid] = SQ ? fake_quant(s_pre, ss, sz) : s_pre;
  for (int q = tid + nthreads; q < HWl; q += nthreads) {
    float sv = s_at(q);
    if (SQ) sv = fake_quant(sv, ss, sz);
    sl[q] = sv;
  }
  CDN_STAMP(1);
  __syncthreads();
  CDN_STAMP(2);
  const int lane = tid & 63, wave = tid >> 6;
  const int cq = lane % LPP, sub = lane / LPP;
  float wk[9][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k][e] = wl[(cq * 4 + e) * 9 + k];

  const int rstride = Wc * LPP * 16;
  // stored-cell index -> byte offset (out-of-image cells select the zero row / column)
  auto row_off = [&](int cy) { return (((unsigned)cy < (unsigned)Hl) ? cy : Hl) * rstride; };
  // first cell of a column class: cb in [-1, Wl - 1] as it is (cells cb, cb + 1 lie in [-1, Wl]); outside that
  // range every weight of the class is zero (fold_axis), any pair of cells will do
  constexpr int kCell = LPP * 16;
  auto col_off = [&](int cb) { return min(max(cb, -1), Wl - 1) * kCell; };
  const char *imgb = reinterpret_cast<const char *>(img) + cq * 16;
#if defined(CDN_DIAG) && CDN_DIAG == 2   // diagnostic build: no LDS cell reads (wrong results)
#define CDN_RD(O) make_float4(__int_as_float(O), 1.f, 2.f, 3.f)
#else
#define CDN_RD(O) (*reinterpret_cast<const float4 *>(imgb + (O)))
#endif

  float mn = INFINITY, mx = -INFINITY, mn2 = INFINITY, mx2 = -INFINITY;

  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  BadMask bad = 0;
  Code8 c8 = {1.f, 0.f};
  if (OUT8) c8 = make_code8(qu.state, bad);
  const bool vec_store = ((C & 3) == 0);
  // every wave owns a contiguous range of blocks (= stored pixels).  The second-dispatched half of the waves
  // (4..7) loses the oldest-first arbitration of VALU / LDS issue on every step and finishes ~30 % later than the
  // first half on equal ranges (in-kernel per-wave stamps: 20.2 vs 26.0 us at stage 2); with >= 12 steps per wave
  // the first half therefore takes 9/16 of the steps.  (Fixed assignment: results do not depend on it.)
  int b_begin, b_end;
  {
    const int steps = (HWl + PPW - 1) / PPW, half = nwaves / 2;
    if (kUnevenSplit && (nwaves & 1) == 0 && steps >= 12 * nwaves) {
      const int old_each = (steps * 9 / 16 + half - 1) / half;           // steps per wave, first half
      const int old_total = min(steps, old_each * half);
      const int young_each = (steps - old_total + half - 1) / half;
      const int s0 = wave < half ? wave * old_each : old_total + (wave - half) * young_each;
      const int s1 = wave < half ? min(old_total, s0 + old_each) : s0 + young_each;
      b_begin = min(HWl, s0 * PPW);
      b_end = min(HWl, s1 * PPW);
    } else {
      const int bpw = ((HWl + nwaves - 1) / nwaves + PPW - 1) / PPW * PPW;
      b_begin = min(HWl, wave * bpw);
      b_end = min(HWl, b_begin + bpw);
    }
  }
  for (int bb = b_begin; bb < b_end; bb += 64) {
    // ---- geometry phase: lane i owns block bb + i -----------------------------------------------
    int g_o[9];       // byte offsets: rows {ya: r0,r1; yb: r0,r1; mid}, cols {xa, xb: first cell; mid}; [8]: the
                      // block's first output pixel 2Y * W + 2X
    float g_w[16];    // slot weights [class ya,yb,xa,xb][pixel a/b][slot 0/1]
    {
      const int blk = min(bb + (use_dpp<LPP>() ? owner_item<LPP>(lane) : lane), HWl - 1);
      const int Y = blk / Wl, X = blk - Y * Wl;
      const int h0 = 2 * Y, w0 = 2 * X;
      const float t = sl[blk] - 1.0f;
      // tap row i = 0: pos = (h - 1) - t ; i = 2: pos = (h + 1) + t ; pixel rows h0 and h0 + 1
      const AxisRaw ya_a = make_axis_raw(h0 - 1, -t, H), ya_b = make_axis_raw(h0, -t, H);
      const AxisRaw yb_a = make_axis_raw(h0 + 1, t, H), yb_b = make_axis_raw(h0 + 2, t, H);
      const AxisRaw xa_a = make_axis_raw(w0 - 1, -t, W), xa_b = make_axis_raw(w0, -t, W);
      const AxisRaw xb_a = make_axis_raw(w0 + 1, t, W), xb_b = make_axis_raw(w0 + 2, t, W);
      const int rya = fold_axis(ya_a, ya_b, &g_w[0]);
      const int ryb = fold_axis(yb_a, yb_b, &g_w[4]);
      const int cxa = fold_axis(xa_a, xa_b, &g_w[8]);
      const int cxb = fold_axis(xb_a, xb_b, &g_w[12]);
      g_o[0] = row_off(rya); g_o[1] = row_off(rya + 1);
      g_o[2] = row_off(ryb); g_o[3] = row_off(ryb + 1);
      g_o[4] = row_off(Y);
      g_o[5] = col_off(cxa);
      g_o[6] = col_off(cxb);
      g_o[7] = X * kCell;
      g_o[8] = h0 * W + w0;
    }
    // ---- gather phase: PPW blocks per step -----------------------------------------------------
#pragma unroll 1
    for (int j = 0; j < 64 / PPW; ++j) {
      const int src = j * PPW + sub;
      const int blk = bb + src;
      int o[9];
      float w[16];
      if (use_dpp<LPP>()) {
        fetch_record<LPP == 8>(j, g_o, g_w, o, w);
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q] = __shfl(g_o[q], src, 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = __shfl(g_w[q], src, 64);
      }
      if (bb + j * PPW >= b_end) break;     // wave-uniform
      // The nine taps factor over the block: the three taps of a row class (ya: taps 0-2, exact row: 3-5, yb: 6-8)
      // read the same cell rows, a pixel's corner weights are (row weight) x (column weight), and the column
      // weights do not depend on py nor the row weights on px.  So per cell row: column-mix the five cells for
      // px = 0, 1 and fold the three depthwise weights (7 fma per component and px); then row-mix the two cell
      // rows into the four accumulators (2 fma each).  86 fma per component and block instead of 148
      // (+ 64 scalar weight products) when every tap of every pixel is mixed on its own.
      float4 acc[2][2];     // [py][px]
      auto row_taps = [&](int ro, int K0, float4 (&U)[2]) {
        const int oa = ro + o[5], ob = ro + o[6];
        const float4 va0 = CDN_RD(oa), va1 = CDN_RD(oa + kCell);
        const float4 vc = CDN_RD(ro + o[7]);
        const float4 vb0 = CDN_RD(ob), vb1 = CDN_RD(ob + kCell);
        const float4 t = make_float4(wk[K0 + 1][0] * vc.x, wk[K0 + 1][1] * vc.y, wk[K0 + 1][2] * vc.z,
                                     wk[K0 + 1][3] * vc.w);
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const float a0 = w[8 + 2 * px], a1 = w[8 + 2 * px + 1], b0 = w[12 + 2 * px], b1 = w[12 + 2 * px + 1];
          float4 ua, ub;
          ua.x = fmaf(a1, va1.x, a0 * va0.x); ua.y = fmaf(a1, va1.y, a0 * va0.y);
          ua.z = fmaf(a1, va1.z, a0 * va0.z); ua.w = fmaf(a1, va1.w, a0 * va0.w);
          ub.x = fmaf(b1, vb1.x, b0 * vb0.x); ub.y = fmaf(b1, vb1.y, b0 * vb0.y);
          ub.z = fmaf(b1, vb1.z, b0 * vb0.z); ub.w = fmaf(b1, vb1.w, b0 * vb0.w);
          U[px].x = fmaf(wk[K0][0], ua.x, fmaf(wk[K0 + 2][0], ub.x, t.x));
          U[px].y = fmaf(wk[K0][1], ua.y, fmaf(wk[K0 + 2][1], ub.y, t.y));
          U[px].z = fmaf(wk[K0][2], ua.z, fmaf(wk[K0 + 2][2], ub.z, t.z));
          U[px].w = fmaf(wk[K0][3], ua.w, fmaf(wk[K0 + 2][3], ub.w, t.w));
        }
      };
      {
        float4 M[2];
        row_taps(o[4], 3, M);                      // exact row Y: the same sum for py = 0, 1
        acc[0][0] = acc[1][0] = M[0];
        acc[0][1] = acc[1][1] = M[1];
      }
#pragma unroll
      for (int cls = 0; cls < 2; ++cls) {          // ya (offsets o[0..1], weights w[0..3]), yb (o[2..3], w[4..7])
        float4 U0[2], U1[2];
        row_taps(o[2 * cls], 6 * cls, U0);
        row_taps(o[2 * cls + 1], 6 * cls, U1);
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const float r0 = w[4 * cls + 2 * py], r1 = w[4 * cls + 2 * py + 1];
            float4 &A = acc[py][px];
            A.x = fmaf(r0, U0[px].x, fmaf(r1, U1[px].x, A.x));
            A.y = fmaf(r0, U0[px].y, fmaf(r1, U1[px].y, A.y));
            A.z = fmaf(r0, U0[px].z, fmaf(r1, U1[px].z, A.z));
            A.w = fmaf(r0, U0[px].w, fmaf(r1, U1[px].w, A.w));
          }
      }
      if (blk < b_end) {
        const int cbase = c0 + cq * 4;
        const long e0 = ((long)n * HW + o[8]) * C + cbase;    // element index of the block's first output pixel
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const float4 a = acc[py][px];
            const long e = e0 + (long)(py * W + px) * C;
            float *dp = d + e;
            if (OUT8) {
              if (cbase + 3 < C)
                *reinterpret_cast<unsigned *>(reinterpret_cast<signed char *>(d) + e) = pack_code8(a, c8, bad);
            } else if (vec_store && cbase + 3 < C) {
              *reinterpret_cast<float4 *>(dp) = a;
              track4(a, mn, mn2, mx, mx2);
            } else {
              const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (cbase + e < C) {
                  dp[e] = av[e];
                  mn = fminf(mn, av[e]);
                  mx = fmaxf(mx, av[e]);
                  has_nan |= (av[e] != av[e]);
                }
            }
          }
      }
    }
  }
#undef CDN_RD
  if (OUT8) {
    if (bad) atomicOr(reinterpret_cast<unsigned *>(dmm), 1u);
    return;
  }
  mn = fminf(mn, mn2);
  mx = fmaxf(mx, mx2);
  CDN_STAMP_WAVE();
  CDN_STAMP(3);
  if (dmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), dmm, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, red);
  CDN_STAMP(4);
}

// ------------------------------------------------------------------------------------------
// pw2: R[m][co] = act( sum_c Aq[m][c] * Wp[co][c] + ... ),  m = n*HW + p  (channels-last both
// sides, the batch folds into M).  f32 MFMA 32x32x2 (exact f32).  A is optionally
// fake-quantised while it is staged.  Tile 128 (m) x BN (co) x 16 (k); 4 waves stacked in m,
// each holding BN/32 accumulators.  LDS rows have an odd stride (17) so the operand read
// `[row = lane&31][k = lane>>5]` is conflict-free with K-contiguous global loads on both sides.
// MFMA maps: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
// Here i <-> pixel, j <-> co, so every accumulator register stores 128 contiguous bytes.
// ------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kPwBK = 32, kPwLD = 33;

// Software-pipelined: while the MFMAs of k-tile t run out of LDS buffer t&1, the global loads of
// k-tile t+1 are in flight into registers; they are fake-quantised (A) and written to buffer
// (t+1)&1 after the MFMAs, so there is ONE barrier per 32-deep k-tile and the load latency, the
// quantisation VALU work and the LDS writes all sit behind matrix work.  All operand fragments
// of a k-tile are read from LDS up front, then the MFMAs issue back to back.
// Workgroup = 4 waves arranged WGM (m) x 4/WGM (n); every wave owns TM x TN accumulators of 32x32.
// FAST: C % 32 == 0 (no k guards, 16-byte loads); otherwise guarded scalar loads.
template <int BM, int BN, int WGM, bool AQ, bool FAST>
__global__ void __launch_bounds__(256)
pw3_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq,
           const float *__restrict__ Wp, const float *__restrict__ bias,
           const float *__restrict__ ep_scale, const float *__restrict__ ep_shift,
           float *__restrict__ R, float2 *rmm, cdn::QUpdate qu, long M, int C, int Co, int relu,
           int only_if_wide, int lda, int ldo) {
  if (only_if_wide && !aq[6]) return;   // fallback launch behind pwi8_kernel: nothing to do
  constexpr int WGN = 4 / WGM;
  constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
  constexpr int AI = BM * 8 / 256, BI = BN * 8 / 256;   // float4 loads per thread per k-tile
  static_assert(TM >= 1 && TN >= 1 && AI >= 1 && BI >= 1, "tile too small for 256 threads");
  __shared__ float As[2][BM * kPwLD];
  __shared__ float Bs[2][BN * kPwLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * TM * 32, wn = (wave % WGN) * TN * 32;
  float qs = 1.f, qz = 0.f;
  if (AQ) {
    qs = reinterpret_cast<const float *>(aq)[2];
    qz = reinterpret_cast<const float *>(aq)[3];
  }
  // workgroups walk the (m, n) tiles with a grid stride: the full-size launch visits one tile each,
  // the fallback launch behind pwi8_kernel is a small grid (an EMPTY full-size launch still costs
  // 4.3 us of workgroup dispatch inside the graph)
  const long ntm = (M + BM - 1) / BM, ntiles = ntm * ((Co + BN - 1) / BN);
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const long m0 = (tile % ntm) * BM;
  const int n0 = (int)(tile / ntm) * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){0};
  const int lr = tid >> 3, lk = (tid & 7) * 4;   // staging: row lr + 32*i, k quad lk
  const bool vec4 = !FAST && (C & 3) == 0 && (lda & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Wp)) & 15) == 0;
  const bool vec2 = !FAST && !vec4 && (C & 1) == 0 && (lda & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(A) & 7) == 0;
  float4 a[AI], b[BI];
  // FAST path: rows beyond M / Co are clamped to a valid row (their results are never stored)
  const float *arow[AI];
  const float *brow[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    long m = m0 + lr + 32 * i;
    if (m > M - 1) m = M - 1;
    arow[i] = A + m * lda + lk;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int co = n0 + lr + 32 * i;
    if (co > Co - 1) co = Co - 1;
    brow[i] = Wp + (long)co * C + lk;
  }

  auto load_tile = [&](int k0) {
    if (FAST) {
#pragma unroll
      for (int i = 0; i < AI; ++i) a[i] = *reinterpret_cast<const float4 *>(arow[i] + k0);
#pragma unroll
      for (int i = 0; i < BI; ++i) b[i] = *reinterpret_cast<const float4 *>(brow[i] + k0);
    } else if (vec4) {      // C % 4 == 0, 16-byte aligned rows: whole quads are in or out of the k range
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool in = k0 + lk < C;
#pragma unroll
      for (int i = 0; i < AI; ++i) a[i] = in ? *reinterpret_cast<const float4 *>(arow[i] + k0) : z;
#pragma unroll
      for (int i = 0; i < BI; ++i) b[i] = in ? *reinterpret_cast<const float4 *>(brow[i] + k0) : z;
    } else {
      if (vec2) {           // C % 2 == 0, 8-byte aligned rows (a unit's second half at channel 58)
        const float2 z = make_float2(0.f, 0.f);
        const int k = k0 + lk;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const float2 lo = (k < C) ? *reinterpret_cast<const float2 *>(arow[i] + k0) : z;
          const float2 hi = (k + 2 < C) ? *reinterpret_cast<const float2 *>(arow[i] + k0 + 2) : z;
          a[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      } else {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const int k = k0 + lk;
          a[i].x = (k + 0 < C) ? arow[i][k0 + 0] : 0.0f;
          a[i].y = (k + 1 < C) ? arow[i][k0 + 1] : 0.0f;
          a[i].z = (k + 2 < C) ? arow[i][k0 + 2] : 0.0f;
          a[i].w = (k + 3 < C) ? arow[i][k0 + 3] : 0.0f;
        }
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int k = k0 + lk;
        b[i].x = (k + 0 < C) ? brow[i][k0 + 0] : 0.0f;
        b[i].y = (k + 1 < C) ? brow[i][k0 + 1] : 0.0f;
        b[i].z = (k + 2 < C) ? brow[i][k0 + 2] : 0.0f;
        b[i].w = (k + 3 < C) ? brow[i][k0 + 3] : 0.0f;
      }
    }
  };
  auto store_tile = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      float4 v = a[i];
      if (AQ) {
        if (FAST) {
          v.x = fake_quant(v.x, qs, qz);
          v.y = fake_quant(v.y, qs, qz);
          v.z = fake_quant(v.z, qs, qz);
          v.w = fake_quant(v.w, qs, qz);
        } else {   // keep the zero padding of the k tail exact
          const int k = k0 + lk;
          v.x = (k + 0 < C) ? fake_quant(v.x, qs, qz) : 0.0f;
          v.y = (k + 1 < C) ? fake_quant(v.y, qs, qz) : 0.0f;
          v.z = (k + 2 < C) ? fake_quant(v.z, qs, qz) : 0.0f;
          v.w = (k + 3 < C) ? fake_quant(v.w, qs, qz) : 0.0f;
        }
      }
      float *p = &As[buf][(lr + 32 * i) * kPwLD + lk];
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      float *p = &Bs[buf][(lr + 32 * i) * kPwLD + lk];
      p[0] = b[i].x; p[1] = b[i].y; p[2] = b[i].z; p[3] = b[i].w;
    }
  };

  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  const int nk = (C + kPwBK - 1) / kPwBK;
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) load_tile((t + 1) * kPwBK);
    // all fragments of this k-tile, then the MFMAs back to back
    float av[kPwBK / 2][TM], bv[kPwBK / 2][TN];
#pragma unroll
    for (int kk = 0; kk < kPwBK / 2; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[kk][i] = As[buf][(wm + i * 32 + (lane & 31)) * kPwLD + 2 * kk + (lane >> 5)];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[kk][j] = Bs[buf][(wn + j * 32 + (lane & 31)) * kPwLD + 2 * kk + (lane >> 5)];
    }
#pragma unroll
    for (int kk = 0; kk < kPwBK / 2; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][i], bv[kk][j], acc[i][j], 0, 0, 0);
    if (t + 1 < nk) store_tile(buf ^ 1, (t + 1) * kPwBK);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn + j * 32 + (lane & 31);
    float bsv = 0.f, es = 1.f, eh = 0.f;
    if (co < Co) {
      if (bias) bsv = bias[co];
      if (ep_scale) {
        es = ep_scale[co];
        eh = ep_shift[co];
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && co < Co) {
          float v = acc[i][j][r] + bsv;
          if (ep_scale) v = fmaf(v, es, eh);
          if (relu) v = cdn::relu_keep_nan(v);
          R[m * ldo + co] = v;
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
          has_nan |= (v != v);
        }
      }
  }
  }   // tile loop
  if (rmm)   // (block_minmax_finish syncs before reusing As as scratch)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, blockIdx.x, gridDim.x, qu, &As[0][0]);
}

// ------------------------------------------------------------------------------------------
// pws: the f32 pointwise conv for launches with FEW output tiles (round 5; cfg2's stage 0: M = 2048 rows, K = 1024,
// Co = 256 -- pw3_kernel's 64 x 128 tiles are 64 workgroups on 256 CUs, each a chain of 32 dependent k tiles with 32
// MFMAs of 64 cycles per wave and tile: 27 us of matrix time on a quarter of the SIMDs, 47 us measured).
// Every WAVE owns one 32 (m) x 32*TN (co) output tile over a k range and runs its own pipeline -- no workgroup barrier
// in the k loop: 32-k windows of both operands (32 + 32 TN rows x 128 B) are fetched by LDS-DMA
// (global_load_lds_dwordx4: 8 lanes per row = whole 128-byte lines, no VGPRs) into a wave-private LDS ring R - 1
// windows ahead, read back in MFMA layout (lane (i = l & 31, h = l >> 5) takes the 16 consecutive k values
// [16 h, 16 h + 16) of row i as four ds_read_b128; a sum over k does not care which of the two k slots of
// v_mfma_f32_32x32x2_f32 a k lands in, only that A and B agree) and multiplied.  The DMA's LDS side is lane-linear
// (rows of exactly 128 B), so the 16-byte chunks of a row are stored XOR-swizzled -- chunk c of row r at position
// c ^ ((r >> 1) & 7), applied to the per-lane SOURCE address -- which makes the fragment reads conflict-free.
// (First form of this kernel, measured: lanes loading their 64-byte operand pieces straight from global memory -- every
// load instruction touched 32 lines for 16 bytes each; 15.3 us at cfg2's stage 0 and SLOWER, 18.3 us, with three windows
// in flight instead of one: bound by the texture path's line touches, not by latency or matrix time.)
// The four waves of a workgroup are
//   KS = 4: the four quarters of K of ONE tile (wave w: [w K/4, (w+1) K/4)); the partial tiles are added through LDS in
//           a FIXED order ((w0 + w1) + (w2 + w3)) -- for launches whose tiles alone would not give every SIMD a wave;
//   KS = 1: four consecutive m tiles, the whole K each; accumulators go straight to the epilogue.
// The launcher picks the widest tile and the least splitting that still gives >= 4 waves per CU.  Reproducible bit for
// bit from call to call; against pw3_kernel the k order differs (fp32 re-association; the fp32 path's parity bound is
// 1e-3 against the oracle).  Workgroups that share A rows (the co tiles of one m block) get ids 8 apart = the same XCD
// under round-robin dispatch, so an A block is fetched into one L2.
// Needs K % (32 KS) == 0 and 16-byte aligned rows; plain f32 operands (no quantise-on-load).
// ------------------------------------------------------------------------------------------
// Sum over the 32 lanes of each wave half (pws_sum32: the total arrives in lanes 16-31 / 48-63) or over the wave
// (pws_sum64: lanes 48-63) on the VALU's DPP path -- quad_perm xor 1, xor 2, row_ror 4, row_ror 8, row_bcast15
// [, row_bcast31]: five / six moves + adds per value in ONE fixed order, where a __shfl_xor tree is as many ds_bpermute
// round trips through the LDS crossbar (measured: +2.8 us on a 13-us launch for 16 rows x 5 steps per wave).
__device__ __forceinline__ float pws_dpp(float v, const int ctrl) {
  switch (ctrl) {      // (the DPP control is an immediate)
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    case 3: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    case 4: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  }
}
__device__ __forceinline__ float pws_sum32(float v) {
  v += pws_dpp(v, 0);
  v += pws_dpp(v, 1);
  v += pws_dpp(v, 2);
  v += pws_dpp(v, 3);       // every lane: the sum of its 16-lane row
  v += pws_dpp(v, 4);       // rows 1 and 3: + the row in front of them
  return v;
}
__device__ __forceinline__ float pws_sum64(float v) {
  v = pws_sum32(v);
  v += pws_dpp(v, 5);       // row 3: + lane 31's total of the first half
  return v;
}

constexpr int kPwsLD = 72;      // floats per LDS row of a partial tile: the two lane halves (rows 4 apart) hit disjoint banks
template <int TN> struct PwsGeom {
  static constexpr int R = TN == 2 ? 3 : 4;                 // ring slots per wave
  static constexpr int kSlot = (32 + 32 * TN) * 128;        // bytes per window: A rows, then B rows, 128 B each
  static constexpr int kRing = R * kSlot;
  static constexpr int kLds = 4 * kRing;                    // 144 KB (TN = 2) / 128 KB (TN = 1)
};

template <int TN, int KS>
__global__ void __launch_bounds__(256)
pws_kernel(const float *__restrict__ A, const float *__restrict__ Wp, const float *__restrict__ bias,
           const float *__restrict__ ep_scale, const float *__restrict__ ep_shift, float *__restrict__ R, float2 *rmm,
           cdn::QUpdate qu, long M, int K, int Co, int relu, int lda, int ldo, const float *__restrict__ nws,
           float *__restrict__ sparts) {
  // nws / sparts (round 6, chained fp32 stages): nws [Co] = the NEXT stage's conv_scale weights; this workgroup's column
  // tile leaves sparts[nt][m] = sum over its columns of out[m][co] * nws[co] (lane tree in a fixed order), which the
  // next stage's gather sums over the tiles in plane order (cdn::ScaleFromSums::fparts) -- the separate scale launch of
  // a stage whose input has no QuantAct in front of it (the fp32 model) is gone.
  using G = PwsGeom<TN>;
  constexpr int BN = 32 * TN;
  constexpr int BM = KS == 4 ? 32 : 128;          // rows of A per workgroup
  constexpr int NL = 4 + 4 * TN;                  // DMA instructions per window and wave
  static_assert(BN <= 64 && (KS == 1 || KS == 4) && 32 * kPwsLD * 4 <= G::kRing, "partial tile aliases the ring");
  extern __shared__ float4 pws_lds[];
  char *lds = reinterpret_cast<char *>(pws_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int ntm = (int)((M + BM - 1) / BM), ntn = (Co + BN - 1) / BN;
  // (m block, n tile) of this workgroup: within the part of the m range that is a multiple of 8 blocks, the n tiles of
  // one m block are 8 ids apart (same XCD); the remainder is mapped plainly.  A bijection on the grid.
  int mt, nt;
  {
    const int b = blockIdx.x, full_m = ntm & ~7, full = full_m * ntn;
    if (b < full) {
      mt = (b & 7) + 8 * (b / (8 * ntn));
      nt = (b >> 3) % ntn;
    } else {
      const int rem = ntm - full_m, q = b - full;
      mt = full_m + q % rem;
      nt = q / rem;
    }
  }
  const long m0 = (long)mt * BM + (KS == 4 ? 0 : 32 * w);      // first row of this WAVE's tile (wave-uniform)
  const int n0 = nt * BN;
  const int Kw = KS == 4 ? K >> 2 : K, kbase = KS == 4 ? w * Kw : 0;
  // ---- DMA side: lane <-> (row 8 u + (l >> 3), 16-byte chunk l & 7) of instruction u; rows beyond M / Co are clamped
  const float *abase = A + (m0 < M ? m0 : M - 1) * lda + kbase;          // wave-uniform bases, 32-bit lane offsets
  const float *bbase = Wp + (long)n0 * K + kbase;
  const int dr = lane >> 3, dc = lane & 7;
  unsigned aoff[4], boff[TN][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = 8 * u + dr;
    const long rmax = M - 1 - m0;                                         // last valid row of this tile (may be < 0)
    const int rl = (int)(rmax < 0 ? 0 : (row < rmax ? row : rmax));
    aoff[u] = (unsigned)rl * (unsigned)lda * 4u + (unsigned)((dc ^ ((row >> 1) & 7)) * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cmax = Co - 1 - n0 - 32 * j;
      const int cl = cmax < 0 ? -32 * j : (row < cmax ? row : cmax);      // (cmax < 0: the whole co tile is padding)
      boff[j][u] = (unsigned)(32 * j + cl) * (unsigned)K * 4u + (unsigned)((dc ^ ((row >> 1) & 7)) * 16);
    }
  }
  const unsigned ring = lds_addr_uniform(lds) + (unsigned)w * G::kRing;
  auto issue = [&](int t) {
    const unsigned dst = ring + (unsigned)(t % G::R) * G::kSlot, koff = (unsigned)t * 128u;
#pragma unroll
    for (int u = 0; u < 4; ++u) glds16(abase, aoff[u] + koff, dst + u * 1024);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) glds16(bbase, boff[j][u] + koff, dst + 4096 + j * 4096 + u * 1024);
  };
  // ---- MFMA side
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = (f32x16){0};
  const char *wring = lds + w * G::kRing;
  const int frow = i * 128, fsw = (i >> 1) & 7;
  const int nit = Kw >> 5;
  int issued = 0;
  for (; issued < G::R - 1 && issued < nit; ++issued) issue(issued);
  for (int t = 0; t < nit; ++t) {
    if (issued < nit) issue(issued++);
    // windows t .. issued - 1 are in flight (NL DMAs each, completing in order): wait for window t
    switch (issued - 1 - t) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: if (NL == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 2: if (NL == 12) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;      // (TN = 1, three windows behind this one)
    }
    const char *slot = wring + (t % G::R) * G::kSlot;
    float4 fa[4], fb[TN][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pos = ((4 * h + q) ^ fsw) * 16;
      fa[q] = *reinterpret_cast<const float4 *>(slot + frow + pos);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j][q] = *reinterpret_cast<const float4 *>(slot + 4096 + j * 4096 + frow + pos);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ae[4] = {fa[q].x, fa[q].y, fa[q].z, fa[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float be = e == 0 ? fb[j][q].x : e == 1 ? fb[j][q].y : e == 2 ? fb[j][q].z : fb[j][q].w;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], be, acc[j], 0, 0, 0);
        }
    }
  }
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  if (KS == 4) {
    // ---- the four k slices of the tile through LDS (each wave's partial tile over its own, drained ring), added in a
    //      fixed order
    float *Pw = reinterpret_cast<float *>(lds + w * G::kRing);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Pw[((r & 3) + 8 * (r >> 2) + 4 * h) * kPwsLD + j * 32 + i] = acc[j][r];
    __syncthreads();
    constexpr int RPT = 32 * BN / 256;            // rows per thread: thread <-> one column, RPT consecutive rows
    const int col = tid % BN, r0 = (tid / BN) * RPT;
    const int co = n0 + col;
    float bsv = 0.f, es = 1.f, eh = 0.f;
    if (co < Co) {
      if (bias) bsv = bias[co];
      if (ep_scale) {
        es = ep_scale[co];
        eh = ep_shift[co];
      }
    }
    const float *P0 = reinterpret_cast<const float *>(lds), *P1 = reinterpret_cast<const float *>(lds + G::kRing),
                *P2 = reinterpret_cast<const float *>(lds + 2 * G::kRing),
                *P3 = reinterpret_cast<const float *>(lds + 3 * G::kRing);
    float v[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int o = (r0 + rr) * kPwsLD + col;
      v[rr] = ((P0[o] + P1[o]) + (P2[o] + P3[o])) + bsv;
      if (ep_scale) v[rr] = fmaf(v[rr], es, eh);
      if (relu) v[rr] = cdn::relu_keep_nan(v[rr]);
    }
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr)
      if (m0 + r0 + rr < M && co < Co) {
        R[(m0 + r0 + rr) * ldo + co] = v[rr];
        mn = fminf(mn, v[rr]);
        mx = fmaxf(mx, v[rr]);
        has_nan |= (v[rr] != v[rr]);
      }
    if (sparts) {           // (wave-uniform) thread <-> column: the BN lanes of a row group hold one row each step
      const float wn = co < Co ? nws[co] : 0.0f;
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) {
        float t = co < Co ? v[rr] * wn : 0.0f;
        t = BN == 64 ? pws_sum64(t) : pws_sum32(t);
        if (col == BN - 1 && m0 + r0 + rr < M) sparts[(long)nt * M + m0 + r0 + rr] = t;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + j * 32 + i;
      float bsv = 0.f, es = 1.f, eh = 0.f;
      if (co < Co) {
        if (bias) bsv = bias[co];
        if (ep_scale) {
          es = ep_scale[co];
          eh = ep_shift[co];
        }
      }
      const float wn = (sparts && co < Co) ? nws[co] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[j][r] + bsv;
        if (ep_scale) v = fmaf(v, es, eh);
        if (relu) v = cdn::relu_keep_nan(v);
        if (m < M && co < Co) {
          R[m * ldo + co] = v;
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
          has_nan |= (v != v);
        }
        // (the accumulator is free now: it carries this lane's term of the row's partial sum, the 32-column sub-tiles in
        // j order)
        if (sparts) {
          const float term = co < Co ? v * wn : 0.0f;
          acc[0][r] = j == 0 ? term : acc[0][r] + term;
        }
      }
    }
    if (sparts) {           // lane <-> column i of the 32-column sub-tiles; the two lane halves hold different rows
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = pws_sum32(acc[0][r]);
        const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (i == 31 && m < M) sparts[(long)nt * M + m] = t;
      }
    }
  }
  if (rmm) {
    __syncthreads();      // (the scratch of block_minmax_finish aliases wave 0's ring / partial tile)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, blockIdx.x, gridDim.x, qu, reinterpret_cast<float *>(lds));
  }
}

// ------------------------------------------------------------------------------------------
// pwi8: the W4A8 pointwise conv on INTEGER CODES with v_mfma_i32_32x32x32_i8.
//   levels   L = q + zp,  q = round(sc*d - zp)  (QuantAct codes, NOT clamped by the reference)
//   weights  qw in [-8, 7], W' = qw / sw[co]     (per-channel symmetric 4 bit)
//   y[m][co] = sum_c (L/sc)*(qw/sw) + b = (sum_c L*qw) / (sc*sw[co]) + b      -- exact integer sum
// a = L - 128 is in [-128,127] for in-range data but the tracked range lags the batch, so codes a
// few LSB outside int8 are routine.  Instead of a second accumulator the K dimension is doubled:
//   a = 16*a1 + a0,  a0 in [0,15], a1 in [-128,127]   (|a| <= 2039: x up to 8x outside the range;
//                                                     beyond that the code saturates)
//   sum a*qw = sum a0*qw + sum a1*(16*qw),   16*qw in [-128,112] is still int8.
// sum L*qw = sum a*qw + 128*colsum(qw).  All integer arithmetic is exact; the only roundings are
// the final fp32 scale and bias add.  Operand lane map: lane (r = l&31, h = l>>5) supplies 16
// consecutive k bytes [16h, 16h+16) of its row's 32-byte k-step for BOTH operands (any k order works as long as A
// and B agree; checked with exact integer data, tools/probes/probe_i8.hip); C/D map as f32.
// Tried in round 3 and removed (source kept in tools/probes/archive/pwi8r_kernel.inc): the same kernel with the raw fp32 A / int8 B tiles fetched by
// global_load_lds into a 3-slot LDS ring two tiles ahead (counted vmcnt, no compiler-visible load in the k loop,
// bit-identical results) -- stage 0 (K = 1024) 39.7 vs 38.3 us, stage 1 (K = 256, only two 72-KB workgroups per CU
// instead of four) 31.6 vs 23.6 us: the k loop is not waiting for its global loads.
// ------------------------------------------------------------------------------------------
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
constexpr int kI8LD = 48;   // bytes per LDS row: 32 k + 16 pad -> conflict-free ds_read_b128
constexpr int kMixedMaxC = 512;   // channels of a mixed-generation input (per-channel state table in LDS)

// The rare branch of the int8 pointwise kernels: this batch's codes are too wide for the nibble split (state[6]), so
// it runs on f32 MFMA with the fake-quantised weights Wp.  As / Bs: BM x 17 / BN x 17 floats of LDS, red: scratch of
// block_minmax_finish.
template <int BM, int BN, int WGM>
__device__ __forceinline__ void pwi8_wide_path(const float *__restrict__ A, const float *__restrict__ Wp,
                                               const float *__restrict__ bias, float *__restrict__ R, float2 *rmm,
                                               const cdn::QUpdate &qu, long M, int C, int Co, int relu, int lda,
                                               int ldo, const int *__restrict__ omap, float qs, float qz,
                                               float *As, float *Bs, float *red, long m0, int n_first, int nrep, int pidx,
                                               int npart) {
  // the workgroup's tile: rows [m0, m0 + BM), nrep column tiles of BN from n_first on, walked one after the other
  // (pwi8s_kernel: all its columns); Co: one past the last column it may touch; pidx / npart: its range partial
  constexpr int WGN = 4 / WGM;
  constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
  constexpr int LDF = 17;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * TM * 32, wn = (wave % WGN) * TN * 32;
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  for (int rep = 0; rep < nrep; ++rep) {
    const int n0 = n_first + rep * BN;
    if (rep) __syncthreads();
    f32x16 accf[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) accf[i][j] = (f32x16){0};
    for (int k0 = 0; k0 < C; k0 += 16) {
      for (int q = tid; q < (BM + BN) * 4; q += 256) {     // one k quad of one row per item
        const bool isA = q < BM * 4;
        const int row = (isA ? q : q - BM * 4) >> 2, kq = (q & 3) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (isA) {
          const long m = min(m0 + row, M - 1);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + kq + e < C) v[e] = fake_quant(A[m * lda + k0 + kq + e], qs, qz);
        } else {
          const int co = min(n0 + row, Co - 1);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + kq + e < C) v[e] = Wp[(long)co * C + k0 + kq + e];
        }
        float *dst = (isA ? As : Bs) + row * LDF + kq;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = v[e];
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = As[(wm + i * 32 + (lane & 31)) * LDF + 2 * kk + (lane >> 5)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = Bs[(wn + j * 32 + (lane & 31)) * LDF + 2 * kk + (lane >> 5)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            accf[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], accf[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + wn + j * 32 + (lane & 31);
      const float bsv = (co < Co && bias) ? bias[co] : 0.f;
      const int oc = (co < Co && omap) ? omap[co] : co;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M && co < Co) {
            float v = accf[i][j][r] + bsv;
            if (relu) v = cdn::relu_keep_nan(v);
            if (R) R[m * ldo + oc] = v;
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            has_nan |= (v != v);
          }
        }
    }
  }
  if (rmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, pidx, npart, qu, red);
}

// (round 6: the kernel body as a device function of (m0, n0, partial index, partial count), so that the detection heads'
// launch -- three 64-column problems on one A operand, pwi8h_kernel below -- maps its workgroups itself)
template <int BM, int BN, int WGM, bool FAST>
__device__ __forceinline__ void
pwi8_body(const float *__restrict__ A, const unsigned *__restrict__ aq,
          const signed char *__restrict__ Wq, const float *__restrict__ wscale,
          const int *__restrict__ wsum, const float *__restrict__ Wp,
          const float *__restrict__ bias, float *__restrict__ R,
          float2 *rmm, const cdn::QUpdate &qu, long M, int C, int Cpad, int Co, int relu, int lda,
          int ldo, const int *__restrict__ omap, int Cw, const long m0, const int n0, const int pidx, const int npart) {
  // C: the K extent of the int8 path (the channels of A, or its padded row length when the pad repeats a real channel
  // against zero weight codes); Cw: the logical channel count = row length of the f32 weights Wp (wide-code branch)
  constexpr int WGN = 4 / WGM;
  constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
  constexpr int AI = BM * 8 / 256;        // float4 loads of A per thread per k-tile
  constexpr int BI = (BN * 2 + 255) / 256;  // 16-byte loads of W per thread per k-tile
  __shared__ __attribute__((aligned(16))) unsigned char A0[2][BM * kI8LD];
  __shared__ __attribute__((aligned(16))) unsigned char A1[2][BM * kI8LD];
  __shared__ __attribute__((aligned(16))) unsigned char B0[2][BN * kI8LD];
  __shared__ __attribute__((aligned(16))) unsigned char B1[2][BN * kI8LD];
  CDN_STAMPR(2, 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * TM * 32, wn = (wave % WGN) * TN * 32;
  // Round 6: the first k tile and the epilogue constants go out BEFORE the input quantiser's state is read -- the branch
  // on its wide-code flag below needs the state, and with the loads behind that branch every workgroup began with one
  // memory round trip for three words and only then asked for its data (a second trip; the workgroups of these launches
  // live for two to eight k tiles).  The wide path ignores what was loaded.
  const int lr = tid >> 3, lk = (tid & 7) * 4;      // A staging: row lr + 32*i, k quad lk
  const bool vec4 = !FAST && (C & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const int br = tid >> 1, bh = (tid & 1) * 16;     // B staging: row br + 128*i, 16-byte half bh
  float4 a[AI];
  i32x4 b[BI];
  const float *arow[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    long m = m0 + lr + 32 * i;
    if (m > M - 1) m = M - 1;
    arow[i] = A + m * lda + lk;
  }
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (FAST) {
        a[i] = *reinterpret_cast<const float4 *>(arow[i] + k0);
      } else if (vec4) {
        a[i] = (k0 + lk < C) ? *reinterpret_cast<const float4 *>(arow[i] + k0)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = k0 + lk;
        a[i].x = (k + 0 < C) ? arow[i][k0 + 0] : 0.0f;
        a[i].y = (k + 1 < C) ? arow[i][k0 + 1] : 0.0f;
        a[i].z = (k + 2 < C) ? arow[i][k0 + 2] : 0.0f;
        a[i].w = (k + 3 < C) ? arow[i][k0 + 3] : 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int co = n0 + br + 128 * i;
      if (co > Co - 1) co = Co - 1;
      b[i] = *reinterpret_cast<const i32x4 *>(Wq + (long)co * Cpad + k0 + bh);
    }
  };
  // (128-row tiles keep the old order: their 16 + 8 registers of tile and row pointers live across the branch took the
  // kernel from 108 to 140 VGPRs = from four to three workgroups per CU; they serve the small-M launches only)
  constexpr bool kEarly = BM <= 64;
  if (kEarly) load_tile(0);
  // Epilogue constants of this lane's TN output columns, requested NOW (branch-free, clamped column): read after
  // the k loop they were 2 dependent round trips per column with the workgroup idle (the ISA waited for bias /
  // scale / colsum, then for the channel map), ~3 us at the end of every launch.
  float e_bias[TN], e_ws[TN];
  int e_sum[TN], e_oc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn + j * 32 + (lane & 31);
    const int cc = min(co, Co - 1);
    e_ws[j] = wscale[cc];
    e_sum[j] = wsum[cc];
    e_bias[j] = 0.f;
    e_oc[j] = co;
  }
  if (bias) {
#pragma unroll
    for (int j = 0; j < TN; ++j) e_bias[j] = bias[min(n0 + wn + j * 32 + (lane & 31), Co - 1)];
  }
  if (omap) {
#pragma unroll
    for (int j = 0; j < TN; ++j) e_oc[j] = omap[min(n0 + wn + j * 32 + (lane & 31), Co - 1)];
  }
  const float qs = reinterpret_cast<const float *>(aq)[2];
  const float qz = reinterpret_cast<const float *>(aq)[3];
  if (__builtin_amdgcn_readfirstlane((int)aq[6])) {      // (workgroup-uniform; said so: the loads above stay live past this branch)
    // Codes too wide for the nibble split (the tracked range is far narrower than the batch: the first
    // ~100 calls of a fresh EMA): this batch runs on f32 MFMA with the fake-quantised weights, inside
    // the same launch (a separate fallback launch costs 4.3 us per stage even when it has nothing to
    // do).  Simple single-buffered 16-deep k-tiles in the int8 path's LDS arrays: the rare path.
    static_assert(BM * 17 * 4 <= 2 * BM * kI8LD && BN * 17 * 4 <= 2 * BN * kI8LD, "LDS reuse");
    pwi8_wide_path<BM, BN, WGM>(A, Wp, bias, R, rmm, qu, M, Cw, Co, relu, lda, ldo, omap, qs, qz,
                                reinterpret_cast<float *>(&A0[0][0]), reinterpret_cast<float *>(&B0[0][0]),
                                reinterpret_cast<float *>(&A1[0][0]), m0, n0, 1, pidx, npart);
    return;
  }
  // as_uint(t + 1.5*2^23) = 0x4B400000 + rint(t) for |t| < 2^22 (guaranteed when state[6] == 0)
  const int ioff = (int)qz + (2048 - 128) - 0x4B400000;
  i32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (i32x16){0};
  auto ucode = [&](float v, bool live) -> unsigned {
#pragma clang fp contract(off)
#if defined(CDN_DIAG) && CDN_DIAG == 7   // diagnostic build: no fp32 -> code arithmetic (wrong results)
    return (__float_as_uint(v) & 0xfffu) | 8u;
#endif
    // t = sc*d - zp (two roundings, as the reference); rint(t) by the 1.5*2^23 trick; then integer:
    // u = rint(t) + zp - 128 + 2048 in [8, 4087]
    const float y_p = qs * v;      // (plain operators under fp contract(off): two roundings, cdn_common.h)
  const float y = (y_p - qz) + 12582912.0f;
    int u = (int)__float_as_uint(y) + ioff;
    u = min(max(u, 8), 4087);       // |L - 128| <= 2040: never active unless state[6] lied
    return live ? (unsigned)u : 2048u;
  };
  auto store_tile = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int k = k0 + lk;
      const unsigned u0 = ucode(a[i].x, FAST || k + 0 < C), u1 = ucode(a[i].y, FAST || k + 1 < C);
      const unsigned u2 = ucode(a[i].z, FAST || k + 2 < C), u3 = ucode(a[i].w, FAST || k + 3 < C);
      const unsigned p01 = u0 | (u1 << 16), p23 = u2 | (u3 << 16);
      // a0 = u & 15 in [0,15];  a1 = (u >> 4) - 128 (byte ^ 0x80);  bytes {p01.b0, p01.b2, p23.b0, p23.b2}
      const unsigned lo = __builtin_amdgcn_perm(p23, p01, 0x06040200u) & 0x0F0F0F0Fu;
      const unsigned hi = __builtin_amdgcn_perm(p23 >> 4, p01 >> 4, 0x06040200u) ^ 0x80808080u;
      *reinterpret_cast<unsigned *>(&A0[buf][(lr + 32 * i) * kI8LD + lk]) = lo;
      *reinterpret_cast<unsigned *>(&A1[buf][(lr + 32 * i) * kI8LD + lk]) = hi;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if (br + 128 * i < BN) {
        i32x4 s16;   // 16*qw per byte: shift each byte left by 4 (qw in [-8,7] -> no carry loss)
#pragma unroll
        for (int e = 0; e < 4; ++e) s16[e] = (int)(((unsigned)b[i][e] << 4) & 0xF0F0F0F0u);
        *reinterpret_cast<i32x4 *>(&B0[buf][(br + 128 * i) * kI8LD + bh]) = b[i];
        *reinterpret_cast<i32x4 *>(&B1[buf][(br + 128 * i) * kI8LD + bh]) = s16;
      }
  };

  if (!kEarly) load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  CDN_STAMPR(2, 1);
  const int nk = (C + 31) / 32;   // (Cpad >= 32 * nk: the weight rows are zero padded to 64)
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) load_tile((t + 1) * 32);
    i32x4 a0[TM], a1[TM], b0[TN], b1[TN];
    const int fo = (lane & 31) * kI8LD + (lane >> 5) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      a0[i] = *reinterpret_cast<const i32x4 *>(&A0[buf][(wm + i * 32) * kI8LD + fo]);
      a1[i] = *reinterpret_cast<const i32x4 *>(&A1[buf][(wm + i * 32) * kI8LD + fo]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      b0[j] = *reinterpret_cast<const i32x4 *>(&B0[buf][(wn + j * 32) * kI8LD + fo]);
      b1[j] = *reinterpret_cast<const i32x4 *>(&B1[buf][(wn + j * 32) * kI8LD + fo]);
    }
    // (low nibbles for every accumulator, then the high parts: two MFMAs into the same accumulator back to back
    // wait for each other; the integer sum does not depend on the order)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[i], b1[j], acc[i][j], 0, 0, 0);
    if (t + 1 < nk) store_tile(buf ^ 1, (t + 1) * 32);
    __syncthreads();
  }
  CDN_STAMPR(2, 2);
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn + j * 32 + (lane & 31);
    const float bsv = e_bias[j], rinv = __fdiv_rn(1.0f, __fmul_rn(qs, e_ws[j]));
    const int t128 = 128 * e_sum[j], oc = e_oc[j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && co < Co) {
          float v = fmaf((float)(acc[i][j][r] + t128), rinv, bsv);
          if (relu) v = cdn::relu_keep_nan(v);
          if (R) R[m * ldo + oc] = v;                          // (R == NULL: range-only pass)
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
          has_nan |= (v != v);
        }
      }
  }
  CDN_STAMPR(2, 3);
  if (rmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, pidx, npart, qu,
                             reinterpret_cast<float *>(&A0[0][0]));
  CDN_STAMPR(2, 4);
}

template <int BM, int BN, int WGM, bool FAST>
__global__ void __launch_bounds__(256)
pwi8_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq,
            const signed char *__restrict__ Wq, const float *__restrict__ wscale,
            const int *__restrict__ wsum, const float *__restrict__ Wp,
            const float *__restrict__ bias, float *__restrict__ R,
            float2 *rmm, cdn::QUpdate qu, long M, int C, int Cpad, int Co, int relu, int lda,
            int ldo, const int *__restrict__ omap, int Cw) {
  pwi8_body<BM, BN, WGM, FAST>(A, aq, Wq, wscale, wsum, Wp, bias, R, rmm, qu, M, C, Cpad, Co, relu, lda, ldo, omap, Cw,
                               (long)blockIdx.x * BM, blockIdx.y * BN, blockIdx.y * gridDim.x + blockIdx.x,
                               gridDim.x * gridDim.y);
}

// pwi8h_kernel (round 6; VERDICT r5 "next" #1a): the FIRST 1x1 convs of the detection heads -- NH problems of 64 output
// columns each on ONE A operand (the last stage's output r, 67 MB at batch 64, which three separate launches on three
// streams read three times) -- as one launch.  Weights / scales / column sums / biases are the heads' concatenated
// (row h * 64 + co); head h writes its own [M][64] buffer R + h * head_stride (omap: column -> column % 64) and updates
// its own QuantAct (qus.q[h], own arrival counters: the range epilogue per column group).  A 1-D grid: the NH workgroups
// of one 64-row block get ids 8 apart -- one XCD under round-robin dispatch, dispatched together -- so the second and
// third reading of the rows are L2 hits.  Same body, same sums: bit-identical to the per-head launches.
struct QUpdateN {
  cdn::QUpdate q[4];
};
template <int NH>
__global__ void __launch_bounds__(256)
pwi8h_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq, const signed char *__restrict__ Wq,
             const float *__restrict__ wscale, const int *__restrict__ wsum, const float *__restrict__ Wp,
             const float *__restrict__ bias, float *__restrict__ R, float2 *rmm, QUpdateN qus, long M, int C, int Cpad,
             int relu, int lda, const int *__restrict__ omap, long head_stride, long part_stride) {
  const int ntm = (int)((M + 63) / 64), grp = 8 * NH, b = blockIdx.x, full = (ntm / 8) * grp;
  int mt, head;
  if (b < full) {
    const int lid = b % grp;
    head = lid >> 3;
    mt = (b - lid) / NH + (lid & 7);
  } else {
    const int q = b - full, rem = ntm - (full / NH);
    head = q / rem;
    mt = full / NH + q % rem;
  }
  pwi8_body<64, 64, 2, true>(A, aq, Wq, wscale, wsum, Wp, bias, R + (long)head * head_stride, rmm + (long)head * part_stride,
                             qus.q[head], M, C, Cpad, 64 * NH, relu, lda, 64, omap, C, (long)mt * 64, head * 64, mt, ntm);
}


// ------------------------------------------------------------------------------------------
// pwi8s_kernel (round 5): pwi8_kernel's arithmetic as a STREAMING kernel for long-K launches (stage 0: K = 1024) --
// pws_kernel's structure.  tools/probes/probe_rowstream.hip: the stage-0 pointwise's 67-MB A operand can be read in
// 12-15 us (4.3-5.4 TB/s) whatever the ring depth; pwi8_kernel takes 36 us there because each of its 32 dependent k
// tiles is the serial sum of a global round trip, the fp32 -> code conversion, LDS writes, a barrier, LDS reads and the
// MFMAs, with two workgroups per CU to overlap (DESIGN_HISTORY section 4.1).  Here a workgroup owns a 32-row block x
// 32 TN columns, its four WAVES are the four quarters of K and each runs its own pipeline without workgroup barriers:
//   A   32 rows x 32 k of fp32 d per window by LDS-DMA into a wave-private ring RR - 1 windows ahead (whole 128-byte
//       lines, XOR-swizzled source chunks: pws_kernel); the fragment -- lane (i, h): k in [16 h, 16 h + 16) of row i --
//       is read as four ds_read_b128 and converted IN REGISTERS to the two nibble operands (pwi8_kernel's ucode / perm
//       expressions: every element is converted once per workgroup);
//   B   the weight codes from a K-BLOCKED copy [window][column][32 B] (behind the row-major codes, made by the host:
//       CDN_X_WCODES_KB): the operand of a 32-column tile is one fully coalesced 1-KB load per window, straight into
//       registers, one window ahead; 16 * qw by a byte shift.
// The int32 partial sums of the four k quarters are added through LDS (exact: order-independent).  Same integer sums
// and the same epilogue expression as pwi8_kernel: bit-identical outputs and ranges (tests/test_gpu_parity.py).
// ncg = 2 (Co = 256): two workgroups per row block, 128 columns each, 8 ids apart -- the same XCD, dispatched
// together, so the second reading of the A rows is an L2 hit; 1024 three-per-CU workgroups instead of 512 two-per-CU
// ones, which is what lets prologues (4 us: cold instruction fetch + the first DMA round trip) and epilogues (7 us)
// of some workgroups overlap the k loops of others -- with one round of workgroups the launch was the plain sum.
// Wide codes (state[6]): pwi8_wide_path.
// vmcnt discipline: the weight loads are inline asm as well.  A compiler-visible load in this loop is waited for
// with a count the compiler derives WITHOUT the DMAs it cannot see, and in-order completion then makes every such
// wait a wait for the newest DMA (first version: 33 us).  Their destination registers are released by an empty asm
// that names them right behind the explicit s_waitcnt, and loaded UNCONDITIONALLY inside the loop (the last steps are
// peeled) -- a conditional load makes the register a phi, and the copy the compiler then inserts reads it before the
// wait (second version: wrong sums); tools/check_asm_loads.py scans the ISA for exactly that.
// ------------------------------------------------------------------------------------------
// the int8 part of pwi8s_kernel (a function of its own so that the kernel is a plain if / else of the two paths: with
// the wide branch written as an early return in front of it the compiler saw that branch's pending stores on a path
// into the prologue below and put an s_waitcnt vmcnt(0) between the first weight loads -- i.e. behind the first DMAs)
template <int TN, int RR>
__device__ __forceinline__ void pwi8s_body(const float *__restrict__ A, const signed char *__restrict__ Wkb,
                                           const float *__restrict__ wscale, const int *__restrict__ wsum,
                                           const float *__restrict__ bias, float *__restrict__ R, float2 *rmm,
                                           const cdn::QUpdate &qu, long M, int K, int Co, int relu, int lda, int ldo,
                                           const int *__restrict__ omap, int ncg, long m0, int cb, float qs, float qz,
                                           char *lds) {
  constexpr int kRing = RR * 4096;
  // Nothing is in flight here, and the compiler is told so with a REAL s_waitcnt vmcnt(0) (the builtin, which its
  // wait-count bookkeeping sees): the structurised control flow has an edge from the end of the wide branch to this
  // block, and for the stores pending on that edge it otherwise puts its own vmcnt(0) in front of the first asm load
  // that overwrites one of their registers -- in the middle of the prologue, behind the first DMAs.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int Cot = 32 * TN * ncg;                     // columns per window of the k-blocked codes (zero rows behind Co)
  const int ioff = (int)qz + (2048 - 128) - 0x4B400000;
  // k windows (32 channels) of this wave: whole PAIRS of windows, the first (K / 64) % 4 waves one pair more -- every
  // wave runs an even number of steps (the two weight-register sets alternate without a copy)
  const int pairs = K >> 6;
  const int nit = 2 * ((pairs >> 2) + (w < (pairs & 3) ? 1 : 0));
  const int win0 = 2 * (w * (pairs >> 2) + min(w, pairs & 3)), kbase = 32 * win0;
  // ---- A: DMA side --------------------------------------------------------------------------------------------------
  const float *abase = A + m0 * lda + kbase;
  unsigned aoff[4];
  {
    const int dr = lane >> 3, dc = lane & 7;
    const long rmax = M - 1 - m0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 8 * u + dr;
      const int rl = (int)(row < rmax ? row : rmax);
      aoff[u] = (unsigned)rl * (unsigned)lda * 4u + (unsigned)((dc ^ ((row >> 1) & 7)) * 16);
    }
  }
  const unsigned ring = cdn::lds_addr_uniform(lds) + (unsigned)w * kRing;
  auto issue = [&](int t) {
    const unsigned dst = ring + (unsigned)(t % RR) * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) cdn::glds16(abase, aoff[u] + (unsigned)t * 128u, dst + u * 1024);
  };
  // ---- B: one coalesced 16-byte load per lane, tile and window (inline asm: see the header) -----------------------------
  const unsigned long long wb = (unsigned long long)(Wkb + ((long)win0 * Cot + cb) * 32);
  const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)wb), whi = __builtin_amdgcn_readfirstlane((unsigned)(wb >> 32));
  const unsigned long long wbu = ((unsigned long long)whi << 32) | wlo;
  const unsigned bo0 = (unsigned)(i * 32 + 16 * h), wstep = (unsigned)Cot * 32u;
  i32x4 bA[TN], bB[TN];
  auto loadB = [&](i32x4 (&b)[TN], int t) {
    static_assert(TN <= 4, "one offset register: the tiles of a window within the 13-bit immediate");
    const unsigned v0 = bo0 + (unsigned)t * wstep;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(b[j]) : "v"(v0), "s"(wbu), "n"(1024 * j) : "memory");
  };
  auto release = [&](i32x4 (&b)[TN]) {      // names the registers a wait has made valid: no use moves above this point
    if constexpr (TN == 4)
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3 % TN]));
    else
      asm volatile("" : "+v"(b[0]), "+v"(b[1]));
  };
  auto ucode = [&](float v) -> unsigned {
#pragma clang fp contract(off)
    const float y_p = qs * v;      // (two roundings, as the reference: see pwi8_kernel)
    const float y = (y_p - qz) + 12582912.0f;
    int u = (int)__float_as_uint(y) + ioff;
    u = min(max(u, 8), 4087);
    return (unsigned)u;
  };
  i32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = (i32x16){0};
  const char *wring = lds + w * kRing;
  const int frow = i * 128, fsw = (i >> 1) & 7;
  // issue order (vmcnt counts DMAs and loads alike, completing in order):
  //   A(0) .. A(RR-3), B(0), A(RR-2) | step t: B(t+1), A(t+RR-1), wait, compute(t)
  // so that behind B(t) -- and A(t), older still -- there are A(t+RR-2), B(t+1), A(t+RR-1): TN + 8 in the loop
  CDN_STAMPR(2, 1);      // (in front of the first DMA: a store here would be one more count behind every wait)
#pragma unroll
  for (int t = 0; t < RR - 2; ++t) issue(t);
  loadB(bA, 0);
  issue(RR - 2);
  auto step = [&](int t, i32x4 (&bcur)[TN], i32x4 (&bnxt)[TN], auto more_b, auto more_a, auto prev_a) {
    // behind B(t) there are: A(t+RR-2) if the step before (or the prologue) issued it, B(t+1), A(t+RR-1)
    constexpr int MB = decltype(more_b)::value, MA = decltype(more_a)::value, PA = decltype(prev_a)::value;
    if constexpr (MB) loadB(bnxt, t + 1);
    if constexpr (MA) issue(t + RR - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TN * MB + 4 * MA + 4 * PA) : "memory");
    release(bcur);
    const char *slot = wring + (t % RR) * 4096;
    i32x4 a0, a1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(slot + frow + ((4 * h + q) ^ fsw) * 16);
      const unsigned u0 = ucode(v.x), u1 = ucode(v.y), u2 = ucode(v.z), u3 = ucode(v.w);
      const unsigned p01 = u0 | (u1 << 16), p23 = u2 | (u3 << 16);
      a0[q] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) & 0x0F0F0F0Fu);
      a1[q] = (int)(__builtin_amdgcn_perm(p23 >> 4, p01 >> 4, 0x06040200u) ^ 0x80808080u);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bcur[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      i32x4 s16;
#pragma unroll
      for (int e = 0; e < 4; ++e) s16[e] = (int)(((unsigned)bcur[j][e] << 4) & 0xF0F0F0F0u);
      acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, s16, acc[j], 0, 0, 0);
    }
  };
  // nit is even and >= RR: every step of the loop fetches the A window RR - 1 ahead and the weights one ahead; the last
  // steps are peeled with their own (static) counts -- no branch and no register copy anywhere near a wait
  static_assert(RR == 3, "the peeled tail is written for a ring of three windows");
  {
    const std::true_type y;
    const std::false_type n;
    int t = 0;
    for (; t + 2 < nit; t += 2) {
      step(t, bA, bB, y, y, y);
      step(t + 1, bB, bA, y, y, y);
    }
    step(t, bA, bB, y, n, y);
    step(t + 1, bB, bA, n, n, n);
  }
  CDN_STAMPR(2, 2);
  // ---- epilogue: the four k quarters' int32 partial tiles through LDS (exact), two 32-column tiles per round ----------
  // P[wave][column][row], rows contiguous: a lane's four consecutive accumulator registers are four consecutive rows
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  {
    int *P = reinterpret_cast<int *>(lds);
    constexpr int PLD = 36;                           // ints per column: 32 rows + 4 (16-byte aligned, bank-staggered)
    const int col = tid & 63, r0 = (tid >> 6) * 8;
    for (int j0 = 0; j0 < TN; j0 += 2) {
      __syncthreads();                                // (rings drained / the previous round's sums read)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const i32x16 &c = acc[j0 + jj];
          *reinterpret_cast<i32x4 *>(&P[((w * 64) + jj * 32 + i) * PLD + 8 * g + 4 * h]) =
              (i32x4){c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
      __syncthreads();
      const int co = cb + j0 * 32 + col;
      if (co < Co) {
        const float bsv = bias ? bias[co] : 0.0f;
        const float rinv = __fdiv_rn(1.0f, __fmul_rn(qs, wscale[co]));
        const int t128 = 128 * wsum[co], oc = omap ? omap[co] : co;
        i32x4 s[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const i32x4 p0 = *reinterpret_cast<const i32x4 *>(&P[(0 * 64 + col) * PLD + r0 + 4 * hh]);
          const i32x4 p1 = *reinterpret_cast<const i32x4 *>(&P[(1 * 64 + col) * PLD + r0 + 4 * hh]);
          const i32x4 p2 = *reinterpret_cast<const i32x4 *>(&P[(2 * 64 + col) * PLD + r0 + 4 * hh]);
          const i32x4 p3 = *reinterpret_cast<const i32x4 *>(&P[(3 * 64 + col) * PLD + r0 + 4 * hh]);
          s[hh] = (p0 + p1) + (p2 + p3);
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = r0 + rr;
          if (m0 + row < M) {
            float v = fmaf((float)(s[rr >> 2][rr & 3] + t128), rinv, bsv);
            if (relu) v = cdn::relu_keep_nan(v);
            if (R) R[(m0 + row) * ldo + oc] = v;
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            has_nan |= (v != v);
          }
        }
      }
    }
  }
  CDN_STAMPR(2, 3);
  if (rmm) {
    __syncthreads();
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, blockIdx.x, gridDim.x, qu,
                             reinterpret_cast<float *>(lds));
  }
  CDN_STAMPR(2, 4);
}

template <int TN, int RR>
__global__ void __launch_bounds__(256)
pwi8s_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq, const signed char *__restrict__ Wkb,
             const float *__restrict__ wscale, const int *__restrict__ wsum, const float *__restrict__ Wp,
             const float *__restrict__ bias, float *__restrict__ R, float2 *rmm, cdn::QUpdate qu, long M, int K,
             int Co, int relu, int lda, int ldo, const int *__restrict__ omap, int Cw, int ncg) {
  extern __shared__ float4 p8_lds[];
  char *lds = reinterpret_cast<char *>(p8_lds);
  CDN_STAMPR(2, 0);
  // workgroup -> (row block, column group): the ncg workgroups of a row block are 8 ids apart (one XCD)
  int rb = blockIdx.x, cg = 0;
  if (ncg == 2) {
    const int nrb8 = (int)((gridDim.x >> 1) & ~7u), b = blockIdx.x;
    if (b < 2 * nrb8) { rb = ((b >> 4) << 3) | (b & 7); cg = (b >> 3) & 1; }
    else { rb = nrb8 + ((b - 2 * nrb8) >> 1); cg = b & 1; }
  }
  const long m0 = (long)rb * 32;
  const int cb = cg * 32 * TN;                       // first column of this workgroup
  const float qs = reinterpret_cast<const float *>(aq)[2];
  const float qz = reinterpret_cast<const float *>(aq)[3];
  if (aq[6] == 0) {
    pwi8s_body<TN, RR>(A, Wkb, wscale, wsum, bias, R, rmm, qu, M, K, Co, relu, lda, ldo, omap, ncg, m0, cb, qs, qz, lds);
  } else {      // codes too wide for the nibble split: this batch on f32 MFMA (the rare path, as in pwi8_kernel)
    float *fl = reinterpret_cast<float *>(lds);
    const int cend = min(Co, cb + 32 * TN);
    pwi8_wide_path<32, 128, 1>(A, Wp, bias, R, rmm, qu, M, Cw, cend, relu, lda, ldo, omap, qs, qz, fl, fl + 32 * 17,
                               fl + (32 + 128) * 17, m0, cb, (32 * TN + 127) / 128, (int)blockIdx.x, (int)gridDim.x);
  }
}
// ------------------------------------------------------------------------------------------
// pwb3: the W4 pointwise conv on FINAL (already fake-quantised, or fp32) activations -- the first 1x1 of a
// ShuffleNetV2 unit, whose input channels carry different generations of the layer's running QuantAct and
// so have no common integer grid.  Exact products on the bf16 matrix cores:
//   x = hi + mid + lo   (three bf16 terms by truncation: 8 + 8 + 8 significant bits, the split is EXACT)
//   weights  qw in [-8, 7] are exact in bf16,  W' = qw / sw[co]
//   y[m][co] = (sum_c hi*qw + sum_c mid*qw + sum_c lo*qw) / sw[co] + b
// every product is exact in fp32 and the accumulation is fp32, as in the f32 MFMA kernel, at 3 of the
// 16x faster v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 v_mfma_f32_32x32x2f32 (measured: the f32 form
// was matrix-core bound at conv5, 191 us for 15.6 GFLOP).  Operand lane map: lane (r = l&31, h = l>>5)
// holds k = 8h..8h+7 of row r.  LDS rows: 32 k (64 B) + 16 B pad = 80 B, conflict-free ds_read_b128.
// One LDS buffer + register prefetch of the next k-tile (two barriers per tile; 35-40 KiB per workgroup).
// ------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int kB3LD = 80;

template <int BM, int BN, int WGM>
__global__ void __launch_bounds__(256)
pwb3_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq,
            const signed char *__restrict__ Wq, const float *__restrict__ wscale,
            const float *__restrict__ bias, float *__restrict__ R, float2 *rmm, cdn::QUpdate qu,
            long M, int C, int Cpad, int Co, int relu, int lda, int ldo,
            const unsigned char *__restrict__ agen, const int *__restrict__ omap) {
  constexpr int WGN = 4 / WGM;
  constexpr int TM = BM / (WGM * 32), TN = BN / (WGN * 32);
  constexpr int AI = BM * 8 / 256;          // float4 loads of A per thread per k-tile
  constexpr int BI = (BN * 2 + 255) / 256;  // 16-byte loads of W per thread per k-tile
  // agen != NULL: channel c of A is fake-quantised with the QuantAct state aq + 8 * agen[c] (the
  // generations of a layer's running block-output QuantAct)
  __shared__ float2 qt[kMixedMaxC];
  __shared__ __attribute__((aligned(16))) unsigned char Ah[BM * kB3LD];
  __shared__ __attribute__((aligned(16))) unsigned char Am[BM * kB3LD];
  __shared__ __attribute__((aligned(16))) unsigned char Al[BM * kB3LD];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * kB3LD];
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * TM * 32, wn = (wave % WGN) * TN * 32;
  const bool has_q = aq != nullptr;
  const bool mixed = agen != nullptr;
  float qs = 1.f, qz = 0.f;
  if (has_q && !mixed) {
    qs = reinterpret_cast<const float *>(aq)[2];
    qz = reinterpret_cast<const float *>(aq)[3];
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16){0};
  const int lr = tid >> 3, lk = (tid & 7) * 4;      // A staging: row lr + 32*i, k quad lk
  const bool vec4 = (C & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool vec2 = !vec4 && (C & 1) == 0 && (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 7) == 0;
  const int br = tid >> 1, bh = (tid & 1) * 16;     // B staging: row br + 128*i, 16 codes from bh
  float4 a[AI];
  i32x4 b[BI];
  const float *arow[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    long m = m0 + lr + 32 * i;
    if (m > M - 1) m = M - 1;
    arow[i] = A + m * lda + lk;
  }
  auto load_tile = [&](int k0) {
    const int k = k0 + lk;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (vec4) {
        a[i] = (k < C) ? *reinterpret_cast<const float4 *>(arow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (vec2) {
        const float2 z = make_float2(0.f, 0.f);
        const float2 lo = (k < C) ? *reinterpret_cast<const float2 *>(arow[i] + k0) : z;
        const float2 hi = (k + 2 < C) ? *reinterpret_cast<const float2 *>(arow[i] + k0 + 2) : z;
        a[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
      } else {
        a[i].x = (k + 0 < C) ? arow[i][k0 + 0] : 0.0f;
        a[i].y = (k + 1 < C) ? arow[i][k0 + 1] : 0.0f;
        a[i].z = (k + 2 < C) ? arow[i][k0 + 2] : 0.0f;
        a[i].w = (k + 3 < C) ? arow[i][k0 + 3] : 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int co = n0 + br + 128 * i;
      if (co > Co - 1) co = Co - 1;
      b[i] = *reinterpret_cast<const i32x4 *>(Wq + (long)co * Cpad + k0 + bh);
    }
  };
  // (a.hi16 << 16) | b.hi16
  auto pack_hi = [](unsigned a_, unsigned b_) -> unsigned { return __builtin_amdgcn_perm(a_, b_, 0x07060302u); };
  auto store_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      float v[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
      unsigned hb[4], mb[4], lb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (mixed) {
          const float2 q2 = qt[min(k0 + lk + e, C - 1)];
          v[e] = (k0 + lk + e < C) ? fake_quant(v[e], q2.x, q2.y) : 0.0f;
        } else if (has_q) {
          v[e] = (k0 + lk + e < C) ? fake_quant(v[e], qs, qz) : 0.0f;
        }
        hb[e] = __float_as_uint(v[e]);
        const float r1 = __fsub_rn(v[e], __uint_as_float(hb[e] & 0xFFFF0000u));   // exact
        mb[e] = __float_as_uint(r1);
        lb[e] = __float_as_uint(__fsub_rn(r1, __uint_as_float(mb[e] & 0xFFFF0000u)));   // exact, <= 8 bits
      }
      const int o = (lr + 32 * i) * kB3LD + lk * 2;
      *reinterpret_cast<uint2 *>(&Ah[o]) = make_uint2(pack_hi(hb[1], hb[0]), pack_hi(hb[3], hb[2]));
      *reinterpret_cast<uint2 *>(&Am[o]) = make_uint2(pack_hi(mb[1], mb[0]), pack_hi(mb[3], mb[2]));
      *reinterpret_cast<uint2 *>(&Al[o]) = make_uint2(pack_hi(lb[1], lb[0]), pack_hi(lb[3], lb[2]));
    }
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if (br + 128 * i < BN) {
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int word = b[i][e];
          const unsigned f0 = __float_as_uint((float)((word << 24) >> 24));
          const unsigned f1 = __float_as_uint((float)((word << 16) >> 24));
          const unsigned f2 = __float_as_uint((float)((word << 8) >> 24));
          const unsigned f3 = __float_as_uint((float)(word >> 24));
          w[2 * e] = pack_hi(f1, f0);
          w[2 * e + 1] = pack_hi(f3, f2);
        }
        unsigned char *dstp = &Bs[(br + 128 * i) * kB3LD + bh * 2];
        *reinterpret_cast<i32x4 *>(dstp) = (i32x4){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        *reinterpret_cast<i32x4 *>(dstp + 16) = (i32x4){(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
      }
  };

  const int nk = (C + 31) / 32;   // (Cpad >= 32 * nk: the weight rows are zero padded to 64)
  load_tile(0);
  if (mixed) {
    for (int c = tid; c < C; c += 256) {
      const float *sp = reinterpret_cast<const float *>(aq) + cdn::kQStateWords * (agen[c] == 255 ? 0 : agen[c]);   // (255: unused channel)
      qt[c] = make_float2(sp[2], sp[3]);
    }
    __syncthreads();
  }
  for (int t = 0; t < nk; ++t) {
    if (t) __syncthreads();       // the previous tile's fragment reads are done
    store_tile(t * 32);
    __syncthreads();
    if (t + 1 < nk) load_tile((t + 1) * 32);
    const int fo = (lane & 31) * kB3LD + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fh[TM], fm[TM], fl[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = (wm + i * 32) * kB3LD + fo + ks * 32;
        fh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(&Ah[o]));
        fm[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(&Am[o]));
        fl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(&Al[o]));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(&Bs[(wn + j * 32) * kB3LD + fo + ks * 32]));
      // smallest terms first
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[i], fb[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm[i], fb[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn + j * 32 + (lane & 31);
    float bsv = 0.f, rinv = 0.f;
    int oc = co;
    if (co < Co) {
      if (bias) bsv = bias[co];
      rinv = __fdiv_rn(1.0f, wscale[co]);
      if (omap) oc = omap[co];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && co < Co) {
          float v = fmaf(acc[i][j][r], rinv, bsv);
          if (relu) v = cdn::relu_keep_nan(v);
          R[m * ldo + oc] = v;
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
          has_nan |= (v != v);
        }
      }
  }
  if (rmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, reinterpret_cast<float *>(&Ah[0]));
}

// ------------------------------------------------------------------------------------------
// pwd3: the same arithmetic as pwb3 (exact bf16 x 3 split against 4-bit weight codes) as a STREAMING
// kernel for the shapes of the ShuffleNetV2 units (Co <= a few hundred, C <= 512): in pwb3 no two waves
// share a row of A (each wave owns 32 rows x the whole N tile), so staging A through LDS only re-shapes it
// -- and costs two barriers per 32 k with the load latency exposed between them (measured 44 us for
// 91 MB at layer 2).  Here a lane loads its MFMA operand straight from global memory: lane (r = l&31,
// h = l>>5) owns k = 32w + 16h .. + 15 of row r in window w -- 64 contiguous bytes, four dwordx4 loads
// (the k order inside a window is free as long as B agrees) -- two windows ahead in registers, no LDS,
// no barrier in the k loop.  B (the N tile's codes as bf16, [n][k]) and the per-channel quantiser table
// are staged in LDS once per workgroup.  A workgroup = 4 waves = 4 x 32 rows, one N tile of 32*TN columns.
// Fake-quantisation while loading: n = rint(s*x - z) + z (integer valued), x' = n / s by Markstein's
// sequence q0 = n*r, q = fma(fma(-q0, s, n), r, q0) with r = RN(1/s): the correctly rounded quotient in
// 3 instructions instead of the ~10 of the IEEE expansion (this loop is VALU-bound next to the MFMAs).
// ------------------------------------------------------------------------------------------
// (blockDim.x = 256, or 512 where the B tile leaves room for only ONE workgroup per CU -- layer4: 125 KB -- so that the CU
// still runs two waves per SIMD behind one staged tile: round 4)
template <int TN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(TN == 2 ? 3 : 2, TN == 2 ? 3 : 2)))
pwd3_kernel(const float *__restrict__ A, const unsigned *__restrict__ aq,
            const signed char *__restrict__ Wq, const float *__restrict__ wscale,
            const float *__restrict__ bias, float *__restrict__ R, float2 *rmm, cdn::QUpdate qu,
            long M, int C, int Cpad, int Co, int relu, int lda, int ldo,
            const unsigned char *__restrict__ agen, const int *__restrict__ omap, int ngen) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CDN_STAMPR(2, 0);
  const int nwin = (C + 31) >> 5, Kp = nwin * 32;
  const int ldb = Kp * 2 + 16;                               // bytes per B row: conflict-free ds_read_b128
  // quantiser table: three float arrays [Kp] (s | z | 1/s): a lane reads its 16 channels of a window as 4 + 4 + 4
  // ds_read_b128 issued together (one {s, z, r} record per channel cost 16 reads, each waited for: 1600 cycles
  // per half window in the ISA)
  float *qtab = reinterpret_cast<float *>(smem + (size_t)32 * TN * ldb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = (int)blockDim.x, nwv = nthr >> 6;
  const int n0 = blockIdx.y * 32 * TN;
  const bool has_q = aq != nullptr;
  auto pack_hi = [](unsigned a_, unsigned b_) -> unsigned { return __builtin_amdgcn_perm(a_, b_, 0x07060302u); };

  // ---- A stream: a wave walks the 32-row blocks rb = first, first + stride, ...; its windows (rb, w) form
  // one sequence that is prefetched PF windows ahead across block boundaries -------------------------
  const long nrb = (M + 31) >> 5;
  const long rb_first = (long)blockIdx.x * nwv + wave, rb_stride = (long)gridDim.x * nwv;
  constexpr int PF = 2;
  float4 buf[PF][4];
  long prb = rb_first;         // block and window of the NEXT prefetch
  // Round 4: only the 32-channel windows in which THIS column tile has a non-zero weight code are streamed.  In the
  // shuffle-free layout the pass-through half of a unit's input row meets zero weight columns (FusedBackbone.
  // _mixed_plan); its slots form a few runs, so 1-7 of the 4 / 8 / 15 windows of a layer-1 / 2 / 3 row hold nothing
  // but zeros: x * 0 adds an exact zero to every accumulator, skipping the window is bit-identical, and the A stream
  // (and the chain of dependent window loads) shrinks by that share.  The host marks such columns with generation 255 in
  // the per-unit a_gen array (FusedBackbone._mixed_plan); the mask is built from those bytes in the table staging.
  __shared__ unsigned s_wmask;
  if (tid == 0) s_wmask = 0u;
  // Round 6: everything of the prologue that depends on nothing is ISSUED here -- the first pass of the B tile's weight
  // codes and the epilogue's per-column constants -- so that their round trip runs under the generation bytes' and the
  // quantiser states' (two dependent trips) instead of after them: the prologue was three to four round trips of
  // ~2 us each with every workgroup of the launch (one resident set) waiting at the same time.
  const int chunks = Kp >> 4;                                // 16-code chunks per row
  const int nitems = 32 * TN * chunks;
  i32x4 cw0[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int q = tid + nthr * u;
    const int r_ = q / chunks, ch = q - r_ * chunks;
    cw0[u] = (i32x4){0, 0, 0, 0};
    if (q < nitems && n0 + r_ < Co) cw0[u] = *reinterpret_cast<const i32x4 *>(Wq + (long)(n0 + r_) * Cpad + ch * 16);
  }
  float bsv[TN], rinv[TN], ws_[TN];
  int oc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {                             // branch-free, clamped column
    const int cc = min(n0 + j * 32 + (lane & 31), Co - 1);
    ws_[j] = wscale[cc];
    bsv[j] = bias ? bias[cc] : 0.f;
    oc[j] = omap ? omap[cc] : cc;
  }
  // ngen in 1 .. 16 (the caller says how many generations a_qstate holds): ALL their states and the first pass of
  // generation bytes go out in the same round trip and the table is filled from LDS -- without it the states are loaded
  // at addresses the generation bytes give: a second, dependent trip
  __shared__ float2 s_states[16];
  const bool pre_states = has_q && agen != nullptr && ngen >= 1 && ngen <= 16;
  int gen0[2] = {0, 0};
  if (pre_states) {
#pragma unroll
    for (int u = 0; u < 2; ++u) gen0[u] = agen[min(tid + nthr * u, C - 1)];
    if (tid < 16)
      s_states[tid] = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(aq) +
                                                        cdn::kQStateWords * min(tid, ngen - 1) + 2);
  }
  __syncthreads();                    // (before the other waves OR their bits in)
  unsigned wmask = 0u, prem = 0u;     // the tile's window set; windows of the current block not yet prefetched
  auto load_next = [&](float4 (&d)[4]) {
    const bool live = prb < nrb;
    const long row = min(prb * 32 + (lane & 31), M - 1);
    const int pw_ = __builtin_ctz(prem);
    const int k = 32 * pw_ + 16 * (lane >> 5);
    // unconditional loads from clamped addresses (C % 4 == 0, C >= 4), zeroed afterwards: a load under a
    // per-lane condition is a branch + a wait of its own in the ISA
    const float *rowp = A + row * lda;
    float4 t4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t4[i] = *reinterpret_cast<const float4 *>(rowp + min(k + 4 * i, C - 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = (live && k + 4 * i < C) ? t4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    prem &= prem - 1;
    if (prem == 0u) {
      prem = wmask;
      prb += rb_stride;
    }
  };

  // ---- quantiser table and B tile -> LDS (loads batched: the prologue is latency, not work) -----------
  if (has_q) {
    // (a table entry per channel; both generation bytes first, then both states: 2 round trips per pass)
    for (int c0_ = 0; c0_ < Kp; c0_ += 2 * nthr) {
      float4 te[2];
      int gen[2] = {0, 0};
      if (agen) {
#pragma unroll
        for (int u = 0; u < 2; ++u) gen[u] = (pre_states && c0_ == 0) ? gen0[u] : agen[min(c0_ + tid + nthr * u, C - 1)];
#pragma unroll
        for (int u = 0; u < 2; ++u) {         // generation 255 = "this column meets only zero weight codes" (host)
          const int c = c0_ + tid + nthr * u;
          // a wave's 64 channels are two windows: one LDS atomic per wave (one per channel was ~1000 atomics on one word)
          const bool live = c < C && gen[u] != 255;
          const unsigned long long lv = __ballot(live);
          const int w0 = (c0_ + (tid & ~63) + nthr * u) >> 5;
          const unsigned bits = ((lv & 0xffffffffull) ? 1u << w0 : 0u) | ((lv >> 32) ? 2u << w0 : 0u);
          if (lane == 0 && bits) atomicOr(&s_wmask, bits);
          if (gen[u] == 255) gen[u] = 0;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float2 sz2 = pre_states ? s_states[min(gen[u], 15)]
                                      : *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(aq) +
                                                                          cdn::kQStateWords * gen[u] + 2);
        te[u] = make_float4(sz2.x, sz2.y, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = c0_ + tid + nthr * u;
        if (c < Kp) {
          qtab[c] = te[u].x;
          qtab[Kp + c] = te[u].y;
          qtab[2 * Kp + c] = c < C ? __fdiv_rn(1.0f, te[u].x) : 0.0f;
        }
      }
    }
  }
  // the window set is known after ONE round trip (the generation bytes): the A stream starts here, beside the staging of
  // the B tile below
  __syncthreads();
  wmask = (has_q && agen) ? __builtin_amdgcn_readfirstlane(s_wmask) : 0u;
  if (wmask == 0u) wmask = nwin >= 32 ? 0xffffffffu : ((1u << nwin) - 1u);      // no mask given: every window
#if defined(CDN_NO_WSKIP)               // A/B build: every window, as in round 3
  wmask = nwin >= 32 ? 0xffffffffu : ((1u << nwin) - 1u);
#endif
  prem = wmask;
  const int nlist = __builtin_popcount(wmask);
#pragma unroll
  for (int p_ = 0; p_ < PF; ++p_) load_next(buf[p_]);
  for (int q0 = tid; q0 < nitems; q0 += nthr * 4) {
    i32x4 cw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + nthr * u;
      const int r_ = q / chunks, ch = q - r_ * chunks;
      cw[u] = cw0[u];                                          // (first pass: loaded at the top of the kernel)
      if (q0 != tid) {
        cw[u] = (i32x4){0, 0, 0, 0};
        if (q < nitems && n0 + r_ < Co)
          cw[u] = *reinterpret_cast<const i32x4 *>(Wq + (long)(n0 + r_) * Cpad + ch * 16);
      }
    }

#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + nthr * u;
      if (q < nitems) {
        const int r_ = q / chunks, ch = q - r_ * chunks;
        unsigned w8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int word = cw[u][e];
          const unsigned f0 = __float_as_uint((float)((word << 24) >> 24));
          const unsigned f1 = __float_as_uint((float)((word << 16) >> 24));
          const unsigned f2 = __float_as_uint((float)((word << 8) >> 24));
          const unsigned f3 = __float_as_uint((float)(word >> 24));
          w8[2 * e] = pack_hi(f1, f0);
          w8[2 * e + 1] = pack_hi(f3, f2);
        }
        unsigned char *dp = smem + (size_t)r_ * ldb + ch * 32;
        *reinterpret_cast<i32x4 *>(dp) = (i32x4){(int)w8[0], (int)w8[1], (int)w8[2], (int)w8[3]};
        *reinterpret_cast<i32x4 *>(dp + 16) = (i32x4){(int)w8[4], (int)w8[5], (int)w8[6], (int)w8[7]};
      }
    }
  }
  // epilogue constants of this lane's TN output columns (loaded at the top of the kernel)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const bool live = n0 + j * 32 + (lane & 31) < Co;
    rinv[j] = live ? __fdiv_rn(1.0f, ws_[j]) : 0.f;
    if (!live) {
      bsv[j] = 0.f;
      oc[j] = -1;                                             // dead column
    }
  }
  __syncthreads();
  CDN_STAMPR(2, 1);

  const unsigned char *bbase = smem + (size_t)(lane & 31) * ldb + 32 * (lane >> 5);
  const float *qrow = qtab + 16 * (lane >> 5);
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not

  for (long rb = rb_first; rb < nrb; rb += rb_stride) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = (f32x16){0};
    unsigned crem = wmask;                                      // windows of this block not yet computed
    for (int w0 = 0; w0 < nlist; w0 += PF) {
#pragma unroll
      for (int p_ = 0; p_ < PF; ++p_) {
        if (w0 + p_ < nlist) {                                  // wave-uniform
          const int w = __builtin_ctz(crem);
          crem &= crem - 1;
          float v[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[4 * i] = buf[p_][i].x; v[4 * i + 1] = buf[p_][i].y;
            v[4 * i + 2] = buf[p_][i].z; v[4 * i + 3] = buf[p_][i].w;
          }
          load_next(buf[p_]);
          unsigned hb[16], mb[16], lb[16];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {                      // 8 channels at a time: 6 table reads in flight
            float ts[8], tz[8], tr[8];
            if (has_q) {
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const float4 s4 = *reinterpret_cast<const float4 *>(qrow + 32 * w + 8 * hh + 4 * i);
                const float4 z4 = *reinterpret_cast<const float4 *>(qrow + Kp + 32 * w + 8 * hh + 4 * i);
                const float4 r4 = *reinterpret_cast<const float4 *>(qrow + 2 * Kp + 32 * w + 8 * hh + 4 * i);
                ts[4 * i] = s4.x; ts[4 * i + 1] = s4.y; ts[4 * i + 2] = s4.z; ts[4 * i + 3] = s4.w;
                tz[4 * i] = z4.x; tz[4 * i + 1] = z4.y; tz[4 * i + 2] = z4.z; tz[4 * i + 3] = z4.w;
                tr[4 * i] = r4.x; tr[4 * i + 1] = r4.y; tr[4 * i + 2] = r4.z; tr[4 * i + 3] = r4.w;
              }
            }
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const int e = 8 * hh + e8;
              float x = v[e];
              if (has_q) {
#pragma clang fp contract(off)
                const float sx = ts[e8] * x;
                const float n = rintf(sx - tz[e8]) + tz[e8];
                const float q0 = __fmul_rn(n, tr[e8]);
                x = fmaf(fmaf(-q0, ts[e8], n), tr[e8], q0);
              }
              hb[e] = __float_as_uint(x);
              const float r1 = __fsub_rn(x, __uint_as_float(hb[e] & 0xFFFF0000u));
              mb[e] = __float_as_uint(r1);
              lb[e] = __float_as_uint(__fsub_rn(r1, __uint_as_float(mb[e] & 0xFFFF0000u)));
            }
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            i32x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ph[e] = (int)pack_hi(hb[8 * ks + 2 * e + 1], hb[8 * ks + 2 * e]);
              pm[e] = (int)pack_hi(mb[8 * ks + 2 * e + 1], mb[8 * ks + 2 * e]);
              pl[e] = (int)pack_hi(lb[8 * ks + 2 * e + 1], lb[8 * ks + 2 * e]);
            }
            const bf16x8 fh = __builtin_bit_cast(bf16x8, ph), fm = __builtin_bit_cast(bf16x8, pm),
                         fl = __builtin_bit_cast(bf16x8, pl);
            // lo, mid, hi terms in this order into every accumulator (the fp32 sums are those of the j-major
            // form), but interleaved over the TN accumulators: three MFMAs into one accumulator back to back stall
            // (TN = 2 sits at its 168-VGPR cap for three waves per SIMD: one B fragment at a time there)
            if constexpr (TN >= 4) {
              bf16x8 fb[TN];
#pragma unroll
              for (int j = 0; j < TN; ++j)
                fb[j] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const i32x4 *>(bbase + (size_t)j * 32 * ldb + 64 * w + 16 * ks));
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fb[j], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, fb[j], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fb[j], acc[j], 0, 0, 0);
            } else {
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const bf16x8 fb = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const i32x4 *>(bbase + (size_t)j * 32 * ldb + 64 * w + 16 * ks));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fb, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, fb, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fb, acc[j], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    if (rb == rb_first) CDN_STAMPR(2, 2);
    // The prefetch stream alternates between the two buffers item by item, the loop above indexes them by the
    // window's position in its pair (static register indexing).  With an ODD number of windows per block the two
    // disagree from the second block of a wave on (window 0 of the next block sits in buf[1]): swap them.  Without
    // this, layer4 (K = 464: 15 windows) computed every second 32-row block of a wave from swapped k windows
    // whenever a wave walked more than one block (M > 8192 rows: batch 64 at 512 x 512 only; found by
    // tests/test_harness.py::test_whole_network_512_batch64_fused_vs_module_path in round 3).
    if (nlist & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = buf[0][i];
        buf[0][i] = buf[1][i];
        buf[1][i] = t;
      }
    }
    const long mb0 = rb * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = mb0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && oc[j] >= 0) {
          float v = fmaf(acc[j][r], rinv[j], bsv[j]);
          if (relu) v = cdn::relu_keep_nan(v);
          R[m * ldo + oc[j]] = v;
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
          has_nan |= (v != v);
        }
      }
  }
  CDN_STAMPR(2, 3);
  if (rmm)
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), rmm, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, reinterpret_cast<float *>(smem));
  CDN_STAMPR(2, 4);
}

// ------------------------------------------------------------------------------------------
// Final materialisation for the consumer outside the fused path (the detection heads):
// out[n][c][2h+dy][2w+dx] = fq(r[n][h*W+w][c])   (nearest x2, NCHW, optional fake-quant).
// One workgroup = (n, one stored row h): reads W*C contiguous floats, transposes through LDS,
// writes 2 output rows per channel.  up == 0 writes the same resolution (plain NHWC -> NCHW).
// ------------------------------------------------------------------------------------------
template <bool RQ>
__global__ void __launch_bounds__(256)
unpack_kernel(const float *__restrict__ r, const unsigned *__restrict__ rq, float *__restrict__ out,
              int C, int H, int W, int up) {
  extern __shared__ float tile[];  // [W][C+1]
  const int n = blockIdx.y, h = blockIdx.x;
  const int ld = C + 1;
  float qs = 1.f, qz = 0.f, qr_ = 1.f;
  if (RQ) {
    qs = reinterpret_cast<const float *>(rq)[2];
    qr_ = __fdiv_rn(1.0f, qs);   // Markstein division in fake_quant_r
    qz = reinterpret_cast<const float *>(rq)[3];
  }
  const float *rp = r + ((long)n * H + h) * W * C;
  if ((C & 3) == 0) {
    for (int q = threadIdx.x; q < W * C / 4; q += 256) {
      const int px = (q * 4) / C, c = q * 4 - px * C;
      float4 v = *reinterpret_cast<const float4 *>(rp + (long)q * 4);
      if (RQ) {
        v.x = cdn::fake_quant_r(v.x, qs, qz, qr_);
        v.y = cdn::fake_quant_r(v.y, qs, qz, qr_);
        v.z = cdn::fake_quant_r(v.z, qs, qz, qr_);
        v.w = cdn::fake_quant_r(v.w, qs, qz, qr_);
      }
      float *tp = tile + px * ld + c;
      tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
    }
  } else {
    for (int q = threadIdx.x; q < W * C; q += 256) {
      const int px = q / C, c = q - px * C;
      float v = rp[q];
      if (RQ) v = cdn::fake_quant_r(v, qs, qz, qr_);
      tile[px * ld + c] = v;
    }
  }
  __syncthreads();
  const int f = 1 << up;
  const int Wo = W * f, Ho = H * f;
  if ((Wo & 3) == 0) {   // 16-byte stores: 4 consecutive output columns per thread
    const int wq = Wo >> 2, per_c = f * wq;
    for (int q = threadIdx.x; q < C * per_c; q += 256) {
      const int c = q / per_c, rem = q - c * per_c;
      const int dy = rem / wq, xo = (rem - dy * wq) * 4;
      float4 v;
      v.x = tile[((xo + 0) >> up) * ld + c];
      v.y = tile[((xo + 1) >> up) * ld + c];
      v.z = tile[((xo + 2) >> up) * ld + c];
      v.w = tile[((xo + 3) >> up) * ld + c];
      *reinterpret_cast<float4 *>(out + (((long)n * C + c) * Ho + h * f + dy) * Wo + xo) = v;
    }
  } else {
    const int per_c = f * Wo;
    for (int q = threadIdx.x; q < C * per_c; q += 256) {
      const int c = q / per_c, rem = q - c * per_c;
      const int dy = rem / Wo, xo = rem - dy * Wo;
      out[(((long)n * C + c) * Ho + h * f + dy) * Wo + xo] = tile[(xo >> up) * ld + c];
    }
  }
}

using cdn::kMaxPartials;

// dynamic LDS of dw2_kernel / dw2u_kernel: the leading zero cell (dw2u), (Hl+1) x (Wl+1) cells of CCH floats, the
// chunk's depthwise weights, the scale plane, reduction scratch
static size_t dw2_lds_bytes(int Hl, int Wl, int CCH) {
  return ((size_t)CCH + (size_t)(Hl + 1) * (Wl + 1) * CCH + (size_t)CCH * 9 + (size_t)Hl * Wl +
          2 * kDw2MaxThreads / 64 + 4) * sizeof(float);
}

// gather schedule for an NCHW input (`gmode`, a PER-CALL argument: bits 8-9 of the stage entry points' layout
// argument, CDN_X_GATHER_* in codenet_dcn.h; the library keeps no process-wide setting): 0 = automatic (persistent
// LDS-DMA kernel where it applies), 1 = always the per-item kernels (dw2_kernel), 2 = persistent wherever its shape
// conditions hold (also on small grids)

static size_t dw0p_lds_bytes(int H, int W) {
  return ((size_t)64 + (size_t)(H + 1) * (W + 1) * 64 + (size_t)64 * H * W + 2 * 64 * 9 + 2 * (size_t)H * W +
          2 * 16 + 4) * sizeof(float);
}
// shape conditions of dw0p_kernel (see there)
static bool dw0p_applies(int C, int H, int W) {      // C: the row length of d (a multiple of 64 >= the channels of x)
  const int HW = H * W;
  return (W & 3) == 0 && (HW & 63) == 0 && (HW & (HW - 1)) == 0 && (C & 63) == 0 &&
         dw0p_lds_bytes(H, W) <= 160 * 1024;
}
template <bool OUT8>
static int launch_dw0p(const float *x, const float *s_raw, const unsigned *sq, const float *wd, float *d,
                       float2 *dmm, cdn::QUpdate qu, int N, int C, int H, int W, hipStream_t st, int Cx = 0) {
  if (Cx == 0) Cx = C;
  const int nitems = N * (C / 64);
  const int grid = std::min(nitems, cdn::kCUs);
  const size_t lds = dw0p_lds_bytes(H, W);
#define CDN_GOP(SQ_)                                                                                      \
  {                                                                                                       \
    auto kern = dw0p_kernel<SQ_, OUT8>;                                                                   \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    kern<<<grid, 1024, lds, st>>>(x, s_raw, sq, wd, d, dmm, qu, C, H, W, nitems, Cx);                     \
  }
  if (sq) CDN_GOP(true)
  else CDN_GOP(false)
#undef CDN_GOP
  return cdn::check_launch("codenet fused dw (persistent)");
}

// ------------------------------------------------------------------------------------------
// dwg: the gather + depthwise for stored planes that do NOT fit LDS even in 8-channel chunks (inputs above
// ~1100 px; round 4, VERDICT r3 missing #5): no staging, the four corners of every tap are read from global
// memory / L2 with bounds tests -- dw_kernel<false> of the module path (codenet_stage.hip) in the fused
// schedule's layout: channels-last d, x NCHW or channels-last (optionally stored at half resolution and
// fake-quantised while loading), s fake-quantised while loading, range partials.  A size fallback: one lane per
// (pixel, channel quad), the per-channel expressions of dw_kernel in the same order.
// ------------------------------------------------------------------------------------------
template <bool NHWC_IN, bool XQ, bool SQ>
__global__ void __launch_bounds__(256)
dwg_kernel(const float *__restrict__ x, const unsigned *__restrict__ xq, const float *__restrict__ s_raw,
           const unsigned *__restrict__ sq, const float *__restrict__ wd, float *__restrict__ d, float2 *mm,
           cdn::QUpdate qu, int C, int H, int W, int up, long total) {
  __shared__ float red[16];
  const int Hl = H >> up, Wl = W >> up, HWl = Hl * Wl, HW = H * W, CQ = (C + 3) >> 2;
  float xs = 1.f, xz = 0.f, xr_ = 1.f, ss = 1.f, sz = 0.f;
  if (XQ) {
    xs = reinterpret_cast<const float *>(xq)[2];
    xz = reinterpret_cast<const float *>(xq)[3];
    xr_ = __fdiv_rn(1.0f, xs);
  }
  if (SQ) {
    ss = reinterpret_cast<const float *>(sq)[2];
    sz = reinterpret_cast<const float *>(sq)[3];
  }
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
    const int cq = (int)(q % CQ);
    const long pn = q / CQ;
    const int n = (int)(pn / HW), p = (int)(pn - (long)n * HW);
    const int h = p / W, w = p - h * W, c = cq * 4;
    const int nc = min(4, C - c);
    float sv = s_raw[(long)n * HWl + (h >> up) * Wl + (w >> up)];
    if (SQ) sv = fake_quant(sv, ss, sz);
    const float t = sv - 1.0f;
    const Axis ya = make_axis(h - 1, -t, H), yb = make_axis(h + 1, t, H);
    const Axis xa = make_axis(w - 1, -t, W), xb = make_axis(w + 1, t, W);
    Axis ym, xm;
    ym.i0 = h; ym.w0 = 1.0f; ym.w1 = 0.0f; ym.ok = true;
    xm.i0 = w; xm.w0 = 1.0f; xm.w1 = 0.0f; xm.ok = true;
    // the four channels of full-resolution cell (yy, xx): zero outside the image
    auto rd = [&](int yy, int xx) __attribute__((always_inline)) -> float4 {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const int cell = (yy >> up) * Wl + (xx >> up);
        if (NHWC_IN) {
          if (nc == 4) {
            v = *reinterpret_cast<const float4 *>(x + ((long)n * HWl + cell) * C + c);
          } else {
            const float *xp = x + ((long)n * HWl + cell) * C + c;
            v.x = xp[0];
            if (nc > 1) v.y = xp[1];
            if (nc > 2) v.z = xp[2];
          }
        } else {
          const float *xp = x + ((long)n * C + c) * HWl + cell;
          v.x = xp[0];
          if (nc > 1) v.y = xp[HWl];
          if (nc > 2) v.z = xp[2 * (long)HWl];
          if (nc > 3) v.w = xp[3 * (long)HWl];
        }
        if (XQ) {
          v.x = cdn::fake_quant_r(v.x, xs, xz, xr_);
          v.y = cdn::fake_quant_r(v.y, xs, xz, xr_);
          v.z = cdn::fake_quant_r(v.z, xs, xz, xr_);
          v.w = cdn::fake_quant_r(v.w, xs, xz, xr_);
        }
      }
      return v;
    };
    auto tap = [&](const Axis &Y, const Axis &X) __attribute__((always_inline)) -> float4 {
      const float4 v00 = rd(Y.i0, X.i0), v01 = rd(Y.i0, X.i0 + 1), v10 = rd(Y.i0 + 1, X.i0), v11 = rd(Y.i0 + 1, X.i0 + 1);
      const float w00 = Y.w0 * X.w0, w01 = Y.w0 * X.w1, w10 = Y.w1 * X.w0, w11 = Y.w1 * X.w1;
      float4 r;
      r.x = (w00 * v00.x + w01 * v01.x) + w10 * v10.x + w11 * v11.x;
      r.y = (w00 * v00.y + w01 * v01.y) + w10 * v10.y + w11 * v11.y;
      r.z = (w00 * v00.z + w01 * v01.z) + w10 * v10.z + w11 * v11.z;
      r.w = (w00 * v00.w + w01 * v01.w) + w10 * v10.w + w11 * v11.w;
      return r;
    };
    const float4 v[9] = {tap(ya, xa), tap(ya, xm), tap(ya, xb), tap(ym, xa), rd(h, w), tap(ym, xb),
                         tap(yb, xa), tap(yb, xm), tap(yb, xb)};
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float *wk = wd + (long)min(c + e, C - 1) * 9;
      const float ve[9] = {(&v[0].x)[e], (&v[1].x)[e], (&v[2].x)[e], (&v[3].x)[e], (&v[4].x)[e],
                           (&v[5].x)[e], (&v[6].x)[e], (&v[7].x)[e], (&v[8].x)[e]};
      float a = wk[0] * ve[0];
#pragma unroll
      for (int k = 1; k < 9; ++k) a = fmaf(wk[k], ve[k], a);
      acc[e] = a;
    }
    float *dp = d + ((long)n * HW + p) * C + c;
    if (nc == 4 && (C & 3) == 0) {
      *reinterpret_cast<float4 *>(dp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      for (int e = 0; e < nc; ++e) dp[e] = acc[e];
    }
    for (int e = 0; e < nc; ++e) {
      mn = fminf(mn, acc[e]);
      mx = fmaxf(mx, acc[e]);
      has_nan |= (acc[e] != acc[e]);
    }
  }
  if (mm) cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), mm, blockIdx.x, gridDim.x, qu, red);
}

int launch_dwg(bool nhwc, const float *x, const unsigned *xq, const float *s_raw, const unsigned *sq, const float *wd,
               float *d, float2 *dmm, cdn::QUpdate qu, int N, int C, int H, int W, int up, hipStream_t st) {
  const long total = (long)N * H * W * ((C + 3) / 4);
  const unsigned blocks = (unsigned)std::min<long>(cdn::ceil_div(total, 256), kMaxPartials);
#define CDN_GOG(NH, XQ_, SQ_) dwg_kernel<NH, XQ_, SQ_><<<blocks, 256, 0, st>>>(x, xq, s_raw, sq, wd, d, dmm, qu, C, H, W, up, total)
  const bool XQ = xq != nullptr, SQ = sq != nullptr;
  if (nhwc) {
    if (XQ && SQ) CDN_GOG(true, true, true);
    else if (XQ) CDN_GOG(true, true, false);
    else if (SQ) CDN_GOG(true, false, true);
    else CDN_GOG(true, false, false);
  } else {
    if (SQ) CDN_GOG(false, false, true);
    else CDN_GOG(false, false, false);
  }
#undef CDN_GOG
  return cdn::check_launch("codenet fused dw (global gather)");
}

template <int CCH>
int launch_dw2(bool nhwc, const float *x, const unsigned *xq, const float *s_raw,
               const unsigned *sq, const float *wd, float *d, float2 *dmm, cdn::QUpdate qu, int N,
               int C, int H, int W, int up, hipStream_t st, int gmode, int ldd = 0,
               cdn::ScaleFromSums si = cdn::ScaleFromSums{nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, 0}) {
  const int Hl = H >> up, Wl = W >> up;
  const size_t lds = dw2_lds_bytes(Hl, Wl, CCH);
  dim3 grid((unsigned)cdn::ceil_div(C, CCH), (unsigned)N);
  // d with rows padded to a multiple of 64 channels (stage_fused_forward decides; a ragged channel count, CoDeNet2x)
  if (ldd > C) return launch_dw0p<false>(x, s_raw, sq, wd, d, dmm, qu, N, ldd, H, W, st, C);
  // NCHW input at output resolution (stage 0): the persistent LDS-DMA form once every CU gets >= 2 items to pipeline
  if (CCH == 64 && !nhwc && up == 0 && xq == nullptr && gmode != 1 && dw0p_applies(C, H, W) &&
      (gmode == 2 || (long)grid.x * grid.y >= 2L * cdn::kCUs))
    return launch_dw0p<false>(x, s_raw, sq, wd, d, dmm, qu, N, C, H, W, st);
  // two 512-thread workgroups per CU when LDS allows and the grid is large enough to fill them
  // (staging of one overlaps compute of the other); otherwise one 1024-thread workgroup per CU.
  const bool two_per_cu = lds * 2 <= 160 * 1024 && (long)grid.x * grid.y >= 2L * cdn::kCUs;
  const int threads = two_per_cu ? 512 : 1024;
  // (tried and removed: two 256-thread workgroups per CU at up to 256 VGPRs with all 25 reads of a step in flight,
  // 62 us vs 56 us at stage 0; a persistent, double-buffered form -- next item's global loads in flight during
  // the gather -- 66 us: at 2 waves/SIMD the per-step LDS latency chain is exposed.  DESIGN.md section 4.1)
#define CDN_GO(NH, XQ_, SQ_)                                                                  \
  {                                                                                           \
    auto kern = dw2_kernel<CCH, NH, XQ_, SQ_, kDw2MaxThreads>;                                \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                      \
    kern<<<grid, threads, lds, st>>>(x, xq, s_raw, sq, wd, d, dmm, qu, C, H, W, up);          \
  }
  const bool XQ = xq != nullptr, SQ = sq != nullptr;
  const bool blocks = nhwc && up == 1;
  if (blocks) {   // 2x2-block kernel for up-sampled inputs
#define CDN_GOU(XQ_, SQ_)                                                                     \
  {                                                                                           \
    auto kern = dw2u_kernel<CCH, XQ_, SQ_>;                                                   \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                      \
    kern<<<grid, 512, lds, st>>>(x, xq, s_raw, sq, wd, d, dmm, qu, C, H, W, si);              \
  }
    if (XQ && SQ) CDN_GOU(true, true)
    else if (XQ) CDN_GOU(true, false)
    else if (SQ) CDN_GOU(false, true)
    else CDN_GOU(false, false)
#undef CDN_GOU
  } else if (nhwc) {
    if (XQ && SQ) CDN_GO(true, true, true)
    else if (XQ) CDN_GO(true, true, false)
    else if (SQ) CDN_GO(true, false, true)
    else CDN_GO(true, false, false)
  } else {
    if (XQ && SQ) CDN_GO(false, true, true)
    else if (XQ) CDN_GO(false, true, false)
    else if (SQ) CDN_GO(false, false, true)
    else CDN_GO(false, false, false)
  }
#undef CDN_GO
  return cdn::check_launch("codenet fused dw");
}

}  // namespace

// ---- launchers of the frozen-range schedule (codenet_frozen.hip): the same scale / gather kernels with byte
// codes on one or both sides.  x_kind: 0 = NCHW fp32 final values, 1 = channels-last fp32 pre-quantisation
// values + their quantiser state xq, 2 = channels-last byte codes of the quantiser xq.  The kernel choice
// mirrors cdn_codenet_stage_fused_forward, so that every fp32 sum is formed in the same order (bit-identical
// s_raw and d codes).
int cdn::launch_frozen_scale(const void *x, int x_kind, const unsigned *xq, const float *w_scale,
                             const float *b_scale, float *s_raw, int64_t N, int64_t C, int64_t HWl, float lo,
                             float hi, hipStream_t st) {
  const cdn::QUpdate none{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 8, 0};
  const float *xf = static_cast<const float *>(x);
  const long npix = (long)(N * HWl);
  if (x_kind == 0) {
    dim3 grid((unsigned)cdn::ceil_div(HWl, 64), (unsigned)N);
    scale_nchw_kernel<<<grid, kScaleWaves * 64, 0, st>>>(xf, w_scale, b_scale, s_raw, nullptr, none, (int)C,
                                                          (int)HWl, lo, hi);
  } else if (C <= 256 && npix >= 32768 && cdn::ceil_div(npix, kScaleTilePix) <= cdn::kMaxPartials) {
    const int blocks = (int)cdn::ceil_div(npix, kScaleTilePix);
    const int CQ = (int)C >> 2, LD = ((CQ + 31) & ~31) + 4;
    const size_t lds = ((size_t)kScaleTilePix * LD + 16) * sizeof(float);
    if (x_kind == 2)
      scale_nhwc_tile_kernel<true, true><<<blocks, 256, lds, st>>>(xf, xq, w_scale, b_scale, s_raw, nullptr, none,
                                                                   (int)C, npix, lo, hi);
    else
      scale_nhwc_tile_kernel<true><<<blocks, 256, lds, st>>>(xf, xq, w_scale, b_scale, s_raw, nullptr, none, (int)C,
                                                             npix, lo, hi);
  } else {
    const int blocks = (int)std::min<long>(cdn::ceil_div(npix, 4), (long)cdn::kCUs * 8);
    if (x_kind == 2)
      scale_nhwc_kernel<true, true><<<blocks, 256, 0, st>>>(xf, xq, w_scale, b_scale, s_raw, nullptr, none, (int)C,
                                                            npix, lo, hi);
    else
      scale_nhwc_kernel<true><<<blocks, 256, 0, st>>>(xf, xq, w_scale, b_scale, s_raw, nullptr, none, (int)C, npix,
                                                      lo, hi);
  }
  return cdn::check_launch("codenet frozen scale");
}

namespace {
template <int CCH>
int launch_frozen_dw_t(const float *x, int x_kind, const unsigned *xq, const float *s_raw, const unsigned *sq,
                       const float *wd, float *d8, unsigned *dstate, float2 *oflow, int N, int C, int H, int W,
                       int up, hipStream_t st, cdn::ScaleFromSums si, int gmode) {
  const int Hl = H >> up, Wl = W >> up;
  const size_t lds = dw2_lds_bytes(Hl, Wl, CCH);
  dim3 grid((unsigned)cdn::ceil_div(C, CCH), (unsigned)N);
  const bool two_per_cu = lds * 2 <= 160 * 1024 && (long)grid.x * grid.y >= 2L * cdn::kCUs;
  const int threads = two_per_cu ? 512 : 1024;
  const cdn::QUpdate qu{nullptr, nullptr, dstate, nullptr, 0.f, 0.f, 8, 0};
#define CDN_FGO(KERN, THREADS, ...)                                                                         \
  {                                                                                                         \
    auto kern = KERN;                                                                                       \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    kern<<<grid, THREADS, lds, st>>>(x, xq, s_raw, sq, wd, d8, oflow, qu, C, H, W, __VA_ARGS__);            \
  }
  if (x_kind != 0 && up == 1) {
    if (x_kind == 2) CDN_FGO((dw2u_kernel<CCH, true, true, true, true>), 512, si)
    else CDN_FGO((dw2u_kernel<CCH, true, true, false, true>), 512, si)
  } else if (x_kind == 2) {
    CDN_FGO((dw2_kernel<CCH, true, true, true, kDw2MaxThreads, true, true>), threads, up)
  } else if (x_kind == 1) {
    CDN_FGO((dw2_kernel<CCH, true, true, true, kDw2MaxThreads, false, true>), threads, up)
  } else if (CCH == 64 && up == 0 && gmode != 1 && dw0p_applies(C, H, W) &&
             (gmode == 2 || (long)grid.x * grid.y >= 2L * cdn::kCUs)) {
    return launch_dw0p<true>(x, s_raw, sq, wd, d8, oflow, qu, N, C, H, W, st);
  } else {
    CDN_FGO((dw2_kernel<CCH, false, false, true, kDw2MaxThreads, false, true>), threads, up)
  }
#undef CDN_FGO
  return cdn::check_launch("codenet frozen gather");
}
}  // namespace

// The gather's channel chunk for THIS launch: the LDS-fit chunk (stage_channel_chunk), thinned while the grid is below one
// workgroup per CU.  cfg2 (32 images of 256 x 256) ran stage 1 on 128 and stage 2 on 64 workgroups of 64 channels -- half
// and a quarter of the GPU (in-kernel stamps, round 6); with chunks of 32 / 16: gather 13.0 -> 11.4 and 21.6 -> 12.7 us,
// step 0.0937 -> 0.0835 ms; CoDeNet2x at 32 images per GPU 0.2154 -> 0.2026 ms.  Thinner than one workgroup per CU, or
// chunks of 8, lose again (0.0866 / 0.0909).  The gather is per channel: the chunking changes no value.
#ifndef CDN_THIN_CHUNK_MIN
#define CDN_THIN_CHUNK_MIN 16      // (64 and 32 channels fetch the tap record with DPP, 16 and 8 with ds_bpermute)
#endif
#ifndef CDN_THIN_CHUNK_FILL
#define CDN_THIN_CHUNK_FILL 1      // workgroups per CU the thinning aims at (A/B: 2 loses at every shape tried)
#endif
int cdn::thin_channel_chunk(int cch, int64_t C, int64_t N) {
#ifndef CDN_NO_THIN_CHUNKS
  while (cch > CDN_THIN_CHUNK_MIN && cdn::ceil_div(C, (int64_t)cch) * N < (int64_t)CDN_THIN_CHUNK_FILL * cdn::kCUs) cch >>= 1;
#endif
  return cch;
}

int cdn::launch_frozen_dw(const void *x, int x_kind, const unsigned *xq, const float *s_raw, const unsigned *sq,
                          const float *wd, signed char *d8, unsigned *dstate, unsigned *oflow, int N, int C, int H,
                          int W, int up, hipStream_t st, cdn::ScaleFromSums si, int gmode) {
  if (si.sums && !(x_kind != 0 && up == 1))
    return cdn::fail(CDN_ERR_UNSUPPORTED, "scale sums are consumed by the up-sampled channels-last gather only");
  int cch = cdn::stage_channel_chunk(H >> up, W >> up);
  if (cch == 0) return cdn::fail(CDN_ERR_UNSUPPORTED, "stored plane too large for the LDS-resident gather");
  cch = cdn::thin_channel_chunk(cch, C, N);      // (few workgroups: see stage_fused_forward_impl)
  if ((long)cdn::ceil_div(C, cch) * N > cdn::kMaxPartials) return cdn::fail(CDN_ERR_UNSUPPORTED, "too many workgroups");
  auto fn = cch == 64 ? launch_frozen_dw_t<64> : cch == 32 ? launch_frozen_dw_t<32>
            : cch == 16 ? launch_frozen_dw_t<16> : launch_frozen_dw_t<8>;
  return fn(static_cast<const float *>(x), x_kind, xq, s_raw, sq, wd, reinterpret_cast<float *>(d8), dstate,
            reinterpret_cast<float2 *>(oflow), N, C, H, W, up, st, si, gmode);
}

// Pointwise (1x1) convolution on a channels-last activation A [M][C] -> R [M][Co]: int8 MFMA on codes
// when the A quantiser state and the integer weights are given, f32 MFMA otherwise.  Shared by the
// stage schedule and the stand-alone entry point (detection heads).
// pws_kernel instead of pw3_kernel: plain f32 operands, K a multiple of 32 with 16-byte aligned rows, and a launch
// whose pw3 tiling (64- or 128-row tiles x 128 / 64 columns) would leave the chip with at most one workgroup per CU
// -- the latency-chain regime (cfg2: all three stages).  CDN_PWS_MAX_TILES: A/B switch.
#ifndef CDN_PWS_MAX_TILES
#define CDN_PWS_MAX_TILES (cdn::kCUs)
#endif
static bool pws_applies(long M, int64_t K, int64_t Co, int64_t lda, const float *a, const float *w) {
  if (K < 32 || (K & 31) || (lda & 3) || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15))
    return false;
  const long tiles = cdn::ceil_div(M, Co > 64 ? 64 : 128) * cdn::ceil_div(Co, Co > 64 ? 128 : 64);
  return tiles <= (long)CDN_PWS_MAX_TILES && cdn::ceil_div(M, 32) * cdn::ceil_div(Co, 32) <= (1L << 20);
}

// tile width (TN) and K split (KS) of pws_kernel as TN * 10 + KS: the widest tile and the least K splitting that give
// >= 4 waves per CU -- (64 columns, whole K), (32, whole K), (64, K / 4), (32, K / 4)
static int pws_choice(long M, int64_t C, int64_t Co) {
  const long want = 4L * cdn::kCUs, mt32 = cdn::ceil_div(M, 32);
  const long w21 = mt32 * cdn::ceil_div(Co, 64), w11 = mt32 * cdn::ceil_div(Co, 32);
  const bool can_split = (C & 127) == 0;
  if (w21 >= want || (!can_split && Co > 32)) return 21;
  if (w11 >= want || !can_split) return 11;
  if (4 * w21 >= want) return 24;
  return 14;
}

// pwi8s_kernel instead of pwi8_kernel: K >= 512 whole 64-channel blocks (every wave an even number of >= 2 windows),
// Co <= 256, the k-blocked copy of the codes given (CDN_X_WCODES_KB).  Measured (round 5, one box, interleaved):
// CoDeNet1x stage 0 (16384 x 1024 -> 256) 37.5 -> 31 us, CoDeNet2x stage 0 (8192 x 2176 -> 256) 47.6 -> 25.9 us.
// CDN_NO_PWI8S: A/B switch.
static int64_t wcodes_kb_columns(int64_t Kpad, int64_t Co) {
#if defined(CDN_NO_PWI8S)
  return 0;
#endif
  if (Kpad < 512 || (Kpad & 63) || Co > 256 || Co < 1) return 0;
  return Co <= 64 ? 64 : Co <= 128 ? 128 : 256;
}
static bool pwi8s_applies(long M, int64_t Kt, int Cpad, int64_t Co, bool pw_fast, const signed char *w_kb) {
  return w_kb != nullptr && pw_fast && Kt == Cpad && wcodes_kb_columns(Kt, Co) != 0 && M >= 1 &&
         2 * cdn::ceil_div(M, 32) <= (long)kMaxPartials;
}
extern "C" int64_t cdn_codenet_wcodes_kb_columns(int64_t C, int64_t Co) {
  return C > 0 ? wcodes_kb_columns((C + 63) / 64 * 64, Co) : 0;
}
extern "C" int64_t cdn_codenet_wcodes_kb_offset(int64_t C, int64_t Co) {
  return C > 0 && Co > 0 ? (Co * ((C + 63) / 64 * 64) + 255) / 256 * 256 : 0;
}

static int launch_pointwise(const float *d, unsigned *dst, long M, int64_t C, int64_t Co,
                            const float *w_pw, const signed char *w_pw_codes, const float *w_pw_scale,
                            const int *w_pw_colsum, const float *bias_pw, const float *ep_scale,
                            const float *ep_shift, int relu, float *r_out, float2 *rmm,
                            const cdn::QUpdate &qu_r, int ptag, hipStream_t st, int64_t lda = 0,
                            int64_t ldo = 0, const unsigned char *a_gen = nullptr,
                            const int *out_map = nullptr, bool a_padded = false,
                            const signed char *w_kb = nullptr, const float *next_ws = nullptr,
                            float *sparts = nullptr, int n_gens = 0) {
  // n_gens: how many states a_gen can name (0: unknown) -- pwd3_kernel then loads them all at once
  // next_ws / sparts: chained fp32 stages (pws_kernel only; the caller asked cdn_codenet_stage_chain_parts first)
  // w_kb: the k-blocked copy of w_pw_codes (include/codenet_dcn.h, CDN_X_WCODES_KB) or NULL
  // a_padded: the rows of A hold lda = round_up(C, 64) valid floats (the pad repeats channel C - 1) and the weight
  // codes are zero beyond C: the int8 path runs its whole-tile form over K = lda; the f32 branch for wide codes keeps C
  if (lda == 0) lda = C;      // row strides of A / R in floats (views into wider channels-last tensors)
  if (ldo == 0) ldo = Co;
  // tile choice: keep >= 2 workgroups per CU when M is small (stage 0), wide N tiles otherwise
  const int pw_bn = Co > 64 ? 128 : 64;
  const int pw_bm = (Co > 64 && cdn::ceil_div(M, 128) * cdn::ceil_div(Co, 128) <= cdn::kCUs) ? 64 : 128;
  const int n_part_r = (int)(cdn::ceil_div(M, pw_bm) * cdn::ceil_div(Co, pw_bn));
  CDN_REQUIRE(n_part_r <= kMaxPartials, CDN_ERR_UNSUPPORTED, "too many pointwise workgroups");
  const int64_t Kt = a_padded ? lda : C;
  const bool pw_fast = (Kt % 32) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0;
#define CDN_PW1(BM_, BN_, WGM_, AQ_, FAST_)                                                      \
  pw3_kernel<BM_, BN_, WGM_, AQ_, FAST_><<<(unsigned)std::min<long>(                             \
      cdn::ceil_div(M, BM_) * cdn::ceil_div(Co, BN_), only_if_wide ? 2L * cdn::kCUs : (1L << 30)), 256, 0, st>>>( \
      d, dst, w_pw, bias_pw, ep_scale, ep_shift, r_out, rmm, qu_r, M, (int)C, (int)Co, relu, only_if_wide, \
      (int)lda, (int)ldo)
#define CDN_PW(BM_, BN_, WGM_, AQ_)                                       \
  do {                                                                    \
    if (pw_fast) CDN_PW1(BM_, BN_, WGM_, AQ_, true);                      \
    else CDN_PW1(BM_, BN_, WGM_, AQ_, false);                             \
  } while (0)
  const bool use_i8 = w_pw_codes != nullptr && dst != nullptr && ep_scale == nullptr && a_gen == nullptr;
  // final-valued input (no QuantAct state to derive integer codes from) with 4-bit weight codes: exact
  // bf16 x 3 split instead of f32 MFMA
  const bool use_b3 = w_pw_codes != nullptr && ep_scale == nullptr && w_pw_scale != nullptr &&
                      (a_gen != nullptr || dst == nullptr);
  CDN_REQUIRE(a_gen == nullptr || (use_b3 && dst != nullptr && C <= kMixedMaxC), CDN_ERR_UNSUPPORTED,
              "a mixed-generation input needs the states, 4-bit weight codes and C <= %d", kMixedMaxC);
  CDN_REQUIRE(out_map == nullptr || use_i8 || use_b3, CDN_ERR_UNSUPPORTED,
              "an output channel map needs the 4-bit weight codes");
  const int only_if_wide = 0;
  if (use_i8) {
    CDN_REQUIRE(w_pw_scale && w_pw_colsum, CDN_ERR_ARG, "int8 pointwise needs scale and colsum");
    CDN_REQUIRE((reinterpret_cast<uintptr_t>(w_pw_codes) & 15) == 0, CDN_ERR_ARG,
                "w_pw_codes must be 16-byte aligned");
    const int Cpad = (int)((C + 63) / 64 * 64);
    CDN_REQUIRE(!a_padded || lda == Cpad, CDN_ERR_ARG, "a padded A has round_up(C, 64) floats per row");
    cdn::ProfScope ps(cdn::kProfPointwise, ptag, st);
#define CDN_PWI(BM_, BN_, WGM_)                                                                  \
  do {                                                                                           \
    dim3 g((unsigned)cdn::ceil_div(M, BM_), (unsigned)cdn::ceil_div(Co, BN_));                   \
    if (pw_fast)                                                                                 \
      pwi8_kernel<BM_, BN_, WGM_, true><<<g, 256, 0, st>>>(d, dst, w_pw_codes, w_pw_scale,        \
                                                            w_pw_colsum, w_pw, bias_pw, r_out, rmm, \
                                                            qu_r, M, (int)Kt, Cpad, (int)Co, relu, (int)lda, (int)ldo, out_map, (int)C); \
    else                                                                                         \
      pwi8_kernel<BM_, BN_, WGM_, false><<<g, 256, 0, st>>>(d, dst, w_pw_codes, w_pw_scale,       \
                                                             w_pw_colsum, w_pw, bias_pw, r_out, rmm, \
                                                             qu_r, M, (int)Kt, Cpad, (int)Co, relu, (int)lda, (int)ldo, out_map, (int)C); \
  } while (0)
    // long-K launches with the k-blocked copy of the codes at hand (the fused stage, stage 0): the streaming kernel
    if (pwi8s_applies(M, Kt, Cpad, Co, pw_fast, w_kb)) {
      const long nrb = cdn::ceil_div(M, 32);
      const int ncg = Co > 128 ? 2 : 1;
      const unsigned grid = (unsigned)(nrb * ncg);
      constexpr int kLds = 4 * 3 * 4096;
      auto kern = Co <= 64 ? pwi8s_kernel<2, 3> : pwi8s_kernel<4, 3>;
      kern<<<grid, 256, kLds, st>>>(d, dst, w_kb, w_pw_scale, w_pw_colsum, w_pw, bias_pw, r_out, rmm, qu_r, M, (int)Kt,
                                    (int)Co, relu, (int)lda, (int)ldo, out_map, (int)C, ncg);
      return cdn::check_launch("codenet int8 pointwise (streaming)");
    }
    // Co > 64: 64-row tiles (36 KiB LDS, 112 VGPRs: four workgroups per CU; measured at stage 1
    // 26.3 us vs 30.6 us with 128-row tiles; 32-row tiles change nothing at stage 0: 40.2 vs 40.7 us)
    // round 4: when 64-row tiles leave the chip with at most ONE workgroup per CU (CoDeNet2x stage 0 at 32 images:
    // M = 8192, K = 2176 -- a chain of 68 dependent k tiles per workgroup and nothing beside it), 32-row tiles give
    // every CU two chains to interleave
#if !defined(CDN_NO_PWI32)
    if (pw_bn == 128 && cdn::ceil_div(M, 64) * cdn::ceil_div(Co, 128) <= cdn::kCUs && Kt >= 1024) CDN_PWI(32, 128, 1);
    else
#endif
    // ... and 32-row tiles for the K < 256 launches of layers 2-3 (unit exits, the stride-2 units' branch convs; round 4):
    // whole network 2.564 / 2.603 -> 2.544 / 2.533 ms on one box.  NOT for K = 256 (stage 1 of the deform path: 0.2442 ->
    // 0.2500 ms per step with them) and not at K >= 512, where nothing moved.
#if !defined(CDN_PWI_NO_32ROWS)
    if (pw_bn == 128 && Kt < 256) CDN_PWI(32, 128, 1);
    else
#endif
    if (pw_bn == 128) CDN_PWI(64, 128, 2);
    // Co <= 64 (layer 1's units, the heads' first conv, stage 2): 64 x 64 tiles on the large-M launches (round 4).  Alone
    // on the GPU the 128 x 64 tiles are as fast (33.1 vs 31.6 us at 262 144 x 58 -> 58, 38.4 vs 39.9 at K = 128), inside
    // the network -- beside the other branch / the other heads -- the smaller tiles are worth 38 us per batch (whole
    // network 2.594-2.599 -> 2.552-2.561 ms, three interleaved pairs on one box; the deform step itself: unchanged)
#if !defined(CDN_PWI_NO_64X64)
    else if (M >= 65536) CDN_PWI(64, 64, 2);
#endif
    else CDN_PWI(128, 64, 4);
#undef CDN_PWI
    // (wide codes, state[6] != 0, are handled by the f32 branch inside pwi8_kernel)
  } else if (use_b3) {
    CDN_REQUIRE((reinterpret_cast<uintptr_t>(w_pw_codes) & 15) == 0, CDN_ERR_ARG,
                "w_pw_codes must be 16-byte aligned");
    const int Cpad = (int)((C + 63) / 64 * 64);
    cdn::ProfScope ps(cdn::kProfPointwise, ptag, st);
#define CDN_PWB(BM_, BN_, WGM_)                                                                  \
  do {                                                                                           \
    dim3 g((unsigned)cdn::ceil_div(M, BM_), (unsigned)cdn::ceil_div(Co, BN_));                   \
    pwb3_kernel<BM_, BN_, WGM_><<<g, 256, 0, st>>>(d, dst, w_pw_codes, w_pw_scale, bias_pw, r_out, \
                                                   rmm, qu_r, M, (int)C, Cpad, (int)Co, relu,    \
                                                   (int)lda, (int)ldo, a_gen, out_map);          \
  } while (0)
    // streaming form when the rows are 16-byte aligned quads and the N tile's weights fit in LDS
    int tn = Co > 64 ? 4 : 2;
    const int Kp = (int)((C + 31) / 32 * 32);
#if !defined(CDN_PWD3_NO_TN2_SMALLM)
    // Round 6: 64-column tiles where the 128-column tile's B (> 80 KB: K = 464) leaves ONE workgroup per CU and M is too
    // small for its 512-thread form -- layer 4's units at batch 64 ran 256 four-wave workgroups, one wave per SIMD; with
    // 64 columns two workgroups share a CU and the grid doubles (the A rows are read four times instead of twice, from
    // L2): 16384 x 464 -> 232 38.0 -> 35.6 us, whole network 2.509 -> 2.495 ms (three interleaved pairs)
    if (tn == 4 && (size_t)32 * 4 * (Kp * 2 + 16) + (size_t)Kp * 16 > 80 * 1024 &&
        cdn::ceil_div(M, 256) * cdn::ceil_div(Co, 128) < cdn::kCUs)
      tn = 2;
#endif
    const size_t lds_d3 = (size_t)32 * tn * (Kp * 2 + 16) + (size_t)Kp * 16;
    const long nblk_d3 = cdn::ceil_div(M, 128) * cdn::ceil_div(Co, 32 * tn);
    if ((C & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 &&
        lds_d3 <= 150 * 1024 && nblk_d3 <= kMaxPartials) {
      // persistent: as many workgroups as stay resident (VGPRs, LDS), each walks row blocks
      const unsigned ny = (unsigned)cdn::ceil_div(Co, 32 * tn);
      const long per_cu = std::min<long>(tn == 2 ? 3 : 2, (long)(160 * 1024 / (lds_d3 + 512)));   // 168 / 256 VGPRs
      const long gx = std::max<long>(1, std::min<long>(cdn::ceil_div(M, 128), per_cu * cdn::kCUs / ny));
      dim3 g((unsigned)gx, ny);
      auto kern = tn == 4 ? pwd3_kernel<4> : pwd3_kernel<2>;
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d3);
      // one workgroup per CU (a B tile above 80 KB: layer4): 8 waves behind the one staged tile instead of 4
#if defined(CDN_PWD3_256)
      const int d3_threads = 256;
#else
      // (only where 256-row workgroups still fill the chip: layer 3's units have 64 x 2 of them and keep 256 threads)
      const int d3_threads = (per_cu == 1 && cdn::ceil_div(M, 256) * ny >= cdn::kCUs) ? 512 : 256;
#endif
      const long gx2 = std::max<long>(1, std::min<long>(cdn::ceil_div(M, d3_threads / 2), per_cu * cdn::kCUs / ny));
      g = dim3((unsigned)gx2, ny);
      kern<<<g, d3_threads, lds_d3, st>>>(d, dst, w_pw_codes, w_pw_scale, bias_pw, r_out, rmm, qu_r, M, (int)C, Cpad,
                                          (int)Co, relu, (int)lda, (int)ldo, a_gen, out_map, n_gens);
    } else if (pw_bn == 128 && pw_bm == 64) CDN_PWB(64, 128, 2);
    else if (pw_bn == 128) CDN_PWB(128, 128, 4);
    else CDN_PWB(128, 64, 4);
#undef CDN_PWB
  } else if (!dst && a_gen == nullptr && out_map == nullptr && !a_padded && pws_applies(M, C, Co, lda, d, w_pw)) {
    // few output tiles: streaming waves (pws_kernel) -- the widest tile and the least K splitting that give >= 4 waves
    // per CU: (64 columns, whole K), (32, whole K), (64, K / 4), (32, K / 4)
    cdn::ProfScope ps(cdn::kProfPointwise, ptag, st);
#define CDN_PWS(TN_, KS_)                                                                                         \
  do {                                                                                                            \
    auto kern = pws_kernel<TN_, KS_>;                                                                             \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PwsGeom<TN_>::kLds); \
    kern<<<(unsigned)(cdn::ceil_div(M, KS_ == 4 ? 32 : 128) * cdn::ceil_div(Co, 32 * TN_)), 256, PwsGeom<TN_>::kLds, st>>>( \
        d, w_pw, bias_pw, ep_scale, ep_shift, r_out, rmm, qu_r, M, (int)C, (int)Co, relu, (int)lda, (int)ldo,      \
        next_ws, sparts);                                                                                         \
  } while (0)
    const int choice = pws_choice(M, C, Co);
    if (choice == 21) CDN_PWS(2, 1);
    else if (choice == 11) CDN_PWS(1, 1);
    else if (choice == 24) CDN_PWS(2, 4);
    else CDN_PWS(1, 4);
#undef CDN_PWS
  } else if (sparts) {
    return cdn::fail(CDN_ERR_UNSUPPORTED, "chained scale sums need the streaming f32 pointwise (cdn_codenet_stage_chain_parts)");
  } else {
    cdn::ProfScope ps(cdn::kProfPointwise, ptag, st);
    if (pw_bn == 128 && pw_bm == 64) {
      if (dst) CDN_PW(64, 128, 2, true); else CDN_PW(64, 128, 2, false);
    } else if (pw_bn == 128) {
      if (dst) CDN_PW(128, 128, 4, true); else CDN_PW(128, 128, 4, false);
    } else {
      if (dst) CDN_PW(128, 64, 4, true); else CDN_PW(128, 64, 4, false);
    }
  }
#undef CDN_PW
#undef CDN_PW1
  return cdn::check_launch("codenet fused pointwise");
}

// --act-percentile (CDN_X_ACT_PERCENTILE): the radix select's histograms + the two order statistics it returns
constexpr int64_t kPctBytes = 32 * 1024;

// Range commit of a percentile QuantAct: the producer (launched with running = 0) left the batch's TRUE extremes in
// state words [4], [5]; the range follows the order statistics `pct` (quant_modules.py:203-219), the code-width flag
// the extremes.
__global__ void quantact_commit_percentile_kernel(cdn::QUpdate u, const float *__restrict__ pct) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float *sf = reinterpret_cast<const float *>(u.state);
    cdn::quantact_update_device(u, pct[0], pct[1], true, false, 0.f, 0.f, true, sf[4], sf[5]);
  }
}

extern "C" int cdn_quantact_commit_range(float *x_min, float *x_max, void *state, const float *range, int bits,
                                         double momentum, int running, void *stream) {
  CDN_REQUIRE(x_min && x_max && state && range, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits out of range");
  const cdn::QUpdate u{x_min, x_max, static_cast<unsigned *>(state), nullptr, (float)(momentum - 1.0),
                       (float)(1.0 - momentum), bits, running};
  quantact_commit_percentile_kernel<<<1, 64, 0, cdn::as_stream(stream)>>>(u, range);
  return cdn::check_launch("quantact commit range");
}

extern "C" size_t cdn_codenet_stage_workspace_bytes(int64_t N, int64_t C, int64_t H, int64_t W,
                                                    int x_up) {
  const int64_t HWl = (H >> x_up) * (W >> x_up);
  // s_raw [N*HWl] + d [N*H*W*C] + 3 regions of per-workgroup {min,max} partials, each rounded
  // up to 256 bytes
  auto r = [](int64_t b) { return (b + 255) / 256 * 256; };
  const int64_t Cd = (C + 63) / 64 * 64;        // (rows of d may be padded to whole 64-channel chunks)
  return (size_t)(r(N * HWl * 4) + r(N * H * W * Cd * 4) + r(kPctBytes) + 3 * r(kMaxPartials * 8) +
                  3 * r(cdn::kArriveWords * 4));
}

// LDS budget of the gather kernel decides the channel chunk: 64 or 32 channels x the whole stored plane at the
// BASELINE resolutions (planes up to 34 x 34, inputs up to 544 px); larger planes keep the whole-plane form -- the tap
// offsets reach +-8 pixels, so a spatial tile would need a +-9-cell halo and re-read 2-2.6x the bytes -- with thinner
// chunks: 16 channels (4 lanes per pixel, planes up to ~49 x 49) or 8 (2 lanes per pixel, up to ~69 x 69, inputs up to
// ~1100 px).  Thin chunks fetch the tap record with ds_bpermute instead of DPP and read 64- / 32-byte pieces of every
// cell row (bank conflicts, partial lines): a size fallback, not a tuned path.  0: the plane does not fit.
int cdn::stage_channel_chunk(int Hl, int Wl) {
  const size_t cells = (size_t)(Hl + 1) * (Wl + 1) + 1;   // + the zero row and zero column, + the leading zero cell
  const long lds_max = 160 * 1024 - 64 * 9 * 4 - 256 - (long)Hl * Wl * 4;   // scale plane, weights, scratch
  if (lds_max <= 0) return 0;
  for (int cch = 64; cch >= 8; cch >>= 1)
    if (cells * cch * 4 <= (size_t)lds_max) return cch;
  return 0;
}

extern "C" int cdn_codenet_stage_supported(int64_t N, int64_t C, int64_t H, int64_t W, int x_nhwc, int x_up) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (x_up != 0 && x_up != 1)) return 0;
  if (x_up && ((H & 1) || (W & 1))) return 0;
  if (x_nhwc && (C & 3)) return 0;
  if (x_up && !x_nhwc) return 0;
  if (N > 65535 || N * C * H * W >= (1ll << 31)) return 0;
  if ((H >> x_up) > 4096 || (W >> x_up) > 4096) return 0;
  const int cch = cdn::stage_channel_chunk((int)(H >> x_up), (int)(W >> x_up));
  return cch != 0 && cdn::ceil_div(C, cch) * N <= kMaxPartials;
}

// cdn_codenet_stage_fused_forward's own limits: as above, plus the global-memory gather for stored planes that do not
// fit LDS (round 4) -- any plane the 32-bit element counts allow.
extern "C" int cdn_codenet_stage_fused_supported(int64_t N, int64_t C, int64_t H, int64_t W, int x_nhwc, int x_up) {
  if (cdn_codenet_stage_supported(N, C, H, W, x_nhwc, x_up)) return 1;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (x_up != 0 && x_up != 1)) return 0;
  if (x_up && ((H & 1) || (W & 1))) return 0;
  if (x_nhwc && (C & 3)) return 0;
  if (x_up && !x_nhwc) return 0;
  if (N > 65535 || N * C * H * W >= (1ll << 31)) return 0;
  if ((H >> x_up) > 4096 || (W >> x_up) > 4096) return 0;
  return cdn::stage_channel_chunk((int)(H >> x_up), (int)(W >> x_up)) == 0;      // (too many chunk workgroups: no)
}

// Rows of d padded to whole 64-channel chunks?  (NCHW input with a ragged channel count feeding the int8 pointwise.)
static bool stage_pads_d(int64_t N, int64_t C, int64_t H, int64_t W, int x_nhwc, int x_up, int gmode, bool int8_pw) {
  const int64_t Cd = (C + 63) / 64 * 64;
  return !x_nhwc && x_up == 0 && Cd != C && gmode != 1 && int8_pw && cdn::stage_channel_chunk((int)H, (int)W) == 64 &&
         dw0p_applies((int)Cd, (int)H, (int)W) && N * Cd * H * W < (1ll << 31);
}

// Diagnostics / tests: where cdn_codenet_stage_fused_forward leaves its two intermediates in the workspace -- the
// clamped scale plane s (pre-quantisation, [N][stored pixels]) at byte offset 0 and the gather output d
// (pre-quantisation, channels-last rows of *d_row_floats floats, of which the first C are the channels) at
// *d_offset_bytes.  int8_pointwise: the call passes the integer form of the pointwise weights and a d quantiser
// (W4A8 without --act-percentile).  Both stay valid until the next call that uses the workspace.
extern "C" int cdn_codenet_stage_fused_intermediates(int64_t N, int64_t C, int64_t H, int64_t W, int x_nhwc, int x_up,
                                                     int int8_pointwise, int64_t *d_offset_bytes,
                                                     int64_t *d_row_floats) {
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && d_offset_bytes && d_row_floats, CDN_ERR_ARG, "bad argument");
  const int gmode = (x_nhwc & CDN_X_GATHER_MASK) >> 8;
  const int64_t HWl = (H >> x_up) * (W >> x_up);
  *d_offset_bytes = (N * HWl * 4 + 255) / 256 * 256;
  *d_row_floats = stage_pads_d(N, C, H, W, x_nhwc & 1, x_up, gmode, int8_pointwise != 0 && !(x_nhwc & CDN_X_ACT_PERCENTILE))
                      ? (C + 63) / 64 * 64 : C;
  return CDN_OK;
}

// parts_in / n_parts_in: the scale prediction of THIS stage as the previous stage's partial sums (no scale launch);
// next_w_scale / parts_out: leave the next stage's partial sums (cdn_codenet_stage_fused_forward_chain)
static int stage_fused_forward_impl(
    const float *x, int x_nhwc, int x_up, const void *x_qstate, int64_t N, int64_t C, int64_t Co,
    int64_t H, int64_t W, const float *w_scale, const float *b_scale, float lo, float hi,
    const float *w_dw, const float *w_pw, const signed char *w_pw_codes, const float *w_pw_scale,
    const int *w_pw_colsum, const float *bias_pw, const float *ep_scale, const float *ep_shift,
    int relu, float *s_min, float *s_max, void *s_state, float *d_min,
    float *d_max, void *d_state, float *r_min, float *r_max, void *r_state, int bits,
    double momentum, int running, void *workspace, size_t workspace_bytes, float *r_out,
    void *stream, const float *parts_in, int n_parts_in, const float *next_w_scale, float *parts_out) {
  CDN_REQUIRE(x && w_scale && w_dw && w_pw && r_out && workspace, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && Co > 0 && H > 0 && W > 0, CDN_ERR_ARG, "non-positive size");
  CDN_REQUIRE((x_nhwc & ~(1 | CDN_X_GATHER_MASK | CDN_X_ACT_PERCENTILE | CDN_X_WCODES_KB | CDN_X_DEFER_RANGE |
                          CDN_X_PHASE_MASK)) == 0 &&
                  ((x_nhwc & CDN_X_GATHER_MASK) >> 8) <= 2, CDN_ERR_ARG,
              "x_nhwc: 0 / 1, optionally | CDN_X_GATHER_PER_ITEM or CDN_X_GATHER_PERSISTENT, | CDN_X_ACT_PERCENTILE, "
              "| CDN_X_WCODES_KB, | CDN_X_DEFER_RANGE, | CDN_X_PHASE_*");
  // split call (multi-process global ranges): which steps run, and whether the producers only measure
  const int phases = (x_nhwc & CDN_X_PHASE_MASK) ? (x_nhwc & CDN_X_PHASE_MASK) : CDN_X_PHASE_MASK;
  const bool defer = (x_nhwc & CDN_X_DEFER_RANGE) != 0;
  CDN_REQUIRE(!defer || (running != 0 && s_state && d_state && r_state && !(x_nhwc & CDN_X_ACT_PERCENTILE)), CDN_ERR_ARG,
              "CDN_X_DEFER_RANGE needs running != 0 and the three QuantActs, and excludes CDN_X_ACT_PERCENTILE");
  const int gmode = (x_nhwc & CDN_X_GATHER_MASK) >> 8;      // per-call schedule choice (tests); no library state
  // the k-blocked copy of the weight codes behind the row-major ones (the streaming int8 pointwise kernel)
  CDN_REQUIRE(!(x_nhwc & CDN_X_WCODES_KB) || (w_pw_codes && cdn_codenet_wcodes_kb_columns(C, Co) != 0), CDN_ERR_ARG,
              "CDN_X_WCODES_KB: no k-blocked form for C = %lld, Co = %lld (cdn_codenet_wcodes_kb_columns)",
              (long long)C, (long long)Co);
  const signed char *w_kb = (x_nhwc & CDN_X_WCODES_KB) ? w_pw_codes + cdn_codenet_wcodes_kb_offset(C, Co) : nullptr;
  // --act-percentile: the three QuantActs follow the 0.1 % / 99.9 % order statistics of their input instead of its
  // extremes (only while the ranges are tracked)
  const bool pct = (x_nhwc & CDN_X_ACT_PERCENTILE) != 0 && running != 0;
  x_nhwc &= 1;
  CDN_REQUIRE(!pct || (s_state && d_state && r_state), CDN_ERR_ARG, "CDN_X_ACT_PERCENTILE needs the three QuantActs");
  CDN_REQUIRE(x_up == 0 || x_up == 1, CDN_ERR_ARG, "x_up must be 0 or 1");
  CDN_REQUIRE(!x_up || ((H & 1) == 0 && (W & 1) == 0), CDN_ERR_SHAPE,
              "x_up needs even H, W (got %lld x %lld)", (long long)H, (long long)W);
  CDN_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), CDN_ERR_ARG,
              "ep_scale / ep_shift must both be set or both be NULL");
  CDN_REQUIRE((s_state == nullptr) == (s_min == nullptr) && (s_state == nullptr) == (s_max == nullptr) &&
                  (d_state == nullptr) == (d_min == nullptr) && (d_state == nullptr) == (d_max == nullptr) &&
                  (r_state == nullptr) == (r_min == nullptr) && (r_state == nullptr) == (r_max == nullptr),
              CDN_ERR_ARG, "each QuantAct needs x_min, x_max and state together");
  CDN_REQUIRE(!x_nhwc || (C & 3) == 0, CDN_ERR_UNSUPPORTED,
              "channels-last input needs C %% 4 == 0 (got %lld)", (long long)C);
  CDN_REQUIRE(x_up == 0 || x_nhwc, CDN_ERR_UNSUPPORTED, "an up-sampled input must be channels-last");
  CDN_REQUIRE(N <= 65535 && N * C * H * W < (1ll << 31) && N * Co * H * W < (1ll << 31),
              CDN_ERR_UNSUPPORTED, "shape too large");
  CDN_REQUIRE(workspace_bytes >= cdn_codenet_stage_workspace_bytes(N, C, H, W, x_up),
              CDN_ERR_WORKSPACE, "workspace too small");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(r_out) & 15) == 0,
              CDN_ERR_ARG, "x / r_out must be 16-byte and workspace 256-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  const int Hl = (int)(H >> x_up), Wl = (int)(W >> x_up);
  const int64_t HWl = (int64_t)Hl * Wl;
  auto r256 = [](int64_t b) { return (b + 255) / 256 * 256; };
  char *wsp = static_cast<char *>(workspace);
  float *s_raw = reinterpret_cast<float *>(wsp);
  float *d = reinterpret_cast<float *>(wsp + r256(N * HWl * 4));
  // Rows of d: C floats, or -- NCHW input with a ragged channel count feeding the int8 pointwise (CoDeNet2x stage 0,
  // C = 2153; round 4) -- padded to the next multiple of 64 so that the persistent LDS-DMA gather (whole 64-channel
  // chunks, 16-byte stores) and the int8 pointwise's 16-byte row loads apply; the pad channels duplicate channel
  // C - 1 and meet zero weight codes (see dw0p_kernel).  Same values as the unpadded schedule, bit for bit.
  const int64_t Cd = (C + 63) / 64 * 64;
  const bool pad_d = !pct && stage_pads_d(N, C, H, W, x_nhwc, x_up, gmode,
                                          w_pw_codes != nullptr && d_state != nullptr && ep_scale == nullptr &&
                                              w_pw_scale != nullptr && w_pw_colsum != nullptr);
  const int64_t ldd = pad_d ? Cd : C;
  char *pct_ws = wsp + r256(N * HWl * 4) + r256(N * H * W * Cd * 4);
  float *pct_out = reinterpret_cast<float *>(pct_ws + kPctBytes - 256);
  float2 *part_s = reinterpret_cast<float2 *>(pct_ws + r256(kPctBytes));
  float2 *part_d = part_s + kMaxPartials;
  float2 *part_r = part_d + kMaxPartials;
  // arrival counters: the LAST bytes of the workspace (the caller zeroes them once)
  unsigned *arrive = reinterpret_cast<unsigned *>(wsp + workspace_bytes / 256 * 256 -
                                                  3 * r256(cdn::kArriveWords * 4));
  const int arr_stride = (int)(r256(cdn::kArriveWords * 4) / 4);
  unsigned *sst = static_cast<unsigned *>(s_state), *dst = static_cast<unsigned *>(d_state),
           *rst = static_cast<unsigned *>(r_state);
  const unsigned *xq = static_cast<const unsigned *>(x_qstate);

  // 0: the stored plane does not fit LDS even in 8-channel chunks -> the global-memory gather (dwg_kernel)
  const int cch = (Hl <= 4096 && Wl <= 4096) ? cdn::stage_channel_chunk(Hl, Wl) : 0;
  CDN_REQUIRE(cch != 0 || xq == nullptr || x_nhwc, CDN_ERR_UNSUPPORTED, "quant-on-load needs a channels-last input");

  // Range tracking runs inside the producing kernels (last workgroup to finish), see
  // cdn::block_minmax_finish: no separate update launches.  Python evaluates (momentum - 1.) and
  // (1. - momentum) in double, then the tensor op rounds the scalar to fp32 (quant_modules.py:217-219).
  const float mm1 = (float)(momentum - 1.0), omm = (float)(1.0 - momentum);
  // (percentile: the producers only measure -- running = 0 leaves the range alone and parks the extremes in the
  // state --, the commit follows each of them)
  const int prun = (pct || defer) ? 0 : running;
  const cdn::QUpdate qu_s{s_min, s_max, sst, arrive, mm1, omm, bits, prun};
  const cdn::QUpdate qu_d{d_min, d_max, dst, arrive + arr_stride, mm1, omm, bits, prun};
  const cdn::QUpdate qu_r{r_min, r_max, rst, arrive + 2 * arr_stride, mm1, omm, bits, prun};
  // t holds n / rep elements, each standing for `rep` equal elements of the tensor the reference ranks (rep = 4: the
  // scale plane of an up-sampled input is computed at stored resolution; nearest x2 replicates every value four times)
  auto commit_percentile = [&](const float *t, int64_t n, int rep, const cdn::QUpdate &qu) -> int {
    // the reference's ranks: round(n * 0.1 * 0.01), round(n * 99.9 * 0.01), Python's round (half to even)
    int64_t k_lo = (int64_t)std::nearbyint(((double)n * 0.1) * 0.01);
    int64_t k_hi = (int64_t)std::nearbyint(((double)n * 99.9) * 0.01);
    CDN_REQUIRE(k_lo >= 1 && k_hi <= n, CDN_ERR_UNSUPPORTED,
                "kthvalue(): selected number k out of range (a tensor of %lld elements has no 0.1 %% order statistic)",
                (long long)n);
    k_lo = (k_lo + rep - 1) / rep;             // the k-th smallest of the replicated multiset
    k_hi = (k_hi + rep - 1) / rep;
    n /= rep;
    int rc2 = cdn_kth_values(t, n, k_lo, k_hi, pct_out, pct_out + 1, pct_ws, (size_t)(kPctBytes - 256), stream);
    if (rc2) return rc2;
    cdn::QUpdate q2 = qu;
    q2.running = running;
    quantact_commit_percentile_kernel<<<1, 64, 0, st>>>(q2, pct_out);
    return cdn::check_launch("codenet fused percentile commit");
  };
  const int ptag = (int)(H > 0xffff ? 0xffff : H);
  int rc = 0;
  // 1. scale prediction at stored resolution (+ min/max of s)
  float2 *smm = sst ? part_s : nullptr;
  int n_part_s = 0;
  if (parts_in) {
    CDN_REQUIRE(n_parts_in >= 1 && !s_state && !xq && x_nhwc && x_up && !pct && !defer &&
                    cdn::stage_channel_chunk(Hl, Wl) != 0,
                CDN_ERR_UNSUPPORTED, "scale partial sums feed the LDS gather of an up-sampled channels-last fp32 input only");
  } else if (phases & CDN_X_PHASE_SCALE) {
  {
  cdn::ProfScope ps(cdn::kProfScale, ptag, st);
  // tiled kernel for large planes (measured: 18 vs 22 us at 65536 pixels x 128 channels; the
  // wave-per-pixel kernel is ahead at 16384 x 256: 13.7 vs 15.0 us)
  if (x_nhwc && C <= 256 && N * HWl >= 32768 &&
      cdn::ceil_div(N * HWl, kScaleTilePix) <= kMaxPartials) {
    const long npix = (long)(N * HWl);
    const int blocks = (int)cdn::ceil_div(npix, kScaleTilePix);
    n_part_s = blocks;
    const int CQ = (int)C >> 2, LD = ((CQ + 31) & ~31) + 4;
    const size_t lds = ((size_t)kScaleTilePix * LD + 16) * sizeof(float);
    if (xq)
      scale_nhwc_tile_kernel<true><<<blocks, 256, lds, st>>>(x, xq, w_scale, b_scale, s_raw, smm,
                                                             qu_s, (int)C, npix, lo, hi);
    else
      scale_nhwc_tile_kernel<false><<<blocks, 256, lds, st>>>(x, nullptr, w_scale, b_scale, s_raw,
                                                              smm, qu_s, (int)C, npix, lo, hi);
  } else if (x_nhwc) {
    const long npix = (long)(N * HWl);
    const int blocks = (int)std::min<long>(cdn::ceil_div(npix, 4), (long)cdn::kCUs * 8);
    n_part_s = blocks;
    if (xq)
      scale_nhwc_kernel<true><<<blocks, 256, 0, st>>>(x, xq, w_scale, b_scale, s_raw, smm, qu_s,
                                                      (int)C, npix, lo, hi);
    else
      scale_nhwc_kernel<false><<<blocks, 256, 0, st>>>(x, nullptr, w_scale, b_scale, s_raw, smm,
                                                       qu_s, (int)C, npix, lo, hi);
  } else {
    CDN_REQUIRE(xq == nullptr, CDN_ERR_UNSUPPORTED, "quant-on-load needs a channels-last input");
    dim3 grid((unsigned)cdn::ceil_div(HWl, 64), (unsigned)N);
    n_part_s = (int)(grid.x * grid.y);
    CDN_REQUIRE(n_part_s <= kMaxPartials, CDN_ERR_UNSUPPORTED, "too many scale workgroups");
    scale_nchw_kernel<<<grid, kScaleWaves * 64, 0, st>>>(x, w_scale, b_scale, s_raw, smm, qu_s,
                                                          (int)C, (int)HWl, lo, hi);
  }
  }
  rc = cdn::check_launch("codenet fused scale");
  if (rc) return rc;
  (void)n_part_s;
  if (pct && (rc = commit_percentile(s_raw, N * H * W, x_up ? 4 : 1, qu_s))) return rc;
  }
  // 2. gather + depthwise (+ min/max of d)
  float2 *dmm = dst ? part_d : nullptr;   // always: the batch extremes also gate the int8 path
  if (!(phases & CDN_X_PHASE_GATHER)) {
  } else if (cch == 0) {
    cdn::ProfScope ps(cdn::kProfDw, ptag, st);
    rc = launch_dwg(x_nhwc != 0, x, xq, s_raw, sst, w_dw, d, dmm, qu_d, (int)N, (int)C, (int)H, (int)W, x_up, st);
  } else {
    // Few workgroups (cfg2: 32 images of 256 x 256 -- stage 1 had 128 and stage 2 64 workgroups of 64 channels on 256
    // CUs; in-kernel stamps, round 6): thinner chunks while the grid is below one workgroup per CU.  The gather is
    // per channel: the chunking changes no value.
    const int cch_g = cdn::thin_channel_chunk(cch, C, N);
    const int n_part_d = (int)(cdn::ceil_div(C, cch_g) * N);
    CDN_REQUIRE(n_part_d <= kMaxPartials, CDN_ERR_UNSUPPORTED, "too many gather workgroups");
    cdn::ProfScope ps(cdn::kProfDw, ptag, st);
    auto fn = cch_g == 64 ? launch_dw2<64> : cch_g == 32 ? launch_dw2<32> : cch_g == 16 ? launch_dw2<16> : launch_dw2<8>;
    rc = fn(x_nhwc != 0, x, xq, s_raw, sst, w_dw, d, dmm, qu_d, (int)N, (int)C, (int)H, (int)W, x_up, st, gmode,
            (int)ldd, cdn::ScaleFromSums{nullptr, nullptr, b_scale, lo, hi, parts_in, n_parts_in});
  }
  if (rc) return rc;
  if (pct && (phases & CDN_X_PHASE_GATHER) && (rc = commit_percentile(d, N * H * W * C, 1, qu_d))) return rc;
  if (!(phases & CDN_X_PHASE_POINTWISE)) return CDN_OK;
  // 3. pointwise MFMA (+ bias / affine / ReLU, min/max of the result)
  rc = launch_pointwise(d, dst, (long)(N * H * W), C, Co, w_pw, w_pw_codes, w_pw_scale, w_pw_colsum,
                        bias_pw, ep_scale, ep_shift, relu, r_out, rst ? part_r : nullptr, qu_r, ptag,
                        st, pad_d ? ldd : 0, 0, nullptr, nullptr, pad_d, w_kb, next_w_scale, parts_out);
  if (rc) return rc;
  if (pct) rc = commit_percentile(r_out, N * H * W * Co, 1, qu_r);
  return rc;
}

extern "C" int cdn_codenet_stage_fused_forward(
    const float *x, int x_nhwc, int x_up, const void *x_qstate, int64_t N, int64_t C, int64_t Co,
    int64_t H, int64_t W, const float *w_scale, const float *b_scale, float lo, float hi,
    const float *w_dw, const float *w_pw, const signed char *w_pw_codes, const float *w_pw_scale,
    const int *w_pw_colsum, const float *bias_pw, const float *ep_scale, const float *ep_shift,
    int relu, float *s_min, float *s_max, void *s_state, float *d_min,
    float *d_max, void *d_state, float *r_min, float *r_max, void *r_state, int bits,
    double momentum, int running, void *workspace, size_t workspace_bytes, float *r_out,
    void *stream) {
  return stage_fused_forward_impl(x, x_nhwc, x_up, x_qstate, N, C, Co, H, W, w_scale, b_scale, lo, hi, w_dw, w_pw,
                                  w_pw_codes, w_pw_scale, w_pw_colsum, bias_pw, ep_scale, ep_shift, relu, s_min, s_max,
                                  s_state, d_min, d_max, d_state, r_min, r_max, r_state, bits, momentum, running,
                                  workspace, workspace_bytes, r_out, stream, nullptr, 0, nullptr, nullptr);
}

// Chained fp32 stages (round 6): number of partial-sum planes the pointwise conv of a stage (N, C -> Co, H x W) leaves for
// the next stage's scale prediction -- 0 when this stage's pointwise is not the streaming f32 kernel (pws_kernel) or the
// next stage (Co channels at 2H x 2W, up-sampled channels-last input) has no LDS-resident gather.
extern "C" int cdn_codenet_stage_chain_parts(int64_t N, int64_t C, int64_t Co, int64_t H, int64_t W) {
  if (N <= 0 || C <= 0 || Co <= 0 || H <= 0 || W <= 0 || N > 65535) return 0;
  const long M = (long)(N * H * W);
  if (!pws_applies(M, C, Co, C, nullptr, nullptr)) return 0;
  if (H > 4096 || W > 4096 || cdn::stage_channel_chunk((int)H, (int)W) == 0) return 0;      // (the next stage's stored plane)
  const int tn = pws_choice(M, C, Co) / 10;
  return (int)cdn::ceil_div(Co, 32 * tn);
}

extern "C" int cdn_codenet_stage_fused_forward_chain(
    const float *x, int x_nhwc, int x_up, const void *x_qstate, int64_t N, int64_t C, int64_t Co,
    int64_t H, int64_t W, const float *w_scale, const float *b_scale, float lo, float hi,
    const float *w_dw, const float *w_pw, const float *bias_pw, const float *ep_scale, const float *ep_shift,
    int relu, void *workspace, size_t workspace_bytes, float *r_out, const float *parts_in, int n_parts_in,
    const float *next_w_scale, float *parts_out, void *stream) {
  CDN_REQUIRE((next_w_scale == nullptr) == (parts_out == nullptr), CDN_ERR_ARG,
              "next_w_scale and parts_out come together");
  CDN_REQUIRE(!parts_out || cdn_codenet_stage_chain_parts(N, C, Co, H, W) != 0, CDN_ERR_UNSUPPORTED,
              "no chained form for this stage (cdn_codenet_stage_chain_parts)");
  return stage_fused_forward_impl(x, x_nhwc, x_up, x_qstate, N, C, Co, H, W, w_scale, b_scale, lo, hi, w_dw, w_pw,
                                  nullptr, nullptr, nullptr, bias_pw, ep_scale, ep_shift, relu, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 8, 0.99, 0,
                                  workspace, workspace_bytes, r_out, stream, parts_in, n_parts_in, next_w_scale,
                                  parts_out);
}

extern "C" int cdn_codenet_unpack_nchw(const float *r_nhwc, const void *r_qstate, float *out_nchw,
                                       int64_t N, int64_t C, int64_t H, int64_t W, int up,
                                       void *stream) {
  CDN_REQUIRE(r_nhwc && out_nchw, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (up == 0 || up == 1), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE(N <= 65535 && H <= 65535 && (size_t)W * (C + 1) * 4 <= 150 * 1024, CDN_ERR_UNSUPPORTED,
              "row of %lld pixels x %lld channels does not fit the LDS transpose tile", (long long)W,
              (long long)C);
  hipStream_t st = cdn::as_stream(stream);
  const size_t lds = (size_t)W * (C + 1) * sizeof(float);
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void *)unpack_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)unpack_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  dim3 grid((unsigned)H, (unsigned)N);
  cdn::ProfScope ps(cdn::kProfUnpack, (int)(H > 0xffff ? 0xffff : H), st);
  if (r_qstate)
    unpack_kernel<true><<<grid, 256, lds, st>>>(r_nhwc, static_cast<const unsigned *>(r_qstate),
                                               out_nchw, (int)C, (int)H, (int)W, up);
  else
    unpack_kernel<false><<<grid, 256, lds, st>>>(r_nhwc, nullptr, out_nchw, (int)C, (int)H, (int)W, up);
  return cdn::check_launch("codenet unpack");
}

// ---- stand-alone entry points for the layers around the hot path (detection heads) ------------
extern "C" size_t cdn_codenet_aux_workspace_bytes(void) { return cdn::aux_workspace_bytes(); }


extern "C" int cdn_codenet_pointwise_nhwc_forward(
    const float *a, const void *a_qstate, int64_t M, int64_t C, int64_t Co, int64_t lda, int64_t ldo,
    const float *w,
    const signed char *w_codes, const float *w_scale, const int *w_colsum, const float *bias,
    const float *ep_scale, const float *ep_shift, int relu, float *r_min, float *r_max, void *r_state,
    int bits, double momentum, int running, void *workspace, size_t workspace_bytes, float *out,
    void *stream) {
  return cdn_codenet_pointwise_mixed_forward(a, a_qstate, nullptr, M, C, Co, lda, ldo, w, w_codes, w_scale,
                                             w_colsum, bias, ep_scale, ep_shift, relu, nullptr, r_min, r_max,
                                             r_state, bits, momentum, running, workspace, workspace_bytes,
                                             out, stream);
}

extern "C" int cdn_codenet_pointwise_mixed_forward(
    const float *a, const void *a_qstate, const unsigned char *a_gen, int64_t M, int64_t C, int64_t Co,
    int64_t lda, int64_t ldo, const float *w, const signed char *w_codes, const float *w_scale,
    const int *w_colsum, const float *bias, const float *ep_scale, const float *ep_shift, int relu,
    const int *out_map, float *r_min, float *r_max, void *r_state, int bits, double momentum, int running,
    void *workspace, size_t workspace_bytes, float *out, void *stream) {
  return cdn_codenet_pointwise_mixed_forward_n(a, a_qstate, a_gen, 0, M, C, Co, lda, ldo, w, w_codes, w_scale, w_colsum,
                                               bias, ep_scale, ep_shift, relu, out_map, r_min, r_max, r_state, bits,
                                               momentum, running, workspace, workspace_bytes, out, stream);
}

extern "C" int cdn_codenet_pointwise_mixed_forward_n(
    const float *a, const void *a_qstate, const unsigned char *a_gen, int n_gens, int64_t M, int64_t C, int64_t Co,
    int64_t lda, int64_t ldo, const float *w, const signed char *w_codes, const float *w_scale,
    const int *w_colsum, const float *bias, const float *ep_scale, const float *ep_shift, int relu,
    const int *out_map, float *r_min, float *r_max, void *r_state, int bits, double momentum, int running,
    void *workspace, size_t workspace_bytes, float *out, void *stream) {
  CDN_REQUIRE(a && w, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(n_gens >= 0, CDN_ERR_ARG, "n_gens must be >= 0 (0: unknown)");
  // out == NULL: RANGE-ONLY pass (round 4) -- the kernel computes everything and stores nothing; only the int8 kernel
  // on one input state implements it
  CDN_REQUIRE(out || (r_state && a_qstate && !a_gen && w_codes && !ep_scale), CDN_ERR_ARG,
              "a range-only pass (out == NULL) needs the output QuantAct and the int8 form");
  CDN_REQUIRE(a_gen == nullptr || a_qstate != nullptr, CDN_ERR_ARG, "a_gen needs the states in a_qstate");
  CDN_REQUIRE(M > 0 && C > 0 && Co > 0 && M * std::max(C, Co) < (1ll << 31), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), CDN_ERR_ARG,
              "ep_scale / ep_shift must both be set or both be NULL");
  CDN_REQUIRE((r_state == nullptr) == (r_min == nullptr) && (r_state == nullptr) == (r_max == nullptr),
              CDN_ERR_ARG, "the output QuantAct needs x_min, x_max and state together");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(a) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0,
              CDN_ERR_ARG, "a / out must be 4-byte aligned");
  CDN_REQUIRE((lda == 0 || lda >= C) && (ldo == 0 || ldo >= Co), CDN_ERR_ARG,
              "row strides must be 0 (dense) or >= the channel counts");
  CDN_REQUIRE(M * std::max(lda, ldo) < (1ll << 31), CDN_ERR_UNSUPPORTED, "shape too large");
  cdn::AuxWs ws{nullptr, nullptr};
  if (r_state)
    CDN_REQUIRE(cdn::aux_workspace(workspace, workspace_bytes, &ws), CDN_ERR_WORKSPACE,
                "workspace missing, too small or not 256-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  const cdn::QUpdate qu{r_min, r_max, static_cast<unsigned *>(r_state), ws.arrive,
                        (float)(momentum - 1.0), (float)(1.0 - momentum), bits, running};
  return launch_pointwise(a, static_cast<unsigned *>(const_cast<void *>(a_qstate)), (long)M, C, Co, w,
                          w_codes, w_scale, w_colsum, bias, ep_scale, ep_shift, relu, out,
                          r_state ? ws.partials : nullptr, qu, 0, st, lda, ldo, a_gen, out_map, false, nullptr, nullptr,
                          nullptr, n_gens);
}

// The first 1x1 convs of NH detection heads (64 -> 64 each, one shared input) as ONE launch: pwi8h_kernel.
extern "C" int cdn_codenet_heads_pointwise_supported(int64_t M, int64_t C, int n_heads) {
  return (M > 0 && C >= 32 && (C & 31) == 0 && n_heads >= 2 && n_heads <= 4 && M * 64 * n_heads < (1ll << 31) &&
          cdn::ceil_div(M, 64) <= kMaxPartials) ? 1 : 0;
}

extern "C" int cdn_codenet_heads_pointwise_forward(
    const float *a, const void *a_qstate, int64_t M, int64_t C, int n_heads, const float *w,
    const signed char *w_codes, const float *w_scale, const int *w_colsum, const float *bias, int relu,
    float *const *r_min, float *const *r_max, void *const *r_state, int bits, double momentum, int running,
    void *const *workspaces, size_t workspace_bytes, const int *out_map, float *out, int64_t head_stride, void *stream) {
  CDN_REQUIRE(a && a_qstate && w && w_codes && w_scale && w_colsum && r_min && r_max && r_state && workspaces && out_map &&
                  out,
              CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(cdn_codenet_heads_pointwise_supported(M, C, n_heads), CDN_ERR_UNSUPPORTED,
              "heads launch: 2-4 heads of 64 columns, C %% 32 == 0 (cdn_codenet_heads_pointwise_supported)");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  CDN_REQUIRE(head_stride >= M * 64 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w_codes)) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(out) & 3) == 0,
              CDN_ERR_ARG, "head_stride >= M * 64; a / w_codes 16-byte aligned");
  QUpdateN qus;
  float2 *part = nullptr;
  for (int h = 0; h < 4; ++h) {
    const int j = h < n_heads ? h : 0;
    CDN_REQUIRE(r_min[j] && r_max[j] && r_state[j] && workspaces[j], CDN_ERR_ARG, "null pointer in head %d", j);
    cdn::AuxWs ws{nullptr, nullptr};
    CDN_REQUIRE(cdn::aux_workspace(workspaces[j], workspace_bytes, &ws), CDN_ERR_WORKSPACE,
                "workspace of head %d missing, too small or not 256-byte aligned", j);
    if (h == 0) part = ws.partials;
    qus.q[h] = cdn::QUpdate{r_min[j], r_max[j], static_cast<unsigned *>(r_state[j]), ws.arrive,
                            (float)(momentum - 1.0), (float)(1.0 - momentum), bits, running};
  }
  const int Cpad = (int)((C + 63) / 64 * 64);
  const unsigned grid = (unsigned)(cdn::ceil_div(M, 64) * n_heads);
  hipStream_t st = cdn::as_stream(stream);
  const unsigned *aq = static_cast<const unsigned *>(a_qstate);
#define CDN_PWH(NH_)                                                                                                \
  pwi8h_kernel<NH_><<<grid, 256, 0, st>>>(a, aq, w_codes, w_scale, w_colsum, w, bias, out, part, qus, (long)M, (int)C, \
                                          Cpad, relu, (int)C, out_map, (long)head_stride, 0L)
  if (n_heads == 2) CDN_PWH(2);
  else if (n_heads == 3) CDN_PWH(3);
  else CDN_PWH(4);
#undef CDN_PWH
  return cdn::check_launch("codenet heads pointwise");
}
