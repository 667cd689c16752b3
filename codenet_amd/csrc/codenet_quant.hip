// codenet_quant.hip -- activation fake-quantisation (QuantAct) for gfx950, entirely on device.
//
// Restates, without any host round trip, what the reference's QuantAct.forward does per call
// (portable_quantizer/quant_modules.py:202-225 + quantization_utils/quant_utils.py:33-75,172-200):
//   1. batch-global min / max of the activation tensor            (quant_modules.py:205-206)
//   2. range tracking: "+=" initialisation while x_min == x_max, else EMA with m = 0.99 (:211-219)
//   3. scale = (2^k - 1) / clamp(x_max - x_min, 1e-10); zp = round(scale*x_min) + 2^(k-1)
//                                                                  (quant_utils.py:60-75)
//   4. q = round(scale*x - zp) (NOT clamped); out = (q + zp) / scale  (quant_utils.py:33-52,193-200)
// Every fp32 operation is issued with explicit _rn intrinsics in the reference's order so the
// compiler cannot contract them into FMAs: identical fp32 inputs give identical integer codes.
#include "cdn_common.h"

#include <algorithm>

namespace {

using cdn::f2ord;
using cdn::ord2f;
using cdn::quant_code;

// state layout (device, 8 x 4 bytes):
//   [0] ~ordered-uint batch min, [1] ordered-uint batch max while a range pass is in flight (zero between calls)
//   [2] scale  [3] zero_point  [4] batch min (float)  [5] batch max (float)  [6] wide-code flag
//   [7] arrival ticket of the range pass (zero between calls)
constexpr int kStateWords = cdn::kQStateWords;

// RELU: the extremes of max(x, 0) (the block ReLU -> QuantAct of the training path reads the pre-ReLU tensor once);
// min / max commute with the monotone ReLU, so it is applied to the two reduced values
// The LAST workgroup to arrive (ticket in state word [7]) folds the batch extremes into the running range and derives
// (scale, zero-point) -- cdn::quantact_update_device, the expressions of the separate update kernel -- and leaves words
// [0], [1], [7] zero again for the next call: one launch instead of init + min/max + update (the QAT step has ~10
// QuantAct calls; each launch costs ~4.5 us inside the step's graph).  Words [0] / [1] hold ~ord(min) / ord(max) under
// atomic MAX, so a zero-initialised state is the identity.
template <bool RELU>
__global__ void __launch_bounds__(256)
minmax_kernel(const float *__restrict__ x, long n, cdn::QUpdate qu) {
  unsigned *state = qu.state;
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  const long n4 = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // 4 independent 16-byte loads in flight per thread (the grid is capped, see minmax_grid)
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    mn = fminf(fminf(fminf(mn, v0.x), fminf(v0.y, fminf(v0.z, v0.w))), fminf(fminf(v1.x, v1.y), fminf(v1.z, v1.w)));
    mn = fminf(fminf(fminf(mn, v2.x), fminf(v2.y, fminf(v2.z, v2.w))), fminf(fminf(v3.x, v3.y), fminf(v3.z, v3.w)));
    mx = fmaxf(fmaxf(fmaxf(mx, v0.x), fmaxf(v0.y, fmaxf(v0.z, v0.w))), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w)));
    mx = fmaxf(fmaxf(fmaxf(mx, v2.x), fmaxf(v2.y, fmaxf(v2.z, v2.w))), fmaxf(fmaxf(v3.x, v3.y), fmaxf(v3.z, v3.w)));
    // (v_cmp_u_f32 tests two values at once; the flag is a scalar mask)
    has_nan |= __builtin_isunordered(v0.x, v0.y) | __builtin_isunordered(v0.z, v0.w) | __builtin_isunordered(v1.x, v1.y) |
            __builtin_isunordered(v1.z, v1.w) | __builtin_isunordered(v2.x, v2.y) | __builtin_isunordered(v2.z, v2.w) |
            __builtin_isunordered(v3.x, v3.y) | __builtin_isunordered(v3.z, v3.w);
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
    mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    has_nan |= __builtin_isunordered(v.x, v.y) | __builtin_isunordered(v.z, v.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    mn = fminf(mn, x[i]);
    mx = fmaxf(mx, x[i]);
    has_nan |= (x[i] != x[i]);
  }
  if (RELU) {      // the extremes of max(x, 0); a NaN stays one (the flag), as torch.relu keeps it
    mn = fmaxf(mn, 0.0f);
    mx = fmaxf(mx, 0.0f);
  }
  // NaN-propagating from here on: ordered-uint keys, integer maxima (cdn_common.h)
  unsigned klo = cdn::key_lo(cdn::nan_lo(mn, has_nan)), khi = cdn::key_hi(cdn::nan_hi(mx, has_nan));
  cdn::wave_key_max(klo, khi);
  __shared__ unsigned smn[4], smx[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    smn[wave] = klo;
    smx[wave] = khi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    klo = max(max(smn[0], smn[1]), max(smn[2], smn[3]));
    khi = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
    // ordering without a fence (an agent-scope release writes L2 back: +5..10 us per launch, measured): the two
    // extremes are RETURNING atomics whose results the ticket waits for, so they have been applied at L2 (where all
    // three words live and are only ever touched by atomics) before the ticket is taken
    const unsigned r0 = __hip_atomic_fetch_max(&state[0], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned r1 = __hip_atomic_fetch_max(&state[1], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(r0), "v"(r1) : "memory");
    const unsigned t = __hip_atomic_fetch_add(&state[7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      unsigned k0 = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned k1 = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!RELU) cdn::empty_pair_is_nan(k0, k1);      // (an all-NaN tensor; with RELU the per-thread clamp already gave (inf, 0))
      const float bmin = cdn::unkey_lo(k0), bmax = cdn::unkey_hi(k1);
      cdn::quantact_update_device(qu, bmin, bmax, true);
      __hip_atomic_store(&state[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One thread: range tracking + quantisation parameters (steps 2 and 3 above).
// running != 0: update x_min/x_max from the batch statistics (reference behaviour even in
// eval()); running == 0: frozen ranges, only derive scale / zp.
__global__ void __launch_bounds__(256)
quantact_update_kernel(float *x_min, float *x_max, unsigned *state, const float *ext_min,
                       const float *ext_max, const float2 *partials, int n_partials, int bits,
                       float m_minus_1, float one_minus_m, int running, int relu, unsigned *state_copy) {
  __shared__ float red[8];
  __shared__ float2 pr;
  const bool from_partials = !ext_min && partials;
  if (from_partials) {   // reduce the producers' per-workgroup {min,max}
    float mn = INFINITY, mx = -INFINITY;
    bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
    for (int i = threadIdx.x; i < n_partials; i += blockDim.x) cdn::fold_pair(partials[i], mn, mx, has_nan);
    cdn::block_minmax_store(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), &pr, red);
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const bool have_stats = ext_min || from_partials;
  float bmin = 0.f, bmax = 0.f;
  if (have_stats) {
    // (running without external extremes or partials is the fused range pass, minmax_kernel: never here)
    bmin = ext_min ? ext_min[0] : (from_partials ? pr.x : 0.0f);
    bmax = ext_max ? ext_max[0] : (from_partials ? pr.y : 0.0f);
    if (from_partials && bmin == INFINITY && bmax == -INFINITY)      // every producer's pair is empty: an all-NaN tensor
      bmin = bmax = __uint_as_float(0x7fc00000u);
    if (relu) {      // the extremes of max(x, 0) from those of x (ReLU is monotone)
      bmin = (bmin != bmin) ? bmin : fmaxf(bmin, 0.0f);      // (a poisoned pair stays NaN)
      bmax = (bmax != bmax) ? bmax : fmaxf(bmax, 0.0f);
    }
  }
  cdn::QUpdate u{x_min, x_max, state, nullptr, m_minus_1, one_minus_m, bits, running};
  cdn::quantact_update_device(u, bmin, bmax, have_stats);
  if (state_copy) {      // a snapshot for consumers that outlive the next call of this QuantAct (the training backward)
    for (int i = 0; i < kStateWords; ++i) state_copy[i] = (i >= 2 && i <= 6) ? state[i] : 0u;
  }
}

// out = (q + zp) / scale; optionally also the integer codes (int16: codes are NOT clamped to
// int8 by the reference, quant_utils.py:193-200, so int8 alone could not hold them).
__global__ void __launch_bounds__(256)
fake_quant_kernel(const float *__restrict__ x, float *__restrict__ out, int16_t *__restrict__ codes,
                  long n, const unsigned *__restrict__ state) {
  const float scale = reinterpret_cast<const float *>(state)[2];
  const float zp = reinterpret_cast<const float *>(state)[3];
  const long n4 = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float4 *o4 = reinterpret_cast<float4 *>(out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    const float q0 = quant_code(v.x, scale, zp), q1 = quant_code(v.y, scale, zp);
    const float q2 = quant_code(v.z, scale, zp), q3 = quant_code(v.w, scale, zp);
    if (out) {
      float4 r;
      r.x = __fdiv_rn(__fadd_rn(q0, zp), scale);
      r.y = __fdiv_rn(__fadd_rn(q1, zp), scale);
      r.z = __fdiv_rn(__fadd_rn(q2, zp), scale);
      r.w = __fdiv_rn(__fadd_rn(q3, zp), scale);
      o4[i] = r;
    }
    if (codes) {
      short4 c;
      c.x = (short)fminf(fmaxf(q0, -32768.f), 32767.f);
      c.y = (short)fminf(fmaxf(q1, -32768.f), 32767.f);
      c.z = (short)fminf(fmaxf(q2, -32768.f), 32767.f);
      c.w = (short)fminf(fmaxf(q3, -32768.f), 32767.f);
      reinterpret_cast<short4 *>(codes)[i] = c;
    }
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float q = quant_code(x[i], scale, zp);
    if (out) out[i] = __fdiv_rn(__fadd_rn(q, zp), scale);
    if (codes) codes[i] = (int16_t)fminf(fmaxf(q, -32768.f), 32767.f);
  }
}

// Block `ReLU(inplace) -> QuantAct -> Upsample(x2, nearest)` that follows every deform stage
// (lib/models/networks/shufflenetv2_dcn.py:303-308; quantize_model.py:79-81), training path: one pass that reads the
// pre-ReLU tensor y [rows = planes*H][W] and writes the fake-quantised values of max(y, 0) to the 2x2 replicas
// out [rows*2][2W] -- instead of relu (read + write), fake_quant (read + write) and upsample (read + 4 writes).
// A thread handles two neighbouring inputs of a row: one 16-byte store per output row.
__global__ void __launch_bounds__(256)
relu_fq_up2_kernel(const float *__restrict__ y, float *__restrict__ out, long rows, int W,
                   const unsigned *__restrict__ state) {
  const float scale = reinterpret_cast<const float *>(state)[2];
  const float zp = reinterpret_cast<const float *>(state)[3];
  const int Wh = (W + 1) >> 1;
  const long total = rows * Wh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Wh;
    const int w0 = (int)(i - r * Wh) * 2;
    const bool two = w0 + 1 < W;
    // (ReLU that propagates NaN like torch's threshold, so that a diverged step is not hidden: ADVICE r3)
    const float ya = y[r * W + w0], yb = two ? y[r * W + w0 + 1] : 0.0f;
    const float a = ya > 0.0f ? ya : (ya != ya ? ya : 0.0f), b = yb > 0.0f ? yb : (yb != yb ? yb : 0.0f);
    const float qa = __fdiv_rn(__fadd_rn(quant_code(a, scale, zp), zp), scale);
    const float qb = __fdiv_rn(__fadd_rn(quant_code(b, scale, zp), zp), scale);
    float *o = out + (r * 2) * (2L * W) + 2 * w0;
    if (two && (W & 1) == 0) {
      const float4 v = make_float4(qa, qa, qb, qb);
      *reinterpret_cast<float4 *>(o) = v;
      *reinterpret_cast<float4 *>(o + 2L * W) = v;
    } else {
      o[0] = qa; o[1] = qa; o[2L * W] = qa; o[2L * W + 1] = qa;
      if (two) { o[2] = qb; o[3] = qb; o[2L * W + 2] = qb; o[2L * W + 3] = qb; }
    }
  }
}

// The same block WITHOUT the up-sampling (round 4: stages 1-2 of the QAT step read their input at stored resolution,
// cdn_codenet_dw_up2_*): out[i] = fake_quant(max(y[i], 0)), and its backward grad_y = grad_out where y > 0.
__global__ void __launch_bounds__(256)
relu_fq_kernel(const float *__restrict__ y, float *__restrict__ out, long n, const unsigned *__restrict__ state) {
  const float scale = reinterpret_cast<const float *>(state)[2];
  const float zp = reinterpret_cast<const float *>(state)[3];
  const long n4 = n >> 2;
  auto fq = [&](float v) {
    const float a = v > 0.0f ? v : (v != v ? v : 0.0f);         // ReLU that propagates NaN like torch's
    return __fdiv_rn(__fadd_rn(quant_code(a, scale, zp), zp), scale);
  };
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4 *>(y)[i];
    reinterpret_cast<float4 *>(out)[i] = make_float4(fq(v.x), fq(v.y), fq(v.z), fq(v.w));
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = fq(y[i]);
}
__global__ void __launch_bounds__(256)
relu_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y, float *__restrict__ gy, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4 *>(y)[i], a = reinterpret_cast<const float4 *>(g)[i];
    reinterpret_cast<float4 *>(gy)[i] = make_float4(v.x > 0.f ? a.x : 0.f, v.y > 0.f ? a.y : 0.f, v.z > 0.f ? a.z : 0.f,
                                                    v.w > 0.f ? a.w : 0.f);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    gy[i] = y[i] > 0.f ? g[i] : 0.f;
}

// backward of that block: grad_y[r][w] = (sum of the 2x2 replicas' gradients) * (y > 0)   (straight-through QuantAct,
// quant_utils.py:202-204; threshold backward of ReLU; nearest-upsample backward = sum over the replicas, added in
// row-major order like at::native::upsample_nearest2d_backward)
__global__ void __launch_bounds__(256)
up2_relu_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y, float *__restrict__ gy, long rows, int W) {
  const int Wh = (W + 1) >> 1;
  const long total = rows * Wh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Wh;
    const int w0 = (int)(i - r * Wh) * 2;
    const bool two = w0 + 1 < W;
    const float *p = g + (r * 2) * (2L * W) + 2 * w0;
    float sa, sb = 0.0f;
    if (two && (W & 1) == 0) {
      const float4 t = *reinterpret_cast<const float4 *>(p), u = *reinterpret_cast<const float4 *>(p + 2L * W);
      sa = ((t.x + t.y) + u.x) + u.y;
      sb = ((t.z + t.w) + u.z) + u.w;
    } else {
      sa = ((p[0] + p[1]) + p[2L * W]) + p[2L * W + 1];
      if (two) sb = ((p[2] + p[3]) + p[2L * W + 2]) + p[2L * W + 3];
    }
    gy[r * W + w0] = y[r * W + w0] > 0.0f ? sa : 0.0f;
    if (two) gy[r * W + w0 + 1] = y[r * W + w0 + 1] > 0.0f ? sb : 0.0f;
  }
}

// Every workgroup ends with two atomics on the SAME two words, and one contended word sustains only ~88
// atomics/us on MI355X: 2048 workgroups cost ~35 us per launch whatever the tensor size (measured in the QAT
// step, 9 launches).  512 workgroups (2 per CU, 4 loads in flight per thread) keep the tail at ~6 us.
inline int minmax_grid(long n) {
  long b = cdn::ceil_div(n >> 2 > 0 ? n >> 2 : 1, 256 * 4);
  const long cap = (long)cdn::kCUs * 2;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

inline int stream_grid(long n) {
  long b = cdn::ceil_div(n >> 2 > 0 ? n >> 2 : 1, 256);
  const long cap = (long)cdn::kCUs * 8;
  return (int)(b < cap ? b : cap);
}


// ------------------------------------------------------------------------------------------
// kth_values: the k_lo-th and k_hi-th smallest elements of a tensor -- torch.kthvalue on the flattened tensor, which is
// what --act-percentile / --wt-percentile compute per QuantAct call (quant_utils.py:18-30: the 0.1 % and 99.9 %
// order statistics as the activation range; a full sort per call in the reference).  Radix select on the
// order-preserving 32-bit keys (cdn::f2ord): three passes over the tensor (11 + 11 + 10 bits), each a histogram of the
// elements that still match a rank's prefix, each followed by a one-workgroup scan that picks the digit.  Exact: the
// result is an element of x (NaNs order last, as in torch; -0.0 orders below +0.0, numerically the same value).
// ------------------------------------------------------------------------------------------
constexpr int kKthBins = 2048, kKthThreads = 512;
struct KthState {            // device-side, in the workspace behind the two histograms
  unsigned prefix[2], mask[2], rank[2];
};

__global__ void __launch_bounds__(kKthThreads)
kth_hist_kernel(const float *__restrict__ x, long n, int shift, int width, const KthState *__restrict__ st,
                unsigned *__restrict__ hist) {
  __shared__ unsigned lh[2][kKthBins];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * kKthBins; i += kKthThreads) (&lh[0][0])[i] = 0u;
  const unsigned p0 = st->prefix[0], p1 = st->prefix[1], m0 = st->mask[0], m1 = st->mask[1];
  const bool same = p0 == p1 && m0 == m1;        // both ranks still in one group (always at the first level)
  const unsigned wm = (1u << width) - 1u;
  __syncthreads();
  auto tally = [&](float v) __attribute__((always_inline)) {
    const unsigned k = v != v ? 0xFFFFFFFFu : cdn::f2ord(v);      // every NaN orders last, whatever its sign bit (torch)
    if ((k & m0) == p0) atomicAdd(&lh[0][(k >> shift) & wm], 1u);
    else if (!same && (k & m1) == p1) atomicAdd(&lh[1][(k >> shift) & wm], 1u);
  };
  const long nq = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n >> 2 : 0;
  for (long q = (long)blockIdx.x * kKthThreads + tid; q < nq; q += (long)gridDim.x * kKthThreads) {
    const float4 v = reinterpret_cast<const float4 *>(x)[q];
    tally(v.x); tally(v.y); tally(v.z); tally(v.w);
  }
  for (long i = nq * 4 + (long)blockIdx.x * kKthThreads + tid; i < n; i += (long)gridDim.x * kKthThreads) tally(x[i]);
  __syncthreads();
  for (int i = tid; i < 2 * kKthBins; i += kKthThreads) {
    const unsigned c = (&lh[0][0])[i];
    if (c) atomicAdd(&hist[i], c);
  }
}

// One workgroup: for both ranks, the digit whose bin holds the rank-th smallest element of the group; extends the
// prefix, rebases the rank, zeroes the histograms for the next level; after the last level writes the values.
__global__ void __launch_bounds__(1024)
kth_pick_kernel(unsigned *__restrict__ hist, KthState *__restrict__ st, int shift, int width, int last,
                float *__restrict__ out_lo, float *__restrict__ out_hi) {
  __shared__ unsigned scan[1024];
  __shared__ unsigned s_digit[2], s_below[2];
  const int tid = threadIdx.x;
  const bool same = st->prefix[0] == st->prefix[1] && st->mask[0] == st->mask[1];
  for (int j = 0; j < 2; ++j) {
    const unsigned *h = hist + ((j == 1 && !same) ? kKthBins : 0);
    const unsigned need = st->rank[j];                       // 1-based rank inside the group
    const unsigned lo = h[2 * tid], hi = h[2 * tid + 1];
    scan[tid] = lo + hi;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // inclusive prefix sums over bin pairs
      const unsigned v = tid >= off ? scan[tid - off] : 0u;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const unsigned incl = scan[tid], excl = incl - (lo + hi);
    if (excl < need && need <= incl) {                       // exactly one pair crosses the rank
      if (need <= excl + lo) {
        s_digit[j] = 2 * tid;
        s_below[j] = excl;
      } else {
        s_digit[j] = 2 * tid + 1;
        s_below[j] = excl + lo;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < 2 * kKthBins; i += 1024) hist[i] = 0u;
  __syncthreads();
  if (tid < 2) {
    const unsigned wm = (1u << width) - 1u;
    const unsigned prefix = st->prefix[tid] | (s_digit[tid] << shift), mask = st->mask[tid] | (wm << shift);
    st->prefix[tid] = prefix;
    st->mask[tid] = mask;
    st->rank[tid] -= s_below[tid];
    if (last) *(tid == 0 ? out_lo : out_hi) = cdn::ord2f(prefix);
  }
}

__global__ void kth_init_kernel(unsigned *hist, KthState *st, unsigned k_lo, unsigned k_hi) {
  for (int i = threadIdx.x; i < 2 * kKthBins; i += blockDim.x) hist[i] = 0u;
  if (threadIdx.x < 2) {
    st->prefix[threadIdx.x] = 0u;
    st->mask[threadIdx.x] = 0u;
    st->rank[threadIdx.x] = threadIdx.x == 0 ? k_lo : k_hi;
  }
}

}  // namespace

namespace cdn {
void launch_quantact_update(float *x_min, float *x_max, unsigned *state, const float *ext_min,
                            const float *ext_max, const float2 *partials, int n_partials, int bits,
                            double momentum, int running, hipStream_t st, int relu, unsigned *state_copy) {
  // Python evaluates (momentum - 1.) and (1. - momentum) in double, then the tensor op rounds
  // the scalar to fp32 (quant_modules.py:217-219).
  const int threads = (!ext_min && partials) ? 256 : 64;
  quantact_update_kernel<<<1, threads, 0, st>>>(x_min, x_max, state, ext_min, ext_max, partials,
                                                n_partials, bits, (float)(momentum - 1.0),
                                                (float)(1.0 - momentum), running, relu, state_copy);
}
}  // namespace cdn

extern "C" size_t cdn_quantact_state_bytes(void) { return kStateWords * sizeof(unsigned); }

extern "C" int cdn_quantact_forward(const float *x, float *out, int16_t *codes, int64_t numel,
                                    float *x_min, float *x_max, void *state,
                                    const float *batch_min, const float *batch_max, int bits,
                                    double momentum, int running, void *stream) {
  CDN_REQUIRE(x && x_min && x_max && state, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(numel > 0, CDN_ERR_ARG, "empty tensor");
  CDN_REQUIRE((batch_min == nullptr) == (batch_max == nullptr), CDN_ERR_ARG,
              "batch_min and batch_max must both be set or both be NULL");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                  (!out || (reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                  (!codes || (reinterpret_cast<uintptr_t>(codes) & 7) == 0),
              CDN_ERR_ARG, "tensors must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  unsigned *stt = static_cast<unsigned *>(state);
  if (running && !batch_min) {     // range pass + update in one launch (the last workgroup updates)
    const cdn::QUpdate qu{x_min, x_max, stt, nullptr, (float)(momentum - 1.0), (float)(1.0 - momentum), bits, 1};
    minmax_kernel<false><<<minmax_grid(numel), 256, 0, st>>>(x, (long)numel, qu);
  } else {
    cdn::launch_quantact_update(x_min, x_max, stt, batch_min, batch_max, nullptr, 0, bits, momentum,
                                running, st);
  }
  if (out || codes)
    fake_quant_kernel<<<stream_grid(numel), 256, 0, st>>>(x, out, codes, (long)numel, stt);
  return cdn::check_launch("quantact forward");
}

// QuantAct.forward with the batch extremes given as per-workgroup {min, max} pairs of the PRODUCING kernel
// (cdn_codenet_{scale,dw,pointwise}_forward_range): update + fake-quantisation, no range pass over x.
extern "C" int cdn_quantact_forward_partials(const float *x, float *out, int64_t numel, float *x_min, float *x_max,
                                             void *state, const float *partials, int64_t n_partials, int bits,
                                             double momentum, int running, void *state_copy, void *stream) {
  CDN_REQUIRE(x_min && x_max && state && partials, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE((out == nullptr || (x != nullptr && numel > 0)) && n_partials > 0 && n_partials < (1ll << 31), CDN_ERR_ARG,
              "bad size");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(partials) & 7) == 0,
              CDN_ERR_ARG, "tensors must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  unsigned *stt = static_cast<unsigned *>(state);
  cdn::launch_quantact_update(x_min, x_max, stt, nullptr, nullptr, reinterpret_cast<const float2 *>(partials),
                              (int)n_partials, bits, momentum, running, st, 0, static_cast<unsigned *>(state_copy));
  if (out) fake_quant_kernel<<<stream_grid(numel), 256, 0, st>>>(x, out, nullptr, (long)numel, stt);
  return cdn::check_launch("quantact forward (partials)");
}

// ---- apply-only forms (round 6): the producer has already updated the QuantAct (cdn_codenet_*_forward_update) --------
extern "C" int cdn_quantact_apply(const float *x, float *out, int64_t numel, const void *state, void *stream) {
  CDN_REQUIRE(x && out && state && numel > 0 && numel < (1ll << 31), CDN_ERR_ARG, "null pointer or bad size");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, CDN_ERR_ARG,
              "tensors must be 16-byte aligned");
  fake_quant_kernel<<<stream_grid(numel), 256, 0, cdn::as_stream(stream)>>>(
      x, out, nullptr, (long)numel, static_cast<unsigned *>(const_cast<void *>(state)));
  return cdn::check_launch("quantact apply");
}

extern "C" int cdn_quantact_relu_apply(const float *y, float *out, int64_t planes, int64_t H, int64_t W, int up,
                                       const void *state, void *stream) {
  CDN_REQUIRE(y && out && state, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * H * W < (1ll << 31), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
              CDN_ERR_ARG, "tensors must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  unsigned *stt = static_cast<unsigned *>(const_cast<void *>(state));
  const long numel = (long)(planes * H * W), rows = (long)(planes * H);
  if (up)
    relu_fq_up2_kernel<<<(unsigned)std::min<long>(cdn::ceil_div(rows * cdn::ceil_div(W, 2), 256), (long)cdn::kCUs * 16),
                         256, 0, st>>>(y, out, rows, (int)W, stt);
  else
    relu_fq_kernel<<<stream_grid(numel), 256, 0, st>>>(y, out, numel, stt);
  return cdn::check_launch("quantact relu [up2] apply");
}

static int relu_up2_impl(const float *y, float *out, int64_t planes, int64_t H, int64_t W, float *x_min, float *x_max,
                         void *state, const float *partials, int64_t n_partials, int bits, double momentum, int running,
                         void *stream, int up = 1) {
  CDN_REQUIRE(y && out && x_min && x_max && state, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * H * W < (1ll << 31), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
              CDN_ERR_ARG, "tensors must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  unsigned *stt = static_cast<unsigned *>(state);
  const long numel = (long)(planes * H * W);
  if (running && partials) {     // extremes of y from its producer, clamped at zero: those of max(y, 0)
    cdn::launch_quantact_update(x_min, x_max, stt, nullptr, nullptr, reinterpret_cast<const float2 *>(partials),
                                (int)n_partials, bits, momentum, running, st, 1);
  } else if (running) {
    const cdn::QUpdate qu{x_min, x_max, stt, nullptr, (float)(momentum - 1.0), (float)(1.0 - momentum), bits, 1};
    minmax_kernel<true><<<minmax_grid(numel), 256, 0, st>>>(y, numel, qu);
  } else {
    cdn::launch_quantact_update(x_min, x_max, stt, nullptr, nullptr, nullptr, 0, bits, momentum, running, st);
  }
  const long rows = (long)(planes * H);
  if (up)
    relu_fq_up2_kernel<<<(unsigned)std::min<long>(cdn::ceil_div(rows * cdn::ceil_div(W, 2), 256), (long)cdn::kCUs * 16),
                         256, 0, st>>>(y, out, rows, (int)W, stt);
  else
    relu_fq_kernel<<<stream_grid(numel), 256, 0, st>>>(y, out, numel, stt);
  return cdn::check_launch("quantact relu [up2] forward");
}

extern "C" int cdn_quantact_relu_forward(const float *y, float *out, int64_t numel, float *x_min, float *x_max,
                                         void *state, const float *partials, int64_t n_partials, int bits,
                                         double momentum, int running, void *stream) {
  CDN_REQUIRE(partials == nullptr || (n_partials > 0 && n_partials < (1ll << 31)), CDN_ERR_ARG, "bad partials");
  return relu_up2_impl(y, out, numel, 1, 1, x_min, x_max, state, partials, n_partials, bits, momentum, running, stream,
                       0);
}

extern "C" int cdn_relu_backward(const float *grad_out, const float *y, float *grad_y, int64_t numel, void *stream) {
  CDN_REQUIRE(grad_out && y && grad_y && numel > 0 && numel < (1ll << 31), CDN_ERR_ARG, "null pointer / bad size");
  CDN_REQUIRE(((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(y) |
                reinterpret_cast<uintptr_t>(grad_y)) & 15) == 0, CDN_ERR_ARG, "tensors must be 16-byte aligned");
  relu_bwd_kernel<<<stream_grid(numel), 256, 0, cdn::as_stream(stream)>>>(grad_out, y, grad_y, (long)numel);
  return cdn::check_launch("relu backward");
}

extern "C" int cdn_quantact_relu_up2_forward(const float *y, float *out, int64_t planes, int64_t H, int64_t W,
                                             float *x_min, float *x_max, void *state, int bits, double momentum,
                                             int running, void *stream) {
  return relu_up2_impl(y, out, planes, H, W, x_min, x_max, state, nullptr, 0, bits, momentum, running, stream);
}

extern "C" int cdn_quantact_relu_up2_forward_partials(const float *y, float *out, int64_t planes, int64_t H, int64_t W,
                                                      float *x_min, float *x_max, void *state, const float *partials,
                                                      int64_t n_partials, int bits, double momentum, int running,
                                                      void *stream) {
  CDN_REQUIRE(partials && n_partials > 0 && n_partials < (1ll << 31), CDN_ERR_ARG, "bad partials");
  return relu_up2_impl(y, out, planes, H, W, x_min, x_max, state, partials, n_partials, bits, momentum, running,
                       stream);
}

extern "C" int cdn_up2_relu_backward(const float *grad_out, const float *y, float *grad_y, int64_t planes, int64_t H,
                                     int64_t W, void *stream) {
  CDN_REQUIRE(grad_out && y && grad_y, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(planes > 0 && H > 0 && W > 0 && planes * H * W < (1ll << 29), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(grad_out) & 15) == 0, CDN_ERR_ARG, "grad_out must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  const long rows = (long)(planes * H);
  up2_relu_bwd_kernel<<<(unsigned)std::min<long>(cdn::ceil_div(rows * cdn::ceil_div(W, 2), 256), (long)cdn::kCUs * 16),
                        256, 0, st>>>(grad_out, y, grad_y, rows, (int)W);
  return cdn::check_launch("up2 relu backward");
}


extern "C" size_t cdn_kth_values_workspace_bytes(void) { return 2 * kKthBins * sizeof(unsigned) + 256; }

extern "C" int cdn_kth_values(const float *x, int64_t numel, int64_t k_lo, int64_t k_hi, float *out_lo, float *out_hi,
                              void *workspace, size_t workspace_bytes, void *stream) {
  CDN_REQUIRE(x && out_lo && out_hi && workspace, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(numel > 0 && numel < (1ll << 32), CDN_ERR_UNSUPPORTED, "1 <= numel < 2^32");
  CDN_REQUIRE(k_lo >= 1 && k_lo <= numel && k_hi >= 1 && k_hi <= numel, CDN_ERR_ARG,
              "kthvalue(): selected number k out of range (k_lo %lld, k_hi %lld, numel %lld)", (long long)k_lo,
              (long long)k_hi, (long long)numel);
  CDN_REQUIRE(workspace_bytes >= cdn_kth_values_workspace_bytes() && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
              CDN_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  unsigned *hist = static_cast<unsigned *>(workspace);
  KthState *state = reinterpret_cast<KthState *>(hist + 2 * kKthBins);
  kth_init_kernel<<<1, 256, 0, st>>>(hist, state, (unsigned)k_lo, (unsigned)k_hi);
  const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>(cdn::ceil_div((long)numel, 4L * kKthThreads * 4), 4L * cdn::kCUs));
  const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
  for (int lvl = 0; lvl < 3; ++lvl) {
    kth_hist_kernel<<<blocks, kKthThreads, 0, st>>>(x, (long)numel, shifts[lvl], widths[lvl], state, hist);
    kth_pick_kernel<<<1, 1024, 0, st>>>(hist, state, shifts[lvl], widths[lvl], lvl == 2, out_lo, out_hi);
  }
  return cdn::check_launch("kth values");
}
