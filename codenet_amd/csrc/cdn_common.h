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
// 8 x 4-byte words, zero-initialised by the owner: [0] ~ordered-uint batch min and [1] ordered-uint batch max while
// a range pass (codenet_quant.hip::minmax_kernel) is in flight, zero between calls; [2] scale (f32), [3] zero-point
// (f32), [4] batch min (f32), [5] batch max (f32), [6] wide-code flag, [7] the range pass's arrival ticket.
constexpr int kQStateWords = 8;

// Range tracking + scale / zero-point derivation, see quantact_update_kernel.  Batch statistics
// come from (in this order of precedence) ext_min/ext_max device scalars or `partials`
// ([n_partials] {min,max} pairs written by producer workgroups, reduced here -- no atomics); without either only
// frozen ranges (running == 0) are meaningful.
void launch_quantact_update(float *x_min, float *x_max, unsigned *state, const float *ext_min,
                            const float *ext_max, const float2 *partials, int n_partials, int bits,
                            double momentum, int running, hipStream_t st, int relu = 0,
                            unsigned *state_copy = nullptr);

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
// Cross-lane fetch of a tap-geometry record WITHOUT the LDS pipeline: v_mov_b32 dpp row_newbcast:J
// copies the value held by lane J of each 16-lane row to every lane of that row (VALU only; the
// compiler folds single-use movs into the consuming instruction's DPP operand).  ds_bpermute costs
// ~8 LDS-pipeline cycles per instruction: with 18 (dw2) / 26 (dw2u) of them next to 25 ds_read_b128
// (4 cycles each) per step the gather was LDS-issue bound on the bpermutes (in-kernel stamps,
// tools/probes/probe_dw.hip).  The DPP control is an immediate, so the step index is dispatched
// through a (wave-uniform) switch; HALVES: the two 8-lane halves of a row fetch different lanes
// (J and 8 + J) with complementary bank masks.
template <int J, bool HALVES>
__device__ __forceinline__ int row_fetch(int v) {
  // (mov_dpp: no defined "old" value, so no zero-initialising v_mov per fetch)
  if (HALVES) {
    const int a = __builtin_amdgcn_mov_dpp(v, 0x150 + J, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(a, v, 0x150 + 8 + J, 0xf, 0xc, false);
  }
  return __builtin_amdgcn_mov_dpp(v, 0x150 + J, 0xf, 0xf, false);
}
template <bool HALVES, int NI, int NF>
__device__ __forceinline__ void fetch_record(int j, const int (&gi)[NI], const float (&gf)[NF],
                                             int (&oi)[NI], float (&of)[NF]) {
#define CDN_CASE(J)                                                                          \
  case J: {                                                                                  \
    _Pragma("unroll") for (int q = 0; q < NI; ++q) oi[q] = row_fetch<J, HALVES>(gi[q]);       \
    _Pragma("unroll") for (int q = 0; q < NF; ++q)                                           \
        of[q] = __int_as_float(row_fetch<J, HALVES>(__float_as_int(gf[q])));                 \
  } break;
  if (HALVES) {
    switch (j) { CDN_CASE(0) CDN_CASE(1) CDN_CASE(2) CDN_CASE(3) CDN_CASE(4) CDN_CASE(5) CDN_CASE(6) CDN_CASE(7) }
  } else {
    switch (j) {
      CDN_CASE(0) CDN_CASE(1) CDN_CASE(2) CDN_CASE(3) CDN_CASE(4) CDN_CASE(5) CDN_CASE(6) CDN_CASE(7)
      CDN_CASE(8) CDN_CASE(9) CDN_CASE(10) CDN_CASE(11) CDN_CASE(12) CDN_CASE(13) CDN_CASE(14) CDN_CASE(15)
    }
  }
#undef CDN_CASE
}
// lane -> owned item of a 64-item batch such that the record of (step j, group g) sits in the DPP row
// of the lanes that consume it: row lane j (16 lanes per item) or half-row lane j (8 lanes per item)
template <int LPP>
__device__ __forceinline__ int owner_item(int lane) {
  return LPP == 16 ? (lane & 15) * 4 + (lane >> 4)
         : LPP == 8 ? (lane & 7) * 8 + ((lane >> 4) * 2 + ((lane >> 3) & 1))
                    : (lane & 3) * 16 + (lane >> 2);      // 4 lanes per item: quad lane j (fetch_record_quad)
}
// the 4-lanes-per-item form: lane j of every quad to the quad's four lanes (quad_perm [j, j, j, j])
template <int NI, int NF>
__device__ __forceinline__ void fetch_record_quad(int j, const int (&gi)[NI], const float (&gf)[NF],
                                                  int (&oi)[NI], float (&of)[NF]) {
#define CDN_CASE(J)                                                                                                \
  case J: {                                                                                                        \
    _Pragma("unroll") for (int q = 0; q < NI; ++q) oi[q] = __builtin_amdgcn_mov_dpp(gi[q], J * 0x55, 0xf, 0xf, false); \
    _Pragma("unroll") for (int q = 0; q < NF; ++q)                                                                 \
        of[q] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(gf[q]), J * 0x55, 0xf, 0xf, false));        \
  } break;
  switch (j) { CDN_CASE(0) CDN_CASE(1) CDN_CASE(2) CDN_CASE(3) }
#undef CDN_CASE
}
// LDS-DMA (global_load_lds) as inline asm: see the note in front of dw0p_kernel (codenet_fused.hip) for why not the builtin.
__device__ __forceinline__ void glds16(const void *gbase, unsigned voff, unsigned lds_dst) {
  // gbase: wave-uniform 64-bit base (SGPR pair), voff: the lane's byte offset, lds_dst: wave-uniform LDS byte address
  unsigned keep;
  const unsigned long long b = (unsigned long long)gbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(bu), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
__device__ __forceinline__ void glds4(const void *gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned long long b = (unsigned long long)gbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(bu), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
// LDS byte address of a pointer into the workgroup's LDS, wave-uniform by construction (made provable for "s")
__device__ __forceinline__ unsigned lds_addr_uniform(const void *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p);
}

// rint(x) as a 64-bit integer for |x| < 2^51, three instructions (v_cvt_f64_f32, v_add_f64, one add on the high word):
// x + 1.5 * 2^52 in double rounds to the nearest integer, ties to even, and its bit pattern is that of the constant
// plus the integer.  __float2ll_rn compiles to ~13 VALU instructions (no 64-bit convert on gfx950); the gather
// backward converts 25 contributions per (pixel, channel) and was VALU-bound on them (profiles/r05/dwbwd_pmc.txt).
// Identical values to __float2ll_rn for finite x in range; NaN / Inf are the caller's business.
__device__ __forceinline__ unsigned long long fixed_rn(float x) {
  const double r = (double)x + 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(r) - 0x4338000000000000LL);
}
// |v| as an unsigned integer image whose order is that of |v| with NaN above Inf: an integer max over these images
// PROPAGATES a NaN (fmaxf drops it)
__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// ---- NaN-propagating range reductions.  The reference's batch statistics are x.min() / x.max()
// (quant_modules.py:203-219), which return NaN when the tensor holds one -- the tracked range is NaN from then on and a
// diverged QAT step is loud.  fminf / fmaxf (v_min_f32 / v_max_f32, v_min3 / v_max3) DROP a NaN, so: every producer keeps
// a per-thread flag next to its running extremes (`has_nan |= v != v`: a compare whose result lives in a scalar mask),
// poisons its pair before the workgroup reduction -- nan_lo / nan_hi: the canonical -NaN / +NaN -- and every reduction
// above a thread runs on the ORDERED-UINT KEYS (key_lo = ~f2ord(min), key_hi = f2ord(max); larger key = more extreme),
// where -NaN / +NaN are the largest keys of their slot: an integer max propagates them at the cost of the float one.
// ReLU / Hardtanh as torch computes them: a NaN stays a NaN (fmaxf / fminf would return the bound)
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.0f ? 0.0f : v; }
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float nan_lo(float mn, bool poisoned) { return poisoned ? __uint_as_float(0xffc00000u) : mn; }
__device__ __forceinline__ float nan_hi(float mx, bool poisoned) { return poisoned ? __uint_as_float(0x7fc00000u) : mx; }
__device__ __forceinline__ unsigned key_lo(float mn) { return ~f2ord(mn); }
__device__ __forceinline__ unsigned key_hi(float mx) { return f2ord(mx); }
__device__ __forceinline__ float unkey_lo(unsigned k) { return ord2f(~k); }
__device__ __forceinline__ float unkey_hi(unsigned k) { return ord2f(k); }
// the two keys over the 64 lanes of a wave
__device__ __forceinline__ void wave_key_max(unsigned &klo, unsigned &khi) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    klo = max(klo, (unsigned)__shfl_xor((int)klo, m, 64));
    khi = max(khi, (unsigned)__shfl_xor((int)khi, m, 64));
  }
}
// A tensor that is NaN throughout -- the state of everything behind a poisoned range: scale and zero-point are NaN, so
// every fake-quantised value is -- never moves a producer's extremes: its reduction ends on the EMPTY pair (+inf, -inf),
// which no real tensor produces.  The final stage of every reduction reads it as NaN (free; this is what lets the
// VALU-bound kernels go without a per-element flag).
__device__ __forceinline__ void empty_pair_is_nan(unsigned &klo, unsigned &khi) {
  if (klo == key_lo(INFINITY) && khi == key_hi(-INFINITY)) {
    klo = key_lo(__uint_as_float(0xffc00000u));
    khi = key_hi(__uint_as_float(0x7fc00000u));
  }
}
// a {min, max} pair some producer stored (possibly poisoned) folded into running extremes + flag
__device__ __forceinline__ void fold_pair(float2 v, float &mn, float &mx, bool &has_nan) {
  mn = fminf(mn, v.x);
  mx = fmaxf(mx, v.y);
  has_nan |= (v.x != v.x) | (v.y != v.y);
}
// q = round(scale*x - zp) (half-even), no FMA contraction: quant_utils.py:33-41
// `#pragma clang fp contract(off)` + plain operators: the ocml _rn intrinsics do NOT keep the compiler from fusing a
// product into the following add / subtract (they bring their own fast-math flags into the caller).  Seen in the ISA:
// pwi8_kernel formed its codes as v_fma_f32(scale, x, -zp) -- ONE rounding where the reference has two
// (quant_utils.py:33-41) -- and the in-kernel range update's EMA came out 1 ulp off the reference arithmetic in ~5 %
// of the updates, depending on the inlining context.
__device__ __forceinline__ float quant_code(float x, float scale, float zp) {
#pragma clang fp contract(off)
  const float p = scale * x;
  return rintf(p - zp);
}
// (q + zp) / scale, true division: quant_utils.py:44-52
__device__ __forceinline__ float fake_quant(float x, float scale, float zp) {
#pragma clang fp contract(off)
  return __fdiv_rn(__fadd_rn(quant_code(x, scale, zp), zp), scale);
}
// The same value with r = RN(1 / scale) precomputed: Markstein's division q0 = n*r, q = fma(fma(-q0, s, n), r, q0)
// is the correctly rounded quotient n / s (the residual fma is exact), 3 instructions instead of the ~10
// of the IEEE expansion -- for loops that fake-quantise every loaded element (checked against true division
// for 3.6e6 (n, s) pairs on the host, tools note in DESIGN.md section 7.3).
__device__ __forceinline__ float fake_quant_r(float x, float scale, float zp, float r) {
#pragma clang fp contract(off)
  const float n = __fadd_rn(quant_code(x, scale, zp), zp);
  const float q0 = __fmul_rn(n, r);
  return fmaf(fmaf(-q0, scale, n), r, q0);
}
// ---- QuantAct range tracking + parameters, one thread (quant_modules.py:211-219,
// quant_utils.py:60-75); shared by the stand-alone update kernel and the in-kernel
// "last workgroup" update of the fused schedule.  have_stats: bmin/bmax are this batch's extremes.
struct QUpdate {
  float *x_min, *x_max;
  unsigned *state;       // kQStateWords words
  unsigned *counters;    // fused schedule only: kArriveWords zero-initialised arrival counters
  float m_minus_1, one_minus_m;
  int bits, running;
};
// Arrival counters: 64 group counters + 1 top counter, one per 64-byte line (a single contended
// word sustains only ~88 atomics/us; 2048 workgroups on one word cost ~25 us).
constexpr int kArriveGroups = 64;
constexpr int kArriveWords = (kArriveGroups + 1) * 16;

// have_wide (--act-percentile): bmin / bmax are the 0.1 % / 99.9 % order statistics the RANGE follows, wmin / wmax
// the batch's true extremes, which still decide how wide the codes get (the reference does not clamp them).
__device__ __forceinline__ void quantact_update_device(const QUpdate &u, float bmin, float bmax,
                                                       bool have_stats, bool preloaded = false,
                                                       float pre_lo = 0.f, float pre_hi = 0.f, bool have_wide = false,
                                                       float wmin = 0.f, float wmax = 0.f) {
#pragma clang fp contract(off)
  float lo = preloaded ? pre_lo : u.x_min[0], hi = preloaded ? pre_hi : u.x_max[0];
  float *sf = reinterpret_cast<float *>(u.state);
  if (have_stats) {
    sf[4] = have_wide ? wmin : bmin;
    sf[5] = have_wide ? wmax : bmax;
  }
  if (u.running) {
    if (lo == hi) {  // "Initialization" branch: += (quant_modules.py:211-213)
      lo = lo + bmin;
      hi = hi + bmax;
    } else {  // x += (m-1)*x + (1-m)*xb  (:217-219): every product and sum rounded on its own
      const float pl = u.m_minus_1 * lo, ql = u.one_minus_m * bmin, ph = u.m_minus_1 * hi, qh = u.one_minus_m * bmax;
      const float sl = pl + ql, sh = ph + qh;
      lo = lo + sl;
      hi = hi + sh;
    }
    u.x_min[0] = lo;
    u.x_max[0] = hi;
  }
  const float nlev = (float)((1 << u.bits) - 1);
  const float range = fmaxf(hi - lo, 1e-10f);          // torch.clamp(min=1e-10)
  // `n / tensor` in torch is Tensor.__rtruediv__ = tensor.reciprocal() * n: two roundings
  const float rcp = __fdiv_rn(1.0f, range);
  const float scale = rcp * nlev;
  const float sl0 = scale * lo;
  const float zp = rintf(sl0) + (float)(1 << (u.bits - 1));
  sf[2] = scale;
  sf[3] = zp;
  // state[6]: 1 when some level L - 128 = round(scale*x - zp) + zp - 128 of this batch cannot be
  // carried by the int8 kernels' nibble split (|.| > 2039) or the batch extremes are unknown --
  // consumers then take their f32 path.  Codes are monotone in x, so the extremes decide.
  unsigned wide = 1u;
  if (have_stats && u.bits == 8) {
    const float a0 = quant_code(have_wide ? wmin : bmin, scale, zp) + (zp - 128.0f);
    const float a1 = quant_code(have_wide ? wmax : bmax, scale, zp) + (zp - 128.0f);
    wide = (fabsf(a0) > 2039.0f || fabsf(a1) > 2039.0f || !(a0 == a0) || !(a1 == a1) ||
            !(fabsf(zp) < 4.0e6f)) ? 1u : 0u;   // (the int8 kernels do integer arithmetic on zp)
  }
  u.state[6] = wide;
}

// Barrier of the range epilogue.  __syncthreads() also waits for the wave's outstanding GLOBAL stores (s_waitcnt
// vmcnt(0)): at the end of a producer kernel that is the drain of its whole output tile, in front of the epilogue's
// atomic round trip.  An LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier) lets the two overlap -- built and
// measured in round 3: step 0.2503 vs 0.2508 ms, i.e. nothing (the kernel cannot end before its stores have drained
// anyway), so the plain form stays.
__device__ __forceinline__ void lds_barrier() {
#if defined(CDN_EPILOGUE_LDS_BARRIER) && CDN_EPILOGUE_LDS_BARRIER      // A/B build (measured: no difference)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

// Epilogue of a producer kernel in the fused schedule: workgroup min/max -> the workgroup's GROUP LINE ->
// arrival ticket; the LAST workgroup to arrive reads the 64 group lines and runs the range update, so no
// separate update launch (~4.4 us each inside a graph) and no heavily contended atomics are needed.
// A group line is one 64-byte line {count, max of ~ord(min), max of ord(max)} (ordered-uint images, so both
// extremes are integer atomicMax starting from 0).  A workgroup issues atomicMax, atomicMax, fetch-add to
// ITS line back to back: the three go to the same L2 channel in issue order, so when a line's count is
// complete every member's extremes are in it -- ONE round trip per workgroup instead of store / wait /
// ticket (measured: the epilogue is the tail of every producing kernel, 9 per step).  Tickets are two-level
// (workgroup -> one of 64 group lines -> top counter) so no word sees more than max(64, nblocks/64)
// arrivals (a single contended word sustains only ~88 atomics/us).  Lines start at zero and the last
// arriver zeroes them again.  Cross-workgroup visibility: agent-scope atomics on both sides (sc1,
// bypassing the non-coherent L1s).  Every thread of the workgroup must call this; `red` is
// >= 2*nwaves + 2 floats of free LDS.  (`partials` is unused by this protocol.)
__device__ __forceinline__ void block_minmax_finish(float mn, float mx, float2 *partials, int bid,
                                                    int nblocks, const QUpdate &u, float *red) {
  (void)partials;
  // (keys: see "NaN-propagating range reductions" above; mn / mx may be the poisoned pair of nan_lo / nan_hi)
  unsigned klo = key_lo(mn), khi = key_hi(mx);
  wave_key_max(klo, khi);
  unsigned *redu = reinterpret_cast<unsigned *>(red);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  lds_barrier();      // `red` may alias tiles other waves are still reading
  if ((threadIdx.x & 63) == 0) {
    redu[2 * wave] = klo;
    redu[2 * wave + 1] = khi;
  }
  lds_barrier();
  float pre_lo = 0.f, pre_hi = 0.f;
  if (threadIdx.x == 0) {
    pre_lo = u.x_min[0];      // in flight during the ticket round trip (only the last arriver uses them)
    pre_hi = u.x_max[0];
    for (int i = 1; i < nw; ++i) {
      klo = max(klo, redu[2 * i]);
      khi = max(khi, redu[2 * i + 1]);
    }
    const int ngroups = nblocks < kArriveGroups ? nblocks : kArriveGroups;
    const int g = bid % kArriveGroups;
    const unsigned gsize = (unsigned)((nblocks - g + kArriveGroups - 1) / kArriveGroups);
    unsigned *line = u.counters + 16 * g;
    bool last = false;
    (void)__hip_atomic_fetch_max(line + 1, klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_max(line + 2, khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned t1 = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t1 == gsize - 1) {
      const unsigned t2 = __hip_atomic_fetch_add(&u.counters[16 * kArriveGroups], 1u,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t2 == (unsigned)(ngroups - 1));
    }
    red[2 * nw] = last ? 1.0f : 0.0f;
  }
  lds_barrier();
  if (red[2 * nw] == 0.0f) return;
  // ---- last workgroup: wave 0 reads the group lines, updates ranges and parameters ---------------
  if (threadIdx.x < 64) {
    const int ngroups = nblocks < kArriveGroups ? nblocks : kArriveGroups;
    klo = key_lo(INFINITY);
    khi = key_hi(-INFINITY);
    if ((int)threadIdx.x < ngroups) {
      unsigned *line = u.counters + 16 * threadIdx.x;
      klo = __hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      khi = __hip_atomic_load(line + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_key_max(klo, khi);
    empty_pair_is_nan(klo, khi);
    if (threadIdx.x == 0) quantact_update_device(u, unkey_lo(klo), unkey_hi(khi), true, true, pre_lo, pre_hi);
    // everybody has arrived: leave the lines zero again
    __hip_atomic_store(&u.counters[16 * threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&u.counters[16 * threadIdx.x + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&u.counters[16 * threadIdx.x + 2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 64)
    __hip_atomic_store(&u.counters[16 * kArriveGroups], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// After block_minmax_finish in a kernel of the training path (round 6): the LAST workgroup (its flag is still in
// red[2 * nwaves]) leaves the snapshot of the updated state a backward pass reads.  Thread 0 wrote the state itself.
__device__ __forceinline__ void last_block_state_copy(const QUpdate &u, unsigned *state_copy, const float *red) {
  const int nw = (blockDim.x + 63) >> 6;
  if (state_copy && threadIdx.x == 0 && red[2 * nw] != 0.0f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) state_copy[i] = u.state[i];
  }
}

// Workgroup-level min/max -> ONE {min,max} pair stored at out[0] (plain store, no atomics: a
// single contended word sustains only ~88 atomics/us on MI355X).  Every thread of the workgroup
// must call it; `red` is >= 2*nwaves floats of LDS that nobody else is using.
__device__ __forceinline__ void block_minmax_store(float mn, float mx, float2 *out, float *red) {
  unsigned klo = key_lo(mn), khi = key_hi(mx);       // (NaN-propagating: see above)
  wave_key_max(klo, khi);
  unsigned *redu = reinterpret_cast<unsigned *>(red);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) {
    redu[2 * wave] = klo;
    redu[2 * wave + 1] = khi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) {
      klo = max(klo, redu[2 * i]);
      khi = max(khi, redu[2 * i + 1]);
    }
    *out = make_float2(unkey_lo(klo), unkey_hi(khi));
  }
}
}  // namespace cdn
#endif

namespace cdn {
// per-kernel capacity of the {min,max} partial arrays; grids are clamped / checked against it
constexpr int kMaxPartials = 16384;
// codenet_fused.hip, shared with the frozen-range schedule (codenet_frozen.hip)
int stage_channel_chunk(int Hl, int Wl);
int thin_channel_chunk(int cch, int64_t C, int64_t N);
int launch_frozen_scale(const void *x, int x_kind, const unsigned *xq, const float *w_scale, const float *b_scale,
                        float *s_raw, int64_t N, int64_t C, int64_t HWl, float lo, float hi, hipStream_t st);
// Frozen schedule, chained stages: the producer's pointwise epilogue leaves sums[m] = sum_co qw_s[co] * level_r[m][co]
// (exact int32) and the consumer's gather forms s_raw = clamp(bias + sums / (sw * sc_r), lo, hi) itself (sc_r: scale of
// its input quantiser).  sums == nullptr: s_raw comes from the scale kernel as usual.
struct ScaleFromSums {
  const int *sums;
  const float *sw;      // device scalar: scale of the conv_scale weight codes (w = qw / sw)
  const float *bias;    // device scalar or nullptr
  float lo, hi;
  // fp32 schedule, chained stages (round 6): the producer's pointwise epilogue (pws_kernel) leaves nparts planes
  // fparts[part][N * HWl] -- per column tile the dot product of its output columns with the conv_scale weights -- and the
  // consumer's gather forms s = clamp(((p0 + p1) + ...) + bias, lo, hi) in plane order: no scale launch, one fixed order
  const float *fparts;
  int nparts;
};
int launch_frozen_dw(const void *x, int x_kind, const unsigned *xq, const float *s_raw, const unsigned *sq,
                     const float *wd, signed char *d8, unsigned *dstate, unsigned *oflow, int N, int C, int H, int W,
                     int up, hipStream_t st, ScaleFromSums si = ScaleFromSums{nullptr, nullptr, nullptr, 0.f, 0.f},
                     int gmode = 0);
// Workspace of the stand-alone layer entry points: partials first, arrival counters in the LAST bytes
// (zeroed once by the caller); size = cdn_codenet_aux_workspace_bytes().
struct AuxWs {
  float2 *partials;
  unsigned *arrive;
};
size_t aux_workspace_bytes();
bool aux_workspace(void *workspace, size_t bytes, AuxWs *w);
}  // namespace cdn

#define CDN_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return cdn::fail(code, __VA_ARGS__); \
  } while (0)
