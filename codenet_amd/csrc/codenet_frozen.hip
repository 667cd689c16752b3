// codenet_frozen.hip -- the stage schedule for FROZEN QuantAct ranges with byte codes in HBM.
//
// In the reference `running_stat` is a plain attribute of QuantAct (portable_quantizer/quant_modules.py:172,181);
// with running_stat = False the range update (:203-219) is skipped and every QuantAct is a fixed affine grid
// (quant_utils.py:60-75,193-200).  The batch-global min / max that makes each QuantAct a global dependency in
// the reference-faithful schedule (codenet_fused.hip) is gone, so:
//   * no range epilogues, no arrival counters;
//   * every quantised tensor crosses HBM as ONE byte per element (the code q = round(scale*x - zp), see Code8 in
//     codenet_fused.hip) instead of four: d (gather output) 4x smaller on both sides, the stage output r too;
//   * the pointwise conv reads its int8 MFMA operand as it lies in memory: no fp32 -> code conversion per tile
//     and no nibble split (stored codes are in [-128,127] by construction), half the MFMAs of pwi8_kernel.
// Same arithmetic as the fp32 schedule wherever a rounding happens (scale sums, bilinear taps, the pointwise
// epilogue fmaf((float)(isum + 128*colsum), 1/(sc*sw), bias), the two-rounding code expression), so as long as
// no code saturates the results are BIT-IDENTICAL to cdn_codenet_stage_fused_forward with running = 0
// (tests/test_gpu_frozen.py).  A saturated code (the reference does not clamp, a byte must) sets *overflow:
// the caller recomputes that batch on the fp32 schedule.
#include "cdn_common.h"

#include <algorithm>
#include <cstdlib>

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

struct Code8 {   // (same as codenet_fused.hip: the byte is the code q = round(scale*x - zp) itself)
  float qs, qz;
};
using BadMask = int;
__device__ __forceinline__ Code8 make_code8(const unsigned *state, BadMask &bad) {
  Code8 c;
  c.qs = reinterpret_cast<const float *>(state)[2];
  c.qz = reinterpret_cast<const float *>(state)[3];
  if (!(fabsf(c.qz) < 4.0e6f)) bad = 1;
  return c;
}
__device__ __forceinline__ int act_code8(float v, const Code8 &c, BadMask &bad) {
#pragma clang fp contract(off)
  const float y_p = c.qs * v;      // (plain operators under fp contract(off): two roundings, cdn_common.h)
  const float y = (y_p - c.qz) + 12582912.0f;
  const int a = (int)__float_as_uint(y) - 0x4B400000;      // rint(scale*v - zp)
  const int s = min(max(a, -128), 127);
  bad |= a ^ s;
  return s;
}

// ------------------------------------------------------------------------------------------------------
// frozen_params_kernel: (scale, zero-point) of up to 64 frozen QuantActs from their x_min / x_max buffers
// into state words [2], [3] (and [6] = 0) -- the expressions of cdn::quantact_update_device without the range update
// (quant_utils.py:60-75).  One launch per step for the whole schedule.
// ------------------------------------------------------------------------------------------------------
constexpr int kMaxFrozen = 64;      // (round 6: the whole serving network's QuantActs in the step's first launch)
struct FrozenList {
  const float *x_min[kMaxFrozen];
  const float *x_max[kMaxFrozen];
  unsigned *state[kMaxFrozen];
  int n, bits;
  int4 *zero;          // optional: zero_n16 16-byte words to clear (the chained schedule's integer scale sums)
  long zero_n16;
};
__global__ void __launch_bounds__(256) frozen_params_kernel(FrozenList f) {
#pragma clang fp contract(off)
  if (blockIdx.x > 0) {      // blocks 1.. clear the sums buffer (a separate memset node costs 4.7 us per buffer)
    for (long q = (long)(blockIdx.x - 1) * 256 + threadIdx.x; q < f.zero_n16; q += (long)(gridDim.x - 1) * 256)
      f.zero[q] = make_int4(0, 0, 0, 0);
    return;
  }
  const int i = threadIdx.x;
  if (i >= f.n) return;
  const float lo = f.x_min[i][0], hi = f.x_max[i][0];
  const float nlev = (float)((1 << f.bits) - 1);
  const float range = fmaxf(hi - lo, 1e-10f);
  const float rcp = __fdiv_rn(1.0f, range);
  const float scale = rcp * nlev;                                        // n / tensor = reciprocal * n in torch
  const float sl0 = scale * lo;
  const float zp = rintf(sl0) + (float)(1 << (f.bits - 1));
  float *sf = reinterpret_cast<float *>(f.state[i]);
  sf[2] = scale;
  sf[3] = zp;
  // word [6] ("this batch has levels too wide for the int8 kernels' nibble split", written by the running-range
  // epilogues from the batch extremes) is stale once the range is frozen: cleared -- a frozen consumer that meets such
  // a level (a value 8 ranges outside the frozen range) sets its overflow flag instead
  f.state[i][6] = 0u;
}

// ------------------------------------------------------------------------------------------------------
// pwq8_kernel: R8[m][co] = code_r( act( (sum_c q[m][c] * qw[co][c] + zp_d * colsum[co]) / (sc_d * sw[co]) + b[co] ) )
//   q   [M][C]      byte codes of the d quantiser, channels-last, as the gather wrote them; the sum in
//                   brackets is sum_c level * qw, the integer pwi8_kernel forms as sum (level - 128) * qw + 128 * colsum
//   qw  [Co][Cpad]  4-bit weight codes as int8, zero padded to a multiple of 64
// v_mfma_i32_32x32x32_i8: lane (r = l & 31, h = l >> 5) supplies the 16 k bytes [16h, 16h+16) of row r of a
// 32-byte k step for both operands; C/D as f32 (column = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)).
// Workgroup tile BM x BN, K tiles of 64 bytes: LDS rows of 64 + 16 bytes (a quarter-wave's 16 ds_read_b128 cover
// the 64 banks), the next tile's 16-byte global loads in flight behind the MFMAs of the current one.
// HBM-bound: A is read once (BN covers all Co up to 256), the weights stay in L2.
// (Round 3, measured and removed: a variant with the WHOLE K extent of both tiles in LDS -- every load of a workgroup
// issued before the first wait -- for the backbone's K <= 512: 67-101 KB of LDS leave 1-2 workgroups per CU and nothing
// to overlap a workgroup's load phase with; 0.97 ms against 0.73 ms over the 42 launches of the frozen network.  Nor
// do 128-byte k tiles (half the round trips per workgroup, twice the loads in flight): 0.751 against 0.735 ms.  At
// M = 16 k .. 262 k rows and K <= 464 these launches sit at 9-23 us whatever the tile pipeline does.  Round 6: the
// 64 x 64 tile asking for ALL its k tiles (K <= 256: at most four) before waiting for the first, in REGISTERS: 80 ->
// 140 VGPRs = three instead of eight workgroups per CU, serving network 1.42 -> 1.59 ms -- what hides a workgroup's
// round trips here is the other seven workgroups of its CU, not a deeper pipeline of its own.)
// ------------------------------------------------------------------------------------------------------
constexpr int kQK = 64, kQLD = kQK + 16;

// The epilogue shared by the int8 pointwise kernels on codes: acc[j][r] is the exact integer sum of row
// m0 + wm + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column n0 + wn + 32 j + (lane & 31).
// What the epilogue needs from memory -- the two quantisers' parameters and this lane's per-column constants.  LOADED
// BEFORE THE K LOOP (round 6): read inside the epilogue they were one more memory round trip per workgroup, behind
// the barriers of the k loop where the compiler cannot hoist them, in kernels whose workgroups live for two to four k
// tiles.  Branch-free (clamped column); dead columns are masked where the values are used.
template <int TN>
struct PwqConsts {
  float qs, qzf, rqs, rqz;
  float bsv[TN], wsc[TN];
  int wsm[TN], nq[TN], oc[TN];
};
template <int TN>
__device__ __forceinline__ PwqConsts<TN> pwq8_consts(const unsigned *__restrict__ aq, const float *__restrict__ wscale,
                                                     const int *__restrict__ wsum, const float *__restrict__ bias,
                                                     const unsigned *__restrict__ rq, int Co,
                                                     const signed char *__restrict__ nsc, const int *__restrict__ omap,
                                                     int n0, int wn, int lane) {
  PwqConsts<TN> k;
  k.qs = reinterpret_cast<const float *>(aq)[2];
  k.qzf = reinterpret_cast<const float *>(aq)[3];
  k.rqs = rq ? reinterpret_cast<const float *>(rq)[2] : 1.f;
  k.rqz = rq ? reinterpret_cast<const float *>(rq)[3] : 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cc = min(n0 + wn + j * 32 + (lane & 31), Co - 1);
    k.wsc[j] = wscale[cc];
    k.wsm[j] = wsum[cc];
    k.bsv[j] = bias ? bias[cc] : 0.f;
    k.nq[j] = nsc ? (int)nsc[cc] : 0;
    k.oc[j] = omap ? omap[cc] : cc;
  }
  return k;
}

template <int TN>
__device__ __forceinline__ void pwq8_epilogue(
    i32x16 (&acc)[TN], const PwqConsts<TN> &kc, signed char *__restrict__ R8, float *__restrict__ Rf,
    unsigned *__restrict__ oflow, long M, int Cpad, int Co, int relu,
    const signed char *__restrict__ nsc, int *__restrict__ sacc, int ldo, long m0, int n0,
    int wm, int wn, int lane) {
  const float qs = kc.qs;
  const float qzf = kc.qzf;
  // epilogue: the expressions of pwi8_kernel, then the output quantiser's code (or fp32 for a consumer
  // that wants pre-quantisation values)
  BadMask bad = 0;
  Code8 c8 = {1.f, 0.f};
  if (R8) {
    c8.qs = kc.rqs;
    c8.qz = kc.rqz;
    if (!(fabsf(c8.qz) < 4.0e6f)) bad = 1;      // (make_code8's check)
  }
  if (!(fabsf(qzf) < 4.0e6f)) bad = 1;
  const int qzi = (int)fminf(fmaxf(qzf, -4.0e6f), 4.0e6f);      // zero-point of the A codes: an integer
  // sum q*qw + zp*colsum is formed in int32: |sum q*qw| <= 128*8*Cpad and |zp*colsum| <= |zp|*8*Cpad, so a narrow
  // frozen range far from zero (large |zp|) can leave int32 long before |zp| reaches 4e6 -- flagged, never silent
  // (the fp32 schedule takes its wide-code f32 branch for such ranges)
  if (((long)abs(qzi) + 128) * 8 * (long)Cpad >= (1L << 31)) bad = 1;
  // Scale prediction of the NEXT stage folded in (frozen schedule only, nsc != NULL): that stage's conv_scale is
  // s_raw[m] = b + sum_co (qw_s[co] / sw) * ((q_r[m][co] + zp_r) / sc_r) -- a sum of integer products over a common
  // denominator, so I[m] = sum_co qw_s[co] * (q_r + zp_r) is accumulated HERE in exact int32 (per row over this wave's
  // columns, then one integer atomic per row: order-independent, reproducible) and the consumer's gather forms
  // s_raw = b + I / (sw * sc_r) with one rounding -- instead of re-reading r in a separate scale launch.
  int rowsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rowsum[r] = 0;
  const int zpr = (int)c8.qz;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn + j * 32 + (lane & 31);
    float bsv = 0.f, rinv = 0.f;
    int t128 = 0, nq = 0, oc = co;
    if (co < Co) {
      bsv = kc.bsv[j];
      rinv = __fdiv_rn(1.0f, __fmul_rn(qs, kc.wsc[j]));
      t128 = qzi * kc.wsm[j];
      nq = kc.nq[j];
      oc = kc.oc[j];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M && co < Co) {
        float v = fmaf((float)(acc[j][r] + t128), rinv, bsv);
        if (relu) v = cdn::relu_keep_nan(v);
        if (R8) {
          const int code = act_code8(v, c8, bad);
          R8[m * ldo + oc] = (signed char)code;
          rowsum[r] += nq * (code + zpr);
        } else {
          Rf[m * ldo + oc] = v;
        }
      }
    }
  }
  if (nsc && R8) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int v = rowsum[r];
#pragma unroll
      for (int msk = 16; msk > 0; msk >>= 1) v += __shfl_xor(v, msk, 64);      // over the 32 columns of this half-wave
      const long m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if ((lane & 31) == 0 && m < M) atomicAdd(&sacc[m], v);
    }
  }
  if (bad) atomicOr(oflow, 1u);
}

template <int BM, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BM == 64 && BN == 64) ? 8 : 1, 8)))
pwq8_kernel(const signed char *__restrict__ A, const unsigned *__restrict__ aq,
            const signed char *__restrict__ Wq, const float *__restrict__ wscale, const int *__restrict__ wsum,
            const float *__restrict__ bias, signed char *__restrict__ R8, float *__restrict__ Rf,
            const unsigned *__restrict__ rq, unsigned *__restrict__ oflow, long M, int C, int Cpad, int Co,
            int relu, const signed char *__restrict__ nsc, int *__restrict__ sacc, int lda, int ldo,
            const int *__restrict__ omap) {
  constexpr int WGM = BM / 32, WGN = 4 / WGM, TN = BN / (32 * WGN);
  constexpr int AI = BM * kQK / 16 / 256;       // 16-byte loads of A per thread per k tile (1 or 2)
  constexpr int BI = BN * kQK / 16 / 256;       // of the weights (1, 2 or 4)
  static_assert(AI >= 1 && BI >= 1 && TN >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * kQLD];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * kQLD];
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * 32, wn = (wave % WGN) * TN * 32;
  const bool a16 = (lda & 15) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;   // (rows 16-byte aligned)
  i32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = (i32x16){0};
  // staging: 4 threads per 64-byte row segment
  const int lr = tid >> 2, lk = (tid & 3) * 16;
  i32x4 ra[AI], rb[BI];
  const PwqConsts<TN> kc = pwq8_consts<TN>(aq, wscale, wsum, bias, R8 ? rq : nullptr, Co, nsc, omap, n0, wn, lane);
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long m = min(m0 + lr + 64 * i, M - 1);
      const int k = k0 + lk;
      if (a16 && k + 15 < lda) {       // (bytes beyond C inside the row are paired with zero weights)
        ra[i] = *reinterpret_cast<const i32x4 *>(A + m * lda + k);
      } else {            // ragged tail: bytes beyond C are paired with zero weights, any finite value will do
        i32x4 t = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + 4 * e + 3 < lda) t[e] = *reinterpret_cast<const int *>(A + m * lda + k + 4 * e);   // (lda % 4 == 0)
        ra[i] = t;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int co = min(n0 + lr + 64 * i, Co - 1);
      rb[i] = *reinterpret_cast<const i32x4 *>(Wq + (long)co * Cpad + k0 + lk);
    }
  };
  load(0);
  const int nk = (C + kQK - 1) / kQK;          // (Cpad >= 64 * nk)
  for (int t = 0; t < nk; ++t) {
    __syncthreads();                            // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<i32x4 *>(&As[(lr + 64 * i) * kQLD + lk]) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<i32x4 *>(&Bs[(lr + 64 * i) * kQLD + lk]) = rb[i];
    __syncthreads();
    if (t + 1 < nk) load((t + 1) * kQK);
    const int fo = (lane & 31) * kQLD + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < kQK / 32; ++ks) {
      const i32x4 a = *reinterpret_cast<const i32x4 *>(&As[wm * kQLD + fo + ks * 32]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const i32x4 b = *reinterpret_cast<const i32x4 *>(&Bs[(wn + j * 32) * kQLD + fo + ks * 32]);
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  pwq8_epilogue<TN>(acc, kc, R8, Rf, oflow, M, Cpad, Co, relu, nsc, sacc, ldo, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------
// expand8_kernel: byte codes [M][C] -> the fake-quantised fp32 values (q + zp) / scale, channels-last (for consumers
// that take fp32 + a quantiser state: fake-quantising level / scale again returns the same value).
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
expand8_kernel(const signed char *__restrict__ a, const unsigned *__restrict__ aq, float *__restrict__ out, long n4) {
  const float scale = reinterpret_cast<const float *>(aq)[2], zp = reinterpret_cast<const float *>(aq)[3];
  const float r = __fdiv_rn(1.0f, scale);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const unsigned u = reinterpret_cast<const unsigned *>(a)[i];
    float4 t;
    const float l0 = __fadd_rn((float)(int)(signed char)(u & 0xff), zp), l1 = __fadd_rn((float)(int)(signed char)((u >> 8) & 0xff), zp);
    const float l2 = __fadd_rn((float)(int)(signed char)((u >> 16) & 0xff), zp), l3 = __fadd_rn((float)(int)(signed char)(u >> 24), zp);
    float q0 = __fmul_rn(l0, r); t.x = fmaf(fmaf(-q0, scale, l0), r, q0);
    q0 = __fmul_rn(l1, r); t.y = fmaf(fmaf(-q0, scale, l1), r, q0);
    q0 = __fmul_rn(l2, r); t.z = fmaf(fmaf(-q0, scale, l2), r, q0);
    q0 = __fmul_rn(l3, r); t.w = fmaf(fmaf(-q0, scale, l3), r, q0);
    reinterpret_cast<float4 *>(out)[i] = t;
  }
}

template <int BM, int BN>
void launch_pwq8(const signed char *A, const unsigned *aq, const signed char *Wq, const float *ws, const int *wsum,
                 const float *bias, signed char *R8, float *Rf, const unsigned *rq, unsigned *oflow, long M, int C,
                 int Co, int relu, hipStream_t st, const signed char *nsc = nullptr, int *sacc = nullptr, int lda = 0,
                 int ldo = 0, const int *omap = nullptr) {
  dim3 g((unsigned)cdn::ceil_div(M, BM), (unsigned)cdn::ceil_div(Co, BN));
  pwq8_kernel<BM, BN><<<g, 256, 0, st>>>(A, aq, Wq, ws, wsum, bias, R8, Rf, rq, oflow, M, C, (C + 63) / 64 * 64, Co,
                                         relu, nsc, sacc, lda ? lda : C, ldo ? ldo : Co, omap);
}


// ------------------------------------------------------------------------------------------------------
// Byte-code kernels for the layers AROUND the hot path in the frozen serving mode (the ShuffleNetV2
// backbone, quant_modules.py:809-907 with every QuantAct at running_stat = False).  Same arithmetic as the fp32
// kernels of codenet_layers.hip on the values (q + zp) / scale -- the accumulation chains below are theirs -- with one
// byte per element in HBM on both sides.
//
// value of a stored code: (q + zp) / scale by Markstein's division, bit-identical to cdn::fake_quant_r of the
// pre-quantisation value that produced the code (codenet_fused.hip::unpack_code8)
__device__ __forceinline__ float code_value(int q, float scale, float zp, float r) {
  const float l = __fadd_rn((float)q, zp);
  const float q0 = __fmul_rn(l, r);
  return fmaf(fmaf(-q0, scale, l), r, q0);
}

// stemq8: stem_kernel<24> (dense 3x3 conv 3 -> 24 + folded BN + ReLU on the NCHW image, one lane per output pixel)
// writing the codes of its QuantAct, rows of ld_out bytes.
__global__ void __launch_bounds__(256)
stemq8_kernel(const float *__restrict__ img, const float *__restrict__ w, const float *__restrict__ bias,
              signed char *__restrict__ out8, const unsigned *__restrict__ rq, unsigned *__restrict__ oflow, int H, int W,
              int Ho, int Wo, int stride, int relu, int ld_out) {
  constexpr int CO = 24;
  const int n = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  BadMask bad = 0;
  const Code8 c8 = make_code8(rq, bad);
  if (p < (long)Ho * Wo) {
    const int oy = (int)(p / Wo), ox = (int)(p - (long)oy * Wo);
    float v[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int y = oy * stride + dy - 1, x = ox * stride + dx - 1;
          v[(c * 3 + dy) * 3 + dx] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                                         ? img[(((long)n * 3 + c) * H + y) * W + x] : 0.0f;
        }
    unsigned *op = reinterpret_cast<unsigned *>(out8 + ((long)n * Ho * Wo + p) * ld_out);
#pragma unroll
    for (int c4 = 0; c4 < CO; c4 += 4) {
      unsigned pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = c4 + e;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = fmaf(w[co * 27 + k], v[k], acc);
        acc += bias ? bias[co] : 0.0f;
        if (relu) acc = cdn::relu_keep_nan(acc);
        pk |= (unsigned)(act_code8(acc, c8, bad) & 0xff) << (8 * e);
      }
      op[c4 >> 2] = pk;
    }
  }
  if (bad) atomicOr(oflow, 1u);
}

// dwq8: depthwise 3x3 (stride 1 / 2, zero padding 1) + folded-BN bias [+ ReLU] on byte codes, channels-last rows of
// ld_in / ld_out bytes.  One thread = 4 channels x a strip of SW output columns x YS output rows, walked top to bottom
// with the three live input rows DECODED ONCE into registers (a rotating window, resolved at compile time by the full
// unroll): a code is expanded to its value ((q + zp) / scale, five operations) (SW + 2) / SW times per output instead of
// nine times, the 36 weights of the channel quad are loaded once per thread, and every load is a 4-byte quad of one
// pixel (a wave reads whole 64-byte row segments).  The accumulation chain is the fp32 kernels' (dws / dwx):
// acc = fmaf(w[dy][dx], x, acc) from zero in (dy, dx) order, then + bias, ReLU, the output code.
template <int STRIDE, int SW, int YS>
__global__ void __launch_bounds__(256)
dwq8_kernel(const signed char *__restrict__ a8, const unsigned *__restrict__ aq, const float *__restrict__ w,
            const float *__restrict__ bias, signed char *__restrict__ out8, const unsigned *__restrict__ rq,
            unsigned *__restrict__ oflow, int C, int ld_in, int ld_out, int Hs, int Ws, int Ho, int Wo, int relu,
            int strips, int ysegs, long total) {
  constexpr int NC = (SW - 1) * STRIDE + 3;          // input columns under a strip
  const float qs = reinterpret_cast<const float *>(aq)[2], qz = reinterpret_cast<const float *>(aq)[3];
  const float qr = __fdiv_rn(1.0f, qs);
  BadMask bad = 0;
  const Code8 c8 = make_code8(rq, bad);
  const int CQ = (C + 3) >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < total) {
    const unsigned iu = (unsigned)i;                    // (total < 2^31: checked by the caller)
    const int cq = (int)(iu % (unsigned)CQ);
    unsigned t = iu / (unsigned)CQ;
    const int sx = (int)(t % (unsigned)strips);
    t /= (unsigned)strips;
    const int ys = (int)(t % (unsigned)ysegs), n = (int)(t / (unsigned)ysegs);
    const int cb = cq * 4, ox0 = sx * SW, oy0 = ys * YS;
    float wk[9][4], bs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = min(cb + e, C - 1);
      const bool live = cb + e < C;
      const unsigned lm = live ? 0xffffffffu : 0u;
#pragma unroll
      for (int k = 0; k < 9; ++k) wk[k][e] = __uint_as_float(__float_as_uint(w[(long)c * 9 + k]) & lm);
      bs[e] = bias ? __uint_as_float(__float_as_uint(bias[c]) & lm) : 0.0f;
    }
    const signed char *ab = a8 + (long)n * Hs * Ws * ld_in + cb;
    signed char *ob = out8 + (long)n * Ho * Wo * ld_out + cb;
    const int xb = STRIDE * ox0 - 1, yb = STRIDE * oy0 - 1;      // input coordinates of window row / column 0
    float v[3][NC][4];
    // input row yb + j decoded into slot j % 3 (the conv's zero padding is the VALUE zero)
#define CDN_DWQ8_ROW(j)                                                                                   \
    {                                                                                                       \
      const int y = yb + (j);                                                                               \
      const bool yin = (unsigned)y < (unsigned)Hs;                                                          \
      const signed char *rp = ab + (long)min(max(y, 0), Hs - 1) * Ws * ld_in;                               \
      unsigned u[NC], msk[NC];                                                                              \
      /* clamped addresses, always valid: unconditional loads issued back to back, then branch-free decoding   \
         (a guarded load or a guarded decode becomes a branch with its own s_waitcnt: the loads serialise) */ \
      _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                      \
        const int x = xb + c;                                                                               \
        u[c] = *reinterpret_cast<const unsigned *>(rp + (long)min(max(x, 0), Ws - 1) * ld_in);              \
        msk[c] = (yin && (unsigned)x < (unsigned)Ws) ? 0xffffffffu : 0u;                                    \
      }                                                                                                     \
      _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
          const int q = (int)(signed char)((u[c] >> (8 * e)) & 0xff);                                       \
          v[(j) % 3][c][e] = __uint_as_float(__float_as_uint(code_value(q, qs, qz, qr)) & msk[c]);          \
        }                                                                                                   \
    }
#pragma unroll
    for (int j = 0; j < 3 - STRIDE; ++j) CDN_DWQ8_ROW(j)
#pragma unroll
    for (int r = 0; r < YS; ++r) {
      const int oy = oy0 + r;
      if (oy < Ho) {                                 // (uniform per thread; rows below the plane need no loads)
#pragma unroll
        for (int j = STRIDE * r + 3 - STRIDE; j < STRIDE * r + 3; ++j) CDN_DWQ8_ROW(j)
#pragma unroll
        for (int sw = 0; sw < SW; ++sw) {
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[e] = fmaf(wk[dy * 3 + dx][e], v[(STRIDE * r + dy) % 3][sw * STRIDE + dx][e], acc[e]);
          unsigned pk = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o = acc[e] + bs[e];
            if (relu) o = cdn::relu_keep_nan(o);
            pk |= (unsigned)(act_code8(o, c8, bad) & 0xff) << (8 * e);
          }
          if (ox0 + sw < Wo) *reinterpret_cast<unsigned *>(ob + ((long)oy * Wo + ox0 + sw) * ld_out) = pk;
        }
      }
    }
#undef CDN_DWQ8_ROW
  }
  if (bad) atomicOr(oflow, 1u);
}

// dwpwq8: a unit's depthwise 3x3 computed INTO the A tile of the 1x1 conv behind it (frozen serving mode; the
// depthwise output never reaches HBM and one launch per unit disappears -- the byte-code backbone is bound by its ~60
// launches of 9-23 us, not by bytes).  Workgroup = 64 consecutive output pixels of one image (TR = 64 / Wseg rows of
// Wseg = min(Wo, 64) columns) x ALL channels:
//   phase 1  dwq8's strip walk per (channel quad, strip of SW columns) item over the tile's TR rows -- same
//            accumulation chain, same output code -- with the packed code quad stored to the LDS A tile [64][KP + 16];
//   phase 2  pwq8's MFMA loop with the A operand resident and the weights streamed in 64-byte k tiles, BN = all Co
//            (<= 256: every column tile would recompute phase 1), pwq8's epilogue.
// Bit-identical to dwq8_kernel followed by pwq8_kernel (tests/test_gpu_frozen.py).
// (waves per SIMD asked of the register allocator: without the request it spends up to 312 VGPRs on the depthwise
// phase's unrolled rows -- one wave per SIMD at 256 columns, two at 128 -- where 153-245 / 107-168 do without scratch:
// serving network 1.402 -> see DESIGN 4.2; the two shapes that would spill keep two waves)
template <int STRIDE, int SW, int TR, int BN, int BM = 64>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu((BN == 256 || (STRIDE == 2 && TR == 4)) ? 2 : 3, 8)))
dwpwq8_kernel(const signed char *__restrict__ a8, const unsigned *__restrict__ aq, const float *__restrict__ wdw,
              const float *__restrict__ bdw, int dw_relu, const unsigned *__restrict__ dq,
              const signed char *__restrict__ Wq, const float *__restrict__ wscale, const int *__restrict__ wsum,
              const float *__restrict__ bias, signed char *__restrict__ R8, const unsigned *__restrict__ rq,
              unsigned *__restrict__ oflow, long M, int C, int KP, int ld_in, int Hs, int Ws, int Ho, int Wo, int Co,
              int relu, int ldo, const int *__restrict__ omap) {
  // (BM = 64: 2 x 2 waves; BM = 32, round 6: 1 x 4 waves -- the 16 x 16 planes of layer 4 at batch 64 are 256 tiles of 64
  // pixels, one four-wave workgroup per CU: 32-pixel tiles put two on every CU)
  constexpr int WGN = 4 / (BM / 32), TN = BN / (32 * WGN), BI = BN * kQK / 16 / 256, NC = (SW - 1) * STRIDE + 3;
  static_assert(BM == 64 || (BM == 32 && BN >= 128), "tile shapes: 64 rows, or 32 rows x >= 128 columns");
  constexpr int Wseg = BM / TR;                    // columns of the tile (== Wo, or a 64-column segment of a row)
  static_assert(TN >= 1 && BI >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dp[];
  const int QA = KP + 16;
  unsigned char *As = lds_dp, *Bs = lds_dp + (size_t)BM * QA;
  const long m0 = (long)blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WGN) * 32, wn = (wave % WGN) * TN * 32;
  // ---- the weights' first k tile goes out before the depthwise phase ------------------------------------------
  const int lr = tid >> 2, lk = (tid & 3) * 16;
  i32x4 rb[BI];
  auto loadB = [&](int k0) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int co = min(lr + 64 * i, Co - 1);
      rb[i] = *reinterpret_cast<const i32x4 *>(Wq + (long)co * KP + k0 + lk);
    }
  };
  loadB(0);
  // ---- phase 1: depthwise into the A tile ---------------------------------------------------------------------
  {
    const float qs = reinterpret_cast<const float *>(aq)[2], qz = reinterpret_cast<const float *>(aq)[3];
    const float qr = __fdiv_rn(1.0f, qs);
    BadMask bad = 0;
    const Code8 c8 = make_code8(dq, bad);
    const int CQ = (C + 3) >> 2;
    constexpr int strips = Wseg / SW;
    const long pix0 = m0 % ((long)Ho * Wo);          // first pixel of the tile inside its image
    const int n = (int)(m0 / ((long)Ho * Wo));
    const int oy0 = (int)(pix0 / Wo), oxs = (int)(pix0 - (long)oy0 * Wo);   // (oxs != 0 only when Wo > 64)
    for (int item = tid; item < CQ * strips; item += 256) {
      const int cq = item % CQ, sx = item / CQ;
      const int cb = cq * 4, ox0 = oxs + sx * SW;
      float wk[9][4], bs[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = min(cb + e, C - 1);
        const unsigned lm = cb + e < C ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k][e] = __uint_as_float(__float_as_uint(wdw[(long)c * 9 + k]) & lm);
        bs[e] = bdw ? __uint_as_float(__float_as_uint(bdw[c]) & lm) : 0.0f;
      }
      const signed char *ab = a8 + (long)n * Hs * Ws * ld_in + cb;
      const int xb = STRIDE * ox0 - 1, yb = STRIDE * oy0 - 1;
      float v[3][NC][4];
#define CDN_DWPW_ROW(j)                                                                                   \
      {                                                                                                     \
        const int y = yb + (j);                                                                             \
        const bool yin = (unsigned)y < (unsigned)Hs;                                                        \
        const signed char *rp = ab + (long)min(max(y, 0), Hs - 1) * Ws * ld_in;                             \
        unsigned u[NC], msk[NC];                                                                            \
        _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                    \
          const int x = xb + c;                                                                             \
          u[c] = *reinterpret_cast<const unsigned *>(rp + (long)min(max(x, 0), Ws - 1) * ld_in);            \
          msk[c] = (yin && (unsigned)x < (unsigned)Ws) ? 0xffffffffu : 0u;                                  \
        }                                                                                                   \
        _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                      \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
            const int q = (int)(signed char)((u[c] >> (8 * e)) & 0xff);                                     \
            v[(j) % 3][c][e] = __uint_as_float(__float_as_uint(code_value(q, qs, qz, qr)) & msk[c]);        \
          }                                                                                                 \
      }
#pragma unroll
      for (int j = 0; j < 3 - STRIDE; ++j) CDN_DWPW_ROW(j)
#pragma unroll
      for (int r = 0; r < TR; ++r) {
#pragma unroll
        for (int j = STRIDE * r + 3 - STRIDE; j < STRIDE * r + 3; ++j) CDN_DWPW_ROW(j)
#pragma unroll
        for (int sw = 0; sw < SW; ++sw) {
          float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[e] = fmaf(wk[dy * 3 + dx][e], v[(STRIDE * r + dy) % 3][sw * STRIDE + dx][e], acc[e]);
          unsigned pk = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o = acc[e] + bs[e];
            if (dw_relu) o = cdn::relu_keep_nan(o);
            pk |= (unsigned)(act_code8(o, c8, bad) & 0xff) << (8 * e);
          }
          *reinterpret_cast<unsigned *>(&As[(r * Wseg + sx * SW + sw) * QA + cb]) = pk;
        }
      }
#undef CDN_DWPW_ROW
    }
    if (bad) atomicOr(oflow, 1u);
  }
  // ---- phase 2: the 1x1 conv on the resident A tile -----------------------------------------------------------
  const PwqConsts<TN> kc = pwq8_consts<TN>(dq, wscale, wsum, bias, rq, Co, nullptr, omap, 0, wn, lane);
  i32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = (i32x16){0};
  const int nk = (C + kQK - 1) / kQK;
  constexpr int QB = kQK + 16;
  for (int t = 0; t < nk; ++t) {
    __syncthreads();                                // (t == 0: also the A tile is complete)
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<i32x4 *>(&Bs[(lr + 64 * i) * QB + lk]) = rb[i];
    __syncthreads();
    if (t + 1 < nk) loadB((t + 1) * kQK);
    const int foA = (lane & 31) * QA + (lane >> 5) * 16, foB = (lane & 31) * QB + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < kQK / 32; ++ks) {
      const i32x4 a = *reinterpret_cast<const i32x4 *>(&As[wm * QA + foA + t * kQK + ks * 32]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const i32x4 b = *reinterpret_cast<const i32x4 *>(&Bs[(wn + j * 32) * QB + foB + ks * 32]);
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  pwq8_epilogue<TN>(acc, kc, R8, nullptr, oflow, M, KP, Co, relu, nullptr, nullptr, ldo, m0, 0, wm, wn, lane);
}

// maxq8: MaxPool2d(3, stride 2, padding 1) of the "S2 + MaxPool" stems (shufflenetv2_dcn.py:209-214; README configs b and
// e) on byte codes: the pool follows ReLU + QuantAct, fake-quantisation is monotone, so the maximum of the codes is the
// code of the maximum -- exactly what maxpool_kernel<true> computes on values.  One thread = one output pixel x 4
// channels; window cells outside the image do not take part (clamped address, code -128 = identity of max).
__global__ void __launch_bounds__(256)
maxq8_kernel(const signed char *__restrict__ a8, signed char *__restrict__ out8, int C, int ld_in, int ld_out, int Hs,
             int Ws, int Ho, int Wo, long total) {
  const int CQ = (C + 3) >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const unsigned iu = (unsigned)i;
    const int cq = (int)(iu % (unsigned)CQ);
    unsigned t = iu / (unsigned)CQ;
    const int ox = (int)(t % (unsigned)Wo);
    t /= (unsigned)Wo;
    const int oy = (int)(t % (unsigned)Ho), n = (int)(t / (unsigned)Ho);
    const signed char *ab = a8 + (long)n * Hs * Ws * ld_in + cq * 4;
    int m[4] = {-128, -128, -128, -128};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int y = 2 * oy + dy - 1, x = 2 * ox + dx - 1;
        const bool in = (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        const unsigned u = *reinterpret_cast<const unsigned *>(ab + ((long)min(max(y, 0), Hs - 1) * Ws + min(max(x, 0), Ws - 1)) * ld_in);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = (int)(signed char)((u >> (8 * e)) & 0xff);
          m[e] = max(m[e], in ? q : -128);
        }
      }
    const unsigned pk = (unsigned)(m[0] & 0xff) | ((unsigned)(m[1] & 0xff) << 8) | ((unsigned)(m[2] & 0xff) << 16) |
                        ((unsigned)(m[3] & 0xff) << 24);
    *reinterpret_cast<unsigned *>(out8 + ((long)n * Ho * Wo + (long)oy * Wo + ox) * ld_out + cq * 4) = pk;
  }
}
}  // namespace

static int frozen_params_impl(int n, float *const *x_min, float *const *x_max, void *const *state, int bits,
                              void *zero, size_t zero_bytes, void *stream);

extern "C" int cdn_quantact_frozen_params(int n, float *const *x_min, float *const *x_max, void *const *state,
                                          int bits, void *stream) {
  return frozen_params_impl(n, x_min, x_max, state, bits, nullptr, 0, stream);
}

extern "C" int cdn_quantact_frozen_params_clear(int n, float *const *x_min, float *const *x_max, void *const *state,
                                                int bits, void *clear, size_t clear_bytes, void *stream) {
  CDN_REQUIRE(clear_bytes == 0 || (clear && (reinterpret_cast<uintptr_t>(clear) & 15) == 0 && (clear_bytes & 15) == 0),
              CDN_ERR_ARG, "the buffer to clear must be 16-byte aligned and a multiple of 16 bytes");
  return frozen_params_impl(n, x_min, x_max, state, bits, clear, clear_bytes, stream);
}

static int frozen_params_impl(int n, float *const *x_min, float *const *x_max, void *const *state, int bits,
                              void *zero, size_t zero_bytes, void *stream) {
  CDN_REQUIRE(n >= 0 && n <= kMaxFrozen, CDN_ERR_ARG, "at most %d QuantActs per call", kMaxFrozen);
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  if (n == 0 && zero_bytes == 0) return CDN_OK;
  CDN_REQUIRE(n == 0 || (x_min && x_max && state), CDN_ERR_ARG, "null pointer");
  FrozenList f;
  f.n = n;
  f.bits = bits;
  for (int i = 0; i < kMaxFrozen && n > 0; ++i) {
    const int j = i < n ? i : 0;
    CDN_REQUIRE(x_min[j] && x_max[j] && state[j], CDN_ERR_ARG, "null pointer in entry %d", j);
    f.x_min[i] = x_min[j];
    f.x_max[i] = x_max[j];
    f.state[i] = static_cast<unsigned *>(state[j]);
  }
  f.zero = static_cast<int4 *>(zero);
  f.zero_n16 = (long)(zero_bytes / 16);
  const int zblocks = (int)std::min<long>(cdn::ceil_div(f.zero_n16, 256 * 4), 256);
  frozen_params_kernel<<<1 + zblocks, 256, 0, cdn::as_stream(stream)>>>(f);
  return cdn::check_launch("frozen QuantAct parameters");
}

extern "C" size_t cdn_codenet_stage_frozen_workspace_bytes(int64_t N, int64_t C, int64_t H, int64_t W, int x_up) {
  const int64_t HWl = (H >> x_up) * (W >> x_up);
  auto r = [](int64_t b) { return (b + 255) / 256 * 256; };
  return (size_t)(r(N * HWl * 4) + r(N * H * W * C));        // s_raw (fp32) + d (byte codes)
}

static int pointwise_q8_impl(const signed char *a, const void *a_state, int64_t M, int64_t C,
                             int64_t Co, const signed char *w_codes, const float *w_scale,
                             const int *w_colsum, const float *bias, int relu,
                             const void *r_state, signed char *r8_out, float *r_out,
                             unsigned *overflow, void *stream, const signed char *nsc, int *sacc,
                             int64_t lda = 0, int64_t ldo = 0, const int *omap = nullptr) {
  CDN_REQUIRE(a && a_state && w_codes && w_scale && w_colsum && overflow, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE((r8_out != nullptr) != (r_out != nullptr), CDN_ERR_ARG, "exactly one of r8_out / r_out");
  CDN_REQUIRE(r8_out == nullptr || r_state != nullptr, CDN_ERR_ARG, "byte output needs the output quantiser state");
  CDN_REQUIRE(M > 0 && C > 0 && Co > 0 && (((lda ? lda : C) & 3) == 0), CDN_ERR_ARG,
              "bad size (the row length -- lda, or C when dense -- must be a multiple of 4)");
  CDN_REQUIRE((lda == 0 || (lda >= C && (lda & 3) == 0)) && (ldo == 0 || ldo >= Co), CDN_ERR_ARG,
              "row strides must be 0 (dense) or >= the channel counts (lda a multiple of 4)");
  CDN_REQUIRE(M * std::max(std::max(C, Co), std::max(lda, ldo)) < (1ll << 31), CDN_ERR_UNSUPPORTED, "shape too large");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(w_codes) & 15) == 0 && (reinterpret_cast<uintptr_t>(a) & 3) == 0,
              CDN_ERR_ARG, "w_codes must be 16-byte, a 4-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  const unsigned *aq = static_cast<const unsigned *>(a_state), *rq = static_cast<const unsigned *>(r_state);
  // Tile choice: the kernel keeps ONE k tile in flight per workgroup, so it needs several workgroups per CU to
  // hide the global-load latency: the largest tile that still gives >= 4 workgroups per CU (measured at stage 0,
  // M = 16384, K = 1024, Co = 256: one 64 x 256 workgroup per CU 25.9 us).  A narrower N tile re-reads A through L2
  // (HBM reads it once: the column tiles of a row block are resident together), never through HBM.
  int bm = 128, bn = Co <= 64 ? 64 : (Co <= 128 ? 128 : 256);
  auto wgs = [&](int bm_, int bn_) { return cdn::ceil_div(M, bm_) * cdn::ceil_div(Co, bn_); };
  if (bn == 256) bm = 64;                          // (a 128 x 256 tile would need 128 accumulator registers per lane)
  // (round 4: "several" = 64 per CU, i.e. in practice the smallest tile.  Chosen with the launch alone on the GPU the
  // threshold was 4; inside the serving network -- 42 of these launches in chains of 9-23 us kernels -- smaller tiles
  // are worth 40 us per batch: 1.465-1.478 -> 1.427-1.437 ms at 64, 1.437-1.445 at 16, no change at 8; the deform
  // step's frozen legs: unchanged)
#ifndef CDN_Q8_WGS
#define CDN_Q8_WGS 64
#endif
  while (wgs(bm, bn) < (long)CDN_Q8_WGS * cdn::kCUs) {
    if (bm == 128) bm = 64;
    else if (bn > 64) bn >>= 1;
    else break;
  }
#define CDN_Q8(BM_, BN_) \
  launch_pwq8<BM_, BN_>(a, aq, w_codes, w_scale, w_colsum, bias, r8_out, r_out, rq, overflow, (long)M, (int)C, (int)Co, relu, st, nsc, sacc, (int)lda, (int)ldo, omap)
  if (bn == 256) CDN_Q8(64, 256);
  else if (bn == 128) { if (bm == 64) CDN_Q8(64, 128); else CDN_Q8(128, 128); }
  else { if (bm == 64) CDN_Q8(64, 64); else CDN_Q8(128, 64); }
#undef CDN_Q8
  return cdn::check_launch("codenet pointwise on byte codes");
}

extern "C" int cdn_codenet_pointwise_q8_forward(const signed char *a, const void *a_state, int64_t M, int64_t C,
                                                int64_t Co, const signed char *w_codes, const float *w_scale,
                                                const int *w_colsum, const float *bias, int relu,
                                                const void *r_state, signed char *r8_out, float *r_out,
                                                unsigned *overflow, void *stream) {
  return pointwise_q8_impl(a, a_state, M, C, Co, w_codes, w_scale, w_colsum, bias, relu, r_state, r8_out, r_out,
                           overflow, stream, nullptr, nullptr);
}

static int stage_frozen_impl(
    const void *x, int x_kind, int x_up, const void *x_state, int64_t N, int64_t C, int64_t Co, int64_t H, int64_t W,
    const float *w_scale, const float *b_scale, float lo, float hi, const float *w_dw,
    const signed char *w_pw_codes, const float *w_pw_scale, const int *w_pw_colsum, const float *bias_pw, int relu,
    const void *s_state, const void *d_state, const void *r_state, void *workspace, size_t workspace_bytes,
    signed char *r8_out, unsigned *overflow, void *stream, const int *s_acc_in, const float *w_scale_sw,
    const signed char *next_scale_codes, int *s_acc_out) {
  CDN_REQUIRE(x && (w_scale || s_acc_in) && w_dw && w_pw_codes && w_pw_scale && w_pw_colsum && s_state && d_state && r_state &&
                  workspace && r8_out && overflow, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE((x_kind & ~(3 | CDN_X_GATHER_MASK)) == 0 && ((x_kind & CDN_X_GATHER_MASK) >> 8) <= 2, CDN_ERR_ARG,
              "x_kind: 0 / 1 / 2, optionally | CDN_X_GATHER_PER_ITEM or CDN_X_GATHER_PERSISTENT");
  const int gmode = (x_kind & CDN_X_GATHER_MASK) >> 8;      // per-call schedule choice (tests); no library state
  x_kind &= 3;
  CDN_REQUIRE(x_kind >= 0 && x_kind <= 2 && (x_up == 0 || x_up == 1), CDN_ERR_ARG, "bad x_kind / x_up");
  CDN_REQUIRE((x_kind == 0) == (x_state == nullptr), CDN_ERR_ARG,
              "x_state goes with channels-last inputs (x_kind 1, 2) and only with them");
  CDN_REQUIRE(N > 0 && C > 0 && Co > 0 && H > 0 && W > 0, CDN_ERR_ARG, "non-positive size");
  CDN_REQUIRE(!x_up || ((H & 1) == 0 && (W & 1) == 0), CDN_ERR_SHAPE, "x_up needs even H, W");
  CDN_REQUIRE(x_up == 0 || x_kind != 0, CDN_ERR_UNSUPPORTED, "an up-sampled input must be channels-last");
  CDN_REQUIRE((C & 3) == 0, CDN_ERR_UNSUPPORTED, "byte codes need C %% 4 == 0 (got %lld)", (long long)C);
  CDN_REQUIRE(N <= 65535 && N * C * H * W < (1ll << 31) && N * Co * H * W < (1ll << 31), CDN_ERR_UNSUPPORTED,
              "shape too large");
  CDN_REQUIRE(workspace_bytes >= cdn_codenet_stage_frozen_workspace_bytes(N, C, H, W, x_up) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, CDN_ERR_WORKSPACE,
              "workspace too small or not 256-byte aligned");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(r8_out) & 15) == 0,
              CDN_ERR_ARG, "x / r8_out must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  const int64_t HWl = (H >> x_up) * (W >> x_up);
  auto r256 = [](int64_t b) { return (b + 255) / 256 * 256; };
  float *s_raw = static_cast<float *>(workspace);
  signed char *d8 = static_cast<signed char *>(workspace) + r256(N * HWl * 4);
  const unsigned *xq = static_cast<const unsigned *>(x_state);
  const int ptag = (int)(H > 0xffff ? 0xffff : H);
  int rc = 0;
  cdn::ScaleFromSums si{nullptr, nullptr, nullptr, 0.f, 0.f};
  if (s_acc_in) {
    // the producer's pointwise epilogue left the integer sums of this stage's scale prediction: no scale launch
    CDN_REQUIRE(x_kind == 2 && x_up == 1 && w_scale_sw, CDN_ERR_ARG,
                "scale sums come with an up-sampled byte-code input and the conv_scale weight scale");
    si = cdn::ScaleFromSums{s_acc_in, w_scale_sw, b_scale, lo, hi};
  } else {
    cdn::ProfScope ps(cdn::kProfScale, ptag, st);
    rc = cdn::launch_frozen_scale(x, x_kind, xq, w_scale, b_scale, s_raw, N, C, HWl, lo, hi, st);
  }
  if (rc) return rc;
  {
    cdn::ProfScope ps(cdn::kProfDw, ptag, st);
    rc = cdn::launch_frozen_dw(x, x_kind, xq, s_raw, static_cast<const unsigned *>(s_state), w_dw, d8,
                               static_cast<unsigned *>(const_cast<void *>(d_state)), overflow, (int)N, (int)C, (int)H,
                               (int)W, x_up, st, si, gmode);
  }
  if (rc) return rc;
  if (s_acc_out)      // (zero on entry: cleared by cdn_quantact_frozen_params_clear at the start of the step)
    CDN_REQUIRE(next_scale_codes, CDN_ERR_ARG, "s_acc_out needs the next stage's conv_scale weight codes");
  cdn::ProfScope ps(cdn::kProfPointwise, ptag, st);
  return pointwise_q8_impl(d8, d_state, N * H * W, C, Co, w_pw_codes, w_pw_scale, w_pw_colsum, bias_pw, relu, r_state,
                           r8_out, nullptr, overflow, stream, s_acc_out ? next_scale_codes : nullptr, s_acc_out);
}

extern "C" int cdn_codenet_stage_frozen_forward(
    const void *x, int x_kind, int x_up, const void *x_state, int64_t N, int64_t C, int64_t Co, int64_t H, int64_t W,
    const float *w_scale, const float *b_scale, float lo, float hi, const float *w_dw,
    const signed char *w_pw_codes, const float *w_pw_scale, const int *w_pw_colsum, const float *bias_pw, int relu,
    const void *s_state, const void *d_state, const void *r_state, void *workspace, size_t workspace_bytes,
    signed char *r8_out, unsigned *overflow, void *stream) {
  return stage_frozen_impl(x, x_kind, x_up, x_state, N, C, Co, H, W, w_scale, b_scale, lo, hi, w_dw, w_pw_codes,
                           w_pw_scale, w_pw_colsum, bias_pw, relu, s_state, d_state, r_state, workspace,
                           workspace_bytes, r8_out, overflow, stream, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int cdn_codenet_stage_frozen_chained_forward(
    const void *x, int x_kind, int x_up, const void *x_state, int64_t N, int64_t C, int64_t Co, int64_t H, int64_t W,
    const float *w_scale, const float *b_scale, float lo, float hi, const float *w_dw,
    const signed char *w_pw_codes, const float *w_pw_scale, const int *w_pw_colsum, const float *bias_pw, int relu,
    const void *s_state, const void *d_state, const void *r_state, void *workspace, size_t workspace_bytes,
    signed char *r8_out, unsigned *overflow, const int *s_sums_in, const float *w_scale_sw,
    const signed char *next_scale_codes, int *s_sums_out, void *stream) {
  return stage_frozen_impl(x, x_kind, x_up, x_state, N, C, Co, H, W, w_scale, b_scale, lo, hi, w_dw, w_pw_codes,
                           w_pw_scale, w_pw_colsum, bias_pw, relu, s_state, d_state, r_state, workspace,
                           workspace_bytes, r8_out, overflow, stream, s_sums_in, w_scale_sw, next_scale_codes,
                           s_sums_out);
}

extern "C" int cdn_codenet_expand_codes(const signed char *a, const void *a_state, float *out, int64_t numel,
                                        void *stream) {
  CDN_REQUIRE(a && a_state && out, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(numel > 0 && (numel & 3) == 0, CDN_ERR_ARG, "numel must be a positive multiple of 4");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(a) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, CDN_ERR_ARG,
              "a must be 4-byte, out 16-byte aligned");
  const long n4 = (long)(numel >> 2);
  const int blocks = (int)std::min<long>(cdn::ceil_div(n4, 256), (long)cdn::kCUs * 8);
  expand8_kernel<<<blocks, 256, 0, cdn::as_stream(stream)>>>(a, static_cast<const unsigned *>(a_state), out, n4);
  return cdn::check_launch("codenet expand codes");
}

extern "C" int cdn_codenet_pointwise_q8_strided_forward(
    const signed char *a, const void *a_state, int64_t M, int64_t C, int64_t Co, int64_t lda, int64_t ldo,
    const signed char *w_codes, const float *w_scale, const int *w_colsum, const float *bias, int relu,
    const int *out_map, const void *r_state, signed char *r8_out, float *r_out, unsigned *overflow, void *stream) {
  return pointwise_q8_impl(a, a_state, M, C, Co, w_codes, w_scale, w_colsum, bias, relu, r_state, r8_out, r_out,
                           overflow, stream, nullptr, nullptr, lda, ldo, out_map);
}

extern "C" int cdn_codenet_stem_q8_forward(const float *img, int64_t N, int64_t H, int64_t W, int64_t Co, int stride,
                                           const float *w, const float *bias, int relu, const void *r_state,
                                           signed char *out8, int64_t ld_out, unsigned *overflow, void *stream) {
  CDN_REQUIRE(img && w && r_state && out8 && overflow, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && H > 0 && W > 0 && stride >= 1 && N <= 65535, CDN_ERR_ARG, "bad size");
  CDN_REQUIRE(Co == 24, CDN_ERR_UNSUPPORTED, "stem kernel is instantiated for 24 output channels (got %lld)",
              (long long)Co);
  CDN_REQUIRE(ld_out >= Co && (ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(out8) & 3) == 0, CDN_ERR_ARG,
              "ld_out must be >= Co and a multiple of 4, out8 4-byte aligned");
  const int Ho = (int)((H + 2 - 3) / stride + 1), Wo = (int)((W + 2 - 3) / stride + 1);
  dim3 grid((unsigned)cdn::ceil_div((long)Ho * Wo, 256), (unsigned)N);
  stemq8_kernel<<<grid, 256, 0, cdn::as_stream(stream)>>>(img, w, bias, out8, static_cast<const unsigned *>(r_state),
                                                          overflow, (int)H, (int)W, Ho, Wo, stride, relu, (int)ld_out);
  return cdn::check_launch("codenet stem (byte codes)");
}

// 1 when cdn_codenet_dwpw_q8_forward implements this geometry (else: the two separate entry points)
extern "C" int cdn_codenet_dwpw_q8_supported(int64_t C, int64_t H, int64_t W, int stride, int64_t Co) {
  if (C <= 0 || H <= 0 || W <= 0 || Co <= 0 || Co > 256 || (stride != 1 && stride != 2)) return 0;
  const int64_t Ho = stride == 2 ? (H - 1) / 2 + 1 : H, Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
  const bool w_ok = Wo == 8 || Wo == 16 || Wo == 32 || (Wo >= 64 && Wo % 64 == 0);
  return (w_ok && (Ho * Wo) % 64 == 0 && (C + 63) / 64 * 64 <= 512) ? 1 : 0;
}

extern "C" int cdn_codenet_dwpw_q8_forward(
    const signed char *a8, const void *a_state, int64_t N, int64_t C, int64_t H, int64_t W, int stride, int64_t ld_in,
    const float *w_dw, const float *b_dw, int dw_relu, const void *d_state, int64_t Co, const signed char *w_codes,
    const float *w_scale, const int *w_colsum, const float *bias, int relu, int64_t ldo, const int *out_map,
    const void *r_state, signed char *r8_out, unsigned *overflow, void *stream) {
  CDN_REQUIRE(a8 && a_state && w_dw && d_state && w_codes && w_scale && w_colsum && r_state && r8_out && overflow,
              CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && cdn_codenet_dwpw_q8_supported(C, H, W, stride, Co), CDN_ERR_UNSUPPORTED,
              "geometry not implemented by the fused depthwise + pointwise kernel (see cdn_codenet_dwpw_q8_supported)");
  const int64_t Cq4 = (C + 3) / 4 * 4;
  CDN_REQUIRE(ld_in >= Cq4 && (ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(a8) & 3) == 0 &&
                  (ldo == 0 || ldo >= Co) && (reinterpret_cast<uintptr_t>(w_codes) & 15) == 0,
              CDN_ERR_ARG, "rows must hold round_up(C, 4) bytes, ld_in % 4 == 0, a8 4-byte / w_codes 16-byte aligned");
  const int Ho = stride == 2 ? (int)((H - 1) / 2 + 1) : (int)H, Wo = stride == 2 ? (int)((W - 1) / 2 + 1) : (int)W;
  const long M = (long)N * Ho * Wo;
  CDN_REQUIRE(N * H * W * ld_in < (1ll << 31) && M * std::max<int64_t>(Co, ldo) < (1ll << 31), CDN_ERR_UNSUPPORTED,
              "shape too large");
  hipStream_t st = cdn::as_stream(stream);
  const unsigned *aq = static_cast<const unsigned *>(a_state), *dq = static_cast<const unsigned *>(d_state);
  const unsigned *rq = static_cast<const unsigned *>(r_state);
  const int KP = (int)((C + 63) / 64 * 64);
  const int bn = Co <= 64 ? 64 : (Co <= 128 ? 128 : 256);
  // 32-row tiles where 64-row tiles leave at most one workgroup per CU (all-Co tiles, small planes: layer 4)
#ifndef CDN_DWPW_BM32_FILL
#define CDN_DWPW_BM32_FILL 1
#endif
  const bool bm32 = bn >= 128 && M / 64 <= (long)CDN_DWPW_BM32_FILL * cdn::kCUs && (Wo == 8 || Wo == 16 || Wo == 32) &&
                    (Ho * Wo) % 32 == 0;
  const int BMr = bm32 ? 32 : 64;
  const size_t lds = (size_t)BMr * (KP + 16) + (size_t)bn * (kQK + 16);
  const unsigned grid = (unsigned)(M / BMr);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<grid, 256, lds, st>>>(a8, aq, w_dw, b_dw, dw_relu, dq, w_codes, w_scale, w_colsum, bias, r8_out, rq, overflow,
                                 M, (int)C, KP, (int)ld_in, (int)H, (int)W, Ho, Wo, (int)Co, relu,
                                 (int)(ldo ? ldo : Co), out_map);
  };
#define CDN_DP_BN(S_, SW_, TR_)                                         \
  {                                                                     \
    if (bn == 64) go(dwpwq8_kernel<S_, SW_, TR_, 64>);                  \
    else if (bn == 128) go(dwpwq8_kernel<S_, SW_, TR_, 128>);           \
    else go(dwpwq8_kernel<S_, SW_, TR_, 256>);                          \
  }
  if (bm32) {
    const int tr32 = 32 / Wo;      // 4, 2 or 1 rows of 8, 16 or 32 columns
#define CDN_DP32(S_, SW_, TR_)                                          \
  {                                                                     \
    if (bn == 128) go(dwpwq8_kernel<S_, SW_, TR_, 128, 32>);            \
    else go(dwpwq8_kernel<S_, SW_, TR_, 256, 32>);                      \
  }
    if (stride == 1) {
      if (tr32 == 1) CDN_DP32(1, 4, 1)
      else if (tr32 == 2) CDN_DP32(1, 4, 2)
      else CDN_DP32(1, 4, 4)
    } else {
      if (tr32 == 1) CDN_DP32(2, 2, 1)
      else if (tr32 == 2) CDN_DP32(2, 2, 2)
      else CDN_DP32(2, 2, 4)
    }
#undef CDN_DP32
    return cdn::check_launch("codenet depthwise + pointwise on byte codes");
  }
  const int tr = Wo >= 64 ? 1 : 64 / Wo;
  if (stride == 1) {
    if (tr == 1) CDN_DP_BN(1, 4, 1)
    else if (tr == 2) CDN_DP_BN(1, 4, 2)
    else if (tr == 4) CDN_DP_BN(1, 4, 4)
    else CDN_DP_BN(1, 4, 8)
  } else {
    if (tr == 1) CDN_DP_BN(2, 2, 1)
    else if (tr == 2) CDN_DP_BN(2, 2, 2)
    else if (tr == 4) CDN_DP_BN(2, 2, 4)
    else CDN_DP_BN(2, 2, 8)
  }
#undef CDN_DP_BN
  return cdn::check_launch("codenet depthwise + pointwise on byte codes");
}

extern "C" int cdn_codenet_maxpool3x3s2_q8_forward(const signed char *a8, int64_t N, int64_t C, int64_t H, int64_t W,
                                                   int64_t ld_in, int64_t ld_out, signed char *out8, void *stream) {
  CDN_REQUIRE(a8 && out8, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, CDN_ERR_ARG, "bad size");
  const int64_t Cq4 = (C + 3) / 4 * 4;
  CDN_REQUIRE(ld_in >= Cq4 && ld_out >= Cq4 && (ld_in & 3) == 0 && (ld_out & 3) == 0 &&
                  (reinterpret_cast<uintptr_t>(a8) & 3) == 0 && (reinterpret_cast<uintptr_t>(out8) & 3) == 0,
              CDN_ERR_ARG, "rows must hold round_up(C, 4) bytes, strides multiples of 4, pointers 4-byte aligned");
  const int Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
  const long total = (long)N * Ho * Wo * ((C + 3) / 4);
  CDN_REQUIRE(N * H * W * ld_in < (1ll << 31) && total < (1ll << 31), CDN_ERR_UNSUPPORTED, "shape too large");
  maxq8_kernel<<<(unsigned)std::min<long>(cdn::ceil_div(total, 256), (long)cdn::kCUs * 32), 256, 0, cdn::as_stream(stream)>>>(
      a8, out8, (int)C, (int)ld_in, (int)ld_out, (int)H, (int)W, Ho, Wo, total);
  return cdn::check_launch("codenet maxpool (byte codes)");
}

extern "C" int cdn_codenet_dw3x3_q8_forward(const signed char *a8, const void *a_state, int64_t N, int64_t C, int64_t H,
                                            int64_t W, int stride, int64_t ld_in, int64_t ld_out, const float *w,
                                            const float *bias, int relu, const void *r_state, signed char *out8,
                                            unsigned *overflow, void *stream) {
  CDN_REQUIRE(a8 && a_state && w && r_state && out8 && overflow, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), CDN_ERR_ARG, "bad size / stride");
  const int64_t Cq4 = (C + 3) / 4 * 4;
  CDN_REQUIRE(ld_in >= Cq4 && ld_out >= Cq4 && (ld_in & 3) == 0 && (ld_out & 3) == 0 &&
                  (reinterpret_cast<uintptr_t>(a8) & 3) == 0 && (reinterpret_cast<uintptr_t>(out8) & 3) == 0,
              CDN_ERR_ARG, "rows must hold round_up(C, 4) bytes, strides multiples of 4, pointers 4-byte aligned");
  const int Ho = stride == 2 ? (int)((H - 1) / 2 + 1) : (int)H, Wo = stride == 2 ? (int)((W - 1) / 2 + 1) : (int)W;
  CDN_REQUIRE(N * H * W * ld_in < (1ll << 31) && N * Ho * (int64_t)Wo * ld_out < (1ll << 31), CDN_ERR_UNSUPPORTED,
              "shape too large");
  hipStream_t st = cdn::as_stream(stream);
  const unsigned *aq = static_cast<const unsigned *>(a_state), *rq = static_cast<const unsigned *>(r_state);
  // rows per thread: 8 while that still gives >= 2 workgroups per CU, else 4, else 2 (more halo rows decoded twice,
  // more threads in flight: the deep layers' planes are 16 x 16)
  const int CQ = (int)((C + 3) / 4);
  auto launch = [&](auto stride_c, auto sw_c, auto ys_c) {
    constexpr int S = decltype(stride_c)::value, SW = decltype(sw_c)::value, YS = decltype(ys_c)::value;
    const int strips = cdn::ceil_div(Wo, SW), ysegs = cdn::ceil_div(Ho, YS);
    const long total = (long)N * ysegs * strips * CQ;
    dwq8_kernel<S, SW, YS><<<(unsigned)cdn::ceil_div(total, 256), 256, 0, st>>>(
        a8, aq, w, bias, out8, rq, overflow, (int)C, (int)ld_in, (int)ld_out, (int)H, (int)W, Ho, Wo, relu, strips,
        ysegs, total);
  };
  auto wgs = [&](int sw, int ys) { return (long)N * cdn::ceil_div(Ho, ys) * cdn::ceil_div(Wo, sw) * CQ / 256; };
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  if (stride == 2) {
    if (wgs(2, 8) >= 2L * cdn::kCUs) launch(I2{}, I2{}, I8{});
    else if (wgs(2, 4) >= 2L * cdn::kCUs) launch(I2{}, I2{}, I4{});
    else launch(I2{}, I2{}, I2{});
  } else {
    if (wgs(4, 8) >= 2L * cdn::kCUs) launch(I1{}, I4{}, I8{});
    else if (wgs(4, 4) >= 2L * cdn::kCUs) launch(I1{}, I4{}, I4{});
    else launch(I1{}, I4{}, I2{});
  }
  return cdn::check_launch("codenet dw3x3 (byte codes)");
}
