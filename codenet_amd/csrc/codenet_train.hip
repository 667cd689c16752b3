// codenet_train.hip -- backward kernels of the two 1x1 convolutions around the gather (config e, the QAT step of
// quant_main.py).  The reference runs them as cuDNN / cuBLAS calls under autograd:
//   conv_scale   (modules/dcn_deform_conv.py:295,324; Quant_Conv2d quant_modules.py:314-321)   C -> 1, bias
//   conv_channel (modules/dcn_deform_conv.py:311-312,328; QuantBnConv2d quant_modules.py:412-419)  C -> Co
// Forward and data-gradient of conv_channel are cdn_codenet_pointwise_forward (the data gradient is the same
// contraction with the transposed weights); here: its weight gradient on f32 MFMA, and the backward of the
// scale prediction.  NCHW fp32, like the module path (codenet_stage.hip).
#include "cdn_common.h"

#include <algorithm>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------------------
// pw_wgrad_kernel: partial[z][co][c] = sum over the k-chunks of slice z of  gy[n][co][p] * d[n][c][p].
// Both operands have the reduction index (pixels) contiguous, so this is an "NT" GEMM with K = N*HW:
//   workgroup tile 64 (co) x 64 (c), 4 waves x one 32x32 accumulator, K tiles of 64 pixels of ONE image
//   staged through LDS as [row][k] (row stride 68 floats: a quarter-wave's 16 ds_read_b128 cover the 64 banks);
//   lane l reads 4 consecutive k (k = 4*(l>>5) + e of every 8): the two lane halves are the two k slots of
//   v_mfma_f32_32x32x2_f32 (A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]) and e = 0..3 are four MFMAs -- a sum
//   over k does not care which slot a k lands in, only that A and B agree.
//   Next tile's global loads (8 x 16 B per thread) are in flight behind the 32 MFMAs of the current one.
// Split-K over gridDim.z (few output tiles: 64 x 128 outputs at stage 2), partial tiles reduced in a fixed
// order by wgrad_reduce_kernel: bitwise reproducible.
// ------------------------------------------------------------------------------------------------------
constexpr int kWgBM = 64, kWgBN = 64, kWgBK = 64, kWgLd = kWgBK + 4;

__global__ void __launch_bounds__(256)
pw_wgrad_kernel(const float *__restrict__ gy, const float *__restrict__ d, float *__restrict__ partial,
                float *__restrict__ partial_b, int C, int Co, int HW, int chunks_per_img, int nchunks,
                int chunks_per_slice, const unsigned *__restrict__ dq) {
  __shared__ __attribute__((aligned(16))) float As[kWgBM][kWgLd];
  __shared__ __attribute__((aligned(16))) float Bs[kWgBN][kWgLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * kWgBN, m0 = blockIdx.y * kWgBM, z = blockIdx.z;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int ch_lo = z * chunks_per_slice, ch_hi = min(nchunks, ch_lo + chunks_per_slice);
  const int lrow = tid >> 4, lk = (tid & 15) * 4;      // staging: rows lrow + 16*i, floats lk .. lk+3
  const bool hw4 = (HW & 3) == 0;
  f32x16 acc = {0};
  float4 ra[4], rb[4];
  // bias gradient = row sums of grad_y: taken by the workgroups of channel tile 0 from the A tiles as they pass
  // through registers (a separate one-workgroup-per-channel kernel took 175 us per step)
  const bool do_bias = partial_b != nullptr && blockIdx.x == 0;
  float dqs = 1.f, dqz = 0.f, dqr = 1.f;
  if (dq) {
    dqs = reinterpret_cast<const float *>(dq)[2];
    dqz = reinterpret_cast<const float *>(dq)[3];
    dqr = __fdiv_rn(1.0f, dqs);
  }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // fast: whole tiles and 16-byte aligned rows (the stage shapes of the QAT step): eight unconditional 16-byte loads
  const bool fast = hw4 && (HW % kWgBK) == 0 && (Co % kWgBM) == 0 && (C % kWgBN) == 0 &&
                    ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
  auto load = [&](int ch) {
    const int n = ch / chunks_per_img, p0 = (ch - n * chunks_per_img) * kWgBK;
    const float *ga = gy + (long)n * Co * HW, *gb = d + (long)n * C * HW;
    if (fast) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = lrow + 16 * i, p = p0 + lk;
        ra[i] = *reinterpret_cast<const float4 *>(ga + (long)(m0 + r) * HW + p);
        rb[i] = *reinterpret_cast<const float4 *>(gb + (long)(c0 + r) * HW + p);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lrow + 16 * i, p = p0 + lk;
      const int ma = m0 + r, cb = c0 + r;
      if (hw4 && p + 3 < HW) {
        ra[i] = ma < Co ? *reinterpret_cast<const float4 *>(ga + (long)ma * HW + p) : make_float4(0, 0, 0, 0);
        rb[i] = cb < C ? *reinterpret_cast<const float4 *>(gb + (long)cb * HW + p) : make_float4(0, 0, 0, 0);
      } else {
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = (ma < Co && p + e < HW) ? ga[(long)ma * HW + p + e] : 0.0f;
          b[e] = (cb < C && p + e < HW) ? gb[(long)cb * HW + p + e] : 0.0f;
        }
        ra[i] = make_float4(a[0], a[1], a[2], a[3]);
        rb[i] = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
  };
  if (ch_lo < ch_hi) load(ch_lo);
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    __syncthreads();                       // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (dq) {      // d holds pre-quantisation values: fake-quantised with its QuantAct state when the tile is parked
                     // (inside load() it consumed the prefetch in front of the MFMAs it was meant to hide behind)
        rb[i].x = cdn::fake_quant_r(rb[i].x, dqs, dqz, dqr); rb[i].y = cdn::fake_quant_r(rb[i].y, dqs, dqz, dqr);
        rb[i].z = cdn::fake_quant_r(rb[i].z, dqs, dqz, dqr); rb[i].w = cdn::fake_quant_r(rb[i].w, dqs, dqz, dqr);
      }
      *reinterpret_cast<float4 *>(&As[lrow + 16 * i][lk]) = ra[i];
      *reinterpret_cast<float4 *>(&Bs[lrow + 16 * i][lk]) = rb[i];
      if (do_bias) bsum[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
    }
    __syncthreads();
    if (ch + 1 < ch_hi) load(ch + 1);      // in flight during the MFMAs below
    const float *ap = &As[wm + (lane & 31)][4 * (lane >> 5)];
    const float *bp = &Bs[wn + (lane & 31)][4 * (lane >> 5)];
#pragma unroll
    for (int kk = 0; kk < kWgBK; kk += 8) {
      const float4 a4 = *reinterpret_cast<const float4 *>(ap + kk);
      const float4 b4 = *reinterpret_cast<const float4 *>(bp + kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
  }
  // C/D layout: column (B row = c) = l & 31, row (A row = co) = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
  float *out = partial + (long)z * Co * C;
  const int col = c0 + wn + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < Co && col < C) out[(long)row * C + col] = acc[r];
  }
  if (do_bias) {       // the 16 threads that share a row: lanes 16*j .. 16*j+15 of a wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int m = 8; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
      const int row = m0 + lrow + 16 * i;
      if ((tid & 15) == 0 && row < Co) partial_b[(long)z * Co + row] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// pw_wgrad_q3_kernel (round 5): the same partial tiles on the bf16 matrix cores with EXACT products, for d given as
// pre-quantisation values + the QuantAct state that quantises them (the QAT step).  d' = (q' + 128) / scale with the
// integer q' = rint(scale x - zp) + zp - 128 (|q'| <= 2040 unless the state's wide flag is set: two bf16 terms hold 16
// bits), grad_y = hi + mid + lo in three bf16 terms by truncation (exact):
//   partial[z][co][c] = sum_k grad_y[co][k] q'[c][k]        6 v_mfma_f32_32x32x16_bf16 per 16 k, every product exact
//   grad_w = (sum_z partial + 128 grad_b[co]) / scale       (wgrad_reduce_q_kernel; grad_b = row sums of grad_y)
// against 8 v_mfma_f32_32x32x2_f32 per 16 k in pw_wgrad_kernel (41 / 37 / 56 us at the step's three stages, matrix-core
// bound).  Workgroup tile 64 (co) x 128 (c): both operands cross L2 less often than with 64 x 64 tiles; waves 2 x 2, each
// a 32 x 64 tile; K tiles of 64 pixels of one image staged through LDS as bf16 rows of 64 k + 16 bytes of padding
// (five planes: 64.5 KB, two workgroups per CU); the next tile's global loads in flight behind the MFMAs.
// Whole tiles only (Co % 64, C % 128, HW % 64, 16-byte aligned rows): everything else stays on pw_wgrad_kernel.
// ------------------------------------------------------------------------------------------------------
using bf16x8w = __attribute__((ext_vector_type(8))) __bf16;
using i32x4w = __attribute__((ext_vector_type(4))) int;
constexpr int kQ3BM = 64, kQ3BN = 128, kQ3BK = 64, kQ3LD = 144;      // bytes per LDS row: 64 bf16 + 16

__global__ void __launch_bounds__(256)
pw_wgrad_q3_kernel(const float *__restrict__ gy, const float *__restrict__ d, float *__restrict__ partial,
                   float *__restrict__ partial_b, int C, int Co, int HW, int chunks_per_img, int nchunks,
                   int chunks_per_slice, const unsigned *__restrict__ dq) {
  extern __shared__ float4 q3_lds[];
  unsigned char *lds = reinterpret_cast<unsigned char *>(q3_lds);
  unsigned char *Ah = lds, *Am = Ah + kQ3BM * kQ3LD, *Al = Am + kQ3BM * kQ3LD;
  unsigned char *Bh = Al + kQ3BM * kQ3LD, *Bl = Bh + kQ3BN * kQ3LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * kQ3BN, m0 = blockIdx.y * kQ3BM, z = blockIdx.z;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
  const int ch_lo = z * chunks_per_slice, ch_hi = min(nchunks, ch_lo + chunks_per_slice);
  const int lrow = tid >> 4, lk = (tid & 15) * 4;      // staging: rows lrow + 16*i, floats lk .. lk+3
  const float qs = reinterpret_cast<const float *>(dq)[2], qz = reinterpret_cast<const float *>(dq)[3];
  const float qoff = qz - 128.0f;                      // (integers: exact)
  f32x16 acc[2];
  acc[0] = (f32x16){0};
  acc[1] = (f32x16){0};
  float4 ra[4], rb[8];
  const bool do_bias = blockIdx.x == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int ch) {
    const int n = ch / chunks_per_img, p0 = (ch - n * chunks_per_img) * kQ3BK;
    const float *ga = gy + (long)n * Co * HW + p0 + lk, *gb = d + (long)n * C * HW + p0 + lk;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4 *>(ga + (long)(m0 + lrow + 16 * i) * HW);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const float4 *>(gb + (long)(c0 + lrow + 16 * i) * HW);
  };
  auto pack_hi = [](unsigned a_, unsigned b_) -> unsigned { return __builtin_amdgcn_perm(a_, b_, 0x07060302u); };
  if (ch_lo < ch_hi) load(ch_lo);
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    __syncthreads();                       // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      unsigned hb[4], mb[4], lb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hb[e] = __float_as_uint(v[e]);
        const float r1 = __fsub_rn(v[e], __uint_as_float(hb[e] & 0xFFFF0000u));
        mb[e] = __float_as_uint(r1);
        lb[e] = __float_as_uint(__fsub_rn(r1, __uint_as_float(mb[e] & 0xFFFF0000u)));
      }
      const int o = (lrow + 16 * i) * kQ3LD + lk * 2;
      *reinterpret_cast<uint2 *>(Ah + o) = make_uint2(pack_hi(hb[1], hb[0]), pack_hi(hb[3], hb[2]));
      *reinterpret_cast<uint2 *>(Am + o) = make_uint2(pack_hi(mb[1], mb[0]), pack_hi(mb[3], mb[2]));
      *reinterpret_cast<uint2 *>(Al + o) = make_uint2(pack_hi(lb[1], lb[0]), pack_hi(lb[3], lb[2]));
      if (do_bias) bsum[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
      unsigned hb[4], lb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float q = cdn::quant_code(v[e], qs, qz) + qoff;      // q' (an integer-valued float)
        hb[e] = __float_as_uint(q);
        lb[e] = __float_as_uint(__fsub_rn(q, __uint_as_float(hb[e] & 0xFFFF0000u)));
      }
      const int o = (lrow + 16 * i) * kQ3LD + lk * 2;
      *reinterpret_cast<uint2 *>(Bh + o) = make_uint2(pack_hi(hb[1], hb[0]), pack_hi(hb[3], hb[2]));
      *reinterpret_cast<uint2 *>(Bl + o) = make_uint2(pack_hi(lb[1], lb[0]), pack_hi(lb[3], lb[2]));
    }
    __syncthreads();
    if (ch + 1 < ch_hi) load(ch + 1);      // in flight during the MFMAs below
    const int fo = (lane & 31) * kQ3LD + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ao = (wm)*kQ3LD + fo + ks * 32;
      const bf16x8w fh = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const i32x4w *>(Ah + ao));
      const bf16x8w fm = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const i32x4w *>(Am + ao));
      const bf16x8w fl = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const i32x4w *>(Al + ao));
      bf16x8w bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int bo = (wn + 32 * j) * kQ3LD + fo + ks * 32;
        bh[j] = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const i32x4w *>(Bh + bo));
        bl[j] = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const i32x4w *>(Bl + bo));
      }
      // smallest terms first, the two accumulators interleaved (back-to-back MFMAs into one accumulator stall)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bh[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh[j], acc[j], 0, 0, 0);
    }
  }
  float *out = partial + (long)z * Co * C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = c0 + wn + 32 * j + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(long)row * C + col] = acc[j][r];
    }
  }
  if (do_bias) {       // the 16 threads that share a row: lanes 16*j .. 16*j+15 of a wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int m = 8; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
      if ((tid & 15) == 0) partial_b[(long)z * Co + m0 + lrow + 16 * i] = v;
    }
  }
}

// grad_w = (sum_z partial + 128 grad_b[co]) / scale and grad_b = sum_z partial_b, fixed orders, one launch: a workgroup
// owns 64 consecutive outputs of ONE row (C % 64 == 0) and first reduces that row's bias sum.
__global__ void __launch_bounds__(256)
wgrad_reduce_q_kernel(const float *__restrict__ partial, float *__restrict__ gw, int C, int Co, int nz,
                      const float *__restrict__ partial_b, float *__restrict__ gb, const unsigned *__restrict__ dq) {
  __shared__ float red[4][64];
  __shared__ float rb[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long n = (long)Co * C;
  const long i = (long)blockIdx.x * 64 + lane;
  const int co = (int)(((long)blockIdx.x * 64) / C);
  // the row's bias sum: wave w takes z = w, w + 4, ... lane-strided, then a fixed-order tree
  float b = 0.f;
  for (int zz = w * 64 + lane; zz < nz; zz += 256) b += partial_b[(long)zz * Co + co];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) b += __shfl_xor(b, m, 64);
  if (lane == 0) rb[w] = b;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int zi = w;
  for (; zi + 12 < nz; zi += 16) {
    s0 += partial[(long)zi * n + i];
    s1 += partial[(long)(zi + 4) * n + i];
    s2 += partial[(long)(zi + 8) * n + i];
    s3 += partial[(long)(zi + 12) * n + i];
  }
  for (; zi < nz; zi += 4) s0 += partial[(long)zi * n + i];
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0) {
    const float gbv = (rb[0] + rb[1]) + (rb[2] + rb[3]);
    const float S = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const float sc = reinterpret_cast<const float *>(dq)[2];
    gw[i] = __fdiv_rn(fmaf(128.0f, gbv, S), sc);
    if (gb != nullptr && lane == 0 && ((long)blockIdx.x * 64) % C == 0) gb[co] = gbv;
  }
}

// gw[i] = sum_z partial[z][i] in a fixed order: a workgroup owns 64 outputs, its 4 waves take z = w, w+4, ...
// with 4 loads in flight, then the 4 wave sums are added in wave order (with few outputs -- 64 x 128 at stage 2 --
// one thread per output walking hundreds of slices was a 36 us latency chain).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw, long n, int nz,
                    const float *__restrict__ partial_b, float *__restrict__ gb, long nb) {
  // (the bias rows ride along: the workgroups behind the weight outputs reduce partial_b [nz][nb] -- one launch)
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wblocks = (n + 63) / 64;
  const bool bias = (long)blockIdx.x >= wblocks;
  const float *src = bias ? partial_b : partial;
  float *dst = bias ? gb : gw;
  const long len = bias ? nb : n;
  const long i = ((long)blockIdx.x - (bias ? wblocks : 0)) * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < len) {
    int z = w;
    for (; z + 12 < nz; z += 16) {
      s0 += src[(long)z * len + i];
      s1 += src[(long)(z + 4) * len + i];
      s2 += src[(long)(z + 8) * len + i];
      s3 += src[(long)(z + 12) * len + i];
    }
    for (; z < nz; z += 4) s0 += src[(long)z * len + i];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && i < len) dst[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------------------------------------------
// scale_bwd_kernel: backward of s_raw[n][p] = sum_c w[c] x[n][c][p] + b given g[n][p] = dL/ds_raw (the caller
// has applied the Hardtanh mask and the straight-through QuantAct):
//   grad_x[n][c][p] += w[c] * g[n][p]           (accumulated into the gather's grad_x)
//   gw_part[n][c]    = sum_p x[n][c][p] g[n][p]  (summed over n by the caller in a fixed order)
// One wave per (image, channel) row; g (N*HW floats) stays in L2.  HBM: read x, read + write grad_x.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
scale_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ sc, float lo,
                 float hi, const float *__restrict__ w, float *__restrict__ gx, float *__restrict__ gw_part, int C,
                 int HW, int ldp) {
  const int n = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float *gp = g + (long)n * HW, *sp = sc ? sc + (long)n * HW : nullptr;
  // Hardtanh backward folded in (sc != NULL): the gradient passes where lo < s_clamped < hi
  auto mask4 = [&](float4 gv, int q) {
    if (sp) {
      const float4 sv = reinterpret_cast<const float4 *>(sp)[q];
      gv.x = (sv.x > lo && sv.x < hi) ? gv.x : 0.0f; gv.y = (sv.y > lo && sv.y < hi) ? gv.y : 0.0f;
      gv.z = (sv.z > lo && sv.z < hi) ? gv.z : 0.0f; gv.w = (sv.w > lo && sv.w < hi) ? gv.w : 0.0f;
    }
    return gv;
  };
  auto mask1 = [&](float gv, int p) { return (sp && !(sp[p] > lo && sp[p] < hi)) ? 0.0f : gv; };
  if (c >= C) {
    // the extra workgroup row (ldp > C): column C of the partials = sum_p g_raw[n][p], the bias gradient's share
    if (c == C && ldp > C && gw_part) {
      float acc = 0.0f;
      if ((HW & 3) == 0) {
        for (int q = lane; q < (HW >> 2); q += 64) {
          const float4 gv = mask4(reinterpret_cast<const float4 *>(gp)[q], q);
          acc += (gv.x + gv.y) + (gv.z + gv.w);
        }
      } else {
        for (int p = lane; p < HW; p += 64) acc += mask1(gp[p], p);
      }
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
      if (lane == 0) gw_part[(long)n * ldp + C] = acc;
    }
    return;
  }
  const float wc = w[c];
  const float *xp = x + ((long)n * C + c) * HW;
  float *gxp = gx ? gx + ((long)n * C + c) * HW : nullptr;
  float acc = 0.0f;
  if ((HW & 3) == 0) {
    for (int q = lane; q < (HW >> 2); q += 64) {
      const float4 xv = reinterpret_cast<const float4 *>(xp)[q];
      const float4 gv = mask4(reinterpret_cast<const float4 *>(gp)[q], q);
      acc = fmaf(xv.x, gv.x, acc); acc = fmaf(xv.y, gv.y, acc);
      acc = fmaf(xv.z, gv.z, acc); acc = fmaf(xv.w, gv.w, acc);
      if (gxp) {
        float4 o = reinterpret_cast<float4 *>(gxp)[q];
        o.x = fmaf(wc, gv.x, o.x); o.y = fmaf(wc, gv.y, o.y);
        o.z = fmaf(wc, gv.z, o.z); o.w = fmaf(wc, gv.w, o.w);
        reinterpret_cast<float4 *>(gxp)[q] = o;
      }
    }
  } else {
    for (int p = lane; p < HW; p += 64) {
      const float gv = mask1(gp[p], p);
      acc = fmaf(xp[p], gv, acc);
      if (gxp) gxp[p] = fmaf(wc, gv, gxp[p]);
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0 && gw_part) gw_part[(long)n * ldp + c] = acc;
}

struct WgPlan {
  int tiles_c, tiles_m, chunks_per_img, nchunks, nz, per_slice;
};

WgPlan wgrad_plan(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  WgPlan p;
  p.tiles_c = (int)cdn::ceil_div(C, kWgBN);
  p.tiles_m = (int)cdn::ceil_div(Co, kWgBM);
  p.chunks_per_img = (int)cdn::ceil_div(HW, kWgBK);
  p.nchunks = (int)(N * p.chunks_per_img);
  // ~3 workgroups per CU in total, at least 2 K tiles per slice, at most 1024 slices
  int nz = (int)std::max<int64_t>(1, (3 * cdn::kCUs) / ((int64_t)p.tiles_c * p.tiles_m));
  nz = std::min(nz, std::max(1, p.nchunks / 2));
  nz = std::min(nz, 1024);
  p.per_slice = (int)cdn::ceil_div(p.nchunks, nz);
  p.nz = (int)cdn::ceil_div(p.nchunks, p.per_slice);
  return p;
}

}  // namespace

extern "C" size_t cdn_codenet_pointwise_wgrad_workspace_bytes(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  if (N <= 0 || C <= 0 || Co <= 0 || HW <= 0) return 0;
  const WgPlan p = wgrad_plan(N, C, Co, HW);
  return ((size_t)p.nz * (size_t)Co * (size_t)(C + 1) * 4 + 255) / 256 * 256;   // weight tiles + bias rows
}

static int pointwise_wgrad_impl(const float *grad_y, const float *d, const void *d_state, float *grad_w, float *grad_b,
                                int64_t N, int64_t C, int64_t Co, int64_t HW, void *workspace, size_t workspace_bytes,
                                void *stream, bool f32_only = false) {
  CDN_REQUIRE(grad_y && d && grad_w && workspace, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && Co > 0 && HW > 0, CDN_ERR_ARG, "non-positive size");
  CDN_REQUIRE(N * C * HW < (1ll << 31) && N * Co * HW < (1ll << 31) && C * Co < (1ll << 31),
              CDN_ERR_UNSUPPORTED, "shape too large");
  CDN_REQUIRE(workspace_bytes >= cdn_codenet_pointwise_wgrad_workspace_bytes(N, C, Co, HW) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              CDN_ERR_WORKSPACE, "workspace too small or not 16-byte aligned");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(grad_y) & 15) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0,
              CDN_ERR_ARG, "grad_y / d must be 16-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
#if !defined(CDN_NO_WGRAD_Q3)
  if (d_state && !f32_only && (Co % kQ3BM) == 0 && (C % kQ3BN) == 0 && (HW % kQ3BK) == 0) {
    // exact bf16 products on the integer form of d (pw_wgrad_q3_kernel): whole 64 x 128 tiles
    const int tiles_c = (int)(C / kQ3BN), tiles_m = (int)(Co / kQ3BM);
    const int chunks_per_img = (int)(HW / kQ3BK), nchunks = (int)(N * chunks_per_img);
    int nz = (int)std::max<int64_t>(1, (2 * cdn::kCUs) / ((int64_t)tiles_c * tiles_m));
    nz = std::min(nz, std::max(1, nchunks / 2));
    const WgPlan pl = wgrad_plan(N, C, Co, HW);
    nz = std::min(nz, pl.nz);                    // (the workspace is sized for the f32 kernel's slice count)
    const int per_slice = (int)cdn::ceil_div(nchunks, nz);
    nz = (int)cdn::ceil_div(nchunks, per_slice);
    float *part = static_cast<float *>(workspace);
    float *part_b = part + (size_t)nz * Co * C;
    constexpr int kLds = (3 * kQ3BM + 2 * kQ3BN) * kQ3LD;
    (void)hipFuncSetAttribute((const void *)pw_wgrad_q3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    pw_wgrad_q3_kernel<<<dim3((unsigned)tiles_c, (unsigned)tiles_m, (unsigned)nz), 256, kLds, st>>>(
        grad_y, d, part, part_b, (int)C, (int)Co, (int)HW, chunks_per_img, nchunks, per_slice,
        static_cast<const unsigned *>(d_state));
    int rc = cdn::check_launch("codenet pointwise weight gradient (bf16 x 3)");
    if (rc) return rc;
    wgrad_reduce_q_kernel<<<(unsigned)((long)Co * C / 64), 256, 0, st>>>(part, grad_w, (int)C, (int)Co, nz, part_b, grad_b,
                                                                        static_cast<const unsigned *>(d_state));
    return cdn::check_launch("codenet pointwise weight / bias gradient reduce");
  }
#endif
  const WgPlan p = wgrad_plan(N, C, Co, HW);
  CDN_REQUIRE(p.tiles_m <= 65535 && p.nz <= 65535, CDN_ERR_UNSUPPORTED, "too many tiles");
  float *partial = static_cast<float *>(workspace);
  float *partial_b = grad_b ? partial + (size_t)p.nz * Co * C : nullptr;
  pw_wgrad_kernel<<<dim3((unsigned)p.tiles_c, (unsigned)p.tiles_m, (unsigned)p.nz), 256, 0, st>>>(
      grad_y, d, partial, partial_b, (int)C, (int)Co, (int)HW, p.chunks_per_img, p.nchunks, p.per_slice,
      static_cast<const unsigned *>(d_state));
  int rc = cdn::check_launch("codenet pointwise weight gradient");
  if (rc) return rc;
  const long n = (long)(Co * C);
  const long blocks = cdn::ceil_div(n, 64) + (grad_b ? cdn::ceil_div(Co, 64) : 0);
  wgrad_reduce_kernel<<<(unsigned)blocks, 256, 0, st>>>(partial, grad_w, n, p.nz, partial_b, grad_b, (long)Co);
  return cdn::check_launch("codenet pointwise weight / bias gradient reduce");
}

extern "C" int cdn_codenet_pointwise_wgrad(const float *grad_y, const float *d, float *grad_w, float *grad_b,
                                           int64_t N, int64_t C, int64_t Co, int64_t HW, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  return pointwise_wgrad_impl(grad_y, d, nullptr, grad_w, grad_b, N, C, Co, HW, workspace, workspace_bytes, stream);
}

extern "C" int cdn_codenet_pointwise_wgrad_q(const float *grad_y, const float *d, const void *d_state, float *grad_w,
                                             float *grad_b, int64_t N, int64_t C, int64_t Co, int64_t HW,
                                             void *workspace, size_t workspace_bytes, void *stream) {
  CDN_REQUIRE(d_state, CDN_ERR_ARG, "null pointer");
  return pointwise_wgrad_impl(grad_y, d, d_state, grad_w, grad_b, N, C, Co, HW, workspace, workspace_bytes, stream);
}
extern "C" int cdn_codenet_pointwise_wgrad_q_f32(const float *grad_y, const float *d, const void *d_state, float *grad_w,
                                                 float *grad_b, int64_t N, int64_t C, int64_t Co, int64_t HW,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
  CDN_REQUIRE(d_state, CDN_ERR_ARG, "null pointer");
  return pointwise_wgrad_impl(grad_y, d, d_state, grad_w, grad_b, N, C, Co, HW, workspace, workspace_bytes, stream, true);
}

static int scale_backward_impl(const float *x, const float *grad_s, const float *s_clamped, float lo, float hi,
                               const float *w_scale, float *grad_x, float *grad_w_partial, int with_bias, int64_t N,
                               int64_t C, int64_t H, int64_t W, void *stream) {
  CDN_REQUIRE(x && grad_s && w_scale, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, CDN_ERR_ARG, "non-positive size");
  CDN_REQUIRE(N <= 65535 && N * C * H * W < (1ll << 31), CDN_ERR_UNSUPPORTED, "shape too large");
  const int HW = (int)(H * W);
  CDN_REQUIRE((HW & 3) != 0 || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                                (reinterpret_cast<uintptr_t>(grad_s) & 15) == 0 &&
                                (reinterpret_cast<uintptr_t>(s_clamped) & 15) == 0 &&
                                (reinterpret_cast<uintptr_t>(grad_x) & 15) == 0),
              CDN_ERR_ARG, "x / grad_s / s_clamped / grad_x must be 16-byte aligned");
  const int extra = (with_bias && grad_w_partial) ? 1 : 0;       // channel index C: the bias column
  dim3 grid((unsigned)cdn::ceil_div(C + extra, 4), (unsigned)N);
  scale_bwd_kernel<<<grid, 256, 0, cdn::as_stream(stream)>>>(x, grad_s, s_clamped, lo, hi, w_scale, grad_x,
                                                             grad_w_partial, (int)C, HW, (int)C + extra);
  return cdn::check_launch("codenet scale backward");
}

extern "C" int cdn_codenet_scale_backward(const float *x, const float *grad_s, const float *w_scale, float *grad_x,
                                          float *grad_w_partial, int64_t N, int64_t C, int64_t H, int64_t W,
                                          void *stream) {
  return scale_backward_impl(x, grad_s, nullptr, 0.f, 0.f, w_scale, grad_x, grad_w_partial, 0, N, C, H, W, stream);
}

extern "C" int cdn_codenet_scale_backward_masked(const float *x, const float *grad_s, const float *s_clamped,
                                                 float lo, float hi, const float *w_scale, float *grad_x,
                                                 float *grad_wb_partial, int64_t N, int64_t C, int64_t H, int64_t W,
                                                 void *stream) {
  CDN_REQUIRE(s_clamped, CDN_ERR_ARG, "null pointer");
  return scale_backward_impl(x, grad_s, s_clamped, lo, hi, w_scale, grad_x, grad_wb_partial, 1, N, C, H, W, stream);
}

namespace {

// ------------------------------------------------------------------------------------------------------
// weight_prep_kernel: the per-step WEIGHT transformation of a quantised convolution in ONE launch --
//   [BN fold]  wf = w * sf[co];  b = (conv_bias - mean) * sf + beta              (quant_modules.py:365-372; the
//              per-channel factor sf = gamma / sqrt(var + eps) itself comes from the caller as two framework ops:
//              PyTorch-ROCm's sqrt is not the correctly rounded one, and sf must be the framework's bit for bit)
//   per output channel (symmetric, per_channel=True, no percentile):
//              mag = max(|min_k wf|, |max_k wf|); scale = n / clamp(mag, 1e-10)  (quant_utils.py:78-84)
//              q = clamp(round(scale * wf), -n-1, n); wq = q / scale             (:33-52, 207-225)
// The reference (and the module mirror) run this as ~15 framework ops per convolution and step, recomputed in every
// forward; three convolutions x three stages = the ~25 % of the QAT step that were PyTorch elementwise / reduction
// kernels (profiles/r02/train_step_kernel_stats.csv).  Every operation is rounded on its own exactly like the
// separate framework kernels (`n / t` = reciprocal * n: two roundings; true divisions; no FMA contraction), so wq and
// b are BIT-IDENTICAL to the torch composition (tests/test_train_step.py).  One wave per output channel.
// ------------------------------------------------------------------------------------------------------
constexpr int kPctMax = 4;
__device__ __forceinline__ void weight_prep_rows(int block, const float *__restrict__ w, int Co, int K, const float *__restrict__ sf_in,
                   const float *__restrict__ bn_b, const float *__restrict__ bn_mean,
                   const float *__restrict__ conv_bias, float nlev, float *__restrict__ wq,
                   float *__restrict__ b_out, int k_low, int k_high, float shrink) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int co = block * 4 + (threadIdx.x >> 6);      // block: this workgroup's index within the tensor
  if (co >= Co) return;
  float sf = 1.0f;
  const bool fold = sf_in != nullptr;
  if (fold) {
    sf = sf_in[co];
    if (lane == 0) {
      const float cb = conv_bias ? conv_bias[co] : 0.0f;
      const float t = cb - bn_mean[co];
      const float p = t * sf;
      b_out[co] = p + bn_b[co];
    }
  }
  const float *wr = w + (long)co * K;
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  if (k_low <= 1 && k_high <= 1) {
    for (int k = lane; k < K; k += 64) {
      const float v = fold ? wr[k] * sf : wr[k];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
      has_nan |= (v != v);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, m, 64));
      mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    }
    if (__any(has_nan)) mn = mx = __uint_as_float(0x7fc00000u);      // (torch's min() / max() of a row with a NaN)
  } else {
    // --wt-percentile (quant_modules.py:287-300): the k_low-th smallest and the k_high-th largest value of the channel
    // (torch.kthvalue counts duplicates).  k <= kPctMax: every lane keeps its kPctMax smallest / largest values sorted,
    // then the wave pops the global extreme k times (the owner of the popped value advances to its next one).
    float lo[kPctMax], hi[kPctMax];
#pragma unroll
    for (int i = 0; i < kPctMax; ++i) {
      lo[i] = INFINITY;
      hi[i] = -INFINITY;
    }
    for (int k = lane; k < K; k += 64) {
      float v = fold ? wr[k] * sf : wr[k], u = v;
#pragma unroll
      for (int i = 0; i < kPctMax; ++i) {       // insertion into the ascending lo[] / descending hi[]
        const float t = fminf(lo[i], v);
        v = fmaxf(lo[i], v);
        lo[i] = t;
        const float t2 = fmaxf(hi[i], u);
        u = fminf(hi[i], u);
        hi[i] = t2;
      }
    }
    for (int r = 0; r < k_low; ++r) {
      float best = lo[0];
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) best = fminf(best, __shfl_xor(best, m, 64));
      mn = best;
      // ONE owner of this value advances (the lowest lane holding it: duplicates are separate elements)
      const unsigned long long own = __ballot(lo[0] == best);
      if (lane == __ffsll((long long)own) - 1) {
#pragma unroll
        for (int i = 0; i + 1 < kPctMax; ++i) lo[i] = lo[i + 1];
        lo[kPctMax - 1] = INFINITY;
      }
    }
    for (int r = 0; r < k_high; ++r) {
      float best = hi[0];
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) best = fmaxf(best, __shfl_xor(best, m, 64));
      mx = best;
      const unsigned long long own = __ballot(hi[0] == best);
      if (lane == __ffsll((long long)own) - 1) {
#pragma unroll
        for (int i = 0; i + 1 < kPctMax; ++i) hi[i] = hi[i + 1];
        hi[kPctMax - 1] = -INFINITY;
      }
    }
  }
  mn = mn * shrink;        // (0.95 for channels of < 10 weights under --wt-percentile, else 1: exact)
  mx = mx * shrink;
  const float mag = fmaxf(fabsf(mn), fabsf(mx));
  const float rcp = __fdiv_rn(1.0f, fmaxf(mag, 1e-10f));
  const float scale = rcp * nlev;
  float *out = wq + (long)co * K;
  for (int k = lane; k < K; k += 64) {
    const float v = fold ? wr[k] * sf : wr[k];
    const float pq = scale * v;
    const float q = fminf(fmaxf(rintf(pq), -(nlev + 1.0f)), nlev);
    out[k] = __fdiv_rn(q, scale);
  }
}

__global__ void __launch_bounds__(256)
weight_prep_kernel(const float *__restrict__ w, int Co, int K, const float *__restrict__ sf_in,
                   const float *__restrict__ bn_b, const float *__restrict__ bn_mean,
                   const float *__restrict__ conv_bias, float nlev, float *__restrict__ wq,
                   float *__restrict__ b_out, int k_low, int k_high, float shrink) {
  weight_prep_rows((int)blockIdx.x, w, Co, K, sf_in, bn_b, bn_mean, conv_bias, nlev, wq, b_out, k_low, k_high, shrink);
}

// Several weight tensors in ONE launch (round 5: the conv_scale and depthwise weights of all three stages of a QAT step,
// and their three BN-folded pointwise weights -- launches of ~4.4 us, each at its launch floor): the workgroups are dealt
// out tensor by tensor, every row is prepared exactly as by weight_prep_kernel.
constexpr int kPrepMulti = 8;
struct PrepMulti {
  const float *w[kPrepMulti];
  const float *sf[kPrepMulti], *bn_b[kPrepMulti], *bn_mean[kPrepMulti], *conv_bias[kPrepMulti];      // BN fold (sf NULL: none)
  float *wq[kPrepMulti], *b_out[kPrepMulti];
  int Co[kPrepMulti], K[kPrepMulti], k_low[kPrepMulti], k_high[kPrepMulti], first_block[kPrepMulti + 1];
  float nlev[kPrepMulti], shrink[kPrepMulti];
  int n;
};
__global__ void __launch_bounds__(256)
weight_prep_multi_kernel(PrepMulti d) {
  int t = 0;
  while (t + 1 < d.n && (int)blockIdx.x >= d.first_block[t + 1]) ++t;
  weight_prep_rows((int)blockIdx.x - d.first_block[t], d.w[t], d.Co[t], d.K[t], d.sf[t], d.bn_b[t], d.bn_mean[t],
                   d.conv_bias[t], d.nlev[t], d.wq[t], d.b_out[t], d.k_low[t], d.k_high[t], d.shrink[t]);
}

}  // namespace

namespace {
// weight_prep_bwd_kernel: backward of the BN fold under a straight-through weight quantiser, one wave per output
// channel (the composition functions/codenet_stage.py::FoldFakeQuantWeight.backward spells out in nine framework ops):
//   grad_w[co][k] = g_wq[co][k] * sf[co]
//   grad_gamma[co] = (sum_k g_wq[co][k] * w[co][k] + g_b[co] * (conv_bias[co] - mean[co])) / std[co]
//   grad_beta[co] = g_b[co];   grad_conv_bias[co] = g_b[co] * sf[co]
__global__ void __launch_bounds__(256)
weight_prep_bwd_kernel(const float *__restrict__ g_wq, const float *__restrict__ g_b, const float *__restrict__ w,
                       const float *__restrict__ sf, const float *__restrict__ stdv, const float *__restrict__ mean,
                       const float *__restrict__ conv_bias, int Co, int K, float *__restrict__ g_w,
                       float *__restrict__ g_gamma, float *__restrict__ g_beta, float *__restrict__ g_cb) {
#pragma clang fp contract(off)
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (co >= Co) return;
  const float f = sf[co];
  const float *gp = g_wq + (long)co * K, *wp = w + (long)co * K;
  float dot = 0.0f;
  for (int k = lane; k < K; k += 64) {
    const float g = gp[k];
    dot += g * wp[k];
    if (g_w) g_w[(long)co * K + k] = g * f;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) dot += __shfl_xor(dot, m, 64);
  if (lane == 0) {
    const float gb = g_b ? g_b[co] : 0.0f;
    const float cbm = (conv_bias ? conv_bias[co] : 0.0f) - mean[co];
    if (g_gamma) g_gamma[co] = __fdiv_rn(dot + gb * cbm, stdv[co]);
    if (g_beta) g_beta[co] = gb;
    if (g_cb) g_cb[co] = gb * f;
  }
}
}  // namespace

extern "C" int cdn_codenet_weight_prep_backward(const float *grad_wq, const float *grad_bias, const float *w,
                                                const float *scale_factor, const float *bn_std, const float *bn_mean,
                                                const float *conv_bias, int64_t Co, int64_t K, float *grad_w,
                                                float *grad_gamma, float *grad_beta, float *grad_conv_bias,
                                                void *stream) {
  CDN_REQUIRE(grad_wq && w && scale_factor && bn_std && bn_mean, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(Co > 0 && K > 0 && Co * K < (1ll << 31), CDN_ERR_ARG, "bad size");
  weight_prep_bwd_kernel<<<(unsigned)cdn::ceil_div(Co, 4), 256, 0, cdn::as_stream(stream)>>>(
      grad_wq, grad_bias, w, scale_factor, bn_std, bn_mean, conv_bias, (int)Co, (int)K, grad_w, grad_gamma, grad_beta,
      grad_conv_bias);
  return cdn::check_launch("codenet weight prep backward");
}

static int weight_prep_impl(const float *w, int64_t Co, int64_t K, const float *scale_factor, const float *bn_bias,
                            const float *bn_mean, const float *conv_bias, int bits, int k_low, int k_high, float shrink,
                            float *w_q, float *bias_out, void *stream) {
  CDN_REQUIRE(w && w_q, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(Co > 0 && K > 0 && Co * K < (1ll << 31), CDN_ERR_ARG, "bad size");
  CDN_REQUIRE(bits >= 2 && bits <= 8, CDN_ERR_ARG, "bits must be in [2, 8]");
  CDN_REQUIRE(k_low >= 1 && k_high >= 1 && k_low <= K && k_high <= K, CDN_ERR_ARG, "k_low / k_high must be in [1, K]");
  CDN_REQUIRE(k_low <= kPctMax && k_high <= kPctMax, CDN_ERR_UNSUPPORTED,
              "percentile ranks above %d are not implemented (channels of more than %d weights)", kPctMax,
              kPctMax * 1000);
  const bool fold = scale_factor != nullptr;
  CDN_REQUIRE(!fold || (bn_bias && bn_mean && bias_out), CDN_ERR_ARG,
              "the BN fold needs scale_factor, bias, running_mean and bias_out together");
  hipStream_t st = cdn::as_stream(stream);
  const float nlev = (float)((1 << (bits - 1)) - 1);
  weight_prep_kernel<<<(unsigned)cdn::ceil_div(Co, 4), 256, 0, st>>>(w, (int)Co, (int)K, scale_factor, bn_bias, bn_mean,
                                                                     conv_bias, nlev, w_q, bias_out, k_low, k_high,
                                                                     shrink);
  return cdn::check_launch("codenet weight prep");
}

extern "C" int cdn_codenet_weight_prep_multi(int n, const float *const *w, const int64_t *Co, const int64_t *K,
                                             const float *const *scale_factor, const float *const *bn_bias,
                                             const float *const *bn_mean, const float *const *conv_bias, const int *bits,
                                             const int *k_low, const int *k_high, const float *shrink,
                                             float *const *w_q, float *const *bias_out, void *stream) {
  CDN_REQUIRE(n >= 1 && n <= kPrepMulti, CDN_ERR_ARG, "1 .. %d tensors per call", kPrepMulti);
  CDN_REQUIRE(w && Co && K && bits && k_low && k_high && shrink && w_q, CDN_ERR_ARG, "null pointer");
  PrepMulti d;
  d.n = n;
  int blocks = 0;
  for (int t = 0; t < n; ++t) {
    CDN_REQUIRE(w[t] && w_q[t] && Co[t] > 0 && K[t] > 0 && Co[t] * K[t] < (1ll << 31), CDN_ERR_ARG, "bad tensor %d", t);
    CDN_REQUIRE(bits[t] >= 2 && bits[t] <= 8, CDN_ERR_ARG, "bits must be in [2, 8]");
    CDN_REQUIRE(k_low[t] <= K[t] && k_high[t] <= K[t], CDN_ERR_ARG, "k_low / k_high must be in [1, K]");
    CDN_REQUIRE(k_low[t] >= 1 && k_high[t] >= 1 && k_low[t] <= kPctMax && k_high[t] <= kPctMax, CDN_ERR_UNSUPPORTED,
                "ranks of 1 .. %d", kPctMax);
    d.sf[t] = scale_factor ? scale_factor[t] : nullptr;
    d.bn_b[t] = (d.sf[t] && bn_bias) ? bn_bias[t] : nullptr;
    d.bn_mean[t] = (d.sf[t] && bn_mean) ? bn_mean[t] : nullptr;
    d.conv_bias[t] = (d.sf[t] && conv_bias) ? conv_bias[t] : nullptr;
    d.b_out[t] = (d.sf[t] && bias_out) ? bias_out[t] : nullptr;
    CDN_REQUIRE(!d.sf[t] || (d.bn_b[t] && d.bn_mean[t] && d.b_out[t]), CDN_ERR_ARG,
                "the BN fold needs scale_factor, bias, running_mean and bias_out together (tensor %d)", t);
    d.w[t] = w[t];
    d.wq[t] = w_q[t];
    d.Co[t] = (int)Co[t];
    d.K[t] = (int)K[t];
    d.k_low[t] = k_low[t];
    d.k_high[t] = k_high[t];
    d.shrink[t] = shrink[t];
    d.nlev[t] = (float)((1 << (bits[t] - 1)) - 1);
    d.first_block[t] = blocks;
    blocks += (int)cdn::ceil_div(Co[t], 4);
  }
  d.first_block[n] = blocks;
  weight_prep_multi_kernel<<<(unsigned)blocks, 256, 0, cdn::as_stream(stream)>>>(d);
  return cdn::check_launch("codenet weight prep (multi)");
}

extern "C" int cdn_codenet_weight_prep(const float *w, int64_t Co, int64_t K, const float *scale_factor,
                                       const float *bn_bias, const float *bn_mean, const float *conv_bias, int bits,
                                       float *w_q, float *bias_out, void *stream) {
  return weight_prep_impl(w, Co, K, scale_factor, bn_bias, bn_mean, conv_bias, bits, 1, 1, 1.0f, w_q, bias_out, stream);
}

extern "C" int cdn_codenet_weight_prep_ranked(const float *w, int64_t Co, int64_t K, const float *scale_factor,
                                              const float *bn_bias, const float *bn_mean, const float *conv_bias,
                                              int bits, int k_low, int k_high, float shrink, float *w_q,
                                              float *bias_out, void *stream) {
  return weight_prep_impl(w, Co, K, scale_factor, bn_bias, bn_mean, conv_bias, bits, k_low, k_high, shrink, w_q,
                          bias_out, stream);
}

namespace {

// ------------------------------------------------------------------------------------------------------
// pwi8n_kernel (round 5): the FORWARD conv_channel of the QAT step on the int8 matrix cores.
// In the QAT step both operands of y = conv1x1(QA_d(d), fq_w(w)) are integers in disguise: the activation is
// (L + zp) / scale with L the 8-bit code of the running QuantAct (quant_modules.py:203-225), the weight is q / ws with q
// the 4-bit per-channel symmetric code (quant_utils.py:207-225).  pointwise_kernel multiplies the two fp32 disguises on
// v_mfma_f32_32x32x2_f32 -- the fp32 VECTOR rate, 71 / 32 / 41 us at the step's three stages (4.3 GFLOP at 157 TFLOP/s
// is 27 us at stage 0 before anything else) -- where the inference schedule (pwi8_kernel, codenet_fused.hip) sums the
// integers exactly:  y = (sum_k (L_k + zp) q_k) / (scale ws) + b, one rounding instead of K.  This is that kernel for
// the training path's NCHW layout, where the MFMA operand layouts need no staging at all:
//   A  (rows = output channels)  the weight codes from a k-blocked copy [window][column][32 B] made by
//      pwn_wcodes_kernel below from the fake-quantised fp32 weights the autograd graph holds (codes = rint(w ws));
//   B  (columns = 32 consecutive pixels)  lane (j, h) loads d[c][p0 + j] for the 16 channels c = 32 t + 16 h + e of
//      window t -- sixteen dword loads, each a pair of full 128-byte lines per wave -- converts them to codes and packs
//      the two nibble operands in registers (pwi8_kernel's expressions); no LDS, no barrier in the k loop;
//   C  lane (j, h) holds pixel j of 4 x 4 output channels per tile: the stores are 128-byte lines of y[co][p].
// KS = 4: the four waves of a workgroup are the four quarters of K of one pixel block, int32 partial tiles added
// through LDS (exact); KS = 1: four pixel blocks.  Codes too wide for the nibble split (state[6]: the first steps of a
// fresh EMA) take the f32-MFMA form of the same loop on the fake-quantised values -- pointwise_kernel's arithmetic.
// ------------------------------------------------------------------------------------------------------
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

__global__ void __launch_bounds__(1024)
pwn_prep_kernel(const float *__restrict__ wq, int Co, int C, int Cot, int Ct, signed char *__restrict__ kb,
                float *__restrict__ wscale, int *__restrict__ wsum, unsigned short *__restrict__ wt,
                float *__restrict__ rws) {
  // Both integer forms of the weights in one launch: workgroup = sixteen output channels (one wave each).
  //   kb [window][column][32]  int8, k-blocked: the A operand of pwi8n_kernel (forward); zero columns behind Co;
  //   wt [co / 16][c][16]      bf16, transposed: the A operand of pwb3n_kernel (data gradient); zero rows behind C.
  // A row holds q / ws with |q| <= 8 (weight_bit <= 4); ws is recovered from the row itself, see below.
  extern __shared__ signed char prep_codes[];      // [16][C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int co = blockIdx.x * 16 + w;
  const float *row = wq + (long)co * C;
  // the row in registers when it fits (C <= 1024: sixteen values per lane): one pass over memory instead of three
  const bool cached = C <= 1024;
  float rv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) rv[u] = (cached && co < Co && lane + 64 * u < C) ? row[lane + 64 * u] : 0.0f;
  float mag = 0.0f;
  if (co < Co) {
    if (cached) {
#pragma unroll
      for (int u = 0; u < 16; ++u) mag = fmaxf(mag, fabsf(rv[u]));
    } else {
      for (int k = lane; k < C; k += 64) mag = fmaxf(mag, fabsf(row[k]));
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) mag = fmaxf(mag, __shfl_xor(mag, m, 64));
  // ws = m / max|w_q| with m the row's largest code magnitude: n = 2^(bits-1) - 1 (the channel's own extreme maps to +-n),
  // n + 1 when --wt-percentile clamps, so m is 1 ... 8 for the <= 4-bit weights this path admits.  The SMALLEST m that
  // makes every w ws an integer is taken: it is the true one divided by the codes' common factor -- the same real
  // values q / ws (a wrong m < the true one leaves a fractional part of a multiple of 1 / 8 or more: never accepted).
  // No m fits (the row is not a <= 4-bit grid): ws = NaN, the outputs of that channel are NaN -- loud, never wrong codes.
  float ws = 1.0f;
  if (co < Co && mag > 0.0f) {
    ws = __builtin_nanf("");
    for (int m = 8; m >= 1; --m) {
      const float cand = __fdiv_rn((float)m, mag);
      bool ok = true;
      if (cached) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float t = rv[u] * cand;
          ok &= fabsf(t - rintf(t)) < 0.02f;
        }
      } else {
        for (int k = lane; k < C; k += 64) {
          const float t = row[k] * cand;
          ok &= fabsf(t - rintf(t)) < 0.02f;
        }
      }
      if (__all(ok)) ws = cand;
    }
  }
  int sum = 0;
  auto emit = [&](int k, float v) {
    int q = 0;
    if (co < Co) q = min(max((int)rintf(v * ws), -8), 7);
    if (co < Cot) kb[((long)(k >> 5) * Cot + co) * 32 + (k & 31)] = (signed char)q;
    prep_codes[w * C + k] = (signed char)q;
    sum += q;
  };
  if (cached) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (lane + 64 * u < C) emit(lane + 64 * u, rv[u]);
  } else {
    for (int k = lane; k < C; k += 64) emit(k, co < Co ? row[k] : 0.0f);
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
  if (lane == 0) {
    if (co < Cot) {
      wscale[co] = ws;
      wsum[co] = sum;
    }
    rws[co] = co < Co ? __fdiv_rn(1.0f, ws) : 0.0f;
  }
  __syncthreads();
  if (blockIdx.x * 16 < ((Co + 15) & ~15)) {
    for (int c = threadIdx.x; c < Ct; c += 1024) {
      unsigned wpk[8];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float f0 = c < C ? (float)prep_codes[e * C + c] : 0.0f, f1 = c < C ? (float)prep_codes[(e + 1) * C + c] : 0.0f;
        wpk[e >> 1] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xFFFF0000u);      // small integers: exact in bf16
      }
      uint4 *dst = reinterpret_cast<uint4 *>(wt + ((long)blockIdx.x * Ct + c) * 16);
      dst[0] = make_uint4(wpk[0], wpk[1], wpk[2], wpk[3]);
      dst[1] = make_uint4(wpk[4], wpk[5], wpk[6], wpk[7]);
    }
  }
}

template <int TN, int KS>
__global__ void __launch_bounds__(256)
pwi8n_kernel(const float *__restrict__ D, const unsigned *__restrict__ dq, const signed char *__restrict__ Wkb,
             const float *__restrict__ wscale, const int *__restrict__ wsum, const float *__restrict__ Wq,
             const float *__restrict__ bias, float *__restrict__ Y, float2 *__restrict__ mm, int C, int Co, int HW,
             int ncg, cdn::QUpdate qu, int relu_range) {
  extern __shared__ float4 pwn_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int n = blockIdx.y;
  // (pixel-block group, column group) of this workgroup.  The column groups of one pixel block read the same rows of d:
  // 8 ids apart they land on one XCD under round-robin dispatch and the second reading is an L2 hit (round 6: adjacent
  // ids -- two XCDs -- made stage 0 read d twice from HBM: 97.7 MB per step for a 33.5-MB tensor, profiles/r05 PMC)
  int cg, pbg;
  {
    const int grp = 8 * ncg, b = blockIdx.x, full = ((int)gridDim.x / grp) * grp;
    if (b < full) {
      const int lid = b % grp;
      cg = lid >> 3;
      pbg = (b - lid) / ncg + (lid & 7);
    } else {
      cg = b % ncg;
      pbg = b / ncg;
    }
  }
  const int pb = KS == 4 ? pbg : pbg * 4 + w;                 // this wave's 32-pixel block
  const int npb = HW >> 5;
  const bool live = pb < npb;                                  // (KS = 1: the last workgroup of an image may be short)
  const int p0 = 32 * (live ? pb : 0);
  const int cb = cg * 32 * TN, Cot = 32 * TN * ncg;
  const int nwin = C >> 5;
  const int t0 = KS == 4 ? w * (nwin >> 2) + min(w, nwin & 3) : 0;
  const int nit = KS == 4 ? (nwin >> 2) + (w < (nwin & 3) ? 1 : 0) : nwin;
  const float qs = reinterpret_cast<const float *>(dq)[2], qz = reinterpret_cast<const float *>(dq)[3];
  const bool wide = dq[6] != 0;
  const float *Dn = D + ((long)n * C + 32 * t0) * HW + p0 + j;
  float mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;      // a NaN among the tracked values: fminf / fmaxf drop it, the reference's min() / max() do not
  // epilogue of one output: value from the reduced sum (int path: exact integer; wide path: fp32 sum), store, range
  auto finish = [&](float v, int co, int p) {
    Y[((long)n * Co + co) * HW + p] = v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    has_nan |= (v != v);
  };
  if (!wide) {
    const int ioff = (int)qz + (2048 - 128) - 0x4B400000;
    auto ucode = [&](float v) -> unsigned {
#pragma clang fp contract(off)
      const float y_p = qs * v;      // (two roundings, as the reference: quant_utils.py:33-41; see pwi8_kernel)
      const float y = (y_p - qz) + 12582912.0f;
      int u = (int)__float_as_uint(y) + ioff;
      u = min(max(u, 8), 4087);
      return (unsigned)u;
    };
    i32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = (i32x16){0};
    const signed char *wp = Wkb + (((long)t0 * Cot + cb + j) * 32 + 16 * h);
    float v[16], vn[16];
    i32x4 a[TN], an[TN];
    auto load = [&](float (&x)[16], i32x4 (&aw)[TN], int t) {
      const float *src = Dn + (long)(32 * t + 16 * h) * HW;
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = src[(long)e * HW];
#pragma unroll
      for (int q = 0; q < TN; ++q) aw[q] = *reinterpret_cast<const i32x4 *>(wp + ((long)t * Cot + 32 * q) * 32);
    };
    if (nit > 0) load(v, a, 0);
    for (int t = 0; t < nit; ++t) {
      if (t + 1 < nit) load(vn, an, t + 1);
      i32x4 blo, bhi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned u0 = ucode(v[4 * q]), u1 = ucode(v[4 * q + 1]), u2 = ucode(v[4 * q + 2]), u3 = ucode(v[4 * q + 3]);
        const unsigned p01 = u0 | (u1 << 16), p23 = u2 | (u3 << 16);
        blo[q] = (int)(__builtin_amdgcn_perm(p23, p01, 0x06040200u) & 0x0F0F0F0Fu);
        bhi[q] = (int)(__builtin_amdgcn_perm(p23 >> 4, p01 >> 4, 0x06040200u) ^ 0x80808080u);
      }
#pragma unroll
      for (int q = 0; q < TN; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q], blo, acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        i32x4 a16;
#pragma unroll
        for (int e = 0; e < 4; ++e) a16[e] = (int)(((unsigned)a[q][e] << 4) & 0xF0F0F0F0u);
        acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, bhi, acc[q], 0, 0, 0);
      }
      if (t + 1 < nit) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = vn[e];
#pragma unroll
        for (int q = 0; q < TN; ++q) a[q] = an[q];
      }
    }
    if (KS == 1) {
      if (live) {
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cb + 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < Co) {
              const float rinv = __fdiv_rn(1.0f, __fmul_rn(qs, wscale[co]));
              finish(fmaf((float)(acc[q][r] + 128 * wsum[co]), rinv, bias ? bias[co] : 0.0f), co, p0 + j);
            }
          }
      }
    } else {
      // P[wave][tile][row quad 2g + h][pixel] of int4: a lane's four consecutive accumulator registers are four
      // consecutive output channels of its pixel
      i32x4 *P = reinterpret_cast<i32x4 *>(pwn_lds);
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          P[((w * TN + q) * 8 + 2 * g + h) * 32 + j] = (i32x4){acc[q][4 * g], acc[q][4 * g + 1], acc[q][4 * g + 2], acc[q][4 * g + 3]};
      __syncthreads();
      const int px = tid & 31, qd = tid >> 5;
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const i32x4 s = (P[((0 * TN + q) * 8 + qd) * 32 + px] + P[((1 * TN + q) * 8 + qd) * 32 + px]) +
                        (P[((2 * TN + q) * 8 + qd) * 32 + px] + P[((3 * TN + q) * 8 + qd) * 32 + px]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = cb + 32 * q + 4 * qd + e;
          if (co < Co) {
            const float rinv = __fdiv_rn(1.0f, __fmul_rn(qs, wscale[co]));
            finish(fmaf((float)(s[e] + 128 * wsum[co]), rinv, bias ? bias[co] : 0.0f), co, p0 + px);
          }
        }
      }
    }
  } else {
    // wide codes: f32 MFMA on the fake-quantised values (A[i = co][k = h], B[k = h][j = pixel] of 32x32x2)
    const float qr = __fdiv_rn(1.0f, qs);
    f32x16 accf[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) accf[t] = (f32x16){0};
    for (int k = 0; k < 32 * nit; k += 2) {
      const float bv = cdn::fake_quant_r(Dn[(long)(k + h) * HW], qs, qz, qr);
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const int co = min(cb + 32 * q + j, Co - 1);
        accf[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wq[(long)co * C + 32 * t0 + k + h], bv, accf[q], 0, 0, 0);
      }
    }
    if (KS == 1) {
      if (live) {
#pragma unroll
        for (int q = 0; q < TN; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cb + 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < Co) finish(accf[q][r] + (bias ? bias[co] : 0.0f), co, p0 + j);
          }
      }
    } else {
      float4 *P = reinterpret_cast<float4 *>(pwn_lds);
#pragma unroll
      for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          P[((w * TN + q) * 8 + 2 * g + h) * 32 + j] = make_float4(accf[q][4 * g], accf[q][4 * g + 1], accf[q][4 * g + 2], accf[q][4 * g + 3]);
      __syncthreads();
      const int px = tid & 31, qd = tid >> 5;
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const float4 s0 = P[((0 * TN + q) * 8 + qd) * 32 + px], s1 = P[((1 * TN + q) * 8 + qd) * 32 + px];
        const float4 s2 = P[((2 * TN + q) * 8 + qd) * 32 + px], s3 = P[((3 * TN + q) * 8 + qd) * 32 + px];
        const float sv[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                             (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = cb + 32 * q + 4 * qd + e;
          if (co < Co) finish(sv[e] + (bias ? bias[co] : 0.0f), co, p0 + px);
        }
      }
    }
  }
  if (qu.counters) {
    // round 6: the QuantAct behind y updated by the last workgroup; relu_range: it sits behind a ReLU (the block after a
    // stage) and tracks the extremes of max(y, 0)
    if (relu_range) {
      mn = cdn::relu_keep_nan(mn);
      mx = cdn::relu_keep_nan(mx);
    }
    __syncthreads();
    cdn::block_minmax_finish(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), nullptr, blockIdx.y * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y, qu, reinterpret_cast<float *>(pwn_lds));
  } else if (mm) {
    __syncthreads();
    cdn::block_minmax_store(cdn::nan_lo(mn, has_nan), cdn::nan_hi(mx, has_nan), &mm[blockIdx.y * gridDim.x + blockIdx.x],
                            reinterpret_cast<float *>(pwn_lds));
  }
}

// ------------------------------------------------------------------------------------------------------
// pwb3n_kernel (round 5): the DATA GRADIENT of conv_channel in the QAT step,
//   grad_d[n][c][p] = sum_co w_q[co][c] grad_y[n][co][p],   w_q = q / ws (4-bit codes),
// on the bf16 matrix cores with EXACT products (pwb3_kernel's split, codenet_fused.hip): g' = grad_y / ws[co] is cut
// into three bf16 terms by truncation (8 + 8 + 8 significant bits: hi + mid + lo == g' exactly), the codes are exact in
// bf16, every product is exact in fp32 and the accumulation is fp32 -- 3 v_mfma_f32_32x32x16_bf16 per 16 k where
// pointwise_kernel issues 8 v_mfma_f32_32x32x2_f32 (39 / 33 / 57 us at the step's three stages, matrix-core bound).
// NCHW needs no staging: A (rows = input channels c) from a transposed bf16 copy of the codes [co / 16][c][16]
// (pwn_prep_kernel, made with the forward's codes), B (columns = 32 pixels) as eight dword loads of grad_y per lane and 16-k step, split in registers;
// a wave owns 32 pixels x 32 TN channels over the whole K = Co; lanes run along pixels in loads and stores.
// Differs from the f32 kernel by rounding noise only (g' is rounded once, the f32 kernel rounds q / ws and the product).
// ------------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int TN>
__global__ void __launch_bounds__(256)
pwb3n_kernel(const float *__restrict__ GY, const unsigned short *__restrict__ WT, const float *__restrict__ rws,
             float *__restrict__ GD, int C, int Co, int HW, int Ct, int ncg) {
  __shared__ float rw[512];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int n = blockIdx.y;
  const int S = (Co + 15) >> 4;
  for (int q = tid; q < 16 * S; q += 256) rw[q] = rws[q];
  __syncthreads();
  // workgroup = 4 waves: consecutive (pixel block, channel group) pairs, channel group fastest (they share grad_y lines)
  const int item = blockIdx.x * 4 + w;
  const int cg = item % ncg, pb = item / ncg;
  if (pb >= (HW >> 5)) return;
  const int p0 = 32 * pb, cb = cg * 32 * TN;
  const float *gyp = GY + (long)n * Co * HW + p0 + j;
  const unsigned short *wtp = WT + ((long)cb + j) * 16 + 8 * h;
  auto pack_hi = [](unsigned a_, unsigned b_) -> unsigned { return __builtin_amdgcn_perm(a_, b_, 0x07060302u); };
  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = (f32x16){0};
  float g[8], gn[8];
  i32x4 a[TN], an[TN];
  auto load = [&](float (&x)[8], i32x4 (&aw)[TN], int s16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = 16 * s16 + 8 * h + e;
      x[e] = gyp[(long)min(co, Co - 1) * HW];            // (co >= Co: multiplied by rw = 0 below)
    }
#pragma unroll
    for (int t = 0; t < TN; ++t)
      aw[t] = *reinterpret_cast<const i32x4 *>(wtp + ((long)s16 * Ct + 32 * t) * 16);
  };
  load(g, a, 0);
  for (int s16 = 0; s16 < S; ++s16) {
    if (s16 + 1 < S) load(gn, an, s16 + 1);
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = __fmul_rn(g[e], rw[16 * s16 + 8 * h + e]);
      hb[e] = __float_as_uint(x);
      const float r1 = __fsub_rn(x, __uint_as_float(hb[e] & 0xFFFF0000u));
      mb[e] = __float_as_uint(r1);
      lb[e] = __float_as_uint(__fsub_rn(r1, __uint_as_float(mb[e] & 0xFFFF0000u)));
    }
    i32x4 ph, pm, pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ph[e] = (int)pack_hi(hb[2 * e + 1], hb[2 * e]);
      pm[e] = (int)pack_hi(mb[2 * e + 1], mb[2 * e]);
      pl[e] = (int)pack_hi(lb[2 * e + 1], lb[2 * e]);
    }
    const bf16x8 fh = __builtin_bit_cast(bf16x8, ph), fm = __builtin_bit_cast(bf16x8, pm), fl = __builtin_bit_cast(bf16x8, pl);
    // lo, mid, hi into every accumulator, interleaved over the tiles (three MFMAs into one accumulator back to back stall)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), fl, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), fm, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), fh, acc[t], 0, 0, 0);
    if (s16 + 1 < S) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = gn[e];
#pragma unroll
      for (int t = 0; t < TN; ++t) a[t] = an[t];
    }
  }
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = cb + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < C) GD[((long)n * C + c) * HW + p0 + j] = acc[t][r];
    }
}

struct PwnPlan {
  int tn, ncg, ks, cot;
  unsigned grid_x;
  size_t lds, off_scale, off_sum, bytes;
  int dtn, dncg, ct;               // data gradient (pwb3n_kernel): channel tiles per wave, channel groups, padded C
  size_t off_wt, off_rws;
};
static bool pwn_plan(int64_t N, int64_t C, int64_t Co, int64_t HW, PwnPlan *p) {
  // C <= 4096: pwn_prep_kernel keeps sixteen code rows of C bytes in LDS (64 KB)
  if (N <= 0 || N > 65535 || C < 32 || (C & 31) || C > 4096 || HW < 32 || (HW & 31) || Co < 1 || Co > 512 ||
      C * HW >= (1ll << 31) || Co * HW >= (1ll << 31))
    return false;
  p->tn = Co <= 64 ? 2 : 4;
  p->ncg = (int)cdn::ceil_div(Co, 32 * p->tn);
  p->cot = 32 * p->tn * p->ncg;
  const long npb = HW >> 5;
  // K split over the four waves of a workgroup while whole-K waves would leave the chip short of waves (stage 0:
  // 256 pixel blocks x 2 column groups at batch 32)
  p->ks = (N * npb * p->ncg < 4096 && (C >> 5) >= 8) ? 4 : 1;
  p->grid_x = (unsigned)((p->ks == 4 ? npb : cdn::ceil_div(npb, 4)) * p->ncg);
  p->lds = p->ks == 4 ? (size_t)4 * p->tn * 8 * 32 * 16 : 256;
  const size_t codes = ((size_t)C * p->cot + 255) / 256 * 256;
  p->off_scale = codes;
  p->off_sum = codes + ((size_t)p->cot * 4 + 255) / 256 * 256;
  p->dtn = C <= 128 ? 4 : 8;
  p->dncg = (int)cdn::ceil_div(C, 32 * p->dtn);
  p->ct = 32 * p->dtn * p->dncg;
  const size_t S = (size_t)(std::max<int64_t>(Co, p->cot) + 15) / 16;      // (the prep kernel's workgroups: 16 columns each)
  p->off_wt = p->off_sum + ((size_t)p->cot * 4 + 255) / 256 * 256;
  p->off_rws = p->off_wt + (S * p->ct * 32 + 255) / 256 * 256;
  p->bytes = p->off_rws + (S * 16 * 4 + 255) / 256 * 256;
  return true;
}

}  // namespace

extern "C" int cdn_codenet_pointwise_i8_supported(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  PwnPlan p;
  return pwn_plan(N, C, Co, HW, &p) ? 1 : 0;
}
extern "C" size_t cdn_codenet_pointwise_i8_workspace_bytes(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  PwnPlan p;
  return pwn_plan(N, C, Co, HW, &p) ? p.bytes : 0;
}
extern "C" int64_t cdn_codenet_pointwise_i8_range_partials(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  PwnPlan p;
  return pwn_plan(N, C, Co, HW, &p) ? (int64_t)p.grid_x * N : 0;
}

static int pointwise_i8_forward_impl(const float *d, const void *d_state, const float *w_q, const float *bias, float *y,
                                     int64_t N, int64_t C, int64_t Co, int64_t HW, float *partials, void *workspace,
                                     size_t workspace_bytes, void *stream, const cdn::QUpdate &qu, int relu_range);

extern "C" int cdn_codenet_pointwise_i8_forward_range(const float *d, const void *d_state, const float *w_q,
                                                      const float *bias, float *y, int64_t N, int64_t C, int64_t Co,
                                                      int64_t HW, float *partials, void *workspace,
                                                      size_t workspace_bytes, void *stream) {
  const cdn::QUpdate none{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 8, 0};
  return pointwise_i8_forward_impl(d, d_state, w_q, bias, y, N, C, Co, HW, partials, workspace, workspace_bytes, stream,
                                   none, 0);
}

// The same launch with the QuantAct BEHIND y updated by its last workgroup (round 6; see cdn_codenet_scale_forward_update):
// relu_range != 0: that QuantAct sits behind a ReLU and tracks the extremes of max(y, 0) (the block after a stage).
extern "C" int cdn_codenet_pointwise_i8_forward_update(const float *d, const void *d_state, const float *w_q,
                                                       const float *bias, float *y, int64_t N, int64_t C, int64_t Co,
                                                       int64_t HW, void *workspace, size_t workspace_bytes,
                                                       int relu_range, float *x_min, float *x_max, void *state,
                                                       void *counters, int bits, double momentum, void *stream) {
  CDN_REQUIRE(x_min && x_max && state && counters, CDN_ERR_ARG, "null QuantAct pointer");
  CDN_REQUIRE(bits >= 2 && bits <= 16, CDN_ERR_ARG, "bits must be in [2,16], got %d", bits);
  const cdn::QUpdate qu{x_min, x_max, static_cast<unsigned *>(state), static_cast<unsigned *>(counters),
                        (float)(momentum - 1.0), (float)(1.0 - momentum), bits, 1};
  return pointwise_i8_forward_impl(d, d_state, w_q, bias, y, N, C, Co, HW, nullptr, workspace, workspace_bytes, stream,
                                   qu, relu_range);
}

static int pointwise_i8_forward_impl(const float *d, const void *d_state, const float *w_q, const float *bias, float *y,
                                     int64_t N, int64_t C, int64_t Co, int64_t HW, float *partials, void *workspace,
                                     size_t workspace_bytes, void *stream, const cdn::QUpdate &qu, int relu_range) {
  CDN_REQUIRE(d && d_state && w_q && y && workspace, CDN_ERR_ARG, "null pointer");
  PwnPlan p;
  CDN_REQUIRE(pwn_plan(N, C, Co, HW, &p), CDN_ERR_UNSUPPORTED,
              "int8 forward needs C %% 32 == 0, C <= 4096, HW %% 32 == 0, Co <= 512 (cdn_codenet_pointwise_i8_supported)");
  CDN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0 && workspace_bytes >= p.bytes, CDN_ERR_WORKSPACE,
              "workspace missing, too small or not 256-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  signed char *kb = static_cast<signed char *>(workspace);
  float *wscale = reinterpret_cast<float *>(kb + p.off_scale);
  int *wsum = reinterpret_cast<int *>(kb + p.off_sum);
  unsigned short *wt = reinterpret_cast<unsigned short *>(kb + p.off_wt);
  float *rws = reinterpret_cast<float *>(kb + p.off_rws);
  if (16 * C > 48 * 1024)
    (void)hipFuncSetAttribute((const void *)pwn_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(16 * C));
  pwn_prep_kernel<<<(unsigned)(p.cot / 16), 1024, (size_t)16 * C, st>>>(w_q, (int)Co, (int)C, p.cot, p.ct, kb, wscale, wsum,
                                                                        wt, rws);
  int rc = cdn::check_launch("codenet weight codes");
  if (rc) return rc;
  dim3 grid(p.grid_x, (unsigned)N);
  float2 *mm = reinterpret_cast<float2 *>(partials);
  const unsigned *dq = static_cast<const unsigned *>(d_state);
#define CDN_PWN(TN_, KS_)                                                                                           \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void *)pwi8n_kernel<TN_, KS_>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)p.lds);                                                                          \
    pwi8n_kernel<TN_, KS_><<<grid, 256, p.lds, st>>>(d, dq, kb, wscale, wsum, w_q, bias, y, mm, (int)C, (int)Co,    \
                                                     (int)HW, p.ncg, qu, relu_range);                               \
  } while (0)
  if (p.tn == 4 && p.ks == 4) CDN_PWN(4, 4);
  else if (p.tn == 4) CDN_PWN(4, 1);
  else if (p.ks == 4) CDN_PWN(2, 4);
  else CDN_PWN(2, 1);
#undef CDN_PWN
  return cdn::check_launch("codenet int8 pointwise forward (NCHW)");
}

// ---- data gradient of conv_channel on bf16 MFMA (pwb3n_kernel) ----------------------------------------------------------
extern "C" int cdn_codenet_pointwise_dgrad_q4_supported(int64_t N, int64_t C, int64_t Co, int64_t HW) {
  PwnPlan p;
  return pwn_plan(N, C, Co, HW, &p) ? 1 : 0;
}
extern "C" int cdn_codenet_pointwise_dgrad_q4(const float *grad_y, const void *fwd_workspace, float *grad_d, int64_t N,
                                              int64_t C, int64_t Co, int64_t HW, void *stream) {
  CDN_REQUIRE(grad_y && fwd_workspace && grad_d, CDN_ERR_ARG, "null pointer");
  PwnPlan p;
  CDN_REQUIRE(pwn_plan(N, C, Co, HW, &p), CDN_ERR_UNSUPPORTED,
              "fwd_workspace is that of cdn_codenet_pointwise_i8_forward_range: the shape must be supported there");
  hipStream_t st = cdn::as_stream(stream);
  const char *base = static_cast<const char *>(fwd_workspace);
  const unsigned short *wt = reinterpret_cast<const unsigned short *>(base + p.off_wt);
  const float *rws = reinterpret_cast<const float *>(base + p.off_rws);
  const long items = (HW >> 5) * p.dncg;
  dim3 grid((unsigned)cdn::ceil_div(items, 4), (unsigned)N);
  if (p.dtn == 8) pwb3n_kernel<8><<<grid, 256, 0, st>>>(grad_y, wt, rws, grad_d, (int)C, (int)Co, (int)HW, p.ct, p.dncg);
  else pwb3n_kernel<4><<<grid, 256, 0, st>>>(grad_y, wt, rws, grad_d, (int)C, (int)Co, (int)HW, p.ct, p.dncg);
  return cdn::check_launch("codenet pointwise data gradient (bf16 x 3)");
}
