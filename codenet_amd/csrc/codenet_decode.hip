// codenet_decode.hip -- ctdet_decode as two kernels.
//
// Reference (lib/models/decode.py:10-16 _nms, :110-127 _topk, :474-505 ctdet_decode; lib/models/
// utils.py _gather_feat / _transpose_and_gather_feat):
//     heat = heat * (max_pool2d(heat, 3, 1, 1) == heat)              3x3 peak filter, non-peaks -> 0
//     per class top-K of heat.view(B, cat, H*W), then top-K of the cat*K survivors
//     xs, ys = index % W, index / W (+ reg at that pixel, or + 0.5); wh at that pixel
//     dets[b][k] = [xs - w/2, ys - h/2, xs + w/2, ys + h/2, score, class]
// as ~25 framework launches incl. two sorts.  The two-level top-K equals the global top-K of the
// masked map (the global top K can hold at most K entries of one class), so here:
//   decode_keys_kernel   one workgroup per (image, class, row band): band -> LDS, (sigmoid,) 3x3 peak test,
//                        32-bit order-preserving key of the masked score of every pixel -> per-image histogram
//                        of the keys' top 11 bits, and (round 4) the CANDIDATES -- pixels whose masked score is
//                        above zero, i.e. the peaks -- appended as (key, inverted index) pairs to a per-image
//                        list (count, ONE global atomic per workgroup to reserve a range, write): ~11 % of the
//                        pixels of a heat map instead of a 4-byte key for every pixel (84 MB at batch 64)
//   decode_select_kernel one workgroup per image: radix refinement of the K-th key (11 + 11 + 10 bits,
//                        re-histogramming only the threshold group), one collect pass, bitonic sort of
//                        the <= 4096 collected (key, index) pairs in LDS, gather of reg / wh, boxes -- over the
//                        candidate list when it holds at least K entries (then the K-th key is above zero and
//                        nothing outside the list can be selected).  Otherwise (fewer than K peaks: zeros fill
//                        up by ascending index; a list that overflowed; more equal keys at the threshold than
//                        the LDS list holds) the same algorithm runs over keys recomputed from the heat map
//                        pixel by pixel: exact for every input, slow, and not met by a detector's heat maps.
// Ties (equal scores) are ordered by ascending flat index class*H*W + y*W + x; torch.topk leaves that
// order unspecified (lib/models/decode.py:114,120), the oracle states the same rule.
#include "cdn_common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kSelThreads = 1024;
constexpr int kCap = 4096;          // capacity of the LDS candidate list
constexpr int kBins = 2048;         // level-0 histogram: top 11 bits of the key
constexpr int kKeyThreads = 512;

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// Words per image in the histogram region: kBins histogram bins + the candidate counter (+ padding).
// Words per image in the histogram region: kBins histogram bins + the candidate counter (+ padding).
constexpr int kHistStride = kBins + 16;

__device__ __forceinline__ unsigned masked_key(float v, float m9) {
  // v = the pixel, m9 = the maximum of its 3x3 neighbourhood INCLUDING itself:
  // hmax == heat  <=>  fmaxf(m8, v) == v  <=>  m9 == v (NaN compares false, as in the reference)
  const float s = (m9 == v) ? v : v * 0.0f;
  return cdn::f2ord(s + 0.0f);                  // (+0.0f: -0 and +0 get the same key)
}

// Masked keys of pixel quad q (4 consecutive pixels of a row; W % 4 == 0) of the band held in `plane`
// ([(rows + 2)][Wp], interior from column 4, row 0 = image row y0 - 1).
__device__ __forceinline__ uint4 quad_keys(const float *plane, int Wp, int y0, int W, int q) {
  const int p = q * 4, y = p / W, x = p - y * W;
  const float *ctr = plane + (y - y0 + 1) * Wp + 4 + x;
  float hm[3][4];                           // horizontal 3-maxima of the three rows
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float *row = ctr + (r - 1) * Wp;
    const float4 cv = *reinterpret_cast<const float4 *>(row);
    const float l = row[-1], rr = row[4];
    hm[r][0] = fmaxf(fmaxf(l, cv.x), cv.y);
    hm[r][1] = fmaxf(fmaxf(cv.x, cv.y), cv.z);
    hm[r][2] = fmaxf(fmaxf(cv.y, cv.z), cv.w);
    hm[r][3] = fmaxf(fmaxf(cv.z, cv.w), rr);
  }
  const float4 cv = *reinterpret_cast<const float4 *>(ctr);
  uint4 k4;
  k4.x = masked_key(cv.x, fmaxf(fmaxf(hm[0][0], hm[1][0]), hm[2][0]));
  k4.y = masked_key(cv.y, fmaxf(fmaxf(hm[0][1], hm[1][1]), hm[2][1]));
  k4.z = masked_key(cv.z, fmaxf(fmaxf(hm[0][2], hm[1][2]), hm[2][2]));
  k4.w = masked_key(cv.w, fmaxf(fmaxf(hm[0][3], hm[1][3]), hm[2][3]));
  return k4;
}
__device__ __forceinline__ unsigned pixel_key(const float *plane, int Wp, int y0, int W, int p) {
  const int y = p / W, x = p - y * W;
  const float *q = plane + (y - y0 + 1) * Wp + 4 + x;
  float m = fmaxf(fmaxf(q[-Wp - 1], q[-Wp]), fmaxf(q[-Wp + 1], q[-1]));
  m = fmaxf(m, fmaxf(fmaxf(q[1], q[Wp - 1]), fmaxf(q[Wp], q[Wp + 1])));
  return masked_key(q[0], fmaxf(m, q[0]));
}

__global__ void __launch_bounds__(kKeyThreads)
decode_keys_kernel(const float *heat, unsigned long long *__restrict__ cand, unsigned *__restrict__ hist,
                   float *heat_out, int cat, int H, int W, int apply_sigmoid, int RB, unsigned cap) {
  // heat_out may alias heat ONLY when a workgroup owns whole planes (one band) or no sigmoid is applied
  // (then the store rewrites the value it read); the host entry routes the other in-place case through
  // sigmoid_store_kernel instead, because a neighbouring band's halo rows would otherwise be read after
  // this band has overwritten them with their sigmoid (no __restrict__ on the two pointers).
  // One workgroup = (class plane, band of RB rows).  LDS: [(RB + 2)][Wp] with a -inf border where the image
  // ends (band halo rows are loaded); Wp = W + 8 and the interior starts at column 4, so that every interior
  // row is 16-byte aligned: a thread handles 4 consecutive pixels with 3 ds_read_b128 + 6 ds_read_b32
  // (9 scalar reads per PIXEL, each waited for, made this phase 440 cycles per pixel and wave in the ISA)
  extern __shared__ __attribute__((aligned(16))) float plane[];
  __shared__ unsigned lh[kBins];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int y0 = blockIdx.z * RB, y1 = min(H, y0 + RB), nr = y1 - y0;      // own rows [y0, y1)
  const int HW = H * W, Wp = ((W + 3) & ~3) + 8;
  const float *hp = heat + ((long)b * cat + c) * HW;
  float *op = heat_out ? heat_out + ((long)b * cat + c) * HW : nullptr;
  for (int i = tid; i < kBins; i += kKeyThreads) lh[i] = 0;
  // LDS row r holds image row y0 - 1 + r.  -inf: columns -1 and W of every row; whole rows outside the image
  for (int i = tid; i < 2 * (nr + 2); i += kKeyThreads)
    plane[(i >> 1) * Wp + ((i & 1) ? 4 + W : 3)] = -INFINITY;
  if (y0 == 0)
    for (int i = tid; i < W; i += kKeyThreads) plane[4 + i] = -INFINITY;
  if (y1 == H)
    for (int i = tid; i < W; i += kKeyThreads) plane[(nr + 1) * Wp + 4 + i] = -INFINITY;
  const int ylo = max(y0 - 1, 0), yhi = min(y1 + 1, H);                    // rows to load
  const bool vec = (W & 3) == 0;
  if (vec) {                                    // 16-byte loads, 8 in flight per thread
    const int q_lo = (ylo * W) >> 2, q_hi = (yhi * W) >> 2;
    for (int base = q_lo; base < q_hi; base += kKeyThreads * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = base + u * kKeyThreads + tid;
        v[u] = q < q_hi ? reinterpret_cast<const float4 *>(hp)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = base + u * kKeyThreads + tid;
        if (q < q_hi) {
          float4 t = v[u];
          if (apply_sigmoid) {
            t.x = sigmoidf_ref(t.x); t.y = sigmoidf_ref(t.y); t.z = sigmoidf_ref(t.z); t.w = sigmoidf_ref(t.w);
          }
          const int p = q * 4, y = p / W, x = p - y * W;
          *reinterpret_cast<float4 *>(plane + (y - y0 + 1) * Wp + 4 + x) = t;
          if (op && y >= y0 && y < y1) reinterpret_cast<float4 *>(op)[q] = t;   // halo rows: their own band
        }
      }
    }
  } else {
    for (int p = ylo * W + tid; p < yhi * W; p += kKeyThreads) {
      float v = hp[p];
      if (apply_sigmoid) v = sigmoidf_ref(v);
      const int y = p / W;
      plane[(y - y0 + 1) * Wp + 4 + p % W] = v;
      if (op && y >= y0 && y < y1) op[p] = v;
    }
  }
  __syncthreads();
  // non-peaks all carry the key of 0.0: counted per thread, ONE histogram atomic per thread at the end
  // (16384 same-address LDS atomics per plane cost 250 us per launch).  Plain locals and straight-line code:
  // counters captured by reference in closures ended up in scratch memory (a round trip per increment, 20 us).
  const unsigned zkey = cdn::f2ord(0.0f);
  unsigned zeros = 0, mine = 0;
#define CDN_TALLY(k_)                                   \
  do {                                                  \
    const unsigned kk_ = (k_);                          \
    if (kk_ == zkey) {                                  \
      ++zeros;                                          \
    } else {                                            \
      atomicAdd(&lh[kk_ >> (32 - 11)], 1u);             \
      mine += kk_ > zkey ? 1u : 0u;                     \
    }                                                   \
  } while (0)
  // Usual shape (the band's pixel quads fit kKeepQ per thread): the keys stay in registers between the counting
  // and the writing pass; otherwise they are recomputed from the LDS plane.
  constexpr int kKeepQ = 4;
  const int q_first = (y0 * W) >> 2, q_end = (y1 * W) >> 2;
  const bool keep = vec && (q_end - q_first) <= kKeepQ * kKeyThreads;
  uint4 kept[kKeepQ];
  if (keep) {
#pragma unroll
    for (int u = 0; u < kKeepQ; ++u) {
      const int q = q_first + tid + u * kKeyThreads;
      kept[u] = make_uint4(zkey, zkey, zkey, zkey);
      if (q < q_end) {
        const uint4 k4 = quad_keys(plane, Wp, y0, W, q);
        CDN_TALLY(k4.x); CDN_TALLY(k4.y); CDN_TALLY(k4.z); CDN_TALLY(k4.w);
        kept[u] = k4;
      }
    }
  } else if (vec) {
    for (int q = q_first + tid; q < q_end; q += kKeyThreads) {
      const uint4 k4 = quad_keys(plane, Wp, y0, W, q);
      CDN_TALLY(k4.x); CDN_TALLY(k4.y); CDN_TALLY(k4.z); CDN_TALLY(k4.w);
    }
  } else {
    for (int p = y0 * W + tid; p < y1 * W; p += kKeyThreads) CDN_TALLY(pixel_key(plane, Wp, y0, W, p));
  }
#undef CDN_TALLY
  if (zeros) atomicAdd(&lh[zkey >> (32 - 11)], zeros);
  // Every WAVE reserves a range of the image's list for its candidates: prefix sum of the lanes' counts, one global
  // atomic by the last lane, no barrier and no LDS atomic (512 same-address LDS atomics with a return value per
  // workgroup cost 13 us per launch, a workgroup-wide reservation behind a barrier 20 us; the other waves of the CU
  // cover a wave's round trip).  Order inside the list does not matter.
  unsigned incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(incl, d, 64);
    if ((tid & 63) >= d) incl += t;
  }
  unsigned *gh = hist + (long)b * kHistStride;
  unsigned wbase = 0u;
  if ((tid & 63) == 63 && incl) wbase = atomicAdd(&gh[kBins], incl);
  wbase = __shfl(wbase, 63, 64);
  unsigned long long *lp = cand + (size_t)b * cap;
  const unsigned pix0 = (unsigned)c * (unsigned)HW;
  unsigned slot = wbase + incl - mine;
#define CDN_PUT(k_, p_)                                                                                         \
  do {                                                                                                          \
    const unsigned kk_ = (k_);                                                                                  \
    if (kk_ > zkey) {                                                                                           \
      if (slot < cap)                                                                                           \
        lp[slot] = ((unsigned long long)kk_ << 32) | (unsigned long long)(0xFFFFFFFFu - (pix0 + (unsigned)(p_))); \
      ++slot;                                                                                                   \
    }                                                                                                           \
  } while (0)
  if (keep) {
#pragma unroll
    for (int u = 0; u < kKeepQ; ++u) {
      const int p = (q_first + tid + u * kKeyThreads) * 4;        // (quads past the band hold zkey: skipped)
      CDN_PUT(kept[u].x, p); CDN_PUT(kept[u].y, p + 1); CDN_PUT(kept[u].z, p + 2); CDN_PUT(kept[u].w, p + 3);
    }
  } else if (vec) {                                               // recomputing sweep: same pixels, same order
    for (int q = q_first + tid; q < q_end; q += kKeyThreads) {
      const uint4 k4 = quad_keys(plane, Wp, y0, W, q);
      CDN_PUT(k4.x, q * 4); CDN_PUT(k4.y, q * 4 + 1); CDN_PUT(k4.z, q * 4 + 2); CDN_PUT(k4.w, q * 4 + 3);
    }
  } else {
    for (int p = y0 * W + tid; p < y1 * W; p += kKeyThreads) CDN_PUT(pixel_key(plane, Wp, y0, W, p), p);
  }
#undef CDN_PUT
  __syncthreads();
  for (int i = tid; i < kBins; i += kKeyThreads)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// In-place sigmoid for the banded case (heat_out aliases heat): runs AFTER decode_keys_kernel in stream
// order, so every band has read its halo rows as logits before any of them is overwritten.
__global__ void __launch_bounds__(256) sigmoid_store_kernel(const float *in, float *out, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 t = *reinterpret_cast<const float4 *>(in + i);
      t.x = sigmoidf_ref(t.x); t.y = sigmoidf_ref(t.y); t.z = sigmoidf_ref(t.z); t.w = sigmoidf_ref(t.w);
      *reinterpret_cast<float4 *>(out + i) = t;
    } else {
      for (long j = i; j < n; ++j) out[j] = sigmoidf_ref(in[j]);
    }
  }
}

// Test-time flip augmentation (lib/detectors/ctdet.py:32-38 with opt.flip_test; every test command of the reference's
// README passes --flip_test): the network ran on [image, its W-mirror];
//   hm.sigmoid_() in place;  hm' = (hm[0] + flip_W(hm[1])) / 2,   wh' = (wh[0] + flip_W(wh[1])) / 2,   reg' = reg[0]
// in one launch (the reference: sigmoid_, two flips, two adds, two divisions).  P pairs: originals 0 .. P-1, mirrors
// P .. 2P-1 (P = 1 is the reference's layout).
__global__ void __launch_bounds__(256)
flip_merge_kernel(float *__restrict__ hm, const float *__restrict__ wh, float *__restrict__ hm_out,
                  float *__restrict__ wh_out, long n_hm, long n_wh, int W, long pair_hm, long pair_wh) {
  // n_hm = P * cat * H * W output elements; pair_hm = P * cat * H * W (offset of the mirrors); same for wh
  const long total = n_hm + n_wh;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const bool is_hm = i < n_hm;
    const long j = is_hm ? i : i - n_hm;
    const long row = j / W;
    const int x = (int)(j - row * W);
    const long jm = row * W + (W - 1 - x);
    if (is_hm) {
      const float a = sigmoidf_ref(hm[j]), b = sigmoidf_ref(hm[pair_hm + jm]);
      hm[j] = a;                      // the reference's in-place hm.sigmoid_() on the whole pair (ctdet.py:32):
      hm[pair_hm + jm] = b;           // every element is read and written by exactly one thread
      hm_out[j] = __fdiv_rn(__fadd_rn(a, b), 2.0f);
    } else {
      wh_out[j] = __fdiv_rn(__fadd_rn(wh[j], wh[pair_wh + jm]), 2.0f);
    }
  }
}

// From the top bin down: the bin D with (count of bins > D) < need <= (count of bins >= D), by a block-wide
// suffix scan over bin pairs (a single thread walking 2048 LDS words costs ~55 us per level).
// Every thread of the workgroup calls it; nbins <= 2 * kSelThreads.  Results in *digit / *above (LDS).
__device__ void find_digit(const unsigned *h, int nbins, unsigned need, unsigned *scan, int *digit,
                           unsigned *above) {
  // Inclusive suffix sums over the threads' pair counts: inside a wave by shuffles, across the sixteen waves through
  // LDS -- two barriers per call (round 6; the Hillis-Steele scan over 1024 LDS words it replaces took twenty, ~5 us
  // of a kernel that is the last link of both graphs' critical paths and runs on 64 CUs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, np = nbins >> 1;
  const unsigned lo = tid < np ? h[2 * tid] : 0u, hi = tid < np ? h[2 * tid + 1] : 0u;
  unsigned v = lo + hi;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_down(v, off, 64);
    if (lane + off < 64) v += t;
  }
  __syncthreads();                                         // (scan[] may still be read by the previous call's threads)
  if (lane == 0) scan[wave] = v;
  __syncthreads();
  unsigned tail = 0u;
  for (int w = wave + 1; w < kSelThreads / 64; ++w) tail += scan[w];
  const unsigned mine = v + tail, next = mine - (lo + hi);
  if (tid < np && mine >= need && next < need) {           // the pair that crosses `need`
    if (next + hi >= need) {
      *digit = 2 * tid + 1;
      *above = next;
    } else {
      *digit = 2 * tid;
      *above = next + hi;
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kSelThreads)
decode_select_kernel(const unsigned long long *__restrict__ cand, unsigned *__restrict__ hist, unsigned cap,
                     const float *__restrict__ src, int need_sigmoid, const float *__restrict__ wh,
                     const float *__restrict__ reg, float *__restrict__ dets, int cat, int H, int W, int wh_ch, int K) {
  __shared__ unsigned h[2048];
  __shared__ unsigned long long list[kCap];
  __shared__ unsigned scan[kSelThreads];
  __shared__ int s_digit;
  __shared__ unsigned s_above, s_n, s_ncand;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int HW = H * W;
  const long total = (long)cat * HW;
  unsigned *gh = hist + (long)b * kHistStride;
  // candidates the keys kernel counted (above cap: not all were stored).  ONE thread reads the counter and the
  // workgroup takes it from LDS behind a barrier: thread 0 resets the word below, and a wave that started late
  // would otherwise read 0 and disagree with the others about the workgroup-uniform `slow`.
  // (the level-0 histogram is requested in the same round trip: it depends on nothing)
  unsigned h0[kBins / kSelThreads];
#pragma unroll
  for (int u = 0; u < kBins / kSelThreads; ++u) h0[u] = gh[tid + u * kSelThreads];
  if (tid == 0) s_ncand = gh[kBins];
  __syncthreads();
  const unsigned ncand = s_ncand;
  const unsigned long long *lp = cand + (size_t)b * cap;
  const float *sp = src + (long)b * total;
  const unsigned zkey = cdn::f2ord(0.0f);
  // The masked key of flat index i recomputed from the heat map (slow path only): the keys kernel's expression.
  auto key_at = [&](long i) __attribute__((always_inline)) -> unsigned {
    const int cls = (int)(i / HW), pix = (int)(i - (long)cls * HW), y = pix / W, x = pix - y * W;
    const float *pl = sp + (long)cls * HW;
    float m9 = -INFINITY, v = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          float t = pl[yy * W + xx];
          if (need_sigmoid) t = sigmoidf_ref(t);
          if (dy == 0 && dx == 0) v = t;
          m9 = fmaxf(m9, t);
        }
      }
    return masked_key(v, m9);
  };
  // fast: the K-th key is above zero and every key at or above it is in the list.  Workgroup-uniform.
  bool slow = !(ncand >= (unsigned)K && ncand <= cap);
  // One pass over the keys of the image: fn(key, flat index) -- over the candidate list (two 8-byte entries per
  // load, 8 loads in flight per thread: one workgroup per image, nothing else hides the round trips) or, on the
  // slow path, over every pixel.
  auto for_each_key = [&](auto &&fn) __attribute__((always_inline)) {
    if (!slow) {
      const unsigned npair = (ncand + 1) >> 1;
      for (unsigned q0 = tid; q0 < npair; q0 += kSelThreads * 8) {
        ulonglong2 e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned q = q0 + u * kSelThreads;
          e[u] = q < npair ? reinterpret_cast<const ulonglong2 *>(lp)[q] : make_ulonglong2(0ull, 0ull);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned q = q0 + u * kSelThreads;
          if (q < npair) {
            fn((unsigned)(e[u].x >> 32), (long)(0xFFFFFFFFu - (unsigned)(e[u].x & 0xFFFFFFFFull)));
            if (2 * q + 1 < ncand)
              fn((unsigned)(e[u].y >> 32), (long)(0xFFFFFFFFu - (unsigned)(e[u].y & 0xFFFFFFFFull)));
          }
        }
      }
    } else {
      for (long i = tid; i < total; i += kSelThreads) fn(key_at(i), i);
    }
  };
#pragma unroll
  for (int u = 0; u < kBins / kSelThreads; ++u) {
    h[tid + u * kSelThreads] = h0[u];
    gh[tid + u * kSelThreads] = 0;               // leave the global histogram zero for the next call
  }
  if (tid == 0) {
    gh[kBins] = 0;                               // ... and the candidate counter
    s_n = 0;
  }
  __syncthreads();
  unsigned need, pmask, pval;
  bool exhausted;
  for (;;) {                                     // at most two rounds: the list, then (rarely) the heat map itself
    need = (unsigned)K, pmask = 0, pval = 0;
    exhausted = false;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int lvl = 0;; ++lvl) {
      find_digit(h, 1 << widths[lvl], need, scan, &s_digit, &s_above);
      const unsigned digit = (unsigned)s_digit, group = h[digit];
      need -= s_above;
      pmask |= ((1u << widths[lvl]) - 1u) << shifts[lvl];
      pval |= digit << shifts[lvl];
      __syncthreads();
      if (group + (unsigned)K <= (unsigned)kCap) break;      // the threshold group fits the list
      if (lvl == 2) { exhausted = true; break; }             // a huge group of IDENTICAL keys
      for (int i = tid; i < 2048; i += kSelThreads) h[i] = 0;
      __syncthreads();
      const int sh = shifts[lvl + 1];
      const unsigned wm = (1u << widths[lvl + 1]) - 1u;
      for_each_key([&](unsigned k, long) {
        if ((k & pmask) == pval) atomicAdd(&h[(k >> sh) & wm], 1u);
      });
      __syncthreads();
    }
    if (!exhausted || slow) break;
    // equal keys at the threshold that do not fit the LDS list must be taken by ascending index, which the
    // unordered candidate list cannot give: start over on the pixels (level-0 histogram rebuilt from them)
    slow = true;
    for (int i = tid; i < 2048; i += kSelThreads) h[i] = 0;
    __syncthreads();
    {
      unsigned zeros = 0;
      for (long i = tid; i < total; i += kSelThreads) {
        const unsigned k = key_at(i);
        if (k == zkey) ++zeros;
        else atomicAdd(&h[k >> 21], 1u);
      }
      if (zeros) atomicAdd(&h[zkey >> 21], zeros);
    }
    __syncthreads();
  }
  // ---- collect: everything above the threshold group, and the group (or its first `need` by index)
  if (!exhausted) {
    for_each_key([&](unsigned k, long i) {
      if ((k & pmask) >= pval) {
        const unsigned slot = atomicAdd(&s_n, 1u);
        if (slot < (unsigned)kCap)
          list[slot] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      }
    });
  } else {
    // (slow path) every thread owns a contiguous index range, so a block scan gives each group member its rank
    const long chunk = (total + kSelThreads - 1) / kSelThreads;
    const long i0 = (long)tid * chunk, i1 = min(total, i0 + chunk);
    unsigned cnt = 0;
    for (long i = i0; i < i1; ++i) {
      const unsigned k = key_at(i);
      if (k > pval) {
        const unsigned slot = atomicAdd(&s_n, 1u);
        if (slot < (unsigned)kCap)
          list[slot] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      } else if (k == pval) {
        ++cnt;
      }
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kSelThreads; off <<= 1) {     // inclusive Hillis-Steele scan
      const unsigned v = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    unsigned rank = scan[tid] - cnt;                      // exclusive
    for (long i = i0; i < i1 && rank < need; ++i)
      if (key_at(i) == pval) {
        const unsigned slot = atomicAdd(&s_n, 1u);
        if (slot < (unsigned)kCap)
          list[slot] = ((unsigned long long)pval << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        ++rank;
      }
  }
  __syncthreads();
  const unsigned n = min(s_n, (unsigned)kCap);
  // ---- descending order (score desc, index asc through the inverted index; the entries are distinct) -----------
  constexpr unsigned kRankMax = 256;
  if (n <= kRankMax) {
    // few entries (the usual case: K plus the threshold group): every entry COUNTS the entries above it -- n broadcast
    // LDS reads per thread, two barriers, against 28-45 barrier-separated steps of the bitonic network (round 6: the
    // sort was 7 of this kernel's 20 us, and the kernel is the last link of both graphs)
    const unsigned long long mine = (unsigned)tid < n ? list[tid] : 0ull;
    unsigned rank = 0;
    if ((unsigned)tid < n)
      for (unsigned j = 0; j < n; ++j) rank += list[j] > mine ? 1u : 0u;
    __syncthreads();
    if ((unsigned)tid < n) list[rank] = mine;
    else if (tid < K) list[tid] = 0ull;                  // below every real entry (fewer than K collected)
    __syncthreads();
  } else {
  int nsort = 2;
  while ((unsigned)nsort < n) nsort <<= 1;               // sort only as much as was collected
  for (int i = tid; i < nsort; i += kSelThreads)
    if ((unsigned)i >= n) list[i] = 0ull;                 // below every real entry
  __syncthreads();
  // ---- bitonic sort, descending ----------------------------------------------------------------
  for (int k2 = 2; k2 <= nsort; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < nsort; i += kSelThreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = list[i], c = list[ixj];
          const bool up = (i & k2) == 0;                  // descending blocks first
          if (up ? (a < c) : (a > c)) {
            list[i] = c;
            list[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- boxes ---------------------------------------------------------------------------------
  for (int k = tid; k < K; k += kSelThreads) {
    const unsigned long long e = list[k];
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull);
    const float score = cdn::ord2f((unsigned)(e >> 32));
    const int cls = (int)(idx / (unsigned)HW), pix = (int)(idx % (unsigned)HW);
    float xs = (float)(pix % W), ys = (float)(pix / W);
    if (reg) {
      xs += reg[((long)b * 2 + 0) * HW + pix];
      ys += reg[((long)b * 2 + 1) * HW + pix];
    } else {
      xs += 0.5f;
      ys += 0.5f;
    }
    const int wc = wh_ch == 2 ? 0 : 2 * cls;              // cat_spec_wh: channel pair of the class
    const float w = wh[((long)b * wh_ch + wc) * HW + pix];
    const float hh = wh[((long)b * wh_ch + wc + 1) * HW + pix];
    float *o = dets + ((long)b * K + k) * 6;
    o[0] = xs - w / 2;
    o[1] = ys - hh / 2;
    o[2] = xs + w / 2;
    o[3] = ys + hh / 2;
    o[4] = score;
    o[5] = (float)cls;
  }
}

}  // namespace

// Row bands of the keys kernel: ~36 KiB of plane per workgroup (3 workgroups per CU with the histogram), at least
// 8 rows, balanced.
static int band_rows(int64_t H, int64_t W) {
  const size_t wp_bytes = (size_t)(((W + 3) & ~3) + 8) * 4;
#ifndef CDN_DECODE_BAND_KIB
#define CDN_DECODE_BAND_KIB 36
#endif
  constexpr int kb_kib = CDN_DECODE_BAND_KIB;
  int RB = (int)std::max<long>(8, (long)((size_t)kb_kib * 1024 / wp_bytes) - 2);
  RB = (int)std::min<long>(RB, H);
  const int nbands = (int)cdn::ceil_div(H, RB);
  return (int)cdn::ceil_div(H, nbands);
}

extern "C" size_t cdn_ctdet_decode_workspace_bytes(int64_t B, int64_t cat, int64_t H, int64_t W) {
  auto r = [](size_t b) { return (b + 255) / 256 * 256; };
  return r((size_t)(B * cat * H * W) * 4) + r((size_t)B * kHistStride * 4);
}

extern "C" int cdn_ctdet_decode(const float *heat, const float *wh, const float *reg, int64_t B,
                                int64_t cat, int64_t H, int64_t W, int cat_spec_wh, int K,
                                int apply_sigmoid, float *heat_out, float *dets, void *workspace,
                                size_t workspace_bytes, void *stream) {
  CDN_REQUIRE(heat && wh && dets && workspace, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(B > 0 && cat > 0 && H > 0 && W > 0 && K > 0, CDN_ERR_ARG, "non-positive size");
  CDN_REQUIRE(K <= 1024 && K <= cat * H * W, CDN_ERR_UNSUPPORTED, "K = %d unsupported", K);
  CDN_REQUIRE(B <= 65535 && cat <= 65535 && cat * H * W < (1ll << 31), CDN_ERR_UNSUPPORTED,
              "shape too large");
  const size_t wp_bytes = (size_t)(((W + 3) & ~3) + 8) * 4;
  CDN_REQUIRE(3 * wp_bytes <= 128 * 1024, CDN_ERR_UNSUPPORTED, "heat-map row of %lld pixels does not fit LDS",
              (long long)W);
  CDN_REQUIRE(workspace_bytes >= cdn_ctdet_decode_workspace_bytes(B, cat, H, W) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
              CDN_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
  hipStream_t st = cdn::as_stream(stream);
  auto r = [](size_t b) { return (b + 255) / 256 * 256; };
  // candidate lists: cat * H * W * 4 bytes per image = room for half the pixels as 8-byte entries
  unsigned long long *cand = static_cast<unsigned long long *>(workspace);
  const unsigned cap = (unsigned)((cat * H * W) / 2) & ~1u;
  const int RB = band_rows(H, W);
  const int nbands = (int)cdn::ceil_div(H, RB);
  // the per-image histograms (+ candidate counters) live in the LAST bytes: zero them once, every call leaves them zero
  unsigned *hist = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + workspace_bytes / 256 * 256 -
                                                r((size_t)B * kHistStride * 4));
  const size_t lds = (size_t)(RB + 2) * wp_bytes;
  CDN_REQUIRE(lds <= 128 * 1024, CDN_ERR_UNSUPPORTED, "heat-map band does not fit LDS");
  (void)hipFuncSetAttribute((const void *)decode_keys_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  // heat_out overlapping heat with a sigmoid and more than one band per plane: a band's halo rows belong to
  // its neighbours, which may already have stored their sigmoid -- keep the keys kernel read-only there and
  // store the sigmoid from a second kernel behind it
  const long n_heat = (long)(B * cat * H * W);
  const bool overlap = heat_out && heat_out < heat + n_heat && heat < heat_out + n_heat;
  const bool deferred = overlap && apply_sigmoid && nbands > 1;
  CDN_REQUIRE(!overlap || heat_out == heat, CDN_ERR_ARG, "heat_out partially overlaps heat");
  decode_keys_kernel<<<dim3((unsigned)cat, (unsigned)B, (unsigned)nbands), kKeyThreads, lds, st>>>(
      heat, cand, hist, deferred ? nullptr : heat_out, (int)cat, (int)H, (int)W, apply_sigmoid, RB, cap);
  int rc = cdn::check_launch("ctdet decode keys");
  if (rc) return rc;
  if (deferred) {
    const unsigned blocks = (unsigned)std::min<long>(cdn::ceil_div(n_heat, 1024), 4096);
    sigmoid_store_kernel<<<blocks, 256, 0, st>>>(heat, heat_out, n_heat);
    rc = cdn::check_launch("ctdet decode in-place sigmoid");
    if (rc) return rc;
  }
  // the slow path of the select kernel recomputes keys from the scores: the stored sigmoid if there is one
  const float *src = heat_out ? heat_out : heat;
  const int need_sigmoid = (apply_sigmoid && !heat_out) ? 1 : 0;
  decode_select_kernel<<<(unsigned)B, kSelThreads, 0, st>>>(cand, hist, cap, src, need_sigmoid, wh, reg, dets, (int)cat,
                                                           (int)H, (int)W, cat_spec_wh ? (int)(2 * cat) : 2, K);
  return cdn::check_launch("ctdet decode select");
}

extern "C" int cdn_ctdet_flip_merge(float *hm, const float *wh, int64_t P, int64_t cat, int64_t wh_ch, int64_t H,
                                    int64_t W, float *hm_out, float *wh_out, void *stream) {
  CDN_REQUIRE(hm && wh && hm_out && wh_out, CDN_ERR_ARG, "null pointer");
  CDN_REQUIRE(P > 0 && cat > 0 && wh_ch > 0 && H > 0 && W > 0 && 2 * P * std::max(cat, wh_ch) * H * W < (1ll << 31),
              CDN_ERR_ARG, "bad size");
  CDN_REQUIRE(hm_out != hm && wh_out != wh, CDN_ERR_ARG, "the merge is not an in-place operation (it reads mirrored columns)");
  const long n_hm = (long)(P * cat * H * W), n_wh = (long)(P * wh_ch * H * W);
  {
    // the four ranges {hm (2P images, written in place), wh (read), hm_out, wh_out} must be pairwise disjoint except
    // hm / wh among themselves never are the same tensor anyway: a shared hm_out == wh_out buffer (two heads of equal
    // shape keyed by shape alone) used to be written by both halves of the launch
    auto disjoint = [](const float *a, long na, const float *b, long nb) { return a + na <= b || b + nb <= a; };
    CDN_REQUIRE(disjoint(hm_out, n_hm, wh_out, n_wh) && disjoint(hm_out, n_hm, hm, 2 * n_hm) &&
                    disjoint(hm_out, n_hm, wh, 2 * n_wh) && disjoint(wh_out, n_wh, hm, 2 * n_hm) &&
                    disjoint(wh_out, n_wh, wh, 2 * n_wh),
                CDN_ERR_ARG, "hm_out / wh_out overlap each other or the inputs");
  }
  const unsigned blocks = (unsigned)std::min<long>(cdn::ceil_div(n_hm + n_wh, 256), (long)cdn::kCUs * 16);
  flip_merge_kernel<<<blocks, 256, 0, cdn::as_stream(stream)>>>(hm, wh, hm_out, wh_out, n_hm, n_wh, (int)W, n_hm, n_wh);
  return cdn::check_launch("ctdet flip merge");
}
