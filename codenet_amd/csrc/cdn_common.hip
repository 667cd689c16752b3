// cdn_common.hip -- error plumbing + geometry checks.
#include "cdn_common.h"

#include <climits>
#include <vector>

namespace cdn {

char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int make_geom(Geom *g, int64_t N, int64_t C, int64_t H, int64_t W, int64_t Co, int kH, int kW,
              int sH, int sW, int pH, int pW, int dH, int dW, int group, int dg) {
  CDN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Co > 0, CDN_ERR_ARG,
              "non-positive tensor size N=%lld C=%lld H=%lld W=%lld Co=%lld", (long long)N,
              (long long)C, (long long)H, (long long)W, (long long)Co);
  CDN_REQUIRE(kW > 0 && kH > 0, CDN_ERR_SHAPE,
              "kernel size should be greater than zero, but got kH: %d kW: %d", kH, kW);
  CDN_REQUIRE(sW > 0 && sH > 0, CDN_ERR_SHAPE,
              "stride should be greater than zero, but got dH: %d dW: %d", sH, sW);
  CDN_REQUIRE(dW > 0 && dH > 0, CDN_ERR_SHAPE,
              "dilation should be greater than 0, but got dilationH: %d dilationW: %d", dH, dW);
  CDN_REQUIRE(pH >= 0 && pW >= 0, CDN_ERR_SHAPE, "negative padding %d %d", pH, pW);
  CDN_REQUIRE(group > 0 && dg > 0, CDN_ERR_SHAPE, "group / deformable_group must be positive");
  CDN_REQUIRE(C % group == 0 && Co % group == 0, CDN_ERR_SHAPE,
              "channels (%lld in, %lld out) not divisible by group %d", (long long)C,
              (long long)Co, group);
  CDN_REQUIRE(C % dg == 0, CDN_ERR_SHAPE, "input channels must divide deformable group size");
  const int64_t Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
  const int64_t Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
  CDN_REQUIRE(H + 2 * pH >= dH * (kH - 1) + 1 && W + 2 * pW >= dW * (kW - 1) + 1 && Ho >= 1 &&
                  Wo >= 1,
              CDN_ERR_SHAPE,
              "Given input size: (%lld x %lld x %lld). Calculated output size: (%lld x %lld x "
              "%lld). Output size is too small",
              (long long)C, (long long)H, (long long)W, (long long)Co, (long long)Ho,
              (long long)Wo);
  CDN_REQUIRE(H >= kH && W >= kW, CDN_ERR_SHAPE, "input image is smaller than kernel");
  const int64_t lim = INT_MAX;
  CDN_REQUIRE(N * C * H * W < lim && N * Co * Ho * Wo < lim &&
                  N * dg * 2 * kH * kW * Ho * Wo < lim && C * kH * kW * N * Ho * Wo / group < lim * 64,
              CDN_ERR_UNSUPPORTED, "tensor too large for 32-bit indexing");
  g->N = (int)N; g->C = (int)C; g->H = (int)H; g->W = (int)W; g->Co = (int)Co;
  g->kH = kH; g->kW = kW; g->sH = sH; g->sW = sW; g->pH = pH; g->pW = pW; g->dH = dH; g->dW = dW;
  g->G = group; g->DG = dg; g->Ho = (int)Ho; g->Wo = (int)Wo;
  return CDN_OK;
}

namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;   // 2 per record
  std::vector<int> kid, tag;
  size_t used = 0;              // records
};
Prof &prof() {
  static thread_local Prof p;
  return p;
}
}  // namespace

ProfScope::ProfScope(int kernel_id, int tag, hipStream_t st) : slot_(-1), st_(st) {
  Prof &p = prof();
  if (!p.on) return;
  if (p.ev.size() < 2 * (p.used + 1)) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    p.ev.push_back(a);
    p.ev.push_back(b);
    p.kid.push_back(0);
    p.tag.push_back(0);
  }
  slot_ = (int)p.used++;
  p.kid[slot_] = kernel_id;
  p.tag[slot_] = tag;
  (void)hipEventRecord(p.ev[2 * slot_], st);
}

ProfScope::~ProfScope() {
  if (slot_ >= 0) (void)hipEventRecord(prof().ev[2 * slot_ + 1], st_);
}

}  // namespace cdn

extern "C" int cdn_profile_enable(int on) {
  auto &p = cdn::prof();
  p.on = on != 0;
  p.used = 0;
  return CDN_OK;
}

extern "C" int cdn_profile_read(int max_records, int *kernel_ids, int *tags, float *ms) {
  auto &p = cdn::prof();
  int n = 0;
  for (size_t i = 0; i < p.used && n < max_records; ++i) {
    if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) continue;
    kernel_ids[n] = p.kid[i];
    tags[n] = p.tag[i];
    ms[n] = t;
    ++n;
  }
  p.used = 0;
  return n;
}

extern "C" int cdn_abi_version(void) { return CDN_ABI_VERSION; }
extern "C" const char *cdn_last_error(void) { return cdn::err_buf(); }

namespace cdn {
size_t aux_workspace_bytes() {
  auto r = [](size_t b) { return (b + 255) / 256 * 256; };
  return r((size_t)kMaxPartials * 8) + r((size_t)kArriveWords * 4);
}
bool aux_workspace(void *workspace, size_t bytes, AuxWs *w) {
  if (!workspace || bytes < aux_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0)
    return false;
  char *p = static_cast<char *>(workspace);
  const size_t cnt = ((size_t)kArriveWords * 4 + 255) / 256 * 256;
  w->partials = reinterpret_cast<float2 *>(p);
  w->arrive = reinterpret_cast<unsigned *>(p + bytes / 256 * 256 - cnt);
  return true;
}
}  // namespace cdn
