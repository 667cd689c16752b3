"""Shared pieces of the fused schedules: the synthetic hot path (DeconvLayers / build_hot_path / make_input), the
stage-call flags of include/codenet_dcn.h, QuantAct predicates, weight forms, byte accounting.

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn

from ..modules.dcn_deform_conv import DeformConvWithOffsetScaleBoundPositive
from ..portable_quantizer.quantization_utils.quantize_model import quantize_deform_stages


BN_MOMENTUM = 0.1


def stage_shapes(input_res=512, w2=False):
    """[(C_in, C_out, H, W)] of the three deform stages."""
    c0 = 2153 if w2 else 1024          # shufflenetv2_dcn.py:199-202,293-296
    r = input_res // 32
    return [(c0, 256, r, r), (256, 128, 2 * r, 2 * r), (128, 64, 4 * r, 4 * r)]


class DeconvLayers(nn.Module):
    """Container with the reference's attribute name so the quantiser's surgery applies."""

    def __init__(self, w2=False, planes=None):
        super().__init__()
        planes = planes or [2153 if w2 else 1024, 256, 128, 64]   # C_in of stage 0, then every C_out
        layers = []
        for cin, cout in zip(planes[:-1], planes[1:]):
            layers += [
                DeformConvWithOffsetScaleBoundPositive(cin, cout, 3, 1, 1, groups=cout, bias=False,
                                                        hidden_state=128, BN_MOMENTUM=BN_MOMENTUM),
                nn.BatchNorm2d(cout, momentum=BN_MOMENTUM),
                nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="nearest"),
            ]
        self.deconv_layers = nn.Sequential(*layers)

    def forward(self, x):
        from ..functions.codenet_stage import forward_stage_blocks
        return forward_stage_blocks(self.deconv_layers, x)      # (== self.deconv_layers(x); fused blocks in the QAT step)


def build_hot_path(w2=False, quantized=True, seed=317, scale_std=3.0, planes=None, wt_percentile=False):
    """Seeded synthetic weights: reference initialisers, except a non-degenerate
    conv_scale (weight ~ N(0, scale_std/sqrt(C)), bias 1 => s ~ N(1, scale_std) on unit-power inputs,
    clipped to [-7, 8] with ~1 % of pixels at each clamp) and non-trivial BN running statistics."""
    g = torch.Generator().manual_seed(seed)
    net = DeconvLayers(w2=w2, planes=planes)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, DeformConvWithOffsetScaleBoundPositive):
                C = m.in_channels
                m.conv_scale.weight.copy_(torch.randn(1, C, 1, 1, generator=g) * (scale_std / C ** 0.5))
                bound = 1.0 / (9 * C) ** 0.5
                m.conv.weight.copy_(torch.empty(C, 1, 3, 3).uniform_(-bound, bound, generator=g))
                m.conv_channel.weight.copy_(
                    torch.randn(m.out_channels, C, 1, 1, generator=g) * (2.0 / C) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    if quantized:
        # (the README's QAT command passes --wt-percentile, its test commands do not: README.md:87-116)
        quantize_deform_stages(net, 4, 8, "symmetric", "asymmetric", True, bool(wt_percentile), False)
    return net.eval()


def make_input(batch, input_res=512, w2=False, seed=0, device="cpu"):
    """Stage-0 input: what layer4 (conv1x1 + BN + ReLU [+ QuantAct]) hands over -- non-negative,
    unit power."""
    C, _, H, W = stage_shapes(input_res, w2)[0]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, C, H, W, generator=g).abs_() * 1.66   # E[x^2] ~ 1 after the fold
    return x.to(device)


def set_running_stat(net, flag):
    """running_stat=True is the reference's behaviour even in eval();
    False freezes the QuantAct ranges."""
    from ..portable_quantizer.quant_modules import QuantAct
    for m in net.modules():
        if isinstance(m, QuantAct):
            m.running_stat = flag


# per-call gather schedule choice for an NCHW stage-0 input (include/codenet_dcn.h: CDN_X_GATHER_*), set as
# `path.gather_flag` on a FusedHotPath / FrozenHotPath by the tests that compare the two schedules bit for bit
GATHER_PER_ITEM, GATHER_PERSISTENT = 0x100, 0x200


class OverflowFlags:
    """The sticky saturation flags of a byte-code schedule: one int32 word PER LAUNCH GROUP (the kernels only
    ``atomicOr(flag, 1)``, so a distinct word per launch attributes a saturated code to the QuantAct(s) whose codes
    that launch writes).  ``any()`` is the old single-flag question; ``acts()`` names the QuantActs to widen
    (calibrate_serving)."""

    def __init__(self, n, dev):
        self.words = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
        self.who = [[] for _ in range(max(1, n))]
        self.off = 0

    def ptr(self, i=0):
        return self.words.data_ptr() + 4 * (self.off + i)

    def data_ptr(self):
        return self.ptr(0)

    def count(self):
        return len(self.who) - self.off

    def slice(self, lo):
        """a view of the words from `lo` on, sharing words and names (for a consumer that numbers its own launches from 0)"""
        v = OverflowFlags.__new__(OverflowFlags)
        v.words, v.who, v.off = self.words, self.who, self.off + lo
        return v

    def name(self, i, acts):
        self.who[self.off + i] = list(acts)

    def any(self, reset=True):
        """True when some code saturated since the last reset (synchronises)."""
        hit = bool(self.words.any().item())
        if hit and reset:
            self.words.zero_()
        return hit

    def acts(self, reset=True):
        """The QuantActs of the launches that saturated since the last reset (synchronises)."""
        w = self.words.tolist()
        out = []
        for i, v in enumerate(w):
            if v:
                for a in self.who[i]:
                    if all(a is not b for b in out):
                        out.append(a)
        if reset and any(w):
            self.words.zero_()
        return out


ACT_PERCENTILE = 0x400      # CDN_X_ACT_PERCENTILE (include/codenet_dcn.h)


WCODES_KB = 0x800           # CDN_X_WCODES_KB


DEFER_RANGE = 0x1000        # CDN_X_DEFER_RANGE


PHASE_SCALE, PHASE_GATHER, PHASE_POINTWISE = 0x2000, 0x4000, 0x8000      # CDN_X_PHASE_*


def stage_int8_codes(convbn, kblocked=True):
    """(i8 triple or None, flag) for the pointwise conv of a fused stage: the int8 form of the folded weights, with the
    k-blocked copy behind the codes -- and CDN_X_WCODES_KB to OR into the stage call's layout argument -- where the
    library has a use for it (long-K stages: cdn_codenet_wcodes_kb_columns)."""
    from .. import _native as N_
    lib = N_.lib()
    conv = convbn.conv
    cols = lib.cdn_codenet_wcodes_kb_columns(conv.in_channels, conv.out_channels) if kblocked else 0
    if cols:
        i8 = convbn.folded_int8_kblocked(cols, lib.cdn_codenet_wcodes_kb_offset(conv.in_channels, conv.out_channels))
        return i8, (WCODES_KB if i8 is not None else 0)
    return convbn.folded_int8(), 0


def act_fusable(act, allow_percentile=False, allow_global=False):
    """The fused schedules implement the reference's default QuantAct: plain batch min/max tracking,
    asymmetric, quantising (quant_modules.py:163-225 with percentile=False).  Symmetric activations and
    full_precision_flag stay on the module path; --act-percentile too, except in the three deform stages
    (allow_percentile: FusedHotPath, round 4 -- the stage entry point follows the order statistics with
    cdn_kth_values between its kernels); the multi-process global-range mode too, except in the three deform stages
    (allow_global: FusedHotPath, round 5 -- the stage call is split at its QuantActs, `_global_commit`)."""
    glob = getattr(act, "global_range", False) and act.running_stat
    return (act.quant_mode == "asymmetric" and (allow_percentile or not act.percentile)
            and not act.full_precision_flag
            and (not glob or (allow_global and not act.percentile)))


def global_range_active(acts):
    """True when the QuantActs of a fused stage call must see batch extremes reduced over the ranks: the mode is on, the
    ranges are tracked and there is more than one rank (one rank: the plain call computes the same thing)."""
    import torch.distributed as dist
    acts = [a for a in acts if a is not None]
    if not any(getattr(a, "global_range", False) and a.running_stat for a in acts):
        return False
    if not all(getattr(a, "global_range", False) for a in acts):
        raise NotImplementedError("the QuantActs of one fused stage must all or none be in global-range mode")
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def uniform_act_settings(acts, what, allow_percentile=False, allow_global=False):
    """(bits, momentum, running) shared by the QuantActs of one fused C call, which takes them once.
    allow_percentile: the call's QuantActs may all be percentile ones (never a mixture)."""
    acts = [a for a in acts if a is not None]
    if not acts:
        return 8, 0.99, 0
    if allow_percentile and len({bool(a.percentile) for a in acts}) != 1:
        raise NotImplementedError("%s: the QuantActs of one fused call must all or none use percentile ranges" % what)
    for a in acts:
        if not act_fusable(a, allow_percentile, allow_global):
            raise NotImplementedError("%s: QuantAct(percentile=%s, quant_mode=%s, full_precision_flag=%s) is not "
                                      "implemented by the fused schedule; use the module path"
                                      % (what, a.percentile, a.quant_mode, a.full_precision_flag))
    st = {(a.activation_bit, float(a.momentum), int(bool(a.running_stat))) for a in acts}
    if len(st) != 1:
        raise NotImplementedError("%s: the QuantActs of one fused call must share activation_bit / momentum / "
                                  "running_stat (got %s)" % (what, sorted(st)))
    return st.pop()


def algorithmic_bytes(batch, input_res=512, w2=False, fused=False, act_bytes=4):
    """Per-step algorithmic bytes of the hot path: unfused 3-kernel definition
    scale (C+1)HW*4 + gather (2C+1)HW*4 + pointwise (C+Co)HW*4, or the fused-stage definition."""
    tot = 0
    per = {}
    for (C, Co, H, W) in stage_shapes(input_res, w2):
        HW = H * W
        if fused:
            b = (C + Co) * HW * act_bytes
            per[(C, Co, H)] = {"fused": b * batch}
        else:
            sc, ga, pw = (C + 1) * HW * 4, (2 * C + 1) * HW * 4, (C + Co) * HW * 4
            b = sc + ga + pw
            per[(C, Co, H)] = {"scale": sc * batch, "dw": ga * batch, "pointwise": pw * batch}
        tot += b * batch
    return tot, per


def bn_affine(cache, bn):
    """BatchNorm (eval) as a per-channel affine, cached until one of its tensors changes (keyed on
    data_ptr + version like the weight caches; refreshed in place so captured graphs stay valid)."""
    from ..portable_quantizer.quant_modules import refresh_in_place
    src = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version, t.device) for t in src)
    ent = cache.get(id(bn))
    if ent is None or ent[0] != key:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var + bn.eps)
            es = (bn.weight * inv).contiguous()
            new = (es, (bn.bias - bn.running_mean * es).contiguous())
            ent = (key, refresh_in_place(ent[1] if ent else None, new))
        cache[id(bn)] = ent
    return ent[1]
