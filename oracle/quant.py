"""oracle/quant.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (torch) restatement of the reference's fake-quantisation arithmetic and of the W4A8
composition of the CoDeNet deform stage, each function citing what it follows:

    portable_quantizer/quantization_utils/quant_utils.py   (linear_quantize :33-41,
        linear_dequantize :44-52, asymmetric params :60-75, symmetric params :78-84,
        AsymmetricQuantFunction :172-200, SymmetricQuantFunction :207-225)
    portable_quantizer/quant_modules.py  (QuantAct :163-225, Quant_Conv2d :228-321,
        QuantBnConv2d :324-419, QuantDeformConv2d :422-517,
        QuantDeformConvWithOffsetScaleBoundPositive :621-671, QuantBaseNode :809-907,
        QuantDepthwiseNode :1013-1071)

Pinned against the reference's own Python modules (imported in the build container only) by
tests/golden/make_golden.py -> tests/golden/*.npz, checked in tests/test_quant_oracle.py.
All arithmetic is fp32 torch on CPU, in the reference's expression order.
"""
import math

import torch
import torch.nn.functional as F

from . import dcn as O

ANCHOR = torch.tensor([-1, -1, -1, 0, -1, 1, 0, -1, 0, 0, 0, 1, 1, -1, 1, 0, 1, 1],
                      dtype=torch.float32).view(1, 18, 1, 1)


# ---- activations: asymmetric, per tensor (quant_utils.py:60-75,193-200) ------------------------

def act_params(x_min, x_max, bits=8):
    n = 2 ** bits - 1
    scale = n / torch.clamp(x_max - x_min, min=1e-10)
    zp = (scale * x_min).round() + 2 ** (bits - 1)
    return scale, zp


def act_codes(x, scale, zp):
    return torch.round(scale * x - zp)           # NOT clamped


def act_dequant(q, scale, zp):
    return (q + zp) / scale


class QuantActState:
    """x_min / x_max tracking of QuantAct.forward (quant_modules.py:202-225)."""

    def __init__(self, bits=8, momentum=0.99, x_min=0.0, x_max=0.0):
        self.bits = bits
        self.momentum = momentum
        self.x_min = torch.tensor([x_min], dtype=torch.float32)
        self.x_max = torch.tensor([x_max], dtype=torch.float32)

    def batch_stats(self, x, percentile=False):
        if not percentile:
            return x.min(), x.max()
        flat = x.reshape(-1)                     # quant_utils.py:18-30
        n = flat.shape[0]
        lo = torch.kthvalue(flat, k=round(n * 0.1 * 0.01)).values
        hi = torch.kthvalue(flat, k=round(n * 99.9 * 0.01)).values
        return lo, hi

    def update(self, x, percentile=False):
        bmin, bmax = self.batch_stats(x, percentile)
        if self.x_min == self.x_max:             # :211-213  (note: +=)
            self.x_min += bmin
            self.x_max += bmax
        else:                                    # :217-219
            self.x_min += (self.momentum - 1.) * self.x_min + (1. - self.momentum) * bmin
            self.x_max += (self.momentum - 1.) * self.x_max + (1. - self.momentum) * bmax

    def __call__(self, x, running=True, percentile=False, return_codes=False):
        if running:
            self.update(x, percentile)
        scale, zp = act_params(self.x_min, self.x_max, self.bits)
        q = act_codes(x, scale, zp)
        out = act_dequant(q, scale, zp)
        return (out, q) if return_codes else out


# ---- weights: symmetric, per output channel (quant_utils.py:78-84,207-225) ---------------------

def weight_range(w2d, percentile=False):
    """Per-row (w_min, w_max) as the Quant*Conv modules derive them (quant_modules.py:281-300)."""
    if not percentile:
        return w2d.min(dim=1).values, w2d.max(dim=1).values
    L = w2d.shape[1]
    if L < 10:
        return w2d.min(dim=1).values * 0.95, w2d.max(dim=1).values * 0.95
    lo = math.ceil(L * 0.1 * 0.01)
    hi = math.ceil(L * 99.9 * 0.01)
    return (torch.kthvalue(w2d, k=lo, dim=1).values, torch.kthvalue(w2d, k=hi, dim=1).values)


def weight_fake_quant(w, bits=4, percentile=False, return_codes=False):
    """w [Co, ...] -> fake-quantised w' (and integer codes, per-channel scale)."""
    Co = w.shape[0]
    w_min, w_max = weight_range(w.contiguous().view(Co, -1), percentile)
    mag = torch.max(torch.stack([w_min.abs(), w_max.abs()], dim=1), dim=1).values
    n = 2 ** (bits - 1) - 1
    scale = n / torch.clamp(mag, min=1e-10)
    sv = scale.view(-1, *([1] * (w.dim() - 1)))
    q = torch.clamp(torch.round(sv * w - 0.0), -(n + 1), n)
    wq = (q + 0.0) / sv
    return (wq, q, scale) if return_codes else wq


def fold_bn(conv_w, conv_b, bn_w, bn_b, bn_mean, bn_var, eps):
    """QuantBnConv2d.forward :365-372."""
    running_std = torch.sqrt(bn_var + eps)
    scale_factor = bn_w / running_std
    w = conv_w * scale_factor.reshape([conv_w.shape[0], 1, 1, 1])
    b = conv_b if conv_b is not None else torch.zeros_like(bn_mean)
    b = (b - bn_mean) * scale_factor + bn_b
    return w, b


# ---- stage compositions -----------------------------------------------------------------------

def stage_fp32(x, w_scale, b_scale, w_dw, w_pw, lo=-7.0, hi=8.0):
    """DeformConvWithOffsetScaleBoundPositive.forward (modules/dcn_deform_conv.py:323-330)."""
    C = x.shape[1]
    s = torch.clamp(F.conv2d(x, w_scale, b_scale), lo, hi)
    o = ANCHOR * (s - 1)
    d = O.deform_conv_forward(x, o, w_dw, 1, 1, 1, C, 1)
    y = F.conv2d(d, w_pw) if w_pw is not None else d
    return {"s": s, "d": d, "y": y}


def stage_w4a8(x, w_scale, b_scale, w_dw, w_pw, bn, act_s, act_d, w_bits=4, running=True,
               wt_percentile=False, act_percentile=False, lo=-7.0, hi=8.0):
    """QuantDeformConvWithOffsetScaleBoundPositive.forward (quant_modules.py:668-671).
    bn = (weight, bias, running_mean, running_var, eps); act_s / act_d are QuantActState."""
    C = x.shape[1]
    wq_s = weight_fake_quant(w_scale, w_bits, wt_percentile)
    s_raw = F.conv2d(x, wq_s, b_scale)
    s_q, s_codes = act_s(torch.clamp(s_raw, lo, hi), running, act_percentile, return_codes=True)
    o = ANCHOR * (s_q - 1)
    wq_d = weight_fake_quant(w_dw, w_bits, wt_percentile)
    d = O.deform_conv_forward(x, o, wq_d, 1, 1, 1, C, 1)
    d_q, d_codes = act_d(d, running, act_percentile, return_codes=True)
    wf, bf = fold_bn(w_pw, None, *bn)
    wq_p = weight_fake_quant(wf, w_bits, wt_percentile)
    y = F.conv2d(d_q, wq_p, bf)
    return {"s": s_q, "s_codes": s_codes, "d": d, "d_q": d_q, "d_codes": d_codes, "y": y}


# ---- detection heads ------------------------------------------------

def head_fp32(x, w1, bn1, w2, bn2, w3, b3, eps=1e-5):
    """The fp32 head nn.Sequential of lib/models/networks/shufflenetv2_dcn.py:244-262:
    conv1x1 -> BN -> ReLU -> depthwise 3x3 -> BN -> ReLU -> conv1x1 + bias (BN in eval mode).
    bn = (weight, bias, running_mean, running_var)."""
    def bn_eval(t, bn):
        w, b, m, v = bn
        return F.batch_norm(t, m, v, w, b, False, 0.0, eps)
    y1 = torch.relu(bn_eval(F.conv2d(x, w1), bn1))
    y2 = torch.relu(bn_eval(F.conv2d(y1, w2, None, 1, 1, 1, w2.shape[0]), bn2))
    return {"y1": y1, "y2": y2, "out": F.conv2d(y2, w3, b3)}


def head_w4a8(x, w1, bn1, w2, bn2, w3, b3, act1, act3, w_bits=4, running=True, wt_percentile=False,
              act_percentile=False, eps=1e-5):
    """QuantDepthwiseNode.forward (quant_modules.py:1059-1071): QuantBnConv2d -> ReLU -> QuantAct ->
    QuantBnConv2d (depthwise) -> ReLU -> QuantAct -> Quant_Conv2d.  act1 / act3 are QuantActState."""
    wf1, bf1 = fold_bn(w1, None, *bn1, eps)
    y1 = F.conv2d(x, weight_fake_quant(wf1, w_bits, wt_percentile), bf1)
    y1q, c1 = act1(torch.relu(y1), running, act_percentile, return_codes=True)
    wf2, bf2 = fold_bn(w2, None, *bn2, eps)
    y2 = F.conv2d(y1q, weight_fake_quant(wf2, w_bits, wt_percentile), bf2, 1, 1, 1, w2.shape[0])
    y2q, c2 = act3(torch.relu(y2), running, act_percentile, return_codes=True)
    out = F.conv2d(y2q, weight_fake_quant(w3, w_bits, wt_percentile), b3)
    return {"y1": y1, "y1q": y1q, "y1_codes": c1, "y2": y2, "y2q": y2q, "y2_codes": c2, "out": out}


# ---- backbone unit -----------------------------------------------------

def channel_shuffle(x, groups=2):
    """lib/models/networks/shufflenetv2_dcn.py:43-54."""
    n, c, h, w = x.shape
    return x.view(n, groups, c // groups, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


def base_node_w4a8(x, p, acts, shared, stride, w_bits=4, running=True, wt_percentile=False, eps=1e-5):
    """QuantBaseNode.forward (quant_modules.py:880-907).  p: dict of conv weights w1..w5 and BN tuples
    bn1..bn5 (weight, bias, mean, var); acts: dict of QuantActState act1, act2 (, act4); shared: the
    layer's block-output QuantActState.  Order of the shared QuantAct's updates: branch 1, then branch 2."""
    def convbn(t, w, bn, stride_=1, groups=1, pad=0):
        wf, bf = fold_bn(w, None, *bn, eps)
        return F.conv2d(t, weight_fake_quant(wf, w_bits, wt_percentile), bf, stride_, pad, 1, groups)
    if stride == 1:
        half = x.shape[1] // 2
        x1, x2 = x[:, :half], x[:, half:]
    else:
        x1 = acts["act4"](convbn(x, p["w4"], p["bn4"], 2, x.shape[1], 1), running)
        x1 = shared(torch.relu(convbn(x1, p["w5"], p["bn5"])), running)
        x2 = x
    x2 = acts["act1"](torch.relu(convbn(x2, p["w1"], p["bn1"])), running)
    x2 = acts["act2"](convbn(x2, p["w2"], p["bn2"], stride, x2.shape[1], 1), running)
    x2 = shared(torch.relu(convbn(x2, p["w3"], p["bn3"])), running)
    return channel_shuffle(torch.cat((x1, x2), 1), 2)
