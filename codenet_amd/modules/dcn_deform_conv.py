"""nn.Module operator API of the deformable-convolution hot path -- the class names, constructor
signatures, attribute and parameter names (hence checkpoint keys) of the
reference's lib/models/external/modules/dcn_deform_conv.py, on the MI355X HIP library.

The CoDeNet operator is ``DeformConvWithOffsetScaleBoundPositive`` (reference :285-330).  On a GPU
float32 input in inference it runs three hand-written kernels (scale prediction, LDS-staged
bilinear-gather depthwise conv, f32-MFMA pointwise conv) and never builds the 18-channel offset
tensor; under autograd the gather runs on the same forward kernel with a dedicated backward
kernel and the two 1x1 convolutions stay on PyTorch so their gradients come from autograd.
Everything else goes through the generic ``deform_conv`` / ``modulated_deform_conv``.
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from .. import ops
from ..functions.dcn_deform_conv import deform_conv, modulated_deform_conv

__all__ = [
    "DeformConv", "DeformConvPack", "DeformConvPack1x1", "DeformConvPackDW",
    "ModulatedDeformConv", "ModulatedDeformConvPack", "DeformConvWithOffsetBound",
    "DeformConvWithOffsetRound", "DeformConvWithOffsetScale", "DeformConvWithOffsetScaleBound",
    "DeformConvWithOffsetScaleBoundPositive", "ModulatedDeformConvWithOffsetScaleBoundPositive",
    "ModulatedDeformConvWithOffset1x1ScaleBoundPositive",
]


def _uniform_fan_in_(weight, in_channels, kernel_size):
    """weight ~ U(-1/sqrt(C*kH*kW), +...)  (reference reset_parameters :49-54, :163-170)."""
    fan = in_channels * kernel_size[0] * kernel_size[1]
    bound = 1.0 / math.sqrt(fan)
    with torch.no_grad():
        weight.uniform_(-bound, bound)


def make_anchor_offset():
    """The fixed 3x3 anchor (dy,dx) for k = 0..8 as a [1,18,1,1] tensor (reference :319-321):
    offset = anchor * (s - 1) dilates the regular 3x3 grid by s."""
    pairs = [(float(dy), float(dx)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    return torch.tensor([v for pr in pairs for v in pr], dtype=torch.float32).view(1, 18, 1, 1)


def _zero_init(conv, bias_value=0.0):
    with torch.no_grad():
        conv.weight.zero_()
        if conv.bias is not None:
            conv.bias.fill_(bias_value)


class DeformConv(nn.Module):
    """Deformable convolution v1 (reference :12-58).  forward(x, offset[N, dg*2*kH*kW, Ho, Wo])."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, \
            "in_channels {} cannot be divisible by groups {}".format(in_channels, groups)
        assert out_channels % groups == 0, \
            "out_channels {} cannot be divisible by groups {}".format(out_channels, groups)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in_(self.weight, self.in_channels, self.kernel_size)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation,
                           self.groups, self.deformable_groups)

    def is_codenet_depthwise(self):
        """True when this is the depthwise 3x3 / stride 1 / pad 1 / dil 1 / dg 1 configuration the
        CoDeNet fast-path kernels implement."""
        return (self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.deformable_groups == 1
                and self.groups == self.in_channels == self.out_channels)


class _PackBase(DeformConv):
    """DeformConv + an internal conv that predicts the offsets (zero-initialised)."""

    def forward(self, x):  # pylint: disable=arguments-differ
        return super().forward(x, self._offsets(x))


class DeformConvPack(_PackBase):
    """Offsets from a kxk conv with the op's own stride/padding (reference :61-84)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(
            self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
            kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
            bias=True)
        self.init_offset()

    def init_offset(self):
        _zero_init(self.conv_offset)

    def _offsets(self, x):
        return self.conv_offset(x)


class DeformConvPack1x1(_PackBase):
    """Offsets from a 1x1 conv (reference :87-110)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(
            self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
            kernel_size=1, stride=1, padding=0, bias=True)
        self.init_offset()

    def init_offset(self):
        _zero_init(self.conv_offset)

    def _offsets(self, x):
        return self.conv_offset(x)


class DeformConvPackDW(_PackBase):
    """Offsets from a depthwise 3x3 + pointwise pair (reference :113-129)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        inp = int(self.in_channels)
        self.conv_dw = nn.Conv2d(inp, inp, 3, 1, 1, groups=inp, bias=True)
        self.conv_pw = nn.Conv2d(inp, int(self.deformable_groups * 18), 1, 1, 0, bias=True)
        _zero_init(self.conv_pw)

    def _offsets(self, x):
        return self.conv_pw(self.conv_dw(x))


class ModulatedDeformConv(nn.Module):
    """Deformable convolution v2 (reference :132-175).  forward(x, offset, mask); int stride /
    padding / dilation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in_(self.weight, self.in_channels, self.kernel_size)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups,
                                     self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """Offsets and sigmoid mask from one kxk conv (reference :178-205)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset_mask = nn.Conv2d(
            self.in_channels, self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
            kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
            bias=True)
        self.init_offset()

    def init_offset(self):
        _zero_init(self.conv_offset_mask)

    def forward(self, x):  # pylint: disable=arguments-differ
        o1, o2, mask = torch.chunk(self.conv_offset_mask(x), 3, dim=1)
        return super().forward(x, torch.cat((o1, o2), dim=1), torch.sigmoid(mask))


# ---- offset-policy wrappers around DeformConv (reference :208-384) -----------------------------

def _offset_conv3x3(in_channels, kernel_size, deformable_groups):
    conv = nn.Conv2d(in_channels, kernel_size * kernel_size * 2 * deformable_groups, kernel_size=3,
                     stride=1, padding=1, bias=True)
    _zero_init(conv)
    return conv


class DeformConvWithOffsetBound(nn.Module):
    """18-channel offsets from a 3x3 conv, clamped to +-offset_bound (reference :208-221)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False, offset_bound=8):
        super().__init__()
        self.conv_offset = _offset_conv3x3(in_channels, kernel_size, deformable_groups)
        self.conv_bound = nn.Hardtanh(min_val=-offset_bound, max_val=offset_bound, inplace=True)
        self.conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=groups,
                               deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        return self.conv(x, self.conv_bound(self.conv_offset(x)))


class DeformConvWithOffsetRound(nn.Module):
    """18-channel offsets from a 3x3 conv, rounded to integers (reference :224-235)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super().__init__()
        self.conv_offset = _offset_conv3x3(in_channels, kernel_size, deformable_groups)
        self.conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=groups,
                               deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        return self.conv(x, self.conv_offset(x).round_())


class _ScaleOffset(nn.Module):
    """Shared plumbing of the `*OffsetScale*` family: a conv predicting one scale per pixel
    (weight 0, bias 1 => s == 1 == regular grid at init) and the fixed anchor."""

    def _make_scale(self, in_channels, deformable_groups, ksize, stride=1):
        conv = nn.Conv2d(in_channels, deformable_groups, kernel_size=ksize, stride=stride,
                         padding=ksize // 2, bias=True)
        _zero_init(conv, bias_value=1.0)
        self.anchor_offset = make_anchor_offset()
        return conv

    def _offsets_from(self, s):
        return self.anchor_offset.to(s.device) * (s - 1)


class DeformConvWithOffsetScale(_ScaleOffset):
    """Unbounded scale from a 3x3 conv (reference :238-255)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super().__init__()
        self.conv_scale = self._make_scale(in_channels, deformable_groups, 3)
        self.conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=groups,
                               deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        return self.conv(x, self._offsets_from(self.conv_scale(x)))


class DeformConvWithOffsetScaleBound(_ScaleOffset):
    """Scale from a 3x3 conv clamped to +-offset_bound (reference :258-282)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False, offset_bound=8):
        super().__init__()
        self.conv_scale = self._make_scale(in_channels, deformable_groups, 3)
        self.conv_bound = nn.Hardtanh(min_val=-offset_bound, max_val=offset_bound, inplace=True)
        self.conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=groups,
                               deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        return self.conv(x, self._offsets_from(self.conv_bound(self.conv_scale(x))))


class DeformConvWithOffsetScaleBoundPositive(_ScaleOffset):
    """The co-designed deformable convolution of CoDeNet (reference :285-330):

        s = Hardtanh(1 - offset_bound, offset_bound)(conv_scale(x))       1x1, C -> 1, bias
        d = depthwise 3x3 deformable conv of x, stencil dilated per pixel by s
        y = conv_channel(d)                                               1x1, C -> C_out, no bias

    `groups`, `hidden_state` and `BN_MOMENTUM` are accepted and ignored, as in the reference
    (:290-309); the depthwise conv always has groups = in_channels.
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False, offset_bound=8, hidden_state=64,
                 BN_MOMENTUM=0.1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.conv_scale = nn.Conv2d(in_channels, deformable_groups, kernel_size=1, stride=stride,
                                    padding=0, bias=True)
        _zero_init(self.conv_scale, bias_value=1.0)
        self.anchor_offset = make_anchor_offset()
        self.conv_bound = nn.Hardtanh(min_val=-offset_bound + 1, max_val=offset_bound, inplace=True)
        self.conv = DeformConv(in_channels, in_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=in_channels,
                               deformable_groups=deformable_groups, bias=bias)
        if in_channels != out_channels:
            self.conv_channel = nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False)
            nn.init.kaiming_normal_(self.conv_channel.weight, nonlinearity="relu")

    def _fast_path_ok(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and self.conv.is_codenet_depthwise() and self.conv_scale.out_channels == 1
                and self.conv_scale.stride == (1, 1))

    def forward(self, x):
        if not self._fast_path_ok(x):
            d = self.conv(x, self._offsets_from(self.conv_bound(self.conv_scale(x))))
            return self.conv_channel(d) if self.in_channels != self.out_channels else d
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            # training: scale 1x1, gather and pointwise 1x1 forward AND backward on the HIP kernels
            from ..functions.codenet_stage import codenet_stage
            pw = self.conv_channel if self.in_channels != self.out_channels else None
            return codenet_stage(x, self.conv_scale.weight, self.conv_scale.bias, self.conv.weight,
                                 pw.weight if pw is not None else None, pw.bias if pw is not None else None,
                                 self.conv_bound.min_val, self.conv_bound.max_val)
        s = ops.codenet_scale(x, self.conv_scale.weight, self.conv_scale.bias,
                              self.conv_bound.min_val, self.conv_bound.max_val)
        d = ops.codenet_dw(x, s, self.conv.weight)
        if self.in_channels == self.out_channels:
            return d
        return ops.codenet_pointwise(d, self.conv_channel.weight, self.conv_channel.bias)


class _ModulatedScaleBase(_ScaleOffset):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                 deformable_groups, bias, offset_bound, pred_ksize):
        super().__init__()
        self.conv_mask = nn.Conv2d(in_channels, deformable_groups * 9, kernel_size=pred_ksize,
                                   stride=1, padding=pred_ksize // 2, bias=True)
        self.conv_scale = self._make_scale(in_channels, deformable_groups, pred_ksize)
        self.conv_bound = nn.Hardtanh(min_val=0, max_val=offset_bound, inplace=True)
        self.conv = ModulatedDeformConv(in_channels, out_channels, kernel_size=kernel_size,
                                        stride=stride, padding=padding, dilation=dilation,
                                        groups=groups, deformable_groups=deformable_groups,
                                        bias=bias)

    def forward(self, x):
        m = self.conv_mask(x)
        s = self.conv_bound(self.conv_scale(x))
        return self.conv(x, self._offsets_from(s), m)


class ModulatedDeformConvWithOffsetScaleBoundPositive(_ModulatedScaleBase):
    """Modulated variant, 3x3 predictors, scale in [0, offset_bound] (reference :333-357)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False, offset_bound=8):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         deformable_groups, bias, offset_bound, pred_ksize=3)


class ModulatedDeformConvWithOffset1x1ScaleBoundPositive(_ModulatedScaleBase):
    """Modulated variant with 1x1 predictors (reference :360-384)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1,
                 groups=1, deformable_groups=1, bias=False, offset_bound=8):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         deformable_groups, bias, offset_bound, pred_ksize=1)
