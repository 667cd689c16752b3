"""Fake-quantisation nn.Modules of the CoDeNet deform stage with the class names, ``set_param``
protocol, sub-module / buffer names (checkpoint keys) and arithmetic of
the reference's portable_quantizer/quant_modules.py:

    QuantAct :163-225          Quant_Conv2d :228-321        QuantBnConv2d :324-419
    QuantDeformConv2d :422-517 QuantDeformConvWithOffsetScaleBoundPositive :621-671
    and, for the class surface (`from portable_quantizer.quant_modules import ...`; none of them is reached by the
    shufflenetv2 configs): QuantLinear :23-160, QuantBnDeformConv2d :520-617,
    QuantDeformConvWithOffsetScaleBoundPositiveBn :674-720, QuantSflUnit :723-806, QuantBaseNodeDeform :910-1010

MI355X specifics
  * QuantAct on a GPU tensor is one C-ABI call (batch min/max reduction, range tracking with
    the reference's "+=" / EMA semantics, quantise + de-quantise) with no host round trip.
  * Weight fake-quantisation (tiny tensors) is derived with torch ops in the reference's fp32
    expression order; in inference the result is cached per weight version instead of being
    recomputed every forward (reference: quant_modules.py:278-321,364-419,473-517).
  * The compound stage module runs scale -> QuantAct -> LDS gather/depthwise -> QuantAct ->
    f32-MFMA pointwise (+ folded BN bias) on the HIP kernels; under autograd it composes the
    same sub-modules so gradients are the reference's (straight-through estimators).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Module, Parameter

from .. import ops
from ..functions.dcn_deform_conv import deform_conv
from .quantization_utils.quant_utils import (AsymmetricQuantFunction, SymmetricQuantFunction,
                                             get_percentile_min_max)

__all__ = ["QuantAct", "Quant_Conv2d", "QuantBnConv2d", "QuantDeformConv2d",
           "QuantDeformConvWithOffsetScaleBoundPositive", "QuantBaseNode", "QuantDepthwiseNode",
           "channel_shuffle", "QuantLinear", "QuantBnDeformConv2d",
           "QuantDeformConvWithOffsetScaleBoundPositiveBn", "QuantSflUnit", "QuantBaseNodeDeform"]


def _quant_function(mode):
    if mode == "symmetric":
        return SymmetricQuantFunction.apply
    if mode == "asymmetric":
        return AsymmetricQuantFunction.apply
    raise ValueError("unknown quant mode: {}".format(mode))


def _channel_range(w2d, percentile):
    """Per-output-channel (w_min, w_max): plain min/max, or with --wt-percentile the
    0.1 / 99.9 percentile k-th values (0.95*min/max when a channel has < 10 weights, e.g. every
    depthwise 3x3) -- reference :281-300."""
    if not percentile:
        return w2d.min(dim=1).values, w2d.max(dim=1).values
    length = w2d.shape[1]
    if length < 10:
        return w2d.min(dim=1).values * 0.95, w2d.max(dim=1).values * 0.95
    lo = math.ceil(length * 0.1 * 0.01)
    hi = math.ceil(length * 99.9 * 0.01)
    return (torch.kthvalue(w2d, k=lo, dim=1).values, torch.kthvalue(w2d, k=hi, dim=1).values)


def refresh_in_place(old, new):
    """Derived tensors are cached per source version.  When a source changes, the new values are written
    INTO the previously cached tensors wherever shape / dtype / device still match, so raw pointers held by a
    captured HIP graph stay valid (and the graph sees the new weights); otherwise the entry is replaced."""
    if isinstance(new, torch.Tensor):
        if (isinstance(old, torch.Tensor) and old.shape == new.shape and old.dtype == new.dtype
                and old.device == new.device and old.is_contiguous() and old.data_ptr() != new.data_ptr()
                and not old.requires_grad):
            old.copy_(new)
            return old
        return new
    if isinstance(new, (tuple, list)):
        olds = old if isinstance(old, (tuple, list)) and len(old) == len(new) else [None] * len(new)
        return type(new)(refresh_in_place(o, n) for o, n in zip(olds, new))
    return new


class _WeightQuantizer:
    """Shared weight fake-quantisation of the three Quant*Conv modules + an inference cache."""

    def _init_weight_quant(self, weight_bit, bias_bit, full_precision_flag, quant_mode, per_channel,
                           weight_percentile):
        self.full_precision_flag = full_precision_flag
        self.weight_bit = weight_bit
        self.quant_mode = quant_mode
        self.per_channel = per_channel
        self.weight_percentile = weight_percentile
        self.bias_bit = bias_bit
        self.quantize_bias = bias_bit is not None
        self.weight_function = _quant_function(quant_mode)
        self._wq_cache = None

    def _fake_quant_weight(self, w, out_channels):
        if self.full_precision_flag:
            return w
        from ..functions import codenet_stage as CS
        if CS.native_weight_prep_ok(w, self):      # QAT step on the GPU: one launch, bit-identical values
            return CS.FakeQuantWeight.apply(w, self.weight_bit, bool(self.weight_percentile))
        if self.per_channel:
            if self.quantize_bias:
                raise NotImplementedError("channel-wise quantize bias is not supported")
            w_min, w_max = _channel_range(w.data.contiguous().view(out_channels, -1),
                                          self.weight_percentile)
        else:
            if self.quantize_bias:
                raise NotImplementedError("bias quantisation is outside the hot path")
            if self.weight_percentile:
                w_min, w_max = get_percentile_min_max(w.view(-1), 0.1, 99.9, output_tensor=True)
            else:
                w_min, w_max = w.data.min(), w.data.max()
        return self.weight_function(w, self.weight_bit, w_min, w_max, self.per_channel,
                                    self.weight_percentile)

    def _int8_ok(self, kernel_size, groups):
        return not (self.full_precision_flag or not self.per_channel or self.weight_bit > 4
                    or self.quant_mode != "symmetric" or tuple(kernel_size) != (1, 1) or groups != 1)

    def _int8_codes(self, w):
        """Integer form of per-channel symmetric <= 4-bit 1x1 weights for the int8-MFMA pointwise
        kernel: (codes int8 [Co, round_up(C,64)] zero padded, scale fp32 [Co] with w' = codes / scale,
        column sums int32 [Co]).  Same expressions as SymmetricQuantFunction.forward
        (quant_utils.py:207-225)."""
        co = w.shape[0]
        w_min, w_max = _channel_range(w.data.contiguous().view(co, -1), self.weight_percentile)
        mag = torch.max(torch.stack([w_min.abs(), w_max.abs()], dim=1), dim=1).values
        n = 2 ** (self.weight_bit - 1) - 1
        scale = n / torch.clamp(mag, min=1e-10)
        q = torch.clamp(torch.round(scale.view(-1, 1, 1, 1) * w), -(n + 1), n).view(co, -1)
        cpad = (q.shape[1] + 63) // 64 * 64
        codes = torch.zeros(co, cpad, dtype=torch.int8, device=q.device)
        codes[:, :q.shape[1]] = q.to(torch.int8)
        return codes.contiguous(), scale.contiguous(), q.sum(dim=1).to(torch.int32).contiguous()

    def _cached_i8(self, key_tensors, compute):
        if torch.is_grad_enabled():
            with torch.no_grad():
                return compute()
        key = tuple((t.data_ptr(), t._version, t.device) for t in key_tensors)
        cache = getattr(self, "_i8_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                self._i8_cache = (key, refresh_in_place(cache[1] if cache else None, compute()))
        return self._i8_cache[1]

    def _cached(self, key_tensors, compute):
        """In inference (no grad) reuse the derived tensors until a source tensor changes."""
        if torch.is_grad_enabled():
            return compute()
        key = tuple((t.data_ptr(), t._version, t.device) for t in key_tensors)
        if self._wq_cache is None or self._wq_cache[0] != key:
            with torch.no_grad():
                old = self._wq_cache[1] if self._wq_cache else None
                self._wq_cache = (key, refresh_in_place(old, compute()))
        return self._wq_cache[1]

    def invalidate(self):
        """Drop the derived-weight caches (needed only after writes that bypass the version counter, i.e.
        through ``.data``)."""
        self._wq_cache = None
        self._i8_cache = None


def allreduce_extremes(bmin, bmax):
    """{min, max} over all ranks as a 2-element tensor, from ONE MAX all-reduce of {-min, max, nan flag}.  A NaN on any
    rank makes both ends NaN on every rank, as x.min() / x.max() of the union would be (quant_modules.py:203-219 of the
    reference on the whole batch) -- the backends' MAX does not propagate it by itself (gloo drops it).  gloo reduces on
    the host (the tests on one GPU)."""
    import torch.distributed as dist
    t = torch.cat((-bmin.reshape(1), bmax.reshape(1), bmin.new_zeros(1)))
    bad = torch.isnan(t[:2]).any()
    t = torch.where(torch.isnan(t), torch.full_like(t, float("-inf")), t)
    t[2] = bad.to(t.dtype)
    if t.is_cuda and dist.get_backend() == "gloo":
        th = t.cpu()
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        t.copy_(th)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)         # RCCL over xGMI: 12 bytes on the current stream
    out = torch.stack((-t[0], t[1]))
    return torch.where(t[2] > 0, torch.full_like(out, float("nan")), out)


class QuantAct(Module):
    """Activation fake-quantiser with tracked range (reference :163-225).  `running_stat` stays
    True in eval() exactly as in the reference (nothing ever clears it); set it to False for
    frozen-range inference."""

    def __init__(self, activation_bit, momentum=0.99, full_precision_flag=False, running_stat=True,
                 quant_mode="symmetric", show_flag=False, percentile=False):
        super().__init__()
        self.activation_bit = activation_bit
        self.momentum = momentum
        self.full_precision_flag = full_precision_flag
        self.running_stat = running_stat
        self.quant_mode = quant_mode
        self.show_flag = show_flag
        self.percentile = percentile
        self.register_buffer("x_min", torch.zeros(1))
        self.register_buffer("x_max", torch.zeros(1))
        self.act_function = _quant_function(quant_mode)
        self._state = None
        # multi-process parity mode (collective 3; not in the reference, which has one process):
        # the batch extremes are all-reduced (MIN / MAX) over the process group before the range update, so that R
        # ranks with B images each track exactly the ranges of ONE process running all R*B images.  Off by default
        # (per-rank ranges = R independent reference runs); pipeline.set_global_range(model, True) turns it on.
        self.global_range = False

    def __repr__(self):
        return "{0}(activation_bit={1}, full_precision_flag={2}, Act_min: {3:.2f}, " \
               "Act_max: {4:.2f})".format(self.__class__.__name__, self.activation_bit,
                                          self.full_precision_flag, self.x_min.item(),
                                          self.x_max.item())

    def _device_state(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:     # "cuda" == the current device
            device = torch.device("cuda", torch.cuda.current_device())
        if self._state is None or self._state.device != device:
            self._state = ops.quantact_state(device)
        return self._state

    def _global_extremes(self, x):
        """(min, max) of x over ALL ranks as 1-element tensors, or None when the mode is off / there is one rank."""
        import torch.distributed as dist
        if not (getattr(self, "global_range", False) and dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1):
            return None
        if self.percentile:
            raise NotImplementedError("QuantAct: global_range with percentile statistics is not defined "
                                      "(a percentile of the union is not a function of the ranks' percentiles)")
        xd = x.detach()
        t = allreduce_extremes(xd.min().reshape(1), xd.max().reshape(1))
        return t[0:1].contiguous(), t[1:2].contiguous()

    def _native_ok(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and self.quant_mode == "asymmetric"
                and not self.full_precision_flag
                and not (torch.is_grad_enabled() and x.requires_grad))

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad
                and self.quant_mode == "asymmetric" and not self.full_precision_flag and not self.percentile
                and not getattr(self, "global_range", False)):
            # training: the same device kernel, straight-through backward (quant_utils.py:202-204)
            from ..functions.codenet_stage import QuantActSTE
            return QuantActSTE.apply(x, self)
        if self._native_ok(x):
            bmin = bmax = None
            if self.running_stat and self.percentile:
                bmin, bmax = get_percentile_min_max(x.detach().view(-1), 0.1, 99.9,
                                                    output_tensor=True)
                bmin, bmax = bmin.reshape(1).contiguous(), bmax.reshape(1).contiguous()
            if self.running_stat:
                glob = self._global_extremes(x)
                if glob is not None:
                    bmin, bmax = glob
            out, _ = ops.quantact_forward(x, self.x_min, self.x_max, self._device_state(x.device),
                                          bits=self.activation_bit, momentum=self.momentum,
                                          running=self.running_stat, batch_min=bmin, batch_max=bmax)
            return out
        # autograd / CPU tensors / symmetric mode: the reference's torch composition
        if self.running_stat:
            if not self.percentile:
                x_min, x_max = x.data.min(), x.data.max()
            else:
                x_min, x_max = get_percentile_min_max(x.detach().view(-1), 0.1, 99.9,
                                                      output_tensor=True)
            glob = self._global_extremes(x)
            if glob is not None:
                x_min, x_max = glob[0][0], glob[1][0]
            if self.x_min == self.x_max:     # "initialisation": += (reference :211-213)
                self.x_min += x_min
                self.x_max += x_max
            else:                            # EMA (reference :217-219)
                self.x_min += (self.momentum - 1.) * self.x_min + (1. - self.momentum) * x_min
                self.x_max += (self.momentum - 1.) * self.x_max + (1. - self.momentum) * x_max
        if self.full_precision_flag:
            return x
        return self.act_function(x, self.activation_bit, self.x_min, self.x_max)


class Quant_Conv2d(Module, _WeightQuantizer):
    """Conv2d with fake-quantised weights, fp32 bias (reference :228-321)."""

    def __init__(self, weight_bit, bias_bit=None, full_precision_flag=False, quant_mode="symmetric",
                 per_channel=False, weight_percentile=False):
        super().__init__()
        self.momentum = 0.99
        self._init_weight_quant(weight_bit, bias_bit, full_precision_flag, quant_mode, per_channel,
                                weight_percentile)

    def set_param(self, conv):
        for name in ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation",
                     "groups"):
            setattr(self, name, getattr(conv, name))
        self.weight = Parameter(conv.weight.data.clone())
        self.bias = Parameter(conv.bias.data.clone()) if getattr(conv, "bias", None) is not None \
            else None

    def quantized_weight(self):
        return self._cached((self.weight,),
                            lambda: self._fake_quant_weight(self.weight, self.out_channels))

    def int8_form(self):
        """Integer form of the fake-quantised 1x1 weights (see _int8_codes); None when this layer is
        not per-channel symmetric <= 4 bit."""
        if not self._int8_ok(self.kernel_size, self.groups):
            return None
        return self._cached_i8((self.weight,), lambda: self._int8_codes(self.weight))

    def forward(self, x):
        return F.conv2d(x, self.quantized_weight(), self.bias, self.stride, self.padding,
                        self.dilation, self.groups)


class QuantBnConv2d(Module, _WeightQuantizer):
    """Conv2d + BatchNorm folded from the BN running statistics, then weight fake-quantisation;
    the folded bias stays fp32 (reference :324-419).  The BN module is never *called*."""

    def __init__(self, weight_bit, bias_bit=None, full_precision_flag=False, running_stat=True,
                 quant_mode="asymmetric", per_channel=False, weight_percentile=False):
        super().__init__()
        self.running_stat = running_stat
        self._init_weight_quant(weight_bit, bias_bit, full_precision_flag, quant_mode, per_channel,
                                weight_percentile)

    def set_param(self, conv, bn):
        self.conv = conv
        self.bn = bn

    def folded(self):
        """(fake-quantised folded weight, fp32 folded bias)."""
        def compute():
            from ..functions import codenet_stage as CS
            if CS.native_weight_prep_ok(self.conv.weight, self):      # QAT step on the GPU: fold + quantiser fused
                return CS.FoldFakeQuantWeight.apply(self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias,
                                                    self.bn.running_mean, self.bn.running_var, self.bn.eps,
                                                    self.weight_bit, bool(self.weight_percentile))
            running_std = torch.sqrt(self.bn.running_var + self.bn.eps)
            scale_factor = self.bn.weight / running_std
            w = self.conv.weight * scale_factor.reshape([self.conv.out_channels, 1, 1, 1])
            b = self.conv.bias if self.conv.bias is not None \
                else torch.zeros_like(self.bn.running_mean)
            b = (b - self.bn.running_mean) * scale_factor + self.bn.bias
            return self._fake_quant_weight(w, self.conv.out_channels), b
        keys = [self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean,
                self.bn.running_var]
        if self.conv.bias is not None:
            keys.append(self.conv.bias)
        return self._cached(tuple(keys), compute)

    def folded_int8(self):
        """Integer form of the folded, fake-quantised 1x1 weights for the int8-MFMA pointwise kernel
        (see _int8_codes); None when this layer is not per-channel symmetric <= 4 bit."""
        if not self._int8_ok(self.conv.kernel_size, self.conv.groups):
            return None

        def compute():
            running_std = torch.sqrt(self.bn.running_var + self.bn.eps)
            scale_factor = self.bn.weight / running_std
            w = self.conv.weight * scale_factor.reshape([self.conv.out_channels, 1, 1, 1])
            return self._int8_codes(w)
        return self._cached_i8((self.conv.weight, self.bn.weight, self.bn.running_var), compute)

    def folded_int8_kblocked(self, columns, offset):
        """(codes + k-blocked copy, scale, column sums): folded_int8() with the codes followed -- at byte `offset` of the
        same int8 buffer -- by the copy [Cpad / 32][columns][32] the streaming int8 pointwise kernel reads
        (include/codenet_dcn.h, CDN_X_WCODES_KB; columns / offset from cdn_codenet_wcodes_kb_columns / _offset).  Cached
        with the codes; the buffer is rewritten in place when the weights change (HIP graphs keep its address)."""
        i8 = self.folded_int8()
        if i8 is None:
            return None
        codes, scale, colsum = i8
        # keyed on the SOURCE tensors, as _cached_i8 is (with grad enabled folded_int8() returns a fresh
        # temporary per call, whose recycled address + version 0 could hit a stale buffer): under grad nothing is cached
        src = (self.conv.weight, self.bn.weight, self.bn.running_var)
        key = tuple((t.data_ptr(), t._version, t.device) for t in src) + (int(columns), int(offset))
        cache = None if torch.is_grad_enabled() else getattr(self, "_kb_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                co, cpad = codes.shape
                full = torch.zeros(int(columns), cpad, dtype=torch.int8, device=codes.device)
                full[:co] = codes
                kb = full.view(int(columns), cpad // 32, 32).permute(1, 0, 2).contiguous()
                buf = torch.zeros(int(offset) + kb.numel(), dtype=torch.int8, device=codes.device)
                buf[:co * cpad] = codes.reshape(-1)
                buf[int(offset):] = kb.reshape(-1)
                if torch.is_grad_enabled():
                    return buf, scale, colsum
                self._kb_cache = (key, refresh_in_place(cache[1] if cache else None, buf))
        return self._kb_cache[1], scale, colsum

    def forward(self, x):
        w, b = self.folded()
        return F.conv2d(x, w, b, self.conv.stride, self.conv.padding, self.conv.dilation,
                        self.conv.groups)


class QuantDeformConv2d(Module, _WeightQuantizer):
    """DeformConv with fake-quantised weights; forward(x, offset) (reference :422-517)."""

    def __init__(self, weight_bit, bias_bit=None, full_precision_flag=False, quant_mode="symmetric",
                 per_channel=False, weight_percentile=False):
        super().__init__()
        self.momentum = 0.99
        self._init_weight_quant(weight_bit, bias_bit, full_precision_flag, quant_mode, per_channel,
                                weight_percentile)

    def set_param(self, conv):
        for name in ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation",
                     "groups", "deformable_groups"):
            setattr(self, name, getattr(conv, name))
        self.weight = Parameter(conv.weight.data.clone())
        self.bias = Parameter(conv.bias.data.clone()) if getattr(conv, "bias", None) is not None \
            else None

    def quantized_weight(self):
        return self._cached((self.weight,),
                            lambda: self._fake_quant_weight(self.weight, self.out_channels))

    def forward(self, x, offset):
        return deform_conv(x, offset, self.quantized_weight(), self.stride, self.padding,
                           self.dilation, self.groups, self.deformable_groups)

    def forward_scaled(self, x, s):
        """Same operator given the per-pixel scale s instead of offset = anchor*(s-1): the
        CoDeNet fast path (LDS gather kernel, no offset tensor)."""
        return ops.codenet_dw(x, s, self.quantized_weight())


class QuantDeformConvWithOffsetScaleBoundPositive(Module):
    """W4A8 counterpart of DeformConvWithOffsetScaleBoundPositive; absorbs the BatchNorm that
    follows the stage into the pointwise conv (reference :621-671)."""

    def __init__(self, weight_bit, act_bit, full_precision_flag=False, bias_bit=None,
                 act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="symmetric",
                 per_channel=False, weight_percentile=False):
        super().__init__()
        self.act_bit = act_bit
        self.weight_bit = weight_bit
        self.bias_bit = bias_bit
        self.quantize_bias = bias_bit is not None
        self.wt_quant_mode = wt_quant_mode
        self.act_quant_mode = act_quant_mode
        self.full_precision_flag = full_precision_flag
        self.act_percentile = act_percentile
        self.per_channel = per_channel
        self.weight_percentile = weight_percentile

    def set_param(self, deform_conv, bn):
        wkw = dict(quant_mode=self.wt_quant_mode, per_channel=self.per_channel,
                   weight_percentile=self.weight_percentile)
        self.quant_conv_scale = Quant_Conv2d(self.weight_bit, **wkw)
        self.quant_conv_scale.set_param(deform_conv.conv_scale)
        self.quant_act = nn.Sequential(deform_conv.conv_bound,
                                       QuantAct(self.act_bit, quant_mode="asymmetric",
                                                percentile=self.act_percentile))
        self.quant_deform_conv = QuantDeformConv2d(self.weight_bit, **wkw)
        self.quant_deform_conv.set_param(deform_conv.conv)
        self.quant_identity_deform = QuantAct(self.act_bit, quant_mode=self.act_quant_mode,
                                              percentile=self.act_percentile)
        self.anchor_offset = deform_conv.anchor_offset.clone()
        self.quant_conv_channel_bn = QuantBnConv2d(self.weight_bit, **wkw)
        self.quant_conv_channel_bn.set_param(deform_conv.conv_channel, bn)

    def _fast_path_ok(self, x):
        dc = self.quant_deform_conv
        cs = self.quant_conv_scale
        pw = self.quant_conv_channel_bn.conv
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and not (torch.is_grad_enabled() and (x.requires_grad or any(
                    p.requires_grad for p in self.parameters())))
                and tuple(dc.kernel_size) == (3, 3) and tuple(dc.stride) == (1, 1)
                and tuple(dc.padding) == (1, 1) and tuple(dc.dilation) == (1, 1)
                and dc.deformable_groups == 1 and dc.groups == dc.in_channels == dc.out_channels
                and cs.out_channels == 1 and tuple(cs.kernel_size) == (1, 1)
                and tuple(cs.stride) == (1, 1) and tuple(pw.kernel_size) == (1, 1)
                and pw.groups == 1 and self.act_quant_mode == "asymmetric")

    def _train_path_ok(self, x):
        """Forward + backward of the whole stage on the HIP kernels (functions/codenet_stage.py)."""
        from ..functions.codenet_stage import native_act_ok
        dc, cs, pw = self.quant_deform_conv, self.quant_conv_scale, self.quant_conv_channel_bn.conv
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and torch.is_grad_enabled()
                and tuple(dc.kernel_size) == (3, 3) and tuple(dc.stride) == (1, 1)
                and tuple(dc.padding) == (1, 1) and tuple(dc.dilation) == (1, 1)
                and dc.deformable_groups == 1 and dc.groups == dc.in_channels == dc.out_channels
                and cs.out_channels == 1 and tuple(cs.kernel_size) == (1, 1) and tuple(cs.stride) == (1, 1)
                and tuple(pw.kernel_size) == (1, 1) and pw.groups == 1 and tuple(pw.stride) == (1, 1)
                and native_act_ok(self.quant_act[1]) and native_act_ok(self.quant_identity_deform))

    def forward(self, x, want_range=False, x_up=False, pre_w=None):
        """pre_w (forward_stage_blocks only): (w_scale_q, w_dw_q, w_pw_q, b_pw) already prepared -- with the other stages'
        in one launch each -- for the native training path; None entries are prepared here.
        want_range (functions/codenet_stage.forward_stage_blocks only): on the native training path return
        (y, per-workgroup {min, max} pairs of y) for the QuantAct of the block behind the stage.  x_up (the same caller,
        training path only): x is the STORED tensor whose nearest x2 up-sampling is the stage's input."""
        if x_up and not self._train_path_ok(x):
            raise NotImplementedError("x_up is a training-path argument of forward_stage_blocks")
        if not x_up and self._fast_path_ok(x):      # (x_up: x is a stored tensor, only the training path reads through the up-sampling)
            bound = self.quant_act[0]
            s_raw = ops.codenet_scale(x, self.quant_conv_scale.quantized_weight(),
                                      self.quant_conv_scale.bias, bound.min_val, bound.max_val)
            s = self.quant_act[1](s_raw)
            d = self.quant_deform_conv.forward_scaled(x, s)
            d_q = self.quant_identity_deform(d)
            w, b = self.quant_conv_channel_bn.folded()
            return ops.codenet_pointwise(d_q, w, b)
        if self._train_path_ok(x):
            # QAT step (config e): the stage as one native autograd function; the weight transformations
            # (fake-quantisation with straight-through gradients, BN fold) are tiny torch ops under autograd
            from ..functions.codenet_stage import codenet_stage
            bound = self.quant_act[0]
            cb = self.quant_conv_channel_bn
            pre_w = pre_w if pre_w is not None else (None, None, None, None)
            w, b = (pre_w[2], pre_w[3]) if pre_w[2] is not None else cb.folded()
            w_sc = pre_w[0] if pre_w[0] is not None else self.quant_conv_scale.quantized_weight()
            w_dw = pre_w[1] if pre_w[1] is not None else self.quant_deform_conv.quantized_weight()
            return codenet_stage(x, w_sc, self.quant_conv_scale.bias, w_dw, w, b, bound.min_val, bound.max_val,
                                 self.quant_act[1], self.quant_identity_deform, want_range, x_up,
                                 cb._int8_ok(cb.conv.kernel_size, cb.conv.groups))
        s = self.quant_act(self.quant_conv_scale(x))
        dc = self.quant_deform_conv
        if (x.is_cuda and x.dtype == torch.float32 and s.shape[1] == 1
                and tuple(dc.kernel_size) == (3, 3) and tuple(dc.stride) == (1, 1)
                and tuple(dc.padding) == (1, 1) and tuple(dc.dilation) == (1, 1)
                and dc.deformable_groups == 1 and dc.groups == dc.in_channels == dc.out_channels):
            d = dc.forward_scaled(x, s)
        else:
            d = self.quant_deform_conv(x, self.anchor_offset.to(x.device) * (s - 1))
        return self.quant_conv_channel_bn(self.quant_identity_deform(d))


# ---- callers either side of the stage -----------------------
# Plain compositions of the modules above; their convolutions run on PyTorch-ROCm.

def channel_shuffle(x, G):
    """lib/models/networks/shufflenetv2_dcn.py:29-34."""
    N, C, H, W = x.size()
    return x.view(N, G, C // G, H, W).transpose(1, 2).contiguous().view(N, C, H, W)


class _CompoundQuant(Module):
    def __init__(self, weight_bit, act_bit, full_precision_flag=False, bias_bit=None,
                 act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="symmetric",
                 per_channel=False, weight_percentile=False):
        super().__init__()
        self.act_bit = act_bit
        self.weight_bit = weight_bit
        self.bias_bit = bias_bit
        self.quantize_bias = bias_bit is not None
        self.wt_quant_mode = wt_quant_mode
        self.act_quant_mode = act_quant_mode
        self.full_precision_flag = full_precision_flag
        self.act_percentile = act_percentile
        self.per_channel = per_channel
        self.weight_percentile = weight_percentile

    def _convbn(self, conv, bn):
        m = QuantBnConv2d(self.weight_bit, quant_mode=self.wt_quant_mode,
                          per_channel=self.per_channel, weight_percentile=self.weight_percentile)
        m.set_param(conv, bn)
        return m

    def _act(self, mode=None):
        return QuantAct(self.act_bit, quant_mode=mode or self.act_quant_mode,
                        percentile=self.act_percentile)


class QuantBaseNode(_CompoundQuant):
    """W4A8 ShuffleNetV2 BaseNode; the block-output QuantAct is shared across the nodes of a
    layer via set_act (reference :809-907)."""

    def set_param(self, base_node):
        self.stride = base_node.stride
        b2 = base_node.b2
        assert type(b2[3]) == nn.Conv2d
        self.quant_convbn1 = self._convbn(b2[0], b2[1])
        self.quant_act1 = self._act("asymmetric")
        self.quant_convbn2 = self._convbn(b2[3], b2[4])
        self.quant_act2 = self._act()
        self.quant_convbn3 = self._convbn(b2[5], b2[6])
        if base_node.stride == 2:
            b1 = base_node.b1
            assert type(b1[0]) == nn.Conv2d
            self.quant_convbn4 = self._convbn(b1[0], b1[1])
            self.quant_act4 = self._act()
            self.quant_convbn5 = self._convbn(b1[2], b1[3])

    def set_act(self, share_quant_act):
        self.quant_act = share_quant_act

    def forward(self, x):
        if self.stride == 1:
            half = x.shape[1] // 2
            x1, x2 = x[:, :half], x[:, half:]
        else:
            x1 = self.quant_act4(self.quant_convbn4(x))
            x1 = self.quant_act(F.relu(self.quant_convbn5(x1)))
            x2 = x
        x2 = self.quant_act1(F.relu(self.quant_convbn1(x2)))
        x2 = self.quant_act2(self.quant_convbn2(x2))
        x2 = self.quant_act(F.relu(self.quant_convbn3(x2)))
        return channel_shuffle(torch.cat((x1, x2), dim=1), 2)


class QuantDepthwiseNode(_CompoundQuant):
    """W4A8 detection head: pw -> ReLU/QuantAct -> dw 3x3 -> ReLU/QuantAct -> pw(+bias)
    (reference :1013-1071)."""

    def set_param(self, head_node):
        assert type(head_node[3]) == nn.Conv2d
        self.quant_convbn1 = self._convbn(head_node[0], head_node[1])
        self.quant_act1 = nn.Sequential(head_node[2], self._act("asymmetric"))
        self.quant_convbn2 = self._convbn(head_node[3], head_node[4])
        self.quant_act3 = nn.Sequential(head_node[5], self._act("asymmetric"))
        self.quant_conv = Quant_Conv2d(self.weight_bit, quant_mode=self.wt_quant_mode,
                                       per_channel=self.per_channel,
                                       weight_percentile=self.weight_percentile)
        self.quant_conv.set_param(head_node[6])

    def forward(self, x):
        x = self.quant_act1(self.quant_convbn1(x))
        x = self.quant_act3(self.quant_convbn2(x))
        return self.quant_conv(x)


# ---- the rest of the reference's class surface ---------------------------------------------------
# Not reached by any shufflenetv2 configuration; kept so that code written against the
# reference's module imports and runs.  Compositions of the modules above; the deformable ones use the
# LDS gather kernel when the geometry is CoDeNet's.

def _codenet_geometry(conv):
    return (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.deformable_groups == 1
            and conv.groups == conv.in_channels == conv.out_channels)


class QuantBnDeformConv2d(Module, _WeightQuantizer):
    """DeformConv + BatchNorm folded from the running statistics, weights fake-quantised after the fold,
    folded bias added in fp32; forward(x, offset) (reference :520-617; its per-tensor --wt-percentile branch
    reads two undefined names, here it uses the 0.1 / 99.9 percentiles of every sibling class)."""

    def __init__(self, weight_bit, bias_bit=None, full_precision_flag=False, running_stat=True,
                 quant_mode="symmetric", per_channel=False, weight_percentile=False):
        super().__init__()
        self.running_stat = running_stat
        self._init_weight_quant(weight_bit, bias_bit, full_precision_flag, quant_mode, per_channel,
                                weight_percentile)

    def set_param(self, conv, bn):
        self.conv = conv
        self.bn = bn

    def folded(self):
        """(fake-quantised folded weight, fp32 folded bias)."""
        def compute():
            running_std = torch.sqrt(self.bn.running_var + self.bn.eps)
            scale_factor = self.bn.weight / running_std
            w = self.conv.weight * scale_factor.reshape([self.conv.out_channels, 1, 1, 1])
            b = getattr(self.conv, "bias", None)
            b = b if b is not None else torch.zeros_like(self.bn.running_mean)
            b = (b - self.bn.running_mean) * scale_factor + self.bn.bias
            return self._fake_quant_weight(w, self.conv.out_channels), b
        keys = [self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        if getattr(self.conv, "bias", None) is not None:
            keys.append(self.conv.bias)
        return self._cached(tuple(keys), compute)

    def forward(self, x, offset):
        w, b = self.folded()
        c = self.conv
        out = deform_conv(x, offset, w, c.stride, c.padding, c.dilation, c.groups, c.deformable_groups)
        return out + b.view(1, -1, 1, 1)

    def forward_scaled(self, x, s):
        """The same operator given the per-pixel scale instead of offset = anchor * (s - 1)."""
        w, b = self.folded()
        return ops.codenet_dw(x, s, w) + b.view(1, -1, 1, 1)


class QuantDeformConvWithOffsetScaleBoundPositiveBn(Module):
    """The CoDeNet operator with the BatchNorm folded into the DEFORMABLE conv (a dense deformable conv, no
    pointwise stage behind it): scale 1x1 -> bound -> QuantAct -> folded deformable conv (reference :674-720)."""

    def __init__(self, weight_bit, act_bit, full_precision_flag=False, bias_bit=None,
                 act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="symmetric",
                 per_channel=False, weight_percentile=False):
        super().__init__()
        self.act_bit = act_bit
        self.weight_bit = weight_bit
        self.bias_bit = bias_bit
        self.quantize_bias = bias_bit is not None
        self.wt_quant_mode = wt_quant_mode
        self.act_quant_mode = act_quant_mode
        self.full_precision_flag = full_precision_flag
        self.act_percentile = act_percentile
        self.per_channel = per_channel
        self.weight_percentile = weight_percentile

    def set_param(self, deform_conv, bn):
        wkw = dict(quant_mode=self.wt_quant_mode, per_channel=self.per_channel,
                   weight_percentile=self.weight_percentile)
        self.quant_conv_scale = Quant_Conv2d(self.weight_bit, **wkw)
        self.quant_conv_scale.set_param(deform_conv.conv_scale)
        self.quant_act = nn.Sequential(deform_conv.conv_bound,
                                       QuantAct(self.act_bit, quant_mode="asymmetric",
                                                percentile=self.act_percentile))
        self.quant_deform_conv_bn = QuantBnDeformConv2d(self.weight_bit, **wkw)
        self.quant_deform_conv_bn.set_param(deform_conv.conv, bn)
        self.anchor_offset = deform_conv.anchor_offset.clone()

    def forward(self, x):
        s = self.quant_act(self.quant_conv_scale(x))
        dc = self.quant_deform_conv_bn
        if x.is_cuda and x.dtype == torch.float32 and s.shape[1] == 1 and _codenet_geometry(dc.conv):
            return dc.forward_scaled(x, s)
        return dc(x, self.anchor_offset.to(x.device) * (s - 1))


class QuantSflUnit(_CompoundQuant):
    """W4A8 form of pytorchcv's ShuffleUnit (duck-typed: compress_conv1 / compress_bn1, dw_conv2 / dw_bn2,
    expand_conv3 / expand_bn3 and, with `downsample`, dw_conv4 / dw_bn4, expand_conv5 / expand_bn5); the unit-output
    QuantAct is shared through set_act (reference :723-806; like the reference, `use_se` / `use_residual` are recorded
    and not applied, and the two depthwise convs never use --wt-percentile)."""

    def __init__(self, weight_bit, act_bit, full_precision_flag=False, bias_bit=None, act_percentile=False,
                 wt_quant_mode="symmetric", act_quant_mode="symmetric", per_channel=False):
        super().__init__(weight_bit, act_bit, full_precision_flag, bias_bit, act_percentile, wt_quant_mode,
                         act_quant_mode, per_channel, False)

    def _unit_convbn(self, conv, bn):
        m = QuantBnConv2d(weight_bit=self.weight_bit, bias_bit=self.bias_bit, quant_mode=self.wt_quant_mode,
                          per_channel=self.per_channel)
        m.set_param(conv, bn)
        return m

    def _unit_act(self, mode):
        return QuantAct(self.act_bit, quant_mode=mode, full_precision_flag=self.full_precision_flag,
                        percentile=self.act_percentile)

    def set_param(self, sfl_unit):
        self.downsample = sfl_unit.downsample
        self.use_se = sfl_unit.use_se
        self.use_residual = sfl_unit.use_residual
        self.quant_compr_convbn1 = self._unit_convbn(sfl_unit.compress_conv1, sfl_unit.compress_bn1)
        self.quant_act1 = nn.Sequential(nn.ReLU(inplace=True), self._unit_act("asymmetric"))
        self.quant_dw_convbn2 = self._unit_convbn(sfl_unit.dw_conv2, sfl_unit.dw_bn2)
        self.quant_act2 = self._unit_act(self.act_quant_mode)
        self.quant_exp_convbn3 = self._unit_convbn(sfl_unit.expand_conv3, sfl_unit.expand_bn3)
        if self.downsample:
            self.quant_dw_convbn4 = self._unit_convbn(sfl_unit.dw_conv4, sfl_unit.dw_bn4)
            self.quant_act4 = self._unit_act(self.act_quant_mode)
            self.quant_exp_convbn5 = self._unit_convbn(sfl_unit.expand_conv5, sfl_unit.expand_bn5)

    def set_act(self, share_quant_act):
        self.quant_act = share_quant_act

    def forward(self, x):
        if self.downsample:
            y1 = self.quant_act4(self.quant_dw_convbn4(x))
            y1 = self.quant_act(self.quant_exp_convbn5(y1))
            x2 = x
        else:
            y1, x2 = torch.chunk(x, chunks=2, dim=1)
        y2 = self.quant_act1(self.quant_compr_convbn1(x2))
        y2 = self.quant_act2(self.quant_dw_convbn2(y2))
        y2 = self.quant_act(self.quant_exp_convbn3(y2))
        return channel_shuffle(torch.cat((y1, y2), dim=1), 2)


class QuantBaseNodeDeform(_CompoundQuant):
    """W4A8 BaseNode whose 3x3 convs are CoDeNet operators (`PoseShuffleNetV2(deform=True)`, shufflenetv2_dcn.py:
    216-230): b2 = [conv, bn, relu, deform, bn, conv, bn, relu], b1 (stride 2) = [deform, bn, conv, bn, relu]
    (reference :910-1010).  The reference class cannot run -- its set_param reads an undefined name (:950) and its
    forward imports a module that does not exist (:984) -- so this follows its text with those two repaired."""

    def _deform(self, op, bn):
        m = QuantDeformConvWithOffsetScaleBoundPositiveBn(
            self.weight_bit, self.act_bit, act_percentile=self.act_percentile, wt_quant_mode=self.wt_quant_mode,
            act_quant_mode=self.act_quant_mode, per_channel=self.per_channel,
            weight_percentile=self.weight_percentile)
        m.set_param(op, bn)
        return m

    def set_param(self, base_node):
        self.stride = base_node.stride
        b2 = base_node.b2
        self.quant_convbn1 = self._convbn(b2[0], b2[1])
        self.quant_act1 = self._act("asymmetric")
        self.quant_convbn2 = self._deform(b2[3], b2[4])
        self.quant_act2 = self._act()
        self.quant_convbn3 = self._convbn(b2[5], b2[6])
        if base_node.stride == 2:
            b1 = base_node.b1
            self.quant_convbn4 = self._deform(b1[0], b1[1])
            self.quant_act4 = self._act()
            self.quant_convbn5 = self._convbn(b1[2], b1[3])

    def set_act(self, share_quant_act):
        self.quant_act = share_quant_act

    def forward(self, x):
        if self.stride == 1:
            half = x.shape[1] // 2
            x1, x2 = x[:, :half], x[:, half:]
        else:
            x1 = self.quant_act4(self.quant_convbn4(x))
            x1 = self.quant_act(F.relu(self.quant_convbn5(x1)))
            x2 = x
        x2 = self.quant_act1(F.relu(self.quant_convbn1(x2)))
        x2 = self.quant_act2(self.quant_convbn2(x2))
        x2 = self.quant_act(F.relu(self.quant_convbn3(x2)))
        return channel_shuffle(torch.cat((x1, x2), dim=1), 2)


class QuantLinear(nn.Linear):
    """Linear layer with fake-quantised weights and an EMA of the weight range in the buffers x_min / x_max
    (reference :23-160: constructor arguments, buffers, `reset_bits` / `reset_alpha`, range statistics per INPUT
    feature when per_channel, groups of input features with group_quantization, `alpha` blending).  The reference's
    forward fails for every configuration (its quantiser reshapes per-channel statistics for 4-D conv weights and
    hands F.linear a 4-D weight; the percentile branches index out of range) -- tests/golden/make_golden.py
    `probe_quant_linear` records that -- so there is no reference output to pin; this class applies the same
    statistics and the same quantiser expressions along the input-feature axis of the 2-D weight."""

    def __init__(self, weight_bit, input_size, output_size, full_precision_flag=False, quant_mode="symmetric",
                 alpha=None, per_channel=True, group_quantization=False, group_number=60,
                 weight_percentile=False):
        super().__init__(input_size, output_size)
        if quant_mode not in ("symmetric", "asymmetric"):
            raise ValueError("unknown quant mode: {}".format(quant_mode))
        self.full_precision_flag = full_precision_flag
        self.weight_bit = weight_bit
        self.alpha = alpha
        self.quant_mode = quant_mode
        self.input_size = input_size
        self.output_size = output_size
        self.momentum = 0.99
        self.register_buffer("x_min", torch.zeros(1))
        self.register_buffer("x_max", torch.zeros(1))
        self.per_channel = per_channel
        self.weight_percentile = weight_percentile
        self.group_quantization = group_quantization
        self.group_number = group_number

    def reset_bits(self, weight_bit=8):
        self.full_precision_flag = False
        self.weight_bit = weight_bit

    def reset_alpha(self, alpha):
        assert 0.0 <= alpha <= 1.0
        self.alpha = alpha

    def extra_repr(self):
        return super().extra_repr() + ", weight_bit={}, full_precision_flag={}".format(
            self.weight_bit, self.full_precision_flag)

    def _range(self):
        w = self.weight.data
        if not self.per_channel:
            if self.weight_percentile:
                return get_percentile_min_max(w.reshape(-1), 0.1, 99.9, output_tensor=True)
            return w.min().expand(1), w.max().expand(1)
        cols = w.transpose(0, 1).contiguous()                 # one row per input feature
        if self.weight_percentile and not self.group_quantization:
            return _channel_range(cols, True)
        w_min, w_max = cols.min(dim=1).values, cols.max(dim=1).values
        if self.group_quantization:
            glen = w_min.numel() // self.group_number
            for i in range(self.group_number):
                sl = slice(i * glen, (i + 1) * glen)
                if self.weight_percentile:
                    lo, hi = get_percentile_min_max(cols[sl].reshape(-1), 0.1, 99.9, output_tensor=True)
                else:
                    lo, hi = w_min[sl].min(), w_max[sl].max()
                w_min[sl], w_max[sl] = lo, hi
        return w_min, w_max

    def _fake_quant(self):
        """The conv modules' quantiser (quant_utils.py:172-229) with the input features as its channel axis."""
        fn = _quant_function(self.quant_mode)
        w4 = self.weight.transpose(0, 1).reshape(self.in_features, self.out_features, 1, 1)
        q = fn(w4, self.weight_bit, self.x_min, self.x_max, self.per_channel, self.weight_percentile)
        return q.reshape(self.in_features, self.out_features).transpose(0, 1)

    def forward(self, x):
        w_min, w_max = self._range()
        if self.x_min.numel() == 1 and bool(self.x_min == self.x_max):        # first call: take the statistics
            self.x_min, self.x_max = w_min.clone(), w_max.clone()
        self.x_min = self.momentum * self.x_min + (1.0 - self.momentum) * w_min
        self.x_max = self.momentum * self.x_max + (1.0 - self.momentum) * w_max
        if self.full_precision_flag:
            assert self.alpha is None
            return F.linear(x, self.weight, self.bias)
        w = self._fake_quant()
        if self.alpha is not None:
            w = self.alpha * w + (1 - self.alpha) * self.weight
        return F.linear(x, w, self.bias)
