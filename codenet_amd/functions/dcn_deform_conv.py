"""Autograd boundary of the deformable-convolution hot path: same public names and call
signatures as the reference's lib/models/external/functions/dcn_deform_conv.py
(``DeformConvFunction`` :11-110, ``ModulatedDeformConvFunction`` :113-182, ``deform_conv`` /
``modulated_deform_conv`` :185-186), running on the HIP library through the
``_ext.dcn.dcn_deform_conv_cuda`` shim.

Differences from the reference, all supersets:
  * no `im2col_step` divisibility assert (there is no im2col buffer);
  * backward returns one gradient slot per forward input (the reference returns 8 for 9);
  * HIP launch errors raise instead of being printed and swallowed.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .._ext.dcn import dcn_deform_conv_cuda

__all__ = ["DeformConvFunction", "ModulatedDeformConvFunction", "deform_conv",
           "modulated_deform_conv"]


def conv_out_shape(x, weight, stride, padding, dilation):
    """[N, C_out, H_out, W_out] of a (deformable) convolution -- the one formula both functions and the native
    library use (dcn_deform_conv_cuda.cpp:187-190): out = (in + 2 p - (d (k - 1) + 1)) // s + 1 per axis."""
    spatial = [(i + 2 * p - (d * (k - 1) + 1)) // s + 1
               for i, k, s, p, d in zip(x.shape[2:], weight.shape[2:], stride, padding, dilation)]
    if min(spatial) <= 0:
        raise ValueError("convolution input is too small (output would be {})".format(
            "x".join(str(v) for v in [x.shape[0], weight.shape[0]] + spatial)))
    return (x.shape[0], weight.shape[0], *spatial)


def _wh_first(weight, stride, padding, dilation):
    """The eight geometry ints of the three deform_conv_*_cuda entry points, W before H
    (dcn_deform_conv_cuda.cpp:151-156)."""
    return (weight.size(3), weight.size(2), stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0])


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError(
                "Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(input, offset, weight)
        if not input.is_cuda:
            raise NotImplementedError  # as the reference: no CPU path
        output = input.new_empty(conv_out_shape(input, weight, ctx.stride, ctx.padding, ctx.dilation))
        empty = input.new_empty(0)
        dcn_deform_conv_cuda.deform_conv_forward_cuda(
            input, weight, offset, output, empty, empty, *_wh_first(weight, ctx.stride, ctx.padding, ctx.dilation),
            ctx.groups, ctx.deformable_groups, min(im2col_step, input.shape[0]))
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        empty = input.new_empty(0)
        step = min(ctx.im2col_step, input.shape[0])
        geom = _wh_first(weight, ctx.stride, ctx.padding, ctx.dilation) + (ctx.groups, ctx.deformable_groups)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            dcn_deform_conv_cuda.deform_conv_backward_input_cuda(
                input, offset, grad_output, grad_input, grad_offset, weight, empty, *geom, step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)      # the native call accumulates into it (cpp:456-462)
            dcn_deform_conv_cuda.deform_conv_backward_parameters_cuda(
                input, offset, grad_output, grad_weight, empty, empty, *geom, 1, step)
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # placeholder, never read
        if not input.is_cuda:
            raise NotImplementedError
        if weight.requires_grad or mask.requires_grad or offset.requires_grad \
                or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        # (the modulated variant takes int, not pair, stride / padding / dilation: functions/...:127-129)
        output = input.new_empty(conv_out_shape(input, weight, _pair(stride), _pair(padding), _pair(dilation)))
        empty = input.new_empty(0)
        dcn_deform_conv_cuda.modulated_deform_conv_cuda_forward(
            input, weight, bias, empty, offset, mask, output, empty, *ModulatedDeformConvFunction._hw_first(ctx, weight),
            ctx.with_bias)
        return output

    @staticmethod
    def _hw_first(ctx, weight):
        """Geometry ints of the two modulated entry points, H before W (dcn_deform_conv_cuda.cpp:486-492)."""
        return (weight.shape[2], weight.shape[3], ctx.stride, ctx.stride, ctx.padding, ctx.padding, ctx.dilation,
                ctx.dilation, ctx.groups, ctx.deformable_groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_input = torch.zeros_like(input)
        grad_offset = torch.zeros_like(offset)
        grad_mask = torch.zeros_like(mask)
        grad_weight = torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        empty = input.new_empty(0)
        dcn_deform_conv_cuda.modulated_deform_conv_cuda_backward(
            input, weight, bias, empty, offset, mask, empty, grad_input, grad_weight, grad_bias,
            grad_offset, grad_mask, grad_output, *ModulatedDeformConvFunction._hw_first(ctx, weight), ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None,
                None, None)


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply
