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
        output = input.new_empty(
            DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride))
        empty = input.new_empty(0)
        dcn_deform_conv_cuda.deform_conv_forward_cuda(
            input, weight, offset, output, empty, empty, weight.size(3), weight.size(2),
            ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1],
            ctx.dilation[0], ctx.groups, ctx.deformable_groups, min(im2col_step, input.shape[0]))
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
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            dcn_deform_conv_cuda.deform_conv_backward_input_cuda(
                input, offset, grad_output, grad_input, grad_offset, weight, empty,
                weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1],
                ctx.padding[0], ctx.dilation[1], ctx.dilation[0], ctx.groups,
                ctx.deformable_groups, step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            dcn_deform_conv_cuda.deform_conv_backward_parameters_cuda(
                input, offset, grad_output, grad_weight, empty, empty, weight.size(3),
                weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0],
                ctx.dilation[1], ctx.dilation[0], ctx.groups, ctx.deformable_groups, 1, step)
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        channels = weight.size(0)
        output_size = (input.size(0), channels)
        for d in range(input.dim() - 2):
            in_size = input.size(d + 2)
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            output_size += ((in_size + (2 * padding[d]) - kernel) // stride[d] + 1,)
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError("convolution input is too small (output would be {})".format(
                "x".join(map(str, output_size))))
        return output_size


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
        output = input.new_empty(ModulatedDeformConvFunction._infer_shape(ctx, input, weight))
        empty = input.new_empty(0)
        dcn_deform_conv_cuda.modulated_deform_conv_cuda_forward(
            input, weight, bias, empty, offset, mask, output, empty, weight.shape[2],
            weight.shape[3], ctx.stride, ctx.stride, ctx.padding, ctx.padding, ctx.dilation,
            ctx.dilation, ctx.groups, ctx.deformable_groups, ctx.with_bias)
        return output

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
            grad_offset, grad_mask, grad_output, weight.shape[2], weight.shape[3], ctx.stride,
            ctx.stride, ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups,
            ctx.deformable_groups, ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None,
                None, None)

    @staticmethod
    def _infer_shape(ctx, input, weight):
        n = input.size(0)
        channels_out = weight.size(0)
        height, width = input.shape[2:4]
        kernel_h, kernel_w = weight.shape[2:4]
        height_out = (height + 2 * ctx.padding - (ctx.dilation * (kernel_h - 1) + 1)) // ctx.stride + 1
        width_out = (width + 2 * ctx.padding - (ctx.dilation * (kernel_w - 1) + 1)) // ctx.stride + 1
        return n, channels_out, height_out, width_out


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply
