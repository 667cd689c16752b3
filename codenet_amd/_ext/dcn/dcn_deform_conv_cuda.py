"""Tensor-level shim with the names and argument lists of the reference's pybind11 module
``dcn_deform_conv_cuda`` (lib/models/external/src/dcn_deform_conv_cuda.cpp:681-695), over the
C ABI of libcodenet_dcn.so (include/codenet_dcn.h).

The module is importable where the reference expects it (``_ext.dcn.dcn_deform_conv_cuda``,
functions/dcn_deform_conv.py:8); only marshalling happens here: contiguity, shape checks that
need the tensors (cpp:61-149), pointer/stream extraction.  All arithmetic is in the HIP library.
"""
import os

import torch

from ... import _native as N_

__all__ = [
    "deform_conv_forward_cuda", "deform_conv_backward_input_cuda",
    "deform_conv_backward_parameters_cuda", "modulated_deform_conv_cuda_forward",
    "modulated_deform_conv_cuda_backward",
]


def _dtype_enum(t, *others):
    """fp16 (the reference dispatches half too: AT_DISPATCH_FLOATING_TYPES_AND_HALF, _kernel.cu:258,352,450) is native in
    the library since round 6 (CDN_F16: half in memory, fp32 arithmetic, one rounding per stored element); every tensor of
    a call has one dtype, as in the reference."""
    for o in others:
        if o is not None and o.dtype != t.dtype:
            raise RuntimeError("codenet_amd: all tensors of a call must share one dtype (got %s and %s)" % (t.dtype, o.dtype))
    if t.dtype == torch.float32:
        return N_.CDN_F32
    if t.dtype == torch.float64:
        return N_.CDN_F64
    if t.dtype == torch.float16:
        return N_.CDN_F16
    raise RuntimeError("codenet_amd: unsupported dtype %s (float16 / float32 / float64)" % t.dtype)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            # reference: functions/dcn_deform_conv.py:43-45
            raise NotImplementedError("codenet_amd runs on the GPU only (got a %s tensor)" % t.device)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_common(input, offset, weight, kH, kW, dH, dW, padH, padW, dilH, dilW, group, dg,
                  grad_output=None):
    """The tensor-dependent half of shape_check (cpp:61-149); the rest is in the library."""
    if weight.dim() != 4:
        raise RuntimeError("4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d"
                           % weight.dim())
    if weight.size(2) != kH or weight.size(3) != kW:
        raise RuntimeError("kernel size should be consistent with weight, but got kH: %d kW: %d "
                           "weight.size(2): %d, weight.size(3): %d"
                           % (kH, kW, weight.size(2), weight.size(3)))
    if input.dim() != 4:
        raise RuntimeError("4D input tensor expected but got: %d" % input.dim())
    n_in = weight.size(1) * group
    if input.size(1) != n_in:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d"
                           % (n_in, input.size(1)))
    H, W = input.size(2), input.size(3)
    Ho = (H + 2 * padH - (dilH * (kH - 1) + 1)) // dH + 1
    Wo = (W + 2 * padW - (dilW * (kW - 1) + 1)) // dW + 1
    if Ho < 1 or Wo < 1:
        raise RuntimeError("Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). "
                           "Output size is too small" % (n_in, H, W, weight.size(0), Ho, Wo))
    if offset.size(0) != input.size(0):
        raise RuntimeError("invalid batch size of offset")
    if offset.size(2) != Ho or offset.size(3) != Wo:
        raise RuntimeError("invalid spatial size of offset, expected height: %d width: %d, but got "
                           "height: %d width: %d" % (Ho, Wo, offset.size(2), offset.size(3)))
    if offset.size(1) != dg * 2 * kH * kW:
        raise RuntimeError("invalid number of channels of offset")
    if grad_output is not None:
        if grad_output.size(1) != weight.size(0):
            raise RuntimeError("invalid number of gradOutput planes, expected: %d, but got: %d"
                               % (weight.size(0), grad_output.size(1)))
        if grad_output.size(2) != Ho or grad_output.size(3) != Wo:
            raise RuntimeError("invalid size of gradOutput, expected height: %d width: %d , but got "
                               "height: %d width: %d"
                               % (Ho, Wo, grad_output.size(2), grad_output.size(3)))
    return Ho, Wo


def _scratch_from(columns, need, like):
    """`need` bytes of f32 scratch on `like`'s device: the caller's `columns` tensor when it can serve (resized, as the
    reference's C++ resizes it), a fresh allocation otherwise."""
    if (isinstance(columns, torch.Tensor) and columns.is_cuda and columns.device == like.device
            and columns.dtype == torch.float32 and columns.is_contiguous()):
        if columns.numel() * 4 < need:
            columns.resize_((need + 3) // 4)
        return columns
    return torch.empty((need + 3) // 4, dtype=torch.float32, device=like.device)


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW,
                             padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """cpp:151-258.  `ones` and `im2col_step` are accepted and ignored (vestigial).  `columns` -- the reference's scratch
    tensor, resized by its C++ (cpp:196-200) -- is this call's scratch too where the library has a use for one: for the
    CoDeNet call geometry (depthwise 3x3) it receives the per-pixel structure plane of the offsets (N x H x W floats,
    cdn_deform_conv_forward_scratch), so that the offsets' anchor * t structure is tested once per call instead of once
    per channel chunk."""
    _require_gpu(input, weight, offset, output)
    Ho, Wo = _check_common(input, offset, weight, kH, kW, dH, dW, padH, padW, dilationH,
                           dilationW, group, deformable_group)
    x, w, o = input.contiguous(), weight.contiguous(), offset.contiguous()
    Nb, C, H, W = x.shape
    Co = w.size(0)
    if tuple(output.shape) != (Nb, Co, Ho, Wo) or not output.is_contiguous():
        raise RuntimeError("output must be a contiguous [%d,%d,%d,%d] tensor" % (Nb, Co, Ho, Wo))
    lib = N_.lib()
    geom = (Nb, C, H, W, Co, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group)
    need = lib.cdn_deform_conv_forward_scratch_bytes(*geom) if x.dtype == torch.float32 else 0
    if need and os.environ.get("CDN_SEAM_NO_SCRATCH") == "1":      # A/B: every workgroup tests its pixels itself
        need = 0
    scratch = _scratch_from(columns, need, x) if need else None
    rc = lib.cdn_deform_conv_forward_scratch(
        _ptr(x), _ptr(w), _ptr(o), _ptr(output), _dtype_enum(x, w, o, output), *geom, _ptr(scratch), need if scratch is not None else 0,
        _stream(x))
    N_.check(rc, "deform_conv_forward_cuda")
    return 1


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight,
                                    columns, kW, kH, dW, dH, padW, padH, dilationW, dilationH,
                                    group, deformable_group, im2col_step):
    """cpp:260-371.  gradInput is accumulated into (zero-filled by the caller).  `columns` (the reference's scratch
    tensor) receives the offsets' structure plane for the CoDeNet call geometry, as in deform_conv_forward_cuda:
    cdn_deform_conv_backward_input_scratch runs the module backward's geometry when every pixel has it."""
    _require_gpu(input, offset, gradOutput, gradInput, gradOffset, weight)
    _check_common(input, offset, weight, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
                  deformable_group, gradOutput)
    x, o, go, w = (t.contiguous() for t in (input, offset, gradOutput, weight))
    if not (gradInput.is_contiguous() and gradOffset.is_contiguous()):
        raise RuntimeError("gradInput / gradOffset must be contiguous")
    Nb, C, H, W = x.shape
    lib = N_.lib()
    geom = (Nb, C, H, W, w.size(0), kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group)
    need = lib.cdn_deform_conv_backward_input_scratch_bytes(*geom) if x.dtype == torch.float32 else 0
    if need and os.environ.get("CDN_SEAM_NO_SCRATCH") == "1":      # A/B: the generic nine-tap backward
        need = 0
    elif need and os.environ.get("CDN_SEAM_MIN_SCRATCH") == "1":   # A/B: structured kernels with float atomics
        need = lib.cdn_deform_conv_backward_input_scratch_min_bytes(*geom)
    scratch = _scratch_from(columns, need, x) if need else None
    rc = lib.cdn_deform_conv_backward_input_scratch(
        _ptr(x), _ptr(o), _ptr(go), _ptr(gradInput), _ptr(gradOffset), _ptr(w),
        _dtype_enum(x, o, go, gradInput, gradOffset, w), *geom, _ptr(scratch), need if scratch is not None else 0,
        _stream(x))
    N_.check(rc, "deform_conv_backward_input_cuda")
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW,
                                         kH, dW, dH, padW, padH, dilationW, dilationH, group,
                                         deformable_group, scale, im2col_step):
    """cpp:373-484.  gradWeight += scale * dL/dW."""
    _require_gpu(input, offset, gradOutput, gradWeight)
    _check_common(input, offset, gradWeight, kH, kW, dH, dW, padH, padW, dilationH, dilationW,
                  group, deformable_group, gradOutput)
    x, o, go = (t.contiguous() for t in (input, offset, gradOutput))
    if not gradWeight.is_contiguous():
        raise RuntimeError("gradWeight must be contiguous")
    Nb, C, H, W = x.shape
    rc = N_.lib().cdn_deform_conv_backward_parameters(
        _ptr(x), _ptr(o), _ptr(go), _ptr(gradWeight), _dtype_enum(x, o, go, gradWeight), Nb, C, H, W,
        gradWeight.size(0), kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
        deformable_group, float(scale), _stream(x))
    N_.check(rc, "deform_conv_backward_parameters_cuda")
    return 1


def _check_modulated(input, weight, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                     pad_w, dilation_h, dilation_w, group, deformable_group):
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")      # cpp:493
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")     # cpp:494
    if weight.size(2) != kernel_h or weight.size(3) != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)."
                           % (kernel_h, kernel_w, weight.size(2), weight.size(3)))
    if input.size(1) != weight.size(1) * group:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)."
                           % (input.size(1), weight.size(1) * group))
    H, W = input.size(2), input.size(3)
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    K = kernel_h * kernel_w
    if tuple(offset.shape) != (input.size(0), deformable_group * 2 * K, Ho, Wo):
        raise RuntimeError("invalid offset shape %s" % (tuple(offset.shape),))
    if tuple(mask.shape) != (input.size(0), deformable_group * K, Ho, Wo):
        raise RuntimeError("invalid mask shape %s" % (tuple(mask.shape),))
    return Ho, Wo


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns,
                                       kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                       dilation_h, dilation_w, group, deformable_group, with_bias):
    """cpp:486-564."""
    _require_gpu(input, weight, offset, mask, output)
    Ho, Wo = _check_modulated(input, weight, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                              pad_h, pad_w, dilation_h, dilation_w, group, deformable_group)
    o, m = offset.contiguous(), mask.contiguous()
    Nb, C, H, W = input.shape
    Co = weight.size(0)
    if tuple(output.shape) != (Nb, Co, Ho, Wo) or not output.is_contiguous():
        raise RuntimeError("output must be a contiguous [%d,%d,%d,%d] tensor" % (Nb, Co, Ho, Wo))
    b = bias.contiguous() if with_bias else None
    rc = N_.lib().cdn_modulated_deform_conv_forward(
        _ptr(input), _ptr(weight), _ptr(b), _ptr(o), _ptr(m), _ptr(output), _dtype_enum(input),
        Nb, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
        dilation_w, group, deformable_group, int(bool(with_bias)), _stream(input))
    N_.check(rc, "modulated_deform_conv_cuda_forward")


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns,
                                        grad_input, grad_weight, grad_bias, grad_offset, grad_mask,
                                        grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                        pad_w, dilation_h, dilation_w, group, deformable_group,
                                        with_bias):
    """cpp:566-679."""
    _require_gpu(input, weight, offset, mask, grad_output)
    _check_modulated(input, weight, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                     pad_w, dilation_h, dilation_w, group, deformable_group)
    o, m, go = offset.contiguous(), mask.contiguous(), grad_output.contiguous()
    for t in (grad_input, grad_weight, grad_offset, grad_mask):
        if not t.is_contiguous():
            raise RuntimeError("gradient buffers must be contiguous")
    Nb, C, H, W = input.shape
    rc = N_.lib().cdn_modulated_deform_conv_backward(
        _ptr(input), _ptr(weight), _ptr(bias) if with_bias else None, _ptr(o), _ptr(m),
        _ptr(grad_input), _ptr(grad_weight), _ptr(grad_bias) if with_bias else None,
        _ptr(grad_offset), _ptr(grad_mask), _ptr(go), _dtype_enum(input), Nb, C, H, W,
        weight.size(0), kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
        dilation_w, group, deformable_group, int(bool(with_bias)), _stream(input))
    N_.check(rc, "modulated_deform_conv_cuda_backward")
