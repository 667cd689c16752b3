"""oracle/dcn.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of the CPU oracle (oracle/dcn_oracle.c) for the five native entry points of
the reference extension ``dcn_deform_conv_cuda``
(lib/models/external/src/dcn_deform_conv_cuda.cpp:681-695).  Takes / returns CPU torch
tensors (float32 or float64).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdcn_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_threads(n):
    """OpenMP team size of the oracle's loops; returns the previous value."""
    return int(lib().dcn_oracle_set_threads(int(n)))


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64, got %s" % t.dtype)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def out_size(H, W, kH, kW, stride, padding, dilation):
    """functions/dcn_deform_conv.py:96-110 / dcn_deform_conv_cuda.cpp:187-190."""
    sH, sW = _pair(stride)
    pH, pW = _pair(padding)
    dH, dW = _pair(dilation)
    Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) // sH + 1
    Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) // sW + 1
    return Ho, Wo


def _geom(x, weight, stride, padding, dilation, groups, deformable_groups):
    N, C, H, W = x.shape
    Co, _, kH, kW = weight.shape
    sH, sW = _pair(stride)
    pH, pW = _pair(padding)
    dH, dW = _pair(dilation)
    return [N, C, H, W, Co, kH, kW, sH, sW, pH, pW, dH, dW, int(groups), int(deformable_groups)]


def deform_conv_forward(x, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                        deformable_groups=1, mask=None, bias=None):
    """deform_conv_forward_cuda (cpp:151-258); with mask/bias: modulated (cpp:486-564)."""
    x, offset, weight = x.contiguous(), offset.contiguous(), weight.contiguous()
    mask = mask.contiguous() if mask is not None else None
    bias = bias.contiguous() if bias is not None else None
    g = _geom(x, weight, stride, padding, dilation, groups, deformable_groups)
    Ho, Wo = out_size(g[2], g[3], g[5], g[6], stride, padding, dilation)
    out = torch.empty(g[0], g[4], Ho, Wo, dtype=x.dtype)
    fn = getattr(lib(), "dcn_oracle_forward_" + _suffix(x))
    rc = fn(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(out),
            *[ctypes.c_int(v) for v in g])
    if rc != 0:
        raise RuntimeError("oracle forward failed rc=%d" % rc)
    return out


def deform_conv_backward_input(x, offset, weight, grad_out, stride=1, padding=0, dilation=1,
                               groups=1, deformable_groups=1, mask=None):
    """deform_conv_backward_input_cuda (cpp:260-371) -> (grad_input, grad_offset[, grad_mask])."""
    x, offset, weight, grad_out = (t.contiguous() for t in (x, offset, weight, grad_out))
    mask = mask.contiguous() if mask is not None else None
    g = _geom(x, weight, stride, padding, dilation, groups, deformable_groups)
    gx = torch.zeros_like(x)
    goff = torch.zeros_like(offset)
    gmask = torch.zeros_like(mask) if mask is not None else None
    fn = getattr(lib(), "dcn_oracle_backward_input_" + _suffix(x))
    rc = fn(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(grad_out), _ptr(gx), _ptr(goff),
            _ptr(gmask), *[ctypes.c_int(v) for v in g])
    if rc != 0:
        raise RuntimeError("oracle backward_input failed rc=%d" % rc)
    return (gx, goff) if mask is None else (gx, goff, gmask)


def deform_conv_backward_params(x, offset, weight_shape, grad_out, stride=1, padding=0, dilation=1,
                                groups=1, deformable_groups=1, mask=None, with_bias=False,
                                scale=1.0):
    """deform_conv_backward_parameters_cuda (cpp:373-484) -> grad_weight[, grad_bias]."""
    x, offset, grad_out = (t.contiguous() for t in (x, offset, grad_out))
    mask = mask.contiguous() if mask is not None else None
    gw = torch.zeros(weight_shape, dtype=x.dtype)
    gb = torch.zeros(weight_shape[0], dtype=x.dtype) if with_bias else None
    g = _geom(x, gw, stride, padding, dilation, groups, deformable_groups)
    fn = getattr(lib(), "dcn_oracle_backward_params_" + _suffix(x))
    rc = fn(_ptr(x), _ptr(offset), _ptr(mask), _ptr(grad_out), _ptr(gw), _ptr(gb),
            *[ctypes.c_int(v) for v in g], ctypes.c_double(scale))
    if rc != 0:
        raise RuntimeError("oracle backward_params failed rc=%d" % rc)
    return (gw, gb) if with_bias else gw


def im2col(x, offset, kH, kW, stride=1, padding=0, dilation=1, deformable_groups=1, mask=None):
    """deformable_im2col (_kernel.cu:189-276) -> cols [C*kH*kW, N, Ho, Wo]."""
    x, offset = x.contiguous(), offset.contiguous()
    mask = mask.contiguous() if mask is not None else None
    N, C, H, W = x.shape
    sH, sW = _pair(stride)
    pH, pW = _pair(padding)
    dH, dW = _pair(dilation)
    Ho, Wo = out_size(H, W, kH, kW, stride, padding, dilation)
    cols = torch.empty(C * kH * kW, N, Ho, Wo, dtype=x.dtype)
    fn = getattr(lib(), "dcn_oracle_im2col_" + _suffix(x))
    rc = fn(_ptr(x), _ptr(offset), _ptr(mask), _ptr(cols),
            *[ctypes.c_int(v) for v in (N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW, deformable_groups)])
    if rc != 0:
        raise RuntimeError("oracle im2col failed rc=%d" % rc)
    return cols


class _OracleDeformConv(torch.autograd.Function):
    """Autograd wrapper so whole reference-shaped modules can run on the oracle (CPU)."""

    @staticmethod
    def forward(ctx, x, offset, weight, stride, padding, dilation, groups, deformable_groups):
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups)
        ctx.save_for_backward(x, offset, weight)
        return deform_conv_forward(x, offset, weight, *ctx.cfg)

    @staticmethod
    def backward(ctx, go):
        x, offset, weight = ctx.saved_tensors
        gx, goff = deform_conv_backward_input(x, offset, weight, go, *ctx.cfg)
        gw = deform_conv_backward_params(x, offset, tuple(weight.shape), go, *ctx.cfg)
        return gx, goff, gw, None, None, None, None, None


def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64):
    """Same call signature as functions/dcn_deform_conv.py:185 ``deform_conv``."""
    return _OracleDeformConv.apply(x, offset, weight, stride, padding, dilation, groups,
                                   deformable_groups)
