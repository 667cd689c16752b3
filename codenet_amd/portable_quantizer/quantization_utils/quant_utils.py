"""Quantisation arithmetic with the names of the reference's
portable_quantizer/quantization_utils/quant_utils.py.  Device-agnostic torch ops in the
reference's fp32 expression order (they are used for the small weight tensors and as the autograd
(straight-through) wrappers); whole activation tensors on the GPU go through the HIP QuantAct
kernel instead (codenet_amd/csrc/codenet_quant.hip)."""
import torch
from torch.autograd import Function

__all__ = ["clamp", "get_percentile_min_max", "linear_quantize", "linear_dequantize",
           "linear_quantize_clamp", "asymmetric_linear_quantization_params",
           "symmetric_linear_quantization_params", "get_tensor_min_max",
           "AsymmetricQuantFunction", "SymmetricQuantFunction"]


def clamp(input, min, max, inplace=False):
    return input.clamp_(min, max) if inplace else torch.clamp(input, min, max)


def get_percentile_min_max(input, lower_percentile, upper_percentile, output_tensor=False):
    """k-th value bounds of a flat tensor (reference :18-30; k = round(n * pct / 100))."""
    n = input.shape[0]
    k_lo, k_hi = round(n * lower_percentile * 0.01), round(n * upper_percentile * 0.01)
    if (input.is_cuda and input.dtype == torch.float32 and 1 <= k_lo <= n and 1 <= k_hi <= n and n < 2 ** 32
            and not (torch.is_grad_enabled() and input.requires_grad)):
        # GPU: both order statistics from three histogram passes (exact radix select) instead of two sorts
        from ... import ops
        lo, hi = ops.kth_values(input, k_lo, k_hi)
        lo, hi = lo.reshape(()), hi.reshape(())
    else:
        lo = torch.kthvalue(input, k=k_lo).values
        hi = torch.kthvalue(input, k=k_hi).values
    return (lo, hi) if output_tensor else (lo.item(), hi.item())


def _bcast(v):
    return v.view(-1, 1, 1, 1)


def linear_quantize(input, scale, zero_point, inplace=False):
    """round(scale * x - zero_point), round-half-even (reference :33-41)."""
    scale, zero_point = _bcast(scale), _bcast(zero_point)
    if inplace:
        return input.mul_(scale).sub_(zero_point).round_()
    return torch.round(scale * input - zero_point)


def linear_dequantize(input, scale, zero_point, inplace=False):
    """(q + zero_point) / scale -- a true division (reference :44-52)."""
    scale, zero_point = _bcast(scale), _bcast(zero_point)
    if inplace:
        return input.add_(zero_point).div_(scale)
    return (input + zero_point) / scale


def linear_quantize_clamp(input, scale, zero_point, clamp_min, clamp_max, inplace=False):
    return clamp(linear_quantize(input, scale, zero_point, inplace), clamp_min, clamp_max, inplace)


def asymmetric_linear_quantization_params(num_bits, saturation_min, saturation_max,
                                          integral_zero_point=True, signed=True):
    """scale = (2^k-1)/clamp(max-min,1e-10); zp = round(scale*min) (+2^(k-1) if signed) (:60-75)."""
    n = 2 ** num_bits - 1
    scale = n / torch.clamp(saturation_max - saturation_min, min=1e-10)
    zero_point = scale * saturation_min
    if integral_zero_point:
        zero_point = zero_point.round() if isinstance(zero_point, torch.Tensor) \
            else float(round(zero_point))
    if signed:
        zero_point = zero_point + 2 ** (num_bits - 1)
    return scale, zero_point


def symmetric_linear_quantization_params(num_bits, saturation_magnitude, signed=False):
    """scale = (2^(k-1)-1)/clamp(mag,1e-10); zp = 0 (:78-84)."""
    if signed:
        raise NotImplementedError
    n = 2 ** (num_bits - 1) - 1
    scale = n / torch.clamp(saturation_magnitude, min=1e-10)
    return scale, torch.zeros_like(scale)


def get_tensor_min_max(t, per_dim=None):
    if per_dim is None:
        return t.min(), t.max()
    if per_dim > t.dim():
        raise ValueError("Got per_dim={0}, but tensor only has {1} dimensions".format(per_dim, t.dim()))
    tv = t.view(*[t.shape[i] for i in range(per_dim + 1)], -1)
    return tv.min(dim=-1)[0], tv.max(dim=-1)[0]


class AsymmetricQuantFunction(Function):
    """Asymmetric fake-quantisation with straight-through gradient (reference :172-204).
    The per-channel branch clamps codes to [0, 2^k-1]; the per-tensor branch does not clamp."""

    @staticmethod
    def forward(ctx, x, k, x_min=None, x_max=None, per_channel=False, percentile_mode=False,
                show=False):
        if x_min is None or x_max is None:
            x_min, x_max = x.min(), x.max()
        scale, zero_point = asymmetric_linear_quantization_params(k, x_min, x_max)
        q = linear_quantize(x, scale, zero_point)
        if per_channel:
            q = torch.clamp(q, 0, 2 ** k - 1)
        out = linear_dequantize(q, scale, zero_point)
        return (out, scale, zero_point) if show else out

    @staticmethod
    def backward(ctx, grad_output, *unused):
        return grad_output.clone(), None, None, None, None, None, None


class SymmetricQuantFunction(Function):
    """Symmetric fake-quantisation, codes clamped to [-2^(k-1), 2^(k-1)-1], straight-through
    gradient (reference :207-229)."""

    @staticmethod
    def forward(ctx, x, k, x_min=None, x_max=None, per_channel=False, percentile_mode=False,
                show=False):
        if per_channel:
            magnitude = torch.max(torch.stack([x_min.abs(), x_max.abs()], dim=1), dim=1).values
        else:
            magnitude = max(x_min.abs(), x_max.abs())
        scale, zero_point = symmetric_linear_quantization_params(k, magnitude)
        n = 2 ** (k - 1)
        q = torch.clamp(linear_quantize(x, scale, zero_point), -n, n - 1)
        out = linear_dequantize(q, scale, zero_point)
        return (out, scale, zero_point) if show else out

    @staticmethod
    def backward(ctx, grad_output, *unused):
        return grad_output.clone(), None, None, None, None, None, None
