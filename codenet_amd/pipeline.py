"""The hot path as the reference composes it: ``PoseShuffleNetV2.deconv_layers``
(lib/models/networks/shufflenetv2_dcn.py:286-312) -- three
[DeformConvWithOffsetScaleBoundPositive, BatchNorm2d, ReLU, Upsample x2] groups -- in fp32, or after
``quantize_shufflenetv2_dcn`` (quantize_model.py:70-82) three
[QuantDeformConvWithOffsetScaleBoundPositive, Sequential(ReLU, QuantAct), Upsample x2] groups.

Used by bench.py, the smoke test and the multi-process tests.  Data-parallel inference shards
independent image batches over one process per GPU; the only collective is a start-up broadcast
of parameters and buffers (weights, BN statistics, QuantAct ranges) from rank 0 over RCCL
(SURVEY.md section 8e).
"""
import torch
import torch.nn as nn

from .modules.dcn_deform_conv import DeformConvWithOffsetScaleBoundPositive
from .portable_quantizer.quantization_utils.quantize_model import quantize_deform_stages

BN_MOMENTUM = 0.1


def stage_shapes(input_res=512, w2=False):
    """[(C_in, C_out, H, W)] of the three deform stages."""
    c0 = 2153 if w2 else 1024          # shufflenetv2_dcn.py:199-202,293-296
    r = input_res // 32
    return [(c0, 256, r, r), (256, 128, 2 * r, 2 * r), (128, 64, 4 * r, 4 * r)]


class DeconvLayers(nn.Module):
    """Container with the reference's attribute name so the quantiser's surgery applies."""

    def __init__(self, w2=False):
        super().__init__()
        planes_in = [2153 if w2 else 1024, 256, 128]
        layers = []
        for cin, cout in zip(planes_in, [256, 128, 64]):
            layers += [
                DeformConvWithOffsetScaleBoundPositive(cin, cout, 3, 1, 1, groups=cout, bias=False,
                                                        hidden_state=128, BN_MOMENTUM=BN_MOMENTUM),
                nn.BatchNorm2d(cout, momentum=BN_MOMENTUM),
                nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="nearest"),
            ]
        self.deconv_layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.deconv_layers(x)


def build_hot_path(w2=False, quantized=True, seed=317, scale_std=3.0):
    """Seeded synthetic weights: reference initialisers, except a non-degenerate
    conv_scale (weight ~ N(0, scale_std/sqrt(C)), bias 1 => s ~ N(1, scale_std) on unit-power inputs,
    clipped to [-7, 8] with ~1 % of pixels at each clamp) and non-trivial BN running statistics."""
    g = torch.Generator().manual_seed(seed)
    net = DeconvLayers(w2=w2)
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
        quantize_deform_stages(net, 4, 8, "symmetric", "asymmetric", True, False, False)
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
    from .portable_quantizer.quant_modules import QuantAct
    for m in net.modules():
        if isinstance(m, QuantAct):
            m.running_stat = flag


def broadcast_parameters(net, src=0):
    """Start-up broadcast of every parameter and buffer from `src` as ONE flat tensor
    (RCCL over xGMI when the process group backend is nccl; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in net.parameters()] + [b.data for b in net.buffers()
                                                    if b.dtype.is_floating_point]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return flat.numel() * 4


def shard_range(total, rank, world):
    """Contiguous images [lo, hi) of a `total`-image batch owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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
