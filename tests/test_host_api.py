"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the Python operator API has the reference's surface (names, ctor signatures, parameter
/ buffer names), the device-agnostic quantiser code reproduces the reference's golden vectors, and
the product refuses CPU tensors exactly like the reference.  No GPU compute here."""
import inspect
import os
import re
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name)).items()}


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "codenet_dcn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cdn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from codenet_amd import _native
    if not os.path.exists(_native.SO_PATH):
        _native.build()
    lib = _native.lib()
    syms = _header_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), "libcodenet_dcn.so does not export %s" % s
        assert s in _native._SIGNATURES, "no ctypes signature for %s" % s
    assert set(_native._SIGNATURES) == set(syms)
    assert lib.cdn_abi_version() == 1


def test_argument_validation_without_gpu():
    """Shape / argument errors are detected before any HIP call, so they can be checked here."""
    from codenet_amd import _native
    lib = _native.lib()
    one = 4096  # any non-null fake pointer: validation must fail before it is dereferenced
    rc = lib.cdn_deform_conv_forward(one, one, one, one, 0, 1, 4, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1,
                                     3, 1, None)       # 4 channels not divisible by group 3
    assert rc == -2 and b"group" in lib.cdn_last_error()
    rc = lib.cdn_deform_conv_forward(None, one, one, one, 0, 1, 4, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1,
                                     1, 1, None)
    assert rc == -1
    rc = lib.cdn_deform_conv_forward(one, one, one, one, 7, 1, 4, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1,
                                     1, 1, None)
    assert rc == -3
    rc = lib.cdn_deform_conv_forward(one, one, one, one, 0, 1, 4, 2, 2, 4, 3, 3, 1, 1, 0, 0, 1, 1,
                                     1, 1, None)       # 2x2 input, 3x3 kernel, no padding
    assert rc == -2 and b"too small" in lib.cdn_last_error()
    rc = lib.cdn_codenet_pointwise_forward(one, one, None, one, None, one, 1, 4, 4, 16, 0, None)
    assert rc == -1                                     # ep_scale without ep_shift


def test_argument_validation_of_the_layer_entry_points_without_gpu():
    """The entry points around the hot path (heads, backbone layers, decode) validate before any HIP call."""
    from codenet_amd import _native
    lib = _native.lib()
    one = 4096
    assert lib.cdn_codenet_aux_workspace_bytes() >= 16384 * 8
    # pointwise: output QuantAct given without a workspace
    rc = lib.cdn_codenet_pointwise_nhwc_forward(one, None, 64, 16, 16, 0, 0, one, None, None, None, None,
                                                None, None, 1, one, one, one, 8, 0.99, 1, None, 0, one, None)
    assert rc != 0 and b"workspace" in lib.cdn_last_error()
    # pointwise: row stride smaller than the channel count
    rc = lib.cdn_codenet_pointwise_nhwc_forward(one, None, 64, 16, 16, 8, 0, one, None, None, None, None,
                                                None, None, 1, None, None, None, 8, 0.99, 0, None, 0, one, None)
    assert rc != 0 and b"stride" in lib.cdn_last_error()
    # depthwise: up-sampling together with stride 2; row stride not a multiple of 4
    rc = lib.cdn_codenet_dw3x3_nhwc_forward(one * 4, None, 1, 8, 8, 8, 1, 2, 0, 0, one, None, None, None, 1,
                                            None, None, None, 8, 0.99, 0, None, 0, one, None)
    assert rc != 0
    rc = lib.cdn_codenet_dw3x3_nhwc_forward(one * 4, None, 1, 6, 8, 8, 0, 1, 6, 6, one, None, None, None, 1,
                                            None, None, None, 8, 0.99, 0, None, 0, one, None)
    assert rc != 0 and b"ld_in" in lib.cdn_last_error()
    # interleave: destination row narrower than 2 h
    rc = lib.cdn_codenet_interleave_forward(one, 8, None, one, 8, None, 10, 8, one, 12, None)
    assert rc != 0
    # stem: only the 24-channel instantiation exists
    rc = lib.cdn_codenet_stem_forward(one, 1, 32, 32, 16, 4, one, None, 1, None, None, None, 8, 0.99, 0,
                                      None, 0, one, None)
    assert rc != 0 and b"24" in lib.cdn_last_error()
    # decode: K larger than the map, missing workspace
    assert lib.cdn_ctdet_decode_workspace_bytes(2, 20, 128, 128) >= 2 * 20 * 128 * 128 * 4
    rc = lib.cdn_ctdet_decode(one, one, None, 1, 2, 4, 4, 0, 100, 0, None, one, one * 64, 1 << 20, None)
    assert rc != 0
    rc = lib.cdn_ctdet_decode(one, one, None, 1, 2, 8, 8, 0, 10, 0, None, one, None, 0, None)
    assert rc != 0
    rc = lib.cdn_codenet_maxpool3x3s2_nhwc_forward(one * 4, None, 1, 6, 8, 8, one * 4, None)
    assert rc != 0      # C % 4


def test_hot_kernels_stay_inside_their_register_budget():
    """hipcc -S metadata (no GPU): no spills / scratch in the hot kernels and the occupancy-defining VGPR
    budgets hold -- the stage-0 gather sits at 127 of 128 VGPRs and a spill costs 20-150 us per launch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_resources
    res, problems = check_resources.check()
    assert not problems, "\n".join(problems)


def test_cpu_tensors_raise_not_implemented():
    from codenet_amd.functions.dcn_deform_conv import deform_conv, modulated_deform_conv
    from codenet_amd import ops
    x = torch.randn(1, 4, 6, 6)
    off = torch.zeros(1, 18, 6, 6)
    w = torch.randn(4, 1, 3, 3)
    with pytest.raises(NotImplementedError):      # functions/dcn_deform_conv.py:43-45
        deform_conv(x, off, w, 1, 1, 1, 4, 1)
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(x, off, torch.ones(1, 9, 6, 6), w, None, 1, 1, 1, 4, 1)
    with pytest.raises(NotImplementedError):
        ops.codenet_dw(x, torch.ones(1, 1, 6, 6), w)
    with pytest.raises(ValueError):               # :24-27
        deform_conv(x[0], off, w, 1, 1, 1, 4, 1)


def test_module_surface_and_checkpoint_keys():
    from codenet_amd.modules import dcn_deform_conv as M
    m = M.DeformConvWithOffsetScaleBoundPositive(32, 16, 3, 1, 1, groups=16, bias=False,
                                                  hidden_state=128, BN_MOMENTUM=0.1)
    assert sorted(m.state_dict().keys()) == sorted(
        ["conv_scale.weight", "conv_scale.bias", "conv.weight", "conv_channel.weight"])
    assert tuple(m.conv_scale.weight.shape) == (1, 32, 1, 1)
    assert tuple(m.conv.weight.shape) == (32, 1, 3, 3) and m.conv.groups == 32
    assert tuple(m.conv_channel.weight.shape) == (16, 32, 1, 1)
    assert m.conv_scale.weight.abs().max().item() == 0 and m.conv_scale.bias.item() == 1.0
    assert (m.conv_bound.min_val, m.conv_bound.max_val) == (-7, 8)
    assert tuple(m.anchor_offset.shape) == (1, 18, 1, 1)
    assert m.anchor_offset.view(-1).tolist() == [-1, -1, -1, 0, -1, 1, 0, -1, 0, 0, 0, 1, 1, -1, 1, 0, 1, 1]
    # C == Co: no conv_channel (reference :327-330)
    m2 = M.DeformConvWithOffsetScaleBoundPositive(8, 8)
    assert "conv_channel.weight" not in m2.state_dict()
    sig = inspect.signature(M.DeformConv.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "out_channels", "kernel_size", "stride",
                                        "padding", "dilation", "groups", "deformable_groups", "bias"]
    sig = inspect.signature(M.ModulatedDeformConv.__init__)
    assert sig.parameters["padding"].default == 0 and sig.parameters["kernel_size"].default is inspect._empty
    for name in ["DeformConvPack", "DeformConvPack1x1", "DeformConvPackDW", "ModulatedDeformConvPack",
                 "DeformConvWithOffsetBound", "DeformConvWithOffsetRound", "DeformConvWithOffsetScale",
                 "DeformConvWithOffsetScaleBound", "ModulatedDeformConvWithOffsetScaleBoundPositive",
                 "ModulatedDeformConvWithOffset1x1ScaleBoundPositive"]:
        assert hasattr(M, name)
    p = M.ModulatedDeformConvPack(4, 6, 3, stride=1, padding=1, bias=True)
    assert sorted(p.state_dict()) == sorted(["weight", "bias", "conv_offset_mask.weight",
                                             "conv_offset_mask.bias"])


def _toy_stage_stack():
    from codenet_amd.modules import dcn_deform_conv as M

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            layers = []
            for cin, cout in ((32, 16), (16, 8), (8, 4)):
                layers += [M.DeformConvWithOffsetScaleBoundPositive(cin, cout, 3, 1, 1, groups=cout),
                           nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                           nn.Upsample(scale_factor=2, mode="nearest")]
            self.deconv_layers = nn.Sequential(*layers)
    return Net()


def test_quantized_stage_checkpoint_keys():
    from codenet_amd.portable_quantizer import quantize_deform_stages
    net = _toy_stage_stack()
    quantize_deform_stages(net, 4, 8, "symmetric", "asymmetric", True, False, False)
    keys = set(net.state_dict().keys())
    for i, j in ((0, 1), (3, 4), (6, 7)):
        pre = "deconv_layers.%d." % i
        for k in ["quant_conv_scale.weight", "quant_conv_scale.bias", "quant_act.1.x_min",
                  "quant_act.1.x_max", "quant_deform_conv.weight", "quant_identity_deform.x_min",
                  "quant_identity_deform.x_max", "quant_conv_channel_bn.conv.weight",
                  "quant_conv_channel_bn.bn.weight", "quant_conv_channel_bn.bn.bias",
                  "quant_conv_channel_bn.bn.running_mean", "quant_conv_channel_bn.bn.running_var",
                  "quant_conv_channel_bn.bn.num_batches_tracked"]:
            assert pre + k in keys, pre + k
        assert "deconv_layers.%d.1.x_min" % j in keys and "deconv_layers.%d.1.x_max" % j in keys
    assert len(net.deconv_layers) == 9


def test_quantizer_torch_path_matches_reference_golden():
    """The device-agnostic quantiser code (used for weights everywhere and for autograd) against
    vectors from the reference's own modules."""
    from codenet_amd.portable_quantizer import quant_modules as QM
    from codenet_amd.portable_quantizer.quantization_utils import quant_utils as QU
    z = load("quant_ref.npz")
    qa = QM.QuantAct(8, quant_mode="asymmetric")
    for it in range(4):
        y = qa(z["act_x%d" % it])
        assert torch.equal(qa.x_min, z["act_min%d" % it]) and torch.equal(qa.x_max, z["act_max%d" % it])
        assert torch.equal(y, z["act_y%d" % it])
    qp = QM.QuantAct(8, quant_mode="asymmetric", percentile=True)
    for it in range(2):
        y = qp(z["pact_x%d" % it])
        assert torch.equal(qp.x_min, z["pact_min%d" % it]) and torch.equal(y, z["pact_y%d" % it])
    for tag, pct in (("n", False), ("p", True)):
        conv = nn.Conv2d(40, 6, 1, bias=True)
        conv.weight.data, conv.bias.data = z["qconv_%s_w" % tag], z["qconv_%s_b" % tag]
        qc = QM.Quant_Conv2d(4, quant_mode="symmetric", per_channel=True, weight_percentile=pct)
        qc.set_param(conv)
        with torch.no_grad():
            assert (qc(z["qconv_%s_x" % tag]) - z["qconv_%s_y" % tag]).abs().max().item() < 1e-5
            assert qc.quantized_weight() is qc.quantized_weight()          # inference cache
        conv2 = nn.Conv2d(1200, 3, 1, bias=False)
        conv2.weight.data = z["qbn_%s_w" % tag]
        bn = nn.BatchNorm2d(3)
        bn.weight.data, bn.bias.data = z["qbn_%s_bn_weight" % tag], z["qbn_%s_bn_bias" % tag]
        bn.running_mean, bn.running_var = z["qbn_%s_bn_running_mean" % tag], z["qbn_%s_bn_running_var" % tag]
        qb = QM.QuantBnConv2d(4, quant_mode="symmetric", per_channel=True, weight_percentile=pct)
        qb.set_param(conv2, bn)
        with torch.no_grad():
            assert (qb(z["qbn_%s_x" % tag]) - z["qbn_%s_y" % tag]).abs().max().item() < 1e-4
        wd = z["qdw_%s_w" % tag]
        lo, hi = QM._channel_range(wd.view(5, -1), pct)
        assert torch.equal(QU.SymmetricQuantFunction.apply(wd, 4, lo, hi, True, pct), z["qdw_%s_wq" % tag])


def test_weight_cache_invalidated_by_in_place_update():
    from codenet_amd.portable_quantizer import quant_modules as QM
    conv = nn.Conv2d(12, 3, 1, bias=False)
    qc = QM.Quant_Conv2d(4, quant_mode="symmetric", per_channel=True)
    qc.set_param(conv)
    with torch.no_grad():
        a = qc.quantized_weight()
        a0, ptr = a.clone(), a.data_ptr()
        qc.weight.mul_(2.0)
        b = qc.quantized_weight()
    # refreshed IN PLACE: the pointer a captured HIP graph holds stays valid and sees the new values
    assert b.data_ptr() == ptr and torch.allclose(b, 2 * a0)
    # writes through .data bypass the version counter: invalidate() is the explicit way out
    with torch.no_grad():
        qc.weight.data.mul_(0.5)
        assert torch.allclose(qc.quantized_weight(), 2 * a0)      # stale by construction
        qc.invalidate()
        assert torch.allclose(qc.quantized_weight(), a0)


def test_broadcast_style_copy_bumps_versions_and_bn_affine_cache_follows():
    """pipeline.broadcast_parameters copies into the parameters themselves (version bump), and the BN
    affine caches of the fused schedules are keyed on versions -- loading a checkpoint after the first
    fused call must not keep the old statistics."""
    from codenet_amd import pipeline
    bn = nn.BatchNorm2d(4).eval()
    cache = {}
    es0, eh0 = [t.clone() for t in pipeline.bn_affine(cache, bn)]
    ptr = pipeline.bn_affine(cache, bn)[0].data_ptr()
    with torch.no_grad():
        bn.running_var.copy_(torch.full((4,), 4.0))
        bn.running_mean.copy_(torch.full((4,), 1.0))
    es1, eh1 = pipeline.bn_affine(cache, bn)
    assert es1.data_ptr() == ptr
    assert torch.allclose(es1, torch.full((4,), 1.0 / (4.0 + bn.eps) ** 0.5)) and not torch.allclose(es1, es0)
    assert torch.allclose(eh1, -es1)


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_drop_in_under_reference_network():
    """The reference's own PoseShuffleNetV2 (imported from /root/reference, not copied) builds on
    OUR operator module when it is installed at the reference's import path, and our quantiser
    rewrites it to the same module tree / checkpoint keys as the reference's quantiser."""
    code = r'''
import sys, types, os
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
for name in ["pytorchcv", "pytorchcv.model_provider", "pytorchcv.models", "pytorchcv.models.shufflenetv2",
             "pytorchcv.models.common", "thop", "_ext", "_ext.dcn", "_ext.dcn.dcn_deform_conv_cuda"]:
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["pytorchcv.model_provider"].get_model = lambda *a, **k: None
sys.modules["pytorchcv.models.shufflenetv2"].ShuffleUnit = type("ShuffleUnit", (), {})
sys.modules["pytorchcv.models.common"].ChannelShuffle = type("ChannelShuffle", (), {})
sys.modules["thop"].profile = lambda *a, **k: (0, 0)
sys.path.insert(0, os.path.join(REF, "lib")); sys.path.insert(0, REF)
import torch
mode = sys.argv[3]
if mode == "ours":
    import codenet_amd.modules.dcn_deform_conv as ours
    import models, models.external                      # reference packages (namespace)
    pkg = types.ModuleType("models.external.modules"); pkg.__path__ = []
    pkg.dcn_deform_conv = ours
    sys.modules["models.external.modules"] = pkg
    sys.modules["models.external.modules.dcn_deform_conv"] = ours
    from codenet_amd.portable_quantizer import quantize_shufflenetv2_dcn
else:
    from portable_quantizer import quantize_shufflenetv2_dcn
from models.networks.shufflenetv2_dcn import PoseShuffleNetV2
torch.manual_seed(0)
net = PoseShuffleNetV2({"hm": 20, "wh": 2, "reg": 2}, 64)
k_fp = sorted((k, tuple(v.shape)) for k, v in net.state_dict().items())
cls = type(net.deconv_layers[0]).__module__
quantize_shufflenetv2_dcn(net, 4, None, 8, "symmetric", "asymmetric", True, False, False, False)
k_q = sorted((k, tuple(v.shape)) for k, v in net.state_dict().items())
import json
print(json.dumps({"cls": cls, "fp": k_fp, "q": k_q}))
'''
    import json
    import subprocess
    outs = {}
    for mode in ("ours", "ref"):
        r = subprocess.run([sys.executable, "-c", code, ROOT, REF, mode], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["ours"]["cls"] == "codenet_amd.modules.dcn_deform_conv"
    assert outs["ours"]["fp"] == outs["ref"]["fp"]
    assert outs["ours"]["q"] == outs["ref"]["q"]
    assert len(outs["ours"]["q"]) > 300


def test_argument_validation_of_round2_entry_points_without_gpu():
    """Frozen-range schedule, training kernels and capability queries validate before any HIP call."""
    import ctypes
    from codenet_amd import _native
    lib = _native.lib()
    one = 4096
    # capability queries (pure host arithmetic)
    assert lib.cdn_codenet_stage_supported(64, 1024, 16, 16, 0, 0) == 1          # cfg3 stage 0
    assert lib.cdn_codenet_stage_supported(64, 128, 64, 64, 1, 1) == 1           # cfg3 stage 2 (stored 32 x 32)
    assert lib.cdn_codenet_stage_supported(1, 128, 160, 160, 1, 1) == 0          # stored 80 x 80: not LDS resident
    assert lib.cdn_codenet_stage_supported(1, 2153, 16, 16, 1, 0) == 0           # channels-last needs C % 4 == 0
    assert lib.cdn_codenet_stage_supported(1, 128, 15, 16, 1, 1) == 0            # x_up needs even H
    assert lib.cdn_codenet_stage_supported(1, 128, 16, 16, 0, 1) == 0            # an up-sampled input must be channels-last
    assert lib.cdn_codenet_dw_backward_supported(64, 64) == 1 and lib.cdn_codenet_dw_backward_supported(136, 136) == 1
    assert lib.cdn_codenet_dw_backward_supported(400, 400) == 0
    # frozen stage: byte codes need C % 4 == 0; x_state goes with channels-last inputs only; workspace size
    args = lambda **kw: [kw.get("x", one * 4), kw.get("kind", 0), kw.get("up", 0), kw.get("xs", None), 2,   # noqa: E731
                         kw.get("C", 64), 32, 8, 8, one, one, -7.0, 8.0, one, one * 4, one, one, None, 1, one, one, one,
                         kw.get("ws", one * 64), kw.get("wsb", 1 << 30), one * 4, one, None]
    assert lib.cdn_codenet_stage_frozen_forward(*args(C=2153)) == _native.CDN_ERR_UNSUPPORTED
    assert b"C % 4" in lib.cdn_last_error()
    assert lib.cdn_codenet_stage_frozen_forward(*args(kind=2)) == -1 and b"x_state" in lib.cdn_last_error()
    assert lib.cdn_codenet_stage_frozen_forward(*args(kind=0, xs=one)) == -1
    assert lib.cdn_codenet_stage_frozen_forward(*args(kind=0, up=1)) == _native.CDN_ERR_UNSUPPORTED
    assert lib.cdn_codenet_stage_frozen_forward(*args(wsb=16)) == -6
    assert lib.cdn_codenet_stage_frozen_workspace_bytes(64, 1024, 16, 16, 0) == 64 * 256 * 4 + 64 * 256 * 1024
    # frozen parameters: at most 64 QuantActs per call (round 6: the whole serving network's in one launch)
    arr = (ctypes.c_void_p * 65)(*([one] * 65))
    assert lib.cdn_quantact_frozen_params(65, arr, arr, arr, 8, None) == -1
    assert lib.cdn_quantact_frozen_params(0, None, None, None, 8, None) == 0
    # byte-code pointwise: exactly one output form; the byte form needs the output quantiser
    assert lib.cdn_codenet_pointwise_q8_forward(one * 4, one, 64, 64, 16, one * 4, one, one, None, 1, one, one * 4,
                                                one * 4, one, None) == -1
    assert lib.cdn_codenet_pointwise_q8_forward(one * 4, one, 64, 64, 16, one * 4, one, one, None, 1, None, one * 4,
                                                None, one, None) == -1
    assert lib.cdn_codenet_pointwise_q8_forward(one * 4, one, 64, 62, 16, one * 4, one, one, None, 1, one, one * 4,
                                                None, one, None) == -1          # C % 4
    # weight gradient: workspace too small; scale backward: null pointers
    need = lib.cdn_codenet_pointwise_wgrad_workspace_bytes(32, 128, 64, 4096)
    assert need >= 64 * 128 * 4
    assert lib.cdn_codenet_pointwise_wgrad(one * 4, one * 4, one, None, 32, 128, 64, 4096, one * 4, need - 256, None) == -6
    assert lib.cdn_codenet_scale_backward(None, one, one, one, one, 1, 4, 8, 8, None) == -1
    # decode: heat_out partially overlapping heat is an argument error
    rc = lib.cdn_ctdet_decode(one * 64, one, None, 1, 2, 8, 8, 0, 10, 1, one * 64 + 64, one, one * 64, 1 << 20, None)
    assert rc == -1 and b"overlaps" in lib.cdn_last_error()


def test_cover_frozen_ranges_only_widens_and_restores_the_running_flags():
    """pipeline.cover_frozen_ranges (the calibration step of the byte-code serving mode): every QuantAct range ends up
    covering what the frozen network feeds it (+ margin), never narrower than before; running_stat flags are restored;
    a shared QuantAct (called more than once per forward) covers the extremes of all its inputs."""
    import torch.nn as nn
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.shared = QuantAct(8), QuantAct(8)

        def forward(self, x):
            y = self.a(x)
            return self.shared(y * 3.0) + self.shared(y - 5.0)      # one QuantAct, two different tensors

    net = Net()
    x = torch.linspace(-1.0, 2.0, 64).reshape(1, 1, 8, 8)
    net(x)                                                   # running update: ranges initialised from this batch
    net.a.x_min.fill_(-0.5); net.a.x_max.fill_(1.0)          # too narrow on both sides
    net.shared.x_min.fill_(-100.0); net.shared.x_max.fill_(0.5)   # wide below (must stay), narrow above
    flags = [a.running_stat for a in (net.a, net.shared)]
    moved = pipeline.cover_frozen_ranges(net, [x], margin=0.05, passes=3)
    assert moved == 2 and [a.running_stat for a in (net.a, net.shared)] == flags
    assert net.a.x_min.item() <= -1.0 and net.a.x_max.item() >= 2.0
    assert net.shared.x_min.item() == -100.0                 # never narrowed
    with torch.no_grad():
        pipeline.set_running_stat(net, False)
        y = net.a(x)
        assert net.shared.x_max.item() >= (y * 3.0).max().item() and net.shared.x_min.item() <= (y - 5.0).min().item()
    # a second call changes nothing
    before = [t.clone() for t in (net.a.x_min, net.a.x_max, net.shared.x_min, net.shared.x_max)]
    pipeline.set_running_stat(net, True)
    assert pipeline.cover_frozen_ranges(net, [x], margin=0.0) == 0
    assert all(torch.equal(p, q) for p, q in zip(before, (net.a.x_min, net.a.x_max, net.shared.x_min, net.shared.x_max)))


def test_cover_frozen_ranges_reports_the_spread_of_the_per_batch_extremes():
    """Round 6 (the serving tail policy, calibrate_serving(sigmas=...)): cover_frozen_ranges(spread=...) leaves, per QuantAct,
    the standard deviations of its per-BATCH minimum and maximum over the calibration batches -- a QuantAct called twice per
    forward contributes ONE pair per batch (the extremes over both calls); one batch gives no statistic."""
    import torch.nn as nn
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.shared = QuantAct(8), QuantAct(8)

        def forward(self, x):
            y = self.a(x)
            return self.shared(y * 3.0) + self.shared(y - 5.0)

    net = Net()
    base = torch.linspace(-1.0, 2.0, 64).reshape(1, 1, 8, 8)
    net(base)
    net.a.x_min.fill_(-50.0); net.a.x_max.fill_(50.0)        # wide: the frozen QuantAct passes values through its fine grid
    net.shared.x_min.fill_(-500.0); net.shared.x_max.fill_(500.0)
    scales = [1.0, 2.0, 4.0, 3.0]
    batches = [base * k for k in scales]
    spread = {}
    pipeline.cover_frozen_ranges(net, batches, margin=0.0, passes=1, spread=spread)
    lo_a, hi_a = spread[id(net.a)]
    want_hi = torch.tensor([2.0 * k for k in scales], dtype=torch.float64).std().item()
    want_lo = torch.tensor([-1.0 * k for k in scales], dtype=torch.float64).std().item()
    assert abs(hi_a - want_hi) < 1e-6 and abs(lo_a - want_lo) < 1e-6
    lo_s, hi_s = spread[id(net.shared)]
    # shared: per batch the maximum is that of y * 3 (= 6 k up to the 100/255 grid), the minimum the smaller of the two
    # calls' minima (-3 k and -k - 5)
    want_lo_s = torch.tensor([min(-3.0 * k, -k - 5.0) for k in scales], dtype=torch.float64).std().item()
    assert abs(hi_s - 3 * want_hi) < 0.5 and abs(lo_s - want_lo_s) < 0.5
    one = {}
    pipeline.cover_frozen_ranges(net, batches[:1], margin=0.0, passes=1, spread=one)
    assert one == {}


def test_overflow_flags_name_the_quantacts_of_the_launch_that_saturated():
    """pipeline.OverflowFlags (round 4: the byte-code schedules give every launch its own overflow word so that a
    saturated code names the QuantAct to widen): words, views for a consumer numbering its launches from 0, reset."""
    from codenet_amd import pipeline
    f = pipeline.OverflowFlags(6, torch.device("cpu"))
    a, b, c = object(), object(), object()
    f.name(0, [a])
    f.name(1, [a, b])
    heads = f.slice(4)
    heads.name(1, [c])                                   # word 5 of the parent
    assert heads.count() == 2 and heads.ptr(1) == f.ptr(5) and f.who[5] == [c]
    assert not f.any() and f.acts() == []
    f.words[1] = 1
    f.words[5] = 1
    got = f.acts(reset=False)
    assert got == [a, b, c] and f.any(reset=False)
    assert f.acts() == [a, b, c] and not f.any() and int(f.words.sum()) == 0


def test_cover_frozen_ranges_refuses_a_model_whose_quantacts_are_never_called():
    """on a model that runs a fused schedule the QuantAct modules are never called, the hooks never fire and
    nothing was widened -- silently.  Now it raises (calibrate on the module path, or use calibrate_serving)."""
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, planes=[8, 4])
    x = torch.randn(1, 8, 4, 4)
    with pytest.raises(RuntimeError, match="no QuantAct was called"):
        pipeline.cover_frozen_ranges(net, [x], forward=lambda b: None)


def test_pmc_steady_reads_the_steady_state_iterations(tmp_path):
    """tools/pmc_steady.py (bench.py's live roofline.traffic and profiles/<round>/pmc_steady_*.json): iterations are
    delimited by the marker kernel, the tail after the last marker is cut, FETCH_SIZE counts double."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import pmc_steady
    d = tmp_path / "run" / "x"
    d.mkdir(parents=True)
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    names = ["setup", "void (anonymous namespace)::scale_nchw_kernel(float)", "dw0p_kernel<true, false>(x)",
             "pwi8_kernel<64, 128, 2, true>(y)"]
    seq = [0] + [1, 2, 3] * 4 + [1, 2]                     # four whole iterations, then a cut one
    for i, k in enumerate(seq):
        rows.append('%d,"%s",FETCH_SIZE,%g' % (i + 1, names[k], {0: 5.0, 1: 10.0, 2: 20.0, 3: 40.0}[k]))
    (d / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    data = pmc_steady.load(str(tmp_path / "run"), "FETCH_SIZE")
    rows2, spans = pmc_steady.iterations(data, "scale_nchw", 1, 2, ["setup"])
    assert len(spans) == 2 and all(hi - lo == 3 for lo, hi in spans)
    per_iter = sum(v for lo, hi in spans for _, v in rows2[lo:hi]) / len(spans)
    assert per_iter == 70.0


def test_kblocked_weight_codes_layout_and_flag_validation():
    """CDN_X_WCODES_KB (include/codenet_dcn.h): the library names the column count / byte offset of the k-blocked copy,
    folded_int8_kblocked builds exactly that layout behind the row-major codes, and the flag is refused where the
    library has no such form."""
    from codenet_amd import _native, pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantBnConv2d
    lib = _native.lib()
    assert lib.cdn_codenet_wcodes_kb_columns(1024, 256) == 256 and lib.cdn_codenet_wcodes_kb_columns(2153, 256) == 256
    assert lib.cdn_codenet_wcodes_kb_columns(512, 100) == 128 and lib.cdn_codenet_wcodes_kb_columns(512, 64) == 64
    assert lib.cdn_codenet_wcodes_kb_columns(256, 128) == 0          # short K: the tile kernel keeps it
    assert lib.cdn_codenet_wcodes_kb_columns(1024, 257) == 0
    assert lib.cdn_codenet_wcodes_kb_offset(2153, 256) == 256 * 2176
    assert lib.cdn_codenet_wcodes_kb_offset(515, 3) == (3 * 576 + 255) // 256 * 256
    torch.manual_seed(0)
    conv, bn = nn.Conv2d(520, 70, 1, bias=False), nn.BatchNorm2d(70)
    q = QuantBnConv2d(weight_bit=4, quant_mode="symmetric", per_channel=True)
    q.set_param(conv, bn)
    q.eval()
    with torch.no_grad():
        codes, scale, colsum = q.folded_int8()
        cols, off = lib.cdn_codenet_wcodes_kb_columns(520, 70), lib.cdn_codenet_wcodes_kb_offset(520, 70)
        assert cols == 128 and codes.shape == (70, 576)
        buf, scale2, colsum2 = q.folded_int8_kblocked(cols, off)
        assert scale2 is scale and colsum2 is colsum
        assert buf.dtype == torch.int8 and buf.numel() == off + 576 * cols
        assert torch.equal(buf[:70 * 576].view(70, 576), codes)
        kb = buf[off:].view(576 // 32, cols, 32)
        for co in (0, 13, 69):
            assert torch.equal(kb[:, co, :].reshape(-1), codes[co])
        assert int(kb[:, 70:, :].abs().sum()) == 0
        # cached: same buffer until the weights change, then rewritten IN PLACE (captured graphs keep the address)
        again = q.folded_int8_kblocked(cols, off)[0]
        assert again.data_ptr() == buf.data_ptr()
        conv.weight.mul_(-1.0)
        new = q.folded_int8_kblocked(cols, off)[0]
        assert new.data_ptr() == buf.data_ptr()
        codes_n = q.folded_int8()[0]
        assert torch.equal(new[off:].view(576 // 32, cols, 32)[:, 5, :].reshape(-1), codes_n[5])
        i8, flag = pipeline.stage_int8_codes(q)
        assert flag == pipeline.WCODES_KB and i8[0].data_ptr() == buf.data_ptr()
    # the flag where no k-blocked form exists: an argument error before any HIP call
    one = 4096
    rc = lib.cdn_codenet_stage_fused_forward(one, 1 | pipeline.WCODES_KB, 0, None, 1, 256, 128, 8, 8, one, None, -7.0, 8.0,
                                             one, one, one, one, one, None, None, None, 1, *([None] * 9), 8, 0.99, 0,
                                             one, 1 << 20, one, None)
    assert rc == -1 and b"CDN_X_WCODES_KB" in lib.cdn_last_error()


_ASM_OK = """
_Z4demoILi1EEvPf: ; @demo
\ts_load_dwordx2 s[4:5], s[0:1], 0x0
\t;;#ASMSTART
\tglobal_load_dwordx4 v[2:5], v1, s[4:5] offset:0
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_lds_dwordx4 v9, s[6:7]
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\tv_add_u32_e32 v7, 1, v7
.LBB0_2:
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tv_mov_b32_e32 v6, v2
\ts_endpgm
.Lfunc_end0:
"""


def test_asm_load_checker_finds_a_read_before_the_wait(tmp_path):
    """tools/check_asm_loads.py (the guard of pwi8s_kernel's inline-asm loads): clean on a correct sequence, and it
    reports a copy placed between the load and the wait, a wait whose count does not cover the load, and a hazard that
    exists on one branch only; the shipped library's own assembly is checked when the build left it behind."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_loads as C
    ok = tmp_path / "ok.s"
    ok.write_text(_ASM_OK)
    assert C.check(str(ok), "demo", verbose=False) == (0, 1)
    bad = tmp_path / "copy.s"
    bad.write_text(_ASM_OK.replace("\ts_cbranch_scc1 .LBB0_2\n", "\tv_mov_b64_e32 v[10:11], v[4:5]\n\ts_cbranch_scc1 .LBB0_2\n"))
    assert C.check(str(bad), "demo", verbose=False)[0] == 1
    lenient = tmp_path / "lenient.s"
    lenient.write_text(_ASM_OK.replace("vmcnt(1)", "vmcnt(2)"))
    assert C.check(str(lenient), "demo", verbose=False)[0] == 1
    branch = tmp_path / "branch.s"          # the DMA on the fall-through path only: behind the branch the count is short
    dma = "\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v9, s[6:7]\n\t;;#ASMEND\n"
    branch.write_text(_ASM_OK.replace(dma, "").replace("\tv_add_u32_e32 v7, 1, v7\n", dma))
    assert C.check(str(branch), "demo", verbose=False)[0] == 1
    shipped = os.path.join(ROOT, "build", "asm", "codenet_fused-hip-amdgcn-amd-amdhsa-gfx950.s")
    if os.path.exists(shipped):
        nbad, nloads = C.check(shipped, "pwi8s_kernel", verbose=False)
        assert nbad == 0 and nloads >= 12


def test_argument_validation_of_the_round5_training_entry_points_without_gpu():
    """The integer-form 1x1-convolution entry points of the QAT step and the multi-tensor weight prep validate before any
    HIP call; the plan functions are pure host arithmetic."""
    import ctypes
    from codenet_amd import _native
    lib = _native.lib()
    one = 4096
    # shapes: C % 32 == 0, HW % 32 == 0, Co <= 512
    assert lib.cdn_codenet_pointwise_i8_supported(32, 1024, 256, 256) == 1
    assert lib.cdn_codenet_pointwise_i8_supported(32, 1000, 256, 256) == 0
    assert lib.cdn_codenet_pointwise_i8_supported(32, 1024, 256, 100) == 0
    assert lib.cdn_codenet_pointwise_i8_supported(32, 1024, 600, 256) == 0
    need = lib.cdn_codenet_pointwise_i8_workspace_bytes(32, 1024, 256, 256)
    # k-blocked int8 codes + scales + sums + the transposed bf16 codes + reciprocal scales, each 256-byte aligned
    assert need >= 1024 * 256 + 2 * 256 * 4 + 16 * 1024 * 32 + 256 * 4 and need % 256 == 0
    assert lib.cdn_codenet_pointwise_i8_workspace_bytes(32, 1000, 256, 256) == 0
    # stage 0 splits K over the four waves of a workgroup (one workgroup per pixel block and column group), stage 2 not
    assert lib.cdn_codenet_pointwise_i8_range_partials(32, 1024, 256, 256) == 32 * (256 // 32) * 2
    assert lib.cdn_codenet_pointwise_i8_range_partials(32, 128, 64, 4096) == 32 * (4096 // 32 // 4)
    rc = lib.cdn_codenet_pointwise_i8_forward_range(one, None, one, None, one, 32, 1024, 256, 256, None, one, need, None)
    assert rc == -1                                              # no QuantAct state
    rc = lib.cdn_codenet_pointwise_i8_forward_range(one, one, one, None, one, 32, 1000, 256, 256, None, 4096, need, None)
    assert rc != 0 and b"C % 32" in lib.cdn_last_error()
    rc = lib.cdn_codenet_pointwise_i8_forward_range(one, one, one, None, one, 32, 1024, 256, 256, None, 4096, 16, None)
    assert rc != 0 and b"workspace" in lib.cdn_last_error()
    assert lib.cdn_codenet_pointwise_dgrad_q4_supported(32, 1024, 256, 256) == 1
    assert lib.cdn_codenet_pointwise_dgrad_q4(one, None, one, 32, 1024, 256, 256, None) == -1
    # multi-tensor weight prep: 1 .. 8 tensors, a BN fold needs its four companions
    P, I64, I, F = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1, ctypes.c_float * 1
    args = (P(one), I64(8), I64(16))
    tail = (I(4), I(1), I(1), F(1.0), P(one))
    assert lib.cdn_codenet_weight_prep_multi(0, *args, None, None, None, None, *tail, None, None) == -1
    assert lib.cdn_codenet_weight_prep_multi(9, *args, None, None, None, None, *tail, None, None) == -1
    rc = lib.cdn_codenet_weight_prep_multi(1, *args, P(one), None, None, None, *tail, None, None)
    assert rc == -1 and b"BN fold" in lib.cdn_last_error()
    rc = lib.cdn_codenet_weight_prep_multi(1, P(one), I64(8), I64(16), None, None, None, None, I(9), I(1), I(1), F(1.0), P(one),
                                           None, None)
    assert rc == -1 and b"bits" in lib.cdn_last_error()
    # reproducible gather backward: workspace = grad_s partials [chunks][N][plane] + grad_w partials [N][C][9]
    need = lib.cdn_codenet_dw_backward_workspace_bytes(32, 256, 32, 32, 1)          # stored plane 16 x 16
    assert need >= (32 * 256 + 32 * 256 * 9) * 4 and (need - 32 * 256 * 9 * 4) % (32 * 256 * 4) == 0
    need = lib.cdn_codenet_dw_backward_workspace_bytes(32, 1024, 16, 16, 0)
    assert need >= (32 * 256 + 32 * 1024 * 9) * 4
    assert lib.cdn_codenet_dw_backward_workspace_bytes(32, 256, 33, 32, 1) == 0     # odd size: no stored form
    assert lib.cdn_codenet_dw_backward_workspace_bytes(2, 8, 512, 512, 0) == 0      # plane beyond LDS
    assert lib.cdn_codenet_dw_backward_r(one, one, one, one, one, one, one, 2, 8, 16, 16, None, None) == -1
    assert b"workspace" in lib.cdn_last_error()
    assert lib.cdn_codenet_dw_up2_backward_r(one, one, one, one, one, one, one, 2, 8, 16, 16, None, None) == -1


def test_split_stage_call_flags_are_validated_without_gpu():
    """CDN_X_DEFER_RANGE / CDN_X_PHASE_* (round 5, global-range mode on the fused stages) and
    cdn_quantact_commit_range validate before any HIP call."""
    from codenet_amd import _native, pipeline
    lib = _native.lib()
    one = 4096
    assert (pipeline.DEFER_RANGE, pipeline.PHASE_SCALE, pipeline.PHASE_GATHER, pipeline.PHASE_POINTWISE) == \
        (0x1000, 0x2000, 0x4000, 0x8000)
    assert lib.cdn_quantact_commit_range(None, one, one, one, 8, 0.99, 1, None) == -1
    assert lib.cdn_quantact_commit_range(one, one, one, one, 1, 0.99, 1, None) == -1 and b"bits" in lib.cdn_last_error()

    def call(flags, running, with_acts=True):
        a = [one] * 9 if with_acts else [None] * 9
        return lib.cdn_codenet_stage_fused_forward(
            one, flags, 0, None, 2, 8, 8, 8, 8, one, None, -7.0, 8.0, one, one, None, None, None, None, None, None, 1,
            *a, 8, 0.99, running, one, 1 << 30, one, None)
    assert call(pipeline.DEFER_RANGE | pipeline.PHASE_SCALE, 0) == -1 and b"CDN_X_DEFER_RANGE" in lib.cdn_last_error()
    assert call(pipeline.DEFER_RANGE, 1, with_acts=False) == -1 and b"CDN_X_DEFER_RANGE" in lib.cdn_last_error()
    assert call(pipeline.DEFER_RANGE | pipeline.ACT_PERCENTILE, 1) == -1
    assert call(0x10000, 1) == -1 and b"x_nhwc" in lib.cdn_last_error()
