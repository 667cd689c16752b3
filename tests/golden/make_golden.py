"""Generates tests/golden/*.npz.  RUNS IN THE BUILD CONTAINER ONLY (needs /root/reference).

The reference's own Python modules are imported from /root/reference (nothing is copied): stubs
are installed for its missing third-party imports (pytorchcv, thop) and for the CUDA extension
``_ext.dcn.dcn_deform_conv_cuda`` (absent: .MISSING_LARGE_BLOBS), and the module-level name
``deform_conv`` is rebound to the CPU oracle (oracle/dcn.py) -- the only thing of the reference
that cannot run here is its native arithmetic.

Fixtures
  quant_ref.npz     PURE reference (no oracle involved): QuantAct state/codes over 4 calls,
                    Quant_Conv2d / QuantBnConv2d / QuantDeformConv2d weight fake-quantisation
                    (with and without --wt-percentile), AsymmetricQuantFunction outputs.
  stage_fp32.npz    reference DeformConvWithOffsetScaleBoundPositive (native call -> oracle).
  stage_w4a8.npz    reference QuantDeformConvWithOffsetScaleBoundPositive + following
                    Sequential(ReLU, QuantAct), 3 consecutive forwards (EMA state pinned).
  model_io.npz      F5: the reference's whole PoseShuffleNetV2 at 256x256 (image + flip), fp32 and
                    W4A8, through CtdetDetector.process' body: sub-sampled hm/wh/reg, checksums and
                    the decoded detections [1,100,6] (weights: codenet_amd.harness.fill_state_dict_).
  model_io_512.npz / model_noise_512.npz   the same two at the BASELINE resolution 512x512 (seed 52).
  model_deform_backbone.npz   the reference's PoseShuffleNetV2(deform=True), fp32, 128x128, one image.
  model_noise.npz   the reference W4A8 model against ITSELF with 1 vs 8 CPU threads (code-flip noise floor).
  voc_eval_ref.npz  the reference's own VOC evaluator (tools/voc_eval_lib/datasets/voc_eval.py, pure numpy) over a
                    synthetic VOC tree: rec / prec / AP per class (difficult objects, duplicates, partial recall).
  quant_extra.npz   the rest of the reference's quantiser class surface, from the reference classes themselves:
                    QuantBnDeformConv2d (with / without --wt-percentile), QuantDeformConvWithOffsetScaleBoundPositiveBn
                    (2 forwards, native call -> oracle), QuantSflUnit (down-sampling + plain unit chained, 2 forwards;
                    pytorchcv's ChannelShuffle / ShuffleUnit are not installed: the unit is a plain container with
                    the attribute names QuantSflUnit reads, the shuffle is the published view / transpose / view),
                    and `linear_errors`: the exception every QuantLinear configuration raises in the reference.
  deform_raw.npz    oracle-only regression vectors for the generic op (fwd + all grads, plain
                    and modulated); the reference cannot produce these (CUDA-only).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import dcn as O  # noqa: E402


def import_reference():
    """Import the reference's hot-path Python modules with stubs for what is missing."""
    for name in ["pytorchcv", "pytorchcv.model_provider", "pytorchcv.models",
                 "pytorchcv.models.shufflenetv2", "pytorchcv.models.common", "thop", "_ext",
                 "_ext.dcn", "_ext.dcn.dcn_deform_conv_cuda"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pytorchcv.model_provider"].get_model = lambda *a, **k: None
    sys.modules["pytorchcv.models.shufflenetv2"].ShuffleUnit = type("ShuffleUnit", (), {})
    sys.modules["pytorchcv.models.common"].ChannelShuffle = type("ChannelShuffle", (), {})
    sys.modules["thop"].profile = lambda *a, **k: (0, 0)
    sys.modules["_ext"].dcn = sys.modules["_ext.dcn"]
    sys.modules["_ext.dcn"].dcn_deform_conv_cuda = sys.modules["_ext.dcn.dcn_deform_conv_cuda"]
    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, REF)
    import models.external.modules.dcn_deform_conv as ref_mod
    import portable_quantizer.quant_modules as ref_qm
    import portable_quantizer.quantization_utils.quant_utils as ref_qu
    ref_mod.deform_conv = O.deform_conv      # native symbol -> CPU oracle
    ref_qm.deform_conv = O.deform_conv
    return ref_mod, ref_qm, ref_qu


def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
            for k, v in d.items()}


def make_quant_ref(ref_qm, ref_qu):
    g = torch.Generator().manual_seed(11)
    out = {}
    # QuantAct: 4 calls (first: "+=" initialisation, then EMA), asymmetric 8 bit
    qa = ref_qm.QuantAct(8, quant_mode="asymmetric")
    qa.eval()
    for it in range(4):
        x = torch.randn(2, 6, 7, 5, generator=g) * (1.0 + 0.7 * it) + 0.25 * it
        y = qa(x)
        scale, zp = ref_qu.asymmetric_linear_quantization_params(8, qa.x_min, qa.x_max)
        q = ref_qu.linear_quantize(x, scale, zp)
        out["act_x%d" % it] = x
        out["act_y%d" % it] = y
        out["act_q%d" % it] = q
        out["act_min%d" % it] = qa.x_min.clone()
        out["act_max%d" % it] = qa.x_max.clone()
    # percentile-mode QuantAct, 2 calls
    qp = ref_qm.QuantAct(8, quant_mode="asymmetric", percentile=True)
    for it in range(2):
        x = torch.randn(2, 8, 16, 16, generator=g) * 2
        out["pact_x%d" % it] = x
        out["pact_y%d" % it] = qp(x)
        out["pact_min%d" % it] = qp.x_min.clone()
        out["pact_max%d" % it] = qp.x_max.clone()
    # weight fake-quantisation through the three conv wrappers
    for pct in (False, True):
        tag = "p" if pct else "n"
        conv = torch.nn.Conv2d(40, 6, 1, bias=True)
        conv.weight.data = torch.randn(6, 40, 1, 1, generator=g) * 0.3
        conv.bias.data = torch.randn(6, generator=g)
        qc = ref_qm.Quant_Conv2d(4, quant_mode="symmetric", per_channel=True, weight_percentile=pct)
        qc.set_param(conv)
        x = torch.randn(2, 40, 5, 5, generator=g)
        out["qconv_%s_w" % tag] = conv.weight.data
        out["qconv_%s_b" % tag] = conv.bias.data
        out["qconv_%s_x" % tag] = x
        out["qconv_%s_y" % tag] = qc(x)
        # big-L channel so the kthvalue branch (L >= 10, and k > 1 at L = 1200) is exercised
        conv2 = torch.nn.Conv2d(1200, 3, 1, bias=False)
        conv2.weight.data = torch.randn(3, 1200, 1, 1, generator=g) * 0.1
        bn = torch.nn.BatchNorm2d(3)
        bn.weight.data = torch.rand(3, generator=g) + 0.5
        bn.bias.data = torch.randn(3, generator=g) * 0.1
        bn.running_mean = torch.randn(3, generator=g) * 0.1
        bn.running_var = torch.rand(3, generator=g) + 0.5
        qb = ref_qm.QuantBnConv2d(4, quant_mode="symmetric", per_channel=True, weight_percentile=pct)
        qb.set_param(conv2, bn)
        x2 = torch.randn(2, 1200, 3, 3, generator=g)
        out["qbn_%s_w" % tag] = conv2.weight.data
        for k in ("weight", "bias", "running_mean", "running_var"):
            out["qbn_%s_bn_%s" % (tag, k)] = getattr(bn, k).data if k in ("weight", "bias") else getattr(bn, k)
        out["qbn_%s_x" % tag] = x2
        out["qbn_%s_y" % tag] = qb(x2)
        # depthwise 3x3 weights: L = 9 < 10 -> the 0.95 rule under --wt-percentile
        wd = torch.randn(5, 1, 3, 3, generator=g) * 0.2
        w_min = wd.view(5, -1).min(dim=1).values * (0.95 if pct else 1.0)
        w_max = wd.view(5, -1).max(dim=1).values * (0.95 if pct else 1.0)
        out["qdw_%s_w" % tag] = wd
        out["qdw_%s_wq" % tag] = ref_qu.SymmetricQuantFunction.apply(wd, 4, w_min, w_max, True, pct)
    return t2n(out)


def _stage_params(C, Co, g):
    return dict(
        w_scale=torch.randn(1, C, 1, 1, generator=g) * (5.0 / C ** 0.5),   # s ~ N(1, 5): both clamps hit
        b_scale=torch.ones(1),
        w_dw=torch.empty(C, 1, 3, 3).uniform_(-1, 1, generator=g) / 3.0,
        w_pw=torch.randn(Co, C, 1, 1, generator=g) * (2.0 / C) ** 0.5,
    )


def _ref_stage(ref_mod, C, Co, p):
    m = ref_mod.DeformConvWithOffsetScaleBoundPositive(C, Co, 3, 1, 1, groups=Co, hidden_state=128)
    with torch.no_grad():
        m.conv_scale.weight.copy_(p["w_scale"])
        m.conv_scale.bias.copy_(p["b_scale"])
        m.conv.weight.copy_(p["w_dw"])
        m.conv_channel.weight.copy_(p["w_pw"])
    return m


def make_stage_fp32(ref_mod):
    g = torch.Generator().manual_seed(21)
    N, C, Co, H, W = 2, 16, 8, 12, 12
    p = _stage_params(C, Co, g)
    m = _ref_stage(ref_mod, C, Co, p).eval()
    x = torch.randn(N, C, H, W, generator=g)
    with torch.no_grad():
        y = m(x)
        s = m.conv_bound(m.conv_scale(x))
    # gradients through the reference composition (oracle backward under it)
    xg = x.clone().requires_grad_(True)
    m.zero_grad()
    yo = m(xg)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go)
    out = dict(p, x=x, y=y, s=s, go=go, gx=xg.grad, g_w_scale=m.conv_scale.weight.grad,
               g_b_scale=m.conv_scale.bias.grad, g_w_dw=m.conv.weight.grad,
               g_w_pw=m.conv_channel.weight.grad)
    return t2n(out)


def make_stage_w4a8(ref_mod, ref_qm):
    g = torch.Generator().manual_seed(31)
    N, C, Co, H, W = 2, 16, 8, 12, 12
    out = {}
    for pct in (False, True):
        tag = "p" if pct else "n"
        p = _stage_params(C, Co, g)
        m = _ref_stage(ref_mod, C, Co, p)
        bn = torch.nn.BatchNorm2d(Co)
        bn.weight.data = torch.rand(Co, generator=g) + 0.5
        bn.bias.data = torch.randn(Co, generator=g) * 0.1
        bn.running_mean = torch.randn(Co, generator=g) * 0.1
        bn.running_var = torch.rand(Co, generator=g) + 0.5
        q = ref_qm.QuantDeformConvWithOffsetScaleBoundPositive(
            4, 8, act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="asymmetric",
            per_channel=True, weight_percentile=pct)
        q.set_param(m, bn)
        post = torch.nn.Sequential(torch.nn.ReLU(inplace=True),
                                   ref_qm.QuantAct(8, quant_mode="asymmetric"))
        q.eval()
        for k, v in p.items():
            out["%s_%s" % (tag, k)] = v
        for k in ("weight", "bias"):
            out["%s_bn_%s" % (tag, k)] = getattr(bn, k).data
        out["%s_bn_running_mean" % tag] = bn.running_mean
        out["%s_bn_running_var" % tag] = bn.running_var
        cap = {}
        q.quant_act.register_forward_hook(lambda mod, i, o: cap.__setitem__("s", o.clone()))
        q.quant_deform_conv.register_forward_hook(lambda mod, i, o: cap.__setitem__("d", o.clone()))
        q.quant_identity_deform.register_forward_hook(lambda mod, i, o: cap.__setitem__("dq", o.clone()))
        for it in range(3):
            x = torch.randn(N, C, H, W, generator=g) * (1.0 + 0.3 * it)
            with torch.no_grad():
                y = q(x)
                r = post(y.clone())
            out["%s_x%d" % (tag, it)] = x
            out["%s_y%d" % (tag, it)] = y
            out["%s_r%d" % (tag, it)] = r
            for k in ("s", "d", "dq"):
                out["%s_%s%d" % (tag, k, it)] = cap[k]
            out["%s_smin%d" % (tag, it)] = q.quant_act[1].x_min.clone()
            out["%s_smax%d" % (tag, it)] = q.quant_act[1].x_max.clone()
            out["%s_dmin%d" % (tag, it)] = q.quant_identity_deform.x_min.clone()
            out["%s_dmax%d" % (tag, it)] = q.quant_identity_deform.x_max.clone()
            out["%s_rmin%d" % (tag, it)] = post[1].x_min.clone()
            out["%s_rmax%d" % (tag, it)] = post[1].x_max.clone()
    return t2n(out)


def make_stage_w4a8_grads(ref_mod, ref_qm):
    """cfg5 (BASELINE configs[4], quant_main.py QAT step): the reference's W4A8 stage
    [QuantDeformConvWithOffsetScaleBoundPositive, Sequential(ReLU, QuantAct), Upsample x2] in train() mode,
    two consecutive forward + backward passes (the QuantAct ranges move between them).  Gradients flow through
    the straight-through estimators (quant_utils.py:202-204,227-229), the Hardtanh, the BN fold
    (quant_modules.py:365-372) and -- rebound to the oracle -- the native backward
    (dcn_deform_conv_cuda.cpp:260-484)."""
    g = torch.Generator().manual_seed(41)
    N, C, Co, H, W = 2, 16, 8, 12, 12
    out = {}
    for pct in (False, True):
        tag = "p" if pct else "n"
        p = _stage_params(C, Co, g)
        m = _ref_stage(ref_mod, C, Co, p)
        bn = torch.nn.BatchNorm2d(Co)
        bn.weight.data = torch.rand(Co, generator=g) + 0.5
        bn.bias.data = torch.randn(Co, generator=g) * 0.1
        bn.running_mean = torch.randn(Co, generator=g) * 0.1
        bn.running_var = torch.rand(Co, generator=g) + 0.5
        q = ref_qm.QuantDeformConvWithOffsetScaleBoundPositive(
            4, 8, act_percentile=False, wt_quant_mode="symmetric", act_quant_mode="asymmetric",
            per_channel=True, weight_percentile=pct)
        q.set_param(m, bn)
        post = torch.nn.Sequential(torch.nn.ReLU(inplace=True), ref_qm.QuantAct(8, quant_mode="asymmetric"))
        up = torch.nn.Upsample(scale_factor=2, mode="nearest")
        net = torch.nn.Sequential(q, post, up).train()
        for k, v in p.items():
            out["%s_%s" % (tag, k)] = v
        for k in ("weight", "bias"):
            out["%s_bn_%s" % (tag, k)] = getattr(bn, k).data.clone()
        out["%s_bn_running_mean" % tag] = bn.running_mean.clone()
        out["%s_bn_running_var" % tag] = bn.running_var.clone()
        params = {"g_w_scale": q.quant_conv_scale.weight, "g_b_scale": q.quant_conv_scale.bias,
                  "g_w_dw": q.quant_deform_conv.weight, "g_w_pw": q.quant_conv_channel_bn.conv.weight,
                  "g_bn_weight": q.quant_conv_channel_bn.bn.weight, "g_bn_bias": q.quant_conv_channel_bn.bn.bias}
        for it in range(2):
            x = (torch.randn(N, C, H, W, generator=g) * (1.0 + 0.3 * it)).requires_grad_(True)
            net.zero_grad()
            y = net(x)
            go = torch.randn(y.shape, generator=g)
            y.backward(go)
            out["%s_x%d" % (tag, it)] = x.detach()
            out["%s_y%d" % (tag, it)] = y.detach()
            out["%s_go%d" % (tag, it)] = go
            out["%s_gx%d" % (tag, it)] = x.grad
            for k, prm in params.items():
                out["%s_%s%d" % (tag, k, it)] = prm.grad.clone()
            out["%s_smin%d" % (tag, it)] = q.quant_act[1].x_min.clone()
            out["%s_smax%d" % (tag, it)] = q.quant_act[1].x_max.clone()
            out["%s_dmax%d" % (tag, it)] = q.quant_identity_deform.x_max.clone()
            out["%s_rmax%d" % (tag, it)] = post[1].x_max.clone()
    return t2n(out)


def make_head_w4a8(ref_qm):
    """The reference's QuantDepthwiseNode (quant_modules.py:1013-1071) on a head-shaped nn.Sequential
    (shufflenetv2_dcn.py:244-262), 3 consecutive forwards (EMA state), plus the fp32 Sequential."""
    g = torch.Generator().manual_seed(57)
    N, C, classes, H, W = 2, 16, 5, 10, 12
    nn = torch.nn

    def bn(c):
        b = nn.BatchNorm2d(c)
        b.weight.data = torch.rand(c, generator=g) + 0.5
        b.bias.data = torch.randn(c, generator=g) * 0.1
        b.running_mean = torch.randn(c, generator=g) * 0.1
        b.running_var = torch.rand(c, generator=g) + 0.5
        return b
    head = nn.Sequential(
        nn.Conv2d(C, C, 1, bias=False), bn(C), nn.ReLU(inplace=True),
        nn.Conv2d(C, C, 3, 1, 1, groups=C, bias=False), bn(C), nn.ReLU(inplace=True),
        nn.Conv2d(C, classes, 1, bias=True)).eval()
    head[0].weight.data = torch.randn(C, C, 1, 1, generator=g) * (1.5 / C) ** 0.5
    head[3].weight.data = torch.randn(C, 1, 3, 3, generator=g) / 3
    head[6].weight.data = torch.randn(classes, C, 1, 1, generator=g) * (1.0 / C) ** 0.5
    head[6].bias.data = torch.randn(classes, generator=g) * 0.1
    out = {"w1": head[0].weight.data, "w2": head[3].weight.data, "w3": head[6].weight.data,
           "b3": head[6].bias.data}
    for i, k in ((1, "bn1"), (4, "bn2")):
        out[k + "_weight"] = head[i].weight.data
        out[k + "_bias"] = head[i].bias.data
        out[k + "_mean"] = head[i].running_mean
        out[k + "_var"] = head[i].running_var
    xs = [torch.randn(N, C, H, W, generator=g).abs() * (1.0 + 0.3 * it) for it in range(3)]
    with torch.no_grad():
        out["fp32_out"] = head(xs[0].clone())
    for pct in (False, True):
        tag = "p" if pct else "n"
        import copy
        q = ref_qm.QuantDepthwiseNode(4, 8, act_percentile=False, wt_quant_mode="symmetric",
                                      act_quant_mode="asymmetric", per_channel=True,
                                      weight_percentile=pct)
        q.set_param(copy.deepcopy(head))
        q.eval()
        cap = {}
        q.quant_act1.register_forward_hook(lambda mod, i, o: cap.__setitem__("y1q", o.clone()))
        q.quant_act3.register_forward_hook(lambda mod, i, o: cap.__setitem__("y2q", o.clone()))
        for it in range(3):
            with torch.no_grad():
                y = q(xs[it].clone())
            out["x%d" % it] = xs[it]
            out["%s_out%d" % (tag, it)] = y
            out["%s_y1q%d" % (tag, it)] = cap["y1q"]
            out["%s_y2q%d" % (tag, it)] = cap["y2q"]
            out["%s_a1min%d" % (tag, it)] = q.quant_act1[1].x_min.clone()
            out["%s_a1max%d" % (tag, it)] = q.quant_act1[1].x_max.clone()
            out["%s_a3min%d" % (tag, it)] = q.quant_act3[1].x_min.clone()
            out["%s_a3max%d" % (tag, it)] = q.quant_act3[1].x_max.clone()
    return t2n(out)


def make_base_nodes(ref_qm):
    """Two chained reference QuantBaseNode units (stride 2 then stride 1, quant_modules.py:809-907) sharing
    one block-output QuantAct, 3 consecutive forwards.  The reference BaseNode container
    (shufflenetv2_dcn.py:57-114) is rebuilt from plain nn layers with the same b1 / b2 layout."""
    nn = torch.nn
    g = torch.Generator().manual_seed(73)

    def bn(c):
        b = nn.BatchNorm2d(c)
        b.weight.data = torch.rand(c, generator=g) + 0.5
        b.bias.data = torch.randn(c, generator=g) * 0.1
        b.running_mean = torch.randn(c, generator=g) * 0.1
        b.running_var = torch.rand(c, generator=g) + 0.5
        return b

    def conv(i, o, k=1, s=1, groups=1):
        c = nn.Conv2d(i, o, k, s, k // 2, groups=groups, bias=False)
        c.weight.data = torch.randn(c.weight.shape, generator=g) * (1.5 / (i // groups * k * k)) ** 0.5
        return c

    class Node(nn.Module):
        def __init__(self, inp, oup, stride):
            super().__init__()
            self.stride = stride
            h = oup // 2
            cin = inp if stride == 2 else h
            self.b2 = nn.Sequential(conv(cin, h), bn(h), nn.ReLU(inplace=True), conv(h, h, 3, stride, h), bn(h),
                                    conv(h, h), bn(h), nn.ReLU(inplace=True))
            if stride == 2:
                self.b1 = nn.Sequential(conv(inp, inp, 3, 2, inp), bn(inp), conv(inp, h), bn(h),
                                        nn.ReLU(inplace=True))
    inp, oup = 8, 20
    nodes = [Node(inp, oup, 2).eval(), Node(oup, oup, 1).eval()]
    out = {}
    for u, nd in enumerate(nodes):
        for i, k in ((0, "1"), (3, "2"), (5, "3")):
            out["u%d_w%s" % (u, k)] = nd.b2[i].weight.data
        for i, k in ((1, "1"), (4, "2"), (6, "3")):
            b = nd.b2[i]
            out["u%d_bn%s" % (u, k)] = torch.stack([b.weight.data, b.bias.data, b.running_mean, b.running_var])
        if nd.stride == 2:
            out["u%d_w4" % u], out["u%d_w5" % u] = nd.b1[0].weight.data, nd.b1[2].weight.data
            for i, k in ((1, "4"), (3, "5")):
                b = nd.b1[i]
                out["u%d_bn%s" % (u, k)] = torch.stack([b.weight.data, b.bias.data, b.running_mean, b.running_var])
    shared = ref_qm.QuantAct(8, quant_mode="asymmetric")
    qn = []
    for nd in nodes:
        q = ref_qm.QuantBaseNode(4, 8, act_percentile=False, wt_quant_mode="symmetric",
                                 act_quant_mode="asymmetric", per_channel=True, weight_percentile=False)
        q.set_param(nd)
        q.set_act(shared)
        qn.append(q.eval())
    for it in range(3):
        x = torch.randn(2, inp, 12, 10, generator=g).abs() * (1.0 + 0.3 * it)
        with torch.no_grad():
            y0 = qn[0](x.clone())
            y1 = qn[1](y0.clone())
        out["x%d" % it], out["y0_%d" % it], out["y1_%d" % it] = x, y0, y1
        out["shared_min%d" % it], out["shared_max%d" % it] = shared.x_min.clone(), shared.x_max.clone()
        for u, q in enumerate(qn):
            for k in ("quant_act1", "quant_act2") + (("quant_act4",) if q.stride == 2 else ()):
                a = getattr(q, k)
                out["u%d_%s_min%d" % (u, k, it)], out["u%d_%s_max%d" % (u, k, it)] = a.x_min.clone(), a.x_max.clone()
    return t2n(out)


def make_quant_extra(ref_mod, ref_qm):
    nn = torch.nn
    g = torch.Generator().manual_seed(91)
    out = {}

    def bn(c):
        b = nn.BatchNorm2d(c)
        b.weight.data = torch.rand(c, generator=g) + 0.5
        b.bias.data = torch.randn(c, generator=g) * 0.1
        b.running_mean = torch.randn(c, generator=g) * 0.1
        b.running_var = torch.rand(c, generator=g) + 0.5
        return b

    def bn_arr(b):
        return torch.stack([b.weight.data, b.bias.data, b.running_mean, b.running_var])

    def conv(i, o, k=1, s=1, groups=1):
        c = nn.Conv2d(i, o, k, s, k // 2, groups=groups, bias=False)
        c.weight.data = torch.randn(c.weight.shape, generator=g) * (1.5 / (i // groups * k * k)) ** 0.5
        return c

    # ---- QuantBnDeformConv2d: dense (groups = 1) and depthwise, arbitrary offsets, with / without --wt-percentile
    for tag, (C, Co, groups, pct) in {"dense": (6, 8, 1, False), "dw": (8, 8, 8, False), "densep": (6, 8, 1, True)}.items():
        dc = ref_mod.DeformConv(C, Co, 3, 1, 1, 1, groups, 1, bias=False)
        dc.weight.data = torch.randn(dc.weight.shape, generator=g) * 0.3
        dc.bias = None          # the reference class reads conv.bias (:566), which its own DeformConv does not have
        b = bn(Co)
        q = ref_qm.QuantBnDeformConv2d(4, quant_mode="symmetric", per_channel=True, weight_percentile=pct)
        q.set_param(dc, b)
        x = torch.randn(2, C, 9, 7, generator=g)
        off = torch.randn(2, 18, 9, 7, generator=g) * 1.5
        with torch.no_grad():
            y = q(x, off)
        out.update({"bd_%s_w" % tag: dc.weight.data, "bd_%s_bn" % tag: bn_arr(b), "bd_%s_x" % tag: x,
                    "bd_%s_off" % tag: off, "bd_%s_y" % tag: y})

    # ---- QuantDeformConvWithOffsetScaleBoundPositiveBn: the CoDeNet operator, BN folded into the deformable conv
    C = 16
    p = _stage_params(C, C, g)
    m = ref_mod.DeformConvWithOffsetScaleBoundPositive(C, C, 3, 1, 1, groups=C)     # in == out: no pointwise conv
    with torch.no_grad():
        m.conv_scale.weight.copy_(p["w_scale"])
        m.conv_scale.bias.copy_(p["b_scale"])
        m.conv.weight.copy_(p["w_dw"])
    m.conv.bias = None
    b = bn(C)
    q = ref_qm.QuantDeformConvWithOffsetScaleBoundPositiveBn(4, 8, wt_quant_mode="symmetric", act_quant_mode="asymmetric",
                                                             per_channel=True)
    q.set_param(m, b)
    q.eval()
    out.update({"pb_w_scale": p["w_scale"], "pb_b_scale": p["b_scale"], "pb_w_dw": p["w_dw"], "pb_bn": bn_arr(b)})
    for it in range(2):
        x = torch.randn(2, C, 12, 12, generator=g) * (1.0 + 0.4 * it)
        with torch.no_grad():
            y = q(x)
        out["pb_x%d" % it], out["pb_y%d" % it] = x, y
        out["pb_smin%d" % it], out["pb_smax%d" % it] = q.quant_act[1].x_min.clone(), q.quant_act[1].x_max.clone()

    # ---- QuantSflUnit: pytorchcv's unit as a plain container; ChannelShuffle = view / transpose / view
    class Shuffle(nn.Module):
        def __init__(self, channels, groups):
            super().__init__()
            self.groups = groups

        def forward(self, x):
            n, c, h, w = x.shape
            return x.view(n, self.groups, c // self.groups, h, w).transpose(1, 2).contiguous().view(n, c, h, w)
    ref_qm.ChannelShuffle = Shuffle

    class Unit:
        def __init__(self, inp, oup, downsample):
            self.downsample, self.use_se, self.use_residual = downsample, False, False
            mid = oup // 2
            cin = inp if downsample else mid
            self.compress_conv1, self.compress_bn1 = conv(cin, mid), bn(mid)
            self.dw_conv2, self.dw_bn2 = conv(mid, mid, 3, 2 if downsample else 1, mid), bn(mid)
            self.expand_conv3, self.expand_bn3 = conv(mid, mid), bn(mid)
            if downsample:
                self.dw_conv4, self.dw_bn4 = conv(inp, inp, 3, 2, inp), bn(inp)
                self.expand_conv5, self.expand_bn5 = conv(inp, mid), bn(mid)
    units = [Unit(8, 20, True), Unit(20, 20, False)]
    names = ["compress_conv1", "dw_conv2", "expand_conv3", "dw_conv4", "expand_conv5"]
    for u, un in enumerate(units):
        for k in names:
            if hasattr(un, k):
                out["sfl%d_%s" % (u, k)] = getattr(un, k).weight.data
                kb = k.replace("conv", "bn")
                out["sfl%d_%s" % (u, kb)] = bn_arr(getattr(un, kb))
    shared = ref_qm.QuantAct(8, quant_mode="asymmetric")
    qs = []
    for un in units:
        q = ref_qm.QuantSflUnit(4, 8, wt_quant_mode="symmetric", act_quant_mode="asymmetric", per_channel=True)
        q.set_param(un)
        q.set_act(shared)
        qs.append(q.eval())
    for it in range(2):
        x = torch.randn(2, 8, 10, 12, generator=g).abs() * (1.0 + 0.3 * it)
        with torch.no_grad():
            y0 = qs[0](x.clone())
            y1 = qs[1](y0.clone())
        out["sfl_x%d" % it], out["sfl_y0_%d" % it], out["sfl_y1_%d" % it] = x, y0, y1
        out["sfl_shared%d" % it] = torch.cat([shared.x_min, shared.x_max])

    # ---- QuantLinear: what the reference does with each configuration
    errs = []
    for kw in (dict(per_channel=True), dict(per_channel=False), dict(per_channel=True, weight_percentile=True),
               dict(per_channel=False, weight_percentile=True), dict(per_channel=True, group_quantization=True, group_number=4),
               dict(per_channel=True, quant_mode="asymmetric")):
        for io in ((16, 16), (16, 8), (64, 32)):
            try:
                ref_qm.QuantLinear(4, io[0], io[1], **kw)(torch.randn(3, io[0], generator=g))
                errs.append("ok")
            except Exception as e:      # noqa: BLE001
                errs.append(type(e).__name__)
    out["linear_errors"] = np.array(errs)
    return t2n(out)


def make_decode():
    """The reference's ctdet_decode (lib/models/decode.py:474-505) on random score maps; the selected
    scores are distinct floats, so torch.topk's unspecified tie order does not matter."""
    from models.decode import ctdet_decode as ref_decode
    g = torch.Generator().manual_seed(91)
    out = {}
    for tag, (B, cat, H, W, K, spec, use_reg) in {
            "a": (2, 20, 32, 32, 100, False, True), "b": (1, 3, 17, 23, 40, True, False),
            "c": (3, 5, 8, 8, 10, False, True)}.items():
        heat = torch.sigmoid(torch.randn(B, cat, H, W, generator=g) * 2 - 2)
        wh = torch.rand(B, 2 * cat if spec else 2, H, W, generator=g) * 10
        reg = torch.rand(B, 2, H, W, generator=g) if use_reg else None
        dets = ref_decode(heat.clone(), wh, reg=reg, cat_spec_wh=spec, K=K)
        out[tag + "_cfg"] = np.array([B, cat, H, W, K, int(spec), int(use_reg)])
        out[tag + "_heat"], out[tag + "_wh"], out[tag + "_dets"] = heat, wh, dets
        if use_reg:
            out[tag + "_reg"] = reg
    return t2n(out)


def make_deform_raw():
    g = torch.Generator().manual_seed(41)
    out = {}
    cases = {"a": (2, 8, 9, 9, 8, 3, 1, 1, 1, 8, 1), "b": (1, 6, 8, 10, 4, 3, 2, 1, 1, 2, 3)}
    for tag, (N, C, H, W, Co, k, s, p, d, G, DG) in cases.items():
        Ho, Wo = O.out_size(H, W, k, k, s, p, d)
        x = torch.randn(N, C, H, W, generator=g)
        off = torch.randn(N, DG * 2 * k * k, Ho, Wo, generator=g) * 2
        off[0, :, 0, 0] = 50.0          # far out of range -> zero samples, zero grads
        w = torch.randn(Co, C // G, k, k, generator=g)
        m = torch.rand(N, DG * k * k, Ho, Wo, generator=g)
        b = torch.randn(Co, generator=g)
        go = torch.randn(N, Co, Ho, Wo, generator=g)
        cfg = (s, p, d, G, DG)
        out[tag + "_cfg"] = np.array([N, C, H, W, Co, k, s, p, d, G, DG])
        out.update({tag + "_x": x, tag + "_off": off, tag + "_w": w, tag + "_m": m, tag + "_b": b,
                    tag + "_go": go})
        out[tag + "_y"] = O.deform_conv_forward(x, off, w, *cfg)
        gx, goff = O.deform_conv_backward_input(x, off, w, go, *cfg)
        out[tag + "_gx"], out[tag + "_goff"] = gx, goff
        out[tag + "_gw"] = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg)
        out[tag + "_my"] = O.deform_conv_forward(x, off, w, *cfg, mask=m, bias=b)
        mgx, mgoff, mgm = O.deform_conv_backward_input(x, off, w, go, *cfg, mask=m)
        mgw, mgb = O.deform_conv_backward_params(x, off, tuple(w.shape), go, *cfg, mask=m,
                                                 with_bias=True)
        out.update({tag + "_mgx": mgx, tag + "_mgoff": mgoff, tag + "_mgm": mgm, tag + "_mgw": mgw,
                    tag + "_mgb": mgb})
    return t2n(out)


def make_model_io(ref_qm, res=256, seed=51):
    """F5: the reference's whole PoseShuffleNetV2 (256x256, image + W-flip, seed-filled weights) in
    fp32 and W4A8 through the body of CtdetDetector.process; native deform_conv -> oracle.
    res=512 (model_io_512.npz): the same at the BASELINE resolution (VERDICT r3 "next" #5)."""
    import models.networks.shufflenetv2_dcn as ref_net
    from models.decode import ctdet_decode as ref_decode
    from models.utils import flip_tensor
    from portable_quantizer import quantize_shufflenetv2_dcn as ref_quantize
    from codenet_amd.harness import fill_state_dict_
    heads = {"hm": 20, "wh": 2, "reg": 2}
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, res, res, generator=g)
    images = torch.cat([img, torch.flip(img, [3])], dim=0)
    out = {"image_seed": np.array(seed), "res": np.array(res)}     # images are re-generated from the seed by the tests
    for tag, quant in (("fp32", False), ("w4a8", True)):
        net = ref_net.PoseShuffleNetV2(heads, 64)
        fill_state_dict_(net, 317)
        if quant:
            ref_quantize(net, 4, None, 8, "symmetric", "asymmetric", True, False, False, False)
        net.eval()
        n_fwd = 3 if quant else 1          # W4A8: three forwards so QuantAct ranges have EMA history
        with torch.no_grad():
            for it in range(n_fwd):
                o = net(images)[-1]
            hm = o["hm"].clone().sigmoid_()
            wh, reg = o["wh"], o["reg"]
            hm_m = (hm[0:1] + flip_tensor(hm[1:2])) / 2
            wh_m = (wh[0:1] + flip_tensor(wh[1:2])) / 2
            dets = ref_decode(hm_m, wh_m, reg=reg[0:1], cat_spec_wh=False, K=100)
        for k in ("hm", "wh", "reg"):
            t = o[k]
            out["%s_%s_sub" % (tag, k)] = t[:, :, ::4, ::4].contiguous()      # 1/16 of the pixels
            out["%s_%s_sum" % (tag, k)] = t.double().sum()
            out["%s_%s_abs" % (tag, k)] = t.double().abs().sum()
        out["%s_dets" % tag] = dets
        out["%s_nfwd" % tag] = np.array(n_fwd)
    return t2n(out)


def make_model_deform_backbone(res=128, seed=53):
    """The reference's PoseShuffleNetV2(deform=True) -- CoDeNet operators as the 3x3 convs of every backbone unit
    (shufflenetv2_dcn.py:216-230; stride-2 operators included) -- in fp32, one image; native deform_conv -> oracle.
    (Its W4A8 form cannot be produced: QuantBaseNodeDeform raises in set_param.)"""
    import models.networks.shufflenetv2_dcn as ref_net
    from codenet_amd.harness import fill_state_dict_
    heads = {"hm": 20, "wh": 2, "reg": 2}
    net = ref_net.PoseShuffleNetV2(heads, 64, deform=True)
    fill_state_dict_(net, 317)
    net.eval()
    img = torch.randn(1, 3, res, res, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        o = net(img)[-1]
    out = {"image_seed": np.array(seed), "res": np.array(res)}
    for k in ("hm", "wh", "reg"):
        out[k] = o[k]
    return t2n(out)


def make_model_noise(ref_qm, res=256, seed=51):
    """The REFERENCE against itself: its W4A8 PoseShuffleNetV2 (same weights / images as model_io.npz) run with
    1 and with 8 CPU threads.  Only the summation order inside torch's CPU convolutions changes, yet 8-bit codes
    flip where a value sits on a rounding boundary and ~70 re-quantising layers amplify the flips.  These
    statistics are the yardstick for any other implementation of the same network (tests/test_harness.py): a
    correct one differs from the reference by about as much as the reference differs from itself."""
    import copy
    import models.networks.shufflenetv2_dcn as ref_net
    from models.decode import ctdet_decode as ref_decode
    from models.utils import flip_tensor
    from portable_quantizer import quantize_shufflenetv2_dcn as ref_quantize
    from codenet_amd.harness import fill_state_dict_
    heads = {"hm": 20, "wh": 2, "reg": 2}
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, res, res, generator=g)
    images = torch.cat([img, torch.flip(img, [3])], dim=0)
    net = ref_net.PoseShuffleNetV2(heads, 64)
    fill_state_dict_(net, 317)
    ref_quantize(net, 4, None, 8, "symmetric", "asymmetric", True, False, False, False)
    net.eval()
    runs = {}
    for threads in (1, 8):
        torch.set_num_threads(threads)
        O.set_threads(threads)
        m = copy.deepcopy(net)
        with torch.no_grad():
            for _ in range(3):
                o = m(images)[-1]
            hm = o["hm"].clone().sigmoid_()
            dets = ref_decode((hm[0:1] + flip_tensor(hm[1:2])) / 2, (o["wh"][0:1] + flip_tensor(o["wh"][1:2])) / 2,
                              reg=o["reg"][0:1], cat_spec_wh=False, K=100)
        runs[threads] = ({k: o[k].clone() for k in heads}, dets)
    torch.set_num_threads(1)
    O.set_threads(1)
    out = {}
    for k in heads:
        d = (runs[1][0][k] - runs[8][0][k]).abs().flatten()
        out["noise_mean_" + k] = d.mean()
        out["noise_p99_" + k] = torch.quantile(d, 0.99)
        out["noise_max_" + k] = d.max()
        out["std_" + k] = runs[1][0][k].std()
    a, b = runs[8][1][0], runs[1][1][0]
    ca = torch.stack([(a[:, 0] + a[:, 2]) / 2, (a[:, 1] + a[:, 3]) / 2], 1)
    cb = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2], 1)
    hit = sum(int(((a[:, 5] == b[i, 5]) & ((ca - cb[i]).abs().max(dim=1).values <= 0.5)
                   & ((a[:, 4] - b[i, 4]).abs() <= 2e-2)).any()) for i in range(b.shape[0]))
    out["self_agreement"] = np.array(hit / b.shape[0])
    return t2n(out)


def make_voc_eval():
    """The reference's OWN evaluator -- tools/voc_eval_lib/datasets/voc_eval.py::voc_eval / voc_ap, pure numpy,
    imported from /root/reference -- over a small synthetic VOC tree written to a temporary directory (XML annotations,
    image-set file, per-class detection files in the reference's formats): partial recall, difficult objects (hit and
    missed), duplicate detections of one box, detections on images without that class, overlaps just above / below
    0.5, a class without detections, a class whose every detection is wrong.  Scores are distinct (the reference's
    argsort leaves ties unspecified).  The only shim: `np.bool`, an alias numpy >= 1.24 removed (voc_eval.py:139-141).
    Stored: the inputs (boxes, difficult flags, detections) and the reference's rec / prec / AP (VOC07 11-point and
    area form) per class."""
    import importlib.util
    import tempfile
    if not hasattr(np, "bool"):
        np.bool = bool                       # removed alias the reference still uses
    spec = importlib.util.spec_from_file_location(
        "ref_voc_eval", os.path.join(REF, "tools", "voc_eval_lib", "datasets", "voc_eval.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(2007)
    classes = ["car", "dog", "person", "sofa", "bird"]
    n_img = 14
    gt = []           # rows: image, class, x1, y1, x2, y2, difficult
    for im in range(n_img):
        for c in range(3):                                   # sofa: one object only; bird: objects but no detections
            for _ in range(rng.randint(0, 4)):
                x1, y1 = rng.randint(0, 300, 2)
                w, h = rng.randint(20, 160, 2)
                gt.append((im, c, x1, y1, x1 + w, y1 + h, int(rng.rand() < 0.25)))
    gt.append((3, 3, 40, 40, 200, 180, 0))
    gt += [(5, 4, 10, 10, 90, 120, 0), (6, 4, 30, 50, 130, 160, 1)]
    gt = np.array(gt, dtype=np.int64)
    dets = []         # rows: image, class, score, x1, y1, x2, y2
    score = iter(rng.permutation(4000)[:2000] / 4000.0 + 1e-4)          # distinct scores
    for row in gt:
        im, c, x1, y1, x2, y2, _d = row
        if c > 2:
            continue
        r = rng.rand()
        if r < 0.2:
            continue                                           # missed object (partial recall)
        for _ in range(1 if r < 0.7 else 2):                   # duplicates of one box
            j = rng.uniform(-0.18, 0.18, 4) * np.array([x2 - x1, y2 - y1, x2 - x1, y2 - y1])
            dets.append((im, c, next(score), x1 + j[0], y1 + j[1], x2 + j[2], y2 + j[3]))
    for _ in range(25):                                        # false positives, some on images without that class
        im, c = rng.randint(0, n_img), rng.randint(0, 3)
        x1, y1 = rng.uniform(0, 350, 2)
        dets.append((im, c, next(score), x1, y1, x1 + rng.uniform(15, 120), y1 + rng.uniform(15, 120)))
    # IoU exactly around the threshold on a known box: GT 100 x 100 px (+1 convention: 101 x 101)
    gt = np.concatenate([gt, np.array([(13, 0, 200, 200, 300, 300, 0), (13, 1, 200, 200, 300, 300, 0)])])
    dets.append((13, 0, next(score), 200.0, 200.0, 300.0, 249.0))      # 101*50 / 101*101 = 0.495 -> FP
    dets.append((13, 1, next(score), 200.0, 200.0, 300.0, 251.0))      # 101*52 / 101*101 = 0.515 -> TP
    dets += [(3, 3, next(score), 300.0, 300.0, 380.0, 380.0), (7, 3, next(score), 10.0, 10.0, 50.0, 50.0)]   # sofa: all wrong
    dets = np.array(dets, dtype=np.float64)
    out = {"classes": np.array(classes), "n_images": np.array(n_img), "gt": gt, "dets": dets}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "Annotations"))
        names = ["%06d" % i for i in range(n_img)]
        with open(os.path.join(tmp, "test.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        for im, name in enumerate(names):
            objs = "".join(
                "<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
                "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                % (classes[r[1]], r[6], r[2], r[3], r[4], r[5]) for r in gt if r[0] == im)
            with open(os.path.join(tmp, "Annotations", name + ".xml"), "w") as f:
                f.write("<annotation><filename>%s.jpg</filename>%s</annotation>" % (name, objs))
        for c, cname in enumerate(classes):
            with open(os.path.join(tmp, "det_%s.txt" % cname), "w") as f:
                for r in dets[dets[:, 1] == c]:
                    f.write("%s %.6f %.3f %.3f %.3f %.3f\n" % (names[int(r[0])], r[2], r[3], r[4], r[5], r[6]))
        for c, cname in enumerate(classes):
            for tag, use07 in (("07", True), ("area", False)):
                with np.errstate(all="ignore"):
                    rec, prec, ap = ref.voc_eval(os.path.join(tmp, "det_{:s}.txt"),
                                                 os.path.join(tmp, "Annotations", "{:s}.xml"),
                                                 os.path.join(tmp, "test.txt"), cname,
                                                 os.path.join(tmp, "cache"), ovthresh=0.5, use_07_metric=use07)
                out["%s_ap_%s" % (cname, tag)] = np.array(ap, dtype=np.float64)
            out["%s_rec" % cname] = np.asarray(rec, dtype=np.float64)
            out["%s_prec" % cname] = np.asarray(prec, dtype=np.float64)
    # the detection files round coordinates to 3 and scores to 6 decimals: store what the reference actually read
    out["dets"] = np.concatenate([dets[:, :2], np.round(dets[:, 2:3], 6), np.round(dets[:, 3:], 3)], axis=1)
    return out


def make_post_process():
    """The reference's own lib/utils/image.py::transform_preds / get_affine_transform / affine_transform and
    lib/utils/post_process.py::ctdet_post_process, imported from /root/reference, on random detections with square and
    non-square crops.  OpenCV is not installed here; the ONE cv2 function on this path, cv2.getAffineTransform, is given by
    its definition -- the unique affine map through three point pairs, solved in float64 (what OpenCV does) -- in a stub
    module; everything else (the three-point construction in float32, get_dir / get_3rd_point, the per-point loop, the
    class dictionaries) is the reference's code."""
    import importlib.util
    cv2 = types.ModuleType("cv2")

    def get_affine_transform(src, dst):
        src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
        a = np.concatenate([src, np.ones((3, 1))], axis=1)          # [3, 3]: rows (x, y, 1)
        return np.linalg.solve(a, dst).T                             # [2, 3]
    cv2.getAffineTransform = get_affine_transform
    had = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    try:
        pkg = types.ModuleType("ref_utils")
        pkg.__path__ = [os.path.join(REF, "lib", "utils")]
        sys.modules["ref_utils"] = pkg
        mods = {}
        for name in ("image", "ddd_utils", "post_process"):
            spec = importlib.util.spec_from_file_location("ref_utils." + name,
                                                          os.path.join(REF, "lib", "utils", name + ".py"))
            mods[name] = importlib.util.module_from_spec(spec)
            sys.modules["ref_utils." + name] = mods[name]
            spec.loader.exec_module(mods[name])
    finally:
        if had is not None:
            sys.modules["cv2"] = had
        else:
            del sys.modules["cv2"]
    img, pp = mods["image"], mods["post_process"]
    rng = np.random.RandomState(11)
    out = {}
    cases = [((250.5, 187.0), 512.0, (128, 128)), ((320.0, 240.0), 640.0, (128, 128)),
             ((100.0, 333.5), np.array([480.0, 480.0], np.float32), (160, 96)), ((64.0, 64.0), 128.0, (32, 32))]
    for i, (c, sc, osz) in enumerate(cases):
        pts = (rng.rand(40, 2) * np.array(osz)).astype(np.float64)
        out["tp%d_pts" % i] = pts
        out["tp%d_center" % i] = np.array(c, np.float32)
        out["tp%d_scale" % i] = np.asarray(sc, np.float32)
        out["tp%d_osize" % i] = np.array(osz)
        out["tp%d_out" % i] = img.transform_preds(pts, np.array(c, np.float32), sc, osz)
    B, K, ncls = 3, 50, 20
    dets = np.zeros((B, K, 6), np.float32)
    dets[:, :, :2] = rng.rand(B, K, 2) * 100
    dets[:, :, 2:4] = dets[:, :, :2] + rng.rand(B, K, 2) * 28
    dets[:, :, 4] = rng.rand(B, K)
    dets[:, :, 5] = rng.randint(0, ncls, (B, K))
    c = np.array([[256.0, 200.0], [300.5, 187.5], [111.0, 222.0]], np.float32)
    s = np.array([512.0, 600.0, 448.0], np.float32)
    out["pp_dets"], out["pp_c"], out["pp_s"] = dets.copy(), c, s
    res = pp.ctdet_post_process(dets.copy(), c, s, 128, 128, ncls)
    for b in range(B):
        for j in range(1, ncls + 1):
            out["pp_out_%d_%d" % (b, j)] = np.asarray(res[b][j], np.float32).reshape(-1, 5)
    return out


def main():
    assert os.path.isdir(REF), "needs the reference checkout at /root/reference"
    torch.manual_seed(317)
    torch.set_num_threads(1)
    ref_mod, ref_qm, ref_qu = import_reference()
    makers = {
        "quant_ref": lambda: make_quant_ref(ref_qm, ref_qu),
        "stage_fp32": lambda: make_stage_fp32(ref_mod),
        "stage_w4a8": lambda: make_stage_w4a8(ref_mod, ref_qm),
        "stage_w4a8_grads": lambda: make_stage_w4a8_grads(ref_mod, ref_qm),
        "deform_raw": make_deform_raw,
        "model_io": lambda: make_model_io(ref_qm),
        "model_noise": lambda: make_model_noise(ref_qm),
        "model_io_512": lambda: make_model_io(ref_qm, res=512, seed=52),
        "model_noise_512": lambda: make_model_noise(ref_qm, res=512, seed=52),
        "model_deform_backbone": make_model_deform_backbone,
        "head_w4a8": lambda: make_head_w4a8(ref_qm),
        "decode_ref": make_decode,
        "base_nodes": lambda: make_base_nodes(ref_qm),
        "quant_extra": lambda: make_quant_extra(ref_mod, ref_qm),
        "voc_eval_ref": make_voc_eval,
        "post_process_ref": make_post_process,
    }
    only = sys.argv[1:] or list(makers)           # `make_golden.py stage_w4a8_grads` regenerates one fixture
    for name in only:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **makers[name]())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
