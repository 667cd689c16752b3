"""In-place module surgery that turns a PoseShuffleNetV2 into its W4A8 form: same entry point,
argument list and resulting module tree (checkpoint keys) as the reference's
portable_quantizer/quantization_utils/quantize_model.py:7-82."""
import torch.nn as nn

from ..quant_modules import (QuantAct, QuantBaseNode, QuantBaseNodeDeform, QuantBnConv2d, QuantDepthwiseNode,
                             QuantDeformConvWithOffsetScaleBoundPositive)

__all__ = ["quantize_shufflenetv2_dcn", "quantize_deform_stages"]


def _relu_quant(activ, bits, act_percentile, *tail):
    return nn.Sequential(activ, QuantAct(bits, quant_mode="asymmetric", percentile=act_percentile),
                         *tail)


def quantize_deform_stages(model, quant_conv, quant_act, wt_quant_mode, act_quant_mode,
                           wt_per_channel, wt_percentile, act_percentile):
    """The hot-path part (reference :70-82): every [deform, BN, ReLU, Upsample] quadruple of
    ``model.deconv_layers`` becomes [QuantDeform(+BN folded), Sequential(ReLU, QuantAct), Upsample]."""
    deform = model.deconv_layers
    assert len(deform) % 4 == 0
    mods = []
    for i in range(len(deform) // 4):
        op, bn, relu, up = deform[4 * i], deform[4 * i + 1], deform[4 * i + 2], deform[4 * i + 3]
        q = QuantDeformConvWithOffsetScaleBoundPositive(
            quant_conv, quant_act, act_percentile=act_percentile, wt_quant_mode=wt_quant_mode,
            act_quant_mode=act_quant_mode, per_channel=wt_per_channel,
            weight_percentile=wt_percentile)
        q.set_param(op, bn)
        mods += [q, _relu_quant(relu, quant_act, act_percentile), up]
    model.deconv_layers = nn.Sequential(*mods)
    return model


def quantize_shufflenetv2_dcn(model, quant_conv, quant_bn, quant_act, wt_quant_mode, act_quant_mode,
                              wt_per_channel, wt_percentile, act_percentile, deform_backbone,
                              w2=False, maxpool=False):
    """quant_conv / quant_act: weight / activation bit widths (layer0 always uses 8-bit weights,
    reference :28); quant_bn, w2 are accepted for signature parity and unused, as in the
    reference."""
    # (deform_backbone=True cannot run in the reference -- QuantBaseNodeDeform raises in set_param -- and its model
    # factory never builds such a backbone; here it maps the nodes of PoseShuffleNetV2(deform=True) as the text says)
    wkw = dict(quant_mode=wt_quant_mode, per_channel=wt_per_channel, weight_percentile=wt_percentile)
    ckw = dict(act_percentile=act_percentile, wt_quant_mode=wt_quant_mode,
               act_quant_mode=act_quant_mode, per_channel=wt_per_channel,
               weight_percentile=wt_percentile)

    l0 = model.layer0
    q0 = QuantBnConv2d(8, **wkw)
    q0.set_param(l0[0], l0[1])
    tail = (l0[3],) if maxpool else ()
    model.layer0 = nn.Sequential(q0, _relu_quant(l0[2], quant_act, act_percentile, *tail))

    for idx in (1, 2, 3):
        layer = getattr(model, "layer%d" % idx)
        shared = QuantAct(quant_act, quant_mode="asymmetric", percentile=act_percentile)
        nodes = []
        for node in layer.children():
            qn = (QuantBaseNodeDeform if deform_backbone else QuantBaseNode)(quant_conv, quant_act, **ckw)
            qn.set_param(node)
            qn.set_act(shared)
            nodes.append(qn)
        setattr(model, "layer%d" % idx, nn.Sequential(*nodes))

    l4 = model.layer4
    q4 = QuantBnConv2d(quant_conv, **wkw)
    q4.set_param(l4[0], l4[1])
    model.layer4 = nn.Sequential(q4, _relu_quant(l4[2], quant_act, act_percentile))

    for head in model.heads:
        qh = QuantDepthwiseNode(quant_conv, quant_act, **ckw)
        qh.set_param(getattr(model, head))
        setattr(model, head, qh)

    quantize_deform_stages(model, quant_conv, quant_act, wt_quant_mode, act_quant_mode,
                           wt_per_channel, wt_percentile, act_percentile)
