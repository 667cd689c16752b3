"""codenet_amd -- MI355X (gfx950) implementation of CoDeNet's co-designed deformable-convolution
hot path behind the reference's own operator API.

    codenet_amd.functions.dcn_deform_conv   deform_conv / modulated_deform_conv autograd functions
    codenet_amd.modules.dcn_deform_conv     DeformConv, ModulatedDeformConv, ...,
                                            DeformConvWithOffsetScaleBoundPositive
    codenet_amd.portable_quantizer          QuantAct, Quant_Conv2d, QuantBnConv2d, QuantDeformConv2d,
                                            QuantDeformConvWithOffsetScaleBoundPositive, ...
    codenet_amd._ext.dcn.dcn_deform_conv_cuda   tensor-level shim over the C ABI (include/codenet_dcn.h)
"""
__version__ = "0.1.0"
