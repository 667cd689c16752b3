"""Fake-quantisation operator API of the hot path: same import surface as the reference's
``portable_quantizer`` package (portable_quantizer/__init__.py:1-2)."""
from .quantization_utils.quantize_model import quantize_shufflenetv2_dcn, quantize_deform_stages
from .quantization_utils.quant_utils import SymmetricQuantFunction, AsymmetricQuantFunction
