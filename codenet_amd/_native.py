"""ctypes binding of libcodenet_dcn.so (include/codenet_dcn.h).

The library is the product: if it is missing this module raises -- there is no CPU or PyTorch
fallback anywhere in codenet_amd (the reference has none either: its functions raise
NotImplementedError for non-CUDA tensors, functions/dcn_deform_conv.py:43-45).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libcodenet_dcn.so")      # the product library; no environment override
CSRC = os.path.join(_HERE, "csrc")

CDN_F32, CDN_F64, CDN_F16 = 0, 1, 2
_lib = None

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

_SIGNATURES = {
    "cdn_abi_version": (ctypes.c_int, []),
    "cdn_last_error": (ctypes.c_char_p, []),
    "cdn_deform_conv_forward": (_i, [_vp] * 4 + [_i] + [_i64] * 5 + [_i] * 10 + [_vp]),
    "cdn_deform_conv_forward_scratch_bytes": (ctypes.c_size_t, [_i64] * 5 + [_i] * 10),
    "cdn_deform_conv_forward_scratch": (_i, [_vp] * 4 + [_i] + [_i64] * 5 + [_i] * 10 + [_vp, ctypes.c_size_t, _vp]),
    "cdn_deform_conv_backward_input": (_i, [_vp] * 6 + [_i] + [_i64] * 5 + [_i] * 10 + [_vp]),
    "cdn_deform_conv_backward_input_scratch_bytes": (ctypes.c_size_t, [_i64] * 5 + [_i] * 10),
    "cdn_deform_conv_backward_input_scratch_min_bytes": (ctypes.c_size_t, [_i64] * 5 + [_i] * 10),
    "cdn_deform_conv_backward_input_scratch": (_i, [_vp] * 6 + [_i] + [_i64] * 5 + [_i] * 10 + [_vp, ctypes.c_size_t, _vp]),
    "cdn_deform_conv_backward_parameters": (_i, [_vp] * 4 + [_i] + [_i64] * 5 + [_i] * 10 + [_f, _vp]),
    "cdn_modulated_deform_conv_forward": (_i, [_vp] * 6 + [_i] + [_i64] * 5 + [_i] * 11 + [_vp]),
    "cdn_modulated_deform_conv_backward": (_i, [_vp] * 11 + [_i] + [_i64] * 5 + [_i] * 11 + [_vp]),
    "cdn_codenet_scale_forward": (_i, [_vp] * 4 + [_i64] * 4 + [_f, _f, _vp]),
    "cdn_codenet_dw_forward": (_i, [_vp] * 4 + [_i64] * 4 + [_vp]),
    "cdn_codenet_dw_backward_supported": (_i, [_i64, _i64]),
    "cdn_codenet_pointwise_wgrad_workspace_bytes": (ctypes.c_size_t, [_i64] * 4),
    "cdn_codenet_pointwise_wgrad": (_i, [_vp] * 4 + [_i64] * 4 + [_vp, ctypes.c_size_t, _vp]),
    "cdn_codenet_scale_backward": (_i, [_vp] * 5 + [_i64] * 4 + [_vp]),
    "cdn_codenet_scale_range_partials": (_i64, [_i64] * 3),
    "cdn_codenet_scale_forward_range": (_i, [_vp] * 4 + [_i64] * 4 + [_f, _f, _vp, _vp]),
    "cdn_codenet_dw_range_partials": (_i64, [_i64] * 4),
    "cdn_codenet_dw_forward_range": (_i, [_vp] * 4 + [_i64] * 4 + [_vp, _vp]),
    "cdn_codenet_pointwise_range_partials": (_i64, [_i64] * 3),
    "cdn_codenet_pointwise_forward_range": (_i, [_vp] * 7 + [_i64] * 4 + [_i, _vp, _vp]),
    "cdn_quantact_forward_partials": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp]),
    "cdn_codenet_pointwise_wgrad_q": (_i, [_vp] * 5 + [_i64] * 4 + [_vp, ctypes.c_size_t, _vp]),
    "cdn_codenet_pointwise_wgrad_q_f32": (_i, [_vp] * 5 + [_i64] * 4 + [_vp, ctypes.c_size_t, _vp]),
    "cdn_quantact_relu_up2_forward_partials": (
        _i, [_vp, _vp] + [_i64] * 3 + [_vp] * 4 + [_i64, _i, _d, _i, _vp]),
    "cdn_codenet_scale_backward_masked": (_i, [_vp] * 3 + [_f, _f] + [_vp] * 3 + [_i64] * 4 + [_vp]),
    "cdn_codenet_weight_prep_backward": (_i, [_vp] * 7 + [_i64, _i64] + [_vp] * 5),
    "cdn_codenet_weight_prep": (_i, [_vp, _i64, _i64] + [_vp] * 4 + [_i] + [_vp] * 3),
    "cdn_codenet_weight_prep_multi": (_i, [_i] + [_vp] * 14),
    "cdn_codenet_weight_prep_ranked": (_i, [_vp, _i64, _i64] + [_vp] * 4 + [_i, _i, _i, _f] + [_vp] * 3),
    "cdn_codenet_stage_supported": (_i, [_i64] * 4 + [_i, _i]),
    "cdn_codenet_stage_fused_supported": (_i, [_i64] * 4 + [_i, _i]),
    "cdn_codenet_stage_chain_parts": (_i, [_i64] * 5),
    "cdn_codenet_heads_pointwise_supported": (_i, [_i64, _i64, _i]),
    "cdn_codenet_heads_pointwise_forward": (_i, [_vp, _vp, _i64, _i64, _i] + [_vp] * 5 + [_i] + [_vp] * 3 + [_i, _d, _i]
                                            + [_vp, ctypes.c_size_t, _vp, _vp, _i64, _vp]),
    "cdn_codenet_stage_fused_forward_chain": (_i, [_vp, _i, _i, _vp] + [_i64] * 5 + [_vp, _vp, _f, _f] + [_vp] * 5
                                              + [_i, _vp, ctypes.c_size_t, _vp, _vp, _i, _vp, _vp, _vp]),
    "cdn_codenet_stage_fused_intermediates": (_i, [_i64] * 4 + [_i, _i, _i, _vp, _vp]),
    "cdn_codenet_wcodes_kb_columns": (_i64, [_i64, _i64]),
    "cdn_codenet_pointwise_i8_supported": (_i, [_i64] * 4),
    "cdn_quantact_arrive_words": (_i, []),
    "cdn_codenet_scale_forward_update": (_i, [_vp] * 4 + [_i64] * 4 + [_f, _f] + [_vp] * 4 + [_i, _d, _vp, _vp]),
    "cdn_codenet_dw_forward_update_supported": (_i, [_i64] * 4 + [_i]),
    "cdn_codenet_dw_forward_update": (_i, [_vp] * 4 + [_i64] * 4 + [_i] + [_vp] * 4 + [_i, _d, _vp, _vp]),
    "cdn_codenet_pointwise_i8_forward_update": (_i, [_vp] * 5 + [_i64] * 4 + [_vp, ctypes.c_size_t, _i] + [_vp] * 4
                                                + [_i, _d, _vp]),
    "cdn_quantact_apply": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "cdn_quantact_relu_apply": (_i, [_vp, _vp, _i64, _i64, _i64, _i, _vp, _vp]),
    "cdn_codenet_pointwise_dgrad_q4_supported": (_i, [_i64] * 4),
    "cdn_codenet_pointwise_dgrad_q4": (_i, [_vp] * 3 + [_i64] * 4 + [_vp]),
    "cdn_codenet_pointwise_i8_workspace_bytes": (ctypes.c_size_t, [_i64] * 4),
    "cdn_codenet_pointwise_i8_range_partials": (_i64, [_i64] * 4),
    "cdn_codenet_pointwise_i8_forward_range": (_i, [_vp] * 5 + [_i64] * 4 + [_vp, _vp, ctypes.c_size_t, _vp]),
    "cdn_codenet_wcodes_kb_offset": (_i64, [_i64, _i64]),
    "cdn_codenet_maxpool3x3s2_q8_forward": (_i, [_vp] + [_i64] * 6 + [_vp, _vp]),
    "cdn_codenet_dwpw_q8_supported": (_i, [_i64] * 3 + [_i, _i64]),
    "cdn_codenet_dwpw_q8_forward": (
        _i, [_vp, _vp] + [_i64] * 4 + [_i, _i64, _vp, _vp, _i, _vp, _i64] + [_vp] * 4 + [_i, _i64] + [_vp] * 5),
    "cdn_codenet_dw_backward": (_i, [_vp] * 7 + [_i64] * 4 + [_vp]),
    "cdn_codenet_dw_backward_workspace_bytes": (ctypes.c_size_t, [_i64] * 4 + [_i]),
    "cdn_codenet_dw_backward_r": (_i, [_vp] * 7 + [_i64] * 4 + [_vp, _vp]),
    "cdn_codenet_dw_up2_backward_r": (_i, [_vp] * 7 + [_i64] * 4 + [_vp, _vp]),
    "cdn_codenet_pointwise_forward": (_i, [_vp] * 6 + [_i64] * 4 + [_i, _vp]),
    "cdn_quantact_state_bytes": (ctypes.c_size_t, []),
    "cdn_kth_values_workspace_bytes": (ctypes.c_size_t, []),
    "cdn_kth_values": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "cdn_quantact_forward": (_i, [_vp] * 3 + [_i64] + [_vp] * 5 + [_i, _d, _i, _vp]),
    "cdn_quantact_relu_up2_forward": (_i, [_vp, _vp] + [_i64] * 3 + [_vp] * 3 + [_i, _d, _i, _vp]),
    "cdn_up2_relu_backward": (_i, [_vp] * 3 + [_i64] * 3 + [_vp]),
    "cdn_quantact_relu_forward": (_i, [_vp, _vp, _i64] + [_vp] * 4 + [_i64, _i, _d, _i, _vp]),
    "cdn_relu_backward": (_i, [_vp] * 3 + [_i64, _vp]),
    "cdn_codenet_pwdw_s2_supported": (_i, [_i64] * 5),
    "cdn_codenet_pwdw_s2_forward": (
        _i, [_vp, _vp] + [_i64] * 5 + [_vp] * 8 + [_i64, _vp, _vp, _i64] + [_vp] * 3 + [_i, _d, _i, _vp,
                                                                                    ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_pwdw_s2_apply": (
        _i, [_vp, _vp] + [_i64] * 5 + [_vp] * 8 + [_i64, _vp, _vp, _i64] + [_vp] * 3 + [_i, _d, _i, _vp,
                                                                                    ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_dw_up2_supported": (_i, [_i64] * 4),
    "cdn_codenet_dw_up2_range_partials": (_i64, [_i64] * 4),
    "cdn_codenet_dw_up2_forward": (_i, [_vp] * 4 + [_i64] * 4 + [_vp, _vp]),
    "cdn_codenet_dw_up2_backward": (_i, [_vp] * 7 + [_i64] * 4 + [_vp]),
    "cdn_codenet_stage_workspace_bytes": (ctypes.c_size_t, [_i64] * 4 + [_i]),
    "cdn_codenet_stage_fused_forward": (
        _i, [_vp, _i, _i, _vp] + [_i64] * 5 + [_vp, _vp, _f, _f] + [_vp] * 8 + [_i] + [_vp] * 9
        + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_quantact_commit_range": (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp]),
    "cdn_quantact_frozen_params": (_i, [_i, _vp, _vp, _vp, _i, _vp]),
    "cdn_quantact_frozen_params_clear": (_i, [_i, _vp, _vp, _vp, _i, _vp, ctypes.c_size_t, _vp]),
    "cdn_codenet_stage_frozen_workspace_bytes": (ctypes.c_size_t, [_i64] * 4 + [_i]),
    "cdn_codenet_stage_frozen_forward": (
        _i, [_vp, _i, _i, _vp] + [_i64] * 5 + [_vp, _vp, _f, _f] + [_vp] * 5 + [_i] + [_vp] * 4
        + [ctypes.c_size_t, _vp, _vp, _vp]),
    "cdn_codenet_stage_frozen_chained_forward": (
        _i, [_vp, _i, _i, _vp] + [_i64] * 5 + [_vp, _vp, _f, _f] + [_vp] * 5 + [_i] + [_vp] * 4
        + [ctypes.c_size_t, _vp, _vp] + [_vp] * 4 + [_vp]),
    "cdn_codenet_pointwise_q8_strided_forward": (_i, [_vp, _vp] + [_i64] * 5 + [_vp] * 4 + [_i] + [_vp] * 6),
    "cdn_codenet_stem_q8_forward": (_i, [_vp] + [_i64] * 4 + [_i, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "cdn_codenet_dw3x3_q8_forward": (_i, [_vp, _vp] + [_i64] * 4 + [_i] + [_i64] * 2 + [_vp, _vp, _i] + [_vp] * 4),
    "cdn_codenet_pointwise_q8_forward": (_i, [_vp, _vp] + [_i64] * 3 + [_vp] * 4 + [_i] + [_vp] * 5),
    "cdn_codenet_expand_codes": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "cdn_codenet_unpack_nchw": (_i, [_vp] * 3 + [_i64] * 4 + [_i, _vp]),
    "cdn_codenet_aux_workspace_bytes": (ctypes.c_size_t, []),
    "cdn_codenet_pointwise_nhwc_forward": (
        _i, [_vp, _vp] + [_i64] * 5 + [_vp] * 7 + [_i] + [_vp] * 3 + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_dw3x3_nhwc_forward": (
        _i, [_vp, _vp] + [_i64] * 4 + [_i, _i] + [_i64] * 2 + [_vp] * 4 + [_i] + [_vp] * 3
        + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_pointwise_mixed_forward": (
        _i, [_vp] * 3 + [_i64] * 5 + [_vp] * 7 + [_i] + [_vp] * 4 + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_pointwise_mixed_forward_n": (
        _i, [_vp] * 3 + [_i] + [_i64] * 5 + [_vp] * 7 + [_i] + [_vp] * 4 + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_dw3x3_mixed_forward": (
        _i, [_vp] * 3 + [_i64] * 4 + [_i, _i] + [_i64] * 2 + [_vp] * 4 + [_i] + [_vp] * 3
        + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_codenet_head_range_forward": (
        _i, [_vp, _vp] + [_i64] * 4 + [_vp] * 5 + [_i, _d, _i, _vp, ctypes.c_size_t, _vp]),
    "cdn_codenet_head_tail_small_forward": (
        _i, [_vp, _vp] + [_i64] * 4 + [_vp] * 7 + [_i64, _vp, _vp]),
    "cdn_codenet_head_tail_small_q8_forward": (
        _i, [_vp, _vp] + [_i64] * 4 + [_vp] * 7 + [_i64, _vp, _vp, _vp]),
    "cdn_codenet_interleave_forward": (
        _i, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "cdn_codenet_maxpool3x3s2_nhwc_forward": (_i, [_vp, _vp] + [_i64] * 4 + [_vp, _vp]),
    "cdn_codenet_stem_forward": (
        _i, [_vp] + [_i64] * 4 + [_i] + [_vp, _vp, _i] + [_vp] * 3 + [_i, _d, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "cdn_ctdet_decode_workspace_bytes": (ctypes.c_size_t, [_i64] * 4),
    "cdn_ctdet_flip_merge": (_i, [_vp, _vp] + [_i64] * 5 + [_vp, _vp, _vp]),
    "cdn_ctdet_decode": (_i, [_vp] * 3 + [_i64] * 4 + [_i, _i, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "cdn_profile_enable": (_i, [_i]),
    "cdn_profile_read": (_i, [_i, _vp, _vp, _vp]),
}


def build(force=False):
    """Compile the HIP sources for gfx950 in-tree (codenet_amd/csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return SO_PATH


def use_library(path):
    """A/B tooling only (tools/with_lib.py): load a VARIANT build of the library instead of the product one.
    Must be called explicitly, with a path, before the first lib() -- nothing in the environment can redirect
    the product loader (a `make diag` / `make stamps` build computes wrong or slower results on purpose)."""
    global SO_PATH
    if _lib is not None:
        raise RuntimeError("codenet_amd: use_library() after the library was loaded")
    path = os.path.abspath(path)
    if not os.path.exists(path):
        raise RuntimeError("codenet_amd: variant library %s does not exist" % path)
    SO_PATH = path
    return path


def lib():
    """Load the native library or fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "codenet_amd: native library %s is missing -- run `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is no "
                "fallback path." % SO_PATH)
        # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process, so
        # make sure it is loaded BEFORE our library resolves the same SONAME (otherwise /opt/rocm's
        # copy is pulled in as a second runtime and every launch fails with hipErrorNoDevice).
        import torch  # noqa: F401
        l = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.cdn_abi_version() != 1:
            raise RuntimeError("codenet_amd: ABI version mismatch")
        _lib = l
    return _lib


def last_error():
    return lib().cdn_last_error().decode("utf-8", "replace")


class NativeError(RuntimeError):
    status = 0


CDN_ERR_UNSUPPORTED = -5


def check(rc, what):
    """Map a cdn_status to the exception types the reference raises
    (AT_CHECK/AT_ERROR -> RuntimeError, dcn_deform_conv_cuda.cpp:61-149)."""
    if rc != 0:
        err = NativeError("%s failed (cdn_status %d): %s" % (what, rc, last_error()))
        err.status = rc
        raise err
