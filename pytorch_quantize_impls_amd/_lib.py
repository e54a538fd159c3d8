"""ctypes binding of libqt_hip.so (the C-ABI declared in include/qt_hip.h).

The shared library is built IN-TREE (pytorch_quantize_impls_amd/lib/libqt_hip.so) by
``__graft_entry__.build()`` / ``make -C pytorch_quantize_impls_amd/csrc``.  There is no fallback:
if the library is missing or a symbol is absent, every GPU op raises ``QtLibraryError``.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading
from collections import Counter

_HERE = os.path.dirname(os.path.abspath(__file__))
# QT_HIP_LIB: A/B builds of the same ABI (tools/); the product path is the in-tree library
LIB_PATH = os.environ.get("QT_HIP_LIB") or os.path.join(_HERE, "lib", "libqt_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "qt_hip.h"))


class QtLibraryError(RuntimeError):
    """libqt_hip.so could not be loaded (not built, wrong arch, missing symbol)."""


class QtStatusError(RuntimeError):
    """A C-ABI entry point returned a negative qt_status."""


_c_i64 = ctypes.c_int64
_c_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_f32 = ctypes.c_float

# name -> (restype, argtypes).  Must list every function include/qt_hip.h declares
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "qt_version": (_c_int, []),
    "qt_strerror": (ctypes.c_char_p, [_c_int]),
    "qt_target_arch": (ctypes.c_char_p, []),
    "qt_device_info": (_c_int, [ctypes.c_char_p, _c_int]),
    "qt_binarize_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "qt_binarize_stochastic_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p]),
    "qt_ternarize_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "qt_ternarize_stochastic_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p]),
    "qt_ste_mask_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_f32, _c_p]),
    "qt_poison_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_i64, _c_p]),
    "qt_dorefa_quantize_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_p]),
    "qt_xnor_weight_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_shift_batch_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_f32, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64,
                                    _c_i64, _c_p]),
    "qt_xnor_act_work_floats": (_c_i64, []),
    "qt_xnor_act_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_xnor_act_backward_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_int,
                                          _c_p]),
    "qt_code_conv3x3_launch_count": (_c_i64, []),
    "qt_xnor_input_quant_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_conv2d_implicit_taps_rows": (_c_int, [_c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    "qt_sign_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_ternary_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_check_pm1_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p]),
    "qt_lin_quantize_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_int, _c_int, _c_p]),
    "qt_log_quantize_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_int, _c_int, _c_p]),
    "qt_ap2_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "qt_xnor_gemm_variant": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_tern_gemm_variant": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64,
                                      _c_p]),
    "qt_conv2d_implicit_variant": (_c_int, [_c_int, _c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p,
                                                                                     _c_i64, _c_i64, _c_p]),
    "qt_xnor_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64,
                              _c_i64, _c_p]),
    "qt_tern_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64,
                              _c_i64, _c_i64, _c_p]),
    "qt_sign_pack_nib_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_ternary_pack_nib_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_sign0_pack_nib_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_pack_pair_nib_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64,
                                      _c_int, _c_p]),
    "qt_wgrad_pack_grad_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p,
                                        _c_i64, _c_p]),
    "qt_wgrad_pack_act_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                                       _c_i64, _c_i64, _c_f32, _c_p, _c_i64, _c_i64, _c_p]),
    "qt_bf16_gemm_taps": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                                   _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_wgrad_reduce_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_f32, _c_f32, _c_int, _c_p,
                                     _c_p]),
    "qt_wgrad_pm_pack_grad_f32": (_c_int, [_c_p] + [_c_i64] * 11 + [_c_p, _c_p]),
    "qt_wgrad_pm_pack_grad_bias_f32": (_c_int, [_c_p] + [_c_i64] * 10 + [_c_p, _c_p, _c_p]),
    "qt_wgrad_pm_bias_reduce_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_p]),
    "qt_wgrad_pm_pack_act_f32": (_c_int, [_c_p] + [_c_i64] * 13 + [_c_f32, _c_p, _c_p]),
    "qt_wgrad_pm_pack_act_s2d_f32": (_c_int, [_c_p] + [_c_i64] * 17 + [_c_p, _c_p]),
    "qt_wgrad_pm_pack_act_s2d_f16x2": (_c_int, [_c_p] + [_c_i64] * 17 + [_c_p, _c_p, _c_p]),
    "qt_wgrad_pm_pack_grad_f16x2": (_c_int, [_c_p] + [_c_i64] * 11 + [_c_p, _c_p, _c_p, _c_p]),
    "qt_wgrad_pm_pack_act_f16": (_c_int, [_c_p] + [_c_i64] * 13 + [_c_f32, _c_p, _c_p]),
    "qt_wgrad_pm_f16": (_c_int, [_c_p, _c_p, _c_p] + [_c_i64] * 7 + [_c_p]),
    "qt_wgrad_pm_f32": (_c_int, [_c_p, _c_p, _c_p] + [_c_i64] * 7 + [_c_p]),
    "qt_wgrad_pm_reduce_f32": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_f32, _c_f32, _c_p, _c_int, _c_p] + [_c_i64] * 4 + [_c_p]),
    "qt_nib_gemm_describe": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, ctypes.c_char_p, _c_int]),
    "qt_nib_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64,
                             _c_i64, _c_p]),
    "qt_bits_to_nib": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_pack_conv_weight_codes_i8": (_c_int, [_c_p] + [_c_i64] * 8 + [_c_int, _c_p, _c_i64, _c_p]),
    "qt_dorefa_codes_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_p]),
    "qt_bn_eval_device_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_affine_dorefa_codes_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_f32,
                                           _c_int, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_p]),
    "qt_affine_dorefa_codes_halo_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_f32,
                                                _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_p, _c_p,
                                                _c_i64, _c_i64, _c_p]),
    "qt_codes_to_f32": (_c_int, [_c_p] + [_c_i64] * 7 + [_c_f32, _c_p, _c_i64, _c_p, _c_i64, _c_p]),
    "qt_weight_codes_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_i8_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_f32, _c_p, _c_i64, _c_p, _c_i64, _c_i64,
                            _c_i64, _c_i64, _c_p]),
    "qt_pool_affine_sign_pack_nhwc": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p,
                                               _c_p, _c_i64, _c_int, _c_p]),
    "qt_pool_affine_sign_pack_nib_nhwc": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p,
                                                   _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p]),
    "qt_bf16x3_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_bf16x3_s2d_pack_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_i64] + [_c_i64] * 7 + [_c_p]),
    "qt_bf16x6_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_f16x2_scale_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p]),
    "qt_f16x2_absmax_work_words": (_c_i64, []),
    "qt_f16x2_absmax_scale_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p]),
    "qt_conv3x3_first_pack_weight_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
    "qt_conv3x3_first_f32": (_c_int, [_c_p] + [_c_i64] * 8 + [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "qt_code_digits_f32": (_c_int, [_c_p, _c_i64, _c_f32, _c_p, _c_p, _c_p, _c_int, _c_p]),
    "qt_digit_combine_f32": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_f32, _c_p, _c_i64, _c_p]),
    "qt_abs_mean_work_words": (_c_i64, []),
    "qt_abs_mean_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p]),
    "qt_f16x2_absmax_pack_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "qt_f16x2_absmax_ch_work_words": (_c_i64, [_c_i64]),
    "qt_f16x2_absmax_scale_ch_f32": (_c_int, [_c_p] + [_c_i64] * 9 + [_c_p, _c_p, _c_p]),
    "qt_f16x2_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_f16x2_pack_conv_weight_f32": (_c_int, [_c_p] + [_c_i64] * 8 + [_c_int, _c_int, _c_p, _c_i64, _c_p]),
    "qt_f16x2_s2d_pack_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_i64] + [_c_i64] * 7 + [_c_p]),
    "qt_f16x2_s2d_spec_work_words": (_c_i64, [_c_i64] * 4),
    "qt_f16x2_s2d_pack_spec_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_f32, _c_p, _c_p, _c_p, _c_p, _c_i64] + [_c_i64] * 7
                                   + [_c_p]),
    "qt_f16_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_train_chain_partial_floats": (_c_i64, [_c_i64, _c_i64]),
    "qt_bn_train_stats_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_f32, _c_f32, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "qt_bn_act_train_backward_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_int] + [_c_p] * 6),
    "qt_pool_bn_sign_train_f32": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_p, _c_f32, _c_f32, _c_f32, _c_f32] + [_c_p] * 9),
    "qt_pool_bn_sign_train_backward_f32": (_c_int, [_c_p, _c_p, _c_p] + [_c_i64] * 6 + [_c_p] * 4 + [_c_f32] * 3 + [_c_p] * 6),
    "qt_bf16_gemm": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_conv2d_implicit": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_i64,
                                                                    _c_i64, _c_p]),
    "qt_conv2d_implicit_bits": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p,
                                                                         _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    "qt_conv2d_implicit_codes": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p,
                                                                          _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_f32,
                                                                          _c_int, _c_int, _c_p, _c_i64, _c_i64, _c_p]
                                 + [_c_i64] * 6 + [_c_p, _c_p, _c_p]),
    "qt_conv2d_implicit_halo": (_c_int, [_c_int, _c_p] + [_c_i64] * 14 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_i64,
                                                                          _c_i64, _c_p]),
    "qt_conv2d_implicit_halo_bn": (_c_int, [_c_int, _c_p] + [_c_i64] * 14 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p, _c_p,
                                                                             _c_p, _c_i64, _c_i64, _c_p]),
    "qt_conv2d_implicit_nib": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p,
                                                                        _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_conv3x3_direct_nib": (_c_int, [_c_int, _c_p] + [_c_i64] * 4 + [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64,
                                        _c_int, _c_p]),
    "qt_conv3x3_direct_codes": (_c_int, [_c_p] + [_c_i64] * 4 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                          _c_f32, _c_int, _c_int, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "qt_pool_bits_nib": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_p] + [_c_i64] * 4 + [_c_p]),
    "qt_pool_bits": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_p, _c_p]),
    "qt_pool_codes_i8": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_i64, _c_i64, _c_p]),
    "qt_zero_halo": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p]),
    "qt_pad_pixel_plane": (_c_int, [_c_p] + [_c_i64] * 6 + [_c_p, _c_p]),
    "qt_bits_to_nib_pad": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_i64] + [_c_i64] * 6 + [_c_p]),
    "qt_im2col_words": (_c_int, [_c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_conv_first_direct_f32": (_c_int, [_c_p] + [_c_i64] * 14 + [_c_p, _c_p, _c_f32, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p]),
    "qt_conv_first_direct_bits_f32": (_c_int, [_c_p] + [_c_i64] * 14 + [_c_p, _c_p, _c_f32, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p,
                                               _c_i64, _c_p]),
    "qt_conv3x3_direct_pairs": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p]),
    "qt_bits_alpha_pairs_f16x2": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_bits_alpha_digits_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_xnor_head_i8": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_digit_reduce_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_i8_gemm_splitk": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_xnor_tap_prep_work_floats": (_c_i64, [_c_i64, _c_i64]),
    "qt_xnor_tap_prep_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "qt_conv2d_implicit_taps": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p, _c_i64,
                                                                         _c_i64, _c_p]),
    "qt_conv2d_implicit_taps_bits": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p, _c_p,
                                                                              _c_p, _c_i64, _c_i64, _c_p]),
    "qt_conv2d_implicit_taps_nib": (_c_int, [_c_int, _c_p] + [_c_i64] * 12 + [_c_p, _c_i64, _c_p, _c_f32, _c_p, _c_p, _c_p, _c_p,
                                                                             _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "qt_nib_gemm_variant": (_c_int, [_c_int, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64,
                                     _c_i64, _c_i64, _c_p]),
}

_lock = threading.Lock()
_lib = None
#: number of successful calls per entry point in this process — the GPU tests assert on it so a
#: silent non-HIP path cannot pass them.
call_counts: Counter = Counter()


def header_declared_functions(header_path: str = HEADER_PATH):
    """Names of all functions declared in include/qt_hip.h (comments stripped)."""
    with open(header_path, "r", encoding="utf-8") as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(qt_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes handle.  Raises QtLibraryError loudly on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise QtLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "pytorch_quantize_impls_amd/csrc`). There is no CPU/PyTorch fallback for GPU tensors.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the box
            raise QtLibraryError(f"cannot dlopen {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise QtLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def strerror(code: int) -> str:
    return load().qt_strerror(int(code)).decode()


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point and raise on a negative status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise QtStatusError(f"{name} failed: {strerror(rc)} (qt_status {rc})")
    call_counts[name] += 1


def version() -> int:
    return int(load().qt_version())


def target_arch() -> str:
    return load().qt_target_arch().decode()


def device_info():
    buf = ctypes.create_string_buffer(64)
    cus = load().qt_device_info(buf, 64)
    if cus < 0:
        raise QtStatusError(f"qt_device_info failed: {strerror(cus)}")
    return buf.value.decode(), int(cus)
