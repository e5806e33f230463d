"""ctypes binding of the C ABI declared in ``include/bgs.h`` (libbgs.so).

The product path has NO CPU fallback: if the HIP library has not been built, every
entry point raises ``BgsLibraryError`` — a GPU run can never silently execute
something other than the hand-written kernels.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))

c_i64p = ctypes.c_void_p
c_f32p = ctypes.c_void_p
c_ptr = ctypes.c_void_p

BGS_OK = 0
BGS_MAX_BINS = 16


class BgsLibraryError(RuntimeError):
    pass


class BgsCallError(RuntimeError):
    def __init__(self, fn, code, text):
        super().__init__('%s failed with code %d: %s' % (fn, code, text))
        self.code = code


# name -> (restype, argtypes); mirrors include/bgs.h one-to-one (checked by tests/test_capi.py)
SIGNATURES = {
    'bgs_version': (ctypes.c_int, []),
    'bgs_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'bgs_selftest_wave_reduce': (ctypes.c_int, [c_f32p, c_f32p, c_ptr]),
    'bgs_selftest_mfma_peak': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_selftest_mfma_peak_bf16': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_gs_prepare': (ctypes.c_int, [c_i64p, c_i64p, c_f32p, ctypes.c_int, c_f32p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_uint64, c_ptr, c_i64p, c_f32p, c_f32p, c_ptr]),
    'bgs_gs_loss_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'bgs_gs_loss_fwd_bwd': (ctypes.c_int, [c_f32p, c_ptr, c_i64p, c_f32p, c_f32p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_f32p, c_f32p, c_ptr, c_ptr]),
    'bgs_gs_head_loss_fused': (ctypes.c_int, [c_f32p, c_i64p, c_i64p, c_f32p, c_ptr, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_uint64, c_ptr, c_f32p,
                                              c_f32p, c_f32p, c_ptr, c_f32p, c_ptr, c_ptr]),
    'bgs_gs_head_step': (ctypes.c_int, [c_f32p, c_i64p, c_i64p, c_ptr, c_f32p, c_ptr, c_ptr, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_uint64, c_ptr, c_f32p, c_f32p, c_f32p, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_ptr, c_f32p, c_ptr, c_ptr]),
    'bgs_gs_head_debug_timestamps': (None, [c_ptr]),
    'bgs_gs_head_tuning': (None, [ctypes.c_int]),
    'bgs_gs_head_variant': (None, [ctypes.c_int]),
    'bgs_gs_head_variant_used': (ctypes.c_int, [ctypes.c_int]),
    'bgs_sgd_clip_workspace_bytes': (ctypes.c_size_t, [c_ptr, ctypes.c_int]),
    'bgs_sgd_clip_step': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int] + [ctypes.c_float] * 5
                          + [c_ptr, ctypes.c_size_t, c_ptr, c_ptr]),
    'bgs_gs_class_bin_mask': (ctypes.c_int, [c_i64p, ctypes.c_int, ctypes.c_int, c_ptr, c_ptr]),
    'bgs_gs_head_step_scale_grad': (ctypes.c_int, [c_f32p, c_f32p, c_ptr, c_f32p, c_f32p, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, c_ptr]),
    'bgs_gs_loss_reduce': (ctypes.c_int, [c_ptr, ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_gs_scale_grad': (ctypes.c_int, [c_f32p, c_i64p, c_f32p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, c_ptr]),
    'bgs_gs_merge_score': (ctypes.c_int, [c_f32p, c_i64p, c_ptr, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_bbox_loss_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'bgs_bbox_smooth_l1_fwd_bwd': (ctypes.c_int, [c_f32p, c_i64p, c_f32p, c_f32p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                  ctypes.c_float, c_f32p, c_f32p, c_ptr, c_ptr]),
    'bgs_conv2d_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 11
                            + [c_ptr]),
    'bgs_conv2d_workspace_bytes': (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int]),
    'bgs_conv2d_nhwc_f32_ws': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
                               + [ctypes.c_int] * 11 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_conv2d_dgrad_nhwc_f32_ws': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
                                     + [ctypes.c_int] * 10 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_conv2d_dgrad_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
                                  + [ctypes.c_int] * 10 + [c_ptr]),
    'bgs_conv2d_wgrad_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 9),
    'bgs_conv2d_wgrad_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p]
                                  + [ctypes.c_int] * 10 + [c_ptr, c_ptr]),
    'bgs_fold_conv_bn_fwd': (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_float] + [ctypes.c_int] * 5
                             + [c_f32p, c_f32p, c_ptr]),
    'bgs_fold_conv_bn_bwd': (ctypes.c_int, [c_f32p] * 7 + [ctypes.c_float] + [ctypes.c_int] * 5
                             + [c_f32p] * 4 + [c_ptr]),
    'bgs_conv2d_wgrad_bfx_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 9),
    'bgs_conv2d_wgrad_nhwc_f32_bfx': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p]
                                      + [ctypes.c_int] * 11 + [c_ptr, c_ptr]),
    'bgs_conv2d_wgrad_bfx_enable': (None, [ctypes.c_int]),
    'bgs_conv3x3_halo_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 6
                                  + [c_ptr]),
    'bgs_conv_bfx_weight_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'bgs_conv_bfx_split_weights': (ctypes.c_int, [c_f32p, c_ptr, ctypes.c_int, ctypes.c_int, c_ptr]),
    'bgs_conv_dgrad_parity_enable': (None, [ctypes.c_int]),
    'bgs_conv_bfx_split_weights_dgrad': (ctypes.c_int, [c_f32p, c_ptr] + [ctypes.c_int] * 4 + [c_ptr]),
    'bgs_conv_bfx_workspace_bytes': (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]),
    'bgs_conv2d_nhwc_f32_bfx_ws': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_f32p, c_f32p]
                                   + [ctypes.c_int] * 12 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_conv2d_dgrad_nhwc_f32_bfx_ws': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_f32p, c_f32p]
                                         + [ctypes.c_int] * 11 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_conv3x3_halo_bfx_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 5),
    'bgs_conv3x3_c3_fused_nhwc_f32_bfx': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_ptr, c_f32p, c_f32p, c_f32p]
                                          + [ctypes.c_int] * 6 + [c_ptr]),
    'bgs_conv3x3_halo_nhwc_f32_bfx': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_f32p]
                                      + [ctypes.c_int] * 7 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_conv3x3_halo_nhwc_f32_bfx_ex': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_f32p, c_f32p]
                                         + [ctypes.c_int] * 7 + [c_ptr, ctypes.c_size_t, c_ptr]),
    'bgs_gs_loss_tuning': (None, [ctypes.c_int]),
    'bgs_gs_head_fold': (None, [ctypes.c_int]),
    'bgs_gs_loss_wavepriv_min_rows': (None, [ctypes.c_int]),
    'bgs_gs_merge_tuning': (None, [ctypes.c_int, ctypes.c_int]),
    'bgs_conv_bfx_wide_tuning': (None, [ctypes.c_int] * 3),
    'bgs_conv_bfx_wide_last_launch': (ctypes.c_int, []),
    'bgs_conv1x1_planes_enable': (None, [ctypes.c_int]),
    'bgs_conv1x1_planes_last_launch': (ctypes.c_int, []),
    'bgs_conv3x3_planes_enable': (None, [ctypes.c_int]),
    'bgs_conv3x3_planes_last_launch': (ctypes.c_int, []),
    'bgs_conv1x1_bres_enable': (None, [ctypes.c_int]),
    'bgs_conv1x1_bres_last_launch': (ctypes.c_int, []),
    'bgs_launch_census': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'bgs_conv_bfx_tuning': (None, [ctypes.c_int] * 2),
    'bgs_conv_bfx_last_launch': (ctypes.c_int, [c_ptr, c_ptr]),
    'bgs_conv_tuning': (None, [ctypes.c_int] * 4),
    'bgs_conv_last_launch': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    'bgs_conv3x3_halo_bfx_tuning': (None, [ctypes.c_int] * 2),
    'bgs_conv3x3_halo_bfx_last_launch': (ctypes.c_int, [c_ptr, c_ptr]),
    'bgs_conv3x3_halo_bfx_last_wide': (ctypes.c_int, [c_ptr, c_ptr]),
    'bgs_conv3x3_halo_bfx_wide': (ctypes.c_int, [ctypes.c_int]),
    'bgs_grouped_conv3x3_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 7
                                     + [c_ptr]),
    'bgs_grouped_conv3x3_nhwc_bf16ops': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 7
                                         + [c_ptr]),
    'bgs_grouped_conv3x3_dgrad_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 6
                                           + [c_ptr]),
    'bgs_grouped_conv3x3_wgrad_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 6),
    'bgs_grouped_conv3x3_wgrad_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p]
                                           + [ctypes.c_int] * 7 + [c_ptr, c_ptr]),
    'bgs_maxpool3x3s2_bwd_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p, c_f32p] + [ctypes.c_int] * 4 + [c_ptr]),
    'bgs_maxpool3x3s2_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 4 + [c_ptr]),
    'bgs_nchw_to_nhwc4_f32': (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 4 + [c_ptr]),
    'bgs_conv2d_nhwc_bf16s': (ctypes.c_int, [c_ptr, c_ptr, c_f32p, c_ptr, ctypes.c_int, ctypes.c_int, c_ptr]
                              + [ctypes.c_int] * 11 + [c_ptr]),
    'bgs_grouped_conv3x3_nhwc_bf16s': (ctypes.c_int, [c_ptr, c_f32p, c_f32p, c_ptr] + [ctypes.c_int] * 7
                                       + [c_ptr]),
    'bgs_maxpool3x3s2_nhwc_f32_to_bf16': (ctypes.c_int, [c_f32p, c_ptr] + [ctypes.c_int] * 4 + [c_ptr]),
    'bgs_conv_bf16s_tuning': (None, [ctypes.c_int]),
    'bgs_roi_align_nhwc_fwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, c_f32p, c_ptr, c_ptr]),
    'bgs_roi_align_nhwc_bwd': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, c_f32p, c_ptr]),
    'bgs_roi_align_nhwc_fwd_f16': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, c_ptr, c_ptr, c_ptr]),
    'bgs_roi_align_nhwc_bwd_f16': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, c_ptr, c_ptr]),
    'bgs_roi_align_nhwc_fwd_ex': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
                                                 c_ptr, c_ptr]),
    'bgs_roi_align_nhwc_bwd_ex': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_resize_bilinear_nhwc_f32': (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 7 + [c_ptr]),
    'bgs_resize_bilinear_nhwc_bwd_f32': (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 7
                                         + [c_ptr]),
    'bgs_mask_target': (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
                                       ctypes.c_int, c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, c_f32p,
                                       c_ptr]),
    'bgs_stem_fused_weight_bytes': (ctypes.c_size_t, []),
    'bgs_stem_fused_split_weights': (ctypes.c_int, [c_f32p, ctypes.c_int, c_ptr, c_ptr]),
    'bgs_stem_conv7x7s2_relu_maxpool_nchw_f32': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, c_f32p, ctypes.c_int,
                                                                ctypes.c_int, ctypes.c_int, c_ptr]),
    'bgs_mask_paste_u8': (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_int, ctypes.c_int, c_ptr, c_ptr]),
    'bgs_mask_gt_logits': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_ptr, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_mask_bce_partials': (ctypes.c_int, [ctypes.c_int]),
    'bgs_mask_bce': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_ptr, c_f32p, c_ptr, c_f32p,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
                                    c_f32p, c_f32p, c_f32p, c_ptr]),
    'bgs_topk_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'bgs_topk_sorted_f32': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int,
                                           ctypes.c_int, c_f32p, c_ptr, c_ptr, c_ptr]),
    'bgs_nms_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'bgs_nms_batched': (ctypes.c_int, [c_f32p, c_ptr, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_int, ctypes.c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    'bgs_nms_merge_select': (ctypes.c_int, [c_f32p, c_ptr, c_ptr] + [ctypes.c_int] * 4 + [c_f32p, c_ptr, c_ptr]),
    'bgs_nms_gather': (ctypes.c_int, [c_f32p, c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_ptr]),
    'bgs_gather_boxes': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
                                        c_ptr, c_ptr]),
    'bgs_iou_assign_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'bgs_iou_assign': (ctypes.c_int, [c_f32p, ctypes.c_longlong, ctypes.c_int, c_ptr, c_f32p, c_ptr,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, c_ptr, c_f32p, c_ptr, c_ptr]),
    'bgs_rpn_loss_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'bgs_rpn_loss': (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, c_f32p, c_ptr, c_ptr,
                                    c_ptr, c_f32p, c_ptr, ctypes.c_int, c_ptr, c_ptr,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                    c_f32p, c_f32p, c_f32p, c_ptr, c_ptr]),
    'bgs_rpn_loss_grad': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, c_f32p,
                                         c_ptr, c_ptr, c_ptr, c_f32p, c_ptr, ctypes.c_int, c_ptr,
                                         c_ptr, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, c_f32p, c_f32p, c_f32p, c_ptr]),
    'bgs_sample_pos_neg_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'bgs_sample_pos_neg': (ctypes.c_int, [c_ptr, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_uint64, c_ptr, c_ptr, c_ptr, c_ptr,
                                          c_ptr]),
    'bgs_sample_rois': (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_uint64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'bgs_sample_rois_ex': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_uint64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'bgs_random_keys': (ctypes.c_int, [ctypes.c_uint64, c_ptr, ctypes.c_int, c_ptr, c_ptr]),
    'bgs_decode_proposals': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int, ctypes.c_int, c_f32p,
                                            c_ptr, c_f32p, ctypes.c_int, c_ptr, c_ptr, c_ptr,
                                            ctypes.c_float, ctypes.c_int, c_f32p, c_ptr]),
    'bgs_refine_boxes': (ctypes.c_int, [c_f32p, c_ptr, c_f32p, ctypes.c_int, ctypes.c_int, c_ptr,
                                        ctypes.c_int, c_f32p, c_f32p, ctypes.c_float, c_f32p, c_ptr]),
    'bgs_rcnn_targets': (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32p, c_ptr,
                                        ctypes.c_int, ctypes.c_int, c_ptr, c_ptr, ctypes.c_float,
                                        c_f32p, c_ptr, c_f32p, c_f32p, c_f32p, c_ptr]),
}

_LIB = None


def lib_filename():
    """``BGS_LIB_VARIANT=nodpp`` selects the ds_bpermute build (A/B and safety net)."""
    variant = os.environ.get('BGS_LIB_VARIANT', '')
    return 'libbgs_%s.so' % variant if variant else 'libbgs.so'


def lib_path():
    return os.environ.get('BGS_LIB_PATH', os.path.join(_PKG, lib_filename()))


def load():
    """dlopen libbgs.so (once) and attach the prototypes."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise BgsLibraryError(
            '%s not found. Build the HIP kernels first: '
            '`python -m balancedgroupsoftmax_amd.csrc.build` (or __graft_entry__.build()). '
            'There is deliberately no CPU fallback.' % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise BgsLibraryError('cannot load %s: %s' % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise BgsLibraryError('%s does not export %s (stale build?)' % (path, name))
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(fn_name, code):
    if code != BGS_OK:
        raise BgsCallError(fn_name, code, load().bgs_error_string(code).decode())


def host_i64(a):
    """Host int64 array (numpy, contiguous) -> pointer for ``host_*`` parameters."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def ptr(t):
    """Device (or host) address of a tensor as a plain integer, None (= NULL) for None: every pointer parameter of the
    library is declared ``c_void_p``, which converts both (no ``c_void_p`` object per argument on the launch path)."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = []


def raw_stream(device=None):
    """The current HIP stream of ``device`` as an integer handle (``torch._C._cuda_getCurrentRawStream``: no Stream
    object, no device-index normalisation — the Python-level ``torch.cuda.current_stream`` cost ~2 us per launch, one
    tenth of the launching thread's time in an eagerly launched step)."""
    import torch
    if not _RAW_STREAM:
        _RAW_STREAM.append(getattr(torch._C, '_cuda_getCurrentRawStream', None))
    idx = device if isinstance(device, int) else getattr(device, 'index', None)
    if idx is None:
        if isinstance(device, str):
            idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
    fn = _RAW_STREAM[0]
    if fn is None:       # (an older torch: the public API)
        return torch.cuda.current_stream(idx).cuda_stream
    return fn(idx)


def current_stream(device=None):
    """The ``bgs_stream_t`` argument of a launch on ``device``'s current stream (an integer handle, see :func:`ptr`)."""
    return raw_stream(device)
