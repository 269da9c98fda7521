"""ctypes binding of ``libyolo_hip.so`` (C ABI declared in ``include/yolo_hip.h``).

There is deliberately no fallback: if the shared library is missing or an entry point cannot be
resolved, importing the HIP path raises ``RuntimeError`` — GPU results never come from a silent eager
substitute.  Build the library with ``make -C csrc`` (or ``python __graft_entry__.py``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_HIP_LIB: another build of the same library (profiling builds of csrc/Makefile); never a different implementation
LIB_PATH = os.environ.get('YOLO_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'libyolo_hip.so')

YH_F16, YH_F32, YH_I8 = 0, 1, 2
ACT_CODES = {'linear': 0, 'leaky': 1, 'relu': 2, 'relu6': 3, 'h_swish': 4, 'mish': 5}
OP_CONV, OP_STEM, OP_POOL, OP_COPY, OP_ADD, OP_DECODE, OP_DW, OP_SE, OP_QCOPY, OP_QPOOL, OP_QADD = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11

_i32, _f32, _vp, _i64 = C.c_int32, C.c_float, C.c_void_p, C.c_int64


class ConvDesc(C.Structure):
    _fields_ = [('x', _vp), ('w', _vp), ('bias', _vp), ('res', _vp), ('y', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('cin', _i32),
                ('ho', _i32), ('wo', _i32), ('cout', _i32),
                ('kh', _i32), ('kw', _i32), ('stride', _i32), ('pad', _i32),
                ('ldx', _i32), ('ldr', _i32), ('ldy', _i32),
                ('cin_k', _i32), ('m_pad', _i32),
                ('act', _i32), ('slope', _f32), ('ups', _i32), ('out_f32', _i32), ('dtype', _i32), ('tile', _i32),
                ('acc_scale', _f32), ('out_scale', _f32),
                ('y_h', _i32), ('y_w', _i32), ('y_off_h', _i32), ('y_off_w', _i32), ('stats_ws', _vp), ('stats_ws_floats', _i64),
                ('q_rx', _f32), ('q_ra', _f32), ('q_scale_x', _f32), ('q_scale_a', _f32), ('q_inv_scale_sum', _f32),
                # training backward (ABI 2): the block whose gradient this data gradient completes - its backward sums ride along
                ('bwd_z', _vp), ('bwd_gamma', _vp), ('bwd_beta', _vp), ('bwd_mean', _vp), ('bwd_invstd', _vp),
                ('bwd_ldz', _i32), ('bwd_act', _i32), ('bwd_slope', _f32), ('bwd_reserved', _i32)]


class StemDesc(C.Structure):
    _fields_ = [('x', _vp), ('w', _vp), ('bias', _vp), ('y', _vp),
                ('n', _i32), ('cin', _i32), ('h', _i32), ('w_in', _i32), ('ho', _i32), ('wo', _i32),
                ('cout', _i32), ('cout_pad', _i32), ('kh', _i32), ('kw', _i32), ('stride', _i32), ('pad', _i32),
                ('ldy', _i32), ('act', _i32), ('slope', _f32), ('dtype', _i32), ('out_scale', _f32),
                ('stats_ws', _vp), ('stats_ws_floats', _i64)]


class PoolDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ho', _i32), ('wo', _i32), ('k', _i32),
                ('stride', _i32), ('pad_lo', _i32), ('edge_zero', _i32), ('ldx', _i32), ('ldy', _i32), ('dtype', _i32)]


class CopyDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ups', _i32), ('ldx', _i32), ('ldy', _i32),
                ('dtype', _i32)]


class AddDesc(C.Structure):
    _fields_ = [('a', _vp), ('b', _vp), ('y', _vp), ('pixels', _i64),
                ('c', _i32), ('lda', _i32), ('ldb', _i32), ('ldy', _i32), ('dtype', _i32), ('amap', _vp), ('bmap', _vp)]


class DecodeDesc(C.Structure):
    _fields_ = [('p', _vp), ('io', _vp), ('raw', _vp),
                ('n', _i32), ('ny', _i32), ('nx', _i32), ('na', _i32), ('no', _i32), ('ldp', _i32),
                ('rows_total', _i32), ('row_off', _i32), ('stride', _f32),
                ('anchor_w', _f32 * 8), ('anchor_h', _f32 * 8)]


class DwDesc(C.Structure):
    _fields_ = [('x', _vp), ('w', _vp), ('bias', _vp), ('y', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ho', _i32), ('wo', _i32), ('k', _i32),
                ('stride', _i32), ('pad', _i32), ('ldx', _i32), ('ldy', _i32), ('act', _i32), ('slope', _f32),
                ('dtype', _i32)]


class SeDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp), ('w1', _vp), ('w2', _vp), ('pooled', _vp), ('gate', _vp), ('ch_map', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('c_phys', _i32), ('cr', _i32), ('ldx', _i32),
                ('ldy', _i32), ('dtype', _i32)]


class QCopyDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ups', _i32), ('ldx', _i32), ('ldy', _i32),
                ('ratio', _f32)]


class QAddDesc(C.Structure):
    _fields_ = [('x', _vp), ('a', _vp), ('y', _vp), ('pixels', _i64),
                ('c', _i32), ('ldx', _i32), ('lda', _i32), ('ldy', _i32),
                ('rx', _f32), ('ra', _f32), ('scale_x', _f32), ('scale_a', _f32), ('inv_scale_sum', _f32)]


class QPoolDesc(PoolDesc):
    """Same layout as PoolDesc; a distinct Python type so plans record it as YH_OP_QPOOL."""


class BnDesc(C.Structure):
    _fields_ = [('z', _vp), ('dy', _vp), ('res', _vp), ('out', _vp), ('gamma', _vp), ('beta', _vp), ('mean', _vp),
                ('invstd', _vp), ('sum', _vp), ('sumsq', _vp), ('running_mean', _vp), ('running_var', _vp),
                ('pixels', _i64), ('n', _i32), ('h', _i32), ('w_in', _i32),
                ('c', _i32), ('ldz', _i32), ('lddy', _i32), ('ldr', _i32), ('ldo', _i32), ('act', _i32), ('ups', _i32),
                ('dtype', _i32), ('slope', _f32), ('eps', _f32), ('momentum', _f32), ('nparts', _i32), ('ws', _vp), ('ws_floats', _i64)]


class BnStatsDesc(BnDesc):
    """Same layout as BnDesc; the distinct Python types select the plan op kind."""


class BnFinalizeDesc(BnDesc):
    pass


class BnActFwdDesc(BnDesc):
    pass


class BnBwdReduceDesc(BnDesc):
    pass


class BnBwdApplyDesc(BnDesc):
    pass


class WgradDesc(C.Structure):
    _fields_ = [('x', _vp), ('dz', _vp), ('dw', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('cin', _i32), ('ho', _i32), ('wo', _i32), ('cout', _i32),
                ('kh', _i32), ('kw', _i32), ('stride', _i32), ('pad', _i32), ('ldx', _i32), ('lddz', _i32), ('dtype', _i32),
                ('splits', _i32), ('cin_w', _i32), ('ws', _vp), ('ws_floats', _i64)]


class StemWgradDesc(WgradDesc):
    pass


class StemBwdDesc(C.Structure):
    _fields_ = [('x', _vp), ('dy', _vp), ('z', _vp), ('gamma', _vp), ('beta', _vp), ('mean', _vp), ('invstd', _vp),
                ('dgamma', _vp), ('dbeta', _vp), ('dw', _vp), ('ws', _vp), ('ws_floats', _i64),
                ('n', _i32), ('cin', _i32), ('h', _i32), ('w_in', _i32), ('cout', _i32), ('lddy', _i32), ('ldz', _i32), ('act', _i32),
                ('slope', _f32), ('dz1', _vp), ('w1', _vp), ('h1', _i32), ('w1_in', _i32), ('k1', _i32), ('k1_pad', _i32), ('lddz1', _i32)]


class ResampleDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp), ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('big_h', _i32),
                ('big_w', _i32), ('ldx', _i32), ('ldy', _i32), ('dtype', _i32)]


class DilateDesc(ResampleDesc):
    pass


class UpsampleBwdDesc(ResampleDesc):
    pass


class LetterboxDesc(C.Structure):
    _fields_ = [('src', _vp), ('tmp', _vp), ('dst', _vp), ('hbounds', _vp), ('hk', _vp), ('vbounds', _vp), ('vk', _vp),
                ('h0', _i32), ('w0', _i32), ('c', _i32), ('src_pitch', _i32), ('hksize', _i32), ('vksize', _i32),
                ('new_h', _i32), ('new_w', _i32), ('out_h', _i32), ('out_w', _i32), ('top', _i32), ('left', _i32),
                ('pad_value', _i32), ('swap_rb', _i32), ('scale', _f32), ('shift', _f32), ('arith', _i32), ('out_u8', _i32)]


class MosaicDesc(C.Structure):
    _fields_ = [('src', _vp * 4), ('dst', _vp), ('inv', C.c_double * 6), ('hsv_gain', C.c_double * 3),
                ('src_h', _i32 * 4), ('src_w', _i32 * 4), ('src_pitch', _i32 * 4),
                ('x1a', _i32 * 4), ('y1a', _i32 * 4), ('x2a', _i32 * 4), ('y2a', _i32 * 4), ('x1b', _i32 * 4), ('y1b', _i32 * 4),
                ('canvas_h', _i32), ('canvas_w', _i32), ('out_h', _i32), ('out_w', _i32), ('c', _i32), ('pad_value', _i32),
                ('hsv', _i32), ('flip_lr', _i32), ('out_dtype', _i32), ('divisor', _f32), ('arith', _i32), ('lut', _vp)]


class LayoutDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp), ('n', _i32), ('c', _i32), ('h', _i32), ('w_in', _i32), ('c_pad', _i32), ('ldy', _i32),
                ('dtype', _i32)]


class PoolBwdDesc(C.Structure):
    _fields_ = [('x', _vp), ('dy', _vp), ('dx', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ho', _i32), ('wo', _i32), ('k', _i32),
                ('stride', _i32), ('pad_lo', _i32), ('edge_zero', _i32), ('ldx', _i32), ('lddy', _i32), ('lddx', _i32),
                ('dtype', _i32)]


class PackItem(C.Structure):
    _fields_ = [('w', _vp), ('bias', _vp), ('packed', _vp), ('bias_out', _vp),
                ('mode', _i32), ('dtype', _i32), ('cout', _i32), ('cin', _i32), ('kh', _i32), ('kw', _i32), ('k_pad', _i32),
                ('m_pad', _i32), ('pad', _i32), ('pa', _i32), ('pb', _i32), ('cout_pad', _i32)]


class PackBatchDesc(C.Structure):
    _fields_ = [('items', _vp), ('n_items', _i32)]


class DwBwdDesc(C.Structure):
    _fields_ = [('x', _vp), ('dz', _vp), ('w', _vp), ('dx', _vp), ('dw', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('ho', _i32), ('wo', _i32), ('k', _i32), ('stride', _i32),
                ('pad', _i32), ('ldx', _i32), ('lddz', _i32), ('lddx', _i32), ('accumulate', _i32), ('dtype', _i32),
                ('ws', _vp), ('ws_floats', _i64)]


class DwWgradDesc(DwBwdDesc):
    pass


class DwDgradDesc(DwBwdDesc):
    pass


class SeBwdDesc(C.Structure):
    _fields_ = [('x', _vp), ('dy', _vp), ('dx', _vp), ('w1', _vp), ('w2', _vp), ('pooled', _vp), ('gate', _vp), ('dw1', _vp),
                ('dw2', _vp), ('scratch', _vp),
                ('n', _i32), ('h', _i32), ('w_in', _i32), ('c', _i32), ('cr', _i32), ('ldx', _i32), ('lddy', _i32), ('lddx', _i32),
                ('accumulate', _i32), ('dtype', _i32), ('scratch2', _vp), ('scratch2_floats', _i64)]


class LossDesc(C.Structure):
    _fields_ = [('p', _vp), ('grad', _vp), ('tobj', _vp), ('winner', _vp), ('targets', _vp), ('anchors', _vp), ('sums', _vp),
                ('count', _vp), ('scale', _vp),
                ('sb', _i64), ('sa', _i64), ('sy', _i64), ('sx', _i64), ('gb', _i64), ('ga', _i64), ('gy', _i64), ('gx', _i64),
                ('bs', _i32), ('na', _i32), ('ny', _i32), ('nx', _i32), ('no', _i32), ('nc', _i32), ('nt', _i32),
                ('iou_t', _f32), ('gr', _f32), ('cp', _f32), ('cn', _f32), ('cls_pw', _f32), ('obj_pw', _f32),
                ('g_box', _f32), ('g_obj', _f32), ('g_cls', _f32)]


class CastDesc(C.Structure):
    _fields_ = [('x', _vp), ('y', _vp), ('pixels', _i64), ('c', _i32), ('ldx', _i32), ('ldy', _i32), ('dtype', _i32)]


OP_BN_STATS, OP_BN_FINALIZE, OP_BN_ACT_FWD, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY = 12, 13, 14, 15, 16
OP_WGRAD, OP_STEM_WGRAD, OP_DILATE2, OP_UPSAMPLE2_BWD, OP_CAST_F32, OP_NCHW_TO_NHWC, OP_POOL_BWD, OP_PACK_BATCH = 17, 18, 19, 20, 21, 22, 23, 24
OP_DW_WGRAD, OP_DW_DGRAD, OP_SE_BWD, OP_STEM_BWD = 25, 26, 27, 28

OP_KIND = {BnStatsDesc: OP_BN_STATS, BnFinalizeDesc: OP_BN_FINALIZE, BnActFwdDesc: OP_BN_ACT_FWD,
           BnBwdReduceDesc: OP_BN_BWD_REDUCE, BnBwdApplyDesc: OP_BN_BWD_APPLY, WgradDesc: OP_WGRAD,
           StemWgradDesc: OP_STEM_WGRAD, DilateDesc: OP_DILATE2, UpsampleBwdDesc: OP_UPSAMPLE2_BWD, CastDesc: OP_CAST_F32, LayoutDesc: OP_NCHW_TO_NHWC, PoolBwdDesc: OP_POOL_BWD, PackBatchDesc: OP_PACK_BATCH, DwWgradDesc: OP_DW_WGRAD, DwDgradDesc: OP_DW_DGRAD,
           SeBwdDesc: OP_SE_BWD, StemBwdDesc: OP_STEM_BWD,
           ConvDesc: OP_CONV, StemDesc: OP_STEM, PoolDesc: OP_POOL, CopyDesc: OP_COPY, AddDesc: OP_ADD,
           DecodeDesc: OP_DECODE, DwDesc: OP_DW, SeDesc: OP_SE, QCopyDesc: OP_QCOPY, QPoolDesc: OP_QPOOL, QAddDesc: OP_QADD}

_SIGNATURES = {
    'yh_abi_version': (C.c_int, []),
    'yh_error_string': (C.c_char_p, [C.c_int]),
    'yh_conv_pack_weights': (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'yh_conv2d_fwd': (C.c_int, [C.POINTER(ConvDesc), _vp]),
    'yh_conv2d_tile': (C.c_int, [C.POINTER(ConvDesc)]),
    'yh_conv2d_stats_rows': (_i64, [C.POINTER(ConvDesc)]),
    'yh_conv2d_bwd_stats_rows': (_i64, [C.POINTER(ConvDesc)]),
    'yh_plan_set_async_reduce': (C.c_int, [_vp, C.c_int]),
    'yh_qconv_pack_weights': (C.c_int, [_vp, _f32, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'yh_stem_pack_weights': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _vp, _vp, _vp]),
    'yh_conv2d_stem_fwd': (C.c_int, [C.POINTER(StemDesc), _vp]),
    'yh_conv2d_stem_stats_rows': (_i64, [C.POINTER(StemDesc)]),
    'yh_qmish_selftest': (C.c_int, [C.c_uint32, C.c_uint32, _f32, _vp, _vp]),
    'yh_dw_pack_weights': (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'yh_dwconv2d_fwd': (C.c_int, [C.POINTER(DwDesc), _vp]),
    'yh_se_fwd': (C.c_int, [C.POINTER(SeDesc), _vp]),
    'yh_maxpool2d_fwd': (C.c_int, [C.POINTER(PoolDesc), _vp]),
    'yh_qcopy': (C.c_int, [C.POINTER(QCopyDesc), _vp]),
    'yh_qpool': (C.c_int, [C.POINTER(PoolDesc), _vp]),
    'yh_qadd': (C.c_int, [C.POINTER(QAddDesc), _vp]),
    'yh_copy_channels': (C.c_int, [C.POINTER(CopyDesc), _vp]),
    'yh_add_channels': (C.c_int, [C.POINTER(AddDesc), _vp]),
    'yh_yolo_decode': (C.c_int, [C.POINTER(DecodeDesc), _vp]),
    'yh_yolo_decode_candidates': (C.c_int, [C.POINTER(DecodeDesc), _f32, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    'yh_nms_candidates': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    'yh_nms_sort': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'yh_nms_sort_cls': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'yh_nms_sort_tiles': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    'yh_nms_class_scan': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp, _vp, _vp, _vp, _vp]),
    'yh_nms_mask': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp]),
    'yh_nms_reduce': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'yh_nms_merge': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'yh_bn_stats': (C.c_int, [C.POINTER(BnDesc), _vp]),
    'yh_bn_finalize': (C.c_int, [C.POINTER(BnDesc), _vp]),
    'yh_bn_act_fwd': (C.c_int, [C.POINTER(BnDesc), _vp]),
    'yh_bn_act_bwd_reduce': (C.c_int, [C.POINTER(BnDesc), _vp]),
    'yh_bn_act_bwd_apply': (C.c_int, [C.POINTER(BnDesc), _vp]),
    'yh_conv_pack_weights_dgrad': (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'yh_conv_pack_weights_dgrad_phase': (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
    'yh_conv2d_wgrad': (C.c_int, [C.POINTER(WgradDesc), _vp]),
    'yh_conv2d_wgrad_workspace': (_i64, [C.POINTER(WgradDesc)]),
    'yh_conv2d_wgrad_kernel': (C.c_int, [C.POINTER(WgradDesc)]),
    'yh_stem_bwd_workspace': (_i64, [C.POINTER(StemBwdDesc)]),
    'yh_stem_bwd': (C.c_int, [C.POINTER(StemBwdDesc), _vp]),
    'yh_bn_reduce_workspace': (_i64, [C.POINTER(BnDesc)]),
    'yh_stem_wgrad': (C.c_int, [C.POINTER(WgradDesc), _vp]),
    'yh_dilate2': (C.c_int, [C.POINTER(ResampleDesc), _vp]),
    'yh_upsample2_bwd': (C.c_int, [C.POINTER(ResampleDesc), _vp]),
    'yh_cast_f32': (C.c_int, [C.POINTER(CastDesc), _vp]),
    'yh_maxpool2d_bwd': (C.c_int, [C.POINTER(PoolBwdDesc), _vp]),
    'yh_pack_batch': (C.c_int, [_vp, C.c_int, _vp]),
    'yh_dw_wgrad': (C.c_int, [C.POINTER(DwBwdDesc), _vp]),
    'yh_dw_wgrad_workspace': (_i64, [C.POINTER(DwBwdDesc)]),
    'yh_dw_dgrad': (C.c_int, [C.POINTER(DwBwdDesc), _vp]),
    'yh_se_bwd': (C.c_int, [C.POINTER(SeBwdDesc), _vp]),
    'yh_yolo_loss_fwd': (C.c_int, [C.POINTER(LossDesc), _vp]),
    'yh_yolo_loss_bwd': (C.c_int, [C.POINTER(LossDesc), _vp]),
    'yh_nchw_to_nhwc': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    'yh_letterbox_fwd': (C.c_int, [C.POINTER(LetterboxDesc), _vp]),
    'yh_mosaic_affine_hsv': (C.c_int, [C.POINTER(MosaicDesc), _vp]),
    'yh_ptq_search_workspace': (_i64, [_i64]),
    'yh_ptq_cos_search': (C.c_int, [_vp, _i64, _f32, C.c_int, _f32, _f32, C.c_int, _vp, _i64, _vp, _vp, _vp]),
    'yh_absmax': (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'yh_plan_create': (_vp, []),
    'yh_plan_destroy': (None, [_vp]),
    'yh_plan_add': (C.c_int, [_vp, C.c_int, _vp, C.c_int]),
    'yh_plan_add_fixup': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _i64]),
    'yh_plan_bind_slot': (C.c_int, [_vp, C.c_int, _vp]),
    'yh_plan_num_ops': (C.c_int, [_vp]),
    'yh_plan_run': (C.c_int, [_vp, _vp]),
    'yh_plan_run_range': (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    'yh_plan_set_lane': (C.c_int, [_vp, C.c_int, C.c_int]),
    'yh_plan_add_dep': (C.c_int, [_vp, C.c_int, C.c_int]),
    'yh_plan_set_timing': (C.c_int, [_vp, C.c_int]),
    'yh_plan_get_timings': (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    'yh_plan_graph_capture': (C.c_int, [_vp, _vp]),
    'yh_plan_graph_launch': (C.c_int, [_vp, _vp]),
    'yh_plan_graph_reset': (C.c_int, [_vp]),
}

EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 2
_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises HipLibraryError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError('HIP library not built: %s is missing (run `make -C %s`). There is no eager '
                              'fallback for CUDA tensors.' % (LIB_PATH, os.path.join(os.path.dirname(_HERE), 'csrc')))
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError('%s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.yh_abi_version() != ABI_VERSION:
        raise HipLibraryError('ABI mismatch: library %d, binding %d' % (lib.yh_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().yh_error_string(rc)
        raise RuntimeError('libyolo_hip %s failed: %s (code %d)' % (what, msg.decode() if msg else '?', rc))


SLOT_NULL = C.c_void_p(-1).value     # yh_plan_bind_slot: 'bound to nothing' (include/yolo_hip.h YH_SLOT_NULL)


def ptr(t, elem_offset=0):
    """Device address of a torch tensor (+ element offset) as an int usable for c_void_p fields."""
    if t is None:
        return None
    return t.data_ptr() + elem_offset * t.element_size()


def on_device(t):
    """Context that makes ``t``'s GPU the current one: every launch goes to torch's *current* device and stream, so an entry
    point handed a tensor of another GPU must switch first (a no-op for CPU tensors in the host-emulation tests)."""
    import contextlib
    import torch
    if t is not None and getattr(t, 'is_cuda', False):
        return torch.cuda.device(t.device)
    return contextlib.nullcontext()


def stream_ptr():
    """Current torch HIP stream as a void* (NULL stream when torch has no GPU: host-emulation tests)."""
    import torch
    if not torch.cuda.is_available():
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
