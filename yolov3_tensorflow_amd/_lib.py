"""ctypes binding of the C-ABI in include/yolov3_amd.h (libyolov3_amd.so, built by csrc/Makefile).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (YOLO_LIB_PATH: a diagnostic build of the same C-ABI, csrc/Makefile target `diag`, for the stamp probes under tools/probes -- still the native
# library or an exception, never a fallback)
LIB_PATH = os.environ.get('YOLO_LIB_PATH') or os.path.join(_HERE, 'libyolov3_amd.so')
LIB_PATH_FP16 = os.path.join(_HERE, 'libyolov3_amd_fp16.so')

MAX_ANCHORS = 8


class ConvProblem(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('N', 'H', 'W', 'Cin', 'C0', 'Cout', 'R', 'S', 'stride', 'pad_t', 'pad_l', 'Ho', 'Wo')]


class MixProblem(C.Structure):
    _fields_ = [('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32), ('split', C.c_int32 * 5), ('ksize', C.c_int32 * 4)]


class LossConfig(C.Structure):
    _fields_ = [('H', C.c_int32 * 3), ('W', C.c_int32 * 3), ('B', C.c_int32 * 3), ('ldc', C.c_int32 * 3),
                ('anchor_w', (C.c_float * MAX_ANCHORS) * 3), ('anchor_h', (C.c_float * MAX_ANCHORS) * 3),
                ('L', C.c_int32), ('T', C.c_int32), ('iou_thresh', C.c_float),
                ('w_xy', C.c_float * 3), ('w_wh', C.c_float * 3), ('w_noobj', C.c_float * 3), ('w_obj', C.c_float * 3),
                ('w_cls', C.c_float * 3), ('w_rect', C.c_float * 3),
                ('rectified_coord_num', C.c_int32), ('is_focal_loss', C.c_int32),
                ('focal_alpha', C.c_float), ('focal_gamma', C.c_float), ('is_tiou_recall', C.c_int32), ('eps', C.c_float),
                ('grad_scale16', C.c_float)]


class ImageDesc(C.Structure):
    _fields_ = [('offset', C.c_int64), ('h', C.c_int32), ('w', C.c_int32), ('nh', C.c_int32), ('nw', C.c_int32), ('top', C.c_int32),
                ('left', C.c_int32), ('noise', C.c_int32), ('color_order', C.c_int32), ('brightness_delta', C.c_float),
                ('saturation_factor', C.c_float), ('contrast_factor', C.c_float), ('seed0', C.c_uint32), ('seed1', C.c_uint32),
                ('reserved', C.c_int32)]


P, I, I64, F = C.c_void_p, C.c_int, C.c_int64, C.c_float
CP = C.POINTER(ConvProblem)
MP = C.POINTER(MixProblem)
LP = C.POINTER(LossConfig)

# name -> (restype, argtypes); must list every function declared in include/yolov3_amd.h
SIGNATURES = {
    'yolo_abi_version': (I, []),
    'yolo_abi_dtype': (I, []),
    'yolo_last_error': (C.c_char_p, []),
    'yolo_crc32c': (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    'yolo_seq_begin': (I, []),
    'yolo_seq_mark': (I, []),
    'yolo_seq_end': (I, []),
    'yolo_seq_fork': (I, [P, P]),
    'yolo_seq_fork_local': (I, [P, P]),
    'yolo_seq_run': (I, [I, I, I]),
    'yolo_seq_free': (I, [I]),
    'yolo_conv2d_stat_rows': (I, [CP]),
    'yolo_conv2d_fwd_plan': (I, [CP, P]),
    'yolo_set_tuning': (I, [C.c_char_p, I]),
    'yolo_conv2d_fwd': (I, [CP, P, P, P, P, P, I, P, P, P]),
    'yolo_conv2d_dgrad': (I, [CP, P, P, P, I, P]),
    'yolo_conv2d_dgrad_bn_rows': (I, [CP]),
    'yolo_conv2d_stat_group_layout': (I, [CP, P]),
    'yolo_conv2d_fwd_g': (I, [CP, P, P, P, P, P, P, P]),
    'yolo_conv2d_dgrad_bn_group_layout': (I, [CP, P]),
    'yolo_conv2d_dgrad_bn_g': (I, [CP, P, P, P, I, P, P, P, P, P, P, P, P, P, P]),
    'yolo_conv2d_dgrad_bn': (I, [CP, P, P, P, I, P, P, P, P, P, P, P, P, P, P]),
    'yolo_conv2d_dgrad_bn_acc': (I, [CP, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P]),
    'yolo_conv2d_fwd_acc': (I, [CP, P, P, P, P, P, P]),
    'yolo_acc_words': (I64, [I, I]),
    'yolo_zero_words': (I, [P, I64, P]),
    'yolo_bn_finalize_act_fwd_acc': (I, [P, I, F, P, P, F, F, P, P, P, P, P, P, P, P, P, P, I64, I, P]),
    'yolo_bn_bwd_finalize_apply_acc': (I, [P, I, F, P, P, P, P, P, P, P, P, P, P, I, P, I, I64, P]),
    'yolo_conv2d_dgrad_add': (I, [CP, P, P, P, P, P]),
    'yolo_conv2d_dgrad_classed': (I, [CP]),
    'yolo_conv2d_dgrad_even': (I, [CP, P, P, P, I, P]),
    'yolo_conv2d_wgrad': (I, [CP, P, P, P, P, I, P]),
    'yolo_conv2d_wgrad_workspace_bytes': (C.c_size_t, [CP]),
    'yolo_conv2d_wgrad_reduce': (I, [CP, P, P, P, P, P, C.c_size_t, I, P]),
    'yolo_conv2d_wgrad_splits': (I, [CP]),
    'yolo_conv2d_wgrad_slabs': (I, [CP, P, P, P, P, P, C.c_size_t, P]),
    'yolo_wgrad_reduce_batched': (I, [P, I, I, P, P, P]),
    'yolo_stem_pool_bwd_slabs': (I, [CP, I, I, I, I, I]),
    'yolo_stem_pool_bwd_wgrad': (I, [CP, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, P, C.c_size_t, P]),
    'yolo_repack_dgrad_weights': (I, [P, P, I, I, I, I, P]),
    'yolo_repack_dgrad_weights_batched': (I, [P, P, P, I, I, P]),
    'yolo_reduce_rows': (I, [I, I]),
    'yolo_bn_stats': (I, [P, I, I, P, P]),
    'yolo_bn_finalize': (I, [P, P, I, I64, I, F, P, P, F, F, P, P, P, P, P, P, P]),
    'yolo_bn_finalize_act_fwd': (I, [P, P, I, I64, I, F, P, P, F, F, P, P, P, P, P, P, P, P, P, P, I64, I, P]),
    'yolo_bn_bwd_finalize_apply': (I, [P, I, I64, I64, I, F, P, P, P, P, P, P, P, P, P, P, I, P, I, I64, P]),
    'yolo_bn_finalize_grouped': (I, [P, P, I, C.c_int64, I, C.c_float, I, P, P, P, C.c_float, C.c_float, P, P, P, P, P, P, P]),
    'yolo_bn_bwd_finalize_grouped': (I, [P, I, C.c_int64, C.c_int64, I, I, C.c_float, I, P, P, P, P, P, P]),
    'yolo_bn_act_fwd': (I, [P, P, P, P, P, P, P, I64, I, I, P]),
    'yolo_bn_act_fwd_mask': (I, [P, P, P, P, P, P, P, P, I64, I, P]),
    'yolo_bn_pool_fwd': (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    'yolo_bn_act_bwd_reduce': (I, [P, P, I, P, P, P, P, P, P, I, I, P, P]),
    'yolo_bn_bwd_finalize': (I, [P, I, I64, I64, I, I, F, P, P, P, P, P]),
    'yolo_bn_act_bwd_apply': (I, [P, P, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, I, I64, I, P]),
    'yolo_bn_bwd_fused_workspace_floats': (C.c_int64, [I]),
    'yolo_bn_bwd_fused_sync_words': (I, []),
    'yolo_bn_act_bwd_fused': (I, [P, P, I, C.c_int64, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, I, P, P, P]),
    'yolo_bn_act_bwd_fused_grouped': (I, [P, P, I, C.c_int64, I, P, P, P, P, I, P, P, P, P, I, P, P, P, P, P, P, P, P, I, P, P, P]),
    'yolo_bn_fused_timeouts': (I, [P, P]),
    'yolo_bn_fused_set_host_flag': (I, [P, P]),
    'yolo_bn_pool_bwd_reduce': (I, [P, P, P, I, P, P, P, P, P, I, I, I, I, I, I, I, I, P, P]),
    'yolo_bn_pool_bwd_apply': (I, [P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    'yolo_upcat_split_bwd': (I, [P, P, I, P, I, I, I, I, I, I, P]),
    'yolo_pack_input': (I, [P, P, I64, I, P]),
    'yolo_reduce_partials': (I, [P, I, I64, I, P, P]),
    'yolo_bn_eval_scale_shift': (I, [P, P, P, P, F, P, P, I, P]),
    'yolo_dwconv_mix_fwd': (I, [MP, P, P, P, P, P, P, P]),
    'yolo_dwconv_mix_dgrad': (I, [MP, P, P, P, P, P, P, I, P]),
    'yolo_dwconv_mix_wgrad_workspace_bytes': (C.c_size_t, [MP]),
    'yolo_dwconv_mix_wgrad': (I, [MP, P, P, P, P, P, P, P, C.c_size_t, I, P]),
    'yolo_loss_workspace_bytes': (I64, [LP, I]),
    'yolo_loss_fwd_bwd': (I, [LP, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'yolo_decode_head': (I, [P, I, I, I, I, I, I, P, F, P, P, P, P, P]),
    'yolo_filter_boxes': (I, [P, P, I, I, I, I, I, F, I, P, P, P, P]),
    'yolo_nms_max_candidates': (I, []),
    'yolo_nms_heads': (I, [P, P, P, P, P, P, I, I, C.c_double, I, P, P, P, P, P]),
    'yolo_letterbox_workspace_bytes': (I64, [I]),
    'yolo_letterbox_augment': (I, [P, P, I, I, I, I, P, P, P, P]),
    'yolo_radam_schedule': (I, [P, P, F, F, F, F, P]),
    'yolo_optimizer_schedule': (I, [P, P, I, F, F, F, P]),
    'yolo_radam_l2_blocks': (I, [I64]),
    'yolo_radam_l2_step': (I, [P, P, P, P, P, P, P, I64, P, F, F, F, F, I, P, P, P]),
    'yolo_cast_f32_to_bf16': (I, [P, P, I64, P]),
    'yolo_sum_partials': (I, [P, I, P, P, P, P]),
}

_libs = {}


class YoloNativeError(RuntimeError):
    pass


def load(dtype=None):
    """Load the library of the current compute dtype (backend.compute_dtype(); once each).  Raises if it has not been built: there
    is no CPU fallback."""
    if dtype is None:
        from yolov3_tensorflow_amd import backend
        dtype = backend.compute_dtype()
    lib = _libs.get(dtype)
    if lib is not None:
        return lib
    path = LIB_PATH_FP16 if dtype == 'float16' else LIB_PATH
    if not os.path.exists(path):
        raise YoloNativeError('%s not found: build it with `make -C %s` (or __graft_entry__.build()); the MI355X path has no '
                              'fallback' % (path, os.path.join(_HERE, 'csrc')))
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so): it must be in the process BEFORE this library is loaded, or the dynamic
    # loader binds the kernels to the system copy and every launch fails with "no ROCm-capable device is detected" once torch has initialised
    # the other one (seen with `python __graft_entry__.py smoke`: build() loaded the library before anything had imported torch)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.yolo_abi_version() != 1 or lib.yolo_abi_dtype() != (1 if dtype == 'float16' else 0):
        raise YoloNativeError('ABI version / dtype mismatch in ' + path)
    _libs[dtype] = lib
    return lib


def check(status, what=''):
    if status != 0:
        msg = load().yolo_last_error()
        raise YoloNativeError('%s failed with status %d: %s' % (what, status, msg.decode() if msg else ''))
