"""Thin torch-tensor wrappers over the C-ABI (pointers + current HIP stream).  PyTorch is only the allocator / stream
provider here; every computation is a kernel of libyolov3_amd.so.  No fallback paths.
"""
import ctypes as C
import numpy as np
import torch
from . import _lib
from ._lib import ConvProblem, LossConfig, MixProblem, check


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def same_pad(size, k, s):
    """TF 'same' padding: (out, pad_before)"""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_problem(N, H, W, Cin, Cout, k=3, stride=1, padding='same', C0=0):
    if padding == 'same':
        Ho, pt = same_pad(H, k, stride)
        Wo, pl = same_pad(W, k, stride)
    else:
        Ho, Wo, pt, pl = (H - k) // stride + 1, (W - k) // stride + 1, 0, 0
    return ConvProblem(N, H, W, Cin, C0, Cout, k, k, stride, pt, pl, Ho, Wo)


def pad_channels(c):
    """detection-conv output channels are padded to 64 * 2^k (GEMM tile and power-of-two k-chunk constraints)"""
    p = 64
    while p < c:
        p *= 2
    return p


_tuning_epoch = 0


def set_tuning(name, value):
    """test / benchmark hook (yolo_set_tuning): override a kernel-selection heuristic.  Every call advances tuning_epoch(): launch
    decisions recorded earlier (the native step sequence of YOLOv3Model, buffer sizes planned at Graph.finalize) are stale after it"""
    global _tuning_epoch
    check(_lib.load().yolo_set_tuning(name.encode(), int(value)), 'yolo_set_tuning')
    _tuning_epoch += 1


def tuning_epoch():
    return _tuning_epoch


def conv2d_stat_rows(p):
    r = _lib.load().yolo_conv2d_stat_rows(C.byref(p))
    if r < 0:
        raise _lib.YoloNativeError('yolo_conv2d_stat_rows rejected the problem')
    return r


CONV_FAMILIES = ('igemm', 'strip', 'retired', 'stem', 'stream', 's32')      # (2: the big-tile kernel of round 3, removed in round 4)


def conv2d_fwd_plan(p):
    """the kernel yolo_conv2d_fwd would launch for p under the current tuning (yolo_conv2d_fwd_plan): dict(family, bm, bn, tile_pixels,
    workgroups, lds_bytes, ring)"""
    info = (C.c_int32 * 8)()
    check(_lib.load().yolo_conv2d_fwd_plan(C.byref(p), info), 'yolo_conv2d_fwd_plan')
    return dict(family=CONV_FAMILIES[info[0]], bm=info[1], bn=info[2], tile_pixels=info[3], workgroups=info[4], lds_bytes=info[5],
                ring=info[6])


def acc_words(Q, Cc):
    """int64 words of one exact accumulator block for Q quantities of Cc channels (yolo_acc_words)"""
    return int(_lib.load().yolo_acc_words(int(Q), int(Cc)))


def zero_words(t):
    """zero an int64 tensor with a library launch (part of a recorded step, unlike a torch fill)"""
    check(_lib.load().yolo_zero_words(_p(t), t.numel(), _stream()), 'yolo_zero_words')


def conv2d_fwd(p, src1, w_fwd, y, src0=None, bias=None, stat_sum=None, stat_sq=None, stat_acc=None, grouped=False):
    """``grouped``: the statistics rows are laid out as conv2d_stat_group_layout says (two-level rows, yolo_conv2d_fwd_g).  ``stat_acc``: an exact accumulator block (acc_words(2, Cout) int64, zeroed this step) that receives the BatchNorm statistics instead
    of the per-tile rows stat_sum / stat_sq (yolo_conv2d_fwd_acc)"""
    if stat_acc is not None:
        if bias is not None or stat_sum is not None or y.dtype == torch.float32:
            raise ValueError('stat_acc goes with a 16-bit output, no bias and no statistics rows')
        check(_lib.load().yolo_conv2d_fwd_acc(C.byref(p), _p(src0), _p(src1), _p(w_fwd), _p(y), _p(stat_acc), _stream()), 'yolo_conv2d_fwd_acc')
        return
    if grouped:
        if bias is not None or stat_sum is None or y.dtype == torch.float32:
            raise ValueError('grouped statistics rows go with a 16-bit output, no bias and statistics')
        check(_lib.load().yolo_conv2d_fwd_g(C.byref(p), _p(src0), _p(src1), _p(w_fwd), _p(y), _p(stat_sum), _p(stat_sq), _stream()), 'yolo_conv2d_fwd_g')
        return
    check(_lib.load().yolo_conv2d_fwd(C.byref(p), _p(src0), _p(src1), _p(w_fwd), _p(bias), _p(y),
                                      1 if y.dtype == torch.float32 else 0, _p(stat_sum), _p(stat_sq), _stream()), 'yolo_conv2d_fwd')


def conv2d_dgrad(p, dy, w_dgrad, dx, accumulate=False, bn=None, addend=None, even_only=False):
    """``bn`` = dict(mask, y, mean, rstd, partial[, y2, mean2, rstd2]): dx is the output gradient of a BatchNorm unit -- leave the masked
    gradient in dx and the unit's backward partial sums in ``partial`` (yolo_conv2d_dgrad_bn).  ``addend``: dx = addend + gradient
    (the fan-in add with the other contribution read from its own buffer).  ``accumulate`` = 2: only the even / even positions hold a previous
    contribution (3x3 stride-2 problems); ``even_only``: a 1x1 stride-2 gradient that writes just those positions (yolo_conv2d_dgrad_even)"""
    if even_only:
        if bn is not None or addend is not None:
            raise ValueError('even_only data gradient takes neither a fused BatchNorm reduce (bn=) nor an external addend')
        check(_lib.load().yolo_conv2d_dgrad_even(C.byref(p), _p(dy), _p(w_dgrad), _p(dx), int(accumulate), _stream()), 'yolo_conv2d_dgrad_even')
        return
    if bn is None:
        if addend is not None:
            check(_lib.load().yolo_conv2d_dgrad_add(C.byref(p), _p(dy), _p(w_dgrad), _p(dx), _p(addend), _stream()), 'yolo_conv2d_dgrad_add')
        else:
            check(_lib.load().yolo_conv2d_dgrad(C.byref(p), _p(dy), _p(w_dgrad), _p(dx), int(accumulate), _stream()), 'yolo_conv2d_dgrad')
        return
    if bn.get('acc') is not None:            # tile sums into an exact accumulator block (acc_words(3, Cin)) instead of partial rows
        check(_lib.load().yolo_conv2d_dgrad_bn_acc(C.byref(p), _p(dy), _p(w_dgrad), _p(dx), int(accumulate), _p(addend), _p(bn.get('mask')),
                                                   _p(bn['y']), _p(bn['mean']), _p(bn['rstd']), _p(bn.get('y2')), _p(bn.get('mean2')),
                                                   _p(bn.get('rstd2')), None, _p(bn['acc']), _stream()), 'yolo_conv2d_dgrad_bn_acc')
        return
    fn, name = (_lib.load().yolo_conv2d_dgrad_bn_g, 'yolo_conv2d_dgrad_bn_g') if bn.get('grouped') else (_lib.load().yolo_conv2d_dgrad_bn, 'yolo_conv2d_dgrad_bn')
    check(fn(C.byref(p), _p(dy), _p(w_dgrad), _p(dx), int(accumulate), _p(addend), _p(bn.get('mask')), _p(bn['y']),
             _p(bn['mean']), _p(bn['rstd']), _p(bn.get('y2')), _p(bn.get('mean2')), _p(bn.get('rstd2')), _p(bn['partial']), _stream()), name)


def _layout(fn, p, name):
    info = (C.c_int32 * 4)()
    check(fn(C.byref(p), info), name)
    return dict(alloc_rows=int(info[0]), groups=int(info[1]), group=int(info[2]), raw_rows=int(info[3]))


def conv2d_stat_group_layout(p):
    """two-level statistics rows of conv2d_fwd(grouped=True): dict(alloc_rows, groups, group, raw_rows) (yolo_conv2d_stat_group_layout)"""
    return _layout(_lib.load().yolo_conv2d_stat_group_layout, p, 'yolo_conv2d_stat_group_layout')


def conv2d_dgrad_bn_group_layout(p):
    """the same for the [rows][3][Cin] partial buffer of conv2d_dgrad(bn=..., grouped=True)"""
    return _layout(_lib.load().yolo_conv2d_dgrad_bn_group_layout, p, 'yolo_conv2d_dgrad_bn_group_layout')


def conv2d_dgrad_classed(p):
    """True if conv2d_dgrad runs this problem as four parity classes (3x3, stride 2): accumulate = 2 is accepted then"""
    return bool(_lib.load().yolo_conv2d_dgrad_classed(C.byref(p)))


def conv2d_dgrad_bn_rows(p):
    """rows of the [rows][3][Cin] partial buffer of conv2d_dgrad(bn=...); -1 if the problem cannot take the fused form"""
    return int(_lib.load().yolo_conv2d_dgrad_bn_rows(C.byref(p)))


def conv2d_wgrad(p, src1, dy, dw, src0=None, split_k=0):
    check(_lib.load().yolo_conv2d_wgrad(C.byref(p), _p(src0), _p(src1), _p(dy), _p(dw), split_k, _stream()), 'yolo_conv2d_wgrad')


def conv2d_wgrad_workspace_bytes(p):
    return int(_lib.load().yolo_conv2d_wgrad_workspace_bytes(C.byref(p)))


def conv2d_wgrad_reduce(p, src1, dy, dw, workspace, src0=None, accumulate=False):
    """two-phase weight gradient (split slabs -> workspace, then one summing pass): no atomics, dw is overwritten unless accumulate"""
    check(_lib.load().yolo_conv2d_wgrad_reduce(C.byref(p), _p(src0), _p(src1), _p(dy), _p(dw), _p(workspace),
                                               workspace.numel() * workspace.element_size(), int(accumulate), _stream()),
          'yolo_conv2d_wgrad_reduce')


def conv2d_wgrad_splits(p):
    r = _lib.load().yolo_conv2d_wgrad_splits(C.byref(p))
    if r < 1:
        raise _lib.YoloNativeError('yolo_conv2d_wgrad_splits rejected the problem')
    return r


def conv2d_wgrad_slabs(p, src1, dy, dw, slabs, src0=None):
    """the slab pass of the two-phase weight gradient into this layer's private slab region (summed later, per bucket, by
    wgrad_reduce_batched); with a single split the kernel writes dw directly and ``slabs`` may be None"""
    check(_lib.load().yolo_conv2d_wgrad_slabs(C.byref(p), _p(src0), _p(src1), _p(dy), _p(dw), _p(slabs),
                                              0 if slabs is None else slabs.numel() * slabs.element_size(), _stream()), 'yolo_conv2d_wgrad_slabs')


REDUCE_WIDE_SLABS = 64      # YOLO_REDUCE_WIDE_SLABS: layers with this many slabs get 16-column workgroups in wgrad_reduce_batched


def reduce_blocks(n4, nslabs):
    """workgroups of wgrad_reduce_batched for a layer of n4 float4s per slab"""
    cols = 16 if nslabs >= REDUCE_WIDE_SLABS else 64
    return (n4 + cols - 1) // cols


def stem_pool_bwd_slabs(p, C_pool, Ho, Wo, pt, pl):
    """slabs the fused stem backward (stem_pool_bwd_wgrad) writes, 0 if this stem is not covered"""
    return int(_lib.load().yolo_stem_pool_bwd_slabs(C.byref(p), C_pool, Ho, Wo, pt, pl))


def stem_pool_bwd_wgrad(p, x, dout, out, argmax, relu, y, a1, mean, rstd, k1, k2, Ho, Wo, pt, pl, slabs):
    check(_lib.load().yolo_stem_pool_bwd_wgrad(C.byref(p), _p(x), _p(dout), _p(out), _p(argmax), int(relu), _p(y), _p(a1), _p(mean), _p(rstd),
                                               _p(k1), _p(k2), Ho, Wo, pt, pl, _p(slabs), slabs.numel() * 4, _stream()),
          'yolo_stem_pool_bwd_wgrad')


def wgrad_reduce_batched(table_dev, nentries, total_blocks, arena, grads):
    check(_lib.load().yolo_wgrad_reduce_batched(_p(table_dev), nentries, total_blocks, _p(arena), _p(grads), _stream()), 'yolo_wgrad_reduce_batched')


def repack_dgrad_weights(w_fwd, w_dgrad, Cout, R, S, Cin):
    check(_lib.load().yolo_repack_dgrad_weights(_p(w_fwd), _p(w_dgrad), Cout, R, S, Cin, _stream()), 'yolo_repack_dgrad_weights')


def repack_dgrad_weights_batched(w_fwd_flat, w_dgrad_flat, table_dev, nlayers, total_tiles):
    check(_lib.load().yolo_repack_dgrad_weights_batched(_p(w_fwd_flat), _p(w_dgrad_flat), _p(table_dev), nlayers, total_tiles, _stream()),
          'yolo_repack_dgrad_weights_batched')


def reduce_rows(M, Cc):
    r = _lib.load().yolo_reduce_rows(M, Cc)
    if r < 0:
        raise _lib.YoloNativeError('yolo_reduce_rows(%d, %d) rejected' % (M, Cc))
    return r


def bn_stats(x, M, Cc, partial):
    check(_lib.load().yolo_bn_stats(_p(x), M, Cc, _p(partial), _stream()), 'yolo_bn_stats')


def bn_finalize(psum, psq, P, row_stride, Cc, count, gamma, beta, eps, momentum, moving_mean, moving_var, scale, shift, mean, rstd):
    check(_lib.load().yolo_bn_finalize(_p(psum), _p(psq), P, row_stride, Cc, float(count), _p(gamma), _p(beta), eps, momentum,
                                       _p(moving_mean), _p(moving_var), _p(scale), _p(shift), _p(mean), _p(rstd), _stream()),
          'yolo_bn_finalize')


def bn_finalize_act_fwd(psum, psq, P, row_stride, Cc, count, gamma, beta, eps, momentum, moving_mean, moving_var, scale, shift, mean, rstd,
                        y, out, M, relu, res=None, mask=None):
    """bn_finalize + bn_act_fwd in one launch (small maps: few partial rows, C % 32 == 0)"""
    check(_lib.load().yolo_bn_finalize_act_fwd(_p(psum), _p(psq), P, row_stride, Cc, float(count), _p(gamma), _p(beta), eps, momentum,
                                               _p(moving_mean), _p(moving_var), _p(scale), _p(shift), _p(mean), _p(rstd), _p(y), _p(res),
                                               _p(out), _p(mask), M, int(bool(relu)), _stream()), 'yolo_bn_finalize_act_fwd')


def bn_finalize_act_fwd_acc(stat_acc, Cc, count, gamma, beta, eps, momentum, moving_mean, moving_var, scale, shift, mean, rstd, y, out, M, relu,
                            res=None, mask=None):
    """finalize + apply in one launch with the statistics read from an exact accumulator block (conv2d_fwd(stat_acc=...)): any layer size"""
    check(_lib.load().yolo_bn_finalize_act_fwd_acc(_p(stat_acc), Cc, float(count), _p(gamma), _p(beta), eps, momentum, _p(moving_mean),
                                                   _p(moving_var), _p(scale), _p(shift), _p(mean), _p(rstd), _p(y), _p(res), _p(out), _p(mask), M,
                                                   int(bool(relu)), _stream()), 'yolo_bn_finalize_act_fwd_acc')


def bn_bwd_finalize_apply_acc(acc, Cc, count, dgamma, dbeta, k1, k2, g, y, a1, mean, rstd, M, dy, acc_dy=False, dres=None, acc_dres=False):
    check(_lib.load().yolo_bn_bwd_finalize_apply_acc(_p(acc), Cc, float(count), _p(dgamma), _p(dbeta), _p(k1), _p(k2), _p(g), _p(y), _p(a1),
                                                     _p(mean), _p(rstd), _p(dy), int(bool(acc_dy)), _p(dres), int(bool(acc_dres)), M, _stream()),
          'yolo_bn_bwd_finalize_apply_acc')


def bn_bwd_finalize_apply(partial, P, Cc, count, dgamma, dbeta, k1, k2, g, y, a1, mean, rstd, M, dy, acc_dy=False, dres=None, acc_dres=False,
                          row_stride=None, q_stride=None):
    """bn_bwd_finalize (which = 1) + bn_act_bwd_apply (masked gradient g, relu = 0) in one launch"""
    q = Cc if q_stride is None else q_stride
    rs = 3 * q if row_stride is None else row_stride
    check(_lib.load().yolo_bn_bwd_finalize_apply(_p(partial), P, rs, q, Cc, float(count), _p(dgamma), _p(dbeta), _p(k1), _p(k2), _p(g), _p(y),
                                                 _p(a1), _p(mean), _p(rstd), _p(dy), int(bool(acc_dy)), _p(dres), int(bool(acc_dres)), M,
                                                 _stream()), 'yolo_bn_bwd_finalize_apply')


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def bn_finalize_grouped(psum, psq, P, row_stride, Cc, count, split, gammas, betas, eps, momentum, moving_means, moving_vars, scale, shift,
                        mean, rstd):
    """one launch for the BatchNorms of consecutive channel groups (split = boundaries, len(gammas) groups)"""
    n = len(gammas)
    sp = (C.c_int32 * (n + 1))(*split)
    mm = None if moving_means is None else _ptr_array(moving_means)
    mv = None if moving_vars is None else _ptr_array(moving_vars)
    check(_lib.load().yolo_bn_finalize_grouped(_p(psum), _p(psq), P, row_stride, Cc, float(count), n, sp, _ptr_array(gammas),
                                               _ptr_array(betas), eps, momentum, mm, mv, _p(scale), _p(shift), _p(mean), _p(rstd),
                                               _stream()), 'yolo_bn_finalize_grouped')


def bn_bwd_finalize_grouped(partial, P, Cc, which, count, split, dgammas, dbetas, k1, k2, row_stride=None, q_stride=None):
    n = len(dgammas)
    sp = (C.c_int32 * (n + 1))(*split)
    q = Cc if q_stride is None else q_stride
    rs = 3 * q if row_stride is None else row_stride
    check(_lib.load().yolo_bn_bwd_finalize_grouped(_p(partial), P, rs, q, Cc, which, float(count), n, sp, _ptr_array(dgammas),
                                                   _ptr_array(dbetas), _p(k1), _p(k2), _stream()), 'yolo_bn_bwd_finalize_grouped')


def bn_act_fwd(y, scale, shift, out, M, Cc, relu, res=None, res_scale=None, res_shift=None, mask=None):
    """``mask`` (uint8 [M * C / 8], ReLU only): also leave the activation's sign bits for the backward pass (relu = 2 there)"""
    if mask is not None and relu:
        check(_lib.load().yolo_bn_act_fwd_mask(_p(y), _p(scale), _p(shift), _p(res), _p(res_scale), _p(res_shift), _p(out), _p(mask), M, Cc,
                                               _stream()), 'yolo_bn_act_fwd_mask')
        return
    check(_lib.load().yolo_bn_act_fwd(_p(y), _p(scale), _p(shift), _p(res), _p(res_scale), _p(res_shift), _p(out), M, Cc, int(relu),
                                      _stream()), 'yolo_bn_act_fwd')


def bn_pool_fwd(y, scale, shift, out, argmax, N, H, W, Cc, Ho, Wo, pt, pl, relu):
    check(_lib.load().yolo_bn_pool_fwd(_p(y), _p(scale), _p(shift), _p(out), _p(argmax), N, H, W, Cc, Ho, Wo, pt, pl, int(relu),
                                       _stream()), 'yolo_bn_pool_fwd')


def bn_act_bwd_reduce(dout, out, relu, y, mean, rstd, M, Cc, partial, y2=None, mean2=None, rstd2=None):
    check(_lib.load().yolo_bn_act_bwd_reduce(_p(dout), _p(out), int(relu), _p(y), _p(mean), _p(rstd), _p(y2), _p(mean2), _p(rstd2),
                                             M, Cc, _p(partial), _stream()), 'yolo_bn_act_bwd_reduce')


def bn_bwd_finalize(partial, P, Cc, which, count, dgamma, dbeta, k1, k2, row_stride=None, q_stride=None):
    """partial: [P][3][q_stride] (default q_stride = Cc); pass a tensor view offset to a channel sub-range for grouped BatchNorms"""
    q = Cc if q_stride is None else q_stride
    rs = 3 * q if row_stride is None else row_stride
    check(_lib.load().yolo_bn_bwd_finalize(_p(partial), P, rs, q, Cc, which, float(count), _p(dgamma), _p(dbeta), _p(k1), _p(k2), _stream()),
          'yolo_bn_bwd_finalize')


def bn_act_bwd_apply(dout, out, relu, M, Cc, y=None, a1=None, mean=None, rstd=None, k1=None, k2=None, dy=None, acc_dy=False,
                     y2=None, a2=None, mean2=None, rstd2=None, k1b=None, k2b=None, dy2=None, dres=None, acc_dres=False):
    check(_lib.load().yolo_bn_act_bwd_apply(_p(dout), _p(out), int(relu), _p(y), _p(a1), _p(mean), _p(rstd), _p(k1), _p(k2), _p(dy),
                                            int(acc_dy), _p(y2), _p(a2), _p(mean2), _p(rstd2), _p(k1b), _p(k2b), _p(dy2), _p(dres),
                                            int(acc_dres), M, Cc, _stream()), 'yolo_bn_act_bwd_apply')


def bn_bwd_fused_workspace_floats(Cc):
    return int(_lib.load().yolo_bn_bwd_fused_workspace_floats(Cc))


def bn_bwd_fused_sync_words():
    return int(_lib.load().yolo_bn_bwd_fused_sync_words())


def bn_act_bwd_fused(dout, out, relu, M, Cc, y, a1, mean, rstd, dgamma, dbeta, dy, workspace, sync, acc_dy=False, y2=None, a2=None,
                     mean2=None, rstd2=None, dgamma2=None, dbeta2=None, dy2=None, dres=None, acc_dres=False):
    """single-launch BN(+ReLU, + residual / second BN) backward; returns False (nothing launched) if the tensor is too large for it"""
    if isinstance(dgamma, (list, tuple)):          # grouped main BatchNorm: (split, [dgamma...], [dbeta...])
        split, dgs = dgamma
        sp = (C.c_int32 * len(split))(*split)
        rc = _lib.load().yolo_bn_act_bwd_fused_grouped(_p(dout), _p(out), int(relu), M, Cc, _p(y), _p(a1), _p(mean), _p(rstd), len(dgs), sp,
                                                       _ptr_array(dgs), _ptr_array(dbeta), _p(dy), int(acc_dy), _p(y2), _p(a2), _p(mean2),
                                                       _p(rstd2), _p(dgamma2), _p(dbeta2), _p(dy2), _p(dres), int(acc_dres), _p(workspace),
                                                       _p(sync), _stream())
    else:
        rc = _lib.load().yolo_bn_act_bwd_fused(_p(dout), _p(out), int(relu), M, Cc, _p(y), _p(a1), _p(mean), _p(rstd), _p(dgamma), _p(dbeta),
                                               _p(dy), int(acc_dy), _p(y2), _p(a2), _p(mean2), _p(rstd2), _p(dgamma2), _p(dbeta2), _p(dy2),
                                               _p(dres), int(acc_dres), _p(workspace), _p(sync), _stream())
    if rc == 1:
        return False
    check(rc, 'yolo_bn_act_bwd_fused')
    return True


def bn_fused_timeouts(sync):
    n = C.c_int(0)
    check(_lib.load().yolo_bn_fused_timeouts(_p(sync), C.byref(n)), 'yolo_bn_fused_timeouts')
    return n.value


def bn_fused_set_host_flag(sync, host_flag):
    """host_flag: a pinned int32 tensor (or None): set to 1 by the kernel when its grid barrier times out"""
    check(_lib.load().yolo_bn_fused_set_host_flag(_p(sync), _p(host_flag)), 'yolo_bn_fused_set_host_flag')


def bn_pool_bwd_reduce(dout, out, argmax, relu, y, mean, rstd, N, H, W, Cc, Ho, Wo, pt, pl, partial, gamma=None, beta=None):
    check(_lib.load().yolo_bn_pool_bwd_reduce(_p(dout), _p(out), _p(argmax), int(relu), _p(y), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                              N, H, W, Cc, Ho, Wo, pt, pl, _p(partial), _stream()), 'yolo_bn_pool_bwd_reduce')


def bn_pool_bwd_apply(dout, out, argmax, relu, y, a1, mean, rstd, k1, k2, dy, N, H, W, Cc, Ho, Wo, pt, pl):
    check(_lib.load().yolo_bn_pool_bwd_apply(_p(dout), _p(out), _p(argmax), int(relu), _p(y), _p(a1), _p(mean), _p(rstd), _p(k1),
                                             _p(k2), _p(dy), N, H, W, Cc, Ho, Wo, pt, pl, _stream()), 'yolo_bn_pool_bwd_apply')


def upcat_split_bwd(dcat, da, acc_a, db, acc_b, N, H, W, C0, C1):
    check(_lib.load().yolo_upcat_split_bwd(_p(dcat), _p(da), int(acc_a), _p(db), int(acc_b), N, H, W, C0, C1, _stream()),
          'yolo_upcat_split_bwd')


def mix_problem(N, H, W, Cc, split, ksize):
    p = MixProblem(N, H, W, Cc)
    for i, v in enumerate(split):
        p.split[i] = int(v)
    for i, v in enumerate(ksize):
        p.ksize[i] = int(v)
    return p


def dwconv_mix_fwd(p, x, w, y):
    check(_lib.load().yolo_dwconv_mix_fwd(C.byref(p), _p(x), _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(y), _stream()), 'yolo_dwconv_mix_fwd')


def dwconv_mix_dgrad(p, dy, w, dx, accumulate=False):
    check(_lib.load().yolo_dwconv_mix_dgrad(C.byref(p), _p(dy), _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(dx), int(accumulate), _stream()),
          'yolo_dwconv_mix_dgrad')


def dwconv_mix_wgrad_workspace_bytes(p):
    return int(_lib.load().yolo_dwconv_mix_wgrad_workspace_bytes(C.byref(p)))


def dwconv_mix_wgrad(p, x, dy, dw, workspace, accumulate=False):
    check(_lib.load().yolo_dwconv_mix_wgrad(C.byref(p), _p(x), _p(dy), _p(dw[0]), _p(dw[1]), _p(dw[2]), _p(dw[3]), _p(workspace),
                                            workspace.numel() * workspace.element_size(), int(accumulate), _stream()),
          'yolo_dwconv_mix_wgrad')


def reduce_partials(partial, P, row_stride, Cc, out):
    check(_lib.load().yolo_reduce_partials(_p(partial), P, row_stride, Cc, _p(out), _stream()), 'yolo_reduce_partials')


def bn_eval_scale_shift(gamma, beta, mm, mv, eps, scale, shift, Cc):
    check(_lib.load().yolo_bn_eval_scale_shift(_p(gamma), _p(beta), _p(mm), _p(mv), eps, _p(scale), _p(shift), Cc, _stream()),
          'yolo_bn_eval_scale_shift')


def pack_input(images, out, npix, cimg):
    check(_lib.load().yolo_pack_input(_p(images), _p(out), npix, cimg, _stream()), 'yolo_pack_input')


def make_loss_config(head_grid_sizes, class_num, anchor_boxes, iou_thresh, loss_weights, ldc, T,
                     rectified_coord_num=0, rectified_loss_weight=None, is_focal_loss=False, focal_alpha=0.25,
                     focal_gamma=2.0, is_tiou_recall=False, eps=1e-8, grad_scale16=1.0):
    c = LossConfig()
    lw = np.asarray(loss_weights, dtype=np.float32)
    if rectified_loss_weight is None:
        rectified_loss_weight = [0.01, 0.01, 0.01]
    for h in range(3):
        gh, gw = int(head_grid_sizes[h][0]), int(head_grid_sizes[h][1])
        c.H[h], c.W[h], c.B[h], c.ldc[h] = gh, gw, len(anchor_boxes[h]), int(ldc[h])
        if len(anchor_boxes[h]) > _lib.MAX_ANCHORS:
            raise ValueError('at most %d anchors per head' % _lib.MAX_ANCHORS)
        for b, (aw, ah) in enumerate(anchor_boxes[h]):
            c.anchor_w[h][b] = float(np.float32(aw) * np.float32(gw))   # yolov3_decoder.py:38-40, float32 product
            c.anchor_h[h][b] = float(np.float32(ah) * np.float32(gh))
        c.w_xy[h], c.w_wh[h], c.w_noobj[h], c.w_obj[h], c.w_cls[h] = [float(v) for v in lw[h]]
        c.w_rect[h] = float(rectified_loss_weight[h])
    c.L = 5 + class_num
    c.T = T
    c.iou_thresh = iou_thresh
    c.rectified_coord_num = int(rectified_coord_num)
    c.is_focal_loss = int(bool(is_focal_loss))
    c.focal_alpha, c.focal_gamma = focal_alpha, focal_gamma
    c.is_tiou_recall = int(bool(is_tiou_recall))
    c.eps = eps
    c.grad_scale16 = float(grad_scale16)
    return c


def loss_workspace_bytes(cfg, N):
    r = _lib.load().yolo_loss_workspace_bytes(C.byref(cfg), N)
    if r < 0:
        raise _lib.YoloNativeError('yolo_loss_workspace_bytes rejected the config')
    return r


def loss_fwd_bwd(cfg, N, batch_global, logits, labels, current_num, terms, total, workspace, dlogits=(None, None, None),
                 dlogits_bf16=(None, None, None), assign_out=None, resp_iou_out=None):
    check(_lib.load().yolo_loss_fwd_bwd(C.byref(cfg), N, batch_global, _p(logits[0]), _p(logits[1]), _p(logits[2]), _p(labels),
                                        _p(dlogits[0]), _p(dlogits[1]), _p(dlogits[2]), _p(dlogits_bf16[0]), _p(dlogits_bf16[1]),
                                        _p(dlogits_bf16[2]), _p(current_num), _p(terms), _p(total), _p(assign_out), _p(resp_iou_out),
                                        _p(workspace), _stream()), 'yolo_loss_fwd_bwd')


def decode_head(logits, N, H, W, B, L, ldc, anchors_grid, eps, decoded=None, boxes=None, score=None, cls_idx=None):
    check(_lib.load().yolo_decode_head(_p(logits), N, H, W, B, L, ldc, _p(anchors_grid), eps, _p(decoded), _p(boxes), _p(score), _p(cls_idx),
                                       _stream()), 'yolo_decode_head')


def filter_boxes(prediction, boxes, N, H, W, B, L, score_thresh, cap, counts, rows, index):
    check(_lib.load().yolo_filter_boxes(_p(prediction), _p(boxes), N, H, W, B, L, float(score_thresh), cap, _p(counts), _p(rows), _p(index),
                                        _stream()), 'yolo_filter_boxes')


def nms_max_candidates():
    return _lib.load().yolo_nms_max_candidates()


def nms_heads(rows, counts, N, cap, nms_thresh, fixed_indices, keep, status):
    check(_lib.load().yolo_nms_heads(_p(rows[0]), _p(rows[1]), _p(rows[2]), _p(counts[0]), _p(counts[1]), _p(counts[2]), N, cap,
                                     float(nms_thresh), int(fixed_indices), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(status), _stream()),
          'yolo_nms_heads')


def letterbox_workspace_bytes(N):
    r = _lib.load().yolo_letterbox_workspace_bytes(N)
    if r < 0:
        raise _lib.YoloNativeError('yolo_letterbox_workspace_bytes(%d) rejected' % N)
    return r


def letterbox_augment(src, desc, N, H, W, augment, workspace, out_f32=None, out_bf16x8=None):
    check(_lib.load().yolo_letterbox_augment(_p(src), _p(desc), N, H, W, int(augment), _p(workspace), _p(out_f32), _p(out_bf16x8), _stream()),
          'yolo_letterbox_augment')


def radam_schedule(sched, iterations, beta1, beta2, decay, warmup_coef):
    check(_lib.load().yolo_radam_schedule(_p(sched), _p(iterations), beta1, beta2, decay, warmup_coef, _stream()), 'yolo_radam_schedule')


def optimizer_schedule(sched, iterations, kind, beta1, beta2, decay):
    check(_lib.load().yolo_optimizer_schedule(_p(sched), _p(iterations), int(kind), beta1, beta2, decay, _stream()), 'yolo_optimizer_schedule')


def radam_l2_blocks(n):
    r = _lib.load().yolo_radam_l2_blocks(n)
    if r < 0:
        raise _lib.YoloNativeError('yolo_radam_l2_blocks(%d) rejected' % n)
    return r


def radam_l2_step(params, grads, m, v, l2_table, n, sched, beta1, beta2, eps, grad_scale=1.0, zero_grad=True, params_bf16=None,
                  vhat=None, l2_partial=None, nonfinite=None):
    check(_lib.load().yolo_radam_l2_step(_p(params), _p(grads), _p(m), _p(v), _p(vhat), _p(params_bf16), _p(l2_table), n, _p(sched),
                                         beta1, beta2, eps, grad_scale, int(zero_grad), _p(l2_partial), _p(nonfinite), _stream()), 'yolo_radam_l2_step')


def cast_f32_to_bf16(x, y, n):
    check(_lib.load().yolo_cast_f32_to_bf16(_p(x), _p(y), n, _stream()), 'yolo_cast_f32_to_bf16')


def sum_partials(partial, n, add, out, out_plain=None):
    check(_lib.load().yolo_sum_partials(_p(partial), n, _p(add), _p(out), _p(out_plain), _stream()), 'yolo_sum_partials')


# ---------------------------------------------------------------------------------------------------------------- launch sequencer
def stream_fork(from_stream, to_stream, local=False):
    """``to_stream`` waits for everything queued on ``from_stream`` so far (torch streams); recorded when a sequence is being recorded.
    ``local``: only kernels of this device wait behind the edge (yolo_seq_fork_local: no system-scope writeback at the event record)"""
    lib = _lib.load()
    fn, name = (lib.yolo_seq_fork_local, 'yolo_seq_fork_local') if local else (lib.yolo_seq_fork, 'yolo_seq_fork')
    check(fn(C.c_void_p(from_stream.cuda_stream), C.c_void_p(to_stream.cuda_stream)), name)


def seq_begin():
    r = _lib.load().yolo_seq_begin()
    if r < 0:
        check(r, 'yolo_seq_begin')
    return r


def seq_mark():
    return _lib.load().yolo_seq_mark()


def seq_end():
    r = _lib.load().yolo_seq_end()
    if r < 0:
        check(r, 'yolo_seq_end')
    return r


def seq_run(seq, begin, end):
    check(_lib.load().yolo_seq_run(seq, begin, end), 'yolo_seq_run')


def seq_free(seq):
    check(_lib.load().yolo_seq_free(seq), 'yolo_seq_free')
