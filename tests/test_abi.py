"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares, and the
ctypes table covers exactly that set (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'yolov3_amd.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(yolo_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_functions():
    names = declared_functions()
    assert 'yolo_conv2d_fwd' in names and 'yolo_loss_fwd_bwd' in names and 'yolo_radam_l2_step' in names
    assert len(names) >= 25


@pytest.mark.parametrize('which', ['bfloat16', 'float16'])
def test_library_exports_every_declared_symbol(which):
    """both builds of the sources (bf16 default, -DYOLO_FP16) export the whole C-ABI and report their element type"""
    from yolov3_tensorflow_amd import _lib
    path = _lib.LIB_PATH_FP16 if which == 'float16' else _lib.LIB_PATH
    if not os.path.exists(path):
        pytest.fail('%s is not built (run __graft_entry__.build())' % os.path.basename(path))
    lib = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(lib, name), 'missing export ' + name
    assert lib.yolo_abi_version() == 1
    assert lib.yolo_abi_dtype() == (1 if which == 'float16' else 0)
    assert _lib.load(which).yolo_abi_dtype() == (1 if which == 'float16' else 0)


def test_ctypes_table_matches_header():
    from yolov3_tensorflow_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_functions()


def test_argument_validation_without_gpu():
    """argument checks run on the host before any launch, so they are testable here"""
    from yolov3_tensorflow_amd import _lib
    lib = _lib.load()
    p = _lib.ConvProblem(1, 8, 8, 12, 0, 64, 3, 3, 1, 1, 1, 8, 8)      # Cin/8 not a power of two
    assert lib.yolo_conv2d_fwd(ctypes.byref(p), None, None, None, None, None, 0, None, None, None) == -1
    assert b'Cin' in lib.yolo_last_error()
    p = _lib.ConvProblem(1, 8, 8, 64, 0, 100, 3, 3, 1, 1, 1, 8, 8)     # Cout not padded
    assert lib.yolo_conv2d_fwd(ctypes.byref(p), None, None, None, None, None, 0, None, None, None) == -1
    assert lib.yolo_radam_l2_step(None, None, None, None, None, None, None, 256, None, 0.9, 0.999, 1e-8, 1.0, 1, None, None, None) == -1
    assert lib.yolo_reduce_rows(100, 24) == -1       # 256 % (C/8) != 0
    assert lib.yolo_reduce_rows(100, 64) > 0


def test_missing_library_fails_loudly(monkeypatch):
    from yolov3_tensorflow_amd import _lib
    monkeypatch.setattr(_lib, '_libs', {})
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libyolov3_amd.so')
    monkeypatch.setattr(_lib, 'LIB_PATH_FP16', '/nonexistent/libyolov3_amd_fp16.so')
    with pytest.raises(_lib.YoloNativeError):
        _lib.load()
    with pytest.raises(_lib.YoloNativeError):
        _lib.load('float16')


def test_kernel_plans_are_consistent_without_gpu():
    """host-side planning (no launches): the statistics rows of the convolution follow the kernel the tuning selects, workspaces are
    sized, unknown tuning names are rejected"""
    from yolov3_tensorflow_amd import ops
    N, H, W = 32, 52, 52
    p = ops.conv_problem(N, H, W, 128, 128, 3, 1, 'same')
    M = N * H * W
    try:
        # 3x3 / stride-1 layers below 80 columns: the 32x32x16 kernel's automatic rule (conv_s32.hip), one statistics row per pixel tile
        plan = ops.conv2d_fwd_plan(p)
        assert plan['family'] == 's32' and (plan['bm'], plan['bn']) == (256, 64) and plan['workgroups'] == (M + 255) // 256 * 2
        assert plan['lds_bytes'] <= 80 * 1024 and ops.conv2d_stat_rows(p) == (M + 255) // 256         # two workgroups per CU
        p26 = ops.conv_problem(N, 26, 26, 256, 256, 3, 1, 'same')
        assert ops.conv2d_fwd_plan(p26)['family'] == 'strip'          # measured equal in the step: stays on the strip kernel
        assert ops.conv2d_fwd_plan(ops.conv_problem(N, 26, 26, 256, 512, 3, 1, 'same'))['family'] == 's32'     # the wide stride-16 head convolution
        assert ops.conv2d_fwd_plan(ops.conv_problem(N, 13, 13, 512, 512, 3, 1, 'same'))['family'] == 'strip'
        ops.set_tuning('s32', 1)                            # forced 128 x 128 configuration
        assert ops.conv2d_fwd_plan(p)['bn'] == 128 and ops.conv2d_stat_rows(p) == (M + 127) // 128
        ops.set_tuning('s32', 0)
        assert ops.conv2d_fwd_plan(p)['family'] == 'strip'
        ops.set_tuning('s32', -1)
        ops.set_tuning('strip_bm', 0)                      # implicit-GEMM kernel: 128-pixel tiles for this shape
        assert ops.conv2d_fwd_plan(p)['family'] == 'igemm'
        assert ops.conv2d_stat_rows(p) == (M + 127) // 128
        for bm in (64, 128, 256):
            ops.set_tuning('strip_bm', bm)
            assert ops.conv2d_stat_rows(p) == (M + bm - 1) // bm
        ops.set_tuning('s32', 0)
        ops.set_tuning('strip_bm', -1)
        assert ops.conv2d_stat_rows(p) == (M + 127) // 128   # the strip kernel's measured choice for 52 x 52 maps
        ops.set_tuning('s32', -1)
        p1 = ops.conv_problem(N, H, W, 128, 128, 1, 1, 'same')
        assert ops.conv2d_stat_rows(p1) == (M + 127) // 128  # 1x1: never the strip kernel
        # the streaming kernel of the 64-channel layers: automatic for FORWARD launches with >= 512 pixels per workgroup, one statistics row
        # per workgroup; the data gradient (whose rows come from conv2d_dgrad_bn_rows) keeps the strip kernel unless forced
        p64 = ops.conv_problem(32, 104, 104, 64, 64, 3, 1, 'same')
        M64 = 32 * 104 * 104
        plan = ops.conv2d_fwd_plan(p64)
        assert plan['family'] == 'stream' and plan['tile_pixels'] == 1408 and plan['workgroups'] == 246 and plan['lds_bytes'] == 160 * 1024
        assert ops.conv2d_stat_rows(p64) == 246 and ops.conv2d_dgrad_bn_rows(p64) == (M64 + 255) // 256
        assert ops.conv2d_fwd_plan(ops.conv_problem(2, 104, 104, 64, 64, 3, 1, 'same'))['family'] == 'strip'       # 128 pixels per workgroup: no
        assert ops.conv2d_fwd_plan(ops.conv_problem(16, 152, 152, 64, 64, 3, 1, 'same'))['family'] == 'stream'     # 608 x 608 input
        assert ops.conv2d_fwd_plan(ops.conv_problem(16, 160, 191, 64, 64, 3, 1, 'same'))['family'] == 'stream'     # the widest map the ring reaches
        assert ops.conv2d_fwd_plan(ops.conv_problem(16, 160, 192, 64, 64, 3, 1, 'same'))['family'] == 'strip'
        ops.set_tuning('stream', 1)
        assert ops.conv2d_dgrad_bn_rows(p64) == 246
        ops.set_tuning('stream', 0)
        assert ops.conv2d_fwd_plan(p64)['family'] == 'strip' and ops.conv2d_stat_rows(p64) == (M64 + 255) // 256
        ops.set_tuning('stream', -1)
    finally:
        ops.set_tuning('strip_bm', -1)
        ops.set_tuning('s32', -1)
    ws = ops.conv2d_wgrad_workspace_bytes(p)
    assert ws % (128 * 9 * 128 * 4) == 0 and 2 <= ws // (128 * 9 * 128 * 4) <= 384     # whole slabs, one round of workgroups
    mp = ops.mix_problem(N, 104, 104, 64, [0, 32, 48, 56, 64], [3, 5, 7, 9])
    assert ops.dwconv_mix_wgrad_workspace_bytes(mp) > 0
    assert ops.bn_bwd_fused_workspace_floats(512) >= (256 * 3 + 3) * 512 and ops.bn_bwd_fused_sync_words() >= 17
    with pytest.raises(Exception):
        ops.set_tuning('no_such_knob', 1)
