"""GPU parity tests of the weights-in-registers streaming 3x3 / stride-1 convolution of the 64-channel layers (csrc/conv_stream.hip: one
512-thread workgroup per compute unit walks a contiguous pixel range through a ring of 896 LDS rows, the 64 x 576 weight tile lives in
registers) against a float32 reference of the same op -- forward with BatchNorm statistics, data gradient (plain, accumulating, external
addend) and the data gradient with the BatchNorm-backward reduce in its epilogue.  The kernel is FORCED through
yolo_set_tuning('stream', 1) (auto takes only forward launches with ranges of >= 512 pixels per workgroup); the plan's family is asserted.  Tolerances as in
test_kernels_gpu.py: bf16 operands are exact in the float32 reference, the differences are float32 summation order and the bf16 rounding
of the stored outputs (2^-8 relative)."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEFAULT_STREAM = -1       # the library's default for the 'stream' tuning


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def ACT():
    from yolov3_tensorflow_amd import backend
    return backend.torch_dtype()


@pytest.fixture
def fp16():
    from yolov3_tensorflow_amd import backend
    backend.set_compute_dtype('float16')
    yield
    backend.set_compute_dtype('bfloat16')


@pytest.fixture
def forced():
    from yolov3_tensorflow_amd import ops

    def force(v):
        ops.set_tuning('stream', v)
    yield force
    ops.set_tuning('stream', DEFAULT_STREAM)


CASES = [
    # N, H, W, Cout   (Cin = 64)
    (1, 3, 64, 64),          # the narrowest map the kernel takes: 192 pixels, 3 workgroups of one step
    (2, 104, 104, 64),       # the benchmark layer's map: 21632 pixels, ranges that start in the middle of image rows
    (3, 7, 100, 64),         # W + 1 not a multiple of 8; the last workgroup's range ends inside a step
    (2, 5, 152, 128),        # 608 x 608 input's map, two channel tiles
    (3, 4, 160, 64),         # 640 x 640 input's map
    (2, 6, 191, 64),         # the widest map the ring reaches: 512 + roundup8(W + 1) + W + 1 == 896 rows exactly
    (5, 1, 70, 64),          # one-row images: every tap row but the centre one is padding
    (1, 70, 65, 192),        # three channel tiles
    (9, 9, 81, 64),          # ranges that span several images
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_stream_fwd_dgrad(dev, forced, case):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cout = case
    Cin = 64
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT())
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * (1.0 / math.sqrt(9 * Cin))).to(ACT())
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    xr = x.float().requires_grad_(True)
    wr = w.float()
    y_ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT())
    y_ref.backward(dy.float())
    M = N * H * W

    forced(1)
    plan = ops.conv2d_fwd_plan(p)
    assert plan['family'] == 'stream' and plan['tile_pixels'] % 64 == 0
    rows = ops.conv2d_stat_rows(p)
    assert rows == -(-M // plan['tile_pixels']) and rows * (Cout // 64) == plan['workgroups']
    xd = x.to(dev)
    w_fwd = w.permute(3, 0, 1, 2).contiguous().to(dev)
    y = torch.full((N, H, W, Cout), float('nan'), dtype=ACT(), device=dev)
    ssum = torch.full((rows, Cout), float('nan'), device=dev)
    ssq = torch.full((rows, Cout), float('nan'), device=dev)
    ops.conv2d_fwd(p, xd, w_fwd, y, stat_sum=ssum, stat_sq=ssq)
    torch.cuda.synchronize()
    yc = y.float().cpu()
    torch.testing.assert_close(yc, y_ref.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(ssum.sum(0).cpu(), yc.sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(ssq.sum(0).cpu(), (yc * yc).sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    # against the other kernels on the same input: same values up to summation order (rounding ties of the 16-bit store)
    forced(0)
    assert ops.conv2d_fwd_plan(p)['family'] != 'stream'
    y_old = torch.empty_like(y)
    ops.conv2d_fwd(p, xd, w_fwd, y_old)
    torch.cuda.synchronize()
    torch.testing.assert_close(yc, y_old.float().cpu(), rtol=2 ** -7, atol=1e-3)
    forced(1)
    # run-to-run determinism
    y2 = torch.empty_like(y)
    ops.conv2d_fwd(p, xd, w_fwd, y2)
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))

    if Cout == 64:
        dyd = dy.to(dev)
        w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
        ops.repack_dgrad_weights(w_fwd, w_dg, Cout, 3, 3, Cin)
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dyd, w_dg, dx)
        torch.testing.assert_close(dx.float().cpu(), xr.grad, rtol=1e-2, atol=1e-2)
        ops.conv2d_dgrad(p, dyd, w_dg, dx, accumulate=True)
        torch.testing.assert_close(dx.float().cpu(), 2 * xr.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('case', [
    # N, H, W, accumulate, relu, shortcut BN      (Cin = Cout = 64)
    (2, 104, 104, True, True, False),
    (3, 7, 100, False, True, False),
    (2, 40, 80, True, True, True),
    (5, 9, 70, False, False, False),
    (4, 20, 65, False, True, True),
])
def test_stream_dgrad_with_bn_reduce_epilogue(dev, forced, case):
    """as test_kernels_gpu.py::test_dgrad_with_bn_reduce_epilogue, on the streaming kernel: the fused launch stores the masked gradient of the
    plain launch bit for bit, and its partial rows sum to the masked reduce of that gradient"""
    from yolov3_tensorflow_amd import ops
    N, H, W, acc, relu, has2 = case
    Cin = Cout = 64
    g = torch.Generator().manual_seed(sum(case[:3]))
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(ACT()).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT()).to(dev)
    base = torch.randn(N, H, W, Cin, generator=g).to(ACT()).to(dev)
    M = N * H * W
    y = (torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(ACT()).to(dev)
    y2 = torch.randn(M, Cin, generator=g).to(ACT()).to(dev) if has2 else None
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mean2, rstd2 = ((torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)) if has2 else (None, None)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev) if relu else None

    forced(1)
    assert ops.conv2d_fwd_plan(p)['family'] == 'stream'
    rows = ops.conv2d_dgrad_bn_rows(p)
    assert rows == ops.conv2d_stat_rows(p)
    plain = base.clone()
    ops.conv2d_dgrad(p, dy, w_dg, plain, accumulate=acc)
    partial = torch.full((rows, 3, Cin), float('nan'), device=dev)
    fused = base.clone()
    bn = dict(mask=mask, y=y, mean=mean, rstd=rstd, partial=partial)
    if has2:
        bn.update(y2=y2, mean2=mean2, rstd2=rstd2)
    ops.conv2d_dgrad(p, dy, w_dg, fused, accumulate=acc, bn=bn)
    torch.cuda.synchronize()
    want = plain.reshape(M, Cin // 8, 8).float()
    if relu:
        bits = ((mask.to(torch.int32).reshape(M, Cin // 8, 1) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).float()
        want = want * bits
    want = want.reshape(M, Cin)
    assert torch.equal(fused.reshape(M, Cin).float(), want)
    if acc:
        other = torch.full_like(base, float('nan'))
        ops.conv2d_dgrad(p, dy, w_dg, other, addend=base)
        torch.cuda.synchronize()
        assert torch.equal(other.view(torch.int16), plain.view(torch.int16))
    # against the tile kernel's fused launch: the same stored gradient
    forced(0)
    rows_old = ops.conv2d_dgrad_bn_rows(p)
    partial_old = torch.zeros(rows_old, 3, Cin, device=dev)
    fused_old = base.clone()
    bn_old = dict(bn, partial=partial_old)
    ops.conv2d_dgrad(p, dy, w_dg, fused_old, accumulate=acc, bn=bn_old)
    torch.cuda.synchronize()
    torch.testing.assert_close(fused.float(), fused_old.float(), rtol=2 ** -7, atol=1e-3)
    nq = 3 if has2 else 2
    got = partial.double().sum(0)[:nq]
    gd, yd = want.double(), y.double()
    scale = float(gd.abs().sum(0).max())
    torch.testing.assert_close(got[0], gd.sum(0), rtol=1e-5, atol=1e-6 * max(scale, 1.0))
    torch.testing.assert_close(got[1], (gd * ((yd - mean.double()) * rstd.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    if has2:
        torch.testing.assert_close(got[2], (gd * ((y2.double() - mean2.double()) * rstd2.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))


def test_stream_is_what_the_benchmark_layers_run(dev):
    """under the default tuning the 64 -> 64 channel layers of the 104 x 104 maps at batch 32 take the streaming kernel, one workgroup per
    compute unit; small batches keep the tile kernel"""
    from yolov3_tensorflow_amd import ops
    plan = ops.conv2d_fwd_plan(ops.conv_problem(32, 104, 104, 64, 64, 3, 1, 'same'))
    assert plan['family'] == 'stream' and plan['workgroups'] <= 256 and plan['tile_pixels'] == 1408
    assert ops.conv2d_fwd_plan(ops.conv_problem(2, 104, 104, 64, 64, 3, 1, 'same'))['family'] == 'strip'
    assert ops.conv2d_fwd_plan(ops.conv_problem(32, 52, 52, 128, 128, 3, 1, 'same'))['family'] != 'stream'


def test_stream_fp16(dev, fp16, forced):
    """the float16 build of the same kernel (libyolov3_amd_fp16.so)"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = 3, 30, 72, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).half()
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * (1.0 / math.sqrt(9 * Cin))).half()
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    y_ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    forced(1)
    assert ops.conv2d_fwd_plan(p)['family'] == 'stream'
    y = torch.empty(N, H, W, Cout, dtype=torch.float16, device=dev)
    ops.conv2d_fwd(p, x.to(dev), w.permute(3, 0, 1, 2).contiguous().to(dev), y)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.float().cpu(), y_ref, rtol=2e-3, atol=2e-3)
