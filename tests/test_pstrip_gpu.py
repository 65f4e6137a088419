"""GPU parity tests of the big-tile 3x3 / stride-1 convolution (csrc/conv_pstrip.hip: one 512-thread workgroup per tile of up to 352 pixels,
padded-coordinate strip, K split over the wave groups) against a float32 reference of the same op -- forward with BatchNorm statistics, data
gradient (plain, accumulating, external addend) and the data gradient with the BatchNorm-backward reduce in its epilogue.  The kernel is
FORCED through yolo_set_tuning('pstrip', 1 + variant); a shape the forced variant cannot take falls back to the other kernels, which the
statistics-row count gives away (asserted).  Tolerances as in test_kernels_gpu.py: bf16 operands are exact in the float32 reference, the
differences are float32 summation order and the bf16 rounding of the stored outputs (2^-8 relative)."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEFAULT_PSTRIP = 0        # the library's default for the 'pstrip' tuning


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
        ops.set_tuning('pstrip', v)
    yield force
    ops.set_tuning('pstrip', DEFAULT_PSTRIP)


def expected_rows(M, Cout, variant, H, W):
    """statistics rows of the forced variant (None: the variant does not take the shape) -- mirrors ps_plan_variant's tile stride"""
    bm, bn = ((352, 64), (176, 128), (384, 64), (192, 128))[variant]
    if Cout % bn:
        return None
    tn = Cout // bn
    ntm_min = -(-M // bm)
    rounds = -(-(ntm_min * tn) // 256)
    ntm = max(rounds * 256 // tn, ntm_min)
    even = -(-M // ntm)
    return [(-(-even // W)) * W, even, bm]


CASES = [
    # N, H, W, Cin, Cout
    (2, 13, 13, 128, 64),        # one tile that spans two images (separator line inside the strip)
    (32, 13, 13, 128, 128),      # 5408 pixels: tiles start in the middle of image rows and span 2-3 images
    (5, 26, 26, 128, 64),        # Wp = 32
    (3, 52, 52, 128, 128),       # Wp = 56, tiles of a few rows
    (2, 20, 12, 64, 64),         # one 64-channel slice (9 K steps), non-square
    (7, 9, 17, 256, 256),        # W + 1 not a multiple of 8, tiles that span several images
    (1, 40, 104, 128, 128),      # Wp = 112: the widest map of the benchmark model
    (2, 1, 1, 64, 64),           # degenerate map: everything but the centre tap is padding
    (3, 2, 5, 64, 128),
    (16, 26, 26, 256, 256),      # four slices, 10816 pixels
]


@pytest.mark.parametrize('variant', [0, 1, 2, 3])
@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_pstrip_fwd_dgrad(dev, forced, case, variant):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + variant)
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT())
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * (1.0 / math.sqrt(9 * Cin))).to(ACT())
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    xr = x.float().requires_grad_(True)
    wr = w.float()
    y_ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT())
    y_ref.backward(dy.float())
    M = N * H * W

    forced(1 + variant)
    rows = ops.conv2d_stat_rows(p)
    cands = expected_rows(M, Cout, variant, H, W)
    if cands is None or all(rows != -(-M // c) for c in cands):
        pytest.skip('the forced variant does not take this shape (rows %d)' % rows)
    xd = x.to(dev)
    w_fwd = w.permute(3, 0, 1, 2).contiguous().to(dev)
    y = torch.full((N, H, W, Cout), float('nan'), dtype=ACT(), device=dev)
    ssum = torch.zeros(rows, Cout, device=dev)
    ssq = torch.zeros(rows, Cout, device=dev)
    ops.conv2d_fwd(p, xd, w_fwd, y, stat_sum=ssum, stat_sq=ssq)
    torch.cuda.synchronize()
    yc = y.float().cpu()
    torch.testing.assert_close(yc, y_ref.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(ssum.sum(0).cpu(), yc.sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(ssq.sum(0).cpu(), (yc * yc).sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    # against the other kernels on the same input: same values up to summation order (rounding ties of the 16-bit store)
    forced(0)
    y_old = torch.empty_like(y)
    ops.conv2d_fwd(p, xd, w_fwd, y_old)
    torch.cuda.synchronize()
    torch.testing.assert_close(yc, y_old.float().cpu(), rtol=2 ** -7, atol=1e-3)
    forced(1 + variant)
    # run-to-run determinism
    y2 = torch.empty_like(y)
    ops.conv2d_fwd(p, xd, w_fwd, y2)
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))

    if Cin % 64 == 0 and Cin % (64 if variant in (0, 2) else 128) == 0:
        dyd = dy.to(dev)
        w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
        ops.repack_dgrad_weights(w_fwd, w_dg, Cout, 3, 3, Cin)
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dyd, w_dg, dx)
        torch.testing.assert_close(dx.float().cpu(), xr.grad, rtol=1e-2, atol=1e-2)
        ops.conv2d_dgrad(p, dyd, w_dg, dx, accumulate=True)
        torch.testing.assert_close(dx.float().cpu(), 2 * xr.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, variant, accumulate, relu, shortcut BN
    (8, 26, 26, 256, 128, 0, True, True, True),
    (6, 13, 13, 128, 128, 0, False, True, False),
    (2, 52, 52, 128, 128, 1, True, True, False),
    (9, 13, 13, 128, 256, 1, False, False, False),
    (8, 26, 26, 256, 128, 2, True, True, True),
    (3, 52, 52, 128, 128, 2, False, True, False),
    (5, 26, 26, 128, 256, 3, True, True, False),
])
def test_pstrip_dgrad_with_bn_reduce_epilogue(dev, forced, case):
    """as test_kernels_gpu.py::test_dgrad_with_bn_reduce_epilogue, on the big-tile kernel: the fused launch stores the masked gradient of the
    plain launch bit for bit, and its partial rows sum to the masked reduce of that gradient"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, variant, acc, relu, has2 = case
    g = torch.Generator().manual_seed(sum(case[:5]))
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

    forced(1 + variant)
    rows = ops.conv2d_dgrad_bn_rows(p)
    cands = expected_rows(M, Cin, variant, H, W)
    assert cands is not None and any(rows == -(-M // c) for c in cands), 'the forced variant must take this shape'
    plain = base.clone()
    ops.conv2d_dgrad(p, dy, w_dg, plain, accumulate=acc)
    partial = torch.zeros(rows, 3, Cin, device=dev)
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
    nq = 3 if has2 else 2
    got = partial.double().sum(0)[:nq]
    gd, yd = want.double(), y.double()
    scale = float(gd.abs().sum(0).max())
    torch.testing.assert_close(got[0], gd.sum(0), rtol=1e-5, atol=1e-6 * max(scale, 1.0))
    torch.testing.assert_close(got[1], (gd * ((yd - mean.double()) * rstd.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    if has2:
        torch.testing.assert_close(got[2], (gd * ((y2.double() - mean2.double()) * rstd2.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))


def test_pstrip_fp16(dev, fp16, forced):
    """the float16 build of the same kernel (libyolov3_amd_fp16.so)"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = 6, 26, 26, 128, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).half()
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * (1.0 / math.sqrt(9 * Cin))).half()
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    y_ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    forced(1)
    y = torch.empty(N, H, W, Cout, dtype=torch.float16, device=dev)
    ops.conv2d_fwd(p, x.to(dev), w.permute(3, 0, 1, 2).contiguous().to(dev), y)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.float().cpu(), y_ref, rtol=2e-3, atol=2e-3)
