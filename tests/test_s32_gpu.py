"""conv3x3_s32_kernel (conv_s32.hip: 3x3 / stride-1 convolution on v_mfma_f32_32x32x16 with 64 x 64 wave tiles, optional K split over wave
groups) through the C-ABI, every tile configuration forced with yolo_set_tuning("s32", 1 + id):
 * forward + BatchNorm statistics, plain / accumulating data gradient against the implicit-GEMM kernel on the same inputs (same bf16
   products, float32 sums in another order, one 16-bit rounding: within one ulp of the 16-bit type) and against a float32 torch reference;
 * the data gradient with the fused BatchNorm-backward reduce: masked gradient bit-identical to mask(plain result), tile sums against double.
Shapes put image boundaries, row wraps, maps narrower than a 32-pixel block, ragged last tiles and several 64-channel slices into the strips.
"""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CONFIGS = {0: (128, 128), 1: (256, 64), 2: (128, 64), 3: (64, 128), 4: (64, 64), 5: (256, 128), 6: (128, 128), 7: (256, 64), 8: (384, 128)}


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


def bf(x):
    return x.to(ACT())


@pytest.fixture(params=['bf16', 'fp16'])
def dtype(request):
    from yolov3_tensorflow_amd import backend
    if request.param == 'fp16':
        backend.set_compute_dtype('float16')
    yield request.param
    backend.set_compute_dtype('bfloat16')


SHAPES = [
    # N, H, W, Cin, Cout
    (3, 21, 19, 128, 256),      # image boundaries and row wraps inside every strip, ragged last tile, two slices
    (40, 13, 13, 64, 128),      # strips that span several images
    (2, 52, 52, 64, 128),
    (1, 104, 104, 64, 128),     # widest benchmark map
    (2, 5, 3, 256, 128),        # a map smaller than the halo, four slices
    (1, 1, 1, 64, 128),         # every tap but the centre is padding
]


@pytest.mark.parametrize('shape', SHAPES, ids=str)
@pytest.mark.parametrize('cfg', sorted(CONFIGS))
def test_s32_matches_implicit_gemm_and_reference(dev, dtype, cfg, shape):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = shape
    bm, bn = CONFIGS[cfg]
    g = torch.Generator().manual_seed(1000 + cfg)
    x = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / math.sqrt(9 * Cin))).to(dev)
    dy = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
    base = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)

    def run():
        rows = ops.conv2d_stat_rows(p)
        y = torch.empty(N, H, W, Cout, dtype=ACT(), device=dev)
        ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
        ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
        dx = torch.empty(N, H, W, Cin, dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dy, w_dg, dx)
        acc = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, acc, accumulate=True)
        other = torch.full_like(base, float('nan'))
        ops.conv2d_dgrad(p, dy, w_dg, other, addend=base)
        torch.cuda.synchronize()
        assert torch.equal(other.view(torch.int16), acc.view(torch.int16))      # fan-in read from its own buffer: the same bits
        return y.float().cpu(), ss.sum(0).cpu(), sq.sum(0).cpu(), dx.float().cpu(), acc.float().cpu(), rows

    try:
        ops.set_tuning('s32', 0)
        ops.set_tuning('strip_bm', 0)
        ops.set_tuning('stream', 0)
        ref = run()
        ops.set_tuning('s32', 1 + cfg)
        plan = ops.conv2d_fwd_plan(p)
        assert plan['family'] == 's32' and (plan['bm'], plan['bn']) == (bm, bn)
        got = run()
        assert got[5] == (N * H * W + bm - 1) // bm
    finally:
        ops.set_tuning('s32', -1)
        ops.set_tuning('strip_bm', -1)
        ops.set_tuning('stream', -1)
    torch.testing.assert_close(got[0], ref[0], rtol=2 ** -7, atol=1e-3)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-3, atol=0.5)
    torch.testing.assert_close(got[2], ref[2], rtol=1e-3, atol=0.5)
    torch.testing.assert_close(got[3], ref[3], rtol=2 ** -7, atol=2e-3)
    # (base + gradient can cancel: two sums that differ by one 16-bit ulp of the GRADIENT term, |dgrad| <= 4, leave results one such ulp apart)
    torch.testing.assert_close(got[4], ref[4], rtol=2 ** -6, atol=2 ** -6)
    # float32 reference (TF 'same' padding of a 3x3 / stride-1 convolution is 1 on every side)
    y_ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(got[0], y_ref, rtol=1e-2, atol=1e-2)
    # statistics are sums of the stored (rounded) outputs
    torch.testing.assert_close(got[1], got[0].sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(got[2], (got[0] * got[0]).sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, accumulate, relu, shortcut BN
    (2, 52, 52, 128, 128, False, True, False),
    (4, 26, 26, 128, 256, True, True, True),
    (3, 13, 13, 256, 128, True, True, False),
    (2, 21, 19, 128, 128, False, False, False),
], ids=str)
@pytest.mark.parametrize('cfg', sorted(CONFIGS))
def test_s32_dgrad_with_bn_reduce_epilogue(dev, cfg, case):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, acc, relu, has2 = case
    g = torch.Generator().manual_seed(sum(case[:5]) + cfg)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
    base = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    M = N * H * W
    y = bf(torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(dev)
    y2 = bf(torch.randn(M, Cin, generator=g)).to(dev) if has2 else None
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mean2, rstd2 = ((torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)) if has2 else (None, None)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev) if relu else None
    try:
        ops.set_tuning('s32', 1 + cfg)
        ops.set_tuning('stream', 0)
        assert ops.conv2d_fwd_plan(ops.conv_problem(N, H, W, Cout, Cin, 3, 1, 'same'))['family'] == 's32'
        plain = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, plain, accumulate=acc)
        rows = ops.conv2d_dgrad_bn_rows(p)
        assert rows == (M + CONFIGS[cfg][0] - 1) // CONFIGS[cfg][0]
        partial = torch.zeros(rows, 3, Cin, device=dev)
        fused = base.clone()
        bn = dict(mask=mask, y=y, mean=mean, rstd=rstd, partial=partial)
        if has2:
            bn.update(y2=y2, mean2=mean2, rstd2=rstd2)
        ops.conv2d_dgrad(p, dy, w_dg, fused, accumulate=acc, bn=bn)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning('s32', -1)
        ops.set_tuning('stream', -1)
    want = plain.reshape(M, Cin // 8, 8).float()
    if relu:
        bits = ((mask.to(torch.int32).reshape(M, Cin // 8, 1) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).float()
        want = want * bits
    want = want.reshape(M, Cin)
    assert torch.equal(fused.reshape(M, Cin).float(), want)
    nq = 3 if has2 else 2
    got = partial.double().sum(0)[:nq]
    gd, yd = want.double(), y.double()
    scale = float(gd.abs().sum(0).max())
    torch.testing.assert_close(got[0], gd.sum(0), rtol=1e-5, atol=1e-6 * max(scale, 1.0))
    torch.testing.assert_close(got[1], (gd * ((yd - mean.double()) * rstd.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    if has2:
        torch.testing.assert_close(got[2], (gd * ((y2.double() - mean2.double()) * rstd2.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))


@pytest.mark.parametrize('shape', [
    # N, H, W (input of the stride-2 convolution = dX grid), Cin, Cout
    (2, 104, 104, 64, 128),     # conv2d_6 of the benchmark model (one 64-channel block per class)
    (3, 52, 52, 128, 256),      # conv2d_11: two blocks per class, four slices
    (5, 26, 26, 256, 512),      # conv2d_16: the 128 x 64 K-split configuration
    (3, 12, 20, 64, 128),       # small map: image boundaries and row wraps inside every strip
    (2, 80, 80, 64, 128), (2, 40, 40, 128, 256), (2, 20, 20, 256, 512),      # BASELINE.json configs[0] (320 x 320, batch 2)
    (1, 2, 2, 64, 64),          # one pixel per class: every tap but (0, 0) is padding
], ids=str)
def test_s32_stride2_classes_match_implicit_gemm(dev, dtype, shape):
    """the stride-2 data gradient's four parity classes on conv3x3_s32_kernel (MODE 1: "s32_s2" = 1, the default) against the implicit-GEMM
    class kernel ("s32_s2" = 0) and a float32 reference: plain, accumulating, onto the even / even class only (accumulate = 2), and with the
    fused BatchNorm-backward reduce (masked gradient bit-identical to mask(plain), tile sums against double)"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 2, 'same')
    assert ops.conv2d_dgrad_classed(p)
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / math.sqrt(9 * Cout / 4))).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = bf(torch.randn(N, p.Ho, p.Wo, Cout, generator=g)).to(dev)
    base = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    M = N * H * W
    y = bf(torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(dev)
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)

    def run(expect_s32):
        rows = ops.conv2d_dgrad_bn_rows(p)
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dy, w_dg, dx)
        acc = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, acc, accumulate=True)
        ee = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, ee, accumulate=2)
        part = torch.zeros(rows, 3, Cin, device=dev)
        fused = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, fused, accumulate=2, bn=dict(mask=mask, y=y, mean=mean, rstd=rstd, partial=part))
        torch.cuda.synchronize()
        return dx.float().cpu(), acc.float().cpu(), ee.float().cpu(), fused.float().cpu(), part.double().sum(0).cpu(), rows

    try:
        ops.set_tuning('s32_s2', 0)
        ref = run(False)
        ops.set_tuning('s32_s2', 1)
        got = run(True)
    finally:
        ops.set_tuning('s32_s2', 1)
    bm = 256 if N * p.Ho * p.Wo >= 16384 else 128
    assert got[5] == 4 * ((N * p.Ho * p.Wo + bm - 1) // bm)          # (the s32 plan took the problem: one row per class and pixel tile)
    torch.testing.assert_close(got[0], ref[0], rtol=2 ** -7, atol=2e-3)
    torch.testing.assert_close(got[1], ref[1], rtol=2 ** -6, atol=2 ** -6)
    torch.testing.assert_close(got[2], ref[2], rtol=2 ** -6, atol=2 ** -6)
    # accumulate = 2: the even / even class adds onto the buffer, the other three overwrite it
    ee = got[2].clone()
    assert torch.equal(ee[:, 1::2], got[0][:, 1::2]) and torch.equal(ee[:, ::2, 1::2], got[0][:, ::2, 1::2])
    # float32 reference: conv_transpose of the stride-2 convolution (TF SAME on an even map pads bottom / right only)
    x_ref = torch.zeros(N, Cin, H + 1, W + 1, requires_grad=True)
    y_ref = F.conv2d(x_ref, w.float().cpu().permute(0, 3, 1, 2), stride=2)
    assert y_ref.shape[2:] == (p.Ho, p.Wo)
    y_ref.backward(dy.float().cpu().permute(0, 3, 1, 2))
    torch.testing.assert_close(got[0], x_ref.grad[:, :, :H, :W].permute(0, 2, 3, 1), rtol=1e-2, atol=1e-2)
    # fused reduce: the masked gradient is mask(accumulate = 2 result) bit for bit; sums against double
    want = got[2].reshape(M, Cin // 8, 8)
    bits = ((mask.cpu().to(torch.int32).reshape(M, Cin // 8, 1) >> torch.arange(8, dtype=torch.int32)) & 1).float()
    want = (want * bits).reshape(M, Cin)
    assert torch.equal(got[3].reshape(M, Cin), want)
    gd, yd = want.double(), y.double().cpu()
    scale = max(float(gd.abs().sum(0).max()), 1.0)
    torch.testing.assert_close(got[4][0], gd.sum(0), rtol=1e-5, atol=1e-6 * scale)
    torch.testing.assert_close(got[4][1], (gd * ((yd - mean.double().cpu()) * rstd.double().cpu())).sum(0), rtol=1e-4, atol=1e-5 * scale)
