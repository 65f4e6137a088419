"""GPU parity tests of the individual HIP kernels (through the C-ABI) against float32 CPU references on the SAME inputs.

Floating-point tolerances (stated per test): bf16 operands are exactly representable in the float32 reference, so the only
differences are accumulation order (float32) and the final bf16 rounding of outputs (<= 2^-8 relative).
"""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd import _lib
    _lib.load()          # fail loudly if the native library is missing
    return torch.device('cuda:0')


def ACT():
    """the 16-bit element type of the library in use (bfloat16, or float16 under the fp16 fixture)"""
    from yolov3_tensorflow_amd import backend
    return backend.torch_dtype()


def bf(x):
    return x.to(ACT())


@pytest.fixture
def fp16():
    from yolov3_tensorflow_amd import backend
    backend.set_compute_dtype('float16')
    yield
    backend.set_compute_dtype('bfloat16')


def ref_conv(x, w_hwio, stride, pt, pl, Ho, Wo):
    """float32 conv of NHWC x with HWIO w and explicit top/left padding (bottom/right as needed)"""
    N, H, W, Cc = x.shape
    k = w_hwio.shape[0]
    pb = max((Ho - 1) * stride + k - H - pt, 0)
    pr = max((Wo - 1) * stride + k - W - pl, 0)
    xin = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xin, w_hwio.permute(3, 2, 0, 1), stride=stride)[:, :, :Ho, :Wo].permute(0, 2, 3, 1)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, padding
    (2, 13, 13, 64, 64, 3, 1, 'same'),
    (3, 12, 20, 64, 128, 3, 2, 'same'),     # TF same s2 on even sizes: pad (0,1)
    (2, 11, 9, 128, 64, 3, 2, 'same'),      # odd sizes: pad (1,1)
    (2, 10, 10, 256, 128, 1, 1, 'same'),
    (2, 10, 14, 64, 128, 1, 2, 'valid'),    # NIN shortcut
    (2, 16, 16, 8, 64, 3, 2, 'same'),       # stem (RGB padded to 8 channels), K = 72 not a multiple of 64
    (2, 64, 96, 8, 64, 3, 2, 'same'),       # stem on its row-walking kernel (stem.hip: Wo % 16 == 0), one output row per workgroup
    (6, 192, 64, 8, 64, 3, 2, 'same'),      # ... two rows per workgroup (N * Ho > 512): the 3-slot input ring rotates, bottom pad row in the last group
    (1, 32, 608, 8, 64, 3, 2, 'same'),      # ... BASELINE.json configs[4]'s width: > 64 KiB of dynamic LDS
    (1, 7, 7, 512, 256, 3, 1, 'same'),
    (2, 1, 1, 64, 64, 3, 1, 'same'),        # degenerate maps: every tap but the centre is padding
    (3, 2, 5, 64, 128, 3, 1, 'same'),
    (2, 3, 2, 64, 64, 3, 2, 'same'),
    # large enough for the LDS-resident strip kernel (3x3 / stride 1): 128- and 256-pixel tiles, 64 / 128 wide, strips that span several
    # images (13 x 13), two 64-channel slices, non-square maps
    (128, 13, 13, 64, 512, 3, 1, 'same'),
    (5, 100, 100, 64, 64, 3, 1, 'same'),
    (4, 160, 160, 64, 64, 3, 1, 'same'),
    (3, 96, 96, 128, 512, 3, 1, 'same'),
    (40, 52, 20, 64, 256, 3, 1, 'same'),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(dev, case):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, k, s, padding = case
    g = torch.Generator().manual_seed(1234)
    x = bf(torch.randn(N, H, W, Cin, generator=g))
    if Cin == 8:
        x[..., 3:] = 0
    w = bf(torch.randn(k, k, Cin, Cout, generator=g) * (1.0 / math.sqrt(k * k * Cin)))
    p = ops.conv_problem(N, H, W, Cin, Cout, k, s, padding)
    Ho, Wo = p.Ho, p.Wo
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    y_ref = ref_conv(xr, wr, s, p.pad_t, p.pad_l, Ho, Wo)
    dy = bf(torch.randn(N, Ho, Wo, Cout, generator=g))
    y_ref.backward(dy.float())

    xd = x.to(dev)
    w_fwd = w.permute(3, 0, 1, 2).contiguous().to(dev)          # [Cout][R][S][Cin]
    y = torch.empty(N, Ho, Wo, Cout, dtype=ACT(), device=dev)
    rows = ops.conv2d_stat_rows(p)
    ssum = torch.zeros(rows, Cout, device=dev)
    ssq = torch.zeros(rows, Cout, device=dev)
    ops.conv2d_fwd(p, xd, w_fwd, y, stat_sum=ssum, stat_sq=ssq)
    torch.cuda.synchronize()
    yc = y.float().cpu()
    # tolerance: bf16 output rounding (2^-8 relative) + float32 accumulation-order noise
    torch.testing.assert_close(yc, y_ref.detach(), rtol=1e-2, atol=1e-2)
    # BN partial statistics are sums of the stored (rounded) outputs
    torch.testing.assert_close(ssum.sum(0).cpu(), yc.sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(ssq.sum(0).cpu(), (yc * yc).sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-2)

    # float32 output + bias
    bias = torch.randn(Cout, generator=g)
    y32 = torch.empty(N, Ho, Wo, Cout, dtype=torch.float32, device=dev)
    ops.conv2d_fwd(p, xd, w_fwd, y32, bias=bias.to(dev))
    torch.testing.assert_close(y32.cpu(), y_ref.detach() + bias, rtol=1e-4, atol=1e-4)

    # wgrad (float32 atomics into a zeroed buffer, [Cout][R][S][Cin])
    dyd = dy.to(dev)
    dw = torch.zeros(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(p, xd, dyd, dw)
    dw_ref = wr.grad.permute(3, 0, 1, 2)
    scale = dw_ref.abs().max().item()
    torch.testing.assert_close(dw.cpu(), dw_ref, rtol=1e-3, atol=1e-4 * max(scale, 1.0))
    # explicit split-K must agree
    dw2 = torch.zeros_like(dw)
    ops.conv2d_wgrad(p, xd, dyd, dw2, split_k=3)
    torch.testing.assert_close(dw2.cpu(), dw_ref, rtol=1e-3, atol=1e-4 * max(scale, 1.0))
    # two-phase path (the one training uses): split slabs in a workspace + summing pass, overwrites dw; accumulate adds
    ws = torch.empty(max(ops.conv2d_wgrad_workspace_bytes(p), 16) // 4, device=dev)
    dw3 = torch.full_like(dw, 7.0)
    ops.conv2d_wgrad_reduce(p, xd, dyd, dw3, ws)
    torch.testing.assert_close(dw3.cpu(), dw_ref, rtol=1e-3, atol=1e-4 * max(scale, 1.0))
    ops.conv2d_wgrad_reduce(p, xd, dyd, dw3, ws, accumulate=True)
    torch.testing.assert_close(dw3.cpu(), 2 * dw_ref, rtol=1e-3, atol=2e-4 * max(scale, 1.0))
    dw4 = torch.empty_like(dw)
    ops.conv2d_wgrad_reduce(p, xd, dyd, dw4, ws)
    ops.conv2d_wgrad_reduce(p, xd, dyd, dw3, ws)
    assert torch.equal(dw3, dw4), 'two-phase weight gradient must be run-to-run deterministic'

    if Cin % 64 == 0:
        w_dg = torch.empty(Cin, k, k, Cout, dtype=ACT(), device=dev)
        ops.repack_dgrad_weights(w_fwd, w_dg, Cout, k, k, Cin)
        ref_dg = w.permute(2, 0, 1, 3).flip(1, 2).contiguous()
        assert torch.equal(w_dg.cpu(), ref_dg)
        dx = torch.empty(N, H, W, Cin, dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dyd, w_dg, dx)
        torch.testing.assert_close(dx.float().cpu(), xr.grad, rtol=1e-2, atol=1e-2)
        ops.conv2d_dgrad(p, dyd, w_dg, dx, accumulate=True)        # fan-in accumulation: dx += dgrad
        torch.testing.assert_close(dx.float().cpu(), 2 * xr.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('C,M,res', [(64, 5000, False), (128, 2704 * 4, True), (512, 700, True)])
def test_relu_byte_mask_equals_reading_the_activation(dev, C, M, res):
    """bn_act_fwd(mask=...) leaves the activation's sign bits, one byte per 8-channel chunk; every BatchNorm-backward entry point given that mask
    (relu = 2) produces bit-identical results to the same call reading the 16-bit activation itself (relu = 1)"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(C + M)
    y = bf(torch.randn(M, C, generator=g)).to(dev)
    r = bf(torch.randn(M, C, generator=g)).to(dev) if res else None
    scale, shift = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    out = torch.empty(M, C, dtype=ACT(), device=dev)
    out2 = torch.empty_like(out)
    mask = torch.zeros(M * C // 8, dtype=torch.uint8, device=dev)
    ops.bn_act_fwd(y, scale, shift, out, M, C, True, res=r, mask=mask)
    ops.bn_act_fwd(y, scale, shift, out2, M, C, True, res=r)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    bits = (out.float() > 0).reshape(M, C // 8, 8).to(torch.int32)
    want = (bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(-1).to(torch.uint8).reshape(-1)
    assert torch.equal(mask, want)
    dout = bf(torch.randn(M, C, generator=g)).to(dev)
    mean, rstd = (torch.randn(C, generator=g) * 0.1).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    P = ops.reduce_rows(M, C)
    res_ = []
    for sign, code in ((out, 1), (mask, 2)):
        part = torch.zeros(P, 3, C, device=dev)
        ops.bn_act_bwd_reduce(dout, sign, code, y, mean, rstd, M, C, part)
        k1, k2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.bn_bwd_finalize(part.view(-1), P, C, 1, M, dg, db, k1, k2, row_stride=3 * C, q_stride=C)
        dy = torch.empty(M, C, dtype=ACT(), device=dev)
        dres = torch.empty(M, C, dtype=ACT(), device=dev) if res else None
        ops.bn_act_bwd_apply(dout, sign, code, M, C, y=y, a1=scale, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy, dres=dres)
        ws = torch.zeros(ops.bn_bwd_fused_workspace_floats(C), device=dev)
        sync = torch.zeros(ops.bn_bwd_fused_sync_words(), dtype=torch.int32, device=dev)
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dy2 = torch.empty(M, C, dtype=ACT(), device=dev)
        dres2 = torch.empty(M, C, dtype=ACT(), device=dev) if res else None
        ops.set_tuning('bn_fused_min_chunks', 1)
        try:
            fused = ops.bn_act_bwd_fused(dout, sign, code, M, C, y, scale, mean, rstd, dg2, db2, dy2, ws, sync, dres=dres2)
        finally:
            ops.set_tuning('bn_fused_min_chunks', 3)
        torch.cuda.synchronize()
        assert fused
        res_.append([t.clone() for t in (part, dg, db, dy, dy2, dg2, db2) + ((dres, dres2) if res else ())])
    for a_, b_ in zip(*res_):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, k, stride, accumulate, relu, shortcut BN      (tile the data-gradient kernel picks)
    (2, 104, 104, 64, 64, 3, 1, True, True, False),      # strip 256 x 64
    (4, 52, 52, 128, 128, 3, 1, False, True, False),     # strip 128-pixel tiles
    (4, 26, 26, 256, 256, 3, 1, True, True, True),       # strip, shortcut-BN quantity
    (3, 13, 13, 512, 256, 3, 1, True, True, False),      # strip 64-pixel tiles, ragged last tile
    (3, 13, 13, 512, 1024, 1, 1, False, False, False),   # 1x1 (implicit GEMM), linear unit (no mask)
    (2, 52, 52, 64, 128, 3, 2, True, True, False),       # stride 2: four parity classes
    (2, 26, 30, 128, 256, 1, 2, False, True, True),      # 1x1 stride 2 (strided gather)
    (1, 27, 27, 64, 64, 3, 2, True, True, False),        # stride 2 on an odd map: classes of different sizes
])
def test_dgrad_with_bn_reduce_epilogue(dev, case):
    """conv2d_dgrad(bn=...) = the plain data gradient followed by the unit's masked reduce: dx holds the masked gradient bit for bit and
    the tile sums add up to what bn_act_bwd_reduce computes from the plain result (float32 sums in a different order)"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, k, s, acc, relu, has2 = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    p = ops.conv_problem(N, H, W, Cin, Cout, k, s, 'same')
    w = bf(torch.randn(Cout, k, k, Cin, generator=g) * 0.05).to(dev)
    w_dg = torch.empty(Cin, k, k, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, k, k, Cin)
    dy = bf(torch.randn(N, p.Ho, p.Wo, Cout, generator=g)).to(dev)
    base = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    M = N * H * W
    y = bf(torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(dev)
    y2 = bf(torch.randn(M, Cin, generator=g)).to(dev) if has2 else None
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mean2, rstd2 = ((torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)) if has2 else (None, None)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev) if relu else None

    plain = base.clone()
    ops.conv2d_dgrad(p, dy, w_dg, plain, accumulate=acc)
    rows = ops.conv2d_dgrad_bn_rows(p)
    assert rows > 0
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
    if acc:   # the fan-in contribution read from its own buffer instead of dx: same bits, dx need not be initialised
        for kw in (dict(), dict(bn=dict(bn, partial=torch.zeros_like(partial)))):
            other = torch.full_like(base, float('nan'))
            ops.conv2d_dgrad(p, dy, w_dg, other, addend=base, **kw)
            torch.cuda.synchronize()
            assert torch.equal(other.view(torch.int16), (fused if kw else plain).view(torch.int16))
            if kw:
                assert torch.equal(kw['bn']['partial'], partial)
    # reference sums from the existing reduce kernel on the plain result
    P = ops.reduce_rows(M, Cin)
    ref = torch.zeros(P, 3, Cin, device=dev)
    ops.bn_act_bwd_reduce(plain, mask, 2 if relu else 0, y, mean, rstd, M, Cin, ref, y2=y2, mean2=mean2, rstd2=rstd2)
    torch.cuda.synchronize()
    nq = 3 if has2 else 2
    got, exp = partial.double().sum(0)[:nq], ref.double().sum(0)[:nq]
    scale = float(exp.abs().max())
    torch.testing.assert_close(got, exp, rtol=1e-4, atol=1e-5 * max(scale, 1.0))
    # and in double from the masked gradient itself
    gd, yd = want.double(), y.double()
    torch.testing.assert_close(got[0], gd.sum(0), rtol=1e-5, atol=1e-5 * max(scale, 1.0))
    torch.testing.assert_close(got[1], (gd * ((yd - mean.double()) * rstd.double())).sum(0), rtol=1e-4, atol=1e-5 * max(scale, 1.0))


def test_stem_kernel_matches_implicit_gemm(dev):
    """the row-walking stem kernel against the implicit-GEMM kernel on the same input (yolo_set_tuning 'stem_direct'): same values up to the
    float32 accumulation order (K is laid out tap-major with 4-channel taps there, 8-channel taps here), statistics rows sum to the same totals"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(9)
    N, H, W = 7, 224, 160
    x = bf(torch.rand(N, H, W, 8, generator=g))
    x[..., 3:] = 0
    w = bf(torch.randn(64, 3, 3, 8, generator=g) * 0.2)
    w[..., 3:] = 0
    p = ops.conv_problem(N, H, W, 8, 64, 3, 2, 'same')
    outs = []
    try:
        for direct in (1, 0):
            ops.set_tuning('stem_direct', direct)
            rows = ops.conv2d_stat_rows(p)
            y = torch.empty(N, p.Ho, p.Wo, 64, dtype=ACT(), device=dev)
            ss, sq = torch.zeros(rows, 64, device=dev), torch.zeros(rows, 64, device=dev)
            ops.conv2d_fwd(p, x.to(dev), w.to(dev), y, stat_sum=ss, stat_sq=sq)
            torch.cuda.synchronize()
            outs.append((rows, y.float().cpu(), ss.sum(0).cpu(), sq.sum(0).cpu()))
    finally:
        ops.set_tuning('stem_direct', 1)
    (r1, y1, s1, q1), (r0, y0, s0, q0) = outs
    assert r1 == N * ((p.Ho + 1) // 2) and r1 < r0              # 7 * 112 = 784 > 512 output rows: two per workgroup, one statistics row each
    torch.testing.assert_close(y1, y0, rtol=2 ** -7, atol=1e-3)
    assert float((y1 != y0).float().mean()) < 0.02               # only rounding ties of the bf16 store differ
    torch.testing.assert_close(s1, s0, rtol=1e-3, atol=5e-2)
    torch.testing.assert_close(q1, q0, rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize('bm,bn', [(64, 64), (64, 128), (128, 64), (128, 128), (256, 64), (256, 128)])
def test_strip_conv_variants_match_implicit_gemm(dev, bm, bn):
    """every tile variant of the LDS-resident strip kernel (3x3 / stride 1) against the implicit-GEMM kernel on the same inputs: forward with
    BatchNorm partial statistics, data gradient with fan-in accumulation; 3 images of 21 x 19 put image boundaries, row wraps and the
    ragged last tile inside the strips, 128 input channels = two 64-channel slices"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(99)
    N, H, W, Cin, Cout = 3, 21, 19, 128, 256
    x = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * 0.03).to(dev)
    dy = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
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
        ops.conv2d_dgrad(p, dy, w_dg, dx, accumulate=True)
        torch.cuda.synchronize()
        return y.float().cpu(), ss.sum(0).cpu(), sq.sum(0).cpu(), dx.float().cpu()

    try:
        ops.set_tuning('strip_bm', 0)
        ref = run()
        ops.set_tuning('strip_bm', bm)
        ops.set_tuning('strip_bn', bn)
        assert ops.conv2d_stat_rows(p) == (N * H * W + bm - 1) // bm
        got = run()
    finally:
        ops.set_tuning('strip_bm', -1)
        ops.set_tuning('strip_bn', 0)
    # same products, different float32 summation order, one bf16 rounding: at most 1 bf16 ulp apart
    torch.testing.assert_close(got[0], ref[0], rtol=2 ** -7, atol=1e-3)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-3, atol=0.5)
    torch.testing.assert_close(got[2], ref[2], rtol=1e-3, atol=0.5)
    torch.testing.assert_close(got[3], ref[3], rtol=2 ** -6, atol=2e-3)


@pytest.mark.parametrize('ring', [2, 3, 'pipe'])
@pytest.mark.parametrize('shape', [(3, 21, 19, 128, 256), (70, 13, 13, 64, 64), (2, 40, 104, 64, 128), (1, 5, 3, 64, 64), (8, 13, 13, 128, 128)])
def test_wgrad_strip_matches_generic(dev, shape, ring):
    """the kernel-row strip weight gradient (3x3 / stride 1) against the generic im2col one: image boundaries and row wraps inside the
    64-pixel stages (13 x 13), maps wider than a stage (W = 104), a map smaller than the halo (5 x 3), two 64-channel slices, both
    output-channel tiles, atomics and two-phase modes"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = shape
    x = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    dy = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')

    def run():
        dw = torch.zeros(Cout, 3, 3, Cin, device=dev)
        ops.conv2d_wgrad(p, x, dy, dw)
        ws = torch.empty(max(ops.conv2d_wgrad_workspace_bytes(p), 16) // 4, device=dev)
        dw2 = torch.empty_like(dw)
        ops.conv2d_wgrad_reduce(p, x, dy, dw2, ws)
        torch.cuda.synchronize()
        return dw.cpu(), dw2.cpu()

    try:
        ops.set_tuning('wgrad_pipe', 1 if ring == 'pipe' else 0)          # software-pipelined stage body (double-buffered) or the plain loop
        ops.set_tuning('wgrad_ring', 2 if ring == 'pipe' else ring)
        ops.set_tuning('wgrad_strip', 0)
        ref = run()
        ops.set_tuning('wgrad_strip', 1)
        got = run()
    finally:
        ops.set_tuning('wgrad_strip', 1)
        ops.set_tuning('wgrad_ring', 2)
        ops.set_tuning('wgrad_pipe', 1)
    scale = ref[1].abs().max().item()
    for t in got:      # same bf16 products, float32 sums in a different order
        torch.testing.assert_close(t, ref[1], rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize('shape', [(3, 12, 20, 64, 128), (2, 11, 9, 128, 64), (2, 26, 26, 128, 256)])
def test_stride2_dgrad_parity_classes_match_strided_gather(dev, shape):
    """stride-2 data gradient: the four-parity-class launch (default) against the strided den = 2 gather (fallback path, still used for
    kernel sizes other than 3) on even and odd maps, overwrite and fan-in accumulate"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(17)
    N, H, W, Cin, Cout = shape
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 2, 'same')
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(dev)
    dy = bf(torch.randn(N, p.Ho, p.Wo, Cout, generator=g)).to(dev)
    base = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)

    def run():
        dx = torch.empty(N, H, W, Cin, dtype=ACT(), device=dev)
        ops.conv2d_dgrad(p, dy, w_dg, dx)
        acc = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, acc, accumulate=True)
        torch.cuda.synchronize()
        return dx.float().cpu(), acc.float().cpu()

    try:
        ops.set_tuning('s2_classes', 0)
        ref = run()
        ops.set_tuning('s2_classes', 1)
        got = run()
    finally:
        ops.set_tuning('s2_classes', 1)
    # same products, different summation order: within a bf16 ulp (of the GRADIENT term where base + gradient cancels: |dgrad| <= 4)
    torch.testing.assert_close(got[0], ref[0], rtol=2 ** -7, atol=2e-3)
    torch.testing.assert_close(got[1], ref[1], rtol=2 ** -6, atol=2 ** -6)


@pytest.mark.parametrize('shape', [(2, 52, 52, 64, 128), (3, 26, 30, 128, 256), (1, 27, 25, 64, 64)], ids=str)
def test_downsampling_block_input_gradient_sparse_shortcut(dev, shape):
    """the input gradient of a down-sampling block = data gradient of its 1x1 / stride-2 shortcut + data gradient of its 3x3 / stride-2
    convolution.  Dense form: the shortcut's gradient written everywhere (zeros at three quarters of the positions), the 3x3 one accumulated
    onto it.  Sparse form: conv2d_dgrad(even_only=True) touches the even / even positions only and the 3x3 launch accumulates onto that
    parity class alone (accumulate=2) -- the same bits, on a buffer that starts as NaN; with and without the BatchNorm reduce epilogue"""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    p3 = ops.conv_problem(N, H, W, Cin, Cout, 3, 2, 'same')
    p1 = ops.conv_problem(N, H, W, Cin, Cout, 1, 2, 'same')
    assert ops.conv2d_dgrad_classed(p3) and not ops.conv2d_dgrad_classed(p1) and (p1.Ho, p1.Wo) == (p3.Ho, p3.Wo)
    w3 = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(dev)
    w1 = bf(torch.randn(Cout, 1, 1, Cin, generator=g) * 0.1).to(dev)
    w3d = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    w1d = torch.empty(Cin, 1, 1, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w3, w3d, Cout, 3, 3, Cin)
    ops.repack_dgrad_weights(w1, w1d, Cout, 1, 1, Cin)
    dy3 = bf(torch.randn(N, p3.Ho, p3.Wo, Cout, generator=g)).to(dev)
    dy1 = bf(torch.randn(N, p3.Ho, p3.Wo, Cout, generator=g)).to(dev)
    M = N * H * W
    y = bf(torch.randn(M, Cin, generator=g)).to(dev)
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    rows = ops.conv2d_dgrad_bn_rows(p3)
    for fused in (False, True):
        res = []
        for sparse in (False, True):
            dx = torch.full((N, H, W, Cin), float('nan'), dtype=ACT(), device=dev)
            ops.conv2d_dgrad(p1, dy1, w1d, dx, even_only=sparse)
            part = torch.zeros(rows, 3, Cin, device=dev)
            kw = dict(bn=dict(mask=mask, y=y, mean=mean, rstd=rstd, partial=part)) if fused else {}
            ops.conv2d_dgrad(p3, dy3, w3d, dx, accumulate=2 if sparse else True, **kw)
            torch.cuda.synchronize()
            res.append((dx.clone(), part.clone()))
        assert torch.isfinite(res[1][0].float()).all()
        assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
        assert torch.equal(res[0][1], res[1][1])
    # the even-only launch by itself: the dense gradient at the even / even positions, nothing elsewhere
    dense = torch.empty(N, H, W, Cin, dtype=ACT(), device=dev)
    ops.conv2d_dgrad(p1, dy1, w1d, dense)
    sparse = torch.full((N, H, W, Cin), 7.0, dtype=ACT(), device=dev)
    ops.conv2d_dgrad(p1, dy1, w1d, sparse, even_only=True)
    torch.cuda.synchronize()
    assert torch.equal(sparse[:, ::2, ::2].float(), dense[:, ::2, ::2].float())
    keep = torch.ones(H, W, dtype=torch.bool, device=dev)
    keep[::2, ::2] = False
    assert bool((sparse[:, keep].float() == 7.0).all()) and float(dense[:, keep].float().abs().max()) == 0.0


def test_conv_fused_upsample_concat(dev):
    """1x1 conv over concat(upsample2x(a), b) without materialising the concat (yolov3_detector.py:115-118)"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(7)
    N, H, W, C0, C1, Cout = 2, 12, 8, 64, 64, 128
    a = bf(torch.randn(N, H // 2, W // 2, C0, generator=g))
    b = bf(torch.randn(N, H, W, C1, generator=g))
    w = bf(torch.randn(1, 1, C0 + C1, Cout, generator=g) * 0.1)
    cat = torch.cat([a.float().repeat_interleave(2, 1).repeat_interleave(2, 2), b.float()], dim=-1).requires_grad_(True)
    wr = w.float().requires_grad_(True)
    y_ref = ref_conv(cat, wr, 1, 0, 0, H, W)
    dy = bf(torch.randn(N, H, W, Cout, generator=g))
    y_ref.backward(dy.float())
    p = ops.conv_problem(N, H, W, C0 + C1, Cout, 1, 1, 'same', C0=C0)
    w_fwd = w.permute(3, 0, 1, 2).contiguous().to(dev)
    y = torch.empty(N, H, W, Cout, dtype=ACT(), device=dev)
    ops.conv2d_fwd(p, b.to(dev), w_fwd, y, src0=a.to(dev))
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), rtol=1e-2, atol=1e-2)
    dw = torch.zeros(Cout, 1, 1, C0 + C1, device=dev)
    ops.conv2d_wgrad(p, b.to(dev), dy.to(dev), dw, src0=a.to(dev))
    torch.testing.assert_close(dw.cpu(), wr.grad.permute(3, 0, 1, 2), rtol=1e-3, atol=1e-3)
    ws = torch.empty(max(ops.conv2d_wgrad_workspace_bytes(p), 16) // 4, device=dev)
    dw_r = torch.empty_like(dw)
    ops.conv2d_wgrad_reduce(p, b.to(dev), dy.to(dev), dw_r, ws, src0=a.to(dev))
    torch.testing.assert_close(dw_r.cpu(), wr.grad.permute(3, 0, 1, 2), rtol=1e-3, atol=1e-3)
    # dgrad over the virtual concat, then the split kernel
    pd = ops.conv_problem(N, H, W, C0 + C1, Cout, 1, 1, 'same')
    w_dg = torch.empty(C0 + C1, 1, 1, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w_fwd, w_dg, Cout, 1, 1, C0 + C1)
    dcat = torch.empty(N, H, W, C0 + C1, dtype=ACT(), device=dev)
    ops.conv2d_dgrad(pd, dy.to(dev), w_dg, dcat)
    da = torch.empty(N, H // 2, W // 2, C0, dtype=ACT(), device=dev)
    db = torch.full((N, H, W, C1), 1.0, dtype=ACT(), device=dev)
    ops.upcat_split_bwd(dcat, da, False, db, True, N, H, W, C0, C1)
    gcat = cat.grad
    da_ref = gcat[..., :C0].reshape(N, H // 2, 2, W // 2, 2, C0).sum(dim=(2, 4))
    torch.testing.assert_close(da.float().cpu(), da_ref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(db.float().cpu(), gcat[..., C0:] + 1.0, rtol=2e-2, atol=3e-2)


def _bn_ref(y, gamma, beta, eps=1e-5):
    m = y.mean(dim=(0, 1, 2))
    v = ((y - m) ** 2).mean(dim=(0, 1, 2))
    return (y - m) * torch.rsqrt(v + eps) * gamma + beta, m, v


@pytest.mark.parametrize('mode', ['plain', 'res', 'res_bn'])
def test_bn_act_fwd_bwd(dev, mode):
    """conv-output statistics -> finalize -> apply(+residual)(+ReLU), and the two backward passes, vs autograd"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(11)
    N, H, W, Cc = 3, 9, 7, 64
    M = N * H * W
    y = bf(torch.randn(N, H, W, Cc, generator=g) * 2 + 0.5)
    y2 = bf(torch.randn(N, H, W, Cc, generator=g))
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    gamma2, beta2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    dout = bf(torch.randn(N, H, W, Cc, generator=g))

    yr, y2r = y.float().requires_grad_(True), y2.float().requires_grad_(True)
    gr, br, g2r, b2r = [t.clone().requires_grad_(True) for t in (gamma, beta, gamma2, beta2)]
    o, m1, v1 = _bn_ref(yr, gr, br)
    if mode == 'res':
        o = o + y2r
    elif mode == 'res_bn':
        o2, m2, v2 = _bn_ref(y2r, g2r, b2r)
        o = o + o2
    out_ref = torch.relu(o)
    out_ref_b = bf(out_ref.detach())
    out_ref.backward(dout.float())      # relu mask of the float32 output == mask of the stored bf16 output (rounding keeps sign)

    d = lambda t: t.to(dev)
    yd, y2d = d(y), d(y2)

    def stats(x):
        rows = ops.reduce_rows(M, Cc)
        part = torch.empty(rows, 2, Cc, device=dev)
        ops.bn_stats(x, M, Cc, part)
        return part, rows

    def finalize(part, rows, ga, be):
        sc, sh, mean, rstd = [torch.empty(Cc, device=dev) for _ in range(4)]
        mm, mv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        ops.bn_finalize(part, part[0, 1], rows, 2 * Cc, Cc, M, d(ga), d(be), 1e-5, 0.9, mm, mv, sc, sh, mean, rstd)
        return sc, sh, mean, rstd, mm, mv

    part, rows = stats(yd)
    sc, sh, mean, rstd, mm, mv = finalize(part, rows, gamma, beta)
    torch.testing.assert_close(mean.cpu(), m1.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.cpu(), torch.rsqrt(v1.detach() + 1e-5), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mm.cpu(), 0.1 * m1.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mv.cpu(), 0.9 + 0.1 * v1.detach() * M / (M - 1), rtol=1e-4, atol=1e-5)
    out = torch.empty(N, H, W, Cc, dtype=ACT(), device=dev)
    kw = {}
    if mode == 'res':
        kw = dict(res=y2d)
    elif mode == 'res_bn':
        part2, rows2 = stats(y2d)
        sc2, sh2, mean2, rstd2, _, _ = finalize(part2, rows2, gamma2, beta2)
        kw = dict(res=y2d, res_scale=sc2, res_shift=sh2)
    ops.bn_act_fwd(yd, sc, sh, out, M, Cc, True, **kw)
    torch.testing.assert_close(out.float().cpu(), out_ref_b.float(), rtol=1e-2, atol=1e-2)

    # ---- backward ----
    out_d, dout_d = d(out_ref_b), d(dout)       # use the reference's stored output so masks agree exactly
    P = ops.reduce_rows(M, Cc)
    partial = torch.empty(P, 3, Cc, device=dev)
    if mode == 'res_bn':
        ops.bn_act_bwd_reduce(dout_d, out_d, True, yd, mean, rstd, M, Cc, partial, y2=y2d, mean2=mean2, rstd2=rstd2)
    else:
        ops.bn_act_bwd_reduce(dout_d, out_d, True, yd, mean, rstd, M, Cc, partial)
    dga, dbe, k1, k2 = [torch.empty(Cc, device=dev) for _ in range(4)]
    ops.bn_bwd_finalize(partial, P, Cc, 1, M, dga, dbe, k1, k2)
    torch.testing.assert_close(dga.cpu(), gr.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(dbe.cpu(), br.grad, rtol=2e-3, atol=2e-3)
    dy = torch.empty(N, H, W, Cc, dtype=ACT(), device=dev)
    if mode == 'plain':
        ops.bn_act_bwd_apply(dout_d, out_d, True, M, Cc, y=yd, a1=sc, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy)
    elif mode == 'res':
        dres = torch.empty_like(dy)
        ops.bn_act_bwd_apply(dout_d, out_d, True, M, Cc, y=yd, a1=sc, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy, dres=dres)
        torch.testing.assert_close(dres.float().cpu(), y2r.grad, rtol=1e-2, atol=1e-2)
    else:
        dga2, dbe2, k1b, k2b = [torch.empty(Cc, device=dev) for _ in range(4)]
        ops.bn_bwd_finalize(partial, P, Cc, 2, M, dga2, dbe2, k1b, k2b)
        torch.testing.assert_close(dga2.cpu(), g2r.grad, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(dbe2.cpu(), b2r.grad, rtol=2e-3, atol=2e-3)
        dy2 = torch.empty_like(dy)
        ops.bn_act_bwd_apply(dout_d, out_d, True, M, Cc, y=yd, a1=sc, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy,
                             y2=y2d, a2=sc2, mean2=mean2, rstd2=rstd2, k1b=k1b, k2b=k2b, dy2=dy2)
        torch.testing.assert_close(dy2.float().cpu(), y2r.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dy.float().cpu(), yr.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('M,C,P,res', [(32 * 13 * 13, 512, 85, True), (32 * 26 * 26, 256, 338, False), (700, 32, 3, True), (5, 64, 1, False),
                                        (32 * 26 * 26, 128, 169, True),
                                        # (round 4: few rows + a large tensor = the streaming form that sums the rows in its own prologue)
                                        (32 * 52 * 52, 128, 22, True), (16 * 104 * 104, 64, 43, False), (32 * 26 * 26, 256, 11, True)], ids=str)
def test_small_map_finalize_plus_apply_in_one_launch(dev, M, C, P, res):
    """yolo_bn_finalize_act_fwd == yolo_bn_finalize + yolo_bn_act_fwd (mask variant) and yolo_bn_bwd_finalize_apply == yolo_bn_bwd_finalize +
    yolo_bn_act_bwd_apply on the same partial rows: identical activations / masks / gradients wherever the per-channel constants agree bit
    for bit (they are column sums in double, summed in another order: 1e-6 relative at most), moving statistics updated once"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(M + C + P)
    d = lambda t: t.to(dev)
    y = d(bf(torch.randn(M, C, generator=g) * 1.5 + 0.3))
    r = d(bf(torch.randn(M, C, generator=g))) if res else None
    # partial rows as the convolution epilogue leaves them: per-tile sums of y and y^2 (tile = consecutive rows)
    rows_per = (M + P - 1) // P
    yf = y.float()
    pad = torch.zeros(P * rows_per - M, C, device=dev)
    yt = torch.cat([yf, pad]).reshape(P, rows_per, C)
    psum, psq = yt.sum(1).contiguous(), (yt * yt).sum(1).contiguous()
    gamma, beta = d(torch.rand(C, generator=g) + 0.5), d(torch.randn(C, generator=g) * 0.2)
    outs = []
    for merged in (False, True):
        mm, mv = d(torch.full((C,), 0.25)), d(torch.full((C,), 2.0))
        sc, sh, mean, rstd = [torch.zeros(C, device=dev) for _ in range(4)]
        out = torch.empty(M, C, dtype=ACT(), device=dev)
        mask = torch.zeros(M * C // 8, dtype=torch.uint8, device=dev)
        if merged:
            ops.bn_finalize_act_fwd(psum.view(-1), psq.view(-1), P, C, C, M, gamma, beta, 1e-5, 0.9, mm, mv, sc, sh, mean, rstd, y, out, M, True,
                                    res=r, mask=mask)
        else:
            ops.bn_finalize(psum.view(-1), psq.view(-1), P, C, C, M, gamma, beta, 1e-5, 0.9, mm, mv, sc, sh, mean, rstd)
            ops.bn_act_fwd(y, sc, sh, out, M, C, True, res=r, mask=mask)
        torch.cuda.synchronize()
        outs.append((mm, mv, sc, sh, mean, rstd, out, mask))
    a, b = outs
    for u, v in zip(a[:6], b[:6]):
        torch.testing.assert_close(u, v, rtol=2e-6, atol=1e-7)
    same = ((a[2] == b[2]) & (a[3] == b[3]))                        # channels whose scale / shift came out bit-identical
    assert same.float().mean() > 0.9
    assert torch.equal(a[6][:, same], b[6][:, same])
    torch.testing.assert_close(a[6].float(), b[6].float(), rtol=1e-2, atol=1e-2)
    if bool(same.all()):
        assert torch.equal(a[7], b[7])
    # reference check of the merged launch on its own: float64 statistics of the stored values
    m64, v64 = yf.double().mean(0), yf.double().var(0, unbiased=False)
    torch.testing.assert_close(b[4].double(), m64, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b[5].double(), 1.0 / torch.sqrt(v64 + 1e-5), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b[0].double(), 0.9 * 0.25 + 0.1 * m64, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b[1].double(), 0.9 * 2.0 + 0.1 * v64 * (M / max(M - 1.0, 1.0)), rtol=1e-5, atol=1e-6)

    # backward: masked gradient g, partial rows [P][3][C] with quantities 0 (sum g) and 1 (sum g * xhat)
    mean, rstd, scale = b[4], b[5], b[2]
    gq = d(bf(torch.randn(M, C, generator=g)))
    gq = torch.where(b[6] > 0, gq, torch.zeros_like(gq))
    xhat = (yf - mean) * rstd
    gt = torch.cat([gq.float(), pad]).reshape(P, rows_per, C)
    xt = torch.cat([gq.float() * xhat, pad]).reshape(P, rows_per, C)
    part = torch.zeros(P, 3, C, device=dev)
    part[:, 0], part[:, 1] = gt.sum(1), xt.sum(1)
    prev_dy = d(bf(torch.randn(M, C, generator=g)))
    prev_dres = d(bf(torch.randn(M, C, generator=g)))
    outs = []
    for acc in (False, True):
        pair = []
        for merged in (False, True):
            dg, db, k1, k2 = [torch.zeros(C, device=dev) for _ in range(4)]
            dy = prev_dy.clone()
            dres = prev_dres.clone() if res else None
            if merged:
                ops.bn_bwd_finalize_apply(part.view(-1), P, C, M, dg, db, k1, k2, gq, y, scale, mean, rstd, M, dy, acc_dy=acc, dres=dres,
                                          acc_dres=acc)
            else:
                ops.bn_bwd_finalize(part.view(-1), P, C, 1, M, dg, db, k1, k2)
                ops.bn_act_bwd_apply(gq, None, 0, M, C, y=y, a1=scale, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy, acc_dy=acc, dres=dres,
                                     acc_dres=acc)
            torch.cuda.synchronize()
            pair.append((dg, db, k1, k2, dy) + ((dres,) if res else ()))
        u, v = pair
        for i in range(4):       # (the two-launch finalize adds up to 3 rows per lane in float32 before widening when P > 128: 1e-7 of the partials)
            torch.testing.assert_close(u[i], v[i], rtol=2e-5, atol=2e-5)
        same = (u[2] == v[2]) & (u[3] == v[3])
        assert same.float().mean() > (0.9 if P <= 128 else 0.3)
        assert torch.equal(u[4][:, same], v[4][:, same])
        torch.testing.assert_close(u[4].float(), v[4].float(), rtol=1e-2, atol=1e-2)
        if res:
            assert torch.equal(u[5], v[5])


@pytest.mark.parametrize('P,C', [(85, 512), (169, 256), (676, 128), (1352, 64), (5408, 64), (1, 64), (129, 72)])
def test_bwd_finalize_small_workgroups_add_in_the_same_order(dev, P, C):
    """yolo_bn_bwd_finalize on 256-thread workgroups (`bwd_fin_small`; a 1024-thread workgroup can starve beside the slab-sum kernel of the
    other stream) against the default 1024-thread form: the same float partial sums, wave butterflies and wave order -- BIT-identical results (the
    float16 loss curve has 1e-4 of margin to north_star's 1e-3: a merely different summation order is not free)"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(P + C)
    part = (torch.randn(P, 3, C, generator=g) * 3.0).to(dev)
    res = []
    try:
        for small in (0, 1):
            ops.set_tuning('bwd_fin_small', small)
            out = [torch.full((C,), float('nan'), device=dev) for _ in range(4)]
            ops.bn_bwd_finalize(part.view(-1), P, C, 1, 1234.0, *out)
            torch.cuda.synchronize()
            res.append(out)
    finally:
        ops.set_tuning('bwd_fin_small', 0)
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # (float partial sums over up to 11 rows before the widening: 1e-6 of the partials' magnitude)
    torch.testing.assert_close(res[1][1].double(), part[:, 0].double().sum(0), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(res[1][0].double(), part[:, 1].double().sum(0), rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize('M,Cc,mode', [(32 * 13 * 13, 512, 'plain'), (32 * 26 * 26, 256, 'res'), (32 * 52 * 52, 128, 'bn2'),
                                       (3001, 64, 'res_acc'), (32 * 104 * 104, 64, 'plain'), (32 * 104 * 104 * 2, 64, 'too_big')])
@pytest.mark.parametrize('small_grid', [0, 128])
def test_bn_bwd_fused_matches_three_kernel_path(dev, M, Cc, mode, small_grid):
    """the single-launch BatchNorm backward (resident grid, device-wide hand-off) against reduce / finalize / apply on the same inputs:
    plain BN+ReLU, identity residual (with and without fan-in accumulation), shortcut-BN branch, 2 / 6 / 11 chunks per thread, a ragged
    size, and the size it has to refuse; two launches in a row reuse the (monotonic) barrier word"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(M % 1000 + Cc)
    y = bf(torch.randn(M, Cc, generator=g) * 1.5 + 0.3).to(dev)
    y2 = bf(torch.randn(M, Cc, generator=g)).to(dev)
    dout = bf(torch.randn(M, Cc, generator=g)).to(dev)
    out = bf(torch.randn(M, Cc, generator=g)).to(dev)           # stands for relu(...) > 0 mask: sign pattern only
    prev = bf(torch.randn(M, Cc, generator=g)).to(dev)
    mean, rstd = torch.randn(Cc, generator=g).to(dev) * 0.2, (torch.rand(Cc, generator=g) + 0.5).to(dev)
    mean2, rstd2 = torch.randn(Cc, generator=g).to(dev) * 0.2, (torch.rand(Cc, generator=g) + 0.5).to(dev)
    a1, a2 = (torch.randn(Cc, generator=g)).to(dev), (torch.randn(Cc, generator=g)).to(dev)
    has2, res, acc = mode == 'bn2', mode in ('res', 'res_acc'), mode == 'res_acc'

    # three-kernel path
    P = ops.reduce_rows(M, Cc)
    partial = torch.zeros(P, 3, Cc, device=dev)
    dg, db, k1, k2 = [torch.zeros(Cc, device=dev) for _ in range(4)]
    dg2, db2, k1b, k2b = [torch.zeros(Cc, device=dev) for _ in range(4)]
    ops.bn_act_bwd_reduce(dout, out, True, y, mean, rstd, M, Cc, partial, y2=y2 if has2 else None, mean2=mean2 if has2 else None,
                          rstd2=rstd2 if has2 else None)
    ops.bn_bwd_finalize(partial.view(-1), P, Cc, 1, M, dg, db, k1, k2)
    kw = {}
    if has2:
        ops.bn_bwd_finalize(partial.view(-1), P, Cc, 2, M, dg2, db2, k1b, k2b)
        dy2_ref = torch.empty_like(y)
        kw.update(y2=y2, a2=a2, mean2=mean2, rstd2=rstd2, k1b=k1b, k2b=k2b, dy2=dy2_ref)
    dy_ref = prev.clone()
    dres_ref = prev.clone()
    if res:
        kw.update(dres=dres_ref, acc_dres=acc)
    ops.bn_act_bwd_apply(dout, out, True, M, Cc, y=y, a1=a1, mean=mean, rstd=rstd, k1=k1, k2=k2, dy=dy_ref, acc_dy=acc, **kw)

    # small sizes: either force them onto the full grid (min chunks 1) or send them to the fixed small grid with its own counters
    need_full = -(-(M * (Cc // 8)) // (256 * 1024))
    on_small = bool(small_grid) and need_full < 3
    ops.set_tuning('bn_fused_min_chunks', 3 if small_grid else 1)
    ops.set_tuning('bn_fused_small_grid', small_grid)
    ws = torch.zeros(ops.bn_bwd_fused_workspace_floats(Cc), device=dev)
    sync = torch.zeros(ops.bn_bwd_fused_sync_words(), dtype=torch.int32, device=dev)
    for rep in range(2):
        fdg, fdb, fdg2, fdb2 = [torch.zeros(Cc, device=dev) for _ in range(4)]
        dy = prev.clone()
        dres = prev.clone()
        dy2 = torch.empty_like(y)
        kw = {}
        if has2:
            kw.update(y2=y2, a2=a2, mean2=mean2, rstd2=rstd2, dgamma2=fdg2, dbeta2=fdb2, dy2=dy2)
        if res:
            kw.update(dres=dres, acc_dres=acc)
        ok = ops.bn_act_bwd_fused(dout, out, True, M, Cc, y, a1, mean, rstd, fdg, fdb, dy, ws, sync, acc_dy=acc, **kw)
        if mode == 'too_big':
            assert not ok
            ops.set_tuning('bn_fused_min_chunks', 3)
            ops.set_tuning('bn_fused_small_grid', 0)
            return
        assert ok
        torch.cuda.synchronize()
        assert ops.bn_fused_timeouts(sync) == 0
        first = int(sync[sync.numel() // 2 if on_small else 0])               # shard 0 of the counter set this grid size uses
        assert first > 0 and first % (2 * (rep + 1)) == 0                     # two grid barriers per launch
        assert int(sync[0 if on_small else sync.numel() // 2]) == 0           # the other set is untouched
        scale = max(dg.abs().max().item(), 1.0)
        torch.testing.assert_close(fdg, dg, rtol=2e-4, atol=2e-5 * scale)
        torch.testing.assert_close(fdb, db, rtol=2e-4, atol=2e-5 * max(db.abs().max().item(), 1.0))
        # same formula, constants folded differently (A g + B y + D): within a bf16 ulp of the reference result
        torch.testing.assert_close(dy.float(), dy_ref.float(), rtol=2 ** -7, atol=2e-2)
        if has2:
            torch.testing.assert_close(fdg2, dg2, rtol=2e-4, atol=2e-5 * max(dg2.abs().max().item(), 1.0))
            torch.testing.assert_close(dy2.float(), dy2_ref.float(), rtol=2 ** -7, atol=2e-2)
        if res:
            assert torch.equal(dres, dres_ref) or torch.allclose(dres.float(), dres_ref.float(), rtol=2 ** -8, atol=0)
    ops.set_tuning('bn_fused_min_chunks', 3)
    ops.set_tuning('bn_fused_small_grid', 0)


def test_bn_pool_relu_fwd_bwd(dev):
    """stem: conv -> BN -> maxpool(3,2,'same') -> ReLU (resnet18.py:59-61) and its backward"""
    from yolov3_tensorflow_amd import ops
    from oracle.nets import same_pad
    g = torch.Generator().manual_seed(5)
    N, H, W, Cc = 2, 12, 10, 64
    M = N * H * W
    # distinct values per channel so the arg-max is unique (ties are resolved first-max by both sides anyway)
    y = bf(torch.randn(N, H, W, Cc, generator=g))
    gamma, beta = torch.rand(Cc, generator=g) - 0.3, torch.randn(Cc, generator=g) * 0.1      # some negative gammas
    yr, gr, br = y.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    o, m1, v1 = _bn_ref(yr, gr, br)
    (pt, pb), (pl, pr) = same_pad(H, 3, 2), same_pad(W, 3, 2)
    pooled = F.max_pool2d(F.pad(o.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf')), 3, 2).permute(0, 2, 3, 1)
    out_ref = torch.relu(pooled)
    Ho, Wo = out_ref.shape[1], out_ref.shape[2]
    dout = bf(torch.randn(N, Ho, Wo, Cc, generator=g))
    out_b = bf(out_ref.detach())
    out_ref.backward(dout.float())
    d = lambda t: t.to(dev)
    yd = d(y)
    rows = ops.reduce_rows(M, Cc)
    part = torch.empty(rows, 2, Cc, device=dev)
    ops.bn_stats(yd, M, Cc, part)
    sc, sh, mean, rstd = [torch.empty(Cc, device=dev) for _ in range(4)]
    ops.bn_finalize(part, part[0, 1], rows, 2 * Cc, Cc, M, d(gamma), d(beta), 1e-5, 0.9, None, None, sc, sh, mean, rstd)
    out = torch.empty(N, Ho, Wo, Cc, dtype=ACT(), device=dev)
    arg = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=dev)
    ops.bn_pool_fwd(yd, sc, sh, out, arg, N, H, W, Cc, Ho, Wo, pt, pl, True)
    torch.testing.assert_close(out.float().cpu(), out_b.float(), rtol=1e-2, atol=1e-2)
    partial = torch.empty(rows, 3, Cc, device=dev)
    for fast in (False, True):        # gather formulation over the pre-pool map, and the pooled-map formulation (xhat from out)
        kw = dict(gamma=d(gamma), beta=d(beta)) if fast else {}
        ops.bn_pool_bwd_reduce(d(dout), out, arg, True, yd, mean, rstd, N, H, W, Cc, Ho, Wo, pt, pl, partial, **kw)
        dga, dbe, k1, k2 = [torch.empty(Cc, device=dev) for _ in range(4)]
        ops.bn_bwd_finalize(partial, rows, Cc, 1, M, dga, dbe, k1, k2)
        torch.testing.assert_close(dga.cpu(), gr.grad, rtol=1e-2, atol=2e-2)
        torch.testing.assert_close(dbe.cpu(), br.grad, rtol=5e-3, atol=5e-3)
    dy = torch.empty(N, H, W, Cc, dtype=ACT(), device=dev)
    ops.bn_pool_bwd_apply(d(dout), out, arg, True, yd, sc, mean, rstd, k1, k2, dy, N, H, W, Cc, Ho, Wo, pt, pl)
    torch.testing.assert_close(dy.float().cpu(), yr.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('shape', [(2, 70, 54, 64), (3, 33, 47, 32), (1, 16, 16, 64), (2, 208, 208, 64)], ids=str)
def test_pool_bwd_scatter_equals_gather(dev, shape):
    """the scatter form of the stem's pooled backward apply (four disjoint window classes, LDS float tile) adds the same contributions in
    the same order as the gather form: bit-identical dy, on maps of several 16 x 16 tiles with ragged edges and both pad parities"""
    from yolov3_tensorflow_amd import ops
    from oracle.nets import same_pad
    N, H, W, Cc = shape
    g = torch.Generator().manual_seed(H * 1000 + W)
    y = bf(torch.randn(N, H, W, Cc, generator=g)).to(dev)
    (pt, _), (pl, _) = same_pad(H, 3, 2), same_pad(W, 3, 2)
    Ho, Wo = -(-H // 2), -(-W // 2)
    sc = (torch.rand(Cc, generator=g) - 0.3).to(dev)
    sh, mean, rstd, k1, k2 = [(torch.randn(Cc, generator=g) * s).to(dev) for s in (0.1, 0.2, 1.0, 0.05, 0.05)]
    out = torch.empty(N, Ho, Wo, Cc, dtype=ACT(), device=dev)
    arg = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=dev)
    ops.bn_pool_fwd(y, sc, sh, out, arg, N, H, W, Cc, Ho, Wo, pt, pl, True)
    dout = bf(torch.randn(N, Ho, Wo, Cc, generator=g)).to(dev)
    res = []
    try:
        for scatter in (0, 1):
            ops.set_tuning('pool_scatter', scatter)
            dy = torch.full((N, H, W, Cc), float('nan'), dtype=ACT(), device=dev)
            ops.bn_pool_bwd_apply(dout, out, arg, True, y, sc, mean, rstd, k1, k2, dy, N, H, W, Cc, Ho, Wo, pt, pl)
            torch.cuda.synchronize()
            res.append(dy.float().cpu())
    finally:
        ops.set_tuning('pool_scatter', 1)
    assert torch.isfinite(res[0]).all()
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize('shape,bn', [((2, 64, 96), True), ((3, 40, 56), True), ((1, 416, 416), True), ((2, 72, 40), False),
                                      ((1, 16, 32), True)], ids=str)
def test_stem_backward_in_one_kernel(dev, shape, bn):
    """stem_pool_bwd_wgrad (un-pool + BatchNorm apply + stem weight gradient, pre-pool gradient never written) against the two-kernel
    path on the same inputs: bn_pool_bwd_apply -> dy (16-bit) -> conv2d_wgrad.  Same products of the same rounded dy, float32 sums in a
    different order.  Ragged tiles (40 x 56 input: 20 x 28 pre-pool map), one tile only (a single slab), and the no-BatchNorm stem of
    ResNet18-v2 (resnet18_v2.py:61-62: dy = the un-pooled gradient)"""
    from yolov3_tensorflow_amd import ops
    from oracle.nets import same_pad
    N, Hi, Wi = shape
    g = torch.Generator().manual_seed(Hi * 7 + Wi)
    p = ops.conv_problem(N, Hi, Wi, 8, 64, 3, 2, 'same')
    H, W, Cc = p.Ho, p.Wo, 64
    x = torch.zeros(N, Hi, Wi, 8)
    x[..., :3] = torch.rand(N, Hi, Wi, 3, generator=g)
    x = bf(x).to(dev)
    y = bf(torch.randn(N, H, W, Cc, generator=g)).to(dev)
    (pt, _), (pl, _) = same_pad(H, 3, 2), same_pad(W, 3, 2)
    Ho, Wo = -(-H // 2), -(-W // 2)
    sc = (torch.rand(Cc, generator=g) - 0.3).to(dev)
    sh, mean, rstd, k1, k2 = [(torch.randn(Cc, generator=g) * s_).to(dev) for s_ in (0.1, 0.2, 1.0, 0.05, 0.05)]
    out = torch.empty(N, Ho, Wo, Cc, dtype=ACT(), device=dev)
    arg = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=dev)
    ops.bn_pool_fwd(y, sc if bn else None, sh if bn else None, out, arg, N, H, W, Cc, Ho, Wo, pt, pl, bn)       # ReLU only on the BN stem
    dout = bf(torch.randn(N, Ho, Wo, Cc, generator=g)).to(dev)
    a = (sc, mean, rstd, k1, k2) if bn else (None,) * 5
    dy = torch.empty(N, H, W, Cc, dtype=ACT(), device=dev)
    ops.bn_pool_bwd_apply(dout, out, arg, bn, y if bn else None, *a, dy, N, H, W, Cc, Ho, Wo, pt, pl)
    dw_ref = torch.zeros(64, 3, 3, 8, device=dev)
    ops.conv2d_wgrad(p, x, dy, dw_ref)
    n = ops.stem_pool_bwd_slabs(p, Cc, Ho, Wo, pt, pl)
    tiles = N * -(-H // 8) * -(-W // 16)
    assert n == min(tiles, 768)
    slabs = torch.full((n, 64, 3, 3, 8), float('nan'), device=dev)
    ops.stem_pool_bwd_wgrad(p, x, dout, out, arg, bn, y, *a, Ho, Wo, pt, pl, slabs)
    torch.cuda.synchronize()
    got = slabs.double().sum(0)
    assert torch.isfinite(got).all() and float(got[..., 3:].abs().max()) == 0.0
    scale = float(dw_ref.abs().max())
    torch.testing.assert_close(got.float(), dw_ref, rtol=2e-3, atol=2e-4 * max(scale, 1.0))
    # the bucket's summing launch takes the many-slab form (16 slab lanes) from 64 slabs on
    if n > 1:
        grads = torch.zeros(64 * 72, device=dev)
        n4 = 64 * 72 // 4
        blocks = ops.reduce_blocks(n4, n)
        tab = torch.tensor([[0, 0, n4, n, 0]], dtype=torch.int64, device=dev)
        ops.wgrad_reduce_batched(tab, 1, blocks, slabs.view(-1), grads)
        torch.cuda.synchronize()
        torch.testing.assert_close(grads.view(64, 3, 3, 8), got.float(), rtol=1e-5, atol=1e-6 * max(scale, 1.0))


@pytest.mark.parametrize('f,N,H,W', [(64, 2, 11, 9), (128, 2, 11, 9), (64, 3, 40, 104), (512, 5, 13, 13), (256, 2, 26, 26), (64, 1, 3, 2)])
def test_mixconv_fwd_dgrad_wgrad(dev, f, N, H, W):
    """mixed depthwise conv (mixnet18.py:38-45) vs 4 x F.conv2d(groups=C_g) on channel slices: ragged strips, several row tiles per
    image (W = 104), 1 / 2 / 4 chunks per workgroup, maps smaller than the largest kernel"""
    from yolov3_tensorflow_amd import ops
    g = torch.Generator().manual_seed(21)
    split = [0, f // 2, 3 * f // 4, 7 * f // 8, f]
    ks = [3, 5, 7, 9]
    x = bf(torch.randn(N, H, W, f, generator=g))
    ws = [bf(torch.randn(k, k, split[i + 1] - split[i], generator=g) * (1.0 / k)) for i, k in enumerate(ks)]
    xr = x.float().requires_grad_(True)
    wr = [w.float().requires_grad_(True) for w in ws]
    outs = []
    for i, k in enumerate(ks):
        xs = xr[..., split[i]:split[i + 1]].permute(0, 3, 1, 2)
        cg = split[i + 1] - split[i]
        outs.append(F.conv2d(xs, wr[i].permute(2, 0, 1).unsqueeze(1), padding=k // 2, groups=cg).permute(0, 2, 3, 1))
    y_ref = torch.cat(outs, dim=-1)
    dy = bf(torch.randn(N, H, W, f, generator=g))
    y_ref.backward(dy.float())
    p = ops.mix_problem(N, H, W, f, split, ks)
    d = lambda t: t.to(dev)
    y = torch.empty(N, H, W, f, dtype=ACT(), device=dev)
    wd = [d(w) for w in ws]
    ops.dwconv_mix_fwd(p, d(x), wd, y)
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), rtol=1e-2, atol=1e-2)
    dx = torch.full((N, H, W, f), 0.5, dtype=ACT(), device=dev)
    ops.dwconv_mix_dgrad(p, d(dy), wd, dx, accumulate=True)
    torch.testing.assert_close(dx.float().cpu(), xr.grad + 0.5, rtol=1e-2, atol=2e-2)
    dw = [torch.full((k, k, split[i + 1] - split[i]), 3.0, device=dev) for i, k in enumerate(ks)]
    wsp = torch.empty(max(ops.dwconv_mix_wgrad_workspace_bytes(p), 16) // 4, device=dev)
    ops.dwconv_mix_wgrad(p, d(x), d(dy), dw, wsp)                      # overwrites
    for a, b in zip(dw, wr):
        torch.testing.assert_close(a.cpu(), b.grad, rtol=1e-3, atol=1e-3 * max(1.0, b.grad.abs().max().item()))
    ops.dwconv_mix_wgrad(p, d(x), d(dy), dw, wsp, accumulate=True)
    for a, b in zip(dw, wr):
        torch.testing.assert_close(a.cpu(), 2 * b.grad, rtol=1e-3, atol=2e-3 * max(1.0, b.grad.abs().max().item()))


@pytest.mark.parametrize('C,k', [(64, 3), (128, 5), (64, 9)])
def test_lone_depthwise_conv_bn_graph(dev, C, k):
    """BasicBackbone.depthwise_conv_bn (reference basic_backbone.py:45-66,140-150) as a graph node: forward of DepthwiseConv2D(k) -> BN -> ReLU
    and the weight gradient through the engine, against F.conv2d(groups=C) + batch_norm + autograd in float32 on the same bf16 operands"""
    from yolov3_tensorflow_amd import engine, ops
    from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone
    N, H, W = 2, 12, 10
    g = engine.Graph(N, dev)
    x = g.input(H, W, 3)
    c = BasicBackbone.conv_bn(x, C, kernel_size=(1, 1))
    a = BasicBackbone.activation(c)
    out = BasicBackbone.activation(BasicBackbone.depthwise_conv_bn(a, kernel_size=(k, k)))
    g.finalize([])
    names = list(g.ps.params.keys())
    assert 'depthwise_conv2d/depthwise_kernel' in names and g.ps.params['depthwise_conv2d/depthwise_kernel'].tf_shape == (k, k, C, 1)
    gen = torch.Generator().manual_seed(4)
    g.images.copy_(torch.rand(N, H, W, 3, generator=gen))
    g.run_forward()
    torch.cuda.synchronize()
    dwop = [op for op in g.tape if isinstance(op, engine.MixConvOp)][0]
    xin = dwop.y.x.buf.float().cpu().requires_grad_(True)
    wdev = g.ps.view(g.ps.params['depthwise_conv2d/depthwise_kernel'], g.ps.bf16).float().cpu().reshape(k, k, C).requires_grad_(True)
    yref = F.conv2d(xin.permute(0, 3, 1, 2), wdev.permute(2, 0, 1).unsqueeze(1), padding=k // 2, groups=C).permute(0, 2, 3, 1)
    torch.testing.assert_close(dwop.y.buf.float().cpu(), yref.detach(), rtol=1e-2, atol=1e-2)
    yq = dwop.y.buf.float().cpu()
    mean, var = yq.mean((0, 1, 2)), yq.var((0, 1, 2), unbiased=False)
    zref = torch.relu((yq - mean) / torch.sqrt(var + 1e-5))
    torch.testing.assert_close(out.buf.float().cpu(), zref, rtol=2e-2, atol=2e-2)
    out.grad.copy_(torch.randn(out.shape, generator=gen).to(ACT()))
    go = out.grad.float().cpu()
    for f in g.bwd[:2]:                                      # the two ops on top: BN+ReLU apply backward, depthwise backward
        f()
    g.flush_wgrad()
    torch.cuda.synchronize()
    yref.backward(dwop.y.dy.float().cpu())
    dw = g.ps.view(g.ps.params['depthwise_conv2d/depthwise_kernel'], g.ps.grad).cpu().reshape(k, k, C)
    torch.testing.assert_close(dw, wdev.grad, rtol=1e-3, atol=1e-3 * max(1.0, wdev.grad.abs().max().item()))
    torch.testing.assert_close(dwop.y.x.grad.float().cpu(), xin.grad, rtol=1e-2, atol=2e-2)
    assert go.abs().sum() > 0


def test_pack_input(dev):
    from yolov3_tensorflow_amd import ops
    img = torch.rand(2, 6, 5, 3)
    out = torch.empty(2, 6, 5, 8, dtype=ACT(), device=dev)
    ops.pack_input(img.to(dev), out, 2 * 6 * 5, 3)
    ref = torch.zeros(2, 6, 5, 8)
    ref[..., :3] = img
    assert torch.equal(out.cpu(), bf(ref))


# ------------------------------------------------------------------------------------------------------------------ loss
ANCHORS = [[(0.06618181818181816, 0.1025177510694752), (0.18544278606965178, 0.13160367921287464), (0.13, 0.32733333333333337)],
           [(0.13, 0.32733333333333337), (0.303806787732042, 0.34370030784316496)],
           [(0.303806787732042, 0.34370030784316496), (0.4667050847457627, 0.5281262429095761),
            (0.7906945888923907, 0.7888860433597275)]]      # /root/reference/configs.py:37-41
LOSS_W = [(5, 5, 0.05, 3, 1), (8, 8, 0.05, 2, 1), (10, 10, 0.05, 2, 1)]     # configs.py:52


def make_labels(gen, N, T, class_num, empty_image=None):
    lab = -torch.ones(N, T, 5)
    for n in range(N):
        if n == empty_image:
            continue
        k = int(torch.randint(1, T + 1, (1,), generator=gen))
        wh = torch.rand(k, 2, generator=gen) * 0.5 + 0.04
        xy = torch.rand(k, 2, generator=gen) * (1 - wh) + wh / 2
        cls = torch.randint(0, max(class_num, 1), (k, 1), generator=gen).float()
        lab[n, :k] = torch.cat([xy, wh, cls], dim=1)
    return lab


LOSS_CASES = [
    dict(name='c13_rect', grid=[(40, 40), (20, 20), (10, 10)], C=13, N=3, T=8, rect=1464, focal=False, tiou=False, empty=None),
    dict(name='c80_norect_empty', grid=[(16, 16), (8, 8), (4, 4)], C=80, N=4, T=6, rect=-1, focal=False, tiou=False, empty=2),
    dict(name='c0_rect', grid=[(12, 16), (6, 8), (3, 4)], C=0, N=2, T=5, rect=0, focal=False, tiou=False, empty=None),
    dict(name='c20_focal_tiou', grid=[(16, 16), (8, 8), (4, 4)], C=20, N=3, T=7, rect=-1, focal=True, tiou=True, empty=None),
    # duplicates in one cell + anchor, a centre exactly on the right / bottom border (floor index = W / H: clamped), a
    # centre exactly on a cell boundary, a box as large as the image and a tiny one
    dict(name='c5_edges', grid=[(16, 16), (8, 8), (4, 4)], C=5, N=4, T=6, rect=-1, focal=False, tiou=False, empty=None, edges=True),
    # the logit of class 0 is -40 everywhere: softmax probability < eps = 1e-8, tf.clip_by_value (yolov3_decoder.py:189-191) clips it and passes
    # no gradient, so a target of class 0 costs -log(eps) and moves no class logit, while the other targets of the same cell + anchor still do
    dict(name='c6_saturated_class', grid=[(16, 16), (8, 8), (4, 4)], C=6, N=3, T=6, rect=-1, focal=False, tiou=False, empty=None, saturate=True),
]


def test_loss_focal_through_the_fp16_library(dev, fp16):
    """BASELINE.json configs[4]: the focal + TIoU case through libyolov3_amd_fp16.so with the static loss scale 1024 -- the float32 terms and
    d(logits) are unscaled, the float16 copy that feeds the backward pass is exactly fp16(1024 * d(logits))"""
    case = dict([c for c in LOSS_CASES if c['name'] == 'c20_focal_tiou'][0], grad_scale16=1024.0)
    test_loss_fwd_bwd_vs_oracle(dev, case)


@pytest.mark.parametrize('case', LOSS_CASES, ids=[c['name'] for c in LOSS_CASES])
def test_loss_fwd_bwd_vs_oracle(dev, case):
    """tolerance: loss terms 1e-4 relative (float32, different summation order); d(logits) 1e-3 relative (north_star);
    responsible-anchor indices exact."""
    from yolov3_tensorflow_amd import ops
    from oracle.loss import YOLOv3LossOracle
    gen = torch.Generator().manual_seed(42)
    grid, Cn, N, T = case['grid'], case['C'], case['N'], case['T']
    L = 5 + Cn
    B = [len(a) for a in ANCHORS]
    ldc = [ops.pad_channels(b * L) for b in B]
    raw = [torch.randn(N, h, w, b, L, generator=gen) * 0.8 for (h, w), b in zip(grid, B)]
    lab = make_labels(gen, N, T, Cn, case['empty'])
    if case.get('saturate'):
        for r in raw:
            r[..., 5] = -40.0
        lab[:, 0::2, 4] = torch.where(lab[:, 0::2, 4] >= 0, torch.zeros(()), lab[:, 0::2, 4])      # every other target is of class 0
        lab[1, 1] = lab[1, 0]
        lab[1, 1, 4] = 3.0                                                                        # same box, classes 0 and 3: one clipped, one live
    if case.get('edges'):
        lab[0, :3] = torch.tensor([[0.40, 0.55, 0.30, 0.20, 2.0], [0.40, 0.55, 0.30, 0.20, 2.0], [0.41, 0.56, 0.30, 0.20, 3.0]])
        lab[1, :2] = torch.tensor([[1.0, 1.0, 0.20, 0.30, 1.0], [0.5, 0.25, 0.10, 0.10, 0.0]])      # border; exact cell boundary at every head
        lab[1, 2:] = -1.0
        lab[2, :2] = torch.tensor([[0.5, 0.5, 1.0, 1.0, 4.0], [0.3, 0.3, 0.004, 0.004, 0.0]])
        lab[2, 2:] = -1.0
    orc = YOLOv3LossOracle(grid, Cn, ANCHORS, 0.5, LOSS_W, rectified_coord_num=case['rect'], rectified_loss_weight=[1.0, 0.5, 2.0],
                           is_focal_loss=case['focal'], focal_alpha=1.0, focal_gamma=2.0, is_tiou_recall=case['tiou'])
    rr = [r.clone().requires_grad_(True) for r in raw]
    total_ref = orc.loss_heads(lab.reshape(N, -1), rr)
    total_ref.backward()
    terms_ref = torch.zeros(6, 3)
    terms_ref[:orc.terms.shape[0]] = orc.terms

    cfg = ops.make_loss_config(grid, Cn, ANCHORS, 0.5, LOSS_W, ldc, T, rectified_coord_num=case['rect'],
                               rectified_loss_weight=[1.0, 0.5, 2.0], is_focal_loss=case['focal'], focal_alpha=1.0, focal_gamma=2.0,
                               is_tiou_recall=case['tiou'], grad_scale16=case.get('grad_scale16', 1.0))
    logits, dl, dlb = [], [], []
    for h in range(3):
        gh, gw = grid[h]
        t = torch.zeros(N, gh, gw, ldc[h])
        t[..., :B[h] * L] = raw[h].reshape(N, gh, gw, B[h] * L)
        logits.append(t.to(dev))
        dl.append(torch.zeros(N, gh, gw, ldc[h], device=dev))
        dlb.append(torch.zeros(N, gh, gw, ldc[h], dtype=ACT(), device=dev))
    ws = torch.empty(ops.loss_workspace_bytes(cfg, N), dtype=torch.uint8, device=dev)
    cur = torch.zeros(1, dtype=torch.int32, device=dev)
    terms = torch.empty(6, 3, device=dev)
    total = torch.empty(1, device=dev)
    assign = torch.empty(N, T, 3, dtype=torch.int32, device=dev)
    riou = torch.empty(N, T, 3, device=dev)
    ops.loss_fwd_bwd(cfg, N, N, logits, lab.to(dev), cur, terms, total, ws, dlogits=dl, dlogits_bf16=dlb, assign_out=assign,
                     resp_iou_out=riou)
    torch.cuda.synchronize()
    # a centre exactly on the border makes the reference's own arithmetic produce NaN in that head's xy term (0 * log 0): the kernel must
    # reproduce it in the same place, not hide it
    nan_ok = bool(case.get('edges'))
    if nan_ok:
        assert torch.isnan(terms_ref).sum() == 1 and torch.equal(torch.isnan(terms.cpu()), torch.isnan(terms_ref))
    torch.testing.assert_close(terms.cpu(), terms_ref, rtol=1e-4, atol=1e-5, equal_nan=nan_ok)
    torch.testing.assert_close(total.cpu()[0], total_ref.detach(), rtol=1e-4, atol=1e-5, equal_nan=nan_ok)
    # rectified counter semantics (yolov3_loss.py:125-130,152)
    expect_cur = N if (case['rect'] >= 0) else 0
    assert int(cur.cpu()[0]) == expect_cur
    # responsible-anchor indices: exact
    a = assign.cpu().numpy()
    for n in range(N):
        valid_t = [t for t in range(T) if lab[n, t, 0] >= 0]
        for h in range(3):
            got = sorted(int(a[n, t, h]) for t in valid_t if a[n, t, h] >= 0)
            ref = orc.last_assign[n][h]
            W_, B_ = grid[h][1], B[h]
            exp = sorted(int((r * W_ + c) * B_ + k) for r, c, k in ref.tolist())
            assert got == exp, (n, h, got, exp)
        for t in range(T):
            if lab[n, t, 0] < 0:
                assert (a[n, t] == -1).all()
    # gradients
    for h in range(3):
        gh, gw = grid[h]
        gref = rr[h].grad.reshape(N, gh, gw, B[h] * L)
        got = dl[h].cpu()
        assert torch.count_nonzero(got[..., B[h] * L:]) == 0
        denom = gref.abs().max().item()
        if nan_ok:
            assert torch.equal(torch.isnan(got[..., :B[h] * L]), torch.isnan(gref))
            denom = torch.nan_to_num(gref).abs().max().item()
        torch.testing.assert_close(got[..., :B[h] * L], gref, rtol=1e-3, atol=1e-5 * max(denom, 1.0), equal_nan=nan_ok)
        torch.testing.assert_close(dlb[h].float().cpu(), (got * case.get('grad_scale16', 1.0)).to(ACT()).float(), rtol=0, atol=0, equal_nan=nan_ok)
    # second call: counter advanced -> rectified term switches off when current_num > rectified_coord_num
    ops.loss_fwd_bwd(cfg, N, N, logits, lab.to(dev), cur, terms, total, ws, dlogits=dl)
    total2 = orc.loss_heads(lab.reshape(N, -1), [r.clone() for r in raw])
    torch.testing.assert_close(total.cpu()[0], total2.detach(), rtol=1e-4, atol=1e-5, equal_nan=nan_ok)


# ------------------------------------------------------------------------------------------------------------------ RAdam
def test_radam_l2_step_vs_oracle(dev):
    """10 steps across the rho_t = 5 switch (t = 6 for beta_2 = .999) with L2 on one segment; tolerance 1e-6 absolute /
    1e-5 relative (float32 elementwise; powf vs numpy power differ by ulps)."""
    from yolov3_tensorflow_amd import ops
    from oracle.optim import RAdamOracle
    rng = np.random.default_rng(0)
    n = 1024
    p0 = rng.normal(size=n).astype(np.float32)
    lam = np.zeros(n // 256, dtype=np.float32)
    lam[1], lam[2] = 5e-4, 1e-5
    lam_e = np.repeat(lam, 256)
    orc = RAdamOracle(lr=1e-3, scalar_dtype=np.float64)   # scalar schedule in float64 on both sides (see optim.hip)
    orc32 = RAdamOracle(lr=1e-3, scalar_dtype=np.float32)
    pr = p0.copy()
    d = lambda a: torch.from_numpy(a).to(dev)
    p, m, v = d(p0.copy()), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=ACT(), device=dev)
    sched = torch.tensor([1e-3, 0, 0, 0], device=dev)
    it = torch.zeros(1, dtype=torch.int64, device=dev)
    l2p = torch.empty(ops.radam_l2_blocks(n), device=dev)
    l2out = torch.empty(1, device=dev)
    for step in range(10):
        g = rng.normal(size=n).astype(np.float32)
        l2_ref = float((lam_e * pr * pr).sum())
        rho, lr_t = orc.step([pr], [g + 2 * lam_e * pr])
        rho32, lr_t32 = orc32.schedule()
        gd = d(g.copy())
        ops.radam_schedule(sched, it, 0.9, 0.999, 0.0, 1.0)
        ops.radam_l2_step(p, gd, m, v, d(lam), n, sched, 0.9, 0.999, 1e-8, 1.0, True, params_bf16=pb, l2_partial=l2p)
        ops.sum_partials(l2p, l2p.numel(), None, l2out)
        s = sched.cpu().numpy()
        assert int(it.cpu()[0]) == step + 1
        assert (s[3] == 1.0) == (rho >= 5.0) and (step + 1 >= 6) == (s[3] == 1.0)
        np.testing.assert_allclose(s[1], lr_t, rtol=1e-5)
        np.testing.assert_allclose(s[1], lr_t32, rtol=3e-2)    # float32 chain as in the reference: ill-conditioned, ~1 %
        np.testing.assert_allclose(p.cpu().numpy(), pr, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(l2out.cpu().numpy()[0], l2_ref, rtol=1e-5)
        assert torch.count_nonzero(gd) == 0
        assert torch.equal(pb.cpu(), p.cpu().to(ACT()))


@pytest.mark.parametrize('kind', ['radam_amsgrad_decay', 'adam_amsgrad', 'adam_decay', 'sgd_nesterov', 'sgd_momentum_decay'])
def test_optimizer_variants_vs_oracle(dev, kind):
    """the branches of the multi-tensor kernel the default RAdam(lr=1e-3) does not take: RAdam's AMSGrad / decay (radam.py:61-64,91-94) and the
    reference trainer's other two optimizers, keras SGD(momentum=0.95, nesterov=True) and Adam(amsgrad=True) (trainer.py:70-73), 12 steps
    each against oracle/optim.py with gradients whose magnitude falls after step 4 (so vhat != v); L2 on one segment."""
    from yolov3_tensorflow_amd import ops
    from oracle.optim import RAdamOracle, AdamOracle, SGDOracle
    rng = np.random.default_rng(1)
    n = 1024
    p0 = rng.normal(size=n).astype(np.float32)
    lam = np.zeros(n // 256, dtype=np.float32)
    lam[1] = 5e-4
    lam_e = np.repeat(lam, 256)
    d = lambda a: torch.from_numpy(a).to(dev)
    if kind == 'radam_amsgrad_decay':
        orc, k, b1, b2, eps, decay, ams = RAdamOracle(lr=1e-3, decay=0.05, amsgrad=True, scalar_dtype=np.float64), 0, 0.9, 0.999, 1e-8, 0.05, True
    elif kind == 'adam_amsgrad':
        orc, k, b1, b2, eps, decay, ams = AdamOracle(lr=2e-4, amsgrad=True), 1, 0.9, 0.999, 1e-8, 0.0, True
    elif kind == 'adam_decay':
        orc, k, b1, b2, eps, decay, ams = AdamOracle(lr=2e-4, decay=0.1), 1, 0.9, 0.999, 1e-8, 0.1, False
    elif kind == 'sgd_nesterov':
        orc, k, b1, b2, eps, decay, ams = SGDOracle(lr=2e-4, momentum=0.95, nesterov=True), 2, 0.95, 0.0, 0.0, 0.0, False
    else:
        orc, k, b1, b2, eps, decay, ams = SGDOracle(lr=2e-4, momentum=0.9, decay=0.1), 3, 0.9, 0.0, 0.0, 0.1, False
    pr = p0.copy()
    p, m, v = d(p0.copy()), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    vhat = torch.zeros(n, device=dev) if ams else None
    sched = torch.tensor([orc.lr, 0, 0, 0], device=dev, dtype=torch.float32)
    it = torch.zeros(1, dtype=torch.int64, device=dev)
    for step in range(12):
        g = (rng.normal(size=n) * (1.0 if step < 4 else 0.1)).astype(np.float32)
        orc.step([pr], [g + 2 * lam_e * pr])
        if k == 0:
            ops.radam_schedule(sched, it, b1, b2, decay, 1.0)
        else:
            ops.optimizer_schedule(sched, it, k, b1, b2, decay)
        ops.radam_l2_step(p, d(g.copy()), m, v, d(lam), n, sched, b1, b2, eps, 1.0, True, vhat=vhat)
        assert int(it.cpu()[0]) == step + 1
        np.testing.assert_allclose(p.cpu().numpy(), pr, rtol=1e-5, atol=1e-6, err_msg='%s step %d' % (kind, step))
    if ams:
        np.testing.assert_allclose(vhat.cpu().numpy(), orc.vhat[0], rtol=1e-5, atol=1e-12)
        assert (vhat.cpu().numpy() > v.cpu().numpy()).mean() > 0.5          # the maximum was the live branch


def test_radam_skips_and_counts_nonfinite_gradients(dev):
    """an inf / NaN gradient element (fp16 overflow in the backward pass) is not applied -- that element takes the step a zero gradient
    would give -- and the wave is counted; every other element updates as usual; without the counter argument the old behaviour remains"""
    from yolov3_tensorflow_amd import ops
    n = 1024
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g)
    bad = grad.clone()
    bad[5], bad[300], bad[301] = float('inf'), float('nan'), float('-inf')           # waves 0 and 4 (64 lanes x 4 elements per wave trip)
    clean = grad.clone()
    clean[5] = clean[300] = clean[301] = 0.0
    lam = torch.zeros(n // 256)

    def step(gr, counter):
        p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sched = torch.tensor([1e-3, 0, 0, 0], device=dev)
        it = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.radam_schedule(sched, it, 0.9, 0.999, 0.0, 1.0)
        ops.radam_l2_step(p, gr.clone().to(dev), m, v, lam.to(dev), n, sched, 0.9, 0.999, 1e-8, 1.0, True, nonfinite=counter)
        torch.cuda.synchronize()
        return p.cpu(), m.cpu()

    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    p_bad, m_bad = step(bad, cnt)
    p_ref, m_ref = step(clean, None)
    assert int(cnt.item()) == 2
    assert torch.equal(p_bad, p_ref) and torch.equal(m_bad, m_ref) and torch.isfinite(p_bad).all()
    cnt.zero_()
    step(grad, cnt)
    assert int(cnt.item()) == 0
    p_unguarded, _ = step(bad, None)
    assert not torch.isfinite(p_unguarded).all()


# ---- the fp16 build (libyolov3_amd_fp16.so: same sources, IEEE half elements, f16 MFMA) through the same checks ----
FP16_CONV = [CONV_CASES[i] for i in (0, 1, 2, len(CONV_CASES) - 5, len(CONV_CASES) - 4, len(CONV_CASES) - 2, len(CONV_CASES) - 1)]


@pytest.mark.parametrize('case', FP16_CONV, ids=[str(c) for c in FP16_CONV])
def test_fp16_build_conv_fwd_dgrad_wgrad(dev, fp16, case):
    from yolov3_tensorflow_amd import _lib
    assert _lib.load().yolo_abi_dtype() == 1
    test_conv_fwd_dgrad_wgrad(dev, case)


def test_fp16_build_strip_and_wgrad_strip(dev, fp16):
    test_strip_conv_variants_match_implicit_gemm(dev, 128, 64)
    test_wgrad_strip_matches_generic(dev, (3, 21, 19, 128, 256), 2)
    test_bn_act_fwd_bwd(dev, 'res_bn')
    test_bn_pool_relu_fwd_bwd(dev)
    test_mixconv_fwd_dgrad_wgrad(dev, 128, 2, 11, 9)


def _acc_totals(acc, Q, C):
    """decode an exact accumulator block (common.h yolo_acc_*): [buckets][Q][2 limbs][C] int64 + flag -> (Q, C) float64 totals, flag"""
    nb = (acc.numel() - 2) // (Q * 2 * C)
    a = acc[:nb * Q * 2 * C].reshape(nb, Q, 2, C).sum(0).cpu()
    return a[:, 0].double() * 2.0 ** -20 + a[:, 1].double() * 2.0 ** -60, int(acc[nb * Q * 2 * C].item())


@pytest.mark.parametrize('case', [(4, 52, 52, 128, 64, True), (3, 104, 104, 64, 64, False), (6, 13, 13, 256, 512, True), (5, 26, 30, 64, 128, False)])
def test_batchnorm_statistics_through_exact_accumulators(dev, case):
    """conv2d_fwd(stat_acc=) + bn_finalize_act_fwd_acc against the statistics-rows path (conv2d_fwd rows + bn_finalize_act_fwd): the block's
    totals ARE the sums of the rows (same per-workgroup float32 sums, added exactly), so scale / shift / mean / rstd agree to float32 rounding and
    the activations are the same bits; the backward twin (conv2d_dgrad(bn=dict(acc=...)) + bn_bwd_finalize_apply_acc) likewise.  Integer
    atomics are associative: two launches leave bit-identical blocks."""
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    M = N * H * W
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    x = bf(torch.randn(N, H, W, Cin, generator=g)).to(dev)
    w = bf(torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(dev)
    res = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.randn(Cout, generator=g) * 0.1).to(dev)
    # rows path
    rows = ops.conv2d_stat_rows(p)
    ssum, ssq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
    y0 = torch.empty(N, H, W, Cout, dtype=ACT(), device=dev)
    ops.conv2d_fwd(p, x, w, y0, stat_sum=ssum, stat_sq=ssq)
    # accumulator path
    acc = torch.zeros(ops.acc_words(2, Cout), dtype=torch.int64, device=dev)
    y1 = torch.empty_like(y0)
    ops.conv2d_fwd(p, x, w, y1, stat_acc=acc)
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    tot, flag = _acc_totals(acc, 2, Cout)
    assert flag == 0
    torch.testing.assert_close(tot[0], ssum.double().sum(0).cpu(), rtol=1e-12, atol=1e-9)
    torch.testing.assert_close(tot[1], ssq.double().sum(0).cpu(), rtol=1e-12, atol=1e-9)
    acc2 = torch.zeros_like(acc)
    ops.conv2d_fwd(p, x, w, y1, stat_acc=acc2)
    torch.cuda.synchronize()
    assert torch.equal(acc, acc2), 'integer accumulation must not depend on the arrival order'

    def state():
        return [torch.zeros(Cout, device=dev) for _ in range(4)] + [torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)]
    outs = []
    for use_acc in (False, True):
        scale, shift, mean, rstd, mm, mv = state()
        out = torch.empty_like(y0)
        mask = torch.zeros(M * Cout // 8, dtype=torch.uint8, device=dev) if relu else None
        if use_acc:
            ops.bn_finalize_act_fwd_acc(acc, Cout, M, gamma, beta, 1e-5, 0.9, mm, mv, scale, shift, mean, rstd, y0, out, M, relu, res=res, mask=mask)
        else:
            if rows > 4096:
                pytest.skip('rows path of the merged launch is not meant for this many rows')
            ops.bn_finalize_act_fwd(ssum.view(-1), ssq.view(-1), rows, Cout, Cout, M, gamma, beta, 1e-5, 0.9, mm, mv, scale, shift, mean, rstd, y0, out,
                                    M, relu, res=res, mask=mask)
        torch.cuda.synchronize()
        outs.append((out.float().cpu(), [t.cpu() for t in (scale, shift, mean, rstd, mm, mv)], None if mask is None else mask.cpu()))
    for a, b in zip(outs[0][1], outs[1][1]):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
    assert float((outs[0][0] != outs[1][0]).float().mean()) < 1e-3          # same bits wherever the float32 constants agree
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=2 ** -7, atol=1e-3)

    # non-finite input: the flag is raised and the unit's outputs are NaN (as the float path's would be)
    xb = x.clone()
    xb[0, 0, 0, 0] = float('inf')
    accb = torch.zeros_like(acc)
    ops.conv2d_fwd(p, xb, w, y1, stat_acc=accb)
    scale, shift, mean, rstd, mm, mv = state()
    out = torch.empty_like(y0)
    ops.bn_finalize_act_fwd_acc(accb, Cout, M, gamma, beta, 1e-5, 0.9, mm, mv, scale, shift, mean, rstd, y1, out, M, False)
    torch.cuda.synchronize()
    assert _acc_totals(accb, 2, Cout)[1] != 0 and bool(torch.isnan(scale).any())

    # ---- backward twin ----
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = bf(torch.randn(N, H, W, Cout, generator=g)).to(dev)
    yb = bf(torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(dev)
    mean_b, rstd_b = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    a1 = (torch.rand(Cin, generator=g) + 0.5).to(dev)
    maskb = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    prow = ops.conv2d_dgrad_bn_rows(p)
    partial = torch.zeros(prow, 3, Cin, device=dev)
    dx0 = torch.empty(N, H, W, Cin, dtype=ACT(), device=dev)
    ops.conv2d_dgrad(p, dy, w_dg, dx0, bn=dict(mask=maskb, y=yb, mean=mean_b, rstd=rstd_b, partial=partial))
    accq = torch.zeros(ops.acc_words(3, Cin), dtype=torch.int64, device=dev)
    dx1 = torch.empty_like(dx0)
    ops.conv2d_dgrad(p, dy, w_dg, dx1, bn=dict(mask=maskb, y=yb, mean=mean_b, rstd=rstd_b, partial=None, acc=accq))
    torch.cuda.synchronize()
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
    totb, flagb = _acc_totals(accq, 3, Cin)
    assert flagb == 0
    ref = partial.double().sum(0).cpu()
    scale_b = float(ref.abs().max())
    torch.testing.assert_close(totb[0], ref[0], rtol=1e-10, atol=1e-9 * max(scale_b, 1.0))
    torch.testing.assert_close(totb[1], ref[1], rtol=1e-10, atol=1e-9 * max(scale_b, 1.0))
    res_b = []
    for use_acc in (False, True):
        dgam, dbet, k1, k2 = [torch.zeros(Cin, device=dev) for _ in range(4)]
        dyo = torch.empty(M, Cin, dtype=ACT(), device=dev)
        if use_acc:
            ops.bn_bwd_finalize_apply_acc(accq, Cin, M, dgam, dbet, k1, k2, dx0, yb, a1, mean_b, rstd_b, M, dyo)
        else:
            if prow > 4096:
                continue
            ops.bn_bwd_finalize_apply(partial.view(-1), prow, Cin, M, dgam, dbet, k1, k2, dx0, yb, a1, mean_b, rstd_b, M, dyo)
        torch.cuda.synchronize()
        res_b.append((dyo.float().cpu(), dgam.cpu(), dbet.cpu(), k1.cpu(), k2.cpu()))
    if len(res_b) == 2:
        for a, b in zip(res_b[0][1:], res_b[1][1:]):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-6 * max(scale_b, 1.0))
        torch.testing.assert_close(res_b[0][0], res_b[1][0], rtol=2 ** -7, atol=1e-3)
