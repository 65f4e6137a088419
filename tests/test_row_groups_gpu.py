"""Two-level partial rows (conv_common.h rows_fold; yolo_conv2d_fwd_g / yolo_conv2d_dgrad_bn_g): the convolution epilogues fold their per-tile
statistics rows in groups -- the workgroup whose arrival completes a group sums that group's raw rows in row order -- so that the BatchNorm
kernels can derive their constants from a few rows in their own prologue (no finalize launch).  Checked here through the C-ABI, for every
kernel family that writes rows (implicit GEMM, strip, 32x32x16 strip, streaming):
 * the convolution output / masked gradient is bit-identical to the plain entry point's;
 * raw rows [P, P + R) are the plain entry point's rows bit for bit, every group row is the float32 sum of its raw rows IN ROW ORDER bit for bit
   (so it does not depend on which workgroup arrived last), the arrival counters are zero again after the launch, a second launch reproduces
   the buffer bit for bit;
 * yolo_bn_finalize_act_fwd / yolo_bn_bwd_finalize_apply on the group rows (their streaming form) agree with finalize + apply on the raw rows.
"""
import math
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


def seq_sum(rows):
    """float32 sum of the rows in row order (what the folding workgroup computes)"""
    t = np.zeros(rows.shape[1:], dtype=np.float32)
    for r in rows:
        t = (t + r).astype(np.float32)
    return t


CASES = [
    # N, H, W, Cin, Cout, k, family the forward launch takes
    (32, 52, 52, 128, 128, 3, 's32'),
    (16, 26, 26, 256, 256, 3, 'strip'),
    (8, 26, 26, 256, 128, 1, 'igemm'),
    (16, 104, 104, 64, 64, 3, 'stream'),
    (3, 21, 19, 64, 128, 3, None),           # ragged last tile and last group
]


@pytest.mark.parametrize('case', CASES, ids=str)
def test_forward_statistics_rows_fold_in_groups(dev, case):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, k, fam = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT()).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(k * k * Cin)).to(ACT()).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, k, 1, 'same')
    if fam is not None:
        assert ops.conv2d_fwd_plan(p)['family'] == fam
    R = ops.conv2d_stat_rows(p)
    lay = ops.conv2d_stat_group_layout(p)
    P, G = lay['groups'], lay['group']
    assert lay['raw_rows'] == R and G in (16, 32, 64) and P == (R + G - 1) // G and P <= 128 and lay['alloc_rows'] > P + R
    y0 = torch.empty(N, H, W, Cout, dtype=ACT(), device=dev)
    s0, q0 = torch.zeros(R, Cout, device=dev), torch.zeros(R, Cout, device=dev)
    ops.conv2d_fwd(p, x, w, y0, stat_sum=s0, stat_sq=q0)
    bufs = []
    for _ in range(2):
        y1 = torch.empty_like(y0)
        st = torch.zeros(2, lay['alloc_rows'], Cout, device=dev) if not bufs else bufs[0][1]      # the second launch reuses the buffer as it was left
        ops.conv2d_fwd(p, x, w, y1, stat_sum=st[0], stat_sq=st[1], grouped=True)
        torch.cuda.synchronize()
        bufs.append((y1, st, st.clone()))
    y1, st, snap = bufs[0]
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    assert torch.equal(bufs[1][2], snap), 'a second launch must reproduce the row buffer bit for bit'
    for arr, plain in ((snap[0], s0), (snap[1], q0)):
        raw = arr[P:P + R]
        assert torch.equal(raw, plain)
        raw_h, grp_h = raw.cpu().numpy(), arr[:P].cpu().numpy()
        for gi in range(P):
            np.testing.assert_array_equal(grp_h[gi], seq_sum(raw_h[gi * G:min(R, (gi + 1) * G)]))
    assert int(snap[0][P + R:].view(torch.int32).abs().sum()) == 0, 'arrival counters must be zero after the launch'
    assert int(snap[1][P + R:].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('case', [(32, 52, 52, 128, 128, True), (8, 104, 104, 64, 64, True), (16, 26, 26, 256, 256, False), (3, 21, 19, 128, 64, True)], ids=str)
def test_dgrad_bn_partial_rows_fold_in_groups(dev, case):
    from yolov3_tensorflow_amd import ops
    N, H, W, Cin, Cout, acc = case
    g = torch.Generator().manual_seed(12)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(ACT()).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT(), device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT()).to(dev)
    base = torch.randn(N, H, W, Cin, generator=g).to(ACT()).to(dev)
    M = N * H * W
    yb = (torch.randn(M, Cin, generator=g) * 1.3 + 0.2).to(ACT()).to(dev)
    mean, rstd = (torch.randn(Cin, generator=g) * 0.2).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    R = ops.conv2d_dgrad_bn_rows(p)
    lay = ops.conv2d_dgrad_bn_group_layout(p)
    P, G = lay['groups'], lay['group']
    assert lay['raw_rows'] == R and G > 0 and P == (R + G - 1) // G and P <= 128
    plain = torch.zeros(R, 3, Cin, device=dev)
    dx0 = base.clone()
    ops.conv2d_dgrad(p, dy, w_dg, dx0, accumulate=acc, bn=dict(mask=mask, y=yb, mean=mean, rstd=rstd, partial=plain))
    part = torch.zeros(lay['alloc_rows'], 3, Cin, device=dev)
    snaps = []
    for _ in range(2):
        dx1 = base.clone()
        ops.conv2d_dgrad(p, dy, w_dg, dx1, accumulate=acc, bn=dict(mask=mask, y=yb, mean=mean, rstd=rstd, partial=part, grouped=True))
        torch.cuda.synchronize()
        snaps.append(part.clone())
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
    assert torch.equal(snaps[0], snaps[1])
    raw = snaps[0][P:P + R]
    assert torch.equal(raw[:, :2], plain[:, :2])
    raw_h, grp_h = raw.cpu().numpy(), snaps[0][:P].cpu().numpy()
    for gi in range(P):
        np.testing.assert_array_equal(grp_h[gi][:2], seq_sum(raw_h[gi * G:min(R, (gi + 1) * G)])[:2])
    assert int(snaps[0][P + R:].view(torch.int32).abs().sum()) == 0


def test_stem_and_stride2_keep_plain_rows(dev):
    from yolov3_tensorflow_amd import ops
    stem = ops.conv_problem(4, 64, 96, 8, 64, 3, 2, 'same')
    lay = ops.conv2d_stat_group_layout(stem)
    assert lay['group'] == 0 and lay['groups'] == lay['alloc_rows'] == lay['raw_rows'] == ops.conv2d_stat_rows(stem)
    s2 = ops.conv_problem(4, 52, 52, 64, 128, 3, 2, 'same')
    lay = ops.conv2d_dgrad_bn_group_layout(s2)
    assert lay['group'] == 0 and lay['groups'] == lay['alloc_rows'] == ops.conv2d_dgrad_bn_rows(s2)
