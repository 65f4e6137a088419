"""Size-independent properties at BASELINE.json's full size (configs[1]: 416x416, batch 32, 80 classes), where the CPU oracle would take
minutes: adjoint identities that tie the three convolution kernels to each other, BatchNorm invariants, and whole-step invariants.

For a linear map Y = conv(X; W):   <Y, dY> == <X, dgrad(dY)> == <W, wgrad(X, dY)>   (all three kernels, any size, no reference needed;
tolerance = bf16 rounding of the stored Y / dX, 2^-9 relative per element, accumulated in float64 on the host side)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYERS = [  # the actual layer shapes of ResNet18-YOLOv3 at 416x416, batch 32: (H, W, Cin, Cout, k, stride, padding)
    (104, 104, 64, 64, 3, 1, 'same'),
    (104, 104, 64, 128, 3, 2, 'same'),
    (52, 52, 128, 128, 3, 1, 'same'),
    (26, 26, 256, 256, 3, 1, 'same'),
    (13, 13, 512, 512, 3, 1, 'same'),
    (52, 52, 64, 128, 1, 2, 'valid'),
    (13, 13, 512, 256, 1, 1, 'same'),
]


@pytest.mark.parametrize('layer', LAYERS, ids=[str(l) for l in LAYERS])
def test_conv_adjoint_identities_full_size(layer):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd import ops
    H, W, Cin, Cout, k, s, pad = layer
    N = 32
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(3)
    p = ops.conv_problem(N, H, W, Cin, Cout, k, s, pad)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(k * k * Cin)).to(torch.bfloat16).to(dev)
    dy = torch.randn(N, p.Ho, p.Wo, Cout, generator=g).to(torch.bfloat16).to(dev)
    y32 = torch.empty(N, p.Ho, p.Wo, Cout, dtype=torch.float32, device=dev)
    ops.conv2d_fwd(p, x, w, y32)                                         # float32 output: no rounding on this side
    ybf = torch.empty(N, p.Ho, p.Wo, Cout, dtype=torch.bfloat16, device=dev)
    rows = ops.conv2d_stat_rows(p)
    ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
    ops.conv2d_fwd(p, x, w, ybf, stat_sum=ss, stat_sq=sq)
    wd = torch.empty(Cin, k, k, Cout, dtype=torch.bfloat16, device=dev)
    ops.repack_dgrad_weights(w, wd, Cout, k, k, Cin)
    dx = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=dev)
    ops.conv2d_dgrad(p, dy, wd, dx)
    dw = torch.zeros(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(p, x, dy, dw)
    torch.cuda.synchronize()
    dot = lambda a, b: float((a.double() * b.double()).sum().item())
    a_fwd = dot(y32, dy)
    a_dg = dot(x, dx)
    a_wg = dot(w, dw)
    scale = float(y32.double().norm().item() * dy.double().norm().item())
    assert abs(a_fwd - a_wg) <= 2e-4 * scale, (a_fwd, a_wg)              # wgrad accumulates in float32: only summation order
    assert abs(a_fwd - a_dg) <= 2e-3 * scale, (a_fwd, a_dg)              # dX is stored in bf16
    # bf16 output == rounded float32 output; the BatchNorm partial sums are the column sums of what was stored
    dev_ = (ybf.float() - y32).abs()
    tol_ = 2 ** -7 * float(y32.abs().max())
    bad = torch.nonzero(dev_ > tol_)
    assert bad.numel() == 0, 'bf16 output differs from the rounded float32 output at %d elements, first (n, h, w, c): %s, values %s vs %s' % (
        bad.shape[0], bad[:8].tolist(), [float(ybf[tuple(i)]) for i in bad[:8].tolist()], [float(y32[tuple(i)]) for i in bad[:8].tolist()])
    torch.testing.assert_close(ss.sum(0), ybf.float().sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-1)
    torch.testing.assert_close(sq.sum(0), (ybf.float() ** 2).sum(dim=(0, 1, 2)), rtol=1e-3, atol=1e-1)
    # linearity of the data-gradient accumulate path: dgrad(dy) + dgrad(dy) == 2 dgrad(dy)
    dx2 = dx.clone()
    ops.conv2d_dgrad(p, dy, wd, dx2, accumulate=True)
    torch.testing.assert_close(dx2.float(), 2 * dx.float(), rtol=2e-2, atol=2e-2)


def test_training_step_invariants_full_size():
    """one full-size step: finite logits/loss, total == sum of the 18 terms, BatchNorm outputs have mean beta / variance gamma^2, every
    variable receives a finite non-zero gradient, the weights move and the loss changes."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import bench
    from yolov3_tensorflow_amd import engine
    model, loss, opt, grids = bench.build_model('resnet-18', 416, 416, 32, 80, torch.device('cuda:0'))
    images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
    model.stage_batch(images, labels)
    model.g.training = True
    model._fwd_bwd()
    torch.cuda.synchronize()
    for h in model.heads:
        assert torch.isfinite(h.buf).all()
    total, terms = float(loss.total.item()), loss.terms.cpu().numpy()
    assert np.isfinite(total) and abs(total - terms.sum()) <= 1e-5 * abs(total)
    assert (terms[5] == 0).all()                                          # rectified_coord_num = -1 in the bench configuration
    # BatchNorm invariant on a mid-network conv->BN->ReLU unit: recompute BN output statistics from the stored conv output
    ap = [op for op in model.g.tape if isinstance(op, engine.ApplyOp) and op.m_bn is not None and op.o_src is None][5]
    y = ap.m_src.buf.float().reshape(-1, ap.C)
    z = y * ap.m_bn.scale + ap.m_bn.shift
    ps = model.g.ps
    torch.testing.assert_close(z.mean(0), ps.view(ap.m_bn.beta), rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(z.var(0, unbiased=False), ps.view(ap.m_bn.gamma) ** 2, rtol=2e-3, atol=2e-3)
    assert torch.equal(ap.out.buf.float().reshape(-1, ap.C), torch.relu(z).to(torch.bfloat16).float()) or \
        float((ap.out.buf.float().reshape(-1, ap.C) - torch.relu(z)).abs().max()) < 0.05
    grad = ps.grad
    assert torch.isfinite(grad).all()
    for p in ps.params.values():
        gp = grad[p.offset:p.offset + p.numel]
        assert float(gp.abs().max()) > 0, p.name
    w_before = ps.flat.clone()
    model._update()
    torch.cuda.synchronize()
    assert float((ps.flat - w_before).abs().max()) > 0 and torch.count_nonzero(ps.grad) == 0      # weights moved; gradient zeroed
    assert torch.equal(ps.bf16, ps.flat.to(torch.bfloat16))                                       # compute copy refreshed
    l0 = float(model.loss_value.item())
    model.run_step()
    assert np.isfinite(float(model.loss_value.item())) and float(model.loss_value.item()) != l0
