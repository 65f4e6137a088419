"""BASELINE.json configs[3] and configs[4] at their FULL sizes (one GPU's share), where the CPU oracle would take minutes: size-independent
invariants of a whole training step -- and the pieces of configs[4] (ResNet18-v2, float16 build, focal loss, static loss scale 1024) that the
224x224 oracle comparison of tests/test_train_step_gpu.py cannot see at that size.
    configs[3]  MixNet18-YOLOv3 416x416, batch 32, bf16         (MixConv depthwise 3/5/7/9)
    configs[4]  ResNet18-v2-YOLOv3 608x608, fp16, focal loss on, batch 128 global = 16 per GPU"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FULL = [('mixnet-18', 416, 32, 'bfloat16', False), ('resnet-18-v2', 608, 16, 'float16', True)]


@pytest.mark.parametrize('backbone,size,batch,dtype,focal', FULL, ids=['configs3_mixnet18_416_b32_bf16', 'configs4_resnet18v2_608_b16_fp16_focal'])
def test_full_size_step_invariants(backbone, size, batch, dtype, focal):
    """finite logits / loss, total == sum of the 18 terms, every variable receives a finite non-zero gradient (after un-scaling: no fp16
    overflow, nothing flushed to zero by the loss scale), BatchNorm outputs have mean beta / variance gamma^2 on a mid-network unit, the
    update moves the weights, refreshes the 16-bit copy and zeroes the gradient, two more steps stay finite and the loss falls at 1e-5"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import bench
    from yolov3_tensorflow_amd import backend, engine
    backend.set_compute_dtype(dtype)
    try:
        dev = torch.device('cuda:0')
        model, loss, opt, grids = bench.build_model(backbone, size, size, batch, 80, dev, focal=focal)
        assert loss.is_focal_loss == focal and (loss.focal_alpha, loss.focal_gamma) == (1.0, 2.0)
        opt.lr = 1e-5
        images, labels = bench.synthetic_batch(batch, size, size, 80, 0)
        model.stage_batch(images, labels)
        model.g.training = True
        model._fwd_bwd()
        torch.cuda.synchronize()
        act = backend.torch_dtype()
        for h in model.heads:
            assert torch.isfinite(h.buf).all() and h.dy.dtype == act
        total, terms = float(loss.total.item()), loss.terms.cpu().numpy()
        assert np.isfinite(total) and abs(total - terms.sum()) <= 1e-5 * abs(total)
        S = backend.loss_scale()
        assert S == (1024.0 if dtype == 'float16' else 1.0)
        ps = model.g.ps
        grad = ps.grad / S
        assert torch.isfinite(grad).all()
        for p in ps.params.values():
            gp = grad[p.offset:p.offset + p.numel]
            assert float(gp.abs().max()) > 0, p.name
        # BatchNorm invariant on a conv -> BN -> ReLU unit in the middle of the network
        aps = [op for op in model.g.tape if isinstance(op, engine.ApplyOp) and op.m_bn is not None and op.o_src is None and op.m_src.kind == 'conv'
               and len(op.m_bn.parts) == 1]
        ap = aps[len(aps) // 2]
        y = ap.m_src.buf.float().reshape(-1, ap.C)
        z = y * ap.m_bn.scale + ap.m_bn.shift
        torch.testing.assert_close(z.mean(0), ps.view(ap.m_bn.beta), rtol=1e-3, atol=3e-3)
        torch.testing.assert_close(z.var(0, unbiased=False), ps.view(ap.m_bn.gamma) ** 2, rtol=3e-3, atol=3e-3)
        if backbone == 'mixnet-18':
            # MixConv: the 4 kernel sizes write disjoint channel groups of ONE tensor (slice / concat are addressing): each group's output
            # depends only on its own input channels -- zeroing the other groups' weights must leave it bit-identical
            mix = [op for op in model.g.tape if isinstance(op, engine.MixConvOp)][3]
            before = mix.y.buf.clone()
            sp = list(mix.y.mp.split)
            keep = mix.w[1].clone()
            for i in (0, 2, 3):
                mix.w[i].zero_()
            mix.forward()
            torch.cuda.synchronize()
            assert torch.equal(mix.y.buf[..., sp[1]:sp[2]], before[..., sp[1]:sp[2]]) and torch.equal(mix.w[1], keep)
            assert float(mix.y.buf[..., :sp[1]].abs().max()) == 0.0 and float(mix.y.buf[..., sp[2]:].abs().max()) == 0.0
            from yolov3_tensorflow_amd import ops
            ops.cast_f32_to_bf16(ps.flat, ps.bf16, ps.n)                   # restore the compute copy
        w_before = ps.flat.clone()
        model._update()
        torch.cuda.synchronize()
        assert float((ps.flat - w_before).abs().max()) > 0 and torch.count_nonzero(ps.grad) == 0
        assert torch.equal(ps.bf16, ps.flat.to(act))
        l0 = float(model.loss_value.item())
        for _ in range(2):
            model.run_step()
        l2 = float(model.loss_value.item())
        model.check_device_protocols()                                     # no non-finite gradient reached the optimizer, no barrier time-out
        assert np.isfinite(l2) and l2 < l0, (l0, l2)
    finally:
        backend.set_compute_dtype('bfloat16')


def test_fp16_loss_scale_can_change_after_compile():
    """backend.set_loss_scale() after compile (what the FloatingPointError of check_device_protocols tells the user to do): the 16-bit
    d(logits) must carry the NEW scale, because the optimizer divides by it per step -- the weights after one step are the same whichever
    scale is used (up to float16 rounding of the scaled gradients), and d(logits) follow the scale exactly"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd import backend
    from test_train_step_gpu import build, make_batch
    backend.set_compute_dtype('float16')
    try:
        H = W = 160
        N, T, Cn = 4, 3, 4
        images, labels = make_batch(N, H, W, T, Cn, seed=9)
        out = []
        for scale in (1024.0, 256.0):
            backend.set_loss_scale(1024.0)
            model, loss, opt, grids = build('resnet-18-v2', H, W, N, Cn, rect=-1, focal=True)
            opt.lr = 1e-4
            backend.set_loss_scale(scale)                                  # AFTER compile
            model.train_on_batch(images, labels)
            torch.cuda.synchronize()
            out.append((model.heads[2].dy.float().clone(), model.g.ps.flat.clone(), float(loss.cfg.grad_scale16)))
        (d0, w0, s0), (d1, w1, s1) = out
        assert (s0, s1) == (1024.0, 256.0)
        torch.testing.assert_close(d0, d1 * 4.0, rtol=2e-3, atol=1e-6)     # powers of two: exact except where fp16 goes subnormal
        ref = build('resnet-18-v2', H, W, N, Cn, rect=-1, focal=True)[0].g.ps.flat
        step0, step1 = w0 - ref, w1 - ref
        assert float(step0.abs().max()) > 0
        assert float((step0 - step1).norm() / step0.norm()) < 2e-2         # a 4x wrong scale would give a 4x different first RAdam step
    finally:
        backend.set_loss_scale(1024.0)
        backend.set_compute_dtype('bfloat16')


def test_validation_pass_updates_nothing_but_the_image_counter():
    """fit(validation_data=...) of the reference (trainer.py:107-110): Model.test_on_batch = forward with batch statistics + loss + L2, no
    weight / moment / moving-average change; the rectified-loss image counter inside the loss graph does advance (yolov3_loss.py:151-152)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_train_step_gpu import build, make_batch
    H = W = 128
    N, T, Cn = 4, 3, 5
    model, loss, opt, grids = build('resnet-18', H, W, N, Cn, rect=1000)
    opt.lr = 1e-5
    images, labels = make_batch(N, H, W, T, Cn, seed=2)
    train_loss = model.train_on_batch(images, labels)
    torch.cuda.synchronize()
    ps = model.g.ps
    snap = (ps.flat.clone(), ps.m.clone(), ps.v.clone(), model.g.bns[3].moving_mean.clone(), model.g.bns[3].moving_var.clone())
    n0, it0 = int(loss.current_num.item()), opt.iterations
    val = model.test_on_batch(images, labels)
    torch.cuda.synchronize()
    for a, b in zip(snap, (ps.flat, ps.m, ps.v, model.g.bns[3].moving_mean, model.g.bns[3].moving_var)):
        assert torch.equal(a, b)
    assert opt.iterations == it0 and int(loss.current_num.item()) == n0 + N
    assert np.isfinite(val) and abs(val - train_loss) < 0.05 * abs(train_loss)      # same batch, weights one 1e-5 step further
    assert model.g.bn_momentum == 0.9 and model.g.training


def test_native_launch_sequencer_replays_the_step_bit_exactly():
    """after two plain steps the third is recorded by the library (every launch + cross-stream edge of the three-stream schedule) and the
    following ones are re-issued with one native call (yolo_seq_run): 7 steps with the sequencer == 7 steps of plain eager launches, bit for
    bit (the kernels are deterministic); a learning-rate change between steps is honoured (it lives in device memory), and a change of a
    launch decision drops the recording (the step is recorded afresh)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from test_train_step_gpu import build, make_batch
    H = W = 160
    N, T, Cn = 4, 3, 4
    batches = [make_batch(N, H, W, T, Cn, seed=30 + i) for i in range(2)]
    out = []
    for native in (True, False):
        model, loss, opt, grids = build('resnet-18', H, W, N, Cn, rect=12)
        model.native_sequencer = native
        curve = []
        for step in range(7):
            opt.lr = 1e-5 if step < 5 else 1e-4
            curve.append(model.train_on_batch(*batches[step % 2]))
        torch.cuda.synchronize()
        out.append((curve, model.g.ps.flat.clone(), model.g.ps.m.clone(), model.g.bns[5].moving_var.clone(), int(loss.current_num.item()), opt.iterations))
        if native:
            assert model._seq is not None and len(model._seq[2]) == 1 and model._seq[2][0][0] > 150       # one segment: the whole step
            sid = model._seq[0]
            model.g.wgrad_batch = 3                      # a launch decision changes: the stale recording is dropped, the step re-recorded
            model.train_on_batch(*batches[0])
            assert model._seq is not None and model._seq[0] != sid
        else:
            assert model._seq is None
            model.train_on_batch(*batches[0])
    (c0, w0, m0, v0, n0, i0), (c1, w1, m1, v1, n1, i1) = out
    assert c0 == c1, (c0, c1)
    assert torch.equal(w0, w1) and torch.equal(m0, m1) and torch.equal(v0, v1) and (n0, i0) == (n1, i1) == (16, 7)      # the image counter stops once it has passed rectified_coord_num = 12
