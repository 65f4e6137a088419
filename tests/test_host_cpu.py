"""CPU tests of the host logic: FLAGS surface, lr schedule, checkpoint naming, the graph builder (kernel launches mocked: graph
construction, Keras variable naming/order, launch plan) and weight layout round trips."""
import collections
import contextlib
import os
import sys
import numpy as np
import pytest
import torch

REFERENCE_FLAG_KEYS = """check_step_epoch check_step_lr train_step_epoch train_step_lr step_epoch step_lr train_set_dir train_label_path
test_set_dir test_label_path input_image_size anchor_boxes class_num box_num box_len head_channel_nums head_grid_sizes head_names iou_thresh
loss_weights train_set_size val_set_size batch_size rectified_coord_num rectified_loss_weight epoch init_lr mode model_backbone optimizer
is_augment is_label_smoothing is_focal_loss focal_alpha focal_gamma is_gradient_harmonized is_tiou_recall type log_path steps_per_epoch
validation_steps ckpt_period stop_patience stop_min_delta lr_func root_path tensorboard_dir checkpoint_path checkpoint_name serving_model_dir
pb_model_dir confidence_thresh nms_thresh save_path image_root_path gpu_mode gpu_num visible_gpu""".split()


def test_flags_surface_and_defaults():
    from yolov3_tensorflow_amd import configs
    F = configs.FLAGS
    for k in REFERENCE_FLAG_KEYS:
        assert k in F, k
    assert list(F.input_image_size) == [384, 480, 3] and F.class_num == 0 and F.batch_size == 3          # configs.py:36,42,56
    assert list(F.box_num) == [3, 2, 3] and F.box_len == 5 and list(F.head_channel_nums) == [15, 10, 15]
    assert [list(g) for g in F.head_grid_sizes] == [[48, 60], [24, 30], [12, 15]]
    assert F.type == 'resnet-18-radam-aug' and F.steps_per_epoch == 7
    assert F.checkpoint_name == 'lp-recognition-resnet-18-radam-aug-{epoch: 3d}-{loss: .5f}.ckpt'
    assert F.checkpoint_name.format(epoch=50, loss=16.2) == 'lp-recognition-resnet-18-radam-aug- 50- 16.20000.ckpt'   # spaces kept


def test_lr_schedule_matches_reference_table():
    from yolov3_tensorflow_amd.configs import lr_func
    from oracle.optim import lr_func as lr_oracle
    # configs.py:16-27: 1e-5 for epochs <= 20, 1e-3 <= 60, 1e-4 <= 80, 1e-3 <= 220, 1e-4 <= 260, 1e-5 <= 280, 1e-6 <= 300
    table = {0: 1e-5, 20: 1e-5, 21: 1e-3, 60: 1e-3, 61: 1e-4, 80: 1e-4, 81: 1e-3, 220: 1e-3, 221: 1e-4, 260: 1e-4, 261: 1e-5, 280: 1e-5,
             281: 1e-6, 300: 1e-6}
    for e, v in table.items():
        assert lr_func(e) == pytest.approx(v) and lr_oracle(e) == pytest.approx(v)


def test_detector_and_loss_argument_errors():
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
    with pytest.raises(ValueError):
        YOLOv3Detector('no-such-net')                                     # yolov3_detector.py:39-42
    with pytest.raises(Exception):
        YOLOv3Detector('resnet-18').build((384, 480), [15, 10, 15], ['a', 'b', 'c'])     # :52-53
    with pytest.raises(ValueError):
        YOLOv3Loss([(4, 4), (2, 2), (1, 1)], 0, [[(1, 1)]] * 3, 0.8, [(1, 1, 1, 1, 1)] * 3, rectified_loss_weight=[1.0])   # yolov3_loss.py:65-66
    assert YOLOv3Detector.BACKBONE_RESNET_18 == 'resnet-18' and YOLOv3Detector.BACKBONE_MIXNET_18 == 'mixnet-18'


def test_radam_facade():
    from yolov3_tensorflow_amd.utils.radam import RAdam
    from yolov3_tensorflow_amd import backend
    o = RAdam(lr=1e-3)
    assert o.epsilon == backend.epsilon() == 1e-8 and o.rho_inf == pytest.approx(1999.0)
    assert o.get_config() == {'lr': 1e-3, 'beta_1': 0.9, 'beta_2': 0.999, 'decay': 0.0, 'epsilon': 1e-8, 'amsgrad': False}
    o.lr = 1e-5
    assert o.lr == 1e-5
    with pytest.raises(TypeError):
        RAdam(bogus=1)


@pytest.fixture
def mocked_kernels(monkeypatch):
    """graph building / launch planning on the CPU: every kernel wrapper records its name instead of launching"""
    from yolov3_tensorflow_amd import ops
    calls = []
    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and getattr(fn, '__module__', None) == ops.__name__ and name not in (
                'same_pad', 'conv_problem', 'mix_problem', 'pad_channels', 'make_loss_config', 'conv2d_stat_rows', 'reduce_rows', 'radam_l2_blocks',
                'loss_workspace_bytes', 'check', '_p', '_stream', 'conv2d_wgrad_workspace_bytes', 'conv2d_wgrad_splits', 'conv2d_dgrad_bn_rows', 'conv2d_dgrad_classed', 'stem_pool_bwd_slabs', 'reduce_blocks', 'bn_bwd_fused_workspace_floats', 'bn_bwd_fused_sync_words', 'dwconv_mix_wgrad_workspace_bytes', 'conv2d_fwd_plan', 'acc_words', 'tuning_epoch', 'set_tuning'):
            monkeypatch.setattr(ops, name, (lambda n: (lambda *a, **k: calls.append(n)))(name))
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    monkeypatch.setattr(torch.cuda, 'device', lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    return calls


@pytest.mark.parametrize('acc', [False, True], ids=['rows', 'accumulators'])
@pytest.mark.parametrize('backbone,n_conv,n_bn', [('resnet-18', 31, 28), ('resnet-18-v2', 31, 30), ('mixnet-18', 23, 52)])
def test_graph_builder_names_and_plan(mocked_kernels, monkeypatch, backbone, n_conv, n_bn, acc):
    """layer counts of the reference's plot_model dumps (images/resnet-18.svg etc.) and Keras auto-names in the
    reference's construction order == the oracle's independent restatement"""
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
    from yolov3_tensorflow_amd.utils.radam import RAdam
    from oracle.nets import DetectorOracle
    monkeypatch.setenv('YOLO_STAT_ACC', '1' if acc else '0')       # BatchNorm statistics through exact accumulators (off by default: measured slower)
    L = 5 + 13
    chans = [3 * L, 2 * L, 3 * L]
    m = YOLOv3Detector(backbone).build((96, 96, 3), chans, ['yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32'], batch_size=2, device='cpu')
    names = list(m.g.ps.params.keys())
    n_dw = 32 if backbone == 'mixnet-18' else 0
    assert sum(1 for n in names if n.endswith('/kernel')) == n_conv and sum(1 for n in names if n.endswith('/gamma')) == n_bn
    assert sum(1 for n in names if n.endswith('/depthwise_kernel')) == n_dw
    det = DetectorOracle(backbone, chans)
    det.forward(torch.rand(1, 96, 96, 3))
    assert names == [n for n, _ in det.params.trainable()]
    for n, t in det.params.trainable():
        assert tuple(t.shape) == m.g.ps.params[n].tf_shape, n
    assert m.count_params() == sum(t.numel() for _, t in det.params.trainable())
    # the launch plan: every conv has fwd + wgrad, all but the stem have dgrad; one loss; one optimizer launch
    loss = YOLOv3Loss([(12, 12), (6, 6), (3, 3)], 13, [[(1, 1)] * 3, [(1, 1)] * 2, [(1, 1)] * 3], 0.5, [(5, 5, .05, 3, 1)] * 3)
    m.compile(RAdam(), loss.loss)
    m.use_hip_graph = False
    m.overlap_wgrad = False           # no HIP streams on the CPU
    mocked_kernels.clear()
    m._fwd_bwd()
    m._update()
    c = collections.Counter(mocked_kernels)
    # (the stem's weight gradient comes out of the fused stem backward kernel together with the un-pooling and the BatchNorm apply)
    assert c['conv2d_fwd'] == n_conv and c['conv2d_wgrad_slabs'] == n_conv - 1 and c['conv2d_dgrad'] == n_conv - 1
    assert c['stem_pool_bwd_wgrad'] == 1 and c['bn_pool_bwd_apply'] == 0
    assert 1 <= c['wgrad_reduce_batched'] <= 3 and c['conv2d_wgrad_reduce'] == 0        # one slab-summing launch per gradient bucket
    assert c['loss_fwd_bwd'] == 1 and c['radam_l2_step'] == 1 and c['radam_schedule'] == 1 and c['upcat_split_bwd'] == 2
    from yolov3_tensorflow_amd import engine
    # (small maps -- every map of this 96 x 96 input -- run finalize + apply as ONE launch where the unit has a single plain BatchNorm)
    # (every unit with a single plain BatchNorm over a convolution runs finalize + apply as ONE launch fed by an exact accumulator block:
    #  no statistics rows, no finalize launch, at any map size; the blocks of a step are zeroed by one launch)
    merged_f = c['bn_finalize_act_fwd'] + c['bn_finalize_act_fwd_acc']
    merged_b = c['bn_bwd_finalize_apply'] + c['bn_bwd_finalize_apply_acc']
    assert c['bn_finalize'] + merged_f + 4 * c['bn_finalize_grouped'] == n_bn      # MixNet: one grouped launch per 4 group BatchNorms
    assert merged_f + c['bn_act_fwd'] == len([op for op in m.g.tape if isinstance(op, engine.ApplyOp)])
    if acc:
        assert c['bn_finalize_act_fwd_acc'] > 0 and c['bn_bwd_finalize_apply_acc'] > 0 and c['zero_words'] == 1
        assert c['bn_finalize_act_fwd'] == 0 and c['bn_bwd_finalize_apply'] == 0    # (what the row-fed merged launch served, the accumulators serve too)
    else:
        assert c['bn_finalize_act_fwd'] > 0 and c['bn_bwd_finalize_apply'] > 0 and c['zero_words'] == 0
        assert c['bn_finalize_act_fwd_acc'] == 0 and c['bn_bwd_finalize_apply_acc'] == 0
    assert merged_b + c['bn_act_bwd_apply'] <= merged_f + c['bn_act_fwd']
    if n_dw:
        assert c['dwconv_mix_fwd'] == 8 and c['dwconv_mix_dgrad'] == 8 and c['dwconv_mix_wgrad'] == 8
    # backward plan of the BatchNorm units: the data gradient that writes a unit's output gradient LAST carries its reduce (every unit whose
    # last writer is a plain convolution: all but the two fed by the upsample + concat split, MixNet's depthwise-fed ones, v2's plain sums);
    # such a unit then launches finalize + apply only; identity shortcuts read the masked gradient in place
    from yolov3_tensorflow_amd import engine
    units = [op for op in m.g.tape if isinstance(op, engine.ApplyOp)]
    fused = [op for op in units if op.producer is not None]
    assert len(units) == {'resnet-18': 23, 'resnet-18-v2': 34, 'mixnet-18': 23}[backbone]
    assert len(fused) == {'resnet-18': 21, 'resnet-18-v2': 22, 'mixnet-18': 13}[backbone]
    for op in units:
        w = op.out.grad_writers
        last_is_conv = bool(w) and isinstance(w[-1], engine.ConvOp) and w[-1].y.x is op.out
        assert (op.producer is not None) == (last_is_conv and (op.m_bn is not None or op.o_bn is not None))
        if op.producer is not None:
            from yolov3_tensorflow_amd import ops as ops_real
            epi = op.producer.bn_epi
            if getattr(op, 'acc_b', None) is not None:        # tile sums into the unit's accumulator block (3 quantities x Cin) instead of rows
                assert epi['partial'] is None and epi['acc'] is op.acc_b and op.acc_b.numel() == ops_real.acc_words(3, op.C)
            else:
                assert epi['partial'] is op.fpartial and op.fpartial.shape[1:] == (3, op.C) and float(op.fpartial.abs().sum()) == 0.0
    # (the mocked single-launch kernel 'declines', so every unit without a producer also takes the three-kernel path here)
    assert c['bn_act_bwd_reduce'] == len([op for op in units if op.producer is None and (op.m_bn is not None or op.o_bn is not None)])
    n_alias = sum(1 for op in units if op.skip_dres)
    assert backbone != 'resnet-18' or n_alias == 4, n_alias          # the second block of each stage has the identity shortcut
    # down-sampling blocks: the 1x1 / stride-2 shortcut's data gradient writes the even / even positions only, the 3x3 / stride-2 one
    # accumulates onto that parity class alone
    sparse = [op for op in m.g.tape if isinstance(op, engine.ConvOp) and op.even_only]
    assert len(sparse) == (3 if backbone != 'mixnet-18' else len(sparse))       # the shortcuts of the three down-sampling stages
    for op in sparse:
        w = op.y.x.grad_writers
        i = w.index(op)
        if op.acc == [False]:                   # first writer: its 3x3 partner covers the rest of the tensor
            assert w[i + 1].acc == [2] and w[i + 1].y.p.R == 3 and w[i + 1].y.p.stride == 2
        else:                                   # (the stage outputs that also feed the FPN heads: the head's gradient arrives first)
            assert i > 0 and op.acc == [True]
    for op in m.g.tape:
        if isinstance(op, engine.ConvOp) and op.addend is not None:
            assert op.acc == [True] and any(u.skip_dres and u.out is op.addend for u in units)


def test_weight_layout_roundtrip_and_checkpoint(mocked_kernels, tmp_path):
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd import model as model_lib
    m = YOLOv3Detector('resnet-18').build((64, 64, 3), [15, 10, 15], ['yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32'], batch_size=1,
                                          device='cpu')
    w = m.get_weights()
    assert w['conv2d/kernel'].shape == (3, 3, 3, 64) and w['yolov3_head_16/kernel'].shape == (1, 1, 512, 10)      # TF HWIO, unpadded
    assert w['yolov3_head_32/bias'].shape == (15,) and w['batch_normalization_v1/moving_variance'].shape == (64,)
    rng = np.random.default_rng(0)
    w2 = {k: rng.normal(size=v.shape).astype(np.float32) for k, v in w.items()}
    m.set_weights(w2)
    w3 = m.get_weights()
    for k in w2:
        np.testing.assert_array_equal(w2[k], w3[k])
    # padded lanes of the device layout stay zero (stem channels 3..7, detection channels beyond B*L)
    p = m.g.ps.params['conv2d/kernel']
    dev = m.g.ps.flat[p.offset:p.offset + p.numel].reshape(p.dev_shape)
    assert p.dev_shape == (64, 3, 3, 8) and torch.count_nonzero(dev[..., 3:]) == 0
    # checkpoint naming / latest pointer (trainer.py:57-64,90-91)
    stem = str(tmp_path / 'models' / 'lp-recognition-x-{epoch: 3d}-{loss: .5f}.ckpt'.format(epoch=50, loss=1.5))
    m.save_weights(stem)
    assert model_lib.latest_checkpoint(str(tmp_path / 'models')) == stem
    m.set_weights(w)
    m.load_weights(stem)
    np.testing.assert_array_equal(m.get_weights()['conv2d_5/kernel'], w2['conv2d_5/kernel'])
    with pytest.raises(KeyError):
        m.set_weights({'conv2d/kernel': w['conv2d/kernel']})


def test_screenshot_is_below_the_xy_loss_floor():
    """the reference's only numbers for the training path (images/tensorboard_loss.jpg: epoch loss 16.2, xy terms 1.25 + 0.0064 + 0.0227 for one
    batch) are unreachable under its current loss code and defaults: the xy cross-entropy (yolov3_loss.py:350-356) is bounded below by the
    entropy of the fractional target, which on the 20 sample labels amounts to 24.4 per image (6.2 for the best 3-image batch)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import reference_default_run as rdr
    floor = rdr.xy_loss_floor()
    assert floor.shape == (20,) and (floor > 0).all()
    shot = rdr.SCREENSHOT
    xy_shot = sum(shot[h]['xy'] for h in ('head_8', 'head_16', 'head_32'))
    assert np.sort(floor)[:3].mean() > 4 * xy_shot             # no 3-image batch can show the screenshot's xy terms
    assert floor.mean() > shot['loss']                         # the xy terms alone exceed the screenshot's whole epoch loss
    assert floor.mean() == pytest.approx(24.38, abs=0.05)


@pytest.mark.parametrize('name,cls,expect', [('sgdm', 'SGD', dict(lr=0.0002, momentum=0.95, nesterov=True)),
                                             ('adam', 'Adam', dict(lr=0.0002, amsgrad=True, beta_1=0.9, beta_2=0.999)),
                                             ('radam', 'RAdam', dict(lr=0.001, amsgrad=False))])
def test_trainer_selects_the_reference_optimizers(mocked_kernels, tmp_path, monkeypatch, name, cls, expect):
    """reference trainer.py:69-75: SGD(lr=FLAGS.init_lr, momentum=0.95, nesterov=True) unless FLAGS.optimizer is 'adam'
    (Adam(lr=FLAGS.init_lr, amsgrad=True)) or 'radam' (RAdam(lr=1e-3), which ignores init_lr); an update launches the schedule of that
    optimizer's kind once, as one launch or as one per gradient bucket"""
    from yolov3_tensorflow_amd import configs
    F = configs.FLAGS
    saved = dict(F)
    try:
        F.update(configs.DEFAULTS)
        F.optimizer, F.root_path, F.input_image_size, F.batch_size = name, str(tmp_path) + os.sep, np.array([64, 96, 3]), 2
        configs.refresh_derived()
        monkeypatch.delenv('WORLD_SIZE', raising=False)
        from yolov3_tensorflow_amd.yolov3.trainer import YOLOv3Trainer
        from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
        build = YOLOv3Detector.build
        monkeypatch.setattr(YOLOv3Detector, 'build', lambda self, *a, **k: build(self, *a, **dict(k, device='cpu')))     # graph on the CPU, kernels mocked
        tr = YOLOv3Trainer()
        opt = tr.optimizer
        assert type(opt).__name__ == cls and tr.batch_size == 2
        for k, v in expect.items():
            assert getattr(opt, k) == pytest.approx(v) if isinstance(v, float) else getattr(opt, k) == v, k
        assert F.type == 'resnet-18-%s-aug' % name and tr.checkpoint_path.endswith('lp-recognition-resnet-18-%s-aug-{epoch: 3d}-{loss: .5f}.ckpt' % name)
        tr.model.overlap_wgrad = False            # no HIP streams on the CPU
        mocked_kernels.clear()
        tr.model._fwd_bwd()
        tr.model._update()
        c = collections.Counter(mocked_kernels)
        assert c['radam_l2_step'] == 1 and c['sum_partials'] == 1                     # the update + the loss / L2 reduction
        assert (c['radam_schedule'], c['optimizer_schedule']) == ((1, 0) if name == 'radam' else (0, 1))
        mocked_kernels.clear()
        for i, (lo, hi) in enumerate(tr.model.g.bucket_ranges):                       # the per-bucket form the training step uses
            opt.launch_range(tr.model, lo, hi, i == 0)
        opt.finish(tr.model)
        c = collections.Counter(mocked_kernels)
        assert c['radam_l2_step'] == 3 and c['radam_schedule'] + c['optimizer_schedule'] == 1 and c['sum_partials'] == 1
        assert sum(hi - lo for lo, hi in tr.model.g.bucket_ranges) == tr.model.g.ps.n
    finally:
        F.clear()
        F.update(saved)
