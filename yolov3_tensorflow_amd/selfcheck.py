"""smoke(): one tiny training step of ResNet18-YOLOv3 on cuda:0 through the HIP path, checked against the CPU oracle
(the oracle is imported here ONLY as the checker, as the task rules allow for __graft_entry__.smoke())."""
import numpy as np
import torch


def run_smoke():
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
    from yolov3_tensorflow_amd.utils.radam import RAdam
    from yolov3_tensorflow_amd.configs import FLAGS
    from oracle.train import OracleTrainer
    H = W = 64
    N, Cn, T = 2, 3, 2
    L = 5 + Cn
    anchors = FLAGS.anchor_boxes
    chans = [len(a) * L for a in anchors]
    grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    names = ['yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32']
    model = YOLOv3Detector('resnet-18').build((H, W, 3), chans, names, batch_size=N, device='cuda:0')
    loss = YOLOv3Loss(grids, Cn, anchors, 0.8, FLAGS.loss_weights, rectified_coord_num=100, rectified_loss_weight=[1.0, 1.0, 1.0])
    model.compile(optimizer=RAdam(lr=1e-3), loss=loss.loss)
    g = torch.Generator().manual_seed(0)
    images = torch.rand(N, H, W, 3, generator=g).numpy()
    labels = -np.ones((N, T, 5), dtype=np.float32)
    labels[0, 0] = [0.4, 0.5, 0.3, 0.4, 1]
    labels[1, 0] = [0.6, 0.3, 0.2, 0.5, 2]
    labels[1, 1] = [0.3, 0.7, 0.5, 0.3, 0]
    labels = labels.reshape(N, T * 5)
    o = OracleTrainer('resnet-18', grids, Cn, anchors, 0.8, FLAGS.loss_weights, rectified_coord_num=100,
                      rectified_loss_weight=[1.0, 1.0, 1.0], emulate_bf16=True)
    o.ensure_params(images)
    o.set_weights(model.get_weights())
    got = model.train_on_batch(images, labels)
    ref = o.step(images, labels)[0]
    if not abs(got - ref) <= 5e-3 * abs(ref):
        raise AssertionError('smoke: loss %.6f vs oracle %.6f' % (got, ref))
    got2 = model.train_on_batch(images, labels)
    if not np.isfinite(got2):
        raise AssertionError('smoke: non-finite loss on the second step')
    print('smoke ok: loss %.5f (oracle %.5f), second step %.5f' % (got, ref, got2))
