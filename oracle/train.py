"""Oracle (test infrastructure): one full training step of the reference restated on PyTorch-CPU.

Follows /root/reference/yolov3/trainer.py:69-84 (RAdam(lr=1e-3), YOLOv3Loss(...).loss, model.compile) and what
keras Model.fit executes per step: forward in training mode, YOLOv3 loss + the L2 regularisers, gradients by
autodiff, RAdam update, BatchNorm moving-average update.  Used as the parity checker and as bench.py's ``cpu_baseline``
("port": this is NOT TensorFlow).  parity unpinned (see oracle/__init__.py).
"""
import numpy as np
import torch
from oracle.nets import DetectorOracle, bf16_round
from oracle.loss import YOLOv3LossOracle
from oracle.optim import RAdamOracle


class OracleTrainer(object):
    def __init__(self, backbone, head_grid_sizes, class_num, anchor_boxes, iou_thresh, loss_weights, rectified_coord_num=0,
                 rectified_loss_weight=None, is_focal_loss=False, focal_alpha=0.25, focal_gamma=2.0, is_tiou_recall=False,
                 lr=1e-3, emulate_bf16=False, seed=800, scalar_dtype=np.float64, emulate_bf16_grads=False):
        L = 5 + class_num
        self.det = DetectorOracle(backbone, [len(a) * L for a in anchor_boxes], seed=seed)
        self.loss = YOLOv3LossOracle(head_grid_sizes, class_num, anchor_boxes, iou_thresh, loss_weights, rectified_coord_num,
                                     rectified_loss_weight, is_focal_loss, focal_alpha, focal_gamma, is_tiou_recall)
        self.opt = RAdamOracle(lr=lr, scalar_dtype=scalar_dtype)
        # emulate_bf16: False, True (bfloat16 storage points) or 'float16' (the product's fp16 build)
        self.round_fn = (lambda x: x.to(torch.float16).to(x.dtype)) if emulate_bf16 == 'float16' else (bf16_round if emulate_bf16 else None)
        self.round_grads = bool(emulate_bf16_grads and emulate_bf16)
        self.box_num = [len(a) for a in anchor_boxes]
        self.L = L

    def ensure_params(self, images):
        if self.det.params is None:
            with torch.no_grad():
                self.det.forward(torch.as_tensor(images[:1]), training=True)

    def set_weights(self, weights):
        """weights: {keras name: ndarray in TF layout}"""
        with torch.no_grad():
            for n, t in self.det.params.p.items():
                t.copy_(torch.as_tensor(np.asarray(weights[n], dtype=np.float32)).reshape(t.shape))

    def forward_loss(self, images, labels, inject=None):
        from oracle import nets
        nets._RoundSTE.ROUND_GRADS = self.round_grads
        heads = self.det.forward(torch.as_tensor(images), training=True, round_fn=self.round_fn,
                                 inject=None if inject is None else iter(inject))
        raw = [h.reshape(h.shape[0], h.shape[1], h.shape[2], b, self.L) for h, b in zip(heads, self.box_num)]
        yolo = self.loss.loss_heads(torch.as_tensor(labels), raw)
        l2 = self.det.l2_regulariser()
        return heads, yolo, l2

    def step(self, images, labels):
        for _, t in self.det.params.trainable():
            t.grad = None
        heads, yolo, l2 = self.forward_loss(images, labels)
        total = yolo + l2
        total.backward()
        names = [n for n, _ in self.det.params.trainable()]
        params = [self.det.params.p[n].detach().numpy() for n in names]
        grads = [self.det.params.p[n].grad.detach().numpy() for n in names]
        self.opt.step(params, grads)        # in place on the tensors' storage
        self.det.g.apply_bn_updates()
        return float(total.item()), float(yolo.item()), float(l2.item())
