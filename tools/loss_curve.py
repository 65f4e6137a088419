#!/usr/bin/env python
"""Loss curve of BASELINE.json configs[0] (ResNet18-YOLOv3 320x320, the reference's 20-image sample set, batch 2, 13 classes, the
reference's first-epoch learning rate 1e-5): the GPU path against the float32 CPU oracle, same initial weights, same batches, step by
step.  Writes a JSON summary (north_star: loss within 1e-3 of the reference run).
Usage: python tools/loss_curve.py [--steps 20] [--out profiles/r01_loss_curve_config1.json]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_config1_gpu import load_fixture
from yolov3_tensorflow_amd.configs import FLAGS
from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
from yolov3_tensorflow_amd.utils.radam import RAdam
from oracle.train import OracleTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--dtype', default='bfloat16', choices=['bfloat16', 'float16'])
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'loss_curve_config1.json'))
a = ap.parse_args()
from yolov3_tensorflow_amd import backend
backend.set_compute_dtype(a.dtype)
images, labels = load_fixture()
H = W = 320
N, Cn = 2, 13
anchors, lw = FLAGS.anchor_boxes, FLAGS.loss_weights
chans = [len(b) * (5 + Cn) for b in anchors]
grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
model = YOLOv3Detector('resnet-18').build((H, W, 3), chans, FLAGS.head_names, batch_size=N)
loss = YOLOv3Loss(grids, Cn, anchors, FLAGS.iou_thresh, lw, rectified_coord_num=FLAGS.rectified_coord_num,
                  rectified_loss_weight=FLAGS.rectified_loss_weight)
opt = RAdam(lr=1e-3)
model.compile(optimizer=opt, loss=loss.loss)
opt.lr = 1e-5
o = OracleTrainer('resnet-18', grids, Cn, anchors, FLAGS.iou_thresh, lw, rectified_coord_num=FLAGS.rectified_coord_num,
                  rectified_loss_weight=FLAGS.rectified_loss_weight, lr=1e-5)
o.ensure_params(images[:N])
o.set_weights(model.get_weights())
gpu, ref = [], []
for step in range(a.steps):
    i = (step * N) % 20
    x, y = images[i:i + N], labels[i:i + N]
    gpu.append(float(model.train_on_batch(x, y)))
    ref.append(float(o.step(x, y)[0]))
    print('step %2d  gpu %.4f  oracle %.4f  rel %.2e' % (step + 1, gpu[-1], ref[-1], abs(gpu[-1] - ref[-1]) / abs(ref[-1])), flush=True)
rel = [abs(g - r) / abs(r) for g, r in zip(gpu, ref)]
out = {'config': 'ResNet18-YOLOv3 320x320, reference sample set (20 images), batch 2, 13 classes, lr 1e-5, %s GPU path vs float32 CPU oracle' % a.dtype,
       'steps': a.steps, 'gpu_loss': gpu, 'oracle_loss': ref, 'relative_deviation': rel, 'max_relative_deviation': max(rel),
       'median_relative_deviation': float(np.median(rel))}
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, 'w'), indent=1)
print('max rel %.2e  median %.2e' % (max(rel), float(np.median(rel))))
