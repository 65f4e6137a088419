"""the configs[0] loss curve's per-head terms on the GPU under two kernel selections (YOLO_TUNE_A / YOLO_TUNE_B): where does a step's loss differ?
usage: YOLO_TUNE_A=s32_s2=0 YOLO_TUNE_B=s32_s2=1 python tools/probes/loss_terms_ab.py [steps]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import loss_curve
from yolov3_tensorflow_amd import ops
from yolov3_tensorflow_amd.configs import FLAGS
from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
from yolov3_tensorflow_amd.utils.radam import RAdam
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
images, labels = loss_curve.load_fixture()
H = W = 320; N, Cn = 2, 13
anchors, lw = FLAGS.anchor_boxes, FLAGS.loss_weights
chans = [len(b) * (5 + Cn) for b in anchors]
grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
res = {}
for tag in ('A', 'B'):
    for kv in filter(None, os.environ.get('YOLO_TUNE_' + tag, '').split(',')):
        k, v = kv.split('=')
        ops.set_tuning(k, int(v))
    model = YOLOv3Detector('resnet-18').build((H, W, 3), chans, FLAGS.head_names, batch_size=N)
    loss = YOLOv3Loss(grids, Cn, anchors, FLAGS.iou_thresh, lw, rectified_coord_num=FLAGS.rectified_coord_num, rectified_loss_weight=FLAGS.rectified_loss_weight)
    opt = RAdam(lr=1e-3)
    model.compile(optimizer=opt, loss=loss.loss)
    rows = []
    for step in range(steps):
        opt.lr = 1e-5 if step < 10 else 1e-3
        i = (step * N) % 20
        l = float(model.train_on_batch(images[i:i + N], labels[i:i + N]))
        rows.append((l, model.loss_obj.terms.detach().cpu().numpy().copy()))
    res[tag] = rows
names = ['xy', 'wh', 'noobj', 'obj', 'cls']
for step in range(steps):
    la, ta = res['A'][step]; lb, tb = res['B'][step]
    d = tb - ta
    k = np.unravel_index(np.abs(d).argmax(), d.shape)
    print('step %2d  A %.4f  B %.4f  rel %.2e | largest term difference: %s head %d: %.4f -> %.4f' % (
        step + 1, la, lb, abs(la - lb) / abs(la), (names[k[0]] if k[0] < len(names) else 'term %d' % k[0]), k[1], ta[k], tb[k]), flush=True)
