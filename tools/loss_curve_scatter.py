"""How much of the configs[0] loss-curve deviation is summation order?  tools/loss_curve.py's 20 steps (320 x 320, the reference's 20 sample
images, batch 2, 13 classes; lr 1e-5 for 10 steps, then 1e-3) are run on the GPU under several kernel-selection settings that change NOTHING but
the order in which float32 partial sums are added (every setting passes the same kernel parity tests), from the same initial weights, and each
curve is compared with ONE float32 oracle curve.  Prints, per 16-bit type and setting: max / median relative deviation and the steps within
north_star's 1e-3.   usage: python tools/loss_curve_scatter.py [--out file.json]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import loss_curve

SETTINGS = [('default', {}), ('wgrad9 off', {'wgrad9': 0}), ('wgrad9 64 wgs', {'wgrad9_wgs': 64}), ('wgrad9 96 wgs', {'wgrad9_wgs': 96}),
            ('wgrad9 192 wgs', {'wgrad9_wgs': 192}), ('wgrad9 256 wgs', {'wgrad9_wgs': 256}), ('s32 off', {'s32': 0}),
            ('s32 off, wgrad9 off', {'s32': 0, 'wgrad9': 0}), ('wgrad strip off', {'wgrad9': 0, 'wgrad_strip': 0}),
            ('stride-2 classes on the implicit GEMM', {'s32_s2': 0})]
DEFAULTS = {'wgrad9': -1, 'wgrad9_wgs': 128, 's32': -1, 'wgrad_strip': 1, 's32_s2': 1}


def gpu_curve(dtype, weights, steps, plateau_after, tune=None):
    from yolov3_tensorflow_amd import backend
    from yolov3_tensorflow_amd.configs import FLAGS
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
    from yolov3_tensorflow_amd.utils.radam import RAdam
    from yolov3_tensorflow_amd import ops
    backend.set_compute_dtype(dtype)
    try:
        for k, v in DEFAULTS.items():                          # (each 16-bit type has its own library, with its own tuning state)
            ops.set_tuning(k, (tune or {}).get(k, v))
        images, labels = loss_curve.load_fixture()
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
        if weights is None:
            weights = model.get_weights()
        else:
            model.set_weights(weights)
        out = []
        for step in range(steps):
            opt.lr = 1e-5 if step < plateau_after else 1e-3
            i = (step * N) % 20
            out.append(float(model.train_on_batch(images[i:i + N], labels[i:i + N])))
        model.check_device_protocols()
        return out, weights
    finally:
        for k, v in DEFAULTS.items():
            ops.set_tuning(k, v)
        backend.set_compute_dtype('bfloat16')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'loss_curve_scatter.json'))
    a = ap.parse_args()
    from yolov3_tensorflow_amd import ops
    from yolov3_tensorflow_amd.configs import FLAGS
    from oracle.train import OracleTrainer
    steps, plateau = 20, 10
    _, weights = gpu_curve('bfloat16', None, 1, plateau)                 # the initial weights every run starts from
    images, labels = loss_curve.load_fixture()
    N, Cn, H, W = 2, 13, 320, 320
    grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    o = OracleTrainer('resnet-18', grids, Cn, FLAGS.anchor_boxes, FLAGS.iou_thresh, FLAGS.loss_weights, rectified_coord_num=FLAGS.rectified_coord_num,
                      rectified_loss_weight=FLAGS.rectified_loss_weight, lr=1e-5, emulate_bf16=False)
    o.ensure_params(images[:N])
    o.set_weights(weights)
    ref = []
    for step in range(steps):
        o.opt.lr = 1e-5 if step < plateau else 1e-3
        i = (step * N) % 20
        ref.append(float(o.step(images[i:i + N], labels[i:i + N])[0]))
    res = {'float32_oracle_loss': ref, 'runs': []}
    for dtype in ('bfloat16', 'float16'):
        for name, tune in SETTINGS:
            g, _ = gpu_curve(dtype, weights, steps, plateau, tune)
            rel = [abs(x - r) / abs(r) for x, r in zip(g, ref)]
            res['runs'].append({'dtype': dtype, 'setting': name, 'relative_deviation': rel, 'max': max(rel), 'median': float(np.median(rel)),
                                'steps_within_1e-3': int(sum(r <= 1e-3 for r in rel))})
            print('%-9s %-22s max %.2e  median %.2e  within 1e-3: %2d/20   late steps: %s' % (
                dtype, name, max(rel), float(np.median(rel)), sum(r <= 1e-3 for r in rel), ' '.join('%.1e' % r for r in rel[14:])), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
