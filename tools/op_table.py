#!/usr/bin/env python
"""Per-op device time of one training step (each fused op's forward and backward timed alone, back to back, with HIP events).
Usage: python tools/op_table.py [--backbone resnet-18] [--batch 32] [--size 416] [--iters 20]"""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from yolov3_tensorflow_amd import engine


def describe(op):
    if isinstance(op, engine.ConvOp):
        p = op.y.p
        return 'conv %dx%d s%d %4d->%4d @%dx%d' % (p.R, p.S, p.stride, p.Cin, p.Cout, p.Ho, p.Wo), \
            2.0 * p.N * p.Ho * p.Wo * p.Cout * p.Cin * p.R * p.S
    if isinstance(op, engine.MixConvOp):
        return 'mixconv %s' % (tuple(op.y.shape),), 0.0
    if isinstance(op, engine.PoolOp):
        return 'pool %s' % (tuple(op.out.shape),), 0.0
    return 'apply %s' % (tuple(op.out.shape),), 0.0


def time_it(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='resnet-18')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model, loss, opt, grids = bench.build_model(a.backbone, a.size, a.size, a.batch, 80, dev)
    model.overlap_wgrad = False
    from yolov3_tensorflow_amd import ops
    for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):
        k, v = kv.split('=')
        ops.set_tuning(k, int(v))
    images, labels = bench.synthetic_batch(a.batch, a.size, a.size, 80, 0)
    model.stage_batch(images, labels)
    model.run_step()
    torch.cuda.synchronize()
    g = model.g
    g.wgrad_stream = None
    tot = {'fwd': 0.0, 'bwd': 0.0, 'wgrad': 0.0}
    print('%-44s %9s %9s %9s   TF/s fwd / bwd(dgrad) / wgrad' % ('op', 'fwd us', 'bwd us', 'wgrad us'))
    for op in g.tape:
        name, fl = describe(op)
        tf = time_it(op.forward, a.iters)
        if isinstance(op, engine.ConvOp):
            tw = time_it(op._wgrad, a.iters)
            tb = time_it(op.backward, a.iters) - tw
        else:
            tw = 0.0
            tb = time_it(op.backward, a.iters)
        tot['fwd'] += tf
        tot['bwd'] += tb
        tot['wgrad'] += tw
        extra = ''
        if fl:
            extra = '%6.0f %6.0f %6.0f' % (fl / tf / 1e6, fl / tb / 1e6 if op.needs_dgrad() else 0, fl / tw / 1e6)
        print('%-44s %9.1f %9.1f %9.1f   %s' % (name, tf, tb, tw, extra))
    print('total us: fwd %.0f  bwd %.0f  wgrad %.0f  sum %.0f' % (tot['fwd'], tot['bwd'], tot['wgrad'], sum(tot.values())))


if __name__ == '__main__':
    main()
