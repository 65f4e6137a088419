"""every convolution's forward and data-gradient launch of the benchmark model timed alone: shape, us, TFLOP/s, operand GB/s"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from yolov3_tensorflow_amd import engine, ops
dev = torch.device('cuda:0')
for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):      # e.g. YOLO_TUNE=s32=0
    k, v = kv.split('=')
    ops.set_tuning(k, int(v))
model, loss, opt, grids = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else 'resnet-18', 416, 416, 32, 80, dev)
images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
model.stage_batch(images, labels)
model.overlap_wgrad = False
for _ in range(2):
    model._fwd_bwd(); model._update()
torch.cuda.synchronize()


def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 100.0


tot = [0.0, 0.0]
for op in model.g.tape:
    if not isinstance(op, engine.ConvOp):
        continue
    y = op.y
    p = y.p
    fl = 2.0 * p.N * p.Ho * p.Wo * p.Cout * p.Cin * p.R * p.S
    by = 2.0 * (p.N * p.H * p.W * p.Cin + p.N * p.Ho * p.Wo * p.Cout)
    us = timed(op.forward)
    tot[0] += us
    line = '%-24s %3dx%-3d Cin %4d (C0 %3d) Cout %4d k%d s%d  fwd %6.1f us %6.1f TF %5.0f GB/s' % (
        y.wp.name[:24], p.H, p.W, p.Cin, p.C0, p.Cout, p.R, p.stride, us, fl / us / 1e6, by / us / 1e3)
    x = y.x
    if op.needs_dgrad() and x.kind != 'cat':
        def dg():
            ops.conv2d_dgrad(y.p, y.dy, op.w_dg, x.grad, accumulate=op.acc[0], bn=op.bn_epi,
                             addend=None if op.addend is None else op.addend.grad, even_only=op.even_only)
        us = timed(dg)
        tot[1] += us
        line += '  | dgrad%s%s%s %6.1f us %6.1f TF' % (' bn' if op.bn_epi else '', ' acc%s' % op.acc[0] if op.acc[0] else '', ' even' if op.even_only else '',
                                                   us, fl / (4 if op.even_only else 1) / us / 1e6)
    print(line)
print('total fwd %.0f us, dgrad %.0f us' % tuple(tot))
