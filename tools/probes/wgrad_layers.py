"""every convolution's weight-gradient launch of the benchmark model timed alone (slab pass only): shape, splits, us, TFLOP/s, operand GB/s"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from yolov3_tensorflow_amd import engine, ops
for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):      # e.g. YOLO_TUNE=wgrad_min_steps=4 (planned when the model is built)
    k, v = kv.split('=')
    ops.set_tuning(k, int(v))
dev = torch.device('cuda:0')
model, loss, opt, grids = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else 'resnet-18', 416, 416, 32, 80, dev)
images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
model.stage_batch(images, labels)
model.overlap_wgrad = False
for _ in range(2):
    model._fwd_bwd(); model._update()
torch.cuda.synchronize()
tot = 0.0
for op in model.g.tape:
    if not isinstance(op, engine.ConvOp) or op.fused_slabs:
        continue
    p = op.y.p
    for _ in range(2): op._wgrad()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): op._wgrad()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100.0
    fl = 2.0 * p.N * p.Ho * p.Wo * p.Cout * p.Cin * p.R * p.S
    by = 2.0 * (p.N * p.H * p.W * p.Cin + p.N * p.Ho * p.Wo * p.Cout) + op.splits * op.y.wp.numel * 4.0 * (op.splits > 1)
    strip = p.R == 3 and p.stride == 1 and p.C0 == 0 and p.Cin % 64 == 0
    tot += us
    print('%-28s %4dx%-4d Cin %4d (C0 %3d) Cout %4d k%d s%d %s splits %3d  %6.1f us  %6.1f TFLOP/s  %6.0f GB/s' % (
        op.y.wp.name[:28], p.H, p.W, p.Cin, p.C0, p.Cout, p.R, p.stride, 'strip' if strip else 'igemm', op.splits, us, fl / us / 1e6, by / us / 1e3))
print('total %.0f us' % tot)
