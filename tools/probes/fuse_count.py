import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bench
from yolov3_tensorflow_amd import engine
for name, size, batch in (('resnet-18', 416, 32), ('mixnet-18', 416, 32), ('resnet-18-v2', 416, 16)):
    model, loss, opt, grids = bench.build_model(name, size, size, batch, 80, torch.device('cuda:0'))
    ops_ = [op for op in model.g.tape if isinstance(op, engine.ApplyOp)]
    fused = [op for op in ops_ if op.producer is not None]
    print(name, 'apply ops', len(ops_), 'fused into dgrad', len(fused), 'elements fused %.1f M of %.1f M' % (
        sum(op.M * op.C for op in fused) / 1e6, sum(op.M * op.C for op in ops_) / 1e6))
    for op in ops_:
        if op.producer is None:
            w = op.out.grad_writers
            print('   not fused: M %d C %d writers %s' % (op.M, op.C, [type(x).__name__ if not isinstance(x, tuple) else 'cat' for x in w]))
    del model, loss, opt
