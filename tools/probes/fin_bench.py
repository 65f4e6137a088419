"""yolo_bn_bwd_finalize timed alone, 1024- vs 256-thread workgroups (bwd_fin_small), over the partial-row counts of the benchmark step"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops
dev = torch.device('cuda:0')
for P, C in ((85, 512), (169, 256), (676, 128), (1352, 64), (5408, 64), (5408, 128)):
    part = torch.randn(P, 3, C, device=dev)
    out = [torch.empty(C, device=dev) for _ in range(4)]
    line = '%5d rows x %3d channels:' % (P, C)
    for small in (0, 1):
        ops.set_tuning('bwd_fin_small', small)
        ts = []
        for _ in range(5):
            ops.bn_bwd_finalize(part.view(-1), P, C, 1, 1000.0, *out); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.bn_bwd_finalize(part.view(-1), P, C, 1, 1000.0, *out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 50)
        line += '  %s %6.1f us' % ('256-thread' if small else '1024-thread', statistics.median(ts))
    print(line)
