#!/usr/bin/env python
"""host enqueue time of one eager training step (python + ctypes + HIP launch), measured with the GPU kernels replaced by nothing:
run K steps and time the launch loop only; the GPU queue absorbs the launches asynchronously."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

model, loss, opt, grids = bench.build_model('resnet-18', 416, 416, 32, 80, torch.device('cuda:0'))
images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
model.stage_batch(images, labels)
for _ in range(3):
    model.run_step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.run_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('host enqueue ms/step: %.3f   step wall ms: %.3f' % (1e3 * sum(a for a, _ in ts) / len(ts), 1e3 * sum(b for _, b in ts) / len(ts)))
