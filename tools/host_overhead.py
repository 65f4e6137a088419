#!/usr/bin/env python
"""host enqueue time of one eager training step (python + ctypes + HIP launch), measured with the GPU kernels replaced by nothing:
run K steps and time the launch loop only; the GPU queue absorbs the launches asynchronously."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

for size, batch in ((416, 32), (320, 8)):
    for native in (False, True):
        model, loss, opt, grids = bench.build_model('resnet-18', size, size, batch, 80, torch.device('cuda:0'))
        model.native_sequencer = native          # True: the step is recorded once and re-issued by yolo_seq_run (one native call per step)
        images, labels = bench.synthetic_batch(batch, size, size, 80, 0)
        model.stage_batch(images, labels)
        for _ in range(5):
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
        t0 = time.perf_counter()
        for _ in range(30):
            model.run_step()
        torch.cuda.synchronize()
        thr = 30 * batch / (time.perf_counter() - t0)
        print('%dx%d batch %d  %s: host enqueue %.3f ms/step, isolated step wall %.3f ms, back-to-back %.0f images/s'
              % (size, size, batch, 'native sequencer' if native else 'python launches ', 1e3 * sum(a for a, _ in ts) / len(ts),
                 1e3 * sum(b for _, b in ts) / len(ts), thr), flush=True)
        del model, loss, opt
