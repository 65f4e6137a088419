#!/usr/bin/env python
"""Throughput of the input pipeline alone: JPEG decode on host threads + GPU letterbox / normalise / augment (FileUtil.get_dataset).
Usage: python tools/input_pipeline_bench.py [--images 256] [--batch 32] [--size 416] [--workers N]"""
import argparse, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from yolov3_tensorflow_amd.dataset.file_util import FileUtil

ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=256)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=416)
ap.add_argument('--workers', type=int, default=None)
ap.add_argument('--batches', type=int, default=40)
a = ap.parse_args()
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    base = rng.integers(0, 255, (375, 500, 3), dtype=np.uint8)
    base = np.asarray(Image.fromarray(base).resize((125, 94)).resize((500, 375)))          # compressible content, VOC-sized
    lines = []
    for i in range(a.images):
        Image.fromarray(np.roll(base, i * 7, axis=1)).save(os.path.join(d, '%d.jpg' % i), quality=90)
        lines.append('%d.jpg 0.5 0.5 0.2 0.2 1' % i)
    open(os.path.join(d, 'label.txt'), 'w').write('\n'.join(lines) + '\n')
    for workers in ([a.workers] if a.workers else [1, 4, 8, 16]):
        it = FileUtil.get_dataset(os.path.join(d, 'label.txt'), d, (a.size, a.size), a.batch, is_augment=True, num_workers=workers)
        next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            x, y = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('workers %2d: %.0f images/s (%d CPUs visible)' % (workers, a.batches * a.batch / dt, len(os.sched_getaffinity(0))))
        it.close()
