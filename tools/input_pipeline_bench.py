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
ap.add_argument('--procs', type=int, nargs='*', default=[8, 16], help='decode PROCESS pool sizes to time (FileUtil.get_dataset(decode_procs=...))')
ap.add_argument('--train', action='store_true', help='feed the headline training step from the files (end to end) instead of timing the pipeline alone')
ap.add_argument('--real', action='store_true', help='use the reference\'s 20 sample JPEGs (tests/golden/test_sample) looped, instead of synthetic files')
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
    label, root = os.path.join(d, 'label.txt'), d
    if a.real:
        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'test_sample')
        label, root = os.path.join(d, 'looped.txt'), os.path.join(here, 'images')
        lines = [l for l in open(os.path.join(here, 'label.txt')).read().splitlines() if l.strip()]
        open(label, 'w').write('\n'.join(lines * (max(a.images, len(lines)) // len(lines))) + '\n')
    if a.train:
        # end to end: files -> decode processes -> upload -> GPU letterbox / augment -> the headline training step (bench.py's model), the loss
        # read once at the end as the trainer does per epoch
        import bench
        model, loss, opt, grids = bench.build_model('resnet-18', a.size, a.size, a.batch, 80, torch.device('cuda:0'))
        images, labels = bench.synthetic_batch(a.batch, a.size, a.size, 80, 0)
        model.stage_batch(images, labels)
        for _ in range(8):
            model.run_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            model.run_step()
        torch.cuda.synchronize()
        print('resident batch (no input pipeline): %.0f images/s' % (a.batches * a.batch / (time.perf_counter() - t0)), flush=True)
        T = labels.shape[1]
        for procs in a.procs:
            it = FileUtil.get_dataset(label, root, (a.size, a.size), a.batch, is_augment=True, decode_procs=procs, prefetch=4)
            x, y = next(it)
            lab = np.full((a.batch, T), -1, np.float32)

            def step():
                x, y = next(it)
                lab[:, :y.shape[1]] = y[:, :T]
                return model.train_on_batch(x, lab, sync=False)
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses = [step() for _ in range(a.batches)]
            final = float(torch.stack(losses).mean())
            dt = time.perf_counter() - t0
            print('files -> training step, %2d decode processes: %.0f images/s (mean loss %.3f)' % (procs, a.batches * a.batch / dt, final), flush=True)
            it.close()
        sys.exit(0)
    for procs in a.procs:
        it = FileUtil.get_dataset(label, root, (a.size, a.size), a.batch, is_augment=True, decode_procs=procs, prefetch=4)
        next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            x, y = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('processes %2d: %.0f images/s (%d CPUs visible)' % (procs, a.batches * a.batch / dt, len(os.sched_getaffinity(0))), flush=True)
        it.close()
    for workers in ([a.workers] if a.workers else [1, 4, 8, 16]):
        it = FileUtil.get_dataset(label, root, (a.size, a.size), a.batch, is_augment=True, num_workers=workers)
        next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            x, y = next(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('workers %2d: %.0f images/s (%d CPUs visible)' % (workers, a.batches * a.batch / dt, len(os.sched_getaffinity(0))))
        it.close()
