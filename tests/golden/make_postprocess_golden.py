"""Generate golden vectors for the post-process path by importing the reference's own NumPy code.

Run in the build container only (needs /root/reference):  python -B tests/golden/make_postprocess_golden.py
The reference module imports cv2 (absent here) but uses it only inside ``visualize`` (never called by this script),
so an empty module object is registered under that name; ``np.float``/``np.int`` (removed from NumPy >= 1.24, used at
yolov3_post_process.py:64,87) are aliased to the builtins.  Inputs are seeded synthetic decoded heads; outputs are
what the reference's filter_boxes -> apply_nms -> resize_boxes return.  Only data (inputs + outputs) is stored.
"""
import os
import sys
import types
import numpy as np

sys.dont_write_bytecode = True
np.float = float
np.int = int
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, '/root/reference')
from yolov3.yolov3_post_process import YOLOv3PostProcessor as P   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_heads(rng, grids, box_nums, class_num, hot):
    """decoded heads: prediction (H,W,B,5+C) in grid units, boxes (H,W,B,4); ``hot`` confident overlapping boxes."""
    heads = []
    for (h, w), b in zip(grids, box_nums):
        pred = np.zeros((h, w, b, 5 + class_num), dtype=np.float32)
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
        pred[..., 0] = xs[..., None] + rng.uniform(0.05, 0.95, (h, w, b))
        pred[..., 1] = ys[..., None] + rng.uniform(0.05, 0.95, (h, w, b))
        pred[..., 2] = rng.uniform(0.5, w / 2, (h, w, b))
        pred[..., 3] = rng.uniform(0.5, h / 2, (h, w, b))
        pred[..., 4] = rng.uniform(0.0, 0.6, (h, w, b))
        if class_num:
            logits = rng.normal(size=(h, w, b, class_num)) * 3
            e = np.exp(logits - logits.max(-1, keepdims=True))
            pred[..., 5:] = e / e.sum(-1, keepdims=True)
        # a few confident boxes clustered around common centres so NMS has work to do across heads
        for _ in range(hot):
            i, j, k = rng.integers(0, h), rng.integers(0, w), rng.integers(0, b)
            pred[i, j, k, 4] = rng.uniform(0.93, 0.999)
            if class_num:
                c = rng.integers(0, min(class_num, 3))
                pred[i, j, k, 5:] = 0.001
                pred[i, j, k, 5 + c] = 0.97
            cx, cy = rng.choice([0.3, 0.6]) * w, rng.choice([0.4, 0.7]) * h
            pred[i, j, k, 0:2] = (cx + rng.normal() * 0.02 * w, cy + rng.normal() * 0.02 * h)
            pred[i, j, k, 2:4] = (0.3 * w * rng.uniform(0.9, 1.1), 0.3 * h * rng.uniform(0.9, 1.1))
        half = pred[..., 2:4] / 2
        boxes = np.concatenate([pred[..., 0:2] - half, pred[..., 0:2] + half], axis=-1).astype(np.float32)
        heads.append((pred, boxes))
    return heads


def run_case(name, seed, grids, box_nums, class_num, hot, score_thresh, nms_thresh, target_size):
    rng = np.random.default_rng(seed)
    heads = synth_heads(rng, grids, box_nums, class_num, hot)
    hs = P.filter_boxes(heads[0][0], heads[0][1], heads[1][0], heads[1][1], heads[2][0], heads[2][1], score_thresh)
    out = {'score_thresh': score_thresh, 'nms_thresh': nms_thresh, 'target_size': np.asarray(target_size, dtype=np.float64)}
    for h in range(3):
        out['pred%d' % h], out['boxes%d' % h] = heads[h]
        out['filtered%d' % h] = np.asarray(hs[h], dtype=np.float64).reshape(-1, 8)
    nms = P.apply_nms([np.array(b) for b in hs], nms_thresh)
    res = P.resize_boxes(nms, target_size=np.asarray(target_size))
    for h in range(3):
        out['nms%d' % h] = np.asarray(nms[h], dtype=np.float64).reshape(-1, 9)
        out['resized%d' % h] = np.asarray(res[h], dtype=np.float64).reshape(-1, 9)
    np.savez_compressed(os.path.join(HERE, 'postprocess_%s.npz' % name), **out)
    print(name, [len(out['filtered%d' % h]) for h in range(3)], '->', [len(out['nms%d' % h]) for h in range(3)])


if __name__ == '__main__':
    run_case('g40_c13', 1, [(40, 40), (20, 20), (10, 10)], [3, 2, 3], 13, 12, 0.8, 0.4, [320, 320, 320, 320])
    run_case('g52_c20', 2, [(52, 52), (26, 26), (13, 13)], [3, 3, 3], 20, 16, 0.8, 0.4, [416, 416, 416, 416])
    run_case('g48x60_c0', 3, [(48, 60), (24, 30), (12, 15)], [3, 2, 3], 0, 10, 0.8, 0.4, [480, 384, 480, 384])
    run_case('empty', 4, [(8, 8), (4, 4), (2, 2)], [3, 2, 3], 13, 0, 0.8, 0.4, [320, 320, 320, 320])
