"""oracle.postprocess vs golden vectors produced by the reference's own yolov3_post_process.py (PINNED)."""
import glob
import os
import numpy as np
import pytest
from oracle import postprocess as pp

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'postprocess_*.npz')))


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_postprocess_matches_reference(path):
    g = np.load(path)
    heads = [(g['pred%d' % h], g['boxes%d' % h]) for h in range(3)]
    idx, filt = pp.filter_boxes(heads, float(g['score_thresh']))
    for h in range(3):
        np.testing.assert_array_equal(np.asarray(filt[h], dtype=np.float64).reshape(-1, 8), g['filtered%d' % h])
        # decoded box indices: flat ((row*W)+col)*B+anchor of every score > thresh
        H, W, B, _ = heads[h][0].shape
        p = heads[h][0]
        sc = p[..., 4] * (p[..., 5:].max(-1) if p.shape[-1] > 5 else 1.0)
        np.testing.assert_array_equal(idx[h], np.flatnonzero(sc.reshape(-1) > float(g['score_thresh'])))
    nms = pp.apply_nms(filt, float(g['nms_thresh']))
    res = pp.resize_boxes(nms, g['target_size'])
    for h in range(3):
        np.testing.assert_array_equal(np.asarray(nms[h], dtype=np.float64).reshape(-1, 9), g['nms%d' % h])
        np.testing.assert_array_equal(np.asarray(res[h], dtype=np.float64).reshape(-1, 9), g['resized%d' % h])


def test_nms_quirk_vs_fixed():
    g = np.load([p for p in GOLD if 'g40' in p][0])
    heads = [(g['pred%d' % h], g['boxes%d' % h]) for h in range(3)]
    _, filt = pp.filter_boxes(heads, float(g['score_thresh']))
    fixed = pp.apply_nms(filt, float(g['nms_thresh']), reference_quirk=False)
    ids = np.concatenate([np.asarray(f, dtype=np.float64).reshape(-1, 9)[:, -1] for f in fixed])
    assert len(set(ids.tolist())) == len(ids)   # globally unique ids in the corrected mode
