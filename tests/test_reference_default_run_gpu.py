"""The one place the reference holds numbers for the training path: its TensorBoard screenshot images/tensorboard_loss.jpg (README.md:30) of the
DEFAULT run -- configs.py:31-61 untouched: ResNet18-YOLOv3 384x480, class_num 0, batch 3, 7 steps / epoch, RAdam under lr_func, rectified
loss for the first 1464 images, augmentation on, 300 epochs over the 20 images of dataset/test_sample.  This test runs that configuration
end to end through run.train (JPEG decode -> GPU letterbox / augmentation -> training step -> callbacks -> checkpoints) on the same 20
data files (tests/golden/test_sample) and compares the curve with the screenshot's values at epoch 218.

It is a SANITY pin, not bit parity: TensorFlow's weight initialiser, shuffle order and augmentation draws cannot be reproduced here, and a
20-image over-fit run is noisy from epoch to epoch, so the comparison is on an 21-epoch window around epoch 218 and on bands:
  * Keras loss (epoch mean incl. L2):  screenshot 16.2 (smoothed 16.69)  -> window mean within [0.5x, 1.6x] = [8.1, 25.9]
  * it has come down from > 80 at the start (the chart's y axis) by a factor > 4
  * head /8 carries most of the remaining loss (screenshot 11.2 of 13.2); every screenshot term that is > 0.5 is matched within a
    factor 3 by the window mean (noobj /8 4.38, obj /8 2.82, wh /8 2.71, xy /8 1.25, noobj /16 0.56, noobj /32 0.54, wh /32 0.82)
  * class losses are exactly 0 (class_num 0) and the rectified term is 0 after image 1464 (epoch 70), as on the screenshot"""
import json
import os
import sys
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_default_run_matches_the_tensorboard_screenshot(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import reference_default_run as rdr
    hist = rdr.run(epochs=230, workdir=str(tmp_path))          # the screenshot's cursor is at epoch 218 of 300
    out = rdr.summarise(hist)
    w = out['window_around_epoch_218']
    print(json.dumps({k: v for k, v in w.items()}, indent=1))
    loss = np.asarray(out['loss'])
    assert np.isfinite(loss).all() and len(loss) == 230
    assert out['lr'][0] == pytest.approx(1e-5) and out['lr'][21] == pytest.approx(1e-3) and out['lr'][61] == pytest.approx(1e-4) \
        and out['lr'][81] == pytest.approx(1e-3) and out['lr'][221] == pytest.approx(1e-4)              # configs.py:16-27
    shot = rdr.SCREENSHOT
    assert 0.5 * shot['loss'] <= w['loss_mean'] <= 1.6 * shot['loss'], w['loss_mean']
    assert loss[:3].mean() > 80 and loss[:3].mean() > 4 * w['loss_mean']
    for head in ('head_8', 'head_16', 'head_32'):
        assert w[head]['class'] == 0.0 and w[head]['rectified'] == 0.0
        for term, ref in shot[head].items():
            if ref > 0.5:
                assert ref / 3 <= w[head][term] <= ref * 3, (head, term, w[head][term], ref)
    sums = {h: sum(w[h][t] for t in ('xy', 'wh', 'noobj', 'obj')) for h in ('head_8', 'head_16', 'head_32')}
    assert sums['head_8'] > sums['head_16'] and sums['head_8'] > sums['head_32'], sums
    terms = np.asarray(out['terms_last_step_of_epoch'])
    assert (terms[:60, 5].sum(axis=1) > 0).all() and (terms[75:, 5] == 0).all()      # rectified term: on for 1464 images = 70 epochs of 21
    # a checkpoint was written at epoch 50 under the reference's naming (trainer.py:90-91, configs.py:93-94)
    ckpts = [f for r, d, fs in os.walk(str(tmp_path)) for f in fs if f.startswith('lp-recognition-resnet-18-radam-aug-')]
    assert any(f.startswith('lp-recognition-resnet-18-radam-aug- 50-') for f in ckpts), ckpts
