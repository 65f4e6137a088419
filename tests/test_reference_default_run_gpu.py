"""The one place the reference holds numbers for the training path: its TensorBoard screenshot images/tensorboard_loss.jpg (README.md:30) of the
DEFAULT run -- configs.py:31-61 untouched: ResNet18-YOLOv3 384x480, class_num 0, batch 3, 7 steps / epoch, RAdam under lr_func, rectified
loss for the first 1464 images, augmentation on, 300 epochs over the 20 images of dataset/test_sample.  This test runs that configuration
end to end through run.train (JPEG decode -> GPU letterbox / augmentation -> training step -> callbacks -> checkpoints) on the same 20
data files (tests/golden/test_sample).

The screenshot CANNOT pin the path (tools/reference_default_run.py, tests/test_host_cpu.py::test_screenshot_is_below_the_xy_loss_floor):
its xy terms (1.28 for one batch) and epoch loss (16.2) lie below the minimum the reference's current xy cross-entropy
(yolov3_loss.py:350-356) can reach on these labels (24.4 per image on average), so it was produced by an earlier loss.  What IS checked:
  * the learning-rate schedule, rectified-term window (1464 images = 70 epochs) and checkpoint naming of the reference's callbacks
  * the epoch loss falls from > 80 (the chart's y axis) by more than 4x and, around epoch 218, sits between the xy floor + L2 and
    2.2x that: the run converges to the floor of ITS loss
  * every term the floor argument does not touch is at least as converged as on the screenshot (window means of wh / obj / noobj of
    head /8 below 2.71 / 2.82 / 4.38, of heads /16 and /32 below the screenshot's head-/8 values too), class terms exactly 0"""
import json
import os
import sys
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_default_run_converges_to_its_loss_floor(tmp_path):
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
    floor = out['xy_floor_per_image_mean']
    assert 24.0 < floor < 25.0
    assert loss[:3].mean() > 80 and loss[:3].mean() > 4 * w['loss_mean']
    assert floor <= w['loss_mean'] <= 2.2 * (floor + 9.0), (w['loss_mean'], floor)       # ~9 = the L2 term of a he_normal ResNet18 (5e-4 * sum w^2)
    assert floor * 0.8 <= w['xy_sum'] <= 2.0 * floor, (w['xy_sum'], floor)                # last-step samples: a 3-image batch, hence the slack
    shot = rdr.SCREENSHOT['head_8']
    for head in ('head_8', 'head_16', 'head_32'):
        assert w[head]['class'] == 0.0 and w[head]['rectified'] == 0.0
        for term in ('wh', 'obj', 'noobj'):
            assert w[head][term] <= shot[term], (head, term, w[head][term], shot[term])
    terms = np.asarray(out['terms_last_step_of_epoch'])
    assert (terms[:60, 5].sum(axis=1) > 0).all() and (terms[75:, 5] == 0).all()      # rectified term: on for 1464 images = 70 epochs of 21
    # a checkpoint was written at epoch 50 under the reference's naming (trainer.py:90-91, configs.py:93-94)
    ckpts = [f for r, d, fs in os.walk(str(tmp_path)) for f in fs if f.startswith('lp-recognition-resnet-18-radam-aug-')]
    assert any(f.startswith('lp-recognition-resnet-18-radam-aug- 50-') for f in ckpts), ckpts
