"""GPU input pipeline: yolo_letterbox_augment against oracle/dataset.py on identical random draws.
Letterbox + x/255 + BGR and the salt-and-pepper decisions are bit-exact; gaussian noise (logf / cosf) and the contrast mean (summation
order) agree to 3e-6 absolute on values in [0, 1].  TensorFlow itself is not available: see the oracle's header (parity unpinned
against TF, Philox pinned against the Random123 known-answer vectors in the CPU suite)."""
import os
import time
import numpy as np
import pytest
import torch

t_start = time.time()

pytestmark = pytest.mark.gpu


def _pipe(N, size):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd.dataset.file_util import DeviceImagePipeline
    return DeviceImagePipeline(N, size)


def _images(rng, sizes):
    return [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in sizes]


SIZES = [(375, 500), (500, 353), (333, 500), (64, 64), (1, 7), (97, 31), (480, 640), (600, 13)]


@pytest.mark.parametrize('size', [(320, 320), (416, 416), (96, 160)], ids=str)
def test_letterbox_is_bit_exact(size):
    from oracle import dataset as ods
    rng = np.random.RandomState(1)
    imgs = _images(rng, SIZES)
    pipe = _pipe(len(imgs), size)
    packed = torch.full((len(imgs), size[0], size[1], 8), 7.0, dtype=torch.bfloat16, device=pipe.device)
    out = pipe(imgs, None, out_bf16x8=packed)
    want = np.stack([ods.to_float_bgr(ods.letterbox(im, size)) for im in imgs])
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    pk = packed.float().cpu().numpy()
    np.testing.assert_array_equal(pk[..., :3], torch.as_tensor(want).to(torch.bfloat16).float().numpy())
    assert (pk[..., 3:] == 0).all()
    out2 = pipe(imgs[::-1], None)                                           # the staging buffers are reused correctly
    np.testing.assert_array_equal(out2.cpu().numpy(), want[::-1])


@pytest.mark.parametrize('noise', [0, 1, 2], ids=['saltpepper', 'gauss', 'nonoise'])
@pytest.mark.parametrize('order', [0, 1, 2, 3], ids=['bsc', 'sbc', 'scb', 'nocolor'])
def test_augmentation_matches_oracle(noise, order):
    from oracle import dataset as ods
    rng = np.random.RandomState(10 + noise * 4 + order)
    sizes = SIZES[:4]
    imgs = _images(rng, sizes)
    imgs[3][:] = imgs[3][:, :, :1]                                           # a grey image: zero saturation range
    size = (160, 192)
    draws = [dict(noise=noise, color_order=order, brightness_delta=float(rng.uniform(-30 / 255., 30 / 255.)),
                  saturation_factor=float(rng.uniform(0.9, 1.1)), contrast_factor=float(rng.uniform(0.9, 1.1)),
                  seed0=int(rng.randint(0, 2 ** 31 - 1)), seed1=int(rng.randint(0, 2 ** 31 - 1))) for _ in imgs]
    pipe = _pipe(len(imgs), size)
    got = pipe(imgs, draws).cpu().numpy()
    for n, (im, d) in enumerate(zip(imgs, draws)):
        base = ods.to_float_bgr(ods.letterbox(im, size))
        want = ods.augment(base, d['noise'], d['color_order'], d['brightness_delta'], d['saturation_factor'], d['contrast_factor'],
                           (d['seed0'], d['seed1']), n)
        if noise != 1 and order == 3:
            np.testing.assert_array_equal(got[n], want)                      # no transcendental, no mean: exact
        else:
            np.testing.assert_allclose(got[n], want, rtol=0, atol=3e-6)
        assert got[n].min() >= 0.0 and got[n].max() <= 1.0
    again = pipe(imgs, draws).cpu().numpy()
    np.testing.assert_array_equal(again, got)                                # deterministic for fixed draws


def test_mixed_batch_and_noise_statistics():
    """every image of a batch follows its own draw; the noise has the reference's rates (p = 0.01 salt-and-pepper, sigma = 0.01)"""
    rng = np.random.RandomState(3)
    img = np.full((208, 208, 3), 128, np.uint8)
    pipe = _pipe(3, (416, 416))
    mk = lambda noise, s: dict(noise=noise, color_order=3, brightness_delta=0.0, saturation_factor=1.0, contrast_factor=1.0, seed0=s, seed1=9)
    out = pipe([img, img, img], [mk(0, 5), mk(1, 6), mk(2, 7)]).cpu().numpy()
    base = np.float32(128) * np.float32(1.0 / 255)
    sp = (out[0] != base).any(-1)
    assert abs(sp.mean() - 0.01) < 0.002
    hit = out[0][sp]
    assert ((hit == 0) | (hit == 1)).all() and (hit[:, 0] == hit[:, 1]).all() and abs((hit[:, 0] == 1).mean() - 0.5) < 0.08
    g = out[1] - base
    assert abs(g.std() - 0.01) < 3e-4 and abs(g.mean()) < 1e-4
    assert abs(np.corrcoef(g[..., 0].ravel(), g[..., 1].ravel())[0, 1]) < 0.01      # channels get independent noise
    assert (out[2] == base).all()


def test_get_dataset_yields_device_batches(tmp_path):
    from PIL import Image
    from oracle import dataset as ods
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(0)
    lines = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 255, (40 + 4 * i, 60, 3), dtype=np.uint8)).save(tmp_path / ('%d.png' % i))
        lines.append('%d.png ' % i + ' '.join('0.5 0.5 0.2 0.2 %d' % j for j in range(1 + i % 3)))
    (tmp_path / 'label.txt').write_text('\n'.join(lines) + '\n')
    it = FileUtil.get_dataset(str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 2, is_augment=True, is_test=False)
    for _ in range(4):
        x, y = next(it)
        assert x.is_cuda and tuple(x.shape) == (2, 64, 64, 3) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        assert y.shape == (2, 15) and ((y == -1) | (y >= 0)).all()
    test_batches = list(FileUtil.get_dataset(str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 2, is_augment=False, is_test=True))
    assert len(test_batches) == 3 and len(test_batches[0]) == 3 and len(test_batches[0][2]) == 2
    x, y, paths = test_batches[0]
    for n in range(2):                                                        # test mode: exactly the letterboxed file, paired with its label
        raw = FileUtil.read_image(paths[n])
        np.testing.assert_array_equal(x[n].cpu().numpy(), ods.to_float_bgr(ods.letterbox(raw, (64, 64))))
        k = 1 + int(paths[n].split('/')[-1][0]) % 3
        assert (y[n].reshape(-1, 5)[:k, 4] == np.arange(k)).all() and (y[n].reshape(-1, 5)[k:] == -1).all()


def test_process_pool_decode_yields_the_same_batches(tmp_path):
    """FileUtil.get_dataset(decode_procs=2): the JPEG / PNG decode runs in two spawned worker processes that write into the shared,
    page-locked staging slots; order, labels, augmentation draws and pixels are those of the thread path -- bit for bit"""
    from PIL import Image
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(1)
    lines = []
    for i in range(7):
        Image.fromarray(rng.integers(0, 255, (40 + 5 * i, 64 - 3 * i, 3), dtype=np.uint8)).save(tmp_path / ('%d.png' % i))
        lines.append('%d.png ' % i + ' '.join('0.5 0.5 0.2 0.2 %d' % j for j in range(1 + i % 3)))
    (tmp_path / 'label.txt').write_text('\n'.join(lines) + '\n')
    args = (str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 3)
    a = FileUtil.get_dataset(*args, is_augment=True, is_test=False, num_workers=2)
    b = FileUtil.get_dataset(*args, is_augment=True, is_test=False, decode_procs=2)
    try:
        for _ in range(6):                                     # crosses an epoch boundary (7 files, batches of 3)
            (xa, ya), (xb, yb) = next(a), next(b)
            assert torch.equal(xa, xb) and np.array_equal(ya, yb)
    finally:
        a.close()
        b.close()
    ta = list(FileUtil.get_dataset(*args, is_augment=False, is_test=True))
    tb = list(FileUtil.get_dataset(*args, is_augment=False, is_test=True, decode_procs=2))
    assert len(ta) == len(tb) == 3
    for (xa, ya, pa), (xb, yb, pb) in zip(ta, tb):
        assert torch.equal(xa, xb) and np.array_equal(ya, yb) and pa == pb
    import glob
    assert not [f for f in glob.glob('/dev/shm/psm_*') if os.path.getmtime(f) > t_start], 'shared staging segments must be unlinked'


def test_process_pool_falls_back_to_threads_without_shared_memory(tmp_path, monkeypatch):
    """decode_procs is the trainer's default; on a host whose /dev/shm cannot hold the staging slots the loader must warn and
    take the thread path (same batches) instead of dying in hipHostRegister or with SIGBUS in a worker"""
    from PIL import Image
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(2)
    lines = []
    for i in range(4):
        Image.fromarray(rng.integers(0, 255, (48, 56 + 2 * i, 3), dtype=np.uint8)).save(tmp_path / ('%d.png' % i))
        lines.append('%d.png 0.5 0.5 0.2 0.2 0' % i)
    (tmp_path / 'label.txt').write_text('\n'.join(lines) + '\n')
    args = (str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 2)
    ref = list(FileUtil.get_dataset(*args, is_augment=False, is_test=True))
    monkeypatch.setattr(FileUtil, '_shm_free_bytes', staticmethod(lambda: 1 << 20))
    with pytest.warns(UserWarning, match='falling back to the decode THREADS'):
        got = list(FileUtil.get_dataset(*args, is_augment=False, is_test=True, decode_procs=2))
    assert len(got) == len(ref) == 2
    for (xa, ya, pa), (xb, yb, pb) in zip(ref, got):
        assert torch.equal(xa, xb) and np.array_equal(ya, yb) and pa == pb


def test_augment_image_entry_point():
    """DatasetUtil.augment_image (reference dataset_util.py:105-115: map _augment over a stream of images): the same draws through the
    oracle's augment give the same pixels; the channel order of the input is kept; non-8-bit inputs are refused"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from oracle import dataset as ods
    from yolov3_tensorflow_amd.dataset.dataset_util import DatasetUtil
    rng = np.random.RandomState(5)
    imgs = [(rng.randint(0, 256, size=(48, 64, 3)).astype(np.float32) * np.float32(1.0 / 255)) for _ in range(3)]
    got = [t.cpu().numpy() for t in DatasetUtil.augment_image(iter(imgs), seed=77)]
    draw_rng = np.random.RandomState(77)
    for x, y in zip(imgs, got):
        d = DatasetUtil.draw(draw_rng)
        want = ods.augment(x, d['noise'], d['color_order'], d['brightness_delta'], d['saturation_factor'], d['contrast_factor'],
                           (d['seed0'], d['seed1']), 0)
        assert y.shape == x.shape and y.min() >= 0.0 and y.max() <= 1.0
        np.testing.assert_allclose(y, want, rtol=0, atol=3e-6)
    with pytest.raises(ValueError):
        list(DatasetUtil.augment_image([np.full((8, 8, 3), 0.3337, np.float32)]))
