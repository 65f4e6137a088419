"""CPU tests of the rows either side of the hot path: the package's YOLOv3PostProcessor against the golden vectors
generated from the reference's own module (pinned), the letterbox / label transform, the dataset iterator and the label decoder."""
import glob
import os
import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'postprocess_*.npz')))


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_product_postprocessor_matches_reference_goldens(path):
    from yolov3_tensorflow_amd.yolov3.yolov3_post_process import YOLOv3PostProcessor as P
    g = np.load(path)
    h = [(g['pred%d' % i], g['boxes%d' % i]) for i in range(3)]
    hs = P.filter_boxes(h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], float(g['score_thresh']))
    for i in range(3):
        np.testing.assert_array_equal(np.asarray(hs[i], dtype=np.float64).reshape(-1, 8), g['filtered%d' % i])
        sc = h[i][0][..., 4] * (h[i][0][..., 5:].max(-1) if h[i][0].shape[-1] > 5 else 1.0)
        np.testing.assert_array_equal(P.filter_indices(h[i][0], float(g['score_thresh'])), np.flatnonzero(sc.reshape(-1) > float(g['score_thresh'])))
    nms = P.apply_nms([np.array(b) for b in hs], float(g['nms_thresh']))
    res = P.resize_boxes(nms, g['target_size'])
    for i in range(3):
        np.testing.assert_array_equal(np.asarray(nms[i], dtype=np.float64).reshape(-1, 9), g['nms%d' % i])
        np.testing.assert_array_equal(np.asarray(res[i], dtype=np.float64).reshape(-1, 9), g['resized%d' % i])


def test_letterbox_and_label_transform():
    """oracle letterbox (the published resize_image_with_pad / nearest-neighbour algorithm), the product's geometry and label transform"""
    from oracle import dataset as ods
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil, DeviceImagePipeline
    img = np.zeros((100, 200, 3), np.uint8)
    img[:, :, 0] = np.arange(200)[None, :]
    lab = np.array([[0.5, 0.5, 0.2, 0.4, 3.0], [0.1, 0.9, 0.1, 0.1, 1.0]], np.float32)
    out = ods.letterbox(img, (160, 160))
    assert out.shape == (160, 160, 3)
    assert out[:40].sum() == 0 and out[120:].sum() == 0 and out[40:120].any()          # 200x100 -> 160x80 centred, 40 px bars
    np.testing.assert_array_equal(out[40, :, 0], (np.arange(160) * 200 // 160).astype(np.uint8))   # nearest neighbour columns
    assert DeviceImagePipeline.geometry(100, 200, 160, 160) == (80, 160, 40, 0) == ods.letterbox_geometry(100, 200, 160, 160)
    assert DeviceImagePipeline.geometry(500, 353, 320, 320) == (320, 225, 0, 47)       # floor(225.92), floor((320 - 225.92) / 2)
    # xy' = xy*r + (1-r)/2, wh' = wh*r with r = (1, 0.5) for (x, y) (dataset/file_util.py:47-55)
    l2 = FileUtil.transform_label(lab, (100, 200), (160, 160))
    np.testing.assert_allclose(l2[0], [0.5, 0.5, 0.2, 0.2, 3.0], rtol=1e-6)
    np.testing.assert_allclose(l2[1], [0.1, 0.7, 0.1, 0.05, 1.0], rtol=1e-6)
    np.testing.assert_array_equal(l2, ods.transform_label(lab, 100, 200, (160, 160)))


def test_augmentation_oracle_properties():
    """oracle/dataset.py: Philox known-answer vectors (Random123 kat_vectors), noise rates, and the colour ops' identities"""
    from oracle import dataset as ods
    kat = ods.philox4x32(np.array([[0, 0, 0, 0], [0xffffffff] * 4, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32),
                         np.array([0, 0], np.uint32))[0]
    np.testing.assert_array_equal(kat, np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint32))
    k2 = ods.philox4x32(np.array([[0xffffffff] * 4], np.uint32), np.array([0xffffffff, 0xffffffff], np.uint32))[0]
    np.testing.assert_array_equal(k2, np.array([0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd], np.uint32))
    k3 = ods.philox4x32(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32), np.array([0xa4093822, 0x299f31d0], np.uint32))[0]
    np.testing.assert_array_equal(k3, np.array([0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1], np.uint32))
    rng = np.random.RandomState(0)
    img = rng.uniform(0.05, 0.95, size=(96, 128, 3)).astype(np.float32)
    sp = ods.pixel_noise(img, 0, (1, 2), 0)
    changed = (sp != img).any(-1)
    assert 0.004 < changed.mean() < 0.02 and set(np.unique(sp[changed])) <= {0.0, 1.0}
    ga = ods.pixel_noise(img, 1, (1, 2), 0) - img
    assert abs(ga.std() - 0.01) < 1e-3 and abs(ga.mean()) < 3e-4
    np.testing.assert_allclose(ods.adjust_saturation(img, 1.0), img, atol=2e-6)          # HSV round trip
    grey = ods.adjust_saturation(img, 0.0)
    np.testing.assert_allclose(grey, np.repeat(img.max(-1, keepdims=True), 3, -1), atol=1e-6)     # S = 0 -> (V, V, V)
    out = ods.augment(img, 2, 2, 0.05, 1.0, 1.0, (0, 0), 0)                              # only brightness acts
    np.testing.assert_allclose(out, np.clip(img + np.float32(0.05), 0, 1), atol=2e-6)
    out = ods.augment(img, 2, 3, 0.05, 1.1, 0.9, (0, 0), 0)                              # order 3: nothing but the clip
    np.testing.assert_array_equal(out, img)


def test_dataset_iterator_shapes(tmp_path):
    from PIL import Image
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    from yolov3_tensorflow_amd.dataset.dataset_util import DatasetUtil
    rng = np.random.default_rng(0)
    lines = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 255, (40 + 4 * i, 60, 3), dtype=np.uint8)).save(tmp_path / ('%d.jpg' % i))
        k = 1 + i % 3
        lines.append('%d.jpg ' % i + ' '.join('0.5 0.5 0.2 0.2 %d' % j for j in range(k)))
    (tmp_path / 'label.txt').write_text('\n'.join(lines) + '\n')
    it = FileUtil.host_batches(str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 2, is_augment=True, is_test=False)
    for _ in range(4):                                   # infinite, always full batches (reference file_util.py:79)
        imgs, y, draws, paths = next(it)
        assert len(imgs) == 2 and all(im.dtype == np.uint8 and im.shape[1:] == (60, 3) for im in imgs) and len(paths) == 2
        assert y.shape == (2, 15) and ((y == -1) | (y >= 0)).all()
        assert len(draws) == 2 and all(d['noise'] in (0, 1, 2) and d['color_order'] in (0, 1, 2, 3) and
                                       abs(d['brightness_delta']) <= DatasetUtil._random_brightness and
                                       0.9 <= d['saturation_factor'] <= 1.1 and 0.9 <= d['contrast_factor'] <= 1.1 for d in draws)
    test_batches = list(FileUtil.host_batches(str(tmp_path / 'label.txt'), str(tmp_path), (64, 64), 2, is_augment=False, is_test=True))
    assert all(b[2] is None for b in test_batches)       # no augmentation draws in test mode
    assert len(test_batches) == 3 and len(test_batches[0]) == 4 and len(test_batches[0][3]) == 2


def test_label_decoder_matches_oracle():
    import torch
    from yolov3_tensorflow_amd.yolov3.label_decoder import LabelDecoder
    from oracle.loss import LabelDecoderOracle
    grids = [(40, 48), (20, 24), (10, 12)]
    t = -np.ones((2, 15), np.float32)
    t[0, :5] = [0.3, 0.6, 0.4, 0.2, 1]
    t[1, :10] = [0.5, 0.5, 0.1, 0.9, 0, 0.2, 0.8, 0.3, 0.3, 2]
    got = LabelDecoder(grids).decode(t)
    ref = LabelDecoderOracle(grids).decode(torch.as_tensor(t))
    for (a, b), (c, d) in zip(got, ref):
        np.testing.assert_array_equal(a, c.numpy())
        np.testing.assert_array_equal(b, d.numpy())


@pytest.mark.parametrize('seed', range(6))
def test_product_nms_matches_oracle_on_adversarial_rows(seed):
    """the product's vectorised host NMS against the pinned oracle on rows with tied scores, duplicate boxes, degenerate (zero-area) boxes,
    empty heads and few classes -- the cases where sort stability, the `w <= 0` short-cut and the per-head id quirk decide the outcome"""
    from oracle import postprocess as pp
    from yolov3_tensorflow_amd.yolov3.yolov3_post_process import YOLOv3PostProcessor as P
    rng = np.random.default_rng(seed)
    heads = []
    for k in (int(rng.integers(0, 40)), 0 if seed % 2 else int(rng.integers(1, 25)), int(rng.integers(0, 30))):
        c = rng.uniform(0.2, 0.8, size=(k, 2))
        half = rng.uniform(0.0, 0.25, size=(k, 2)) * (rng.uniform(size=(k, 1)) > 0.1)         # some zero-area boxes
        rows = np.concatenate([c - half, c + half, rng.uniform(0.5, 1, (k, 1)), rng.uniform(0.5, 1, (k, 1)),
                               rng.integers(0, 3, (k, 1)).astype(np.float64), np.round(rng.uniform(0.3, 1, (k, 1)), 1)], axis=1)   # tied scores
        if k > 4:
            rows[1] = rows[0]                                                                  # exact duplicates
            rows[3, :4] = rows[2, :4]
        heads.append(rows)
    for quirk in (True, False):
        want = pp.apply_nms([h.copy() for h in heads], 0.35, reference_quirk=quirk)
        got = P.apply_nms([h.copy() for h in heads], 0.35, fixed_indices=not quirk)
        for w, g in zip(want, got):
            np.testing.assert_array_equal(np.asarray(g, dtype=np.float64).reshape(-1, 9), np.asarray(w, dtype=np.float64).reshape(-1, 9))
    size = np.array([480, 384, 480, 384])
    for w, g in zip(pp.resize_boxes(want, size), P.resize_boxes(got, size)):
        np.testing.assert_array_equal(np.asarray(g, dtype=np.float64).reshape(-1, 9), np.asarray(w, dtype=np.float64).reshape(-1, 9))


def test_data_parallel_ranks_read_disjoint_slices_of_the_single_process_batches(tmp_path):
    """two ranks with per-rank batch 2 == one process with batch 4 (keras multi_gpu_model splits the one batch on axis 0, reference
    trainer.py:40-43): every global batch is covered exactly once, the ranks' slices are disjoint, and images, labels AND augmentation
    draws concatenate to what the single process sees"""
    from PIL import Image
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    rng = np.random.default_rng(1)
    lines = []
    for i in range(9):
        Image.fromarray(rng.integers(0, 255, (32 + 2 * i, 48, 3), dtype=np.uint8)).save(tmp_path / ('%d.jpg' % i))
        lines.append('%d.jpg 0.5 0.5 0.2 0.2 %d' % (i, i))
    label = str(tmp_path / 'label.txt')
    (tmp_path / 'label.txt').write_text('\n'.join(lines) + '\n')
    single = FileUtil.host_batches(label, str(tmp_path), (64, 64), 4, is_augment=True)
    ranks = [FileUtil.host_batches(label, str(tmp_path), (64, 64), 2, is_augment=True, rank=r, world=2) for r in range(2)]
    for step in range(5):                                                   # 2 global batches per epoch of 9 images: crosses two reshuffles
        imgs, y, draws, paths = next(single)
        parts = [next(it) for it in ranks]
        assert not set(parts[0][3]) & set(parts[1][3])                      # disjoint
        assert parts[0][3] + parts[1][3] == paths                           # together: the single-process batch, in its order
        np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), y)
        assert parts[0][2] + parts[1][2] == draws
        for a, b in zip(parts[0][0] + parts[1][0], imgs):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        next(FileUtil.host_batches(label, str(tmp_path), (64, 64), 2, rank=2, world=2))


def test_letterbox_matches_the_references_own_frames():
    """SURVEY 8f-1 geometry, pinned against the reference itself: dataset/test_result/*.jpg are the reference's OWN 480 x 384 letterboxed frames
    of its 20 sample images (yolov3_post_process.py:199-204 writes the network input of file_util.py:47-59 with the boxes drawn on it;
    fixture tests/golden/letterbox_frames.npz = those files' bytes, made by tests/golden/make_letterbox_golden.py).  oracle.dataset.letterbox
    of the same sample image must match each frame within JPEG noise -- scale, offset and nearest-neighbour sampling -- and EVERY one-pixel
    shift must be strictly worse.  The drawn box lines / labels are masked by the per-pixel outlier rule (|difference| > 48 grey levels)."""
    import io
    from PIL import Image
    from oracle import dataset as ods
    from yolov3_tensorflow_amd.dataset.file_util import FileUtil
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    d = np.load(os.path.join(here, 'letterbox_frames.npz'))
    names = [str(n) for n in d['names']]
    assert len(names) == 20
    H, W = 384, 480                                          # configs.py:36 input_image_size
    for k, name in enumerate(names):
        ref = np.asarray(Image.open(io.BytesIO(d['jpeg_bytes'][d['offsets'][k]:d['offsets'][k + 1]].tobytes())).convert('RGB')).astype(np.int32)
        assert ref.shape == (H, W, 3)
        src = FileUtil.read_image(os.path.join(here, 'test_sample', 'images', name))
        box = ods.letterbox(src, (H, W, 3)).astype(np.int32)
        nh, nw, top, left = ods.letterbox_geometry(src.shape[0], src.shape[1], H, W)

        def score(dy, dx):
            a = box[top + 2:top + nh - 2, left + 2:left + nw - 2]
            b = ref[top + 2 + dy:top + nh - 2 + dy, left + 2 + dx:left + nw - 2 + dx]
            diff = np.abs(a - b).max(-1)
            keep = diff <= 48
            return float(diff[keep].mean()), float(keep.mean())

        aligned, inliers = score(0, 0)
        assert aligned <= 8.0 and inliers >= 0.95, (name, aligned, inliers)          # measured 3.3 .. 5.4 grey levels, >= 97 % inliers
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, -1), (1, -1), (-1, 1)):
            shifted = score(dy, dx)[0]
            assert shifted >= 1.4 * aligned, (name, (dy, dx), shifted, aligned)        # measured >= 1.9 x
        # the zero bars sit where the geometry says (JPEG ringing only), the first content row / column next to them does not
        # (99th percentile: a drawn box label may reach into a bar)
        for bar in (ref[:max(top - 2, 0)], ref[top + nh + 2:], ref[:, :max(left - 2, 0)], ref[:, left + nw + 2:]):
            if bar.size:
                assert np.percentile(bar, 99) <= 24, (name, bar.shape)
        inside = ref[top + 2:top + nh - 2, left + 2:left + nw - 2]
        assert inside.mean() > 24, name                       # (and the content is not itself dark)


def test_decode_worker_writes_into_a_shared_segment(tmp_path):
    """the decode workers of FileUtil.get_dataset(decode_procs=...): header probe and decode-into-shared-memory (called in process here and
    through the two-process pool the GPU path uses)"""
    from multiprocessing import shared_memory
    from PIL import Image
    from yolov3_tensorflow_amd.dataset import decode_worker
    rng = np.random.default_rng(3)
    paths, imgs = [], []
    for i, (h, w) in enumerate(((30, 44), (51, 20), (8, 8))):
        im = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / ('%d.png' % i))
        Image.fromarray(im).save(p)
        paths.append(p)
        imgs.append(im)
    assert [decode_worker.probe_size(p) for p in paths] == [im.shape[:2] for im in imgs]
    offs = np.cumsum([0] + [(im.size + 15) // 16 * 16 for im in imgs])
    shm = shared_memory.SharedMemory(create=True, size=int(offs[-1]) + 64)
    try:
        tasks = [(shm.name, int(offs[i]), imgs[i].shape[0], imgs[i].shape[1], paths[i]) for i in range(3)]
        pool = decode_worker.DecodePool(2)                     # two worker interpreters (python -m ...decode_worker), JSON lines over pipes
        try:
            assert pool.probe_sizes(paths) == [im.shape[:2] for im in imgs]
            pool.submit(tasks[1:])
            pool.submit([(shm.name, 0, 5, 5, paths[0])])         # planned size != decoded size
            pool.wait_oldest()
            with pytest.raises(RuntimeError, match='decodes to'):
                pool.wait_oldest()                               # ... is loud, and the pool keeps working
            pool.submit([tasks[0]])
            pool.wait_oldest()
        finally:
            pool.close()
        for i, im in enumerate(imgs):
            got = np.ndarray(im.shape, np.uint8, buffer=shm.buf, offset=int(offs[i]))
            np.testing.assert_array_equal(got, im)
    finally:
        shm.close()
        shm.unlink()
