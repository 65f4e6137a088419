"""FileUtil.get_dataset with the reference's arguments (dataset/file_util.py:62-114): label file ``name cx cy w h cls ...`` ->
an infinite (train) or single-pass (test) iterator of full batches
    images float32 (N, H, W, 3) in [0, 1], BGR, letterboxed with nearest-neighbour resize  -- a DEVICE tensor,
    labels float32 (N, T*5) = (cx, cy, w, h, cls) normalised to the letterboxed image, padded with -1
(+ image paths when is_test), i.e. the tensors the reference's tf.data pipeline feeds to keras fit (file_util.py:54-59,95-97).
The host only decodes the JPEGs (PIL; TensorFlow's decoder is not available, decoded pixels can differ in the last bit) and transforms
the few label numbers; resize, normalisation, channel order and augmentation run in one GPU kernel over the whole batch
(DeviceImagePipeline -> yolo_letterbox_augment).  Image and label stay paired by construction (the reference relies on identical
shuffle seeds of two dataset branches, file_util.py:79-88)."""
import ctypes
import math
import os
import numpy as np
from yolov3_tensorflow_amd.dataset.dataset_util import DatasetUtil


class DeviceImagePipeline(object):
    """one batch of decoded uint8 RGB images (any sizes) -> float32 (N, H, W, 3) BGR [0, 1] on the GPU (and, optionally, the packed
    bf16 conv input).  Owns a pinned staging buffer so the upload of batch k+1 can overlap the kernels of batch k."""

    def __init__(self, batch_size, image_size, device=None, max_pixels_per_image=4096 * 4096):
        import torch
        from yolov3_tensorflow_amd import ops, _lib
        if not torch.cuda.is_available():
            raise RuntimeError('the input pipeline runs on the GPU (yolo_letterbox_augment): no GPU is visible and there is no CPU fallback')
        self.torch, self.ops, self.lib = torch, ops, _lib
        self.N, (self.H, self.W) = int(batch_size), (int(image_size[0]), int(image_size[1]))
        self.device = torch.device('cuda:%d' % torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_pixels = int(max_pixels_per_image)
        self._stage = None
        self._dev = None
        self.workspace = torch.empty(ops.letterbox_workspace_bytes(self.N), dtype=torch.uint8, device=self.device)
        self._desc_host = torch.empty(self.N * ctypes.sizeof(_lib.ImageDesc), dtype=torch.uint8).pin_memory()
        self._desc_dev = torch.empty_like(self._desc_host, device=self.device)

    @staticmethod
    def geometry(h, w, H, W):
        """tf.image.resize_image_with_pad: ratio = max(w/W, h/H) in float64, resized = floor(dim / ratio), offset = floor((target - dim/ratio) / 2)"""
        ratio = max(float(w) / float(W), float(h) / float(H))
        rh, rw = float(h) / ratio, float(w) / ratio
        return int(math.floor(rh)), int(math.floor(rw)), max(0, int(math.floor((H - rh) / 2))), max(0, int(math.floor((W - rw) / 2)))

    def __call__(self, images_rgb_u8, draws=None, out_f32=True, out_bf16x8=None):
        """images_rgb_u8: list of N uint8 (h, w, 3) arrays; draws: list of DatasetUtil.draw() dicts (None = no augmentation)"""
        torch = self.torch
        if len(images_rgb_u8) != self.N:
            raise ValueError('expected %d images, got %d' % (self.N, len(images_rgb_u8)))
        total = 0
        descs = (self.lib.ImageDesc * self.N)()
        for n, im in enumerate(images_rgb_u8):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('image %d must be uint8 (h, w, 3), got %s %s' % (n, im.dtype, im.shape))
            h, w = im.shape[:2]
            if h * w > self.max_pixels or h < 1 or w < 1:
                raise ValueError('image %d has unsupported size %dx%d' % (n, h, w))
            nh, nw, top, left = self.geometry(h, w, self.H, self.W)
            if nh < 1 or nw < 1:
                raise ValueError('image %d (%dx%d) collapses to nothing at %dx%d' % (n, h, w, self.H, self.W))
            d = descs[n]
            d.offset, d.h, d.w, d.nh, d.nw, d.top, d.left = total, h, w, nh, nw, top, left
            for k, v in (draws[n] if draws is not None else DatasetUtil.NO_AUGMENT).items():
                setattr(d, k, v)
            total += (h * w * 3 + 15) // 16 * 16
        if self._stage is None or self._stage.numel() < total:
            self._stage = torch.empty(int(total * 1.25), dtype=torch.uint8).pin_memory()
            self._dev = torch.empty(self._stage.numel(), dtype=torch.uint8, device=self.device)
        else:
            torch.cuda.current_stream(self.device).synchronize()       # the previous batch's upload has left the staging buffer
        stage = self._stage.numpy()
        for n, im in enumerate(images_rgb_u8):
            o = descs[n].offset
            stage[o:o + im.size] = np.ascontiguousarray(im).reshape(-1)
        ctypes.memmove(self._desc_host.data_ptr(), ctypes.addressof(descs), ctypes.sizeof(descs))
        self._dev[:total].copy_(self._stage[:total], non_blocking=True)
        self._desc_dev.copy_(self._desc_host, non_blocking=True)
        out = torch.empty(self.N, self.H, self.W, 3, device=self.device) if out_f32 else None
        self.ops.letterbox_augment(self._dev, self._desc_dev, self.N, self.H, self.W, draws is not None, self.workspace, out_f32=out,
                                   out_bf16x8=out_bf16x8)
        return out


class FileUtil(object):

    @staticmethod
    def _parse_label_file(label_path):
        """reference :21-38"""
        names, labels = [], []
        with open(label_path) as f:
            for line in f:
                parts = line.split()
                if not parts:
                    continue
                names.append(parts[0])
                labels.append(np.asarray(parts[1:], dtype=np.float32).reshape(-1, 5))
        return names, labels

    @staticmethod
    def transform_label(label, src_hw, image_size):
        """reference :48-53: xy' = xy * r + (1 - r) / 2, wh' = wh * r with r = (src_hw / dst_hw)[::-1] / max(src_hw / dst_hw), float32"""
        lab = np.asarray(label, dtype=np.float32).reshape(-1, 5).copy()
        if len(lab):
            src = np.asarray(src_hw, dtype=np.float32) / np.asarray(image_size, dtype=np.float32)
            ratio = src[::-1] / np.max(src)                      # (rw, rh): the side that fills the canvas has ratio 1
            lab[:, 0:2] = lab[:, 0:2] * ratio + (np.float32(1) - ratio) / np.float32(2.0)
            lab[:, 2:4] = lab[:, 2:4] * ratio
        return lab

    @staticmethod
    def read_image(path):
        """tf.read_file + tf.image.decode_jpeg (reference :44-45) -> uint8 (h, w, 3) RGB"""
        from PIL import Image
        return np.asarray(Image.open(path).convert('RGB'))

    @staticmethod
    def host_batches(file_path, image_dir, image_size, batch_size, is_augment=True, is_test=False, seed=800):
        """the host half of get_dataset: (decoded images, padded transformed labels (N, T*5), augmentation draws or None, paths)"""
        names, labels = FileUtil._parse_label_file(file_path)
        if not names:
            raise ValueError('empty label file ' + file_path)
        t_max = max(len(l) for l in labels)
        rng = np.random.RandomState(seed)
        order = np.arange(len(names))
        while True:
            if not is_test:
                rng.shuffle(order)                                   # shuffle-and-repeat (reference :79)
            for i in range(0, len(order) - batch_size + 1 if not is_test else len(order), batch_size):
                idx = order[i:i + batch_size]
                if len(idx) < batch_size:                            # keras fit gets full batches; pad the last test batch by wrap-around
                    idx = np.concatenate([idx, order[:batch_size - len(idx)]])
                imgs, labs = [], -np.ones((batch_size, t_max, 5), dtype=np.float32)
                for k, j in enumerate(idx):
                    im = FileUtil.read_image(os.path.join(image_dir, names[j]))
                    lb = FileUtil.transform_label(labels[j], im.shape[:2], image_size)
                    imgs.append(im)
                    labs[k, :len(lb)] = lb
                draws = [DatasetUtil.draw(rng) for _ in idx] if (is_augment and not is_test) else None
                yield imgs, labs.reshape(batch_size, t_max * 5), draws, [os.path.join(image_dir, names[j]) for j in idx]
            if is_test:
                return

    @staticmethod
    def get_dataset(file_path, image_dir, image_size, batch_size, is_augment=True, is_test=False, seed=800, device=None):
        """reference :62-114"""
        pipe = DeviceImagePipeline(batch_size, image_size, device=device)

        def batches():
            for imgs, labs, draws, paths in FileUtil.host_batches(file_path, image_dir, image_size, batch_size, is_augment, is_test, seed):
                x = pipe(imgs, draws)
                yield (x, labs, paths) if is_test else (x, labs)
        return batches()
