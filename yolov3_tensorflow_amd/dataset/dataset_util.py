"""DatasetUtil: the reference's augmentation menu (dataset/dataset_util.py:19-115).  The reference maps ``_augment`` over a tf.data set;
here the per-image scalar decisions (noise type, colour-op order, brightness delta, saturation / contrast factors, noise seed) are drawn
on the host and the pixel work -- noise, brightness, saturation, contrast, clip -- runs inside the GPU input kernel
(yolo_letterbox_augment, csrc/dataset.hip) together with the letterbox resize."""
import numpy as np


class DatasetUtil(object):
    _random_brightness = 30. / 255.      # reference :22-27
    _random_low_contrast = 0.9
    _random_up_contrast = 1.1
    _random_low_saturation = 0.9
    _random_up_saturation = 1.1
    _random_normal = 0.01                # gaussian sigma and salt-and-pepper rate (the kernel's constants)
    NO_AUGMENT = dict(noise=2, color_order=3, brightness_delta=0.0, saturation_factor=1.0, contrast_factor=1.0, seed0=0, seed1=0)

    @staticmethod
    def draw(rng):
        """the scalar random decisions of DatasetUtil._augment for one image (reference :47-49 noise type in [0,3), :81-83 colour order in
        [0,4), tf.image.random_brightness / random_saturation / random_contrast draws)"""
        return dict(noise=int(rng.randint(0, 3)), color_order=int(rng.randint(0, 4)),
                    brightness_delta=float(rng.uniform(-DatasetUtil._random_brightness, DatasetUtil._random_brightness)),
                    saturation_factor=float(rng.uniform(DatasetUtil._random_low_saturation, DatasetUtil._random_up_saturation)),
                    contrast_factor=float(rng.uniform(DatasetUtil._random_low_contrast, DatasetUtil._random_up_contrast)),
                    seed0=int(rng.randint(0, 2 ** 31 - 1)), seed1=int(rng.randint(0, 2 ** 31 - 1)))
