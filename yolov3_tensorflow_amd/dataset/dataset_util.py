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

    @staticmethod
    def augment_image(image_set, seed=800, device=None):
        """reference :105-115: map the augmentation (noise -> one of the colour-op orderings -> clip, :29-99) over a stream of images.
        ``image_set`` yields float32 (H, W, 3) images in [0, 1] (any channel order: the ops are applied per channel and on the channel
        max / min, i.e. order-independent except that the reference's dataset is BGR at this point -- kept as given); yields float32
        device tensors of the same shape.  The pixel work runs in the GPU input kernel (yolo_letterbox_augment with an identity
        letterbox), which reads 8-bit pixels: the images must be k / 255 values, as everything the reference's pipeline produces is
        (file_util.py:57, x / 255 of decoded JPEG bytes)."""
        from yolov3_tensorflow_amd.dataset.file_util import DeviceImagePipeline
        rng = np.random.RandomState(seed)
        pipes = {}
        for image in image_set:
            x = np.asarray(image.cpu() if hasattr(image, 'cpu') else image, dtype=np.float32)
            if x.ndim != 3 or x.shape[2] != 3:
                raise ValueError('augment_image expects (H, W, 3) images, got %s' % (x.shape,))
            u8 = np.rint(x * 255.0)
            if np.abs(u8 - x * 255.0).max() > 1e-3 or u8.min() < 0 or u8.max() > 255:
                raise ValueError('augment_image expects x / 255 of 8-bit pixels in [0, 1] (what FileUtil produces)')
            key = x.shape[:2]
            if key not in pipes:
                pipes[key] = DeviceImagePipeline(1, key, device=device)
            # the kernel turns RGB into BGR on the way (file_util.py:58-59): hand it the channels reversed so that they come out as given
            out = pipes[key]([np.ascontiguousarray(u8.astype(np.uint8)[..., ::-1])], draws=[DatasetUtil.draw(rng)])
            yield out[0]
