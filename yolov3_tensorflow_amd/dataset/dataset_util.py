"""DatasetUtil.augment_image: the reference's random noise / colour augmentation menu (dataset/dataset_util.py:19-115) on a float32
BGR image in [0, 1]: one of {salt-and-pepper p = 0.01, gaussian sigma = 0.01, none} and one of {3 colour re-orderings, none}, then clip."""
import numpy as np


class DatasetUtil(object):
    SALT_PEPPER_P = 0.01     # reference :22-27
    GAUSSIAN_STD = 0.01

    @staticmethod
    def augment_image(image, rng):
        choice = rng.randint(0, 3)
        if choice == 0:
            mask = rng.uniform(size=image.shape[:2] + (1,))
            image = np.where(mask < DatasetUtil.SALT_PEPPER_P / 2, 0.0, np.where(mask > 1 - DatasetUtil.SALT_PEPPER_P / 2, 1.0, image))
        elif choice == 1:
            image = image + rng.normal(0.0, DatasetUtil.GAUSSIAN_STD, size=image.shape)
        color = rng.randint(0, 4)
        if color == 1:
            image = image[..., [1, 2, 0]]
        elif color == 2:
            image = image[..., [2, 0, 1]]
        elif color == 3:
            image = image[..., [0, 2, 1]]
        return np.clip(image, 0.0, 1.0).astype(np.float32)
