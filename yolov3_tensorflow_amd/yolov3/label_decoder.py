"""LabelDecoder with the reference's signature (yolov3/label_decoder.py:11-60): labels (N, T*5) normalised -> per head (N, T, 5) in grid
units + corner boxes (N, T, 4).  Pure layout arithmetic on a tiny array (float32 NumPy on the host, as the reference's callers outside
the TF graph would see it); the training path does the same multiplication inside the loss kernel."""
import numpy as np


class LabelDecoder(object):
    def __init__(self, head_grid_sizes):
        (self.head_8_height, self.head_8_width), (self.head_16_height, self.head_16_width), (self.head_32_height, self.head_32_width) = \
            [(int(h), int(w)) for (h, w) in head_grid_sizes]
        self.head_8_wh = np.asarray([self.head_8_width, self.head_8_height], dtype=np.float32)
        self.head_16_wh = np.asarray([self.head_16_width, self.head_16_height], dtype=np.float32)
        self.head_32_wh = np.asarray([self.head_32_width, self.head_32_height], dtype=np.float32)

    def decode(self, targets):
        """reference :26-42"""
        targets = np.asarray(targets, dtype=np.float32)
        targets = np.reshape(targets, [targets.shape[0], -1, 5])
        return [self._decode_single_head(targets, wh) for wh in (self.head_8_wh, self.head_16_wh, self.head_32_wh)]

    @staticmethod
    def _decode_single_head(targets, head_wh):
        """reference :44-60"""
        targets_xy = targets[:, :, 0:2] * head_wh
        targets_wh = targets[:, :, 2:4] * head_wh
        out = np.concatenate([targets_xy, targets_wh, targets[:, :, 4:5]], axis=-1)
        half_wh = targets_wh / 2
        return out, np.concatenate([targets_xy - half_wh, targets_xy + half_wh], axis=-1)
