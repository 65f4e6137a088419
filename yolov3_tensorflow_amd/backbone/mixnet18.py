""""MixNet-18" of the reference (backbone/mixnet18.py:12-82), described as data: the ResNet-18 skeleton of resnet18.py whose second
convolution of every block is a mixed depthwise convolution -- the channels are cut at [0, 1/2, 3/4, 7/8, 1] of the width and the four
groups go through depthwise 3x3 / 5x5 / 7x7 / 9x9 kernels, each followed by its own BatchNorm, then concatenated."""
import numpy as np
from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone
from yolov3_tensorflow_amd.backbone.resnet18 import STAGES, BLOCKS_PER_STAGE


class MixNet18(BasicBackbone):
    MIX_KERNEL_SIZES = [(3, 3), (5, 5), (7, 7), (9, 9)]                     # reference :18
    MIX_KERNEL_RATIO = np.cumsum([0, 8, 4, 2, 2]) / 16.0                    # reference :19-20: cumulative share of the channels

    @classmethod
    def _mix_block(cls, x, width, stride, project):
        """conv3x3(stride)-BN-ReLU-mixconv(+BN per group) + shortcut (1x1 conv + BN when ``project``), ReLU after the sum (reference :22-48)"""
        branch = cls.activation(cls.conv_bn(x, width, strides=(stride, stride)))
        bounds = [int(v) for v in (width * cls.MIX_KERNEL_RATIO).astype(np.int64)]
        branch = branch.g.mix_depthwise_conv_bn(branch, bounds, [k for k, _ in cls.MIX_KERNEL_SIZES])
        return cls.activation(cls.element_wise_add(x, branch, is_nin=project))

    @classmethod
    def build(cls, input_x):
        """-> the (stride-8, stride-16, stride-32) feature maps (reference :65-82)"""
        x = cls.activation(cls.max_pooling(cls.conv_bn(input_x, filters=64, kernel_size=(3, 3), strides=(2, 2), padding='same')))
        taps = []
        for width, stride in STAGES:
            for b in range(BLOCKS_PER_STAGE):
                x = cls._mix_block(x, width, stride if b == 0 else 1, project=(b == 0))
            taps.append(x)
        return tuple(taps[1:])
