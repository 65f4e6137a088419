"""Pre-activation ResNet-18 of the reference (backbone/resnet18_v2.py:10-74), described as data: a bare 3x3 stride-2 stem (no
BatchNorm) and max-pool, four stages of two pre-activation blocks, and a closing BatchNorm + ReLU on each of the three taps."""
from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone
from yolov3_tensorflow_amd.backbone.resnet18 import STAGES, BLOCKS_PER_STAGE


class ResNet18_v2(BasicBackbone):

    @classmethod
    def _preact_block(cls, x, width, stride, project):
        """BN-ReLU-conv3x3(stride)-BN-ReLU-conv3x3 + shortcut.  The projection shortcut (1x1 conv + BN) branches off the PRE-ACTIVATED
        tensor, the identity shortcut off the block input (reference :33-36); nothing follows the sum, so it is stored as it is."""
        pre = cls.bn_activation(x)
        branch = cls.convolution(pre, filters=width, strides=(stride, stride))
        branch = cls.convolution(cls.bn_activation(branch), filters=width, strides=(1, 1))
        total = cls.element_wise_add(pre if project else x, branch, is_nin=project)
        return total.g.materialize(total, relu=False)

    @classmethod
    def build(cls, input_x):
        """-> the (stride-8, stride-16, stride-32) feature maps, each through a final BN + ReLU (reference :54-74)"""
        x = cls.max_pooling(cls.convolution(input_x, filters=64, kernel_size=(3, 3), strides=(2, 2), padding='same'))
        taps = []
        for width, stride in STAGES:
            for b in range(BLOCKS_PER_STAGE):
                x = cls._preact_block(x, width, stride if b == 0 else 1, project=(b == 0))
            taps.append(x)
        return tuple(cls.bn_activation(t) for t in taps[1:])
