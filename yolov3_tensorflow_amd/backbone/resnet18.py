"""ResNet-18 variant of the reference (backbone/resnet18.py:14-69), described as data: a 3x3 stride-2 stem followed by
BatchNorm -> max-pool -> ReLU, then four stages of two basic blocks.  Unlike torchvision's ResNet the first block of EVERY stage
(also the stride-1 one) has a 1x1-conv + BatchNorm projection shortcut.  Layers are created in the reference's order, so the Keras
auto-names (conv2d_k, batch_normalization_v1_k) line up with its checkpoints."""
from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))          # (width, stride of the stage's first block): outputs at /4, /8, /16, /32
BLOCKS_PER_STAGE = 2


class ResNet18(BasicBackbone):

    @classmethod
    def _basic_block(cls, x, width, stride, project):
        """conv3x3(stride)-BN-ReLU-conv3x3-BN, plus the shortcut (1x1 conv + BN when ``project``), ReLU after the sum (reference :17-35)"""
        branch = cls.activation(cls.conv_bn(x, width, strides=(stride, stride)))
        branch = cls.conv_bn(branch, width, strides=(1, 1))
        return cls.activation(cls.element_wise_add(x, branch, is_nin=project))

    @classmethod
    def build(cls, input_x):
        """-> the (stride-8, stride-16, stride-32) feature maps (reference :52-69)"""
        x = cls.activation(cls.max_pooling(cls.conv_bn(input_x, filters=64, kernel_size=(3, 3), strides=(2, 2), padding='same')))
        taps = []
        for width, stride in STAGES:
            for b in range(BLOCKS_PER_STAGE):
                x = cls._basic_block(x, width, stride if b == 0 else 1, project=(b == 0))
            taps.append(x)
        return tuple(taps[1:])
