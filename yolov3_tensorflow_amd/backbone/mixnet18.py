""""MixNet-18" of the reference (backbone/mixnet18.py:12-82): ResNet-18 skeleton whose second conv of every block is a
mixed depthwise conv (channel groups [0,1/2,3/4,7/8,1] x kernels 3/5/7/9, each followed by BN, concatenated)."""
import numpy as np
from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone


class MixNet18(BasicBackbone):
    MIX_KERNEL_SIZES = [(3, 3), (5, 5), (7, 7), (9, 9)]                     # reference :18
    MIX_KERNEL_RATIO = np.array([0, 8, 4, 2, 2], dtype=np.float64)          # reference :19-20
    MIX_KERNEL_RATIO = MIX_KERNEL_RATIO.cumsum() / MIX_KERNEL_RATIO.sum()

    @classmethod
    def _mix_residual_block(cls, input_x, filters, is_nin=True, **conv_params):
        """reference :22-48"""
        residual = cls.conv_bn(input_x, filters, **conv_params)
        residual = cls.activation(residual)
        mix_kernel_nums = (filters * cls.MIX_KERNEL_RATIO).astype(np.int64)         # :38-39
        mix_residuals = residual.g.mix_depthwise_conv_bn(residual, [int(v) for v in mix_kernel_nums],
                                                         [k[0] for k in cls.MIX_KERNEL_SIZES])   # :41-45
        identity = cls.element_wise_add(input_x, mix_residuals, is_nin=is_nin)
        return cls.activation(identity)

    @classmethod
    def _mix_residual_module(cls, input_x, filters, **conv_params):
        """reference :50-63"""
        first_block = cls._mix_residual_block(input_x, filters, is_nin=True, **conv_params)
        return cls._mix_residual_block(first_block, filters, is_nin=False)

    @classmethod
    def build(cls, input_x):
        """reference :65-82"""
        net = cls.conv_bn(input_x, filters=64, kernel_size=(3, 3), strides=(2, 2), padding='same')
        net = cls.max_pooling(net)
        net = cls.activation(net)
        net = cls._mix_residual_module(net, filters=64)
        sub_stride_8_net = cls._mix_residual_module(net, filters=128, strides=(2, 2))
        sub_stride_16_net = cls._mix_residual_module(sub_stride_8_net, filters=256, strides=(2, 2))
        sub_stride_32_net = cls._mix_residual_module(sub_stride_16_net, filters=512, strides=(2, 2))
        return sub_stride_8_net, sub_stride_16_net, sub_stride_32_net
