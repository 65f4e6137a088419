"""ResNet-18 variant of the reference (backbone/resnet18.py:14-69): 3x3 s2 stem, BN -> max-pool -> ReLU, four modules of two
blocks; the first block of EVERY module has a 1x1-conv + BN shortcut."""
from yolov3_tensorflow_amd.backbone.basic_backbone import BasicBackbone


class ResNet18(BasicBackbone):

    @classmethod
    def _residual_block(cls, input_x, filters, is_nin=True, **conv_params):
        """reference :17-35"""
        residual = cls.conv_bn(input_x, filters, **conv_params)
        residual = cls.activation(residual)
        conv_params.update(strides=(1, 1))
        residual = cls.conv_bn(residual, filters, **conv_params)
        identity = cls.element_wise_add(input_x, residual, is_nin=is_nin)
        return cls.activation(identity)

    @classmethod
    def _residual_module(cls, input_x, filters, **conv_params):
        """reference :37-50"""
        first_block = cls._residual_block(input_x, filters, is_nin=True, **conv_params)
        return cls._residual_block(first_block, filters, is_nin=False)

    @classmethod
    def build(cls, input_x):
        """reference :52-69 -> (stride-8, stride-16, stride-32) features"""
        net = cls.conv_bn(input_x, filters=64, kernel_size=(3, 3), strides=(2, 2), padding='same')
        net = cls.max_pooling(net)
        net = cls.activation(net)
        net = cls._residual_module(net, filters=64)
        sub_stride_8_net = cls._residual_module(net, filters=128, strides=(2, 2))
        sub_stride_16_net = cls._residual_module(sub_stride_8_net, filters=256, strides=(2, 2))
        sub_stride_32_net = cls._residual_module(sub_stride_16_net, filters=512, strides=(2, 2))
        return sub_stride_8_net, sub_stride_16_net, sub_stride_32_net
