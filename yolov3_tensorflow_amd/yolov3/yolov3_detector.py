"""YOLOv3Detector with the reference's constants and signatures (yolov3/yolov3_detector.py:15-151).  ``build`` returns a
``YOLOv3Model`` (yolov3_tensorflow_amd.model) instead of a keras Model; the three heads are kept as separate NHWC tensors on
the device and are merged into the reference's (N, H/32, W/32, C) layout (:80-85) only at the ``predict`` API edge."""
import logging
from yolov3_tensorflow_amd.backbone.resnet18 import ResNet18
from yolov3_tensorflow_amd.backbone.resnet18_v2 import ResNet18_v2
from yolov3_tensorflow_amd.backbone.mixnet18 import MixNet18


class YOLOv3Detector(object):
    BACKBONE_RESNET_18 = 'resnet-18'
    BACKBONE_RESNET_18_V2 = 'resnet-18-v2'
    BACKBONE_RESNEXT_18 = 'resnext-18'          # constants kept for configs written against the reference;
    BACKBONE_MIXNET_18 = 'mixnet-18'
    BACKBONE_MOBILENET_V2 = 'mobilenet-v2'      # resnext / mobilenet are outside the hot-path scope
    BACKBONE_TYPE = {
        BACKBONE_RESNET_18: ResNet18,
        BACKBONE_RESNET_18_V2: ResNet18_v2,
        BACKBONE_MIXNET_18: MixNet18,
    }

    def __init__(self, backbone_name):
        """reference :32-42"""
        logging.info('building YOLOv3 model, backbone: %s', backbone_name)
        self.backbone_name = backbone_name
        if backbone_name in self.BACKBONE_TYPE.keys():
            self.backbone = self.BACKBONE_TYPE[backbone_name]
        else:
            raise ValueError('no such backbone type!')

    def build(self, input_image_size, head_channel_nums, head_names, batch_size=None, device=None, seed=800):
        """reference :44-59.  ``batch_size``/``device`` are extra (the native graph is static); defaults come from configs."""
        if len(input_image_size) != 3:
            raise Exception('model input shape must have 3 dimensions')
        from yolov3_tensorflow_amd.model import YOLOv3Model
        return YOLOv3Model(self, [int(v) for v in input_image_size], [int(c) for c in head_channel_nums], list(head_names),
                           batch_size=batch_size, device=device, seed=seed)

    def _detection_head(self, nets, head_channel_nums, head_names):
        """reference :61-86 -> (head_8, head_16, head_32) raw conv outputs (float32, channel-padded)"""
        sub_stride_8_net, sub_stride_16_net, sub_stride_32_net = nets
        stride_8_channel_num, stride_16_channel_num, stride_32_channel_num = head_channel_nums
        stride_8_head_name, stride_16_head_name, stride_32_head_name = head_names
        head_32_feature = self._yolov3_stride_32_head(sub_stride_32_net, stride_32_channel_num, stride_32_head_name)
        merge_net, head_16_feature = self._yolov3_stride_16_head(sub_stride_32_net, sub_stride_16_net,
                                                                 stride_16_channel_num, stride_16_head_name)
        head_8_feature = self._yolov3_stride_8_head(merge_net, sub_stride_8_net, stride_8_channel_num, stride_8_head_name)
        return head_8_feature, head_16_feature, head_32_feature

    def _detect_conv(self, net, channel_num, name):
        """keras Conv2D(1x1, RandomNormal(0.01), bias, no regulariser) (reference :98-100,123-125,148-150)"""
        return self.backbone.convolution(net, channel_num, kernel_size=(1, 1), use_bias=True, name=name,
                                         kernel_initializer='random_normal_0.01')

    def _yolov3_stride_32_head(self, sub_stride_32_net, stride_32_channel_num, stride_32_head_name):
        """reference :88-101"""
        net = self.backbone.conv_bn(sub_stride_32_net, 512)
        net = self.backbone.activation(net)
        return self._detect_conv(net, stride_32_channel_num, stride_32_head_name)

    def _yolov3_stride_16_head(self, stride_32_feature, sub_stride_16_net, stride_16_channel_num, stride_16_head_name):
        """reference :103-126"""
        g = stride_32_feature.g
        net = self.backbone.conv_bn(stride_32_feature, filters=256, strides=(1, 1))
        net = self.backbone.activation(net)
        merge_net = g.concat(g.up_sample(net), sub_stride_16_net)
        merge_net = self.backbone.conv_bn(merge_net, filters=256, kernel_size=(1, 1))
        merge_net = self.backbone.activation(merge_net)
        net = self.backbone.conv_bn(merge_net, filters=512, kernel_size=(3, 3))
        net = self.backbone.activation(net)
        return merge_net, self._detect_conv(net, stride_16_channel_num, stride_16_head_name)

    def _yolov3_stride_8_head(self, stride_16_feature, sub_stride_8_net, stride_8_channel_num, stride_8_head_name):
        """reference :128-151"""
        g = stride_16_feature.g
        net = self.backbone.conv_bn(stride_16_feature, filters=128, kernel_size=(1, 1))
        net = self.backbone.activation(net)
        merge_net = g.concat(g.up_sample(net), sub_stride_8_net)
        merge_net = self.backbone.conv_bn(merge_net, filters=128, kernel_size=(1, 1))
        merge_net = self.backbone.activation(merge_net)
        merge_net = self.backbone.conv_bn(merge_net, filters=256, kernel_size=(3, 3))
        merge_net = self.backbone.activation(merge_net)
        return self._detect_conv(merge_net, stride_8_channel_num, stride_8_head_name)
