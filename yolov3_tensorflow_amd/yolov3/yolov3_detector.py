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

    # the two top-down levels of the neck (reference :103-151): kernel / width of the conv applied to the coarser feature before it is
    # up-sampled, width of the 1x1 conv after the concatenation with the backbone tap, width of the 3x3 conv in front of the detection conv
    TOP_DOWN = (dict(top_kernel=(3, 3), top_width=256, merge_width=256, out_width=512),      # /32 -> /16
                dict(top_kernel=(1, 1), top_width=128, merge_width=128, out_width=256))      # /16 -> /8

    def _cbr(self, x, width, kernel_size):
        """conv -> BatchNorm -> ReLU with the backbone's factories"""
        return self.backbone.activation(self.backbone.conv_bn(x, filters=width, kernel_size=kernel_size, strides=(1, 1)))

    def _detect_conv(self, net, channel_num, name):
        """the detection convolution: 1x1, RandomNormal(0.01) kernel, bias, no regulariser (reference :98-100,123-125,148-150)"""
        return self.backbone.convolution(net, channel_num, kernel_size=(1, 1), use_bias=True, name=name,
                                         kernel_initializer='random_normal_0.01')

    def _detection_head(self, nets, head_channel_nums, head_names):
        """FPN-style neck + the three detection convolutions (reference :61-151) -> (head_8, head_16, head_32) raw outputs (float32,
        channel-padded).  Layers are created coarse to fine, as in the reference: /32 head, then the /16 level, then the /8 level; the
        feature handed down from the /16 level is its merged 1x1 output, the /8 level detects on its 3x3 output."""
        tap_8, tap_16, tap_32 = nets
        outputs = [self._detect_conv(self._cbr(tap_32, 512, (3, 3)), head_channel_nums[2], head_names[2])]
        coarse = tap_32
        for level, tap, channels, name in zip(self.TOP_DOWN, (tap_16, tap_8), (head_channel_nums[1], head_channel_nums[0]),
                                              (head_names[1], head_names[0])):
            g = coarse.g
            top = self._cbr(coarse, level['top_width'], level['top_kernel'])
            merged = self._cbr(g.concat(g.up_sample(top), tap), level['merge_width'], (1, 1))
            feature = self._cbr(merged, level['out_width'], (3, 3))
            outputs.append(self._detect_conv(feature, channels, name))
            coarse = merged
        head_32, head_16, head_8 = outputs
        return head_8, head_16, head_32
