"""MI355X-native (gfx950) YOLOv3 training hot path: hand-written HIP kernels behind the Python surface of
zheng-yuwei/YOLOv3-tensorflow (configs / run / yolov3.trainer / yolov3.yolov3_detector / yolov3.yolov3_loss /
yolov3.yolov3_decoder / yolov3.label_decoder / utils.radam).  See DESIGN.md and INTEGRATION.md at the repository root."""
__version__ = '0.1.0'
