"""Process entry with the reference's mode switch (run.py:123-181): FLAGS.mode in {train, test, predict, save_pb, save_serving}.
    python -m yolov3_tensorflow_amd.run            (single GPU)
    python -m torch.distributed.run --nproc-per-node 8 -m yolov3_tensorflow_amd.run     (data parallel)"""
import logging
import os
import numpy as np

from yolov3_tensorflow_amd import backend
from yolov3_tensorflow_amd.configs import FLAGS
from yolov3_tensorflow_amd.dataset.file_util import FileUtil
from yolov3_tensorflow_amd.yolov3.trainer import YOLOv3Trainer
from yolov3_tensorflow_amd.yolov3.yolov3_decoder import YOLOv3Decoder
from yolov3_tensorflow_amd.yolov3.yolov3_post_process import YOLOv3PostProcessor

backend.set_learning_phase(FLAGS.mode == 'train')     # reference :21-24
backend.set_epsilon(1e-8)                              # reference :26
np.random.seed(6)                                      # reference :27 (the weight-init seed 800 of :28 is the engine's default)


def train(yolov3_trainer):
    """reference :31-38"""
    logging.info('loading training set: %s', FLAGS.train_label_path)
    # FLAGS.batch_size is the global batch; under torchrun every rank reads its slice of it (FileUtil.host_batches)
    train_dataset = FileUtil.get_dataset(FLAGS.train_label_path, FLAGS.train_set_dir, image_size=FLAGS.input_image_size[0:2],
                                         batch_size=yolov3_trainer.batch_size, is_augment=FLAGS.is_augment, is_test=False,
                                         decode_procs=FLAGS.get('decode_procs'))
    # (the trainer's defaults for these two are bound when its module is imported, as in the reference, trainer.py:99; FLAGS edited after that
    #  import are honoured by passing the current values)
    yolov3_trainer.train(train_dataset, None, train_steps=FLAGS.steps_per_epoch, val_steps=FLAGS.validation_steps)
    logging.info('training finished')


def _detect(yolov3_trainer, yolov3_decoder, images):
    """network + decode + score filter + NMS (reference :60-72) without leaving the GPU: forward kernels -> yolo_decode_head ->
    yolo_filter_boxes -> yolo_nms_heads; only the surviving rows come back.  -> per image [head /8, /16, /32] (k, 9) arrays"""
    import torch
    model = yolov3_trainer.model
    if not torch.is_tensor(images):
        images = torch.as_tensor(np.asarray(images, dtype=np.float32))
    n, N = images.shape[0], model.batch_size
    out = []
    for i in range(0, n, N):
        chunk = images[i:i + N]
        valid = chunk.shape[0]
        if valid < N:
            chunk = torch.cat([chunk, torch.zeros((N - valid,) + tuple(chunk.shape[1:]), dtype=chunk.dtype, device=chunk.device)], dim=0)
        decoded, boxes = yolov3_decoder.decode_device(model.forward_only(chunk, training=False))
        dev_boxes = YOLOv3PostProcessor.filter_boxes_device(decoded, boxes, FLAGS.confidence_thresh)
        YOLOv3PostProcessor.apply_nms_device(dev_boxes, FLAGS.nms_thresh)
        out.extend(YOLOv3PostProcessor.boxes_to_host(dev_boxes)[:valid])
    return out


def test(yolov3_trainer, yolov3_decoder, save_path=None):
    """reference :41-80"""
    test_set = FileUtil.get_dataset(FLAGS.test_label_path, FLAGS.test_set_dir, image_size=FLAGS.input_image_size[0:2],
                                    batch_size=yolov3_trainer.batch_size, is_augment=False, is_test=True)
    input_box_size = np.tile(FLAGS.input_image_size[1::-1], [2])          # [W, H, W, H]
    results = []
    for images, labels, image_paths in test_set:
        detections = _detect(yolov3_trainer, yolov3_decoder, images)
        for n, (image, image_path) in enumerate(zip(images, image_paths)):
            nms_boxes = detections[n]
            in_boxes = YOLOv3PostProcessor.resize_boxes(nms_boxes, target_size=input_box_size)
            results.append((image_path, in_boxes))
            if save_path is not None:
                YOLOv3PostProcessor.visualize(image.cpu().numpy() if hasattr(image, 'cpu') else image, in_boxes, src_box_size=input_box_size,
                                              image_path=os.path.join(save_path, os.path.basename(image_path)))
    return results


def predict(yolov3_trainer, yolov3_decoder, image_paths, save_path):
    """reference :83-120"""
    from yolov3_tensorflow_amd.dataset.file_util import DeviceImagePipeline
    input_box_size = np.tile(FLAGS.input_image_size[1::-1], [2])
    pipe = DeviceImagePipeline(1, FLAGS.input_image_size[0:2])
    for image_path in image_paths:
        image = pipe([FileUtil.read_image(image_path)]).cpu().numpy()[0]       # letterbox, x/255, RGB -> BGR on the GPU
        nms_boxes = _detect(yolov3_trainer, yolov3_decoder, np.expand_dims(image, 0))[0]
        in_boxes = YOLOv3PostProcessor.resize_boxes(nms_boxes, target_size=input_box_size)
        YOLOv3PostProcessor.visualize(image, in_boxes, src_box_size=input_box_size, image_path=os.path.join(save_path, os.path.basename(image_path)))


def run():
    """reference :123-181"""
    if FLAGS.gpu_mode == YOLOv3Trainer.CPU_MODE:
        raise RuntimeError("gpu_mode 'cpu' is not available on the MI355X-native path")
    logging.basicConfig(level=logging.INFO)
    import torch
    # the whole program runs on a high-priority stream: the step's main-stream kernels are the critical path and take CUs before the
    # concurrent weight-gradient / communication streams do (+1 % measured on MI355X)
    torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    yolov3_trainer = YOLOv3Trainer()
    if FLAGS.mode == 'train':
        train(yolov3_trainer)
    elif FLAGS.mode in ('test', 'predict'):
        yolov3_decoder = YOLOv3Decoder(head_grid_sizes=FLAGS.head_grid_sizes, class_num=FLAGS.class_num, anchor_boxes=FLAGS.anchor_boxes)
        save_path = FLAGS.save_path
        if save_path is not None and not os.path.exists(save_path):
            os.makedirs(save_path)                      # the reference raises before it can create the directory (run.py:154-157)
        if FLAGS.mode == 'test':
            test(yolov3_trainer, yolov3_decoder, save_path)
        else:
            root = FLAGS.image_root_path
            if root is None or not os.path.isdir(root) or save_path is None:
                raise ValueError('image_root_path must be a directory and save_path must be set')
            predict(yolov3_trainer, yolov3_decoder, [os.path.join(root, f) for f in sorted(os.listdir(root)) if f.endswith('.jpg')], save_path)
    elif FLAGS.mode == 'save_pb':
        yolov3_trainer.save_mobile()
    elif FLAGS.mode == 'save_serving':
        yolov3_trainer.save_serving()
    else:
        raise ValueError('Mode Error!')


if __name__ == '__main__':
    run()
