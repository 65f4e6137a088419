"""Module-global FLAGS with the reference's keys, defaults and derived fields (configs.py:11-106) and ``lr_func`` (:23-27).
``FLAGS`` is an attribute-style dict (the reference uses easydict.EasyDict) filled from the DEFAULTS table below; call
``refresh_derived()`` after editing keys that other fields are derived from (the reference computes them once at import time)."""
import datetime
import numpy as np
from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector


class AttrDict(dict):
    """dict with attribute access (FLAGS.key)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


_MILLI = 1e-3
SCHEDULES = {   # epoch boundaries -> learning rate (reference :14-17): a short sweep for checking, and the training schedule
    'check': (np.array([2, 4, 6, 8, 10, 12, 14], np.int64), np.array([0.00001, 0.0001, 0.001, 0.01, 0.1, 1., 10.0], dtype=np.float64) * _MILLI),
    'train': (np.array([20, 60, 80, 220, 260, 280, 300], np.int64), np.array([0.01, 1., 0.1, 1., 0.1, 0.01, 0.001], dtype=np.float64) * _MILLI),
}

DEFAULTS = dict(
    train_set_dir='dataset/test_sample/images',
    train_label_path='dataset/test_sample/label.txt',
    test_set_dir='dataset/test_sample/images',
    test_label_path='dataset/test_sample/label.txt',
    input_image_size=np.array([384, 480, 3], dtype=np.int64),
    anchor_boxes=[[(0.06618181818181816, 0.1025177510694752), (0.18544278606965178, 0.13160367921287464),
                      (0.13, 0.32733333333333337)],
                      [(0.13, 0.32733333333333337), (0.303806787732042, 0.34370030784316496)],
                      [(0.303806787732042, 0.34370030784316496), (0.4667050847457627, 0.5281262429095761),
                      (0.7906945888923907, 0.7888860433597275)]],
    class_num=0,
    head_names=['yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32', ],
    iou_thresh=0.8,
    loss_weights=[(5, 5, 0.05, 3, 1), (8, 8, 0.05, 2, 1), (10, 10, 0.05, 2, 1)],
    train_set_size=20,
    val_set_size=20,
    batch_size=3,
    rectified_coord_num=1464,
    rectified_loss_weight=[1.0, 1.0, 1.0],
    epoch=300,
    init_lr=0.0002,
    mode='train',
    model_backbone=YOLOv3Detector.BACKBONE_RESNET_18,
    optimizer='radam',
    is_augment=True,
    is_label_smoothing=False,
    is_focal_loss=False,
    focal_alpha=1.0,
    focal_gamma=2.0,
    is_gradient_harmonized=False,
    is_tiou_recall=False,
    ckpt_period=50,
    stop_patience=500,
    stop_min_delta=0.0001,
    root_path='',
    confidence_thresh=0.8,
    nms_thresh=0.4,
    save_path='dataset/test_result/',
    image_root_path=None,
    gpu_mode='gpu',
    gpu_num=1,
    visible_gpu='0',
    full_state_resume=False,
    # (not in the reference) JPEG decode of the training set on this many worker PROCESSES that write into shared page-locked staging memory
    # (dataset/decode_worker.py; the reference's tf.data pipeline decodes with AUTOTUNE parallelism, dataset/file_util.py:80-88).
    # 0 = decode on Python threads (~2000-3000 images/s: the interpreter parts of PIL serialise)
    decode_procs=12,
)

FLAGS = AttrDict()
for _name, (_epochs, _rates) in SCHEDULES.items():
    FLAGS[_name + '_step_epoch'], FLAGS[_name + '_step_lr'] = _epochs, _rates
FLAGS.step_epoch, FLAGS.step_lr = FLAGS.train_step_epoch, FLAGS.train_step_lr


def lr_func(epoch):
    """piecewise-constant schedule: the rate of the first boundary that ``epoch`` does not exceed (reference :23-27)"""
    passed = int(np.searchsorted(np.asarray(FLAGS.step_epoch), epoch, side='left'))     # number of boundaries with boundary < epoch
    return FLAGS.step_lr[passed]


FLAGS.update(DEFAULTS)
FLAGS.lr_func = lr_func


def refresh_derived():
    """the fields the reference derives at import time (:43-48, :73-96)"""
    F = FLAGS
    F.box_num = np.array([len(a) for a in F.anchor_boxes], dtype=np.int64)
    F.box_len = 4 + 1 + F.class_num
    F.head_channel_nums = F.box_num * F.box_len
    F.head_grid_sizes = [np.divide(F.input_image_size[0:2], 8).astype(np.int64),
                         np.divide(F.input_image_size[0:2], 16).astype(np.int64),
                         np.divide(F.input_image_size[0:2], 32).astype(np.int64)]  # [H, W]
    F.type = F.model_backbone + '-' + F.optimizer
    F.type += ('-aug' if F.is_augment else '')
    F.type += ('-smooth' if F.is_label_smoothing else '')
    F.type += ('-focal' if F.is_focal_loss else '')
    F.type += ('-ghm' if F.is_gradient_harmonized else '')
    F.type += ('-TIOU' if F.is_tiou_recall else '')
    F.log_path = 'logs/log-{}.txt'.format(F.type)
    F.steps_per_epoch = int(np.ceil(F.train_set_size / F.batch_size))
    F.validation_steps = int(np.ceil(F.val_set_size / F.batch_size))
    F.tensorboard_dir = F.root_path + 'logs/' + 'lpr-{}-{}'.format(F.type, datetime.datetime.now().strftime('%Y%m%d-%H%M%S'))
    F.checkpoint_path = F.root_path + 'models/{}/'.format(F.type)
    F.checkpoint_name = 'lp-recognition-{}'.format(F.type) + '-{epoch: 3d}-{loss: .5f}.ckpt'      # reference :94 (spaces kept)
    F.serving_model_dir = F.root_path + 'models/serving'
    F.pb_model_dir = F.root_path + 'models/pb'


refresh_derived()
