"""YOLOv3Trainer with the reference's constructor and methods (yolov3/trainer.py:19-185): build model, (data-parallel wrap),
load the latest checkpoint, choose the optimizer, build the loss, compile, run the epoch loop with the reference's callbacks
(checkpoint every ``ckpt_period`` epochs, early stopping on the epoch loss, per-epoch learning-rate schedule, detailed loss log).
"""
import logging
import os
import time
import numpy as np

from yolov3_tensorflow_amd.configs import FLAGS
from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
from yolov3_tensorflow_amd import model as model_lib


class YOLOv3Trainer(object):
    GPU_MODE = 'gpu'
    CPU_MODE = 'cpu'

    def __init__(self):
        """reference :30-97"""
        self.backbone = FLAGS.model_backbone
        self.input_image_size = FLAGS.input_image_size
        self.head_channel_nums = FLAGS.head_channel_nums
        if FLAGS.gpu_mode == YOLOv3Trainer.CPU_MODE:
            raise RuntimeError("gpu_mode 'cpu' is not available: this is the MI355X-native path (no CPU fallback)")
        # reference :40-43 wraps the model with keras multi_gpu_model, which splits the one batch across the towers; here data parallelism is
        # one process per GPU (torch.distributed / RCCL, parallel.setup_data_parallel when WORLD_SIZE > 1) and FLAGS.batch_size stays the
        # GLOBAL batch: each rank builds its static graph for batch_size / world images
        from yolov3_tensorflow_amd import parallel
        self.batch_size = parallel.per_rank_batch(int(FLAGS.batch_size))
        self.model = YOLOv3Detector(self.backbone).build(self.input_image_size, self.head_channel_nums, FLAGS.head_names,
                                                         batch_size=self.batch_size)
        parallel.setup_data_parallel(self.model)
        self.model.summary()
        self.history = None

        # load pre-trained weights if any (reference :47-67)
        self.checkpoint_path = FLAGS.checkpoint_path
        if self.checkpoint_path is None:
            self.checkpoint_path = 'models/'
        if any(os.path.isfile(self.checkpoint_path + ext) for ext in ('', '.index', '.npz')):
            self.model.load_weights(self.checkpoint_path)
            self._resumed_from = self.checkpoint_path
            logging.info('weights loaded')
            self.checkpoint_path = os.path.dirname(self.checkpoint_path)
        if os.path.isdir(self.checkpoint_path):
            latest = model_lib.latest_checkpoint(self.checkpoint_path)
            if latest is not None:
                self.model.load_weights(latest)
                self._resumed_from = latest
                logging.info('weights loaded: %s', latest)
        else:
            self.checkpoint_path = os.path.dirname(self.checkpoint_path)
        self.checkpoint_path = os.path.join(self.checkpoint_path, FLAGS.checkpoint_name)

        # optimizer (reference :69-75): SGD-Nesterov unless 'adam' / 'radam'; RAdam(lr=1e-3) ignores FLAGS.init_lr exactly like the reference
        # (every epoch the LearningRateScheduler overwrites the rate anyway, :94)
        if FLAGS.optimizer == 'radam':
            from yolov3_tensorflow_amd.utils.radam import RAdam
            optimizer = RAdam(lr=1e-3)
        elif FLAGS.optimizer == 'adam':
            from yolov3_tensorflow_amd.utils.optimizers import Adam
            optimizer = Adam(lr=FLAGS.init_lr, amsgrad=True)
        else:
            from yolov3_tensorflow_amd.utils.optimizers import SGD
            optimizer = SGD(lr=FLAGS.init_lr, momentum=0.95, nesterov=True)
        self.loss_object = YOLOv3Loss(FLAGS.head_grid_sizes, FLAGS.class_num, FLAGS.anchor_boxes, FLAGS.iou_thresh,
                                      FLAGS.loss_weights, rectified_coord_num=FLAGS.rectified_coord_num,
                                      rectified_loss_weight=FLAGS.rectified_loss_weight, is_focal_loss=FLAGS.is_focal_loss,
                                      focal_alpha=FLAGS.focal_alpha, focal_gamma=FLAGS.focal_gamma,
                                      is_tiou_recall=FLAGS.is_tiou_recall)
        self.loss_function = self.loss_object.loss
        self.optimizer = optimizer
        self.model.compile(optimizer=optimizer, loss=self.loss_function)
        # opt-in (not in the reference, which resumes with weights only -- SURVEY.md appendix B): restore the optimizer moments, its step
        # counter, the rectified-loss image counter and the epoch, so that the schedule continues instead of restarting
        self.start_epoch = 0
        if FLAGS.get('full_state_resume') and getattr(self, '_resumed_from', None):
            ep = self.model.load_weights(self._resumed_from, full_state=True)
            if ep is not None:
                self.start_epoch = ep + 1
                logging.info('full training state restored: continuing with epoch %d', self.start_epoch + 1)
        self.epoch = FLAGS.epoch
        self.ckpt_period = FLAGS.ckpt_period
        self.stop_patience, self.stop_min_delta = FLAGS.stop_patience, FLAGS.stop_min_delta
        self.lr_func = FLAGS.lr_func
        # callbacks (reference :89-97): checkpoint / early stopping / LR schedule are the loop below; the two custom ones are objects
        from yolov3_tensorflow_amd.utils.logger_callback import DetailLossLogger
        from yolov3_tensorflow_amd.utils.board_callback import MyTensorBoard
        self.log_callback = DetailLossLogger(verbose=2)
        self.tensorboard = MyTensorBoard(log_dir=FLAGS.tensorboard_dir)

    CHECK_EVERY = 50      # steps between two reads of the losses / device-protocol checks inside an epoch

    def _check_protocols_collectively(self):
        """Model.check_device_protocols on every rank; if any rank failed, all ranks raise together (no rank is left waiting in a collective)"""
        from yolov3_tensorflow_amd import parallel
        failure = None
        try:
            self.model.check_device_protocols()
        except (RuntimeError, FloatingPointError) as e:
            failure = e
        if parallel.agree_any(failure is not None, self.model.device, self.model.process_group):
            raise failure if failure is not None else RuntimeError('another rank reported a device-protocol failure')

    def train(self, train_set, val_set, train_steps=FLAGS.steps_per_epoch, val_steps=FLAGS.validation_steps):
        """reference :99-115.  ``train_set`` yields (images float32 (N,H,W,3) in [0,1] BGR, labels float32 (N, T*5) padded -1)."""
        import torch
        from yolov3_tensorflow_amd import parallel
        it = iter(train_set)
        stopper = parallel.EarlyStopping(self.stop_min_delta, self.stop_patience)
        val_it = iter(val_set) if val_set is not None else None
        history = {'loss': [], 'lr': []}
        is_main = self.model.rank == 0
        callbacks = [self.tensorboard, self.log_callback] if is_main else []
        for cb in callbacks:
            cb.set_model(self.model, self.loss_object)
        self.log_callback.on_train_begin(self.epoch, train_steps)
        for epoch in range(self.start_epoch, self.epoch):
            lr = float(self.lr_func(epoch))                                 # LearningRateScheduler (reference :94)
            self.optimizer.lr = lr
            if is_main:
                self.log_callback.on_epoch_begin(epoch)
            # device scalars, read every CHECK_EVERY steps: one synchronisation per 50 steps keeps the host ahead of the GPU (a read per step
            # costs ~25 % of a 4 ms step) while a device-protocol failure or non-finite gradients stop the run within 50 steps, not at the
            # end of a long epoch.  Every rank runs the same number of steps, so the collective decisions below line up.
            pending, losses = [], []
            for step in range(train_steps):
                images, labels = next(it)
                pending.append(self.model.train_on_batch(images, labels, sync=False))
                if len(pending) >= self.CHECK_EVERY or step == train_steps - 1:
                    losses.extend(torch.stack(pending).double().cpu().numpy().tolist())
                    pending = []
                    self._check_protocols_collectively()
            # keras reports the running mean over the epoch (nan for an epoch of zero steps, as np.mean([]) gave before); data parallel: the
            # mean over the ranks' shards, the same number everywhere
            mean = float(np.mean(losses)) if losses else float('nan')
            epoch_loss = parallel.agree_mean(mean, self.model.device, self.model.process_group)
            if not losses:
                self._check_protocols_collectively()
            history['loss'].append(epoch_loss)
            history['lr'].append(lr)
            # the per-head terms as the callbacks see them at the epoch's end (the last step's values, reference yolov3_loss.py:115-134):
            # rows xy, wh, noobj, obj, class, rectified; columns head /8, /16, /32
            history.setdefault('terms', []).append(self.loss_object.terms.detach().cpu().numpy().copy())
            logs = {'loss': epoch_loss, 'lr': lr}
            if val_it is not None:                                           # fit(validation_data=val_set, validation_steps=val_steps) (reference :107-110)
                val = [self.model.test_on_batch(*next(val_it)[:2]) for _ in range(val_steps)]
                logs['val_loss'] = parallel.agree_mean(float(np.mean(val)), self.model.device, self.model.process_group)
                history.setdefault('val_loss', []).append(logs['val_loss'])
            if is_main:
                self.tensorboard.on_epoch_end(epoch, {k: v for k, v in logs.items() if k != 'lr'})   # MyTensorBoard (reference :96)
                self.log_callback.on_epoch_end(epoch, logs)                  # DetailLossLogger (reference :95)
                if (epoch + 1) % self.ckpt_period == 0:                      # ModelCheckpoint(period) (reference :90-91)
                    path = self.checkpoint_path.format(epoch=epoch + 1, loss=epoch_loss)
                    self.model.save_weights(path, full_state=bool(FLAGS.get('full_state_resume')), epoch=epoch)
                    logging.info('saved %s', path)
            if stopper.should_stop(epoch_loss):                              # EarlyStopping(monitor='loss') (reference :92-93)
                logging.info('early stopping at epoch %d', epoch + 1)
                break
        if is_main:
            self.tensorboard.on_train_end()
        self.history = history
        logging.info('training finished')

    def predict(self, test_images):
        """reference :117-124"""
        return self.model.predict(test_images)

    def convert_multi2single(self):
        """reference :126-138: every rank holds the full weights, so the latest checkpoint is already a single-GPU one;
        kept for interface parity -- writes the ``single_`` copy the reference's test flow expects."""
        dir_name = self.checkpoint_path if os.path.isdir(self.checkpoint_path) else os.path.dirname(self.checkpoint_path)
        latest = model_lib.latest_checkpoint(dir_name)
        if latest is None:
            raise RuntimeError('no checkpoint to convert')
        self.model.load_weights(latest)
        self.model.save_weights(os.path.join(dir_name, 'single_' + os.path.basename(latest)))

    def save_mobile(self):
        raise NotImplementedError('frozen .pb export is TensorFlow-specific (out of scope, SURVEY.md section 2)')

    def save_serving(self):
        raise NotImplementedError('TF-Serving SavedModel export is TensorFlow-specific (out of scope, SURVEY.md section 2)')
