"""DetailLossLogger with the reference's message format (utils/logger_callback.py:12-140): per epoch (verbose 2) or per batch (verbose 1)

 - 12s - lr: 0.0010 - loss: 234.3500 - gamma_regular_loss(20): 1.2345e-04 - kernel_regular_loss(30): 4.6500 -  :
 - head: /8:
 - rectified_loss: ... - xy_loss: ... - wh_loss: ... - noobj_iou_loss: ... - obj_iou_loss: ... - cls_loss: ... -  :
 ...
The six (3,) sub-loss vectors are the loss kernel's ``terms`` output; the two regulariser sums (and the number of terms in each) come
from the flat parameter buffer (Model.regularization_losses)."""
import logging
import time


class DetailLossLogger(object):

    def __init__(self, verbose=0):
        self.verbose = verbose
        self.epochs = self.seen = self.target = 0
        self._last_update = 0
        self.model = self.loss_object = None
        self.metrics = ['loss']

    def set_model(self, model, loss_object):
        self.model, self.loss_object = model, loss_object

    def on_train_begin(self, epochs, steps):
        self.epochs, self.target = epochs, steps

    def on_batch_begin(self, batch, logs=None):
        if self.verbose == 1:
            self._last_update = time.time()

    def on_batch_end(self, batch, logs=None):
        self.seen += (logs or {}).get('num_steps', 1)
        if self.verbose == 1 and self.seen < self.target:
            self.log(logs or {})

    def on_epoch_begin(self, epoch, logs=None):
        self.seen = 0
        if self.verbose in (1, 2) and self.epochs > 1:
            logging.info('Epoch %d/%d' % (epoch + 1, self.epochs))
        self._last_update = time.time()

    def on_epoch_end(self, epoch, logs=None):
        if self.verbose in (1, 2):
            self.log(logs or {})

    HEADS = ('/8:', '/16:', '/32:')
    SUB_LOSSES = (('rectified_loss', 'rectified_coord_loss'), ('xy_loss', 'coord_loss_xy'), ('wh_loss', 'coord_loss_wh'),
                  ('noobj_iou_loss', 'noobj_iou_loss'), ('obj_iou_loss', 'obj_iou_loss'), ('cls_loss', 'class_loss'))

    @staticmethod
    def _field(key, value):
        """' - key: value' with the reference's number format: 4 decimals above 1e-3, scientific below (reference :132-138)"""
        if isinstance(value, str):
            return ' - %s: %s' % (key, value)
        return (' - %s: %.4f' if value > 1e-3 else ' - %s: %.4e') % (key, value)

    def format(self, logs):
        """the reference's message (:87-139): a summary row (seconds, lr, metrics, the two regulariser sums with their term counts), then
        per head a title row and a row of the six sub-losses; every row ends with the reference's ' -  : ' separator + newline"""
        gamma_sum, gamma_n, kernel_sum, kernel_n = self.model.regularization_losses()
        summary = [('lr', logs['lr'])] + [(k, logs[k]) for k in self.metrics if k in logs]
        summary += [('gamma_regular_loss(%d)' % gamma_n, gamma_sum), ('kernel_regular_loss(%d)' % kernel_n, kernel_sum)]
        end_of_row = self._field(' ', '\n')
        text = '\n - %.0fs' % (time.time() - self._last_update) + ''.join(self._field(k, v) for k, v in summary) + end_of_row
        per_head = [getattr(self.loss_object, attr) for _, attr in self.SUB_LOSSES]
        for i, head in enumerate(self.HEADS):
            text += self._field('head', head + '\n')
            text += ''.join(self._field(label, values[i]) for (label, _), values in zip(self.SUB_LOSSES, per_head)) + end_of_row
        return text

    def log(self, logs=None):
        logging.info(self.format(logs or {}))
