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

    def format(self, logs):
        """reference :87-139 -> the message string"""
        info = '\n - %.0fs' % (time.time() - self._last_update)
        log_values = [('lr', logs['lr'])]
        for k in self.metrics:
            if k in logs:
                log_values.append((k, logs[k]))
        gamma_sum, gamma_n, kernel_sum, kernel_n = self.model.regularization_losses()
        log_values.append(('gamma_regular_loss({})'.format(gamma_n), gamma_sum))
        log_values.append(('kernel_regular_loss({})'.format(kernel_n), kernel_sum))
        log_values.append((' ', '\n'))
        lo = self.loss_object
        rect, xy, wh, noobj, obj, cls = (lo.rectified_coord_loss, lo.coord_loss_xy, lo.coord_loss_wh, lo.noobj_iou_loss, lo.obj_iou_loss,
                                         lo.class_loss)
        for i, head in enumerate(('/8:\n', '/16:\n', '/32:\n')):
            log_values.append(('head', head))
            log_values.append(('rectified_loss', rect[i]))
            log_values.append(('xy_loss', xy[i]))
            log_values.append(('wh_loss', wh[i]))
            log_values.append(('noobj_iou_loss', noobj[i]))
            log_values.append(('obj_iou_loss', obj[i]))
            log_values.append(('cls_loss', cls[i]))
            log_values.append((' ', '\n'))
        for key, value in log_values:
            if isinstance(value, str):
                info += ' - %s: %s' % (key, value)
            elif value > 1e-3:
                info += ' - %s: %.4f' % (key, value)
            else:
                info += ' - %s: %.4e' % (key, value)
        return info

    def log(self, logs=None):
        logging.info(self.format(logs or {}))
