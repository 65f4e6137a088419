"""MyTensorBoard with the reference's behaviour (utils/board_callback.py:12-148): at every epoch end write
  * ``learning_rate`` and the epoch ``loss`` (and any other fit log) to the main writer in ``log_dir``,
  * each of the 18 sub-losses ``head_{8,16,32}_{rectified,xy,wh,noobj_iou,obj_iou,class}_loss`` as a scalar tagged ``loss`` into its own
    sub-directory (so TensorBoard overlays them in one chart, :99-101),
  * the histogram (1000 numpy bins, first edge dropped, :117-144) of all BatchNorm gammas concatenated, tag ``bn_gamma``, in ``bn_gamma/``.
Event files are written by utils/event_file.py (no TensorFlow)."""
import os
import numpy as np
from yolov3_tensorflow_amd.utils.event_file import EventFileWriter

SUB_LOSSES = (('rectified_loss', 'rectified_coord_loss'), ('xy_loss', 'coord_loss_xy'), ('wh_loss', 'coord_loss_wh'),
              ('noobj_iou_loss', 'noobj_iou_loss'), ('obj_iou_loss', 'obj_iou_loss'), ('class_loss', 'class_loss'))


class MyTensorBoard(object):

    def __init__(self, log_dir='./log', write_graph=True):
        self.log_dir, self.write_graph = log_dir, write_graph       # there is no TF graph to dump; the flag is accepted and ignored
        self.writer = dict()
        self.model = self.loss_object = None
        self.metrics_keys = set('%s_%s' % (head, name) for head in ('head_8', 'head_16', 'head_32') for name, _ in SUB_LOSSES)
        self.histograms_keys = {'bn_gamma'}

    def set_model(self, model, loss_object):
        """reference :33-49"""
        self.model, self.loss_object = model, loss_object
        self.writer['main'] = EventFileWriter(self.log_dir)
        for key in sorted(self.metrics_keys | self.histograms_keys):
            self.writer[key] = EventFileWriter(os.path.join(self.log_dir, key))

    def on_epoch_end(self, epoch, logs=None):
        """reference :84-104"""
        logs = dict(logs or {})
        logs['learning_rate'] = float(self.model.optimizer.lr)
        for name, attr in SUB_LOSSES:
            values = np.asarray(getattr(self.loss_object, attr), dtype=np.float64)
            for i, head in enumerate(('head_8', 'head_16', 'head_32')):
                logs['%s_%s' % (head, name)] = float(values[i])
        logs['bn_gamma'] = self.model.bn_gammas()
        for name, value in logs.items():
            if name in ('batch', 'size'):
                continue
            if name in self.metrics_keys:
                self.writer[name].add_scalar('loss', value, epoch)
            elif name in self.histograms_keys:
                self._log_histogram(self.writer[name], name, value, epoch)
            else:
                self.writer['main'].add_scalar(name, value, epoch)
        for w in self.writer.values():
            w.flush()

    @staticmethod
    def _log_histogram(writer, tag, values, step, bins=1000):
        """reference :112-144"""
        values = np.array(values)
        counts, bin_edges = np.histogram(values, bins=bins)
        writer.add_histogram(tag, float(np.min(values)), float(np.max(values)), int(np.prod(values.shape)), float(np.sum(values)),
                             float(np.sum(values ** 2)), bin_edges[1:], counts, step)

    def on_train_end(self, logs=None):
        for w in self.writer.values():
            w.close()
