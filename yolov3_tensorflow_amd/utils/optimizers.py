"""The two other optimizers the reference trainer can select (yolov3/trainer.py:70-73): ``keras.optimizers.SGD(lr, momentum=0.95,
nesterov=True)`` and ``keras.optimizers.Adam(lr, amsgrad=True)``, with tf.keras' constructor arguments (tensorflow 1.13.1,
python/keras/optimizers.py -- a dependency of the reference, not part of it).  Like RAdam they run as ONE fused launch over the flat
parameter buffer (yolo_radam_l2_step; the update rule is selected on the device by yolo_optimizer_schedule) together with the Keras L2
regularisers."""
import torch
from yolov3_tensorflow_amd import ops, backend


class FlatOptimizer(object):
    """device state + launch sequence shared by RAdam / Adam / SGD: schedule scalars, step counter, (optional) vhat, L2 partial sums"""
    KIND = None           # yolo_optimizer_schedule kind
    amsgrad = False

    def __init__(self, lr, **kwargs):
        allowed = {'clipnorm', 'clipvalue', 'name'}
        for k in kwargs:
            if k not in allowed:
                raise TypeError('Unexpected keyword argument passed to optimizer: ' + str(k))
        if 'clipnorm' in kwargs or 'clipvalue' in kwargs:
            raise NotImplementedError('gradient clipping is not used by the reference trainer')
        self._lr = float(lr)
        self.model = None

    # the LearningRateScheduler callback assigns optimizer.lr every epoch (reference trainer.py:94)
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if self.model is not None:
            self.sched[0:1].fill_(self._lr)

    @property
    def iterations(self):
        return int(self._iterations.item()) if self.model is not None else 0

    def bind(self, model):
        self.model = model
        dev = model.device
        ps = model.g.ps
        with torch.cuda.device(dev):
            self.sched = torch.tensor([self._lr, 0.0, 0.0, 0.0], device=dev)
            self._iterations = torch.zeros(1, dtype=torch.int64, device=dev)
            self.vhat = torch.zeros(ps.n, device=dev) if self.amsgrad else None
            self.l2_partial = torch.zeros(4 * 2048 + ops.radam_l2_blocks(ps.n), device=dev)     # a region per range launched in one step
            self._cursor = 0
            self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)      # waves that met an inf / NaN gradient element

    # betas / epsilon handed to the update kernel (SGD: beta_1 carries the momentum)
    def _coefficients(self):
        raise NotImplementedError

    def launch_schedule(self):
        b1, b2, _ = self._coefficients()
        ops.optimizer_schedule(self.sched, self._iterations, self.KIND, b1, b2, self.initial_decay)

    def launch_range(self, model, lo, hi, first):
        """the update of the parameter range [lo, hi) (a gradient bucket: slots are 256-element aligned) on the current stream; ``first``
        marks the first range of a step (advances the step counter / schedule scalars once)"""
        ps = model.g.ps
        if first:
            self.launch_schedule()
            self._cursor = 0
        b1, b2, eps = self._coefficients()
        n = hi - lo
        blocks = ops.radam_l2_blocks(n)
        part = self.l2_partial[self._cursor:self._cursor + blocks]
        if part.numel() < blocks:
            raise RuntimeError('l2_partial exhausted: too many optimizer ranges in one step')
        self._cursor += blocks
        sl = slice(lo, hi)
        ops.radam_l2_step(ps.flat[sl], ps.grad[sl], ps.m[sl], ps.v[sl], ps.l2_table[lo // 256:hi // 256], n, self.sched, b1, b2, eps,
                          grad_scale=1.0 / (model.world_size * backend.loss_scale()), zero_grad=True, params_bf16=ps.bf16[sl],
                          vhat=None if self.vhat is None else self.vhat[sl], l2_partial=part, nonfinite=self.nonfinite)

    def finish(self, model):
        """reported loss = YOLOv3 loss + sum of L2 regularisers (what keras' compiled loss contains), from the ranges launched this step"""
        ops.sum_partials(self.l2_partial, self._cursor, model.loss_obj.total, model.loss_value, out_plain=model.l2_value)

    def launch(self, model):
        """the whole update in one launch (hipGraph replay, tests); the training step launches it per gradient bucket (launch_range)"""
        self.launch_range(model, 0, model.g.ps.n, True)
        self.finish(model)


class SGD(FlatOptimizer):
    """keras.optimizers.SGD(lr=0.01, momentum=0., decay=0., nesterov=False)"""

    def __init__(self, lr=0.01, momentum=0., decay=0., nesterov=False, **kwargs):
        super(SGD, self).__init__(lr, **kwargs)
        self.momentum, self.decay, self.nesterov = float(momentum), float(decay), bool(nesterov)
        self.initial_decay = float(decay)
        self.KIND = 2 if self.nesterov else 3

    def _coefficients(self):
        return self.momentum, 0.0, 0.0

    def get_config(self):
        return {'lr': float(self._lr), 'momentum': float(self.momentum), 'decay': float(self.decay), 'nesterov': self.nesterov}


class Adam(FlatOptimizer):
    """keras.optimizers.Adam(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False)"""
    KIND = 1

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, **kwargs):
        super(Adam, self).__init__(lr, **kwargs)
        self.beta_1, self.beta_2, self.decay = float(beta_1), float(beta_2), float(decay)
        self.epsilon = float(backend.epsilon() if epsilon is None else epsilon)
        self.initial_decay = float(decay)
        self.amsgrad = bool(amsgrad)

    def _coefficients(self):
        return self.beta_1, self.beta_2, self.epsilon

    def get_config(self):
        return {'lr': float(self._lr), 'beta_1': float(self.beta_1), 'beta_2': float(self.beta_2), 'decay': float(self.decay),
                'epsilon': self.epsilon, 'amsgrad': self.amsgrad}
