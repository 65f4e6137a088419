"""RAdam with the reference's constructor (utils/radam.py:32-54); the update itself (radam.py:56-107) plus the Keras L2
regularisers run as ONE fused kernel over the flat parameter buffer (yolo_radam_l2_step)."""
import torch
from yolov3_tensorflow_amd import ops, backend


class RAdam(object):
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, warmup_coef=1., **kwargs):
        allowed = {'clipnorm', 'clipvalue', 'name'}
        for k in kwargs:
            if k not in allowed:
                raise TypeError('Unexpected keyword argument passed to optimizer: ' + str(k))
        if 'clipnorm' in kwargs or 'clipvalue' in kwargs:
            raise NotImplementedError('gradient clipping is not used by the reference trainer')
        self._lr = float(lr)
        self.beta_1, self.beta_2, self.decay = float(beta_1), float(beta_2), float(decay)
        if epsilon is None:
            epsilon = backend.epsilon()                       # reference :48-49
        self.epsilon = float(epsilon)
        self.initial_decay = float(decay)
        self.amsgrad = bool(amsgrad)
        self.warmup_coef = float(warmup_coef)
        self.rho_inf = 2. / (1. - self.beta_2) - 1            # reference :54
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
            self.l2_partial = torch.zeros(ops.radam_l2_blocks(ps.n), device=dev)
            self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)      # waves that met an inf / NaN gradient element

    def launch(self, model):
        ps = model.g.ps
        ops.radam_schedule(self.sched, self._iterations, self.beta_1, self.beta_2, self.initial_decay, self.warmup_coef)
        ops.radam_l2_step(ps.flat, ps.grad, ps.m, ps.v, ps.l2_table, ps.n, self.sched, self.beta_1, self.beta_2, self.epsilon,
                          grad_scale=1.0 / (model.world_size * backend.loss_scale()), zero_grad=True, params_bf16=ps.bf16, vhat=self.vhat,
                          l2_partial=self.l2_partial, nonfinite=self.nonfinite)
        # reported loss = YOLOv3 loss + sum of L2 regularisers (what keras' compiled loss contains)
        ops.sum_partials(self.l2_partial, self.l2_partial.numel(), None, model.l2_value)
        ops.sum_partials(self.l2_partial, self.l2_partial.numel(), model.loss_obj.total, model.loss_value)

    def get_config(self):
        """reference :109-119"""
        return {'lr': float(self._lr), 'beta_1': float(self.beta_1), 'beta_2': float(self.beta_2), 'decay': float(self.decay),
                'epsilon': self.epsilon, 'amsgrad': self.amsgrad}
