"""RAdam with the reference's constructor (utils/radam.py:32-54); the update itself (radam.py:56-107) plus the Keras L2
regularisers run as ONE fused kernel over the flat parameter buffer (yolo_radam_l2_step)."""
from yolov3_tensorflow_amd import ops, backend
from yolov3_tensorflow_amd.utils.optimizers import FlatOptimizer


class RAdam(FlatOptimizer):
    KIND = 0

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, warmup_coef=1., **kwargs):
        super(RAdam, self).__init__(lr, **kwargs)
        self.beta_1, self.beta_2, self.decay = float(beta_1), float(beta_2), float(decay)
        if epsilon is None:
            epsilon = backend.epsilon()                       # reference :48-49
        self.epsilon = float(epsilon)
        self.initial_decay = float(decay)
        self.amsgrad = bool(amsgrad)
        self.warmup_coef = float(warmup_coef)
        self.rho_inf = 2. / (1. - self.beta_2) - 1            # reference :54

    def _coefficients(self):
        return self.beta_1, self.beta_2, self.epsilon

    def launch_schedule(self):
        ops.radam_schedule(self.sched, self._iterations, self.beta_1, self.beta_2, self.initial_decay, self.warmup_coef)

    def get_config(self):
        """reference :109-119"""
        return {'lr': float(self._lr), 'beta_1': float(self.beta_1), 'beta_2': float(self.beta_2), 'decay': float(self.decay),
                'epsilon': self.epsilon, 'amsgrad': self.amsgrad}
