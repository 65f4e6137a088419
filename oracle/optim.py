"""Oracle (test infrastructure): RAdam update and the learning-rate schedule, restated with NumPy.

Follows /root/reference/utils/radam.py:32-107 and /root/reference/configs.py:14-27; instantiation
``RAdam(lr=1e-3)`` at /root/reference/yolov3/trainer.py:75 (beta_1 .9, beta_2 .999, epsilon = K.epsilon() = 1e-8 from
run.py:26, no decay, no AMSGrad, warmup_coef 1).  The reference evaluates the scalar schedule in float32 (K.floatx());
``scalar_dtype`` selects that (default) or float64.

parity unpinned: no golden trajectory exists in the reference; known-answer checks in tests/ are computed by hand.
"""
import numpy as np

TRAIN_STEP_EPOCH = np.array([20, 60, 80, 220, 260, 280, 300])                         # configs.py:16
TRAIN_STEP_LR = np.array([0.01, 1., 0.1, 1., 0.1, 0.01, 0.001]) * 1e-3                # configs.py:17


def lr_func(epoch, step_epoch=TRAIN_STEP_EPOCH, step_lr=TRAIN_STEP_LR):
    """configs.py:23-27"""
    i = 0
    while i < len(step_epoch) and epoch > step_epoch[i]:
        i += 1
    return step_lr[i]


class RAdamOracle(object):
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, warmup_coef=1.,
                 scalar_dtype=np.float32):
        self.iterations = 0
        self.lr, self.beta_1, self.beta_2, self.decay = lr, beta_1, beta_2, decay
        self.epsilon = 1e-8 if epsilon is None else epsilon                           # radam.py:48-50
        self.initial_decay = decay
        self.amsgrad = amsgrad
        self.warmup_coef = warmup_coef
        self.rho_inf = 2. / (1. - beta_2) - 1                                         # radam.py:54
        self.sd = scalar_dtype
        self.m = self.v = self.vhat = None

    def schedule(self):
        """radam.py:60-85 -> (rho_t, lr_t) after incrementing iterations"""
        f = self.sd
        lr = f(self.lr)
        if self.initial_decay > 0:
            lr = lr * (f(1.) / (f(1.) + f(self.decay) * f(self.iterations)))           # radam.py:61-64
        self.iterations += 1                                                          # radam.py:66
        t = f(self.iterations)
        # beta_1 / beta_2 are float32 K.variables in the reference (radam.py:44-45), so their float32 values are what enters
        # the chain, and rho_inf (radam.py:54) is computed from that float32 beta_2
        beta_1, beta_2 = f(np.float32(self.beta_1)), f(np.float32(self.beta_2))
        b1p = np.power(beta_1, t)                                                     # radam.py:77
        b2p = np.power(beta_2, t)                                                     # radam.py:78
        rho_inf = f(2.) / (f(1.) - beta_2) - f(1.)                                    # radam.py:54
        rho_t = rho_inf - f(2.0) * t * b2p / (f(1.0) - b2p)                           # radam.py:79
        if rho_t >= 5.0:                                                              # radam.py:81-85
            lr_t = np.sqrt((rho_t - f(4.)) * (rho_t - f(2.)) * rho_inf /
                           ((rho_inf - f(4.)) * (rho_inf - f(2.)) * rho_t)) * lr * (np.sqrt(f(1.) - b2p) / (f(1.) - b1p))
        else:
            lr_t = f(self.warmup_coef) * lr / (f(1.) - b1p)
        return f(rho_t), f(lr_t)

    def step(self, params, grads):
        """params, grads: lists of float32 ndarrays (params updated in place).  radam.py:87-106"""
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
            self.v = [np.zeros_like(p) for p in params]
            self.vhat = [np.zeros_like(p) for p in params] if self.amsgrad else None
        rho_t, lr_t = self.schedule()
        b1, b2, eps = np.float32(self.beta_1), np.float32(self.beta_2), np.float32(self.epsilon)
        one = np.float32(1.)
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g.astype(np.float32)
            m_t = b1 * self.m[i] + (one - b1) * g                                     # radam.py:88
            v_t = b2 * self.v[i] + (one - b2) * np.square(g)                          # radam.py:89
            if self.amsgrad:
                vh = np.maximum(self.vhat[i], v_t)                                    # radam.py:92
                self.vhat[i] = vh
            else:
                vh = v_t
            if rho_t >= 5.0:
                p -= np.float32(lr_t) * (m_t / (np.sqrt(vh) + eps))                   # radam.py:93/96
            else:
                p -= np.float32(lr_t) * m_t
            self.m[i], self.v[i] = m_t, v_t
        return rho_t, lr_t


class AdamOracle(object):
    """keras.optimizers.Adam as the reference instantiates it at yolov3/trainer.py:72 (``Adam(lr=FLAGS.init_lr, amsgrad=True)``).  The class
    itself lives in tensorflow (>= 1.13.1, python/keras/optimizers.py: Adam.get_updates), not under /root/reference; this restates its
    published update:  lr' = lr / (1 + decay * iterations);  t = iterations + 1;  lr_t = lr' * sqrt(1 - b2^t) / (1 - b1^t);
    m_t = b1 m + (1 - b1) g;  v_t = b2 v + (1 - b2) g^2;  vhat_t = max(vhat, v_t) if amsgrad;  p -= lr_t * m_t / (sqrt(vhat_t or v_t) + eps)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False, scalar_dtype=np.float64):
        self.iterations, self.lr, self.beta_1, self.beta_2, self.decay = 0, lr, beta_1, beta_2, decay
        self.epsilon = 1e-8 if epsilon is None else epsilon
        self.amsgrad, self.sd = amsgrad, scalar_dtype
        self.m = self.v = self.vhat = None

    def step(self, params, grads):
        f = self.sd
        lr = f(self.lr)
        if self.decay > 0:
            lr = lr * (f(1.) / (f(1.) + f(self.decay) * f(self.iterations)))
        self.iterations += 1
        t = f(self.iterations)
        b1s, b2s = f(np.float32(self.beta_1)), f(np.float32(self.beta_2))
        lr_t = np.float32(lr * (np.sqrt(f(1.) - np.power(b2s, t)) / (f(1.) - np.power(b1s, t))))
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
            self.v = [np.zeros_like(p) for p in params]
            self.vhat = [np.zeros_like(p) for p in params]
        b1, b2, eps, one = np.float32(self.beta_1), np.float32(self.beta_2), np.float32(self.epsilon), np.float32(1.)
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g.astype(np.float32)
            self.m[i] = b1 * self.m[i] + (one - b1) * g
            self.v[i] = b2 * self.v[i] + (one - b2) * np.square(g)
            den = self.v[i]
            if self.amsgrad:
                self.vhat[i] = np.maximum(self.vhat[i], self.v[i])
                den = self.vhat[i]
            p -= lr_t * self.m[i] / (np.sqrt(den) + eps)
        return lr_t


class SGDOracle(object):
    """keras.optimizers.SGD as the reference instantiates it at yolov3/trainer.py:70 (``SGD(lr=FLAGS.init_lr, momentum=0.95, nesterov=True)``);
    tensorflow python/keras/optimizers.py SGD.get_updates restated:  lr' = lr / (1 + decay * iterations);  v = momentum * m - lr' * g;
    m <- v;  p += momentum * v - lr' * g  (Nesterov)  or  p += v."""

    def __init__(self, lr=0.01, momentum=0., decay=0., nesterov=False):
        self.iterations, self.lr, self.momentum, self.decay, self.nesterov = 0, lr, momentum, decay, nesterov
        self.m = None

    def step(self, params, grads):
        lr = np.float64(self.lr)
        if self.decay > 0:
            lr = lr * (1. / (1. + np.float64(self.decay) * self.iterations))
        self.iterations += 1
        lr, mom = np.float32(lr), np.float32(self.momentum)
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g.astype(np.float32)
            v = mom * self.m[i] - lr * g
            self.m[i] = v
            p += (mom * v - lr * g) if self.nesterov else v
        return lr
