"""Oracle (test infrastructure): the reference's conv/BN graph restated on PyTorch-CPU.

Follows:
  /root/reference/backbone/basic_backbone.py:20-163   (layer factories)
  /root/reference/backbone/resnet18.py:17-69          (ResNet18)
  /root/reference/backbone/resnet18_v2.py:13-74       (ResNet18_v2)
  /root/reference/backbone/mixnet18.py:18-82          (MixNet18)
  /root/reference/yolov3/yolov3_detector.py:44-151    (3 FPN heads + merge)

Tensors are NHWC at the interface (basic_backbone.py:15-18); kernels are HWIO (Keras Conv2D) and HW C 1 (DepthwiseConv2D).
Parameters live in an ordered dict keyed by the Keras auto-generated variable names
(``conv2d[_k]/kernel``, ``batch_normalization_v1[_k]/{gamma,beta,moving_mean,moving_variance}``,
``depthwise_conv2d[_k]/depthwise_kernel``, ``yolov3_head_{8,16,32}/{kernel,bias}``), created in the
reference's layer-construction order, so weights can be exchanged with the product by name.

parity unpinned: the arithmetic of these layers lives in TensorFlow (requirements.txt:5), which is not installed.
TF semantics restated from documentation: 'same' padding pad_total = max((ceil(H/s)-1)*s + k - H, 0), before =
pad_total // 2 (so (0,1) for k=3, s=2, even H); max-pool pads with -inf; UpSampling2D nearest = pixel repeat;
BatchNormalization (training) normalises with the biased batch variance; the moving statistics are updated with
momentum 0.9, the moving variance from the UNBIASED batch variance (what tf.nn.fused_batch_norm, used by tf.keras for
4-D NHWC inputs, returns for the running average).  Moving statistics never enter the training loss.
An optional ``round_fn`` emulates the product's bf16 storage points (conv outputs, activations) so the GPU path can
be checked tightly; with ``round_fn=None`` this is the plain float32 reference.
"""
import collections
import math
import numpy as np
import torch
import torch.nn.functional as F

L2_CONV_DECAY = 5e-4       # basic_backbone.py:11
BN_L2_GAMMA_DECAY = 1e-5   # basic_backbone.py:12
BN_MOMENTUM = 0.9          # basic_backbone.py:13
BN_EPSILON = 1e-5          # basic_backbone.py:14


def same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def bf16_round(x):
    return x.to(torch.bfloat16).to(x.dtype)


class Params(object):
    """Ordered parameter store with Keras-style auto names."""

    def __init__(self, seed=800, dtype=torch.float32):
        self.p = collections.OrderedDict()
        self.kind = {}
        self.counters = collections.Counter()
        self.gen = torch.Generator().manual_seed(seed)
        self.dtype = dtype

    def layer_name(self, base):
        k = self.counters[base]
        self.counters[base] += 1
        return base if k == 0 else '%s_%d' % (base, k)

    def add(self, name, tensor, kind, trainable=True):
        self.p[name] = tensor.to(self.dtype).requires_grad_(trainable)
        self.kind[name] = kind
        return self.p[name]

    def he_normal(self, shape, fan_in):
        # keras 'he_normal' = truncated normal, stddev = sqrt(2 / fan_in) / .87962566103423978
        std = math.sqrt(2.0 / fan_in) / .87962566103423978
        t = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=self.gen)
        return t

    def trainable(self):
        return [(n, t) for n, t in self.p.items() if t.requires_grad]


class Graph(object):
    """Functional executor.  ``params`` is created lazily on the first pass (build) and reused afterwards."""

    def __init__(self, params=None, training=True, round_fn=None, dtype=torch.float32, inject=None):
        # inject: optional iterator of tensors; at every ACTIVATION storage point the computed value is replaced by the next
        # injected tensor with a straight-through gradient, so autograd differentiates at exactly that forward state
        self.inject = inject
        self.params = params if params is not None else Params(dtype=dtype)
        self.building = params is None
        self.training = training
        self.round_fn = round_fn
        self.replay = None
        self.bn_updates = {}

    def begin(self):
        self.replay = collections.Counter()
        self.bn_updates = {}

    def _name(self, base):
        if self.building:
            return self.params.layer_name(base)
        k = self.replay[base]
        self.replay[base] += 1
        return base if k == 0 else '%s_%d' % (base, k)

    def _r(self, x, kind='act'):
        if kind == 'act' and self.inject is not None:
            g = next(self.inject).to(x.dtype)
            if tuple(g.shape) != tuple(x.shape):
                raise ValueError('injected tensor shape %s != %s' % (tuple(g.shape), tuple(x.shape)))
            return x + (g - x).detach()
        return x if self.round_fn is None else _RoundSTE.apply(x, self.round_fn)

    # ---------------------------------------------------------------- basic_backbone.py:20-43
    def convolution(self, x, filters, kernel_size=(3, 3), strides=(1, 1), padding='same', use_bias=False,
                    name=None, init='he_normal'):
        name = name or self._name('conv2d')
        kh, kw = kernel_size
        cin = x.shape[-1]
        if self.building:
            if init == 'he_normal':
                w = self.params.he_normal((kh, kw, cin, filters), kh * kw * cin)
                kind = 'conv_kernel'             # L2 5e-4 (basic_backbone.py:41)
            else:                                # RandomNormal(stddev=0.01), no regulariser (yolov3_detector.py:98-100)
                w = torch.randn((kh, kw, cin, filters), generator=self.params.gen) * 0.01
                kind = 'head_kernel'
            self.params.add(name + '/kernel', w, kind)
            if use_bias:
                self.params.add(name + '/bias', torch.zeros(filters), 'bias')
        w = self.params.p[name + '/kernel']
        b = self.params.p[name + '/bias'] if use_bias else None
        wq = self._r(w, kind='w')
        xin = x.permute(0, 3, 1, 2)
        H, W = x.shape[1], x.shape[2]
        if padding == 'same':
            (pt, pb), (pl, pr) = same_pad(H, kh, strides[0]), same_pad(W, kw, strides[1])
            xin = F.pad(xin, (pl, pr, pt, pb))
        y = F.conv2d(xin, wq.permute(3, 2, 0, 1), b, stride=strides)
        return y.permute(0, 2, 3, 1)

    # ---------------------------------------------------------------- basic_backbone.py:45-66
    def depthwise_conv(self, x, kernel_size=(3, 3)):
        name = self._name('depthwise_conv2d')
        kh, kw = kernel_size
        c = x.shape[-1]
        if self.building:
            # keras he_normal fan_in for a (kh, kw, C, 1) depthwise kernel = kh*kw*C
            self.params.add(name + '/depthwise_kernel', self.params.he_normal((kh, kw, c, 1), kh * kw * c), 'conv_kernel')
        w = self._r(self.params.p[name + '/depthwise_kernel'], kind='w')
        xin = F.pad(x.permute(0, 3, 1, 2), (kw // 2, kw // 2, kh // 2, kh // 2))
        y = F.conv2d(xin, w.permute(2, 3, 0, 1), None, groups=c)
        return y.permute(0, 2, 3, 1)

    # ---------------------------------------------------------------- basic_backbone.py:68-78
    def batch_normalization(self, x):
        name = self._name('batch_normalization_v1')
        c = x.shape[-1]
        if self.building:
            self.params.add(name + '/gamma', torch.ones(c), 'bn_gamma')   # L2 1e-5 (basic_backbone.py:76)
            self.params.add(name + '/beta', torch.zeros(c), 'bn_beta')
            self.params.add(name + '/moving_mean', torch.zeros(c), 'bn_mean', trainable=False)
            self.params.add(name + '/moving_variance', torch.ones(c), 'bn_var', trainable=False)
        g, b = self.params.p[name + '/gamma'], self.params.p[name + '/beta']
        if self.training:
            m = x.mean(dim=(0, 1, 2))
            v = ((x - m) ** 2).mean(dim=(0, 1, 2))
            cnt = x.shape[0] * x.shape[1] * x.shape[2]
            self.bn_updates[name] = (m.detach(), v.detach() * (cnt / max(cnt - 1, 1)))
        else:
            m, v = self.params.p[name + '/moving_mean'], self.params.p[name + '/moving_variance']
        return (x - m) * torch.rsqrt(v + BN_EPSILON) * g + b

    def apply_bn_updates(self):
        with torch.no_grad():
            for name, (m, v) in self.bn_updates.items():
                mm, mv = self.params.p[name + '/moving_mean'], self.params.p[name + '/moving_variance']
                mm.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * m)
                mv.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * v)

    def activation(self, x):                      # basic_backbone.py:80-90 (ReLU everywhere)
        return self._r(torch.relu(x))

    def conv_bn(self, x, filters, **kw):          # basic_backbone.py:127-138
        return self.batch_normalization(self._r(self.convolution(x, filters, **kw)))

    def depthwise_conv_bn(self, x, **kw):         # basic_backbone.py:140-150
        return self.batch_normalization(self._r(self.depthwise_conv(x, **kw)))

    def bn_activation(self, x):                   # basic_backbone.py:152-163
        return self.activation(self.batch_normalization(x))

    def element_wise_add(self, identity, residual, is_nin=False):    # basic_backbone.py:102-125
        sh = int(round(identity.shape[1] / residual.shape[1]))
        sw = int(round(identity.shape[2] / residual.shape[2]))
        if is_nin:
            identity = self._r(self.convolution(identity, residual.shape[-1], kernel_size=(1, 1), strides=(sh, sw),
                                                padding='valid'))
            identity = self.batch_normalization(identity)
        return identity + residual

    def max_pool(self, x):                        # MaxPooling2D(3, 2, 'same'): resnet18.py:60
        H, W = x.shape[1], x.shape[2]
        (pt, pb), (pl, pr) = same_pad(H, 3, 2), same_pad(W, 3, 2)
        xin = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf'))
        return F.max_pool2d(xin, 3, 2).permute(0, 2, 3, 1)

    def up_sample(self, x):                       # UpSampling2D(2, nearest): yolov3_detector.py:115
        return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


class _RoundSTE(torch.autograd.Function):
    """value rounding with a straight-through gradient (emulates a bf16 storage point).  With ROUND_GRADS the incoming
    gradient is rounded too: the product stores the gradient of every bf16 tensor in bf16 as well."""
    ROUND_GRADS = False

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return fn(x)

    @staticmethod
    def backward(ctx, g):
        return (ctx.fn(g) if _RoundSTE.ROUND_GRADS else g), None


class _GradRound(torch.autograd.Function):
    """identity whose gradient is rounded (the d(logits) handed to the detection convs' backward are bf16)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g), None


# ======================================================================== backbones
def resnet18(g, x):
    """resnet18.py:52-69"""
    def block(x, filters, is_nin, strides=(1, 1)):                      # :17-35
        r = g.conv_bn(x, filters, strides=strides)
        r = g.activation(r)
        r = g.conv_bn(r, filters)                                        # strides reset to (1,1) (:31)
        return g.activation(g.element_wise_add(x, r, is_nin=is_nin))

    def module(x, filters, strides=(1, 1)):                              # :37-50
        return block(block(x, filters, True, strides), filters, False)

    net = g.conv_bn(x, 64, kernel_size=(3, 3), strides=(2, 2))           # :59
    net = g.activation(g.max_pool(net))                                  # :60-61 (pool BEFORE relu)
    net = module(net, 64)
    c3 = module(net, 128, (2, 2))
    c4 = module(c3, 256, (2, 2))
    c5 = module(c4, 512, (2, 2))
    return c3, c4, c5


def resnet18_v2(g, x):
    """resnet18_v2.py:54-74"""
    def block(x, filters, is_nin, strides=(1, 1)):                      # :13-37
        pre = g.bn_activation(x)
        r = g._r(g.convolution(pre, filters, strides=strides))
        r = g.bn_activation(r)
        r = g._r(g.convolution(r, filters))
        if is_nin:
            return g._r(g.element_wise_add(pre, r, is_nin=True))         # shortcut from the PRE-ACTIVATED tensor (:33-34)
        return g._r(g.element_wise_add(x, r, is_nin=False))              # (:36)

    def module(x, filters, strides=(1, 1)):                              # :39-52
        return block(block(x, filters, True, strides), filters, False)

    net = g._r(g.convolution(x, 64, kernel_size=(3, 3), strides=(2, 2)))  # :61 (no BN)
    net = g._r(g.max_pool(net))                                          # :62 (a stored tensor in the product; rounding is exact here)
    net = module(net, 64)
    c3 = module(net, 128, (2, 2))
    c4 = module(c3, 256, (2, 2))
    c5 = module(c4, 512, (2, 2))
    return g.bn_activation(c3), g.bn_activation(c4), g.bn_activation(c5)  # :70-72


MIX_KERNEL_SIZES = [(3, 3), (5, 5), (7, 7), (9, 9)]                      # mixnet18.py:18
MIX_KERNEL_RATIO = np.array([0, 8, 4, 2, 2], dtype=np.float64).cumsum() / 16.0   # mixnet18.py:19-20


def mixnet18(g, x):
    """mixnet18.py:65-82"""
    def block(x, filters, is_nin, strides=(1, 1)):                      # :22-48
        r = g.activation(g.conv_bn(x, filters, strides=strides))
        nums = (filters * MIX_KERNEL_RATIO).astype(np.int64)             # :38-39
        parts = []
        for i, ks in enumerate(MIX_KERNEL_SIZES):                        # :41-44 (intended static split)
            parts.append(g.depthwise_conv_bn(r[..., nums[i]:nums[i + 1]], kernel_size=ks))
        r = torch.cat(parts, dim=-1)                                     # :45
        return g.activation(g.element_wise_add(x, r, is_nin=is_nin))

    def module(x, filters, strides=(1, 1)):                              # :50-63
        return block(block(x, filters, True, strides), filters, False)

    net = g.conv_bn(x, 64, kernel_size=(3, 3), strides=(2, 2))           # :72
    net = g.activation(g.max_pool(net))                                  # :73-74
    net = module(net, 64)
    c3 = module(net, 128, (2, 2))
    c4 = module(c3, 256, (2, 2))
    c5 = module(c4, 512, (2, 2))
    return c3, c4, c5


BACKBONES = {'resnet-18': resnet18, 'resnet-18-v2': resnet18_v2, 'mixnet-18': mixnet18}   # yolov3_detector.py:19-30


# ======================================================================== detector heads
def detection_heads(g, nets, head_channel_nums, head_names):
    """yolov3_detector.py:61-151.  Returns the three raw head tensors (/8, /16, /32), NHWC."""
    c3, c4, c5 = nets
    ch8, ch16, ch32 = [int(c) for c in head_channel_nums]
    n8, n16, n32 = head_names
    # /32 head (:88-101)
    net = g.activation(g.conv_bn(c5, 512))
    h32 = g.convolution(net, ch32, kernel_size=(1, 1), use_bias=True, name=n32, init='normal')
    # /16 head (:103-126) -- branches from the BACKBONE C5 (:75)
    net = g.activation(g.conv_bn(c5, 256))
    merge = torch.cat([g.up_sample(net), c4], dim=-1)
    merge = g.activation(g.conv_bn(merge, 256, kernel_size=(1, 1)))
    net = g.activation(g.conv_bn(merge, 512, kernel_size=(3, 3)))
    h16 = g.convolution(net, ch16, kernel_size=(1, 1), use_bias=True, name=n16, init='normal')
    # /8 head (:128-151)
    net = g.activation(g.conv_bn(merge, 128, kernel_size=(1, 1)))
    m8 = torch.cat([g.up_sample(net), c3], dim=-1)
    m8 = g.activation(g.conv_bn(m8, 128, kernel_size=(1, 1)))
    m8 = g.activation(g.conv_bn(m8, 256, kernel_size=(3, 3)))
    h8 = g.convolution(m8, ch8, kernel_size=(1, 1), use_bias=True, name=n8, init='normal')
    return h8, h16, h32


class DetectorOracle(object):
    """YOLOv3Detector(backbone).build(...) restated (yolov3_detector.py:32-59)."""

    def __init__(self, backbone_name, head_channel_nums, head_names=('yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32'),
                 seed=800, dtype=torch.float32):
        if backbone_name not in BACKBONES:
            raise ValueError('unknown backbone')                          # yolov3_detector.py:39-42
        self.backbone = BACKBONES[backbone_name]
        self.head_channel_nums = head_channel_nums
        self.head_names = head_names
        self.params = None
        self.seed = seed
        self.dtype = dtype

    def forward(self, images, training=True, round_fn=None, inject=None):
        """images (N,H,W,3) in [0,1] BGR -> (h8, h16, h32) raw logits NHWC; keeps the Graph in self.g"""
        if self.params is None:
            g = Graph(None, training, round_fn, self.dtype)
            g.params.gen.manual_seed(self.seed)
        else:
            g = Graph(self.params, training, round_fn, self.dtype, inject=inject)
        g.begin()
        x = images.to(self.dtype)
        if g.inject is not None:
            x = next(g.inject).to(self.dtype)          # the packed (bf16) input image
        elif round_fn is not None:
            x = round_fn(x)
        heads = detection_heads(g, self.backbone(g, x), self.head_channel_nums, self.head_names)
        if round_fn is not None and _RoundSTE.ROUND_GRADS:
            heads = tuple(_GradRound.apply(h, round_fn) for h in heads)
        self.params = g.params
        self.g = g
        return heads

    def l2_regulariser(self):
        """Keras adds l2(lambda) = lambda * sum(w^2) of every regularised variable to the compiled loss
        (basic_backbone.py:41,64,76); not on beta, biases or the three detection convs."""
        tot = 0.0
        for n, t in self.params.p.items():
            k = self.params.kind[n]
            if k == 'conv_kernel':
                tot = tot + L2_CONV_DECAY * (t ** 2).sum()
            elif k == 'bn_gamma':
                tot = tot + BN_L2_GAMMA_DECAY * (t ** 2).sum()
        return tot
