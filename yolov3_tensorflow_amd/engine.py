"""Graph builder + executor of the MI355X-native training step.

The reference builds a tf.keras graph from layer factories (backbone/basic_backbone.py) and lets TensorFlow run and
differentiate it.  Here the same factory calls build a *lazy* graph of symbolic values; consecutive
conv -> BN -> (+shortcut) -> ReLU / max-pool patterns are lowered onto the fused HIP kernels of libyolov3_amd.so, and the
backward pass is a static tape built once.  All device memory (activations, gradients, statistics, the flat parameter /
gradient / RAdam-state buffers) is allocated up front with torch (allocator only), so a whole step is a fixed launch
sequence that can be captured into a hipGraph.

Layout: activations NHWC bf16; conv kernels OHWI (bf16 compute copy + float32 master, both inside flat buffers whose
slots are padded to 256 elements); detection-conv outputs padded to 64*2^k channels, the RGB stem padded 3 -> 8 channels.
"""
import collections
import math
import os
import numpy as np
import torch
from . import ops, backend

L2_CONV_DECAY = 5e-4       # reference backbone/basic_backbone.py:11
BN_L2_GAMMA_DECAY = 1e-5   # :12
BN_MOMENTUM = 0.9          # :13
BN_EPSILON = 1e-5          # :14
SLOT = 256                 # flat-buffer slot granularity (elements)


def _round_up(n, m):
    return (n + m - 1) // m * m


class Param(object):
    """one variable of the model: a slot of the flat buffers + its Keras name / TF shape for checkpoints"""

    def __init__(self, name, kind, tf_shape, dev_shape, l2):
        self.name, self.kind, self.tf_shape, self.dev_shape, self.l2 = name, kind, tuple(tf_shape), tuple(dev_shape), l2
        self.numel = int(np.prod(dev_shape))
        self.offset = None
        self.init = None          # float32 CPU tensor in device layout


class ParamStore(object):
    """flat float32 params / grads / m / v and the bf16 compute copy; Keras-style auto names"""

    def __init__(self, seed=800):
        self.params = collections.OrderedDict()
        self.state = collections.OrderedDict()     # non-trainable (moving statistics): name -> (tf_shape, init tensor)
        self.counters = collections.Counter()
        self.gen = torch.Generator().manual_seed(seed)
        self.total = 0
        self.flat = None

    def layer_name(self, base):
        k = self.counters[base]
        self.counters[base] += 1
        return base if k == 0 else '%s_%d' % (base, k)

    def add(self, p, init):
        p.offset = self.total
        p.init = init.reshape(-1).float()
        self.total += _round_up(p.numel, SLOT)
        self.params[p.name] = p
        return p

    def he_normal(self, shape, fan_in):
        """keras 'he_normal': truncated normal (+-2 sigma), stddev = sqrt(2/fan_in)/.87962566103423978"""
        std = math.sqrt(2.0 / fan_in) / .87962566103423978
        t = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=self.gen)
        return t

    def allocate(self, device):
        n = max(self.total, SLOT)
        self.n = n
        host = torch.zeros(n)
        l2 = torch.zeros(n // SLOT)
        for p in self.params.values():
            host[p.offset:p.offset + p.numel] = p.init
            l2[p.offset // SLOT:(p.offset + _round_up(p.numel, SLOT)) // SLOT] = p.l2
        self.flat = host.to(device)
        self.grad = torch.zeros(n, device=device)
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.bf16 = torch.zeros(n, dtype=backend.torch_dtype(), device=device)
        self.l2_table = l2.to(device)
        ops.cast_f32_to_bf16(self.flat, self.bf16, n)

    def view(self, p, buf=None):
        buf = self.flat if buf is None else buf
        return buf[p.offset:p.offset + p.numel]


class BNState(object):
    """one keras BatchNormalization: gamma/beta slots of the flat buffers + per-channel work vectors"""

    def __init__(self, g, name, C):
        self.name, self.C, self.g = name, C, g
        ps = g.ps
        self.gamma = ps.add(Param(name + '/gamma', 'bn_gamma', (C,), (C,), BN_L2_GAMMA_DECAY), torch.ones(C))
        self.beta = ps.add(Param(name + '/beta', 'bn_beta', (C,), (C,), 0.0), torch.zeros(C))
        ps.state[name + '/moving_mean'] = self
        ps.state[name + '/moving_variance'] = self
        self.parts = [(self, 0)]

    def allocate(self, dev, ps, work=None, c0=0):
        C = self.C
        self.moving_mean = torch.zeros(C, device=dev)
        self.moving_var = torch.ones(C, device=dev)
        if work is None:
            work = [torch.zeros(C, device=dev) for _ in range(6)]
        (self.scale, self.shift, self.mean, self.rstd, self.k1, self.k2) = [w[c0:c0 + C] for w in work]
        self.v_gamma, self.v_beta = ps.view(self.gamma), ps.view(self.beta)
        self.v_dgamma, self.v_dbeta = ps.view(self.gamma, ps.grad), ps.view(self.beta, ps.grad)

    # psum / psq: flat float32 tensors whose element [p * row_stride + c] is the partial of row p, channel c
    def fwd_finalize(self, psum, psq, P, row_stride, count, training):
        for bn, c0 in self.parts:
            if training:
                ops.bn_finalize(psum[c0:], psq[c0:], P, row_stride, bn.C, count, bn.v_gamma, bn.v_beta, BN_EPSILON, bn.g.bn_momentum,
                                bn.moving_mean, bn.moving_var, bn.scale, bn.shift, bn.mean, bn.rstd)
            else:           # keras learning_phase False (run.py:21-24): normalise with the moving statistics
                ops.bn_eval_scale_shift(bn.v_gamma, bn.v_beta, bn.moving_mean, bn.moving_var, BN_EPSILON, bn.scale, bn.shift, bn.C)

    # partial: flat float32 view of [P][3][Cfull]
    def bwd_finalize(self, partial, P, Cfull, which, count):
        for bn, c0 in self.parts:
            ops.bn_bwd_finalize(partial[c0:], P, bn.C, which, count, bn.v_dgamma, bn.v_dbeta, bn.k1, bn.k2, row_stride=3 * Cfull,
                                q_stride=Cfull)


class MultiBN(object):
    """the 4 BatchNorms that follow the 4 depthwise convs of a MixNet block (mixnet18.py:43): per-channel statistics of the
    concatenated tensor are the same numbers, so they share [C] work vectors and one apply pass; only gamma/beta are separate"""

    def __init__(self, bns, split):
        self.C = split[-1]
        self.parts = [(bn, c0) for bn, c0 in zip(bns, split[:-1]) if bn.C > 0]
        self.name = bns[0].name

    def allocate(self, dev, ps):
        work = [torch.zeros(self.C, device=dev) for _ in range(6)]
        (self.scale, self.shift, self.mean, self.rstd, self.k1, self.k2) = work
        for bn, c0 in self.parts:
            bn.allocate(dev, ps, work, c0)

    def _bounds(self):
        return [c0 for _, c0 in self.parts] + [self.parts[-1][1] + self.parts[-1][0].C]

    def fwd_finalize(self, psum, psq, P, row_stride, count, training):
        if not training or self.parts[0][1] != 0 or self._bounds()[-1] != self.C:
            return BNState.fwd_finalize(self, psum, psq, P, row_stride, count, training)
        bns = [bn for bn, _ in self.parts]                                         # one launch for the (up to 4) groups
        ops.bn_finalize_grouped(psum, psq, P, row_stride, self.C, count, self._bounds(), [b.v_gamma for b in bns], [b.v_beta for b in bns],
                                BN_EPSILON, bns[0].g.bn_momentum, [b.moving_mean for b in bns], [b.moving_var for b in bns], self.scale, self.shift,
                                self.mean, self.rstd)

    def bwd_finalize(self, partial, P, Cfull, which, count):
        if self.parts[0][1] != 0 or self._bounds()[-1] != self.C:
            return BNState.bwd_finalize(self, partial, P, Cfull, which, count)
        bns = [bn for bn, _ in self.parts]
        ops.bn_bwd_finalize_grouped(partial, P, self.C, which, count, self._bounds(), [b.v_dgamma for b in bns], [b.v_dbeta for b in bns],
                                    self.k1, self.k2, row_stride=3 * Cfull, q_stride=Cfull)


class Val(object):
    """symbolic value.  kind: input | act | conv | bn | sum | pool | up | cat"""

    def __init__(self, g, kind, shape, **kw):
        self.g, self.kind, self.shape = g, kind, tuple(shape)    # shape = (N, H, W, C)
        self.__dict__.update(kw)
        self.buf = None
        self.grad = None
        self.grad_init = False
        self.grad_writers = []          # backward ops that write / accumulate into .grad, in backward order (plan_backward)
        self.needs_grad = True
        self.cached = None

    @property
    def M(self):
        return self.shape[0] * self.shape[1] * self.shape[2]


class Graph(object):
    def __init__(self, batch, device, seed=800):
        self.N = batch
        self.dev = device
        self.ps = ParamStore(seed)
        self.tape = []
        self._repack_event = None
        self.fused_bn_bwd = True         # single-launch BatchNorm backward where the tensor fits (ops.bn_act_bwd_fused)
        # BN+ReLU units leave a byte mask of the activation's sign for the backward pass (decided before finalize; YOLO_RELU_MASK=0 for A/B runs)
        self.relu_mask = os.environ.get('YOLO_RELU_MASK', '1') != '0'
        # BatchNorm-backward reduce in the epilogue of the data-gradient convolution that completes the unit's output gradient
        # (ApplyOp.plan_fusion; YOLO_DGRAD_BN=0 for A/B runs): no reduce pass, no grid barrier
        self.dgrad_bn = os.environ.get('YOLO_DGRAD_BN', '1') != '0'
        # the stem's backward pass (un-pool + BatchNorm apply + weight gradient) as one kernel that never writes the pre-pool gradient
        # (PoolOp.plan_fusion; YOLO_STEM_BWD=0 for A/B runs)
        self.stem_bwd = os.environ.get('YOLO_STEM_BWD', '1') != '0'
        # small maps: BatchNorm finalize + apply in one launch when a unit has at most this many partial rows (0 = never; forward and
        # backward; ops.bn_finalize_act_fwd / ops.bn_bwd_finalize_apply)
        # BatchNorm statistics through exact int64 accumulators (ops.conv2d_fwd(stat_acc=...), yolo_acc_*): the producing convolution adds
        # its tile sums with integer atomics -- associative, so bit-reproducible -- into YOLO_ACC_NB = 8 buckets, and the unit's finalize + apply run as ONE
        # launch whatever the layer size (with partial ROWS that only pays below ~128 rows): the separate finalize launches (~5 us of
        # dependent-launch floor each, forward and backward) disappear for every plain conv -> BatchNorm unit.  YOLO_STAT_ACC=0 for A/B runs.
        # Measured (profiles/HISTORY.md, round 3): the one-workgroup-per-CU merged launch wins below ~6 M elements (the 26 x 26 and 13 x 13 maps at
        # batch 32: 8-9 us against 6.5 + 5.5); on larger tensors it is slower than finalize + the 8-workgroups-per-CU streaming apply, and the
        # streaming kernels that derive their constants from the block themselves (yolo_set_tuning acc_stream_kelems) pay ~6 us of prologue --
        # 2048 workgroups re-reading the same 16 KB of accumulator lines -- for the ~5.5 us launch they save.  YOLO_ACC_MAX_ELEMS caps the
        # units that take the accumulator path.
        self.stat_acc = os.environ.get('YOLO_STAT_ACC', '0') != '0'
        self.acc_max_elems = int(float(os.environ.get('YOLO_ACC_MAX_ELEMS', '6e6')))
        self.acc_buf = None
        # two-level partial rows (round 4; ops.conv2d_stat_group_layout, conv_common.h rows_fold): the convolution epilogues fold their per-tile
        # statistics rows in groups of 16-64 tiles, every unit then has <= ~85 rows, and its finalize + apply run as ONE launch (the streaming
        # form of ops.bn_finalize_act_fwd / ops.bn_bwd_finalize_apply sums the rows in its prologue).  Built, bit-exact and deterministic
        # (tests/test_row_groups_gpu.py), measured and OFF: 23 of the 46 finalize launches go and the BatchNorm kernels take 80 us less per step,
        # but every workgroup of every convolution now drains its stores and waits for a returning device-scope atomic before it leaves its CU
        # -- 3-9 us per launch, +205 us per step: 7.90 k against 8.13 k images/s (profiles/r04_row_groups_ab_*).  YOLO_ROW_GROUPS=1 turns it on.
        self.row_groups = os.environ.get('YOLO_ROW_GROUPS', '0') != '0'
        self.fin_merge_rows = int(os.environ.get('YOLO_FIN_MERGE_ROWS', '128'))
        self.fin_merge_bwd_rows = int(os.environ.get('YOLO_FIN_MERGE_BWD_ROWS', str(self.fin_merge_rows)))
        self.vals = []
        self.bns = []            # every keras BatchNormalization (for checkpoints)
        self.bn_groups = []      # allocation units: a BNState or a MultiBN
        self.fwd, self.bwd = [], []
        self.training = True
        self.capturing = False             # inside a hipGraph capture (model._capture): cross-stream edges must be torch's capture-aware ones
        self.bn_momentum = BN_MOMENTUM     # 1.0 while evaluating with batch statistics: the moving averages then stay as they are
        self._alloc = []
        self._repack_table = None
        self.wgrad_stream = None
        self.wgrad_batch = 2               # at most this many weight gradients per main->side stream hand-off (see on_wgrad_stream; with the
                                           # fence-free local edges 2 measures +0.5 % over 4, 1 is -0.3 %: profiles/r04_wgrad_batch_ab.txt)
        self._wgrad_pending = []
        self._wgrad_cost, self.wgrad_cost_limit = 0.0, 18.0   # ... or as soon as the pending ones reach this many GFLOP (a big 3x3 layer goes alone)
        self.bucket_cut, self.bucket_offset, self.on_bucket = -1, 0, None
        self.buckets, self.bucket_tail = [], None
        self.tail_on_main = False          # set by the owner (single GPU, per-bucket updates): see bucket_done

    # ------------------------------------------------------------------------------------------------ allocation helpers
    def _buffer(self, shape, dtype=None):
        cell = {}
        self._alloc.append((cell, tuple(shape), dtype))
        return cell

    # ------------------------------------------------------------------------------------------------ factories
    def input(self, H, W, C=3):
        v = Val(self, 'input', (self.N, H, W, C))
        v.needs_grad = False
        self.input_val = v
        return v

    def _as_conv_input(self, x):
        """a conv reads an activation, the packed input image, or concat(upsample(act), act)"""
        if x.kind in ('act', 'input'):
            return x
        if x.kind == 'cat' and x.a.kind == 'up' and x.a.src.kind == 'act' and x.b.kind == 'act':
            return x
        return self.materialize(x, relu=False)

    def convolution(self, x, filters, kernel_size=(3, 3), strides=(1, 1), padding='same', use_bias=False, name=None,
                    init='he_normal'):
        x = self._as_conv_input(x)
        k, s = int(kernel_size[0]), int(strides[0])
        N, H, W, Cin = x.shape
        name = name or self.ps.layer_name('conv2d')
        cin_dev = 8 if x.kind == 'input' else Cin
        cout_dev = ops.pad_channels(filters) if use_bias else filters
        if init == 'he_normal':
            w = self.ps.he_normal((k, k, Cin, filters), k * k * Cin)
            kind, l2 = 'conv_kernel', L2_CONV_DECAY
        else:                                     # RandomNormal(stddev=0.01), no regulariser (yolov3_detector.py:98-100)
            w = torch.randn((k, k, Cin, filters), generator=self.ps.gen) * 0.01
            kind, l2 = 'head_kernel', 0.0
        wp = Param(name + '/kernel', kind, (k, k, Cin, filters), (cout_dev, k, k, cin_dev), l2)
        self.ps.add(wp, self.kernel_to_dev(w, wp))
        bp = None
        if use_bias:
            bp = self.ps.add(Param(name + '/bias', 'bias', (filters,), (cout_dev,), 0.0), torch.zeros(cout_dev))
        C0 = x.a.src.shape[3] if x.kind == 'cat' else 0
        p = ops.conv_problem(N, H, W, cin_dev, cout_dev, k, s, padding, C0=C0)
        y = Val(self, 'conv', (N, p.Ho, p.Wo, cout_dev), x=x, wp=wp, bp=bp, p=p, f32=use_bias, filters=filters)
        y.cell = self._buffer(y.shape, torch.float32 if use_bias else backend.torch_dtype())
        y.dy_cell = self._buffer(y.shape, backend.torch_dtype())
        lay = ops.conv2d_stat_group_layout(p) if (self.row_groups and not use_bias) else None
        y.stat_grouped = bool(lay and lay['group'] > 0)
        y.stat_rows = lay['alloc_rows'] if y.stat_grouped else ops.conv2d_stat_rows(p)       # rows of the buffer (zeroed once: arrival counters live in it)
        y.stat_groups = lay['groups'] if y.stat_grouped else y.stat_rows                     # rows the BatchNorm kernels read
        y.stat_cell = None if use_bias else self._buffer((2, y.stat_rows, cout_dev), torch.float32)
        y.wants_stats = False
        self.tape.append(ConvOp(self, y))
        return y

    @staticmethod
    def kernel_to_dev(w_hwio, wp):
        """TF HWIO -> device OHWI, zero-padded to the device shape"""
        co, k, _, ci = wp.dev_shape
        out = torch.zeros(wp.dev_shape)
        w = w_hwio.permute(3, 0, 1, 2)
        out[:w.shape[0], :, :, :w.shape[3]] = w
        return out

    @staticmethod
    def kernel_from_dev(w_dev, wp):
        k, _, ci, co = wp.tf_shape
        return w_dev.reshape(wp.dev_shape)[:co, :, :, :ci].permute(1, 2, 3, 0).contiguous()

    def batch_normalization(self, x):
        if x.kind not in ('conv', 'act'):
            x = self.materialize(x, relu=False)
        C = x.shape[3]
        bn = BNState(self, self.ps.layer_name('batch_normalization_v1'), C)
        self.bns.append(bn)
        self.bn_groups.append(bn)
        if x.kind == 'conv':
            x.wants_stats = True
        return Val(self, 'bn', x.shape, src=x, bn=bn)

    def activation(self, x):
        return self.materialize(x, relu=True)

    def add(self, identity, residual):
        if identity.kind not in ('bn', 'conv', 'act'):
            identity = self.materialize(identity, relu=False)
        if residual.kind not in ('bn', 'conv', 'act'):
            residual = self.materialize(residual, relu=False)
        return Val(self, 'sum', residual.shape, a=identity, b=residual)

    def max_pool(self, x):
        N, H, W, C = x.shape
        Ho, pt = ops.same_pad(H, 3, 2)
        Wo, pl = ops.same_pad(W, 3, 2)
        return Val(self, 'pool', (N, Ho, Wo, C), src=x, pt=pt, pl=pl)

    def up_sample(self, x):
        x = self.materialize(x, relu=False)
        N, H, W, C = x.shape
        return Val(self, 'up', (N, 2 * H, 2 * W, C), src=x)

    def concat(self, a, b):
        if b.kind != 'act':
            b = self.materialize(b, relu=False)
        return Val(self, 'cat', b.shape[:3] + (a.shape[3] + b.shape[3],), a=a, b=b)

    def mix_depthwise_conv_bn(self, x, split, ksizes):
        """4 x (channel slice -> DepthwiseConv2D(k) -> BatchNormalization) -> concat (mixnet18.py:38-45) as one depthwise launch
        + one grouped BatchNorm.  Variables are created in the reference's order: dw_0, bn_0, dw_1, bn_1, ..."""
        x = self.materialize(x, relu=False)
        N, H, W, C = x.shape
        wps, bns = [], []
        for i, k in enumerate(ksizes):
            cg = split[i + 1] - split[i]
            name = self.ps.layer_name('depthwise_conv2d')
            w = self.ps.he_normal((k, k, cg, 1), k * k * cg)              # keras he_normal fan_in of a (k,k,Cg,1) kernel
            wp = Param(name + '/depthwise_kernel', 'dw_kernel', (k, k, cg, 1), (k, k, cg), L2_CONV_DECAY)
            self.ps.add(wp, w.reshape(k, k, cg))
            wps.append(wp)
            bn = BNState(self, self.ps.layer_name('batch_normalization_v1'), cg)
            self.bns.append(bn)
            bns.append(bn)
        mbn = MultiBN(bns, split)
        self.bn_groups.append(mbn)
        y = Val(self, 'conv', (N, H, W, C), x=x, wps=wps, mp=ops.mix_problem(N, H, W, C, split, ksizes), f32=False)
        y.cell = self._buffer(y.shape)
        y.dy_cell = self._buffer(y.shape)
        self.tape.append(MixConvOp(self, y))
        return Val(self, 'bn', y.shape, src=y, bn=mbn)

    def depthwise_conv(self, x, kernel_size=(3, 3)):
        """a lone keras DepthwiseConv2D (depth multiplier 1, stride 1, 'same', no bias, he_normal, L2 5e-4: reference basic_backbone.py:45-66) =
        the mixed depthwise launch with one channel group.  The kernel wants C / 8 to be a power of two <= 64."""
        x = self.materialize(x, relu=False)
        N, H, W, C = x.shape
        k = int(kernel_size[0])
        if k not in (1, 3, 5, 7, 9) or int(kernel_size[1]) != k:
            raise NotImplementedError('DepthwiseConv2D kernel sizes on this path: 1, 3, 5, 7, 9 (square)')
        if C % 8 or (C // 8) & (C // 8 - 1) or C > 512:
            raise NotImplementedError('DepthwiseConv2D over %d channels: C / 8 must be a power of two <= 64' % C)
        name = self.ps.layer_name('depthwise_conv2d')
        wp = Param(name + '/depthwise_kernel', 'dw_kernel', (k, k, C, 1), (k, k, C), L2_CONV_DECAY)
        self.ps.add(wp, self.ps.he_normal((k, k, C, 1), k * k * C).reshape(k, k, C))
        y = Val(self, 'conv', (N, H, W, C), x=x, wps=[wp], mp=ops.mix_problem(N, H, W, C, [0, C, C, C, C], [k, 3, 3, 3]), f32=False)
        y.cell = self._buffer(y.shape)
        y.dy_cell = self._buffer(y.shape)
        y.wants_stats = False
        self.tape.append(MixConvOp(self, y))
        return y

    @staticmethod
    def dw_to_dev(w, wp):
        return w.reshape(wp.dev_shape)

    @staticmethod
    def dw_from_dev(w_dev, wp):
        return w_dev.reshape(wp.tf_shape).contiguous()

    # ------------------------------------------------------------------------------------------------ lowering
    def materialize(self, v, relu):
        if v.kind == 'act' and not relu:
            return v
        if v.cached is not None and v.cached[0] == relu:
            return v.cached[1]
        out = Val(self, 'act', v.shape)
        out.cell = self._buffer(out.shape)
        out.grad_cell = self._buffer(out.shape)
        if v.kind in ('bn', 'conv', 'act'):
            op = ApplyOp(self, out, relu, main=v)
        elif v.kind == 'sum':
            a, b = v.a, v.b
            for t in (a, b):
                if t.kind not in ('bn', 'conv', 'act'):
                    raise NotImplementedError('sum operand of kind ' + t.kind)
            main, other = (b, a) if b.kind in ('bn', 'conv') else (a, b)
            op = ApplyOp(self, out, relu, main=main, other=other)
        elif v.kind == 'pool':
            op = PoolOp(self, out, relu, v)
        else:
            raise NotImplementedError('cannot materialise ' + v.kind)
        self.tape.append(op)
        v.cached = (relu, out)
        return out

    # ------------------------------------------------------------------------------------------------ finalisation
    def finalize(self, heads, loss_cfg_kwargs=None):
        """allocate every buffer, build the forward / backward launch lists"""
        dev = self.dev
        self.heads = heads
        self.ps.allocate(dev)
        for cell, shape, dtype in self._alloc:
            cell['t'] = torch.zeros(shape, dtype=dtype if dtype is not None else backend.torch_dtype(), device=dev)
        for bn in self.bn_groups:
            bn.allocate(dev, self.ps)
        N, H, W, C = self.input_val.shape
        self.images = torch.zeros(N, H, W, C, device=dev)
        self.input_val.buf = torch.zeros(N, H, W, 8, dtype=backend.torch_dtype(), device=dev)
        # dgrad weight copies
        n_dg = 0
        for op in self.tape:
            if isinstance(op, ConvOp) and op.needs_dgrad():
                op.dg_off = n_dg
                n_dg += _round_up(op.y.wp.numel, SLOT)
        self.w_dgrad = torch.zeros(max(n_dg, SLOT), dtype=backend.torch_dtype(), device=dev)
        # slab arena of the two-phase weight gradients: every convolution whose plan splits the pixels keeps its partial [Cout][R][S][Cin]
        # slabs in a PRIVATE region until its gradient bucket is complete; one launch per bucket then sums all of them (bucket_done).
        # (the depthwise weight gradients keep their own shared workspace: they run back to back on one stream and sum right away)
        n_slab = 0
        for op in self.tape:
            if isinstance(op, PoolOp) and self.stem_bwd:
                op.plan_fusion()
        for op in self.tape:
            if isinstance(op, ConvOp):
                op.splits = op.fused_slabs or ops.conv2d_wgrad_splits(op.y.p)
                op.slab_off = n_slab
                if op.splits > 1:
                    n_slab += op.splits * op.y.wp.numel
        self.slab_arena = torch.empty(max(n_slab, 4), dtype=torch.float32, device=dev)
        ws_bytes = max([ops.dwconv_mix_wgrad_workspace_bytes(op.y.mp) for op in self.tape if isinstance(op, MixConvOp)] + [16])
        self.wgrad_ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        # workspace + hand-off words of the single-launch BatchNorm backward (all on the main stream, one at a time)
        cmax = max([op.out.shape[3] for op in self.tape if isinstance(op, ApplyOp)] + [8])
        self.bn_ws = torch.zeros(ops.bn_bwd_fused_workspace_floats(cmax), dtype=torch.float32, device=dev)
        self.bn_sync = torch.zeros(ops.bn_bwd_fused_sync_words(), dtype=torch.int32, device=dev)
        for op in self.tape:
            op.bind()
        self.fwd = [lambda: ops.pack_input(self.images, self.input_val.buf, N * H * W, C)]
        for op in self.tape:
            self.fwd.append(op.forward)
        self.bwd = []
        for op in reversed(self.tape):
            op.plan_backward()
            self.bwd.append(op.backward)
        if self.dgrad_bn:
            for op in self.tape:
                if isinstance(op, ApplyOp):
                    op.plan_fusion()
            for op in self.tape:
                if isinstance(op, ApplyOp) and os.environ.get('YOLO_SHORTCUT_ALIAS', '1') != '0':
                    op.plan_shortcut_alias()
        # down-sampling blocks: one tensor feeds a 1x1 / stride-2 shortcut convolution and a 3x3 / stride-2 convolution.  The shortcut's
        # data gradient is zero at three quarters of the positions: it writes the even / even ones only, and the 3x3 gradient (four parity
        # classes) accumulates onto that class alone (YOLO_S2_SPARSE=0 for A/B runs)
        if os.environ.get('YOLO_S2_SPARSE', '1') != '0':
            for v in self.vals_with_writers():
                w = v.grad_writers
                for i, a in enumerate(w):
                    if not (isinstance(a, ConvOp) and a.y.x is v):
                        continue
                    pa = a.y.p
                    if not (pa.R == 1 and pa.S == 1 and pa.stride == 2 and pa.pad_t == 0 and pa.pad_l == 0 and pa.Cout % 64 == 0):
                        continue
                    if a.bn_epi is not None or a.addend is not None:
                        continue                           # (the even-only launch takes neither a fused reduce nor an external addend)
                    if a.acc == [True]:                    # adds onto an earlier (dense) contribution: only where it has one of its own
                        a.even_only = True
                    elif a.acc == [False] and i + 1 < len(w):
                        b = w[i + 1]                       # first writer: the next one must write the other three quarters itself
                        if (isinstance(b, ConvOp) and b.y.x is v and b.acc == [True] and b.addend is None and
                                ops.conv2d_dgrad_classed(b.y.p)):
                            a.even_only = True
                            b.acc = [2]
        self.plan_accumulators()
        # gradient buckets for data-parallel overlap, by backbone stage (parameters are laid out in creation order, backward runs in reverse):
        #   [first stride-32 conv, n)   module512 + the three heads, ~70 % of the parameters: complete ~40 % into the backward pass
        #   [first stride-8 conv, that) the stride-8 / stride-16 stages
        #   [0, first stride-8 conv)    stem + stride-4 stage (~1 % of the parameters): the only bucket whose all-reduce is exposed
        # self.buckets = [(index in the backward launch list after which the bucket is complete, lo, hi), ...] in completion order
        # Data parallel (WORLD_SIZE > 1, or YOLO_BUCKET_MB set): a stage bucket larger than YOLO_BUCKET_MB (default 16 MB of float32
        # gradient) is cut further at convolution boundaries, so that the first all-reduce starts a few layers into the backward pass
        # instead of after module512 + all three heads (~47 MB in one collective).
        H = self.input_val.shape[1]
        convs = [(len(self.tape) - 1 - i, op.y.wp.offset, op.y.shape[1]) for i, op in enumerate(self.tape) if isinstance(op, ConvOp)]
        mb = os.environ.get('YOLO_BUCKET_MB')
        if mb is None and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            mb = '16'
        target = int(float(mb) * (1 << 20) / 4) if mb else None
        self.buckets = plan_buckets(convs, H, self.ps.n, target)
        hi = self.buckets[-1][1] if self.buckets else None
        self.bucket_tail = hi                                  # [0, bucket_tail) remains after the backward pass (None: everything)
        if self.buckets:
            self.bucket_cut, self.bucket_offset = self.buckets[0][0], self.buckets[0][1]     # (kept: first bucket, for introspection / tests)
        # slab-summing tables, one per bucket range (and one for everything): rows {dw float4 offset, slab float4 offset, float4s per slab,
        # slabs, first workgroup} of the bucket's split convolutions
        n = self.ps.n
        self.bucket_ranges = [(lo, n if hi_ is None else hi_) for _, lo, hi_ in self.buckets] + [(0, n if self.bucket_tail is None else self.bucket_tail)]
        self.reduce_tables = {}
        for lo, hi_ in self.bucket_ranges + [(0, n)]:
            rows, blocks = [], 0
            for op in self.tape:
                if isinstance(op, ConvOp) and op.splits > 1 and lo <= op.y.wp.offset < hi_:
                    n4 = op.y.wp.numel // 4
                    rows.append([op.y.wp.offset // 4, op.slab_off // 4, n4, op.splits, blocks])
                    blocks += ops.reduce_blocks(n4, op.splits)
            self.reduce_tables[(lo, hi_)] = (torch.tensor(rows, dtype=torch.int64, device=dev) if rows else None, len(rows), blocks)

    def vals_with_writers(self):
        """every activation value some backward op writes a gradient into"""
        seen, out = set(), []
        for op in self.tape:
            for v in (getattr(op, 'out', None), getattr(getattr(op, 'y', None), 'x', None)):
                if v is not None and id(v) not in seen and getattr(v, 'grad_writers', None):
                    seen.add(id(v))
                    out.append(v)
        return out

    def refresh_dgrad_weights(self):
        """flipped/transposed bf16 weight copies for the data-gradient pass, all layers in one launch"""
        if self._repack_table is None:
            rows, tiles = [], 0
            for op in self.tape:
                if isinstance(op, ConvOp) and op.needs_dgrad():
                    co, k, _, ci = op.y.wp.dev_shape
                    tci, tco = (ci + 31) // 32, (co + 31) // 32
                    rows.append([op.y.wp.offset, op.dg_off, co, k * k, ci, tiles, tci, tco])
                    tiles += k * k * tci * tco
            self._repack_table = (torch.tensor(rows, dtype=torch.int32, device=self.dev), len(rows), tiles)
        tab, n, tiles = self._repack_table
        if n:
            ops.repack_dgrad_weights_batched(self.ps.bf16, self.w_dgrad, tab, n, tiles)

    def run_forward(self):
        for f in self.fwd:
            f()

    def run_backward(self):
        """backward launch list.  With ``wgrad_stream`` set, every weight-gradient GEMM is enqueued on that second stream (forked by an
        event after its dY is complete, joined once at the end): the MFMA-bound wgrads then overlap the bandwidth-bound BatchNorm
        backward kernels and the tails of the data-gradient GEMMs.  Under hipGraph capture this becomes a forked graph."""
        side = self.wgrad_stream
        if self._repack_event is not None:        # the data-gradient weight copies were refreshed on the side stream (refresh_dgrad_async)
            if side is not None:
                self.stream_wait(torch.cuda.current_stream(self.dev), side, local=True)
            self._repack_event = None
        cuts = {cut: (lo, self.ps.n if hi is None else hi) for cut, lo, hi in self.buckets}
        for i, f in enumerate(self.bwd):
            f()
            if i in cuts:
                self.bucket_done(*cuts[i])
        self.bucket_done(*self.bucket_ranges[-1], last=True)
        if side is not None and not self.tail_on_main:
            self.stream_wait(torch.cuda.current_stream(self.dev), side, local=True)

    def plan_accumulators(self):
        """exact accumulator blocks (one flat int64 buffer, zeroed by ONE launch at the start of every step) for the BatchNorm units whose
        finalize + apply can run as one launch: forward statistics from the producing convolution's epilogue, backward tile sums from the
        epilogue of the data gradient that completes the unit's output gradient"""
        self.acc_buf = None
        if not self.stat_acc:
            return
        reqs, total = [], 0
        for op in self.tape:
            if not isinstance(op, ApplyOp):
                continue
            if op.acc_fwd_eligible():
                reqs.append((op, 'f', total))
                total += ops.acc_words(2, op.C)
            if op.acc_bwd_eligible():
                reqs.append((op, 'b', total))
                total += ops.acc_words(3, op.C)
        if not reqs:
            return
        self.acc_buf = torch.zeros(total, dtype=torch.int64, device=self.dev)
        for op, kind, off in reqs:
            if kind == 'f':
                op.acc_f = self.acc_buf[off:off + ops.acc_words(2, op.C)]
                op.m_src.acc_fwd = op.acc_f
            else:
                op.acc_b = self.acc_buf[off:off + ops.acc_words(3, op.C)]
                op.producer.bn_epi = dict(op.producer.bn_epi, acc=op.acc_b, partial=None)
        self.fwd.insert(0, lambda: ops.zero_words(self.acc_buf))

    def stream_wait(self, waiter, signaler, local=False):
        """``waiter`` (a torch stream) waits for everything queued on ``signaler`` so far.  Eager mode goes through the library
        (yolo_seq_fork) so that the edge becomes part of a recorded launch sequence; under hipGraph capture torch's own event does it.
        ``local``: what waits behind the edge are kernels of this device only (main <-> weight-gradient stream): yolo_seq_fork_local, an
        event without the system-scope writeback.  Edges in front of a collective (copy engines, peers, the host read) keep the default."""
        if self.capturing:
            waiter.wait_stream(signaler)
        else:
            ops.stream_fork(signaler, waiter, local=local)

    def reduce_slabs(self, lo, hi):
        tab, n, blocks = self.reduce_tables[(lo, hi)]
        if n:
            ops.wgrad_reduce_batched(tab, n, blocks, self.slab_arena, self.ps.grad)

    def bucket_done(self, lo, hi, last=False):
        """every gradient of the parameter range [lo, hi) has been enqueued (main stream: BatchNorm gradients; weight-gradient stream: the
        slab passes).  Hand what is pending to the weight-gradient stream together with the ONE launch that sums the bucket's slabs, then
        tell the owner (gradient exchange and / or the optimizer launch of this bucket, both behind that launch on the side stream).

        The LAST range of a step (stem + stride-4 stage: nothing is left to overlap it with) with ``tail_on_main``: the main stream joins the
        weight-gradient stream once and runs the slab sum -- and, through ``on_bucket(lo, hi, True)``, the owner's update -- itself: one
        cross-stream hand-off at the end of the step instead of two (each costs 10-20 us of idle GPU on this runtime)"""
        if last and self.tail_on_main and self.wgrad_stream is not None:
            self.flush_wgrad()
            self.stream_wait(torch.cuda.current_stream(self.dev), self.wgrad_stream, local=True)
            self.reduce_slabs(lo, hi)
            if self.on_bucket is not None:
                self.on_bucket(lo, hi, True)
            return
        self.on_wgrad_stream(lambda: self.reduce_slabs(lo, hi), flush=True)
        if self.on_bucket is not None:
            self.on_bucket(lo, hi, False)

    def owned_tensors(self):
        """every device tensor this graph (its parameter store, ops and BatchNorm states) holds"""
        seen, out = set(), []

        def visit(obj, depth):
            if isinstance(obj, torch.Tensor):
                if obj.is_cuda and obj.data_ptr() not in seen:
                    seen.add(obj.data_ptr())
                    out.append(obj)
            elif isinstance(obj, dict) and depth < 3:
                for v in obj.values():
                    visit(v, depth + 1)
            elif isinstance(obj, (list, tuple)) and depth < 3:
                for v in obj:
                    visit(v, depth + 1)
            elif hasattr(obj, '__dict__') and depth < 3 and not isinstance(obj, Graph):
                for v in vars(obj).values():
                    visit(v, depth + 1)

        for v in vars(self).values():
            visit(v, 0)
        return out

    def use_side_stream(self, stream):
        """tell the caching allocator that the graph's buffers are also used on `stream` (weight-gradient kernels, the asynchronous
        data-gradient weight repack): when the graph is dropped while such work is still queued, the memory is not handed to the next
        allocation until that work has finished (without this a following tensor could be scribbled on by a late kernel)"""
        for t in self.owned_tensors():
            t.record_stream(stream)

    def refresh_dgrad_async(self):
        """refresh_dgrad_weights on the weight-gradient stream (eager mode): the copies are first needed by the NEXT step's backward pass,
        so the repack overlaps the next forward instead of sitting between the optimizer and it"""
        if self.wgrad_stream is None:
            self.refresh_dgrad_weights()
            return
        self.on_wgrad_stream(self.refresh_dgrad_weights, flush=True)
        self._repack_event = True                  # the next backward pass first waits for the side stream (run_backward)

    def on_wgrad_stream(self, fn, flush=False, cost=0.0):
        """run fn() on the weight-gradient stream after everything enqueued so far on the current stream.  Hand-offs are BATCHED: an
        event record on the main stream plus the wait on the side stream stalls the main stream for ~15-25 us on this runtime (measured,
        tools/probes/event_cost.py: 4.7 us per kernel in a plain chain, 28 us with a hand-off after each), so the weight gradients of
        consecutive layers share one event (up to ``wgrad_batch`` of them, or ``wgrad_cost_limit`` GFLOP: the small 1x1 / stride-2 layers ride along
        with the next big one) -- their dY buffers are static, they only have to run before the optimizer"""
        side = self.wgrad_stream
        if side is None:
            fn()
            return
        self._wgrad_pending.append(fn)
        self._wgrad_cost += cost
        if flush or len(self._wgrad_pending) >= self.wgrad_batch or (self.wgrad_cost_limit and self._wgrad_cost >= self.wgrad_cost_limit):
            self.flush_wgrad()

    def flush_wgrad(self):
        side = self.wgrad_stream
        if not self._wgrad_pending:
            return
        pending, self._wgrad_pending = self._wgrad_pending, []
        self._wgrad_cost = 0.0
        if side is None:
            for fn in pending:
                fn()
            return
        self.stream_wait(side, torch.cuda.current_stream(self.dev), local=True)
        with torch.cuda.stream(side):
            for fn in pending:
                fn()


# ==================================================================================================================== ops
def plan_buckets(convs, H, n=None, target=None):
    """gradient buckets in completion order.  ``convs`` = [(index in the backward launch list after which this convolution's weight gradient
    (and everything created after it) is complete, offset of its kernel in the flat parameter buffer, output height)] in creation order;
    parameters are laid out in creation order and the backward pass runs in reverse, so the range [offset, n) is complete at that index.
    Stage marks: the first convolution at stride 32 and the first at stride 8 (the range [0, first stride-8 convolution) is the tail the
    caller handles after the backward pass).  ``target`` (elements, needs ``n`` = all parameters): additionally cut at a convolution
    whenever the bucket that ends there has reached that many parameters.  Returns [(cut index, lo, hi or None for "up to n")]."""
    marks = []
    for div in (32, 8):
        for cut, off, h in convs:
            if h == H // div and (not marks or off < marks[-1][1]):
                marks.append((cut, off))
                break
    cuts = {off: cut for cut, off in marks if off > 0}
    if target and n is not None and cuts:
        stop, hi = min(cuts), n                               # nothing below the last stage mark is cut further (the tail stays one range)
        for cut, off, _ in sorted(convs, key=lambda c: -c[1]):                # backward order: descending offsets
            if off < stop:
                break
            if off in cuts:
                hi = off
            elif hi - off >= target:
                cuts[off] = cut
                hi = off
    out, hi = [], None
    for off in sorted(cuts, reverse=True):
        out.append((cuts[off], off, hi))
        hi = off
    return out


class ConvOp(object):
    def __init__(self, g, y):
        self.g, self.y = g, y
        self.fused_slabs = 0        # > 0: a PoolOp's fused backward kernel writes this convolution's weight-gradient slabs (stem)

    def needs_dgrad(self):
        x = self.y.x
        return x.kind != 'input'

    def bind(self):
        g, y = self.g, self.y
        y.buf = y.cell['t']
        y.dy = y.dy_cell['t']
        ps = g.ps
        self.w = ps.view(y.wp, ps.bf16)
        self.dw = ps.view(y.wp, ps.grad)
        self.bias = ps.view(y.bp) if y.bp is not None else None
        self.dbias = ps.view(y.bp, ps.grad) if y.bp is not None else None
        self.ssum = self.ssq = None
        if y.stat_cell is not None and y.wants_stats:
            st = y.stat_cell['t']
            self.ssum, self.ssq = st[0], st[1]
            C = y.shape[3]
            y.stats = (st[0].view(-1), st[1].view(-1), y.stat_groups, C)      # (psum, psq, P, row_stride)
        x = y.x
        if x.kind == 'cat':
            self.src0, self.src1 = x.a.src, x.b
        else:
            self.src0, self.src1 = None, x
        if self.needs_dgrad():
            self.w_dg = g.w_dgrad[self.dg_off:self.dg_off + y.wp.numel]
        self.slabs = g.slab_arena[self.slab_off:self.slab_off + self.splits * y.wp.numel] if self.splits > 1 else None
        if y.bp is not None:
            C = y.shape[3]
            self.brow = ops.reduce_rows(y.M, C)
            self.bpart = torch.zeros(self.brow, 2, C, device=g.dev)

    def repack(self):
        co, k, _, ci = self.y.wp.dev_shape
        ops.repack_dgrad_weights(self.w, self.w_dg, co, k, k, ci)

    def forward(self):
        y = self.y
        acc = getattr(y, 'acc_fwd', None)
        if acc is not None and self.g.training:          # statistics into the unit's accumulator block (Graph.plan_accumulators)
            ops.conv2d_fwd(y.p, self.src1.buf, self.w, y.buf, src0=None if self.src0 is None else self.src0.buf, stat_acc=acc)
            return
        if self.ssum is not None and y.stat_grouped:
            ops.conv2d_fwd(y.p, self.src1.buf, self.w, y.buf, src0=None if self.src0 is None else self.src0.buf, stat_sum=self.ssum,
                           stat_sq=self.ssq, grouped=True)
            return
        ops.conv2d_fwd(y.p, self.src1.buf, self.w, y.buf, src0=None if self.src0 is None else self.src0.buf, bias=self.bias,
                       stat_sum=self.ssum, stat_sq=self.ssq)

    def plan_backward(self):
        y = self.y
        self.acc = []
        self.bn_epi = None          # set by ApplyOp.plan_fusion: this data gradient completes a BatchNorm unit's output gradient
        self.addend = None          # set by ApplyOp.plan_shortcut_alias: the fan-in contribution is read from that Val's .grad
        self.even_only = False      # set by Graph.finalize: 1x1 / stride-2 shortcut whose gradient is written at the even / even positions only
        if not self.needs_dgrad():
            return
        x = y.x
        if x.kind == 'cat':
            self.dcat = torch.zeros(x.shape, dtype=backend.torch_dtype(), device=self.g.dev)
            a, b = x.a.src, x.b
            self.acc = [a.grad_init, b.grad_init]
            a.grad_init = b.grad_init = True
            a.grad_writers.append(('cat', self))
            b.grad_writers.append(('cat', self))
            import copy
            self.pd = copy.copy(y.p)
            self.pd.C0 = 0
        else:
            self.acc = [x.grad_init]
            x.grad_init = True
            x.grad_writers.append(self)

    def _wgrad(self):
        y = self.y
        s0 = None if self.src0 is None else self.src0.buf
        ops.conv2d_wgrad_slabs(y.p, self.src1.buf, y.dy, self.dw, self.slabs, src0=s0)
        if self.dbias is not None:
            C = y.shape[3]
            ops.bn_stats(y.dy, y.M, C, self.bpart)
            ops.reduce_partials(self.bpart, self.brow, 2 * C, C, self.dbias)

    def backward(self):
        y = self.y
        p = y.p
        if not self.fused_slabs:
            self.g.on_wgrad_stream(self._wgrad, cost=2e-9 * p.N * p.Ho * p.Wo * p.Cout * p.Cin * p.R * p.S)
        if not self.needs_dgrad():
            return
        x = y.x
        if x.kind == 'cat':
            ops.conv2d_dgrad(self.pd, y.dy, self.w_dg, self.dcat)
            a, b = x.a.src, x.b
            N, H, W, _ = x.shape
            ops.upcat_split_bwd(self.dcat, a.grad, self.acc[0], b.grad, self.acc[1], N, H, W, a.shape[3], b.shape[3])
        else:
            ops.conv2d_dgrad(y.p, y.dy, self.w_dg, x.grad, accumulate=self.acc[0], bn=self.bn_epi,
                             addend=None if self.addend is None else self.addend.grad, even_only=self.even_only)


class MixConvOp(object):
    """mixed depthwise conv (one launch for the 4 kernel sizes) + per-channel statistics for the grouped BatchNorm"""

    def __init__(self, g, y):
        self.g, self.y = g, y

    def bind(self):
        g, y = self.g, self.y
        y.buf = y.cell['t']
        y.dy = y.dy_cell['t']
        ps = g.ps
        self.w = [ps.view(wp, ps.bf16) for wp in y.wps]
        self.dw = [ps.view(wp, ps.grad) for wp in y.wps]
        self.w += self.w[:1] * (4 - len(self.w))             # a lone DepthwiseConv2D has one (non-empty) channel group: the empty groups'
        self.dw += self.dw[:1] * (4 - len(self.dw))          # pointers are never dereferenced, but the C-ABI wants them non-null
        C = y.shape[3]
        self.P = ops.reduce_rows(y.M, C)
        self.part = torch.zeros(self.P, 2, C, device=g.dev)
        flat = self.part.view(-1)
        y.stats = (flat, flat[C:], self.P, 2 * C)

    def forward(self):
        y = self.y
        ops.dwconv_mix_fwd(y.mp, y.x.buf, self.w, y.buf)
        ops.bn_stats(y.buf, y.M, y.shape[3], self.part)

    def plan_backward(self):
        x = self.y.x
        self.acc = x.grad_init
        x.grad_init = True
        x.grad_writers.append(self)

    def backward(self):
        y = self.y
        self.g.on_wgrad_stream(lambda: ops.dwconv_mix_wgrad(y.mp, y.x.buf, y.dy, self.dw, self.g.wgrad_ws))
        ops.dwconv_mix_dgrad(y.mp, y.dy, self.w, y.x.grad, accumulate=self.acc)


def _branch(v):
    """decompose an apply operand into (tensor value, bn object or None)"""
    if v.kind == 'bn':
        return v.src, v.bn
    return v, None


class ApplyOp(object):
    """out = act(BN?(main) + T),  T = nothing | act | BN(conv)   (bn_act_fwd and its two backward passes)"""

    def __init__(self, g, out, relu, main, other=None):
        self.g, self.out, self.relu = g, out, relu
        self.m_src, self.m_bn = _branch(main)
        self.o_src, self.o_bn = (None, None) if other is None else _branch(other)
        if self.o_src is not None and self.o_bn is None and self.o_src.kind == 'conv':
            raise NotImplementedError('raw conv output as the secondary sum operand')

    def bind(self):
        g, out = self.g, self.out
        out.buf = out.cell['t']
        out.grad = out.grad_cell['t']
        C = out.shape[3]
        self.C, self.M = C, out.M
        # ReLU sign bits of the output, one byte per 8-channel chunk: the backward kernels read this instead of the activation (relu code 2)
        self.mask = torch.zeros(self.M * (C // 8), dtype=torch.uint8, device=g.dev) if (self.relu and g.relu_mask) else None
        self.P = ops.reduce_rows(self.M, C)
        self.partial = torch.zeros(self.P, 3, C, device=g.dev)
        self.pflat = self.partial.view(-1)
        for src, bn in ((self.m_src, self.m_bn), (self.o_src, self.o_bn)):
            if bn is not None and src.kind == 'act':        # BN over a materialised tensor: statistics by a separate pass
                bn.stat_part = torch.zeros(self.P, 2, C, device=g.dev)

    def _finalize_bn(self, src, bn):
        if src.kind == 'conv':
            psum, psq, P, rs = src.stats
        else:
            if self.g.training:
                ops.bn_stats(src.buf, self.M, self.C, bn.stat_part)
            flat = bn.stat_part.view(-1)
            psum, psq, P, rs = flat, flat[self.C:], self.P, 2 * self.C
        bn.fwd_finalize(psum, psq, P, rs, self.M, self.g.training)

    def acc_fwd_eligible(self):
        """forward statistics through an accumulator block: one plain BatchNorm over a (non-stem) convolution's output, nothing normalised
        on the other operand -- the merged finalize + apply launch then serves the unit at any size"""
        g, mb, m = self.g, self.m_bn, self.m_src
        if not (g.stat_acc and mb is not None and self.o_bn is None and m.kind == 'conv' and isinstance(mb, BNState) and len(mb.parts) == 1
                and self.C % 32 == 0 and getattr(m, 'stats', None) is not None and self.M * self.C <= g.acc_max_elems):
            return False
        return ops.conv2d_fwd_plan(m.p)['family'] != 'stem'

    def acc_bwd_eligible(self):
        """backward: the producer's epilogue leaves the unit's tile sums; one BatchNorm, the unit writes its own dy (no alias)"""
        mb = self.m_bn
        if not (self.g.stat_acc and self.producer is not None and self.M * self.C <= self.g.acc_max_elems):
            return False
        y1, b1, y2, b2 = self._reduce_operands()
        return b2 is None and b1 is mb and isinstance(mb, BNState) and len(mb.parts) == 1 and self.C % 32 == 0 and not self.skip_dy

    def _merged_fwd(self):
        """(psum, psq, P, row_stride) if this unit's finalize + apply run as one launch: training, one plain BatchNorm over a conv output
        with few partial rows, no BatchNorm on the other operand"""
        g, mb, m = self.g, self.m_bn, self.m_src
        if not (g.training and g.fin_merge_rows and mb is not None and self.o_bn is None and m.kind == 'conv' and isinstance(mb, BNState)
                and len(mb.parts) == 1 and self.C % 32 == 0):
            return None
        st = m.stats
        return st if st[2] <= g.fin_merge_rows else None

    def forward(self):
        if getattr(self, 'acc_f', None) is not None and self.g.training:
            mb = self.m_bn
            ops.bn_finalize_act_fwd_acc(self.acc_f, self.C, self.M, mb.v_gamma, mb.v_beta, BN_EPSILON, self.g.bn_momentum, mb.moving_mean,
                                        mb.moving_var, mb.scale, mb.shift, mb.mean, mb.rstd, self.m_src.buf, self.out.buf, self.M, self.relu,
                                        res=None if self.o_src is None else self.o_src.buf, mask=self.mask)
            return
        st = self._merged_fwd()
        if st is not None:
            mb = self.m_bn
            ops.bn_finalize_act_fwd(st[0], st[1], st[2], st[3], self.C, self.M, mb.v_gamma, mb.v_beta, BN_EPSILON, self.g.bn_momentum,
                                    mb.moving_mean, mb.moving_var, mb.scale, mb.shift, mb.mean, mb.rstd, self.m_src.buf, self.out.buf,
                                    self.M, self.relu, res=None if self.o_src is None else self.o_src.buf, mask=self.mask)
            return
        if self.m_bn is not None:
            self._finalize_bn(self.m_src, self.m_bn)
        if self.o_bn is not None:
            self._finalize_bn(self.o_src, self.o_bn)
        sc, sh = (self.m_bn.scale, self.m_bn.shift) if self.m_bn is not None else (None, None)
        kw = {}
        if self.o_src is not None:
            kw['res'] = self.o_src.buf
            if self.o_bn is not None:
                kw['res_scale'], kw['res_shift'] = self.o_bn.scale, self.o_bn.shift
        ops.bn_act_fwd(self.m_src.buf, sc, sh, self.out.buf, self.M, self.C, self.relu, mask=self.mask, **kw)

    def plan_backward(self):
        m = self.m_src
        if m.kind == 'conv':
            self.m_dst, self.m_acc = 'dy', False
        else:
            self.m_dst, self.m_acc = 'grad', m.grad_init
            m.grad_init = True
            m.grad_writers.append(self)
        o = self.o_src
        self.o_acc = False
        if o is not None and o.kind == 'act':
            self.o_acc = o.grad_init
            o.grad_init = True
            o.grad_writers.append(self)
        self.producer = None
        self.skip_dres = False
        self.skip_dy = False

    def plan_shortcut_alias(self):
        """identity shortcut of a residual unit whose masked gradient g already sits in out.grad (left there by the producer's epilogue):
        the shortcut's gradient IS g.  If the shortcut tensor's gradient has exactly one more contribution, a data-gradient convolution
        that runs after this unit and accumulates, that launch reads g from out.grad directly and the copy into the shortcut's
        gradient buffer (2 B / element written, then read back) is dropped"""
        o = self.o_src
        # out.grad IS the gradient g of this unit's inputs: left masked by the producer's epilogue, or -- a plain sum without BatchNorm
        # and ReLU (ResNet18-v2's residual adds, resnet18_v2.py:38-58) -- as it arrived
        plain_sum = self.m_bn is None and self.o_bn is None and not self.relu
        if self.producer is None and not plain_sum:
            return
        if plain_sum and self.m_src.kind == 'conv' and self.m_dst == 'dy':
            self.m_src.dy = self.out.grad          # the convolution's output gradient is this buffer: no copy (its kernels read y.dy when launched)
            self.skip_dy = True
        if o is None or self.o_bn is not None or o.kind != 'act' or self.o_acc:
            return
        w = o.grad_writers
        if len(w) == 2 and w[0] is self and isinstance(w[1], ConvOp) and w[1].y.x is o:
            w[1].addend = self.out
            self.skip_dres = True

    def _reduce_operands(self):
        """(y1, bn1, y2, bn2) of the backward reduction: quantity 1 belongs to the main BN if there is one, else to the shortcut BN"""
        m, o, mb, ob = self.m_src, self.o_src, self.m_bn, self.o_bn
        y1, b1 = (m, mb) if mb is not None else (o, ob)
        y2, b2 = (o, ob) if (mb is not None and ob is not None) else (None, None)
        return y1, b1, y2, b2

    def plan_fusion(self):
        """if the LAST writer of this unit's output gradient is a plain data-gradient convolution, that launch takes over the ReLU masking
        and the reduce of this unit's backward pass (ops.conv2d_dgrad(bn=...)): backward() is then finalize + apply on the masked gradient"""
        w = self.out.grad_writers
        if not w or not isinstance(w[-1], ConvOp) or w[-1].y.x is not self.out or (self.m_bn is None and self.o_bn is None):
            return
        if self.relu and self.mask is None:
            # YOLO_RELU_MASK=0: the epilogue could only take the ReLU derivative from a sign-byte mask (a null mask means "linear unit"
            # to yolo_conv2d_dgrad_bn, and backward() then runs the apply with relu = 0): the unit keeps its own reduce pass
            return
        conv = w[-1]
        rows = ops.conv2d_dgrad_bn_rows(conv.y.p)
        if rows <= 0:
            return
        y1, b1, y2, b2 = self._reduce_operands()
        lay = ops.conv2d_dgrad_bn_group_layout(conv.y.p) if self.g.row_groups else None
        grouped = bool(lay and lay['group'] > 0)
        self.frows = lay['groups'] if grouped else rows
        self.fpartial = torch.zeros(lay['alloc_rows'] if grouped else rows, 3, self.C, device=self.g.dev)     # rows a launch does not write stay zero
        conv.bn_epi = dict(mask=self.mask, y=y1.buf, mean=b1.mean, rstd=b1.rstd, partial=self.fpartial, grouped=grouped)
        if b2 is not None:
            conv.bn_epi.update(y2=y2.buf, mean2=b2.mean, rstd2=b2.rstd)
        self.producer = conv

    def backward(self):
        out, m, o = self.out, self.m_src, self.o_src
        mb, ob = self.m_bn, self.o_bn
        sign, relu = (self.mask, 2) if self.mask is not None else (out.buf, self.relu)      # where the ReLU mask comes from
        grouped = mb is not None and len(mb.parts) > 1
        if self.producer is not None:
            # out.grad already holds the masked gradient and self.fpartial its tile sums (left by the producer's epilogue)
            y1, b1, y2, b2 = self._reduce_operands()
            if getattr(self, 'acc_b', None) is not None:
                dres = o.grad if (o is not None and not self.skip_dres) else None
                ops.bn_bwd_finalize_apply_acc(self.acc_b, self.C, self.M, mb.v_dgamma, mb.v_dbeta, mb.k1, mb.k2, out.grad, m.buf, mb.scale, mb.mean,
                                              mb.rstd, self.M, m.dy if self.m_dst == 'dy' else m.grad, acc_dy=self.m_acc, dres=dres,
                                              acc_dres=self.o_acc)
                return
            if (b2 is None and b1 is mb and isinstance(mb, BNState) and len(mb.parts) == 1 and self.C % 32 == 0 and not self.skip_dy
                    and 0 < self.frows <= self.g.fin_merge_bwd_rows):
                dres = o.grad if (o is not None and not self.skip_dres) else None          # (ob is None here: b2 is None and b1 is mb)
                ops.bn_bwd_finalize_apply(self.fpartial.view(-1), self.frows, self.C, self.M, mb.v_dgamma, mb.v_dbeta, mb.k1, mb.k2, out.grad,
                                          m.buf, mb.scale, mb.mean, mb.rstd, self.M, m.dy if self.m_dst == 'dy' else m.grad,
                                          acc_dy=self.m_acc, dres=dres, acc_dres=self.o_acc)
                return
            b1.bwd_finalize(self.fpartial.view(-1), self.frows, self.C, 1, self.M)
            if b2 is not None:
                b2.bwd_finalize(self.fpartial.view(-1), self.frows, self.C, 2, self.M)
            sign, relu = None, 0
        elif self.g.fused_bn_bwd and mb is not None and (ob is None or len(ob.parts) == 1) and \
                (not grouped or (mb.parts[0][1] == 0 and mb._bounds()[-1] == mb.C)):
            kw = {}
            if o is not None:
                if ob is not None:
                    kw.update(y2=o.buf, a2=ob.scale, mean2=ob.mean, rstd2=ob.rstd, dgamma2=ob.v_dgamma, dbeta2=ob.v_dbeta, dy2=o.dy)
                else:
                    kw.update(dres=o.grad, acc_dres=self.o_acc)
            if grouped:
                dgs = (mb._bounds(), [b.v_dgamma for b, _ in mb.parts])
                dbs = [b.v_dbeta for b, _ in mb.parts]
            else:
                dgs, dbs = mb.v_dgamma, mb.v_dbeta
            if ops.bn_act_bwd_fused(out.grad, sign, relu, self.M, self.C, m.buf, mb.scale, mb.mean, mb.rstd, dgs,
                                    dbs, m.dy if self.m_dst == 'dy' else m.grad, self.g.bn_ws, self.g.bn_sync,
                                    acc_dy=self.m_acc, **kw):
                return
        if self.producer is None and (mb is not None or ob is not None):
            y1, b1, y2, b2 = self._reduce_operands()
            ops.bn_act_bwd_reduce(out.grad, sign, relu, y1.buf, b1.mean, b1.rstd, self.M, self.C, self.partial,
                                  y2=None if y2 is None else y2.buf, mean2=None if b2 is None else b2.mean,
                                  rstd2=None if b2 is None else b2.rstd)
            b1.bwd_finalize(self.pflat, self.P, self.C, 1, self.M)
            if b2 is not None:
                b2.bwd_finalize(self.pflat, self.P, self.C, 2, self.M)
        kw = {}
        if self.skip_dy and (o is None or self.skip_dres):
            return                                  # a plain sum whose operands read out.grad in place: nothing to launch
        m_dst = m.dy if self.m_dst == 'dy' else m.grad
        if mb is not None:
            kw.update(y=m.buf, a1=mb.scale, mean=mb.mean, rstd=mb.rstd, k1=mb.k1, k2=mb.k2)       # gamma * rstd == forward scale
        if not self.skip_dy:
            kw.update(dy=m_dst, acc_dy=self.m_acc)
        if o is not None:
            if ob is not None:
                kw.update(y2=o.buf, a2=ob.scale, mean2=ob.mean, rstd2=ob.rstd, k1b=ob.k1, k2b=ob.k2, dy2=o.dy)
            elif not self.skip_dres:
                kw.update(dres=o.grad, acc_dres=self.o_acc)
        ops.bn_act_bwd_apply(out.grad, sign, relu, self.M, self.C, **kw)


class PoolOp(object):
    """out = act(maxpool3x3s2(BN?(conv)))   (stem: resnet18.py:59-61; v2: resnet18_v2.py:61-62)"""

    def __init__(self, g, out, relu, v):
        self.g, self.out, self.relu, self.v = g, out, relu, v
        self.src, self.bn = _branch(v.src)
        if self.src.kind != 'conv':
            raise NotImplementedError('max-pool over a non-conv value')
        self.conv_op = None

    def plan_fusion(self):
        """the stem (image -> conv3x3 s2 -> [BN] -> max-pool -> [ReLU]): its convolution has no data gradient, so the pre-pool gradient
        feeds the weight gradient only -- one kernel does un-pool + BatchNorm apply + weight gradient (ops.stem_pool_bwd_wgrad)"""
        src = self.src
        conv = next((op for op in self.g.tape if isinstance(op, ConvOp) and op.y is src), None)
        if conv is None or conv.needs_dgrad() or src.bp is not None or (self.bn is not None and len(getattr(self.bn, 'parts', [0])) > 1):
            return
        out = self.out
        n = ops.stem_pool_bwd_slabs(src.p, src.shape[3], out.shape[1], out.shape[2], self.v.pt, self.v.pl)
        if n > 0:
            conv.fused_slabs = n
            self.conv_op = conv

    def bind(self):
        g, out = self.g, self.out
        out.buf = out.cell['t']
        out.grad = out.grad_cell['t']
        self.argmax = torch.zeros(out.shape, dtype=torch.uint8, device=g.dev)
        N, H, W, C = self.src.shape
        self.geom = (N, H, W, C, out.shape[1], out.shape[2], self.v.pt, self.v.pl)
        self.P = ops.reduce_rows(N * H * W, C)
        self.partial = torch.zeros(self.P, 3, C, device=g.dev)

    def forward(self):
        src, bn = self.src, self.bn
        N, H, W, C = src.shape
        if bn is not None:
            psum, psq, P, rs = src.stats
            bn.fwd_finalize(psum, psq, P, rs, N * H * W, self.g.training)
        ops.bn_pool_fwd(src.buf, None if bn is None else bn.scale, None if bn is None else bn.shift, self.out.buf, self.argmax,
                        *self.geom, self.relu)

    def plan_backward(self):
        pass

    def backward(self):
        src, bn, out = self.src, self.bn, self.out
        N, H, W, C = src.shape
        if bn is not None:
            ops.bn_pool_bwd_reduce(out.grad, out.buf, self.argmax, self.relu, src.buf, bn.mean, bn.rstd, *self.geom, self.partial,
                                   gamma=getattr(bn, 'v_gamma', None), beta=getattr(bn, 'v_beta', None))
            bn.bwd_finalize(self.partial.view(-1), self.P, C, 1, N * H * W)
        a1, mean, rstd, k1, k2 = (bn.scale, bn.mean, bn.rstd, bn.k1, bn.k2) if bn is not None else (None,) * 5
        conv = self.conv_op
        if conv is not None:
            Ho, Wo, pt, pl = self.geom[4:]
            ops.stem_pool_bwd_wgrad(src.p, conv.src1.buf, out.grad, out.buf, self.argmax, self.relu, src.buf, a1, mean, rstd, k1, k2,
                                    Ho, Wo, pt, pl, conv.slabs if conv.slabs is not None else conv.dw)      # one tile: straight into dW
        else:
            ops.bn_pool_bwd_apply(out.grad, out.buf, self.argmax, self.relu, src.buf if bn is not None else None, a1, mean, rstd, k1, k2,
                                  src.dy, *self.geom)
