"""YOLOv3Model: what ``YOLOv3Detector.build`` returns in place of the reference's keras Model
(yolov3/yolov3_detector.py:55-59) -- the static native graph plus the keras-like calls the trainer uses
(compile / train_on_batch / predict / save_weights / load_weights).

A training step (reference: one session.run of forward, YOLOv3Loss.loss, tf.gradients, RAdam.get_updates, trainer.py:84,113)
is a fixed sequence of kernel launches on one HIP stream, captured once into a hipGraph and replayed:
    pack input -> forward -> loss fwd+bwd -> backward -> [RCCL all-reduce of the flat gradient] -> RAdam+L2 -> weight repack
"""
import os
import numpy as np
import torch
from . import engine, ops


class YOLOv3Model(object):
    def __init__(self, detector, input_image_size, head_channel_nums, head_names, batch_size=None, device=None, seed=800):
        from . import configs
        if not torch.cuda.is_available():
            raise RuntimeError('the MI355X-native path needs a GPU (no CPU fallback)')
        self.name = detector.backbone_name
        self.batch_size = int(batch_size if batch_size is not None else configs.FLAGS.batch_size)
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.input_image_size = list(input_image_size)
        self.head_channel_nums = list(head_channel_nums)
        self.head_names = list(head_names)
        H, W, C = self.input_image_size
        if H % 32 or W % 32:
            raise ValueError('input height/width must be multiples of 32')
        with torch.cuda.device(self.device):
            g = engine.Graph(self.batch_size, self.device, seed)
            x = g.input(H, W, C)
            heads = detector._detection_head(detector.backbone.build(x), self.head_channel_nums, self.head_names)
            g.finalize(heads)
            g.refresh_dgrad_weights()
            # a time-out of the single-launch BatchNorm backward's grid barrier (its cooperative grid was not resident) raises this pinned
            # word; run_step looks at it before it enqueues anything: the failure is loud at the very next step, not at the epoch's end
            self._bn_flag = torch.zeros(1, dtype=torch.int32)
            try:
                self._bn_flag = self._bn_flag.pin_memory()
            except RuntimeError:                           # no device behind torch.cuda (the CPU suite builds graphs with mocked kernels)
                pass
            if self._bn_flag.is_pinned():
                ops.bn_fused_set_host_flag(g.bn_sync, self._bn_flag)
        self.g = g
        self.heads = heads                       # engine.Val (kind 'conv', float32, channel-padded), order /8, /16, /32
        self.ldc = [h.shape[3] for h in heads]
        self.loss_obj = None
        self.optimizer = None
        self._graphs = None
        self.world_size, self.rank = 1, 0
        self.process_group = None
        # Execution mode.  Measured on MI355X (ResNet18 416^2 batch 32): eager launches with the weight-gradient GEMMs on a second
        # stream take 6.6 ms/step; hipGraph replay serialises the forked branch and takes 7.7 ms (the same as one stream), so the default is
        # eager + overlap; the host stays ahead of the GPU (~270 launches per step).
        self.use_hip_graph = False
        self.overlap_wgrad = True          # weight-gradient GEMMs on a second stream (see engine.Graph.run_backward)
        self.overlap_allreduce = True      # data parallel: each stage's gradient bucket is all-reduced while the earlier layers still run backward
        self.bucket_updates = True         # eager mode: RAdam + L2 per gradient bucket, overlapped with the rest of the backward pass
        # eager mode: after two plain steps the step's launch list (kernels + cross-stream edges) is recorded once by the native library
        # and re-issued with one call per step (yolo_seq_run): host cost per step ~2.8 ms of Python + ctypes -> the bare HIP launches
        self.native_sequencer = True
        self._seq, self._recording, self._eager_steps = None, None, 0
        self._comm_stream = None
        self._pending = []
        self._step_ranges = []
        self.loss_value = torch.zeros(1, device=self.device)
        self.l2_value = torch.zeros(1, device=self.device)

    # ---------------------------------------------------------------------------------------------- keras-like surface
    def summary(self):
        ps = self.g.ps
        n = sum(int(np.prod(p.tf_shape)) for p in ps.params.values())
        print('%s: %d trainable parameters in %d variables, %d kernels launches per forward' % (self.name, n, len(ps.params), len(self.g.fwd)))

    def count_params(self):
        return sum(int(np.prod(p.tf_shape)) for p in self.g.ps.params.values())

    def compile(self, optimizer, loss):
        """optimizer: utils.radam.RAdam; loss: the bound ``YOLOv3Loss.loss`` (reference trainer.py:84) or the object"""
        self.optimizer = optimizer
        self.loss_obj = getattr(loss, '__self__', loss)
        self.loss_obj.bind(self)
        optimizer.bind(self)
        self._graphs = None

    def set_distributed(self, world_size, rank, process_group=None):
        self.world_size, self.rank, self.process_group = world_size, rank, process_group
        self._graphs = None
        if world_size > 1:
            # the single-launch BatchNorm backward needs its whole grid resident (one 1024-thread, ~120-VGPR workgroup on every CU); an
            # RCCL kernel of the overlapped gradient all-reduce holds registers on some CUs for the length of the collective, so the
            # grid barrier would wait for it: data-parallel runs keep the three-kernel path
            self.g.fused_bn_bwd = False

    # ---------------------------------------------------------------------------------------------- step
    def _fwd_bwd(self):
        g = self.g
        if self.overlap_wgrad and g.wgrad_stream is None:
            g.wgrad_stream = torch.cuda.Stream(device=self.device)
            g.use_side_stream(g.wgrad_stream)
            for owner in (self.loss_obj, self.optimizer, self):      # the per-bucket optimizer launches run on this stream too
                for t in (getattr(owner, '__dict__', {}) or {}).values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(g.wgrad_stream)
        elif not self.overlap_wgrad:
            g.wgrad_stream = None
        g.run_forward()
        self.loss_obj.launch(self)
        g.run_backward()

    def _update(self):
        g = self.g
        self.optimizer.launch(self)
        if self.use_hip_graph:
            g.refresh_dgrad_weights()
        else:
            g.refresh_dgrad_async()

    def _capture(self):
        if not self.use_hip_graph:
            self._graphs = (None, None)
            return
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.g.capturing = True
        try:
            with torch.cuda.stream(s):
                with torch.cuda.graph(ga, stream=s):
                    self._fwd_bwd()
                with torch.cuda.graph(gb, stream=s):
                    self._update()
        finally:
            self.g.capturing = False
        torch.cuda.current_stream(self.device).wait_stream(s)
        self._graphs = (ga, gb)

    def stage_batch(self, images, labels):
        """host -> device copy of one batch into the static input buffers (outside the timed region of bench.py)"""
        g = self.g
        img = torch.as_tensor(np.asarray(images, dtype=np.float32) if not torch.is_tensor(images) else images)
        lab = torch.as_tensor(np.asarray(labels, dtype=np.float32) if not torch.is_tensor(labels) else labels)
        if tuple(img.shape) != tuple(g.images.shape):
            raise ValueError('images must have shape %s (static graph), got %s' % (tuple(g.images.shape), tuple(img.shape)))
        g.images.copy_(img.to(torch.float32), non_blocking=True)
        self.loss_obj.stage_labels(lab)

    def _allreduce_bucket(self, lo, hi, then_update=False, first=False):
        """enqueue the all-reduce (SUM) of grad[lo:hi] on the communication stream once everything enqueued so far on the main and
        weight-gradient streams has finished -- and, with ``then_update``, the optimizer launch of that range right behind it on the same
        stream (the bucket is updated while the rest of the backward pass still runs); otherwise the optimizer waits for self._pending"""
        import torch.distributed as dist
        g = self.g
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.device)
            for t in g.owned_tensors() + [t for t in vars(self.optimizer).values() if isinstance(t, torch.Tensor) and t.is_cuda]:
                t.record_stream(self._comm_stream)
        cs = self._comm_stream
        g.stream_wait(cs, torch.cuda.current_stream(self.device))
        if g.wgrad_stream is not None:
            g.stream_wait(cs, g.wgrad_stream)

        def collective():                      # host work between launches: a segment boundary of a recorded launch sequence
            with torch.cuda.stream(cs):
                work = dist.all_reduce(g.ps.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
                if then_update:
                    work.wait()                # the communication stream waits for the collective
                else:
                    self._pending.append(work)

        if self._recording is not None:
            self._recording.append((ops.seq_mark(), collective))
        collective()
        if then_update:
            with torch.cuda.stream(cs):
                self.optimizer.launch_range(self, lo, hi, first)

    def _bucket_ready(self, lo, hi, on_main=False):
        """engine callback (eager mode): the gradient range [lo, hi) is complete behind what is queued on the main and weight-gradient
        streams.  Data parallel: all-reduce it, then update it, on the communication stream; single GPU: update it on the weight-gradient
        stream.  Either way the bucket's RAdam + L2 launch overlaps the rest of the backward pass (reference north_star: all-reduce of
        gradients overlapped with the RAdam update); only the last, ~1 % bucket (stem + stride-4 stage) is exposed."""
        first, self._step_ranges = not self._step_ranges, self._step_ranges + [(lo, hi)]
        if self.world_size > 1:
            self._allreduce_bucket(lo, hi, then_update=True, first=first)
            return
        side = self.g.wgrad_stream
        if side is None or on_main:                 # (on_main: the step's last range, already behind the weight-gradient stream: engine.bucket_done)
            self.optimizer.launch_range(self, lo, hi, first)
        else:
            with torch.cuda.stream(side):
                self.optimizer.launch_range(self, lo, hi, first)

    def run_step(self):
        """one training step on the staged batch; returns nothing (loss stays on the device in self.loss_value)"""
        if self.loss_obj is None or self.optimizer is None:
            raise RuntimeError('compile(optimizer, loss) first')
        g = self.g
        if int(self._bn_flag[0]) != 0:
            raise RuntimeError('a grid barrier of yolo_bn_act_bwd_fused timed out in an earlier step (its cooperative grid was not resident: '
                               'another process on the GPU?): results since then are invalid; rerun with g.fused_bn_bwd = False')
        with torch.cuda.device(self.device):
            if self._graphs is None:
                g.training = True
                self._drop_sequence()              # buffers / configuration changed (compile, set_distributed, label slots grew)
                self._eager_steps = 0
                self._capture()
            ga, gb = self._graphs
            dp = self.world_size > 1
            if ga is None and self.bucket_updates and (self.overlap_allreduce or not dp):
                # eager: every gradient bucket is (all-reduced and) updated as soon as it is complete, beside the rest of the backward pass
                sig = self._step_signature()
                if self._seq is not None and self._seq[1] == sig:
                    self._replay()                                   # the recorded launch list, one native call per segment
                    return
                self._drop_sequence()
                self._eager_steps += 1
                record = self.native_sequencer and self._eager_steps > 2      # the first steps run plain: one-time set-up happens there
                if record:
                    seq_id = ops.seq_begin()
                    self._recording = []
                try:
                    self._step_ranges = []
                    g.on_bucket = self._bucket_ready
                    tail = g.tail_on_main = not dp and g.wgrad_stream is not None and os.environ.get('YOLO_TAIL_ON_MAIN', '1') != '0'
                    try:
                        self._fwd_bwd()
                    finally:
                        g.on_bucket = None
                        g.tail_on_main = False
                    main = torch.cuda.current_stream(self.device)
                    probe = getattr(self, '_comm_probe', None)
                    if probe is not None:                            # measure_exposed_comm_ms: main-stream time spent waiting in the joins
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record(main)
                    if self._comm_stream is not None and dp:
                        g.stream_wait(main, self._comm_stream)
                    if g.wgrad_stream is not None and not tail:     # (tail: the main stream joined the side stream before the last range)
                        g.stream_wait(main, g.wgrad_stream, local=True)
                    if probe is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record(main)
                        probe.append((e0, e1))
                    covered = sum(hi - lo for lo, hi in self._step_ranges)
                    if covered != g.ps.n:
                        raise RuntimeError('gradient buckets cover %d of %d parameters' % (covered, g.ps.n))
                    self.optimizer.finish(self)
                    g.refresh_dgrad_async()
                except BaseException:
                    if record:                      # a step that raised part-way must never be replayed: close the recording and drop it
                        ops.seq_end()
                        ops.seq_free(seq_id)
                        self._recording = None
                    raise
                if record:                          # only a step that completed becomes the replayed launch list
                    n = ops.seq_end()
                    segments, self._recording = self._recording, None
                    self._seq = (seq_id, sig, segments + [(n, None)])
                return
            if ga is None:
                self._fwd_bwd()
            else:
                ga.replay()
            if dp:
                self._allreduce_bucket(0, g.ps.n)
            for work in self._pending:
                work.wait()                    # the current stream waits for the collective
            self._pending = []
            if gb is None:
                self._update()
            else:
                gb.replay()

    def measure_exposed_comm_ms(self, steps=3):
        """milliseconds per step the main stream waits, after its own last kernel, for the communication (and weight-gradient) stream:
        the part of the gradient exchange + per-bucket updates that the backward pass did NOT hide.  Runs ``steps`` eagerly enqueued
        steps (collective on every rank)."""
        saved = self.native_sequencer
        self.native_sequencer = False
        self._drop_sequence()
        self._comm_probe = []
        try:
            for _ in range(steps + 1):
                self.run_step()
            torch.cuda.synchronize(self.device)
            spans = [a.elapsed_time(b) for a, b in self._comm_probe[1:]]
        finally:
            self._comm_probe = None
            self.native_sequencer = saved
        return round(sum(spans) / max(len(spans), 1), 4) if spans else None

    # ---------------------------------------------------------------------------------------------- native launch sequencer
    def _step_signature(self):
        """everything a recorded step depends on besides the (static) buffers: a change re-records"""
        from . import backend
        g = self.g
        return (self.overlap_wgrad, self.bucket_updates, self.overlap_allreduce, g.fused_bn_bwd, g.wgrad_batch, g.wgrad_cost_limit, self.world_size,
                backend.loss_scale(), id(self.loss_obj), id(self.optimizer), int(self.loss_obj.T), g.training, g.bn_momentum,
                torch.cuda.current_stream(self.device).cuda_stream, ops.tuning_epoch())

    def _replay(self):
        seq_id, _, segments = self._seq
        at = 0
        for end, host in segments:
            ops.seq_run(seq_id, at, end)
            if host is not None:
                host()
            at = end
        self.g._repack_event = True

    def _drop_sequence(self):
        if self._seq is not None:
            ops.seq_free(self._seq[0])
            self._seq = None

    def check_device_protocols(self):
        """raise if a bounded device-side wait expired (the single-launch BatchNorm backward's grid barrier): results since the last
        check cannot be trusted.  Synchronises the device: call it per epoch / at the end of a run, not per step."""
        sync = getattr(self.g, 'bn_sync', None)
        if sync is not None:
            n = ops.bn_fused_timeouts(sync)
            if n:
                raise RuntimeError('%d grid-barrier time-outs in yolo_bn_act_bwd_fused: the cooperative grid was not resident '
                                   '(another process on the GPU?); rerun with fused_bn_bwd = False' % n)
        nf = getattr(self.optimizer, 'nonfinite', None)
        if nf is not None:
            n = int(nf.item())
            if n:
                nf.zero_()
                from . import backend
                hint = (' (float16: lower the loss scale, backend.set_loss_scale(%g))' % (backend.loss_scale() / 4)
                        if backend.compute_dtype() == 'float16' else '')
                raise FloatingPointError('non-finite gradient elements reached the optimizer in %d waves since the last check; they were '
                                         'not applied%s' % (n, hint))

    def train_on_batch(self, images, labels, sync=True):
        """keras Model.train_on_batch: one step, returns the loss.  ``sync=False`` returns it as a 0-d DEVICE tensor instead of a float: the host
        does not wait for the step, so decoding / uploading / enqueueing the next batch overlaps it (the trainer reads the epoch's losses once,
        at the epoch's end; a float per step costs a device synchronisation per step, ~25 % of a 4 ms step when batches come from files)"""
        self.stage_batch(images, labels)
        self.run_step()
        if not sync:
            return self.loss_value.detach().clone()
        return float(self.loss_value.item())

    def test_on_batch(self, images, labels):
        """keras Model.test_on_batch as ``fit(validation_data=...)`` uses it (reference trainer.py:107-110): loss of one batch, nothing
        updated.  The reference trains with the learning phase forced to 1 (run.py:21-22), so its validation pass also normalises with BATCH
        statistics; the moving averages are only touched by the training function (momentum 1 here leaves them bit-identical).  What the
        reference's loss graph does update even here is the rectified-loss image counter (yolov3_loss.py:151-152 sits inside the loss)."""
        if self.loss_obj is None:
            raise RuntimeError('compile(optimizer, loss) first')
        g = self.g
        self.stage_batch(images, labels)
        with torch.cuda.device(self.device):
            prev = (g.training, g.bn_momentum)
            g.training, g.bn_momentum = True, 1.0
            try:
                g.run_forward()
                self.loss_obj.launch(self)
            finally:
                g.training, g.bn_momentum = prev
            ps = g.ps
            l2 = (ps.flat.view(-1, engine.SLOT).double().pow(2).sum(dim=1) * ps.l2_table.double()).sum()
            return float(self.loss_obj.total.double().item() + l2.item())

    def forward_only(self, images, training=False):
        """run the forward kernels (no capture); returns the three head tensors (device, float32, padded channels)"""
        g = self.g
        with torch.cuda.device(self.device):
            g.images.copy_(torch.as_tensor(np.asarray(images, dtype=np.float32) if not torch.is_tensor(images) else images))
            prev = g.training
            g.training = training
            try:
                g.run_forward()
            finally:
                g.training = prev
        return [h.buf for h in self.heads]

    def predict(self, test_images):
        """reference trainer.py:117-124 / keras Model.predict: float32 (n,H,W,3) in [0,1] BGR -> float32 ndarray
        (n, H/32, W/32, C8*16 + C16*4 + C32) in the reference's merged layout (yolov3_detector.py:80-85)"""
        x = np.asarray(test_images, dtype=np.float32)
        n, N = x.shape[0], self.batch_size
        outs = []
        for i in range(0, n, N):
            chunk = x[i:i + N]
            if chunk.shape[0] < N:
                chunk = np.concatenate([chunk, np.zeros((N - chunk.shape[0],) + chunk.shape[1:], np.float32)], axis=0)
            heads = self.forward_only(chunk, training=False)
            outs.append(self.merge_heads(heads)[:min(N, n - i)])
        return np.concatenate(outs, axis=0)

    def merge_heads(self, heads):
        H32, W32 = self.input_image_size[0] // 32, self.input_image_size[1] // 32
        parts = []
        for t, c in zip(heads, self.head_channel_nums):
            parts.append(t[..., :c].reshape(t.shape[0], H32, W32, -1))
        return torch.cat(parts, dim=-1).float().cpu().numpy()

    # ---------------------------------------------------------------------------------------------- observability (per epoch)
    def regularization_losses(self):
        """(sum of the BN-gamma L2 terms, their count, sum of the kernel L2 terms, their count): what DetailLossLogger prints
        (reference utils/logger_callback.py:49-58,105-108; each term is l2 * sum(w^2), basic_backbone.py:41,64,76)"""
        ps = self.g.ps
        sums = {'bn_gamma': 0.0, 'kernel': 0.0}
        counts = {'bn_gamma': 0, 'kernel': 0}
        flat = ps.flat.detach()
        for p in ps.params.values():
            if p.l2 > 0:
                key = 'bn_gamma' if p.kind == 'bn_gamma' else 'kernel'
                counts[key] += 1
                sums[key] += p.l2 * float((flat[p.offset:p.offset + p.numel].double() ** 2).sum().item())
        return sums['bn_gamma'], counts['bn_gamma'], sums['kernel'], counts['kernel']

    def bn_gammas(self):
        """all BatchNorm gammas concatenated (reference utils/board_callback.py:73-81) -> float32 ndarray"""
        ps = self.g.ps
        flat = ps.flat.detach()
        parts = [flat[p.offset:p.offset + p.numel] for p in ps.params.values() if p.kind == 'bn_gamma']
        return torch.cat(parts).cpu().numpy()

    # ---------------------------------------------------------------------------------------------- weights by Keras name
    def get_weights(self):
        """{keras variable name: float32 ndarray in TF layout (HWIO kernels)}, trainable + moving statistics"""
        ps = self.g.ps
        flat = ps.flat.detach().cpu()
        out = {}
        for p in ps.params.values():
            t = flat[p.offset:p.offset + p.numel]
            if p.kind in ('conv_kernel', 'head_kernel'):
                out[p.name] = engine.Graph.kernel_from_dev(t, p).numpy()
            elif p.kind == 'dw_kernel':
                out[p.name] = engine.Graph.dw_from_dev(t, p).numpy()
            else:
                out[p.name] = t[:p.tf_shape[0]].clone().numpy()
        for bn in self.g.bns:
            out[bn.name + '/moving_mean'] = bn.moving_mean.cpu().numpy()
            out[bn.name + '/moving_variance'] = bn.moving_var.cpu().numpy()
        return out

    def set_weights(self, weights):
        ps = self.g.ps
        flat = ps.flat.detach().cpu()
        for p in ps.params.values():
            if p.name not in weights:
                raise KeyError('missing variable ' + p.name)
            w = torch.as_tensor(np.asarray(weights[p.name], dtype=np.float32))
            if tuple(w.shape) != p.tf_shape:
                raise ValueError('%s: expected shape %s, got %s' % (p.name, p.tf_shape, tuple(w.shape)))
            if p.kind in ('conv_kernel', 'head_kernel'):
                t = engine.Graph.kernel_to_dev(w, p)
            elif p.kind == 'dw_kernel':
                t = engine.Graph.dw_to_dev(w, p)
            else:
                t = torch.zeros(p.dev_shape)
                t[:w.shape[0]] = w
            flat[p.offset:p.offset + p.numel] = t.reshape(-1)
        with torch.cuda.device(self.device):
            ps.flat.copy_(flat)
            ops.cast_f32_to_bf16(ps.flat, ps.bf16, ps.n)
            for bn in self.g.bns:
                if bn.name + '/moving_mean' in weights:
                    bn.moving_mean.copy_(torch.as_tensor(np.asarray(weights[bn.name + '/moving_mean'], dtype=np.float32)))
                    bn.moving_var.copy_(torch.as_tensor(np.asarray(weights[bn.name + '/moving_variance'], dtype=np.float32)))
            self.g.refresh_dgrad_weights()
            torch.cuda.synchronize(self.device)

    def save_weights(self, path, full_state=False, epoch=None):
        """weights-only checkpoint under the reference's file stem (trainer.py:90-91: ``model.save_weights('...ckpt')``) in TensorFlow's
        checkpoint byte format -- ``<path>.index`` (SSTable) + ``<path>.data-00000-of-00001`` with Keras' object-based keys and object graph,
        and the ``checkpoint`` state file naming the latest stem (utils/tf_checkpoint.py; parity unpinned against TensorFlow itself).
        ``full_state`` (opt-in; the reference restores weights only, so the RAdam moments, its step counter, the rectified-loss image
        counter and the epoch -- hence the learning-rate schedule -- restart) additionally writes
        ``<path>.state.npz`` with exactly those, in the flat device layout of this build."""
        from .utils import tf_checkpoint
        d = os.path.dirname(path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        tf_checkpoint.write_checkpoint(path, self.get_weights())
        if full_state:
            ps, opt = self.g.ps, self.optimizer
            state = {'layout_n': np.int64(ps.n), 'm': ps.m.detach().cpu().numpy(), 'v': ps.v.detach().cpu().numpy(),
                     'master': ps.flat.detach().cpu().numpy(),      # float32 master weights in device layout: the resume is bit-exact
                     'iterations': np.int64(opt.iterations if opt is not None else 0), 'epoch': np.int64(-1 if epoch is None else epoch),
                     'current_num': np.int64(int(self.loss_obj.current_num.item()) if self.loss_obj is not None else 0)}
            if opt is not None and opt.vhat is not None:
                state['vhat'] = opt.vhat.detach().cpu().numpy()
            np.savez(path + '.state.npz', **state)
        tf_checkpoint.update_checkpoint_state(d, os.path.basename(path))

    def load_weights(self, path, full_state=False):
        """-> the epoch stored with a full-state checkpoint (or None)"""
        from .utils import tf_checkpoint
        stem = path
        for suffix in ('.npz', '.index'):
            if stem.endswith(suffix):
                stem = stem[:-len(suffix)]
        if tf_checkpoint.exists(stem):                     # TensorFlow checkpoint: ours, or one written by the reference through TensorFlow
            self.set_weights(tf_checkpoint.read_checkpoint(stem))
        else:                                              # round-1 checkpoints of this package (variables by Keras name in an .npz)
            with np.load(stem + '.npz', allow_pickle=False) as z:
                self.set_weights({k: z[k] for k in z.files})
        if not full_state or not os.path.exists(stem + '.state.npz'):
            return None
        if self.optimizer is None or self.loss_obj is None:
            raise RuntimeError('compile(optimizer, loss) before restoring a full-state checkpoint')
        with np.load(stem + '.state.npz', allow_pickle=False) as z:
            ps, opt = self.g.ps, self.optimizer
            if int(z['layout_n']) != ps.n:
                raise ValueError('optimizer state was saved for a different model layout (%d slots, this model has %d)' % (int(z['layout_n']), ps.n))
            with torch.cuda.device(self.device):
                ps.m.copy_(torch.as_tensor(z['m']))
                ps.v.copy_(torch.as_tensor(z['v']))
                ps.flat.copy_(torch.as_tensor(z['master']))
                ops.cast_f32_to_bf16(ps.flat, ps.bf16, ps.n)
                self.g.refresh_dgrad_weights()
                opt._iterations.fill_(int(z['iterations']))
                if 'vhat' in z.files and opt.vhat is not None:
                    opt.vhat.copy_(torch.as_tensor(z['vhat']))
                self.loss_obj.current_num.fill_(int(z['current_num']))
                torch.cuda.synchronize(self.device)
            epoch = int(z['epoch'])
        return epoch if epoch >= 0 else None


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint semantics (reference trainer.py:60): the stem named by the ``checkpoint`` state file, if its files exist"""
    from .utils import tf_checkpoint
    full = tf_checkpoint.latest_checkpoint(directory)
    if full is None:
        return None
    return full if (tf_checkpoint.exists(full) or os.path.exists(full + '.npz')) else None
