#!/usr/bin/env python
"""bench.py -- training throughput of the MI355X-native YOLOv3 hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full training step (pack input, forward, YOLOv3 loss fwd+bwd, backward with the weight-gradient GEMMs on a second
stream, [bucketed RCCL gradient all-reduce on a third], RAdam+L2 update, weight repack) on one synthetic COCO-shaped batch already resident in HBM.  Workload = BASELINE.json configs[1]:
ResNet18-YOLOv3 416x416 bf16, 80 classes, batch 32 per GPU (weak scaling: global batch = 32 * N = configs[2] at N = 8).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# COCO-9 anchors normalised by 416 (w, h), 3/3/3 per head
COCO_ANCHORS = [[(10 / 416., 13 / 416.), (16 / 416., 30 / 416.), (33 / 416., 23 / 416.)],
                [(30 / 416., 61 / 416.), (62 / 416., 45 / 416.), (59 / 416., 119 / 416.)],
                [(116 / 416., 90 / 416.), (156 / 416., 198 / 416.), (373 / 416., 326 / 416.)]]
LOSS_WEIGHTS = [(5, 5, 0.05, 3, 1), (8, 8, 0.05, 2, 1), (10, 10, 0.05, 2, 1)]      # reference configs.py:52
HEAD_NAMES = ['yolov3_head_8', 'yolov3_head_16', 'yolov3_head_32']
TRAIN_GFLOP_PER_IMAGE = 51.727     # ResNet18 416x416, 80 classes, anchors 3/3/3, fwd + dgrad + wgrad
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md chip-level parameters)


def synthetic_batch(N, H, W, class_num, rank, T=8):
    """images U[0,1) seed 800 (+rank); labels n~U{1..8}, cx,cy~U(.05,.95), w,h~U(.03,.6) clipped, seed 6"""
    g = torch.Generator().manual_seed(800 + rank)
    images = torch.rand(N, H, W, 3, generator=g)
    rng = np.random.RandomState(6 + rank)
    labels = -np.ones((N, T, 5), dtype=np.float32)
    for n in range(N):
        k = rng.randint(1, T + 1)
        cx, cy = rng.uniform(0.05, 0.95, k), rng.uniform(0.05, 0.95, k)
        w = np.minimum(rng.uniform(0.03, 0.6, k), 2 * np.minimum(cx, 1 - cx))
        h = np.minimum(rng.uniform(0.03, 0.6, k), 2 * np.minimum(cy, 1 - cy))
        labels[n, :k] = np.stack([cx, cy, w, h, rng.randint(0, class_num, k)], axis=1)
    return images, torch.from_numpy(labels.reshape(N, T * 5))


def build_model(backbone, H, W, N, class_num, device, focal=False):
    from yolov3_tensorflow_amd.yolov3.yolov3_detector import YOLOv3Detector
    from yolov3_tensorflow_amd.yolov3.yolov3_loss import YOLOv3Loss
    from yolov3_tensorflow_amd.utils.radam import RAdam
    L = 5 + class_num
    chans = [len(a) * L for a in COCO_ANCHORS]
    grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    model = YOLOv3Detector(backbone).build((H, W, 3), chans, HEAD_NAMES, batch_size=N, device=device)
    loss = YOLOv3Loss(grids, class_num, COCO_ANCHORS, 0.8, LOSS_WEIGHTS, rectified_coord_num=-1, rectified_loss_weight=[1.0, 1.0, 1.0],
                      is_focal_loss=focal, focal_alpha=1.0, focal_gamma=2.0)     # the reference's FLAGS values (configs.py:69-70)
    opt = RAdam(lr=1e-3)
    model.compile(optimizer=opt, loss=loss.loss)
    return model, loss, opt, grids


HBM_PEAK_GBS = 8000.0              # HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming kernel sustains on it
HBM_ACHIEVABLE_GBS = 6300.0


def _big_tensor_bytes(args, kwargs, floor):
    """operand bytes of a bandwidth-bound launch: every tensor argument of at least ``floor`` bytes counted once (its read or its write)"""
    seen, total = set(), 0
    for t in list(args) + list(kwargs.values()):
        if isinstance(t, (list, tuple)):
            ts = t
        else:
            ts = (t,)
        for u in ts:
            if isinstance(u, torch.Tensor) and u.data_ptr() not in seen and u.numel() * u.element_size() >= floor:
                seen.add(u.data_ptr())
                total += u.numel() * u.element_size()
    return total


def kernel_rooflines(model, steps, overlap):
    """HIP-event timing (on the launch stream) of the step's kernel families in eager passes:
      strip / other : conv3x3_strip_kernel (every 3x3 stride-1 conv forward + data gradient: the dominant kernel) and igemm_fwd_kernel (stem,
                      stride-2, 1x1, fused concat) -- algorithmic FLOPs / duration
      bn_fwd / bn_bwd / loss / optimizer : the bandwidth-bound families -- bytes of their full-size operands / duration
    ``overlap`` False: the weight-gradient stream is folded into the main stream, every launch is timed alone (the kernel's own speed).
    ``overlap`` True: the real two-stream schedule -- a launch's interval then also contains what the concurrent weight-gradient kernels take
    from it (the in-step figure).  Collective-free, so data-parallel runs call it on every rank."""
    from yolov3_tensorflow_amd import ops
    records = []

    def is_strip(p):
        return (p.R == 3 and p.S == 3 and p.stride == 1 and p.pad_t == 1 and p.pad_l == 1 and p.C0 == 0 and p.Ho == p.H and p.Wo == p.W
                and p.Cin % 64 == 0 and p.Cout % 64 == 0)

    def timed_conv(fn):
        def wrapper(p, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(p, *a, **k)
            e1.record()
            cin = 3 if p.Cin == 8 else p.Cin     # algorithmic: the RGB stem is 3 of the 8 padded channels
            fam = 'strip' if (is_strip(p) and k.get('bias') is None) else 'other'
            if fam == 'strip' and fn is saved_fwd and ops.conv2d_fwd_plan(p)['family'] == 'stream':
                records.append((e0, e1, 2.0 * p.N * p.Ho * p.Wo * p.Cout * cin * p.R * p.S, 'stream'))     # (also counted in 'strip' below)
            if k.get('bn') is not None:      # data gradient carrying a BatchNorm unit's masking + backward reduce in its epilogue (another instantiation)
                fam += '_bn'
            records.append((e0, e1, 2.0 * p.N * p.Ho * p.Wo * p.Cout * cin * p.R * p.S, fam))
        return wrapper

    def timed_bytes(fn, family, nbytes=None):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            records.append((e0, e1, float(nbytes(a, k) if nbytes else _big_tensor_bytes(a, k, 1 << 16)), family))
            return r
        return wrapper

    n_params = model.g.ps.n
    saved_fwd = ops.conv2d_fwd
    patched = {'conv2d_fwd': timed_conv(ops.conv2d_fwd), 'conv2d_dgrad': timed_conv(ops.conv2d_dgrad)}

    def timed_wgrad(fn):                  # the slab pass of the weight gradient: 2 N Ho Wo Cout Cin R S FLOP as well (timed on ITS launch stream)
        def wrapper(p, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(p, *a, **k)
            e1.record()
            cin = 3 if p.Cin == 8 else p.Cin
            records.append((e0, e1, 2.0 * p.N * p.Ho * p.Wo * p.Cout * cin * p.R * p.S, 'wgrad3x3' if is_strip(p) else 'wgrad_other'))
        return wrapper
    patched['conv2d_wgrad_slabs'] = timed_wgrad(ops.conv2d_wgrad_slabs)
    patched['wgrad_reduce_batched'] = timed_bytes(ops.wgrad_reduce_batched, 'wgrad_sum')
    for name in ('bn_act_fwd', 'bn_pool_fwd'):
        patched[name] = timed_bytes(getattr(ops, name), 'bn_fwd')
    for name in ('bn_act_bwd_fused', 'bn_act_bwd_reduce', 'bn_act_bwd_apply', 'bn_pool_bwd_reduce', 'bn_pool_bwd_apply', 'stem_pool_bwd_wgrad'):
        patched[name] = timed_bytes(getattr(ops, name), 'bn_bwd')
    patched['loss_fwd_bwd'] = timed_bytes(ops.loss_fwd_bwd, 'loss')
    # p, g, m, v read + p, m, v written + the 16-bit copy = 30 B / parameter (SURVEY 8d counts 28 without the copy; the gradient zeroing adds 4 more)
    patched['radam_l2_step'] = timed_bytes(ops.radam_l2_step, 'optimizer', nbytes=lambda a, k: 30.0 * a[5])
    saved = {n: getattr(ops, n) for n in patched}
    saved_overlap, saved_bucket = model.overlap_wgrad, model.g.on_bucket
    for n, f in patched.items():
        setattr(ops, n, f)
    model.overlap_wgrad = bool(overlap)
    model.g.on_bucket = None           # purely local passes: no gradient all-reduce is issued
    try:
        for _ in range(steps):
            model._fwd_bwd()
            model._update()
        torch.cuda.synchronize()
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        model.overlap_wgrad, model.g.on_bucket = saved_overlap, saved_bucket
    out = {}
    for family in ('strip', 'stream', 'strip_bn', 'other', 'other_bn', 'wgrad3x3', 'wgrad_other', 'wgrad_sum', 'bn_fwd', 'bn_bwd', 'loss', 'optimizer'):
        rs = [r for r in records if r[3] == family]
        t_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rs)
        work = sum(r[2] for r in rs)
        out[family] = {'rate': work / (t_ms * 1e-3) if rs and t_ms > 0 else 0.0, 'avg_launch_ms': t_ms / max(len(rs), 1),
                       'launches_per_step': len(rs) // max(steps, 1), 'ms_per_step': t_ms / max(steps, 1), 'work_per_step': work / max(steps, 1)}
    return out


def strip_hbm_traffic():
    """HBM bytes per launch of the strip kernel: bench.py cannot run the rocprofv3 --pmc passes itself, so this is the launch-weighted
    mean over the kernel's tile variants from the newest committed PMC summary of this same command (tools/hbm_traffic.py); the step's
    34 launches move ~1.46 GB of activations + weights algorithmically (~43 MB per launch)"""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pmc_hbm_traffic.json')))
    for path in reversed(files):
        try:
            rows = [k for k in json.load(open(path))['kernels'] if k['kernel'].startswith(('conv3x3_strip_kernel', 'conv3x3_stream_kernel', 'conv3x3_s32_kernel'))]
            n = sum(k['launches'] for k in rows)
            if n:
                b = sum((k['fetch_corrected_KB_per_launch'] + k['WRITE_SIZE_KB_per_launch']) * 1024.0 * k['launches'] for k in rows) / n
                return int(b), 'from profiles/' + os.path.basename(path)
        except (KeyError, ValueError, OSError):
            continue
    return None, 'no PMC summary committed'


def parity_statement(dtype):
    """which build holds north_star's 1e-3 on the loss curve: the committed 20-step configs[0] curves against the float32 oracle
    (tools/loss_curve.py on the GPU box; newest profiles/r*_loss_curve_config1*.json).  The headline run itself is synthetic data."""
    import glob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
    out = {'tolerance': 1e-3, 'reference': 'float32 CPU oracle (restatement of the reference; parity with TensorFlow itself is unpinned: TF not installable here)',
           'workload': 'BASELINE.json configs[0]: ResNet18 320x320, the reference\'s 20 sample images, batch 2, 20 steps', 'headline_dtype': dtype}
    for key, pat in (('bf16', 'r*_loss_curve_config1.json'), ('fp16', 'r*_loss_curve_config1_fp16.json')):
        files = sorted(glob.glob(os.path.join(here, pat)))
        if not files:
            continue
        try:
            d = json.load(open(files[-1]))['float32_oracle']
            out[key] = {'max': round(d['max'], 6), 'median': round(d['median'], 6), 'within_1e-3': '%d/%d' % (d['steps_within_1e-3'], len(d['relative_deviation'])),
                        'source': 'profiles/' + os.path.basename(files[-1])}
            if d.get('assignment_differs'):       # steps on which the GPU run and the oracle assign a ground truth to different (head, anchor) pairs
                out[key]['assignment_differs_at_steps'] = d['assignment_differs']
                out[key]['max_on_the_other_steps'] = round(d['max_same_assignment'], 6)
        except (KeyError, ValueError, OSError):
            continue
    out['holds_1e-3_on_every_step'] = [k for k in ('bf16', 'fp16') if k in out and out[k]['within_1e-3'].split('/')[0] == out[k]['within_1e-3'].split('/')[1]]
    # how much of that is summation order: the same 20 steps under several kernel selections that only reorder float32 partial sums
    sc = sorted(glob.glob(os.path.join(here, 'r*_loss_curve_scatter.json')))
    if sc:
        try:
            runs = json.load(open(sc[-1]))['runs']
            out['summation_order_scatter'] = {'source': 'profiles/' + os.path.basename(sc[-1]), 'what': 'tools/loss_curve_scatter.py: the curve under '
                                              'kernel selections that differ only in the order of float32 partial sums (all pass the kernel parity tests)'}
            for key, name in (('bf16', 'bfloat16'), ('fp16', 'float16')):
                rs = [r for r in runs if r['dtype'] == name]
                if rs:
                    out['summation_order_scatter'][key] = {'orders': len(rs), 'max_range': [round(min(r['max'] for r in rs), 6), round(max(r['max'] for r in rs), 6)],
                                                           'median_range': [round(min(r['median'] for r in rs), 6), round(max(r['median'] for r in rs), 6)],
                                                           'within_1e-3_range': '%d-%d/20' % (min(r['steps_within_1e-3'] for r in rs), max(r['steps_within_1e-3'] for r in rs))}
        except (KeyError, ValueError, OSError):
            pass
    return out


def fp16_throughput(args, device, steps=20, warmup=5):
    """the float16 build (libyolov3_amd_fp16.so: the build whose loss curve sits closest to the float32 oracle) timed in THIS run on the headline
    configuration, so that one record states both builds' speed next to both builds' parity"""
    from yolov3_tensorflow_amd import backend, ops
    backend.set_compute_dtype('float16')
    try:
        for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):
            k, v = kv.split('=')
            ops.set_tuning(k, int(v))
        model, loss, opt, grids = build_model(args.backbone, args.size, args.size, args.batch, args.classes, device, focal=args.focal)
        images, labels = synthetic_batch(args.batch, args.size, args.size, args.classes, 0)
        model.stage_batch(images, labels)
        for _ in range(warmup):
            model.run_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.run_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        model.check_device_protocols()
        return {'images_per_sec': round(args.batch * steps / dt, 2), 'ms_per_step': round(dt / steps * 1e3, 4), 'steps': steps,
                'final_loss': round(float(model.loss_value.item()), 4)}
    finally:
        backend.set_compute_dtype(args.dtype)


def cu_mask(n, pattern, total=256):
    """8 x 32-bit words selecting n of the 256 CUs (the weight-gradient stream on CUs of its own)"""
    bits = [0] * total
    if pattern == 'low':
        idx = range(n)
    elif pattern == 'high':
        idx = range(total - n, total)
    elif pattern == 'even':
        idx = [i for i in range(total) if i % (total // n) == 0][:n]
    else:                                    # 'xcd': n / 8 low bits of every word
        per = n // 8
        idx = [w * 32 + b for w in range(8) for b in range(per)]
    for i in idx:
        bits[i] = 1
    return [sum(bits[w * 32 + b] << b for b in range(32)) for w in range(total // 32)]


def masked_stream(device, words):
    """torch stream around a HIP stream created with hipExtStreamCreateWithCUMask (ctypes on the HIP runtime torch has loaded)"""
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    if rc != 0 or not st.value:
        raise SystemExit('hipExtStreamCreateWithCUMask failed: %d' % rc)
    return torch.cuda.ExternalStream(st.value, device=device)


def rccl_statement(model, world, device):
    """what the process group looked like to this run (multi-GPU records must describe themselves): backend, world size seen by the group,
    the gradient buckets' bytes and the communication time that was NOT hidden behind the backward pass"""
    import torch.distributed as dist
    g = model.g
    bytes_per = [int((hi - lo) * 4) for (lo, hi) in g.bucket_ranges]      # completion order; the last one is the tail after the backward pass
    exposed = model.measure_exposed_comm_ms() if hasattr(model, 'measure_exposed_comm_ms') else None
    seen = [None] * world
    dist.all_gather_object(seen, {'rank': dist.get_rank(), 'device': str(device), 'name': torch.cuda.get_device_name(device)})
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks': seen, 'bucket_bytes_in_completion_order': bytes_per,
            'gradient_dtype': 'float32', 'exposed_comm_ms_per_step': exposed,
            'overlap': 'each bucket is all-reduced on a communication stream as soon as its last gradient is enqueued; its RAdam + L2 launch follows there'}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def _time_oracle(H, W, class_num, batch, anchors, budget_s, max_steps):
    from oracle.train import OracleTrainer
    grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    images, labels = synthetic_batch(batch, H, W, max(class_num, 1), 0)
    o = OracleTrainer('resnet-18', grids, class_num, anchors, 0.8, LOSS_WEIGHTS, rectified_coord_num=-1)
    o.ensure_params(images.numpy())
    o.step(images.numpy(), labels.numpy())        # warm-up step (allocations, thread pools)
    n_steps, t0 = 0, time.time()
    while n_steps < max_steps and (time.time() - t0 < budget_s or n_steps < 1):
        o.step(images.numpy(), labels.numpy())
        n_steps += 1
    return batch / ((time.time() - t0) / n_steps), n_steps


def cpu_baseline(H, W, class_num, budget_batch=4):
    """the CPU oracle (a restatement of the reference step on PyTorch-CPU, NOT TensorFlow) timed on this host's cores: the bench workload's
    shape at a small batch over a short sweep of thread counts (PyTorch's CPU convolutions do not scale to 128+ threads at this batch:
    ONE thread beat 128 on the GPU box's EPYC 9575F), the best of which is the reported value; plus ONE thread at batch 1, and
    BASELINE.json configs[0] (320x320, batch 2, 13 classes, the reference's 3/2/3 anchors)"""
    from yolov3_tensorflow_amd.configs import FLAGS
    all_threads = int(torch.get_num_threads())
    sweep = {}
    try:
        for threads in sorted({min(16, all_threads), min(64, all_threads), all_threads}):
            torch.set_num_threads(threads)
            ips, n_steps = _time_oracle(H, W, class_num, budget_batch, COCO_ANCHORS, 4.0, 6)
            sweep[threads] = (ips, n_steps)
        best = max(sweep, key=lambda t: sweep[t][0])
        torch.set_num_threads(best)
        c1_ips, c1_steps = _time_oracle(320, 320, 13, 2, FLAGS.anchor_boxes, 3.0, 8)
        torch.set_num_threads(1)
        st_ips, st_steps = _time_oracle(H, W, class_num, 1, COCO_ANCHORS, 4.0, 3)
    finally:
        torch.set_num_threads(all_threads)
    ips, n_steps = sweep[best]
    return {'value': round(ips, 3), 'unit': 'images/sec', 'cores': best, 'kind': 'port', 'cpu_model': _cpu_model(),
            'logical_cpus': os.cpu_count() or 0,
            'thread_sweep': {str(t): round(v[0], 3) for t, v in sorted(sweep.items())},
            'single_thread': {'value': round(st_ips, 4), 'unit': 'images/sec', 'cores': 1, 'sample': '%d timed step(s) of batch 1 at %dx%d' % (st_steps, H, W)},
            'config1': {'value': round(c1_ips, 3), 'unit': 'images/sec', 'cores': best,
                        'sample': '%d timed step(s) of BASELINE.json configs[0]: 320x320, batch 2, 13 classes, anchors 3/2/3' % c1_steps},
            'sample': '%d timed step(s) of batch %d at %dx%d, %d classes on %d threads (PyTorch-CPU float32 restatement of the reference step on %s; '
                      'host has %d logical cpus; best of the thread sweep)' % (n_steps, budget_batch, H, W, class_num, best, _cpu_model(), os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--classes', type=int, default=80)
    ap.add_argument('--backbone', default='resnet-18')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay two hipGraphs instead of eager two-stream launches (measured slower: the '
                    'forked weight-gradient branch is serialised under replay)')
    ap.add_argument('--no-overlap', action='store_true', help='weight-gradient GEMMs on the main stream')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'], help='16-bit compute type: bf16 (default) or fp16 (libyolov3_amd_fp16.so + '
                    'static loss scaling; BASELINE.json configs[4])')
    ap.add_argument('--focal', action='store_true', help='focal loss on (BASELINE.json configs[4])')
    ap.add_argument('--wgrad-batch', type=int, default=None, help='weight gradients per hand-off to the side stream (engine default 2)')
    ap.add_argument('--wgrad-gflop', type=float, default=None, help='also hand over when the pending weight gradients reach this many GFLOP')
    ap.add_argument('--main-priority', type=int, default=-1, help='priority of the stream the step runs on (-1 = high, the default: its kernels are the critical '
                    'path and win CUs from the concurrent weight-gradient stream, +1 %% measured; 0 = the default stream)')
    ap.add_argument('--side-priority', type=int, default=None, help='priority of the weight-gradient stream (A/B probe)')
    ap.add_argument('--side-cus', type=int, default=0, help='A/B probe: create the weight-gradient stream with hipExtStreamCreateWithCUMask over this many '
                    'CUs (0 = an ordinary stream)')
    ap.add_argument('--side-cu-pattern', default='low', choices=['low', 'high', 'even', 'xcd'], help='which CUs the mask selects: the first / last N bits, '
                    'every other bit, or N / 8 bits of every 32-bit word (one word per XCD if the enumeration is XCD-major)')
    ap.add_argument('--main-cus', type=int, default=0, help='A/B probe: run the main stream on a CU-masked stream over the COMPLEMENT of the side mask (1) or unmasked (0)')
    ap.add_argument('--no-sequencer', action='store_true', help='enqueue every launch from Python instead of replaying the recorded launch list')
    ap.add_argument('--no-bucket-updates', action='store_true', help='one RAdam + L2 launch after the backward pass instead of one per gradient bucket')
    ap.add_argument('--no-fused-bn', action='store_true', help='three-kernel BatchNorm backward instead of the single-launch one')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    from yolov3_tensorflow_amd import backend
    backend.set_compute_dtype(args.dtype)
    if os.environ.get('YOLO_BENCH_SHARE_GPU'):          # rehearsal aid: several ranks on one GPU (with YOLO_DIST_BACKEND=gloo)
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda:%d' % local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get('YOLO_DIST_BACKEND', 'nccl'))      # 'nccl' = RCCL; gloo only for single-GPU rehearsals

    H = W = args.size
    from yolov3_tensorflow_amd import ops
    for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):      # kernel-selection overrides for A/B runs (yolo_set_tuning);
        k, v = kv.split('=')                                                 # before the model is built: workspaces are sized from the plans
        ops.set_tuning(k, int(v))
    model, loss, opt, grids = build_model(args.backbone, H, W, args.batch, args.classes, device, focal=args.focal)
    model.use_hip_graph = bool(args.graph)
    model.overlap_wgrad = not args.no_overlap
    model.g.fused_bn_bwd = not args.no_fused_bn
    model.bucket_updates = not args.no_bucket_updates
    model.native_sequencer = not args.no_sequencer
    if args.side_priority is not None:
        model.g.wgrad_stream = torch.cuda.Stream(device=device, priority=args.side_priority)
        model.g.use_side_stream(model.g.wgrad_stream)
    masked_main = None
    if args.side_cus:
        side_mask = cu_mask(args.side_cus, args.side_cu_pattern)
        model.g.wgrad_stream = masked_stream(device, side_mask)
        model.g.use_side_stream(model.g.wgrad_stream)
        if args.main_cus:
            masked_main = masked_stream(device, [(~w) & 0xffffffff for w in side_mask])
    if masked_main is not None:
        torch.cuda.synchronize(device)
        torch.cuda.set_stream(masked_main)
    elif args.main_priority:
        torch.cuda.synchronize(device)          # the model was built on the default stream; streams made here do not wait for it implicitly
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=args.main_priority))
    if args.wgrad_batch is not None:
        model.g.wgrad_batch = max(1, args.wgrad_batch)
    if args.wgrad_gflop is not None:
        model.g.wgrad_cost_limit = args.wgrad_gflop
    if world > 1:
        from yolov3_tensorflow_amd import parallel
        parallel.setup_data_parallel(model)
    images, labels = synthetic_batch(args.batch, H, W, args.classes, rank)
    model.stage_batch(images, labels)        # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(model.loss_value.item())
    model.check_device_protocols()           # a timed-out grid barrier would make the run invalid: fail loudly
    if not np.isfinite(final_loss):
        raise SystemExit('non-finite loss %r' % final_loss)

    ms = elapsed / args.steps * 1e3
    ips = args.batch * world * args.steps / elapsed
    gflop = TRAIN_GFLOP_PER_IMAGE if (args.size == 416 and args.classes == 80 and args.backbone == 'resnet-18') else None
    out = {
        'metric': 'images/sec training 416x416 ResNet18-YOLOv3 (%s)' % args.dtype,
        'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': '%s-YOLOv3 %dx%d, %d classes, anchors 3/3/3 (COCO-9), per-GPU batch %d, full training step '
                               '(fwd + YOLOv3 %sloss + bwd + RAdam/L2%s), random-init weights'
                               % (args.backbone, H, W, args.classes, args.batch, 'focal ' if args.focal else '',
                                  ' + RCCL grad all-reduce' if world > 1 else ''),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'hip_graph': bool(model.use_hip_graph),
                   'final_loss': round(final_loss, 4)},
    }
    if rank == 0 and gflop is not None:
        out['step_tflops'] = round(ips / world * gflop / 1000.0, 2)      # per GPU, whole step, algorithmic conv FLOPs
        out['step_mfma_frac'] = round(out['step_tflops'] / PEAK_BF16_TFLOPS, 4)
    if not args.no_roofline:
        # every rank runs the (collective-free) measurement passes so that the ranks stay in step; rank 0 reports its own numbers
        k = max(2, min(5, args.steps))
        alone = kernel_rooflines(model, k, overlap=False)
        instep = kernel_rooflines(model, k, overlap=True)
    if rank == 0 and not args.no_roofline:
        traffic, traffic_src = strip_hbm_traffic()
        tf = alone['strip']['rate'] / 1e12
        tf_in = instep['strip']['rate'] / 1e12
        out['roofline'] = {'bound': 'mfma', 'kernel': 'conv3x3_strip_kernel<.., false> + conv3x3_s32_kernel<.., false> + conv3x3_stream_kernel<0, false> (every 3x3 stride-1 conv forward + '
                                                      'plain data-gradient launch, all tile variants; the forward launches of the 64-channel layers run '
                                                      'the streaming kernel: the `stream` entry; the data gradients that also carry a BatchNorm reduce are '
                                                      'the dgrad_bn entry)',
                           'achieved': round(tf, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_BF16_TFLOPS, 4),
                           'traffic': traffic, 'traffic_unit': 'bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), ' + traffic_src,
                           'avg_launch_ms': round(alone['strip']['avg_launch_ms'], 5), 'launches_per_step': alone['strip']['launches_per_step'],
                           'timing': 'each launch alone on the stream (weight-gradient stream folded away); in_step = the same launches under the '
                                     'two-stream schedule the headline runs, where concurrent weight-gradient kernels share the CUs',
                           'in_step': {'achieved': round(tf_in, 2), 'frac': round(tf_in / PEAK_BF16_TFLOPS, 4),
                                       'avg_launch_ms': round(instep['strip']['avg_launch_ms'], 5)},
                           'stream': {'kernel': 'conv3x3_stream_kernel<0, false> (weights in registers, pixels through an LDS ring, one workgroup per CU): '
                                                'the forward launches of the 64-channel 3x3 layers; bound by vector-instruction issue (DESIGN.md section 4)',
                                      'achieved': round(alone['stream']['rate'] / 1e12, 2), 'frac': round(alone['stream']['rate'] / 1e12 / PEAK_BF16_TFLOPS, 4),
                                      'in_step_achieved': round(instep['stream']['rate'] / 1e12, 2),
                                      'avg_launch_ms': round(alone['stream']['avg_launch_ms'], 5), 'launches_per_step': alone['stream']['launches_per_step']},
                           'dgrad_bn': {'kernel': 'conv3x3_strip_kernel<.., true> / conv3x3_s32_kernel<.., true> (strip_*: the 3x3 stride-1 layers); igemm_*: the other '
                                                  'layers = the three 3x3 stride-2 ones as parity classes on conv3x3_s32_kernel<.., true, 1> + the 1x1 ones on '
                                                  'igemm_fwd_kernel<.., true>: data gradient + ReLU masking + BatchNorm-backward partial sums of the unit it '
                                                  'completes (reads y and the sign bytes on top of the conv operands)',
                                        'strip_achieved': round(alone['strip_bn']['rate'] / 1e12, 2),
                                        'strip_frac': round(alone['strip_bn']['rate'] / 1e12 / PEAK_BF16_TFLOPS, 4),
                                        'strip_in_step_achieved': round(instep['strip_bn']['rate'] / 1e12, 2),
                                        'strip_avg_launch_ms': round(alone['strip_bn']['avg_launch_ms'], 5),
                                        'strip_launches_per_step': alone['strip_bn']['launches_per_step'],
                                        'igemm_achieved': round(alone['other_bn']['rate'] / 1e12, 2),
                                        'igemm_launches_per_step': alone['other_bn']['launches_per_step']},
                           'wgrad': {'kernel': 'wgrad9_kernel (3x3 stride-1 layers of 20 x 20 pixels and more: stationary dW tile, 128 workgroups = half the CUs, the rest stays free for the main stream) / wgrad3x3_strip_kernel (13 x 13 layers) / igemm_wgrad_kernel (the rest) on the weight-gradient stream; '
                                               'slab_sum = wgrad_reduce_batched_kernel (one launch per gradient bucket, HBM-bound)',
                                     'strip_achieved': round(alone['wgrad3x3']['rate'] / 1e12, 2),
                                     'strip_frac': round(alone['wgrad3x3']['rate'] / 1e12 / PEAK_BF16_TFLOPS, 4),
                                     'strip_in_step_achieved': round(instep['wgrad3x3']['rate'] / 1e12, 2),
                                     'strip_launches_per_step': alone['wgrad3x3']['launches_per_step'],
                                     'strip_ms_per_step': round(alone['wgrad3x3']['ms_per_step'], 4),
                                     'other_achieved': round(alone['wgrad_other']['rate'] / 1e12, 2),
                                     'other_launches_per_step': alone['wgrad_other']['launches_per_step'],
                                     'other_ms_per_step': round(alone['wgrad_other']['ms_per_step'], 4),
                                     'family_ms_per_step': round(alone['wgrad3x3']['ms_per_step'] + alone['wgrad_other']['ms_per_step'], 4),
                                     'family_in_step_ms_per_step': round(instep['wgrad3x3']['ms_per_step'] + instep['wgrad_other']['ms_per_step'], 4),
                                     'slab_sum_ms_per_step': round(alone['wgrad_sum']['ms_per_step'], 4),
                                     'slab_sum_launches_per_step': alone['wgrad_sum']['launches_per_step']},
                           'other_conv': {'kernel': 'igemm_fwd_kernel (stem, stride-2, 1x1, fused-concat launches)',
                                          'achieved': round(alone['other']['rate'] / 1e12, 2), 'in_step_achieved': round(instep['other']['rate'] / 1e12, 2),
                                          'avg_launch_ms': round(alone['other']['avg_launch_ms'], 5),
                                          'launches_per_step': alone['other']['launches_per_step']}}
        names = {'bn_fwd': 'bn_act_fwd / bn_pool_fwd (BatchNorm apply + ReLU + residual, stem BN + max-pool)',
                 'bn_bwd': 'bn_bwd_apply (+ reduce / fused where the data gradient does not carry the reduce) / stem_pool_bwd_wgrad (BatchNorm backward)',
                 'loss': 'loss_assign + loss_main + loss_finalize (YOLOv3 loss forward + d(logits))',
                 'optimizer': 'radam_l2_kernel (RAdam + L2, 30 B / parameter)'}
        out['hbm'] = {'peak': HBM_PEAK_GBS, 'achievable': HBM_ACHIEVABLE_GBS, 'unit': 'GB/s',
                      'bytes': 'operand bytes of the launches (every full-size tensor argument once), not PMC traffic'}
        for fam, label in names.items():
            a, b = alone[fam], instep[fam]
            out['hbm'][fam] = {'kernel': label, 'achieved': round(a['rate'] / 1e9, 1), 'frac_of_achievable': round(a['rate'] / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                               'in_step_achieved': round(b['rate'] / 1e9, 1), 'ms_per_step': round(a['ms_per_step'], 4),
                               'in_step_ms_per_step': round(b['ms_per_step'], 4), 'launches_per_step': a['launches_per_step'],
                               'MB_per_step': round(a['work_per_step'] / 1e6, 1)}
    if rank == 0:
        out['parity'] = parity_statement(args.dtype)
        out['parity'].setdefault(args.dtype, {})['images_per_sec'] = out['value'] if world == 1 else None
        if world == 1 and args.dtype == 'bf16' and not args.no_roofline:      # (the long form of the record: the default run; ~10 s)
            try:
                out['parity'].setdefault('fp16', {}).update(fp16_throughput(args, device))
            except Exception as e:                                            # the headline number must not depend on the second build
                out['parity'].setdefault('fp16', {})['images_per_sec_error'] = repr(e)
    if world > 1:
        out['rccl'] = rccl_statement(model, world, device)       # collective (a tiny all-gather of the per-rank view): every rank calls it
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(H, W, args.classes)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
