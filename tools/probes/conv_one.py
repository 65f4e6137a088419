"""one 3x3 convolution (forward with BatchNorm statistics, or the plain data gradient with --dgrad) timed alone under several tuning
settings, interleaved in ONE process (the CDNA guide's rule for A/B numbers): median / min microseconds and TFLOP/s per setting.

usage: python tools/probes/conv_one.py N H W Cin Cout "s32=0" "s32=2" "strip_bm=0" [--rounds 7] [--iters 20] [--dgrad]
(under rocprofv3 --pmc ... -- python tools/probes/conv_one.py ... --rounds 1 --iters 2 for counters)"""
import argparse, math, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend

ap = argparse.ArgumentParser()
ap.add_argument('dims', nargs=5, type=int)
ap.add_argument('settings', nargs='+')
ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--dgrad', action='store_true')
a = ap.parse_args()
N, H, W, Cin, Cout = a.dims
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
dt = backend.torch_dtype()
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(dt).to(dev)
dy = torch.randn(N, H, W, Cout, generator=g).to(dt).to(dev)
p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
y = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
dx = torch.empty(N, H, W, Cin, dtype=dt, device=dev)
w_dg = torch.empty(Cin, 3, 3, Cout, dtype=dt, device=dev)
ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
fl = 2.0 * N * H * W * Cout * Cin * 9


def apply(setting):
    for k in ('s32', 'strip_bm', 'stream'):
        ops.set_tuning(k, -1)
    for kv in filter(None, setting.split(',')):
        k, v = kv.split('=')
        ops.set_tuning(k, int(v))


def make(setting):
    apply(setting)
    rows = ops.conv2d_stat_rows(p)
    ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
    plan = ops.conv2d_fwd_plan(p)

    def run():
        apply(setting)
        if a.dgrad:
            ops.conv2d_dgrad(p, dy, w_dg, dx)
        else:
            ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
    return run, plan


runs = [make(s) for s in a.settings]
ref = None
times = [[] for _ in runs]
for r in range(a.rounds + 1):
    for i, (run, plan) in enumerate(runs):
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[i].append(e0.elapsed_time(e1) * 1e3 / a.iters)
        if r == 0:
            out = (dx if a.dgrad else y).float()
            if ref is None:
                ref = out.clone()
            else:
                err = float((out - ref).abs().max())
                if err >= 0.1:
                    print('WARNING: setting %s differs from the first by %g' % (a.settings[i], err))
for s, t, (run, plan) in zip(a.settings, times, runs):
    med, mn = statistics.median(t), min(t)
    print('%-28s %-7s tile %3d x %-3d wg %4d  median %6.1f us  min %6.1f us  %6.1f TFLOP/s' % (
        s, plan['family'], plan['tile_pixels'], plan['bn'], plan['workgroups'], med, mn, fl / med / 1e6))
