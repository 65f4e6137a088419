"""every 3x3 / stride-1 layer shape of the benchmark model (ResNet18-YOLOv3 416 x 416, batch 32) under every tile configuration of
conv3x3_s32_kernel and under the shipping choice (tuning s32 = 0): forward with statistics, plain data gradient, data gradient with the fused
BatchNorm-backward reduce (accumulating), timed alone with HIP events, interleaved rounds in ONE process (cdna guide rule 24).
usage: python tools/probes/s32_sweep.py [configs, e.g. 0,1,2] [rounds]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend

dev = torch.device('cuda:0')
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else '0,1,2,3,4,5,6,7,8').split(',')]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ACT = backend.torch_dtype()
LAYERS = [  # name, N, H, W, Cin, Cout
    ('104x64->64', 32, 104, 104, 64, 64),
    ('52x128->128', 32, 52, 52, 128, 128),
    ('26x256->256', 32, 26, 26, 256, 256),
    ('13x512->512', 32, 13, 13, 512, 512),
    ('52x128->256', 32, 52, 52, 128, 256),
    ('26x256->512', 32, 26, 26, 256, 512),
    ('13x512->256', 32, 13, 13, 512, 256),
]
if os.environ.get('S32_LAYERS'):
    keep = os.environ['S32_LAYERS'].split(',')
    LAYERS = [l for l in LAYERS if l[0].split('->')[0].split('x')[0] in keep or l[0] in keep]


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


for name, N, H, W, Cin, Cout in LAYERS:
    g = torch.Generator().manual_seed(1)
    M = N * H * W
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT).to(dev)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(ACT).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT, device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    y = torch.empty(N, H, W, Cout, dtype=ACT, device=dev)
    dx = torch.zeros(N, H, W, Cin, dtype=ACT, device=dev)
    ybn = torch.randn(M, Cin, generator=g).to(ACT).to(dev)
    mean, rstd = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    fl = 2.0 * M * Cout * Cin * 9
    res = {}
    for r in range(rounds):
        for c in [-1] + cfgs:
            ops.set_tuning('s32', 0 if c < 0 else 1 + c)
            ops.set_tuning('stream', -1 if c < 0 else 0)          # (the streaming kernel claims the 64-channel layers first)
            plan = ops.conv2d_fwd_plan(p)
            if c >= 0 and plan['family'] != 's32':
                continue
            rows = ops.conv2d_stat_rows(p)
            ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
            prow = ops.conv2d_dgrad_bn_rows(p)
            part = torch.zeros(prow, 3, Cin, device=dev)
            bn = dict(mask=mask, y=ybn, mean=mean, rstd=rstd, partial=part)
            t_f = timed(lambda: ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq))
            t_d = timed(lambda: ops.conv2d_dgrad(p, dy, w_dg, dx))
            t_b = timed(lambda: ops.conv2d_dgrad(p, dy, w_dg, dx, accumulate=True, bn=bn))
            key = (c, plan['family'], plan['bm'], plan['bn'], plan['workgroups'])
            res.setdefault(key, []).append((t_f, t_d, t_b))
    ops.set_tuning('s32', -1)
    ops.set_tuning('stream', -1)
    for key, v in res.items():
        best = [min(t[i] for t in v) for i in range(3)]
        med = [sorted(t[i] for t in v)[len(v) // 2] for i in range(3)]
        print('%-12s cfg %2d %-6s %3dx%-3d wg %5d | fwd %6.1f us %6.0f TF | dgrad %6.1f us %6.0f TF | dgrad+bn acc %6.1f us %6.0f TF   (median %5.1f %5.1f %5.1f)' % (
            (name,) + key + (best[0], fl / best[0] / 1e6, best[1], fl / best[1] / 1e6, best[2], fl / best[2] / 1e6) + tuple(med)), flush=True)
