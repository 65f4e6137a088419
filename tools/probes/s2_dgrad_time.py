"""the three stride-2 3x3 data gradients of the benchmark model (ResNet18 416 x 416, batch 32) timed alone: parity classes on conv3x3_s32_kernel
("s32_s2" = 1) against the implicit-GEMM class kernel (0); plain, and as the step runs them (accumulate = 2 + fused BatchNorm reduce)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend
dev = torch.device('cuda:0')
ACT = backend.torch_dtype()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


SHAPES = [('conv2d_6', 32, 104, 104, 64, 128), ('conv2d_11', 32, 52, 52, 128, 256), ('conv2d_16', 32, 26, 26, 256, 512)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':      # BASELINE.json configs[0] (320 x 320, batch 2) and its batch-8 variant
    SHAPES = [('conv2d_6', 2, 80, 80, 64, 128), ('conv2d_11', 2, 40, 40, 128, 256), ('conv2d_16', 2, 20, 20, 256, 512),
              ('conv2d_6', 8, 80, 80, 64, 128), ('conv2d_11', 8, 40, 40, 128, 256), ('conv2d_16', 8, 20, 20, 256, 512)]
for name, N, H, W, Cin, Cout in SHAPES:
    g = torch.Generator().manual_seed(1)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 2, 'same')
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(ACT).to(dev)
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT, device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    dy = torch.randn(N, p.Ho, p.Wo, Cout, generator=g).to(ACT).to(dev)
    dx = torch.zeros(N, H, W, Cin, dtype=ACT, device=dev)
    M = N * H * W
    y = torch.randn(M, Cin, generator=g).to(ACT).to(dev)
    mean, rstd = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    fl = 2.0 * N * p.Ho * p.Wo * Cout * Cin * 9
    for rnd in range(2):
        for s2 in (0, 1):
            ops.set_tuning('s32_s2', s2)
            rows = ops.conv2d_dgrad_bn_rows(p)
            part = torch.zeros(rows, 3, Cin, device=dev)
            bn = dict(mask=mask, y=y, mean=mean, rstd=rstd, partial=part)
            t0 = timed(lambda: ops.conv2d_dgrad(p, dy, w_dg, dx))
            t1 = timed(lambda: ops.conv2d_dgrad(p, dy, w_dg, dx, accumulate=2, bn=bn))
            print('%-10s %dx%d %d<-%d  s32_s2 %d  rows %5d | plain %6.1f us %5.0f TFLOP/s | acc=2 + bn %6.1f us %5.0f TFLOP/s' % (
                name, H, W, Cin, Cout, s2, rows, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)
    ops.set_tuning('s32_s2', 1)
