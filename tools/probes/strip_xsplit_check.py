"""strip kernel with the XCD-aware tile split ("strip_xsplit" = 2 / 4) against the contiguous-run dealing: the same tiles, so outputs, statistics
rows and fused-reduce rows must be bit-identical; and the alone time of the 13 x 13 / 26 x 26 layers under each."""
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


ops.set_tuning('s32', 0); ops.set_tuning('stream', 0)
for name, N, H, W, Cin, Cout in [('13x512->512', 32, 13, 13, 512, 512), ('13x512->256', 32, 13, 13, 512, 256), ('26x256->256', 32, 26, 26, 256, 256),
                                 ('26x256->512', 32, 26, 26, 256, 512), ('ragged 7x19x21', 7, 19, 21, 128, 256)]:
    g = torch.Generator().manual_seed(1)
    M = N * H * W
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT).to(dev)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(ACT).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
    w_dg = torch.empty(Cin, 3, 3, Cout, dtype=ACT, device=dev)
    ops.repack_dgrad_weights(w, w_dg, Cout, 3, 3, Cin)
    ybn = torch.randn(M, Cin, generator=g).to(ACT).to(dev)
    mean, rstd = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    mask = torch.randint(0, 256, (M * Cin // 8,), generator=g, dtype=torch.uint8).to(dev)
    ref = None
    for xs in (0, 2, 4, 0):
        ops.set_tuning('strip_xsplit', xs)
        rows = ops.conv2d_stat_rows(p)
        y = torch.empty(N, H, W, Cout, dtype=ACT, device=dev)
        ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
        ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
        prow = ops.conv2d_dgrad_bn_rows(p)
        part = torch.zeros(prow, 3, Cin, device=dev)
        dx = torch.zeros(N, H, W, Cin, dtype=ACT, device=dev)
        bn = dict(mask=mask, y=ybn, mean=mean, rstd=rstd, partial=part)
        ops.conv2d_dgrad(p, dy, w_dg, dx, bn=bn)
        torch.cuda.synchronize()
        cur = (y.view(torch.int16).clone(), ss.clone(), sq.clone(), dx.view(torch.int16).clone(), part.clone())
        if ref is None:
            ref = cur
        same = all(torch.equal(a, b) for a, b in zip(cur, ref))
        t_f = timed(lambda: ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq))
        t_b = timed(lambda: ops.conv2d_dgrad(p, dy, w_dg, dx, accumulate=True, bn=bn))
        fl = 2.0 * M * Cout * Cin * 9
        print('%-16s xsplit %d  %s  %s | fwd %6.1f us %5.0f TF | dgrad+bn acc %6.1f us %5.0f TF' % (
            name, xs, ops.conv2d_fwd_plan(p)['family'], 'bit-identical' if same else 'DIFFERS', t_f, fl / t_f / 1e6, t_b, fl / t_b / 1e6), flush=True)
ops.set_tuning('strip_xsplit', 0); ops.set_tuning('s32', -1); ops.set_tuning('stream', -1)
