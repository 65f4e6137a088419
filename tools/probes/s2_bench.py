"""the three stride-2 3x3 layers of ResNet18 at 416^2 / batch 32, forward and data gradient alone"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend
dev = torch.device('cuda:0'); ACT = backend.torch_dtype()
for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):
    k, v = kv.split('='); ops.set_tuning(k, int(v))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for H, ci, co in ((104, 64, 128), (52, 128, 256), (26, 256, 512)):
    p = ops.conv_problem(32, H, H, ci, co, 3, 2, 'same')
    x = torch.randn(32, H, H, ci).to(ACT).to(dev); w = (torch.randn(co, 3, 3, ci) * 0.05).to(ACT).to(dev)
    y = torch.empty(32, p.Ho, p.Wo, co, dtype=ACT, device=dev); dy = torch.randn(32, p.Ho, p.Wo, co).to(ACT).to(dev)
    wd = torch.empty(ci, 3, 3, co, dtype=ACT, device=dev); ops.repack_dgrad_weights(w, wd, co, 3, 3, ci)
    dx = torch.zeros(32, H, H, ci, dtype=ACT, device=dev)
    fl = 2.0 * 32 * p.Ho * p.Wo * co * ci * 9
    tf = timeit(lambda: ops.conv2d_fwd(p, x, w, y)); td = timeit(lambda: ops.conv2d_dgrad(p, dy, wd, dx, accumulate=True))
    print('%3d^2 %3d->%3d  fwd %5.1f us %5.0f TFLOP/s   dgrad %5.1f us %5.0f TFLOP/s' % (H, ci, co, tf, fl / tf / 1e6, td, fl / td / 1e6))
