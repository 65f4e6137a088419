"""wgrad9_kernel timed alone on one layer shape (any build selected with YOLO_LIB_PATH; the ablation builds -DW9_ABL_* compute wrong sums on purpose).
usage: python tools/probes/wgrad9_time.py N H W Cin Cout [wgs]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend
N, H, W, Cin, Cout = map(int, sys.argv[1:6])
if len(sys.argv) > 6:
    ops.set_tuning('wgrad9_wgs', int(sys.argv[6]))
ops.set_tuning('wgrad9', 1)
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
dt = backend.torch_dtype()
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
dy = torch.randn(N, H, W, Cout, generator=g).to(dt).to(dev)
p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
splits = ops.conv2d_wgrad_splits(p)
n = Cout * 9 * Cin
dw = torch.zeros(n, device=dev)
slabs = torch.empty(splits * n, device=dev) if splits > 1 else None
best = 1e9
for rep in range(5):
    for _ in range(3):
        ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs)
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) * 50.0)
fl = 2.0 * N * H * W * Cout * Cin * 9
print('%dx%d %d->%d  splits %d  %.1f us  %.0f TFLOP/s' % (H, W, Cin, Cout, splits, best, fl / best / 1e6))
