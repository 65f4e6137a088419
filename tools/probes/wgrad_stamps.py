"""where the waves of wgrad3x3_strip_kernel (pipelined variant) spend their cycles (diagnostic build only: make -C yolov3_tensorflow_amd/csrc diag;
the product library has no stamps and no yolo_debug_wg_stamps symbol).
usage: python tools/probes/wgrad_stamps.py N H W Cin Cout      medians over waves, s_memtime ticks (shader cycles)"""
import ctypes as C, math, os, sys
os.environ.setdefault('YOLO_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'yolov3_tensorflow_amd', 'libyolov3_amd_diag.so'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend, _lib
N, H, W, Cin, Cout = map(int, sys.argv[1:6])
dev = torch.device('cuda:0')
lib = _lib.load()
g = torch.Generator().manual_seed(3)
dt = backend.torch_dtype()
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
dy = torch.randn(N, H, W, Cout, generator=g).to(dt).to(dev)
p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
splits = ops.conv2d_wgrad_splits(p)
n = Cout * 9 * Cin
dw = torch.zeros(n, device=dev)
slabs = torch.empty(splits * n, device=dev) if splits > 1 else None
for _ in range(5):
    ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 100
bco = 128 if Cout % 128 == 0 else 64
nwg = 3 * (Cin // 64) * (Cout // bco) * splits
st = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=dev)
lib.yolo_debug_wg_stamps.restype = C.c_int
lib.yolo_debug_wg_stamps.argtypes = [C.c_void_p]
assert lib.yolo_debug_wg_stamps(st.data_ptr()) == 0
ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs)
torch.cuda.synchronize()
lib.yolo_debug_wg_stamps(None)
s = st.cpu().double().reshape(nwg, 8, 16)
s = s[s[:, 0, 0] > 0]
stages = float(s[:, :, 7].median())
fl = 2.0 * N * H * W * Cout * Cin * 9
print('%dx%d %d->%d: %d workgroups (%d splits), %.0f stages each; stamped build %.1f us = %.0f TFLOP/s' % (H, W, Cin, Cout, nwg, splits, stages, us, fl / us / 1e6))
t0 = s[:, :, 0].min()
print('launch span %.0f cycles; workgroup starts: median +%.0f, max +%.0f' % (float(s[:, :, 9].max() - t0), float((s[:, 0, 0] - t0).median()), float((s[:, 0, 0] - t0).max())))
med = lambda i: float(s[:, :, i].median())
print('per wave (median):')
print('  setup                      %8.0f' % med(1))
print('  stage loop                 %8.0f   = %.0f per 64-pixel stage (24 MFMAs 16x16x32 = 384 cycles of matrix pipe per wave)' % (med(2), med(2) / stages))
print('    vmcnt(0) / lgkmcnt wait  %8.0f   = %.0f per stage' % (med(4), med(4) / stages))
print('    barrier                  %8.0f   = %.0f per stage' % (med(5), med(5) / stages))
print('    body                     %8.0f   = %.0f per stage' % (med(6), med(6) / stages))
print('  slab store                 %8.0f' % med(3))
