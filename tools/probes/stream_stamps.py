"""where a workgroup of conv3x3_stream_kernel spends its cycles: s_memtime stamps of wave 0 of every workgroup (diagnostic build only:
make -C yolov3_tensorflow_amd/csrc EXTRA_conv_stream=-DST_STAMPS; the product library has no stamps and no yolo_debug_st_stamps symbol).
usage: python tools/probes/stream_stamps.py N H W Cout [--dgrad]      medians over workgroups, 100 MHz s_memtime ticks converted to ns"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend, _lib
N, H, W, Cout = map(int, sys.argv[1:5])
Cin = 64
ops.set_tuning('stream', 1)
dev = torch.device('cuda:0')
lib = _lib.load()
g = torch.Generator().manual_seed(3)
dt = backend.torch_dtype()
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(dt).to(dev)
p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
y = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
plan = ops.conv2d_fwd_plan(p)
rows = ops.conv2d_stat_rows(p)
ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
for _ in range(5):
    ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
torch.cuda.synchronize()
nwg = plan['workgroups']
st = torch.zeros(nwg * 64 + nwg * 64, dtype=torch.int64, device=dev)
lib.yolo_debug_st_stamps.restype = C.c_int
lib.yolo_debug_st_stamps.argtypes = [C.c_void_p]
assert lib.yolo_debug_st_stamps(st.data_ptr()) == 0
ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
torch.cuda.synchronize()
lib.yolo_debug_st_stamps(None)
allst = st.cpu().double()
s = allst[:nwg * 64].reshape(-1, 64)
ws = allst[nwg * 64:].reshape(nwg, 8, 8)
print(plan)
full = s[:, 4 + 10] > 0                 # workgroups with >= 11 steps
s = s[full]
t0 = s[:, 0].min()
print('%d full workgroups; start spread %.0f; whole workgroup median %.0f, max end %.0f (s_memtime ticks)' % (
    int(full.sum()), float((s[:, 0] - t0).max()), float((s[:, 42] - s[:, 0]).median()), float((s[:, 42] - t0).max())))
d = lambda i, j: float((s[:, i] - s[:, j]).median())
print('  prologue loads + barrier   %8.0f' % d(1, 0))
print('  weights to registers       %8.0f' % d(2, 1))
print('  loop setup                 %8.0f' % d(3, 2))
print('  steps: ' + ' '.join('%.0f' % d(5 + i, 4 + i) for i in range(10)))
print('  loop total                 %8.0f' % d(40, 3))
print('  drain                      %8.0f' % d(41, 40))
print('  statistics row             %8.0f' % d(42, 41))
ws = ws[full]
arr = ws[:, :, 0] - ws[:, :, 0].min(dim=1, keepdim=True).values          # arrival at step 5's barrier, relative to the first wave
rel = ws[:, :, 1] - ws[:, :, 0].min(dim=1, keepdim=True).values
print('  step 5 barrier: arrival of waves 0-7 after the first (median): ' + ' '.join('%.0f' % float(arr[:, w].median()) for w in range(8)))
print('                  release of waves 0-7 after the first arrival:   ' + ' '.join('%.0f' % float(rel[:, w].median()) for w in range(8)))
per = ws[:, :, 2] - ws[:, :, 1]
print('  release(5) -> arrival(6) per wave: ' + ' '.join('%.0f' % float(per[:, w].median()) for w in range(8)))
