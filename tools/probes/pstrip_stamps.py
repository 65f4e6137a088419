"""where a workgroup of conv3x3_pstrip_kernel spends its cycles: s_memtime stamps of wave 0 of every workgroup (diagnostic build only:
make -C yolov3_tensorflow_amd/csrc EXTRA_conv_pstrip=-DPS_STAMPS; the product library has no stamps and no yolo_debug_ps_stamps symbol).
usage: python tools/probes/pstrip_stamps.py N H W Cin Cout [setting]      medians over workgroups, shader cycles"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend, _lib
N, H, W, Cin, Cout = map(int, sys.argv[1:6])
setting = sys.argv[6] if len(sys.argv) > 6 else 'pstrip=1'
for kv in filter(None, setting.split(',')):
    k, v = kv.split('=')
    ops.set_tuning(k, int(v))
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
st = torch.zeros(plan['workgroups'] * 32, dtype=torch.int64, device=dev)
lib.yolo_debug_ps_stamps.restype = C.c_int
lib.yolo_debug_ps_stamps.argtypes = [C.c_void_p]
assert lib.yolo_debug_ps_stamps(st.data_ptr()) == 0
ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
torch.cuda.synchronize()
lib.yolo_debug_ps_stamps(None)
s = st.cpu().reshape(-1, 32).double()
names = {1: 'setup (addresses)', 2: 'prologue loads + barrier', 3: 'first slice (9 K steps)', 4: 'remaining slices', 5: 'drain + barrier',
         6: 'K-group reduction', 7: 'epilogue'}
print(plan)
t0 = s[:, 0].min()
print('workgroup start spread: %.0f cycles; whole workgroup median %.0f, max end %.0f' % (float((s[:, 0] - t0).max()), float((s[:, 7] - s[:, 0]).median()), float((s[:, 7] - t0).max())))
for i in range(1, 8):
    print('  %-28s %8.0f' % (names[i], float((s[:, i] - s[:, i - 1]).median())))
print('  epilogue: first sync %.0f, staging %.0f, sync %.0f, store loop %.0f, statistics %.0f' % tuple(
    float((s[:, i] - s[:, j]).median()) for i, j in ((13, 6), (14, 13), (15, 14), (16, 15), (7, 16))))
print('  one K step (slice 1, tap 4): DMA issue %.0f, reads + MFMAs %.0f, vmcnt wait %.0f, barrier %.0f' % tuple(
    float((s[:, i] - s[:, i - 1]).median()) for i in (9, 10, 11, 12)))
