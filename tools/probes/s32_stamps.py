"""where the waves of conv3x3_s32_kernel spend their cycles (diagnostic build only: make -C yolov3_tensorflow_amd/csrc
EXTRA_conv_s32="-fno-slp-vectorize -DS32_STAMPS"; the product library has no stamps and no yolo_debug_s32_stamps symbol).
usage: python tools/probes/s32_stamps.py cfg N H W Cin Cout      medians over waves, s_memtime ticks (shader cycles)"""
import ctypes as C, math, os, sys
os.environ.setdefault('YOLO_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'yolov3_tensorflow_amd', 'libyolov3_amd_diag.so'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend, _lib
cfg, N, H, W, Cin, Cout = map(int, sys.argv[1:7])
NW = {0: 4, 1: 4, 2: 4, 3: 4, 4: 4, 5: 8, 6: 8, 7: 8}[cfg]
ops.set_tuning('stream', 0)
ops.set_tuning('s32', 1 + cfg)
dev = torch.device('cuda:0')
lib = _lib.load()
g = torch.Generator().manual_seed(3)
dt = backend.torch_dtype()
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(9 * Cin)).to(dt).to(dev)
p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')
y = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
plan = ops.conv2d_fwd_plan(p)
assert plan['family'] == 's32', plan
rows = ops.conv2d_stat_rows(p)
ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
for _ in range(5):
    ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
torch.cuda.synchronize()
nwg = plan['workgroups']
st = torch.zeros(nwg * NW * 16, dtype=torch.int64, device=dev)
lib.yolo_debug_s32_stamps.restype = C.c_int
lib.yolo_debug_s32_stamps.argtypes = [C.c_void_p]
assert lib.yolo_debug_s32_stamps(st.data_ptr()) == 0
ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq)
torch.cuda.synchronize()
lib.yolo_debug_s32_stamps(None)
s = st.cpu().double().reshape(nwg, NW, 16)
taps = 9 * (Cin // 64)
t0 = s[:, :, 0][s[:, :, 0] > 0].min()
print(plan, 'taps', taps)
print('launch span (first start -> last end): %.0f cycles; workgroup starts: median +%.0f, max +%.0f' % (
    float(s[:, :, 9].max() - t0), float((s[:, 0, 0] - t0).median()), float((s[:, 0, 0] - t0).max())))
med = lambda i: float(s[:, :, i].median())
print('per wave (median over all waves of all workgroups):')
print('  setup (addresses, masks, first weight DMA)  %8.0f' % med(1))
print('  tap loop                                    %8.0f   = %.0f per tap' % (med(2), med(2) / taps))
print('    counted vmcnt / lgkmcnt wait              %8.0f   = %.0f per tap' % (med(4), med(4) / taps))
print('    barrier                                   %8.0f   = %.0f per tap' % (med(5), med(5) / taps))
print('    weight LDS-DMA issue                      %8.0f   = %.0f per tap' % (med(6), med(6) / taps))
print('    fragment reads + MFMAs                    %8.0f   = %.0f per tap' % (med(7), med(7) / taps))
print('    slice boundaries (barrier + strip issue)  %8.0f   = %.0f per slice' % (med(8), med(8) / (Cin // 64)))
print('  epilogue (staging, row walk, stores)        %8.0f' % med(3))
for wv in range(NW):
    print('  wave %d: wait %6.0f  barrier %6.0f  issue %6.0f  compute %6.0f' % (wv, float(s[:, wv, 4].median()), float(s[:, wv, 5].median()),
                                                                              float(s[:, wv, 6].median()), float(s[:, wv, 7].median())))
