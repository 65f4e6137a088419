"""s_memtime stamps of wave 0 of every workgroup of one strip-conv launch: where a K-step's time goes (needs tools/probes/strip_stamps.patch applied to csrc/conv_igemm.hip)"""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend, _lib
dev = torch.device('cuda:0'); ACT = backend.torch_dtype()
lib = _lib.load()
lib.yolo_strip_set_stamps.argtypes = [ctypes.c_void_p]
for H, C in ((104, 64), (52, 128), (26, 256), (13, 512)):
    p = ops.conv_problem(32, H, H, C, C, 3, 1, 'same')
    x = torch.randn(32, H, H, C).to(ACT).to(dev); w = (torch.randn(C, 3, 3, C) * 0.05).to(ACT).to(dev)
    y = torch.empty(32, H, H, C, dtype=ACT, device=dev)
    for _ in range(3): ops.conv2d_fwd(p, x, w, y)
    stamps = torch.zeros(8192, 32, dtype=torch.int64, device=dev)
    lib.yolo_strip_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    ops.conv2d_fwd(p, x, w, y)
    torch.cuda.synchronize()
    lib.yolo_strip_set_stamps(None)
    s = stamps.cpu().numpy()
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    d = lambda a, b: float(np.median(s[:, b] - s[:, a]))
    nk = min(12, 9 * (C // 64))
    print('%3d^2 C %3d: %4d workgroups, kernel span %6.0f ticks' % (H, C, len(s), s[:, 28].max() - t0))
    print('   start->first barrier entry %6.0f   first barrier wait %6.0f' % (d(0, 1), d(1, 2)))
    print('   K-step (barrier exit -> next barrier entry), median per step: ' + ' '.join('%5.0f' % d(2 + 2 * k, 3 + 2 * k) for k in range(nk - 1)))
    print('   barrier wait, median per step:                                ' + ' '.join('%5.0f' % d(1 + 2 * k, 2 + 2 * k) for k in range(nk)))
    print('   last stamp in loop -> loop end %6.0f  final sync %5.0f  epilogue %6.0f   whole workgroup %6.0f' % (d(2 * nk, 26), d(26, 27), d(27, 28), d(0, 28)))
