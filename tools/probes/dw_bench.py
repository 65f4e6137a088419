"""micro-benchmark of the mixed depthwise forward / data-gradient kernel (probe): time per launch on the MixNet18 shapes, row-tile kernel
against the tiled kernel at several persistent grid sizes (yolo_set_tuning 'dw_tiled')"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, _lib
import ctypes as C
dev = torch.device('cuda:0')
lib = _lib.load()
for (N, H, W, f) in ((32, 104, 104, 64), (32, 52, 52, 128), (32, 26, 26, 256), (32, 13, 13, 512)):
    split = [0, f // 2, 3 * f // 4, 7 * f // 8, f]
    ks = [3, 5, 7, 9]
    p = ops.mix_problem(N, H, W, f, split, ks)
    x = torch.randn(N, H, W, f, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    w = [torch.randn(k, k, split[i + 1] - split[i], device=dev).to(torch.bfloat16) for i, k in enumerate(ks)]
    res = []
    for tiled in (0, 256, 512, 768, 1024, 2048):
        ops.set_tuning('dw_tiled', tiled)
        for bits in (0,):
            def run():
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                lib.yolo_dwconv_mix_dgrad(C.byref(p), x.data_ptr(), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(), y.data_ptr(), bits, st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            res.append('%s%s %.1f' % ('tiled%d' % tiled if tiled else 'rows', '' if not bits else ' -%s' % '-'.join(n for b, n in ((0x10, 'compute'), (0x20, 'stage'), (0x40, 'store')) if bits & b), e0.elapsed_time(e1) * 50))
    print((N, H, W, f), ' | '.join(res), 'us', flush=True)
ops.set_tuning('dw_tiled', 1)
