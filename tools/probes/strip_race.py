"""Repeat the 52x52x128 forward (bf16 strip kernel with statistics vs float32 implicit-GEMM output) and locate disagreeing elements."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops
dev = torch.device('cuda:0')
H, W, Cin, Cout, k, s, pad = 52, 52, 128, 128, 3, 1, 'same'
N = 32
g = torch.Generator(device='cpu').manual_seed(3)
p = ops.conv_problem(N, H, W, Cin, Cout, k, s, pad)
x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(k * k * Cin)).to(torch.bfloat16).to(dev)
rows = ops.conv2d_stat_rows(p)
ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
y32 = torch.empty(N, H, W, Cout, dtype=torch.float32, device=dev)
ops.conv2d_fwd(p, x, w, y32)
torch.cuda.synchronize()
ref = y32.clone()
bad_total = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    ybf = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.conv2d_fwd(p, x, w, ybf, stat_sum=ss, stat_sq=sq)
    y32.zero_()
    ops.conv2d_fwd(p, x, w, y32)
    torch.cuda.synchronize()
    d = (ybf.float() - ref).abs()
    nb = int((~(d <= 0.05)).sum())
    d32 = (y32 - ref).abs()
    nb32 = int((d32 > 1e-3).sum())
    if nb or nb32:
        bad_total += 1
        idx = torch.nonzero(~(d <= 0.05))[:6].tolist()
        print('iter', it, 'strip bad elements', nb, 'first', idx, 'igemm-f32 bad', nb32, torch.nonzero(d32 > 1e-3)[:4].tolist())
        if nb:
            ii = torch.nonzero(~(d <= 0.05))
            print('   bad n range', int(ii[:, 0].min()), int(ii[:, 0].max()), 'h', int(ii[:, 1].min()), int(ii[:, 1].max()), 'w', int(ii[:, 2].min()), int(ii[:, 2].max()),
                  'c', int(ii[:, 3].min()), int(ii[:, 3].max()))
            lin = (ii[:, 0] * H + ii[:, 1]) * W + ii[:, 2]
            print('   linear pixel range', int(lin.min()), int(lin.max()), 'tiles(128)', sorted(set((lin // 128).tolist()))[:10])
print('iterations with errors:', bad_total)
