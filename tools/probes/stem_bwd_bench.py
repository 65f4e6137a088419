"""stem backward at 416^2 / batch 32: two-kernel path (bn_pool_bwd_apply + conv2d_wgrad_slabs) against stem_pool_bwd_wgrad, alone on the GPU"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov3_tensorflow_amd import ops, backend
dev = torch.device('cuda:0')
N, Hi, Wi = 32, 416, 416
ACT = backend.torch_dtype()
p = ops.conv_problem(N, Hi, Wi, 8, 64, 3, 2, 'same')
H, W, C = p.Ho, p.Wo, 64
g = torch.Generator().manual_seed(1)
x = torch.zeros(N, Hi, Wi, 8); x[..., :3] = torch.rand(N, Hi, Wi, 3, generator=g); x = x.to(ACT).to(dev)
y = torch.randn(N, H, W, C, generator=g).to(ACT).to(dev)
Ho, Wo, pt, pl = H // 2, W // 2, 0, 0
sc = (torch.rand(C, generator=g) + 0.5).to(dev)
sh, mean, rstd, k1, k2 = [(torch.randn(C, generator=g) * s).to(dev) for s in (0.1, 0.2, 1.0, 0.05, 0.05)]
out = torch.empty(N, Ho, Wo, C, dtype=ACT, device=dev); arg = torch.empty(N, Ho, Wo, C, dtype=torch.uint8, device=dev)
ops.bn_pool_fwd(y, sc, sh, out, arg, N, H, W, C, Ho, Wo, pt, pl, True)
dout = torch.randn(N, Ho, Wo, C, generator=g).to(ACT).to(dev)
dy = torch.empty(N, H, W, C, dtype=ACT, device=dev)
splits = ops.conv2d_wgrad_splits(p)
dw = torch.zeros(64, 3, 3, 8, device=dev)
slabs_old = torch.zeros(max(splits, 1) * dw.numel(), device=dev)
n = ops.stem_pool_bwd_slabs(p, C, Ho, Wo, pt, pl)
slabs = torch.zeros(n, 64, 3, 3, 8, device=dev)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

t_apply = timeit(lambda: ops.bn_pool_bwd_apply(dout, out, arg, True, y, sc, mean, rstd, k1, k2, dy, N, H, W, C, Ho, Wo, pt, pl))
t_wgrad = timeit(lambda: ops.conv2d_wgrad_slabs(p, x, dy, dw, slabs_old if splits > 1 else None))
t_fused = timeit(lambda: ops.stem_pool_bwd_wgrad(p, x, dout, out, arg, True, y, sc, mean, rstd, k1, k2, Ho, Wo, pt, pl, slabs))
n4 = dw.numel() // 4
tab = torch.tensor([[0, 0, n4, n, 0]], dtype=torch.int64, device=dev)
grads = torch.zeros(dw.numel(), device=dev)
t_red = timeit(lambda: ops.wgrad_reduce_batched(tab, 1, ops.reduce_blocks(n4, n), slabs.view(-1), grads))
print('splits old %d, slabs new %d' % (splits, n))
print('apply %.1f us  wgrad %.1f us  | fused %.1f us  reduce(%d slabs) %.1f us' % (t_apply, t_wgrad, t_fused, n, t_red))
