"""localise a statistics difference between the plain-rows and the two-level-rows plans: per BatchNorm unit, mean / rstd after one forward +
backward pass on the same weights and batch (YOLO_ROW_GROUPS=0 vs 1)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from yolov3_tensorflow_amd import engine
dev = torch.device('cuda:0')
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 128
out = {}
for mode in ('0', '1'):
    os.environ['YOLO_ROW_GROUPS'] = mode
    model, loss, opt, grids = bench.build_model('resnet-18', S, S, N, 80, dev)
    images, labels = bench.synthetic_batch(N, S, S, 80, 0)
    model.stage_batch(images, labels)
    model.overlap_wgrad = False
    model.g.training = True
    model._fwd_bwd()
    torch.cuda.synchronize()
    rec = []
    for op in model.g.tape:
        if isinstance(op, engine.ApplyOp) and op.m_bn is not None:
            mb = op.m_bn
            src = op.m_src
            rows = getattr(src, 'stat_groups', None)
            rec.append((getattr(getattr(src, 'wp', None), 'name', '?'), op.M, op.C, rows, getattr(src, 'stat_grouped', None), mb.mean.cpu().numpy().copy(), mb.rstd.cpu().numpy().copy(),
                        mb.k1.cpu().numpy().copy() if hasattr(mb, 'k1') else None, getattr(op, 'frows', None)))
    out[mode] = (float(loss.total.item()), rec, model.g.ps.grad.detach().float().cpu().numpy().copy())
    del model, loss, opt
print('loss plain %.6f grouped %.6f' % (out['0'][0], out['1'][0]))
g0, g1 = out['0'][2], out['1'][2]
print('gradient rel L2 diff %.3e' % (np.linalg.norm(g1 - g0) / np.linalg.norm(g0)))
for a, b in zip(out['0'][1], out['1'][1]):
    dm = np.abs(a[5] - b[5]).max() / (np.abs(a[5]).max() + 1e-12)
    dr = np.abs(a[6] - b[6]).max() / (np.abs(a[6]).max() + 1e-12)
    dk = (np.abs(a[7] - b[7]).max() / (np.abs(a[7]).max() + 1e-12)) if a[7] is not None else -1
    print('%-22s M %7d C %4d rows %5s -> %4s grouped %-5s | mean %.2e rstd %.2e | k1 %.2e (bwd rows %s -> %s)' % (a[0][:22], a[1], a[2], a[3], b[3], b[4], dm, dr, dk, a[8], b[8]))
