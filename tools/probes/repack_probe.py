import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
model, loss, opt, grids = bench.build_model('resnet-18', 416, 416, 32, 80, dev)
g = model.g
for _ in range(3): g.refresh_dgrad_weights()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): g.refresh_dgrad_weights()
b.record(); torch.cuda.synchronize()
print('repack_dgrad_batched %.1f us' % (a.elapsed_time(b) * 50))
