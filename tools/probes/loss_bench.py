"""the loss launch (assign + main + finalize) of the benchmark model timed alone on the logits of a real step"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
model, loss, opt, grids = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else 'resnet-18', 416, 416, 32, 80, dev)
images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
model.stage_batch(images, labels)
for _ in range(3):
    model._fwd_bwd(); model._update()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    a.record()
    for _ in range(20):
        model.loss_obj.launch(model)
    b.record(); torch.cuda.synchronize()
    print('loss launch (3 kernels): %.1f us' % (a.elapsed_time(b) * 50.0))
