"""long run on one synthetic batch: loss and the optimizer's non-finite counter every 100 steps (where does a fixed-batch run leave float32?)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
if len(sys.argv) > 2:                       # K.epsilon (reference run.py:26 sets 1e-8, whose 1 - eps is 1.0f: no upper clip); 1e-7 makes the clip real
    from yolov3_tensorflow_amd import backend
    backend.set_epsilon(float(sys.argv[2]))
model, loss, opt, grids = bench.build_model('resnet-18', 416, 416, 32, 80, dev)
images, labels = bench.synthetic_batch(32, 416, 416, 80, 0)
model.stage_batch(images, labels)
for s in range(1, steps + 1):
    model.run_step()
    if s % 100 == 0 or s in (1, 10, 50):
        torch.cuda.synchronize()
        nf = int(model.optimizer.nonfinite.item())
        t = model.loss_obj.terms.detach().cpu().numpy()
        print('step %5d loss %12.5f  nonfinite waves %d  terms xy %s wh %s noobj %s obj %s cls %s' % (
            s, float(model.loss_value.item()), nf, t[0].round(3), t[1].round(3), t[2].round(3), t[3].round(3), t[4].round(3)), flush=True)
        if nf:
            break
