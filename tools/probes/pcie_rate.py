"""PCIe-inclusive rates of the headline step (bench.py's `value` starts with the batch resident in HBM): host float32 images handed to
train_on_batch-style staging (pageable and pinned), and uint8 images as the input pipeline uploads them (dataset/file_util.py)"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
N = 32
model, loss, opt, grids = bench.build_model('resnet-18', 416, 416, N, 80, dev)
images, labels = bench.synthetic_batch(N, 416, 416, 80, 0)
img_np = np.ascontiguousarray(images.numpy() if torch.is_tensor(images) else images, dtype=np.float32)
lab = labels
img_pin = torch.from_numpy(img_np).pin_memory()
img_u8 = (torch.from_numpy(img_np) * 255).to(torch.uint8).pin_memory()
u8_dev = torch.empty_like(img_u8, device=dev)
model.stage_batch(img_pin, lab)
for _ in range(8):
    model.run_step()
torch.cuda.synchronize()


def rate(fn, steps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return N / dt, dt * 1e3


def copy_only(src, dst):
    def f():
        dst.copy_(src, non_blocking=True)
    return f


print('resident batch (bench.py value)      : %8.1f images/s  %.3f ms/step' % rate(model.run_step))
print('H2D float32 pinned, 66 MB, alone      : %8.1f images/s  %.3f ms' % rate(copy_only(img_pin, model.g.images)))
print('H2D uint8 pinned, 16.6 MB, alone      : %8.1f images/s  %.3f ms' % rate(copy_only(img_u8, u8_dev)))


def step_pageable():
    model.stage_batch(img_np, lab); model.run_step()


def step_pinned():
    model.stage_batch(img_pin, lab); model.run_step()


def step_u8():
    u8_dev.copy_(img_u8, non_blocking=True)
    model.g.images.copy_(u8_dev)           # uint8 -> float32 on the device (the pipeline's letterbox kernel writes float32 the same way)
    model.g.images.mul_(1.0 / 255.0)
    model.run_step()


print('float32 pageable numpy + step (serial): %8.1f images/s  %.3f ms/step' % rate(step_pageable, 10))
print('float32 pinned + step (one stream)    : %8.1f images/s  %.3f ms/step' % rate(step_pinned))
print('uint8 pinned + convert + step         : %8.1f images/s  %.3f ms/step' % rate(step_u8))
copy_stream = torch.cuda.Stream(device=dev)
bufs = [torch.empty_like(img_u8, device=dev) for _ in range(2)]
evs = [torch.cuda.Event() for _ in range(2)]
state = {'i': 0}


def step_u8_prefetch():
    i = state['i']; state['i'] = i ^ 1
    with torch.cuda.stream(copy_stream):                       # next batch uploads beside this step
        bufs[i ^ 1].copy_(img_u8, non_blocking=True)
        evs[i ^ 1].record(copy_stream)
    torch.cuda.current_stream().wait_event(evs[i])
    model.g.images.copy_(bufs[i]); model.g.images.mul_(1.0 / 255.0)
    model.run_step()


with torch.cuda.stream(copy_stream):
    bufs[0].copy_(img_u8, non_blocking=True); evs[0].record(copy_stream)
print('uint8 pinned, upload on a copy stream : %8.1f images/s  %.3f ms/step' % rate(step_u8_prefetch))
