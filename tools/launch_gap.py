#!/usr/bin/env python
"""Per-launch floor of a dependent chain of tiny kernels on one stream: eager vs hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_tensorflow_amd import ops
dev = torch.device('cuda:0')
x = torch.zeros(256, device=dev)
out = torch.zeros(8, device=dev)
N = 2000
def chain(n):
    for _ in range(n):
        ops.cast_f32_to_bf16(x, y, 256)
y = torch.zeros(256, dtype=torch.bfloat16, device=dev)
chain(10); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); chain(N); e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
print('eager: device %.2f us/launch, host enqueue %.2f us/launch' % (e0.elapsed_time(e1) * 1e3 / N, (t1 - t0) * 1e6 / N))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        chain(N)
torch.cuda.current_stream().wait_stream(s)
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print('graph: device %.2f us/launch' % (e0.elapsed_time(e1) * 1e3 / N))
# larger tensor: 5.5 MB (13x13x512x32 bf16) elementwise
x2 = torch.zeros(32 * 169 * 512, device=dev); y2 = torch.zeros(32 * 169 * 512, dtype=torch.bfloat16, device=dev)
def chain2(n):
    for _ in range(n):
        ops.cast_f32_to_bf16(x2, y2, x2.numel())
chain2(10); torch.cuda.synchronize()
e0.record(); chain2(500); e1.record(); torch.cuda.synchronize()
print('eager 2.7M-element cast: %.2f us/launch' % (e0.elapsed_time(e1) * 1e3 / 500))
