#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the ResNet18-YOLOv3 layer shapes (batch 32, 416x416).  Usage:
    python tools/conv_bench.py [--layers 64,128,256,512,stem] [--passes fwd,dgrad,wgrad,wgrad2]  (wgrad = one-pass atomics, wgrad2 = two-phase slabs + reduce, the training path) [--iters 30]"""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov3_tensorflow_amd import ops

LAYERS = {   # name: (H, W, Cin, Cout, k, stride, padding)
    'stem': (416, 416, 8, 64, 3, 2, 'same'),
    '64': (104, 104, 64, 64, 3, 1, 'same'),
    '128': (52, 52, 128, 128, 3, 1, 'same'),
    '256': (26, 26, 256, 256, 3, 1, 'same'),
    '512': (13, 13, 512, 512, 3, 1, 'same'),
    '128s2': (104, 104, 64, 128, 3, 2, 'same'),
    '256s2': (52, 52, 128, 256, 3, 2, 'same'),
    '512s2': (26, 26, 256, 512, 3, 2, 'same'),
    '1x1': (52, 52, 256, 128, 1, 1, 'same'),
    'h13': (13, 13, 512, 256, 3, 1, 'same'),
    'h26': (26, 26, 256, 512, 3, 1, 'same'),
    'h52': (52, 52, 128, 256, 3, 1, 'same'),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='64,128,256,512')
    ap.add_argument('--passes', default='fwd,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    N = a.batch
    for kv in filter(None, os.environ.get('YOLO_TUNE', '').split(',')):      # e.g. YOLO_TUNE=s2_classes=0,wgrad_strip=0
        k, v = kv.split('=')
        ops.set_tuning(k, int(v))
    for name in a.layers.split(','):
        H, W, Cin, Cout, k, s, pad = LAYERS[name]
        p = ops.conv_problem(N, H, W, Cin, Cout, k, s, pad)
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.05).to(torch.bfloat16)
        wd = torch.empty(Cin, k, k, Cout, dtype=torch.bfloat16, device=dev)
        ops.repack_dgrad_weights(w, wd, Cout, k, k, Cin)
        y = torch.empty(N, p.Ho, p.Wo, Cout, dtype=torch.bfloat16, device=dev)
        dy = torch.randn(N, p.Ho, p.Wo, Cout, device=dev).to(torch.bfloat16)
        dx = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=dev)
        dw = torch.zeros((160 if os.environ.get('YOLO_WGRAD_EPI') == '3' else 1) * Cout, k, k, Cin, device=dev)
        rows = ops.conv2d_stat_rows(p)
        wsp = torch.empty(max(ops.conv2d_wgrad_workspace_bytes(p), 16) // 4, device=dev)
        ss, sq = torch.zeros(rows, Cout, device=dev), torch.zeros(rows, Cout, device=dev)
        flops = 2.0 * N * p.Ho * p.Wo * Cout * Cin * k * k
        fns = {'fwd': lambda: ops.conv2d_fwd(p, x, w, y, stat_sum=ss, stat_sq=sq),
               'dgrad': lambda: ops.conv2d_dgrad(p, dy, wd, dx),
               'wgrad': lambda: ops.conv2d_wgrad(p, x, dy, dw),
               'wgrad2': lambda: ops.conv2d_wgrad_reduce(p, x, dy, dw, wsp)}
        for ps in a.passes.split(','):
            if ps == 'dgrad' and Cin % 64:
                continue
            f = fns[ps]
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            print('%-6s %-6s %8.1f us  %7.1f TFLOP/s' % (name, ps, us, flops / us / 1e6), flush=True)


if __name__ == '__main__':
    main()
