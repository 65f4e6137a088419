#!/usr/bin/env python
"""CPU-only ablation behind DESIGN.md "precision": which 16-bit storage point of the product moves the training loss of BASELINE.json
configs[0] (ResNet18-YOLOv3 320x320, the reference's sample set, batch 2, lr 1e-5) away from the float32 oracle?  The oracle emulates the
product's storage points (oracle/nets.py `_r`); each variant switches the rounding of one class of tensors off and the loss curve is
compared with the plain float32 run.  Test infrastructure (imports oracle/); nothing here runs on the product path.
Usage: python tools/precision_ablation.py [--steps 8]"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets
from oracle.train import OracleTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--plateau-after', type=int, default=10**9, help='switch the rate from 1e-5 to 1e-3 after this many steps')
ap.add_argument('--variants', default='all,none_w,none_act,none_head_w,none_head_in,w_hi_lo_heads')
a = ap.parse_args()
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'sample20_320.npz'))
images = (z['images_rgb_u8'].astype(np.float32) / 255.0)[..., ::-1].copy()
labels = z['labels']
H = W = 320
N, Cn = 2, 13
ANCHORS = [[(0.06618181818181816, 0.1025177510694752), (0.18544278606965178, 0.13160367921287464), (0.13, 0.32733333333333337)],
           [(0.13, 0.32733333333333337), (0.303806787732042, 0.34370030784316496)],
           [(0.303806787732042, 0.34370030784316496), (0.4667050847457627, 0.5281262429095761), (0.7906945888923907, 0.7888860433597275)]]
LW = [(5, 5, 0.05, 3, 1), (8, 8, 0.05, 2, 1), (10, 10, 0.05, 2, 1)]
grids = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]

POLICY = {'mode': 'all'}
orig_r = nets.Graph._r
orig_conv = nets.Graph.convolution


def patched_r(self, x, kind='act'):
    m = POLICY['mode']
    if self.round_fn is None:
        return x
    if m == 'none_w' and kind == 'w':
        return x
    if m.startswith('w_keep_') and kind == 'w':          # w_keep_<lo>_<hi>: weights of conv number lo..hi-1 (creation order) stay float32
        lo, hi = [int(t) for t in m.split('_')[2:4]]
        k = POLICY['conv_index']
        if lo <= k < hi:
            return x
    if m == 'none_act' and kind == 'act':
        return x
    if m in ('none_head_w', 'w_hi_lo_heads') and kind == 'w' and POLICY.get('in_head'):
        return x
    return orig_r(self, x, kind)


def patched_conv(self, x, filters, kernel_size=(3, 3), strides=(1, 1), padding='same', use_bias=False, name=None, init='he_normal'):
    POLICY['in_head'] = bool(use_bias)              # the three detection convs are the only ones with a bias
    POLICY['conv_index'] = self.replay['conv2d'] + self.replay['yolov3_head'] if self.replay is not None else 0
    if use_bias and POLICY['mode'] == 'none_head_in' and self.round_fn is not None:
        x = POLICY['last_unrounded']                # the detection conv reads the float32 activation
    try:
        return orig_conv(self, x, filters, kernel_size, strides, padding, use_bias, name, init)
    finally:
        POLICY['in_head'] = False


orig_act = nets.Graph.activation


def patched_act(self, x):
    POLICY['last_unrounded'] = torch.relu(x)
    return orig_act(self, x)


nets.Graph._r = patched_r
nets.Graph.convolution = patched_conv
nets.Graph.activation = patched_act


def run(mode, emulate):
    POLICY['mode'] = mode
    o = OracleTrainer('resnet-18', grids, Cn, ANCHORS, 0.8, LW, rectified_coord_num=1464, rectified_loss_weight=[1.0, 1.0, 1.0], lr=1e-5,
                      emulate_bf16=emulate)
    o.ensure_params(images[:N])
    out = []
    for step in range(a.steps):
        i = (step * N) % 20
        o.opt.lr = 1e-5 if step < a.plateau_after else 1e-3
        out.append(o.step(images[i:i + N], labels[i:i + N])[0])
    return np.array(out)


ref = run('all', False)
print('float32       ', np.round(ref, 3))
for v in a.variants.split(','):
    cur = run(v, True)
    rel = np.abs(cur - ref) / np.abs(ref)
    print('%-14s max %.2e median %.2e  per step %s' % (v, rel.max(), np.median(rel), ' '.join('%.1e' % r for r in rel)), flush=True)
