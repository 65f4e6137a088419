"""Where does the bf16 + focal-loss case of tests/test_train_step_gpu.py::test_forward_loss_grads_and_step differ from the bf16-EMULATING
oracle by 1.1e-3 of the loss?  Both round at the same storage points; the convolution sums run in different
orders, so the two sets of logits differ by rounding noise amplified through ~20 BatchNorm layers.  This tool evaluates the ORACLE's loss
on both sets of logits (the GPU's loss kernel agrees with the oracle on the GPU's own logits to 1e-4, asserted by the test) and prints:
the (5 terms x 3 heads) table of both, whether any ground truth changed its responsible (head, cell, anchor), and the cells that carry
the difference of the dominant term.  Output: JSON on stdout (committed as profiles/r03_focal_gap.json).

usage (GPU box): python tools/focal_gap.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_train_step_gpu as T          # the test's own builders: same model, batch, anchors, weights
from oracle.train import OracleTrainer

backbone, rect, focal = 'resnet-18', -1, True
H = W = 224
N, Tn, Cn = 4, 4, 13
model, loss, opt, grids = T.build(backbone, H, W, N, Cn, rect=rect, focal=focal)
images, labels = T.make_batch(N, H, W, Tn, Cn, seed=3)
w0 = model.get_weights()
model.use_hip_graph = False
fk = dict(is_focal_loss=True, focal_alpha=1.0, focal_gamma=2.0)
o = OracleTrainer(backbone, grids, Cn, T.ANCHORS, 0.5, T.LOSS_W, rectified_coord_num=rect, rectified_loss_weight=[1.0, 1.0, 1.0],
                  emulate_bf16=True, emulate_bf16_grads=False, **fk)
o.ensure_params(images)
o.set_weights(w0)
model.stage_batch(images, labels)
model.g.training = True
model._fwd_bwd()
torch.cuda.synchronize()
heads_gpu = [h.buf[..., :c].float().cpu() for h, c in zip(model.heads, model.head_channel_nums)]
loss_gpu = float(loss.total.item())
with torch.no_grad():
    heads_orc, yolo, l2 = o.forward_loss(images, labels)
loss_orc = float(yolo.item())


def evaluate(heads):
    chk = OracleTrainer(backbone, grids, Cn, T.ANCHORS, 0.5, T.LOSS_W, rectified_coord_num=rect, rectified_loss_weight=[1.0, 1.0, 1.0], **fk)
    raw = [h.reshape(N, h.shape[1], h.shape[2], len(a), 5 + Cn) for h, a in zip(heads, T.ANCHORS)]
    with torch.no_grad():
        total = float(chk.loss.loss_heads(torch.as_tensor(labels), raw).item())
    L = chk.loss
    # per-cell no-object terms (yolov3_loss.py:331-338) recomputed from the oracle's own pieces
    dec = L.decoder.decode_heads(raw)
    tgt = L.label_decoder.decode(torch.as_tensor(labels).to(L.dtype))
    cells = []
    for h in range(3):
        per = []
        for i in range(N):
            valid = tgt[0][0][i][:, 0] >= 0
            t, b = tgt[h][0][i][valid], tgt[h][1][i][valid]
            max_iou, rmax, grid_xyz = L._calc_iou(t, b, dec[h][1][i], dec[h][2][i], L.grid[h])
            obj = torch.zeros(L.grid[h][0], L.grid[h][1], L.box_num[h])
            for (r, c, a) in L.last_assign[i][h].tolist():
                obj[r, c, a] = 1.0
            bg = (max_iou < L.iou_thresh).float() * (1 - obj)
            p = dec[h][1][i][..., 4]
            per.append(float(L.w_noobj[h]) * (-torch.log(1 - p)) * p.pow(2.0) * bg / N)
        cells.append(torch.stack(per))
    # per responsible prediction: its wh term (yolov3_loss.py:350, 358-359), keyed by (image, head, ground truth)
    wh = {}
    for i in range(N):
        valid = tgt[0][0][i][:, 0] >= 0
        per_head = []
        for h in range(3):
            t, b = tgt[h][0][i][valid], tgt[h][1][i][valid]
            per_head.append((t,) + tuple(L._calc_iou(t, b, dec[h][1][i], dec[h][2][i], L.grid[h])))
        r = [ph[2] for ph in per_head]
        pos = [(r[0] >= r[1]) & (r[0] >= r[2]), (r[1] >= r[0]) & (r[1] >= r[2]), (r[2] >= r[0]) & (r[2] >= r[1])]
        for h in range(3):
            t, _, _, gxyz = per_head[h]
            Hh, Wh = L.grid[h]
            for k in torch.nonzero(pos[h]).flatten().tolist():
                row, col, a_ = gxyz[k].tolist()
                rp = dec[h][1][i][row, col, a_]
                scale = 2 - t[k, 2] * t[k, 3] / (Hh * Wh)
                v = float(L.w_wh[h]) * float(scale) * float(torch.square(torch.log(t[k, 2:4]) - torch.log(rp[2:4])).sum()) / N
                wh[(i, h, k)] = {'image': i, 'head': ['/8', '/16', '/32'][h], 'ground_truth': k, 'row': row, 'col': col, 'anchor': a_,
                                 'raw_wh_logits': [float(x) for x in raw[h][i, row, col, a_, 2:4]], 'wh_term': v}
    return total, L.per_image.clone(), [[a.tolist() for a in img] for img in L.last_assign], cells, wh


tg, pg, ag, cg, wg = evaluate(heads_gpu)
to, po, ao, co, wo = evaluate([h.detach() for h in heads_orc])
terms = ['xy', 'wh', 'noobj', 'obj', 'class']
dtab = (pg - po).mean(0)                       # (5, 3): contribution of every (term, head) to the difference of the batch-mean loss
k = int(dtab.abs().argmax())
term, head = k // 3, k % 3
out = {
    'case': 'resnet-18 224x224 batch 4, 13 classes, focal alpha 1 gamma 2, bf16 build vs bf16-emulating oracle (test_forward_loss_grads_and_step)',
    'loss_gpu_kernel': loss_gpu, 'oracle_loss_on_gpu_logits': tg, 'oracle_loss_on_emulated_logits': to, 'emulating_oracle_loss': loss_orc,
    'relative_gap': abs(tg - to) / abs(to),
    'logits_rel_l2_per_head': [T.rel_l2(a.numpy(), b.detach().numpy()) for a, b in zip(heads_gpu, heads_orc)],
    'same_responsible_cells': ag == ao,
    'term_table_gpu_logits': pg.mean(0).tolist(), 'term_table_emulated_logits': po.mean(0).tolist(), 'difference_rows_xy_wh_noobj_obj_class': dtab.tolist(),
    'dominant': {'term': terms[term], 'head': ['/8', '/16', '/32'][head], 'difference': float(dtab[term, head]), 'share_of_gap': float(dtab[term, head] / (tg - to))},
}
if terms[term] == 'wh':
    rows = []
    for key in sorted(wg):
        e = dict(wg[key])
        e['raw_wh_logits_emulated'] = wo[key]['raw_wh_logits']
        e['wh_term_emulated'] = wo[key]['wh_term']
        e['difference'] = e['wh_term'] - wo[key]['wh_term']
        rows.append(e)
    rows.sort(key=lambda e: -abs(e['difference']))
    out['dominant']['responsible_predictions'] = len(rows)
    out['dominant']['sum_of_wh_differences_all_heads'] = sum(e['difference'] for e in rows)
    out['dominant']['largest_contributions'] = rows[:8]
    out['reading'] = ('no ground truth changes its responsible (head, cell, anchor); the gap is the squared-log wh term (quadratic in the two raw wh '
                      'logits) at the %d responsible predictions, whose logits differ by the ~2 %% relative L2 that rounding-order noise reaches '
                      'after ~20 BatchNorm layers; the focal factor only shrinks the no-object term, i.e. the denominator of the relative gap'
                      % len(rows))
if terms[term] == 'noobj':
    d = (cg[head] - co[head])
    flat = d.abs().flatten()
    top = torch.topk(flat, 10).indices
    shape = d.shape
    rows = []
    for idx in top.tolist():
        n, r, c, a = np.unravel_index(idx, shape)
        pgc = float(torch.sigmoid(heads_gpu[head].reshape(N, shape[1], shape[2], shape[3], 5 + Cn)[n, r, c, a, 4]))
        poc = float(torch.sigmoid(heads_orc[head].detach().reshape(N, shape[1], shape[2], shape[3], 5 + Cn)[n, r, c, a, 4]))
        rows.append({'image': int(n), 'row': int(r), 'col': int(c), 'anchor': int(a), 'conf_gpu': pgc, 'conf_emulated': poc, 'term_difference': float(d[n, r, c, a])})
    out['dominant']['cells_total'] = int(flat.numel())
    out['dominant']['sum_of_all_cell_differences'] = float(d.sum())
    out['dominant']['top10_cells'] = rows
    out['dominant']['share_of_top10'] = float(sum(x['term_difference'] for x in rows) / float(d.sum())) if float(d.sum()) != 0 else None
    # how is the difference spread: the cells sorted by |difference|
    srt = torch.sort(flat, descending=True).values
    cs = torch.cumsum(srt, 0) / srt.sum()
    out['dominant']['cells_for_half_of_abs_difference'] = int((cs < 0.5).sum()) + 1
print(json.dumps(out, indent=1))
