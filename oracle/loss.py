"""Oracle (test infrastructure): YOLOv3 decoder, label decoder and loss, restated on PyTorch-CPU.

Follows, line by line:
  /root/reference/yolov3/yolov3_decoder.py:62-192   (un-merge + decode)
  /root/reference/yolov3/label_decoder.py:26-60     (label decode)
  /root/reference/yolov3/yolov3_loss.py:81-369      (loss)
``K.epsilon()`` is 1e-8 (/root/reference/run.py:26).  All arithmetic is float32 unless ``dtype=torch.float64``.
Gradients come from torch autograd (the reference relies on TF autodiff); the hand-derived backward in
SURVEY.md Appendix A (what the HIP kernel implements) is checked against this in tests/.

parity unpinned: the reference has no tests / golden vectors for this path and TensorFlow is not installed, so the
TF op semantics used here (clip_by_value gradient, gather_nd, sparse_to_dense with duplicates, reduce_max over an
empty axis = -inf, arg_max first-index tie-break) are taken from TF documentation.
Divergence (documented): a GT centre exactly on the right/bottom border makes TF's gather_nd fail on CPU
(yolov3_loss.py:269-271); here the cell index is clamped into the grid, as the HIP kernel does.
"""
import numpy as np
import torch

EPS = 1e-8  # /root/reference/run.py:26


def merge_heads(head8, head16, head32):
    """yolov3_detector.py:80-85 -- flat row-major reshape of the /8 and /16 heads onto the /32 grid + concat.
    heads are (N, H, W, C) tensors."""
    n, h32, w32, _ = head32.shape
    return torch.cat([head8.reshape(n, h32, w32, -1), head16.reshape(n, h32, w32, -1), head32], dim=-1)


class YOLOv3DecoderOracle(object):
    """yolov3_decoder.py:12-192"""

    def __init__(self, head_grid_sizes, class_num, anchor_boxes, dtype=torch.float32):
        self.dtype = dtype
        self.grid = [(int(h), int(w)) for (h, w) in head_grid_sizes]                 # [H, W]  (:22-25)
        self.box_num = [len(a) for a in anchor_boxes]                                # (:32-34)
        # anchors scaled to grid units [W, H]                                          (:38-40)
        self.anchors = [torch.tensor(np.asarray(a, dtype=np.float32), dtype=torch.float32).to(dtype)
                        * torch.tensor([w, h], dtype=dtype)
                        for a, (h, w) in zip(anchor_boxes, self.grid)]
        self.class_num = class_num
        self.box_len = 4 + 1 + class_num                                             # (:45)
        self.left_top = [self._get_left_top(w, h) for (h, w) in self.grid]            # (:27-29)

    def _get_left_top(self, width, height):
        """(:47-60) meshgrid of cell (x, y), shape (H, W, 1, 2)"""
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing='ij')
        return torch.stack([xs, ys], dim=-1).reshape(height, width, 1, 2).to(self.dtype)

    def unpack(self, predicts):
        """(:89-117) (N, H32, W32, C) -> 3 x (N, H, W, B, L)"""
        out, start = [], 0
        mult = [16, 4, 1]
        for (h, w), b, m in zip(self.grid, self.box_num, mult):
            end = start + b * self.box_len * m
            out.append(predicts[..., start:end].reshape(-1, h, w, b, self.box_len))
            start = end
        return out

    def decode_heads(self, raw_heads):
        """(:119-192) raw_heads: 3 x (N,H,W,B,L) -> [(raw_txywh, decoded, boxes)] x 3"""
        eps_lo = torch.tensor(EPS, dtype=torch.float32).to(self.dtype)
        eps_hi = torch.tensor(np.float32(1 - EPS) if self.dtype == torch.float32 else 1 - EPS, dtype=self.dtype)
        res = []
        for raw, lt, anc in zip(raw_heads, self.left_top, self.anchors):
            xy = torch.sigmoid(raw[..., 0:2])                                         # (:153)
            xy = torch.clamp(xy, eps_lo, eps_hi) + lt                                 # (:154-155)
            wh = torch.exp(raw[..., 2:4]) * anc                                       # (:167-168)
            conf = torch.clamp(torch.sigmoid(raw[..., 4:5]), eps_lo, eps_hi)          # (:178-179)
            if self.class_num >= 1:
                mx = raw[..., 5:].max(dim=-1, keepdim=True)[0]                        # (:189)
                prob = torch.softmax(raw[..., 5:] - mx, dim=-1)                       # (:190)
                prob = torch.clamp(prob, eps_lo, eps_hi)                              # (:191)
                dec = torch.cat([xy, wh, conf, prob], dim=-1)                         # (:132)
            else:
                dec = torch.cat([xy, wh, conf], dim=-1)                               # (:134)
            half = wh / 2
            boxes = torch.cat([xy - half, xy + half], dim=-1)                         # (:137-139)
            res.append((raw[..., 0:4], dec, boxes))
        return res

    def decode(self, predicts):
        """(:62-87)"""
        return self.decode_heads(self.unpack(predicts))


class LabelDecoderOracle(object):
    """label_decoder.py:11-60"""

    def __init__(self, head_grid_sizes, dtype=torch.float32):
        self.wh = [torch.tensor([float(w), float(h)], dtype=dtype) for (h, w) in head_grid_sizes]   # (:21-23)

    def decode(self, targets):
        targets = targets.reshape(targets.shape[0], -1, 5)                           # (:35)
        out = []
        for wh in self.wh:
            xy = targets[:, :, 0:2] * wh                                             # (:53)
            twh = targets[:, :, 2:4] * wh                                            # (:54)
            t = torch.cat([xy, twh, targets[:, :, 4:5]], dim=-1)                     # (:56)
            half = twh / 2
            out.append((t, torch.cat([xy - half, xy + half], dim=-1)))               # (:58-59)
        return out


class YOLOv3LossOracle(object):
    """yolov3_loss.py:13-369.  State: ``current_num`` (the rectified-image counter, :69) and the six (3,) logging
    vectors (:72-79)."""

    def __init__(self, head_grid_sizes, class_num, anchor_boxes, iou_thresh, loss_weights,
                 rectified_coord_num=0, rectified_loss_weight=None,
                 is_focal_loss=False, focal_alpha=0.25, focal_gamma=2.0, is_tiou_recall=False,
                 dtype=torch.float32):
        self.dtype = dtype
        self.decoder = YOLOv3DecoderOracle(head_grid_sizes, class_num, anchor_boxes, dtype)
        self.label_decoder = LabelDecoderOracle(head_grid_sizes, dtype)
        self.grid = [(int(h), int(w)) for (h, w) in head_grid_sizes]
        lw = np.transpose(np.asarray(loss_weights, dtype=np.float32))                # (:46-47)
        (self.w_xy, self.w_wh, self.w_noobj, self.w_obj, self.w_cls) = [torch.tensor(r).to(dtype) for r in lw]
        self.box_num = [len(a) for a in anchor_boxes]
        self.class_num = class_num
        self.iou_thresh = iou_thresh
        self.is_focal_loss = is_focal_loss
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        self.is_tiou_recall = is_tiou_recall
        self.rectified_coord_num = rectified_coord_num
        if rectified_loss_weight is None:
            rectified_loss_weight = [0.01, 0.01, 0.01]                               # (:63-64)
        elif len(rectified_loss_weight) != 3:
            raise ValueError('rectified_loss_weight must have length 3')              # (:65-66)
        self.rectified_weight = list(rectified_loss_weight)
        self.current_num = 0                                                         # (:69)
        self.detail = {k: torch.zeros(3, dtype=dtype) for k in
                       ('rectified_coord_loss', 'coord_loss_xy', 'coord_loss_wh', 'noobj_iou_loss', 'obj_iou_loss',
                        'class_loss')}
        self.last_assign = None   # filled by loss(): per image, per head list of (row, col, anchor) responsible cells

    # ------------------------------------------------------------------ (:254-303)
    def _calc_iou(self, target, target_boxes, predict, predict_boxes, hw):
        H, W = hw
        T = target.shape[0]
        predict_area = predict[..., 2] * predict[..., 3]                             # (:267)
        grid_xy = torch.floor(target[:, 0:2]).long()                                 # (:269)
        col = grid_xy[:, 0].clamp(0, W - 1)    # documented divergence: clamp instead of gather_nd failure
        row = grid_xy[:, 1].clamp(0, H - 1)                                          # (:270) reversed -> (row, col)
        response_area = predict_area[row, col]                                       # (:271) (T, B)
        target_area = target[:, 2] * target[:, 3]                                    # (:273)
        pb = predict_boxes.unsqueeze(-2)                                             # (:275) (H,W,B,1,4)
        lt = torch.maximum(pb[..., 0:2], target_boxes[:, 0:2])                       # (:276)
        rb = torch.minimum(pb[..., 2:4], target_boxes[:, 2:4])                       # (:277)
        inter_wh = torch.clamp(rb - lt, min=0)                                       # (:278)
        inter = inter_wh[..., 0] * inter_wh[..., 1]                                  # (:279) (H,W,B,T)
        rboxes = predict_boxes[row, col]                                             # (:281) (T,B,4)
        tb = target_boxes.unsqueeze(1)                                               # (:282)
        rlt = torch.maximum(rboxes[..., 0:2], tb[..., 0:2])                          # (:283)
        rrb = torch.minimum(rboxes[..., 2:4], tb[..., 2:4])                          # (:284)
        rwh = torch.clamp(rrb - rlt, min=0)                                          # (:285)
        rinter = rwh[..., 0] * rwh[..., 1]                                           # (:286) (T,B)
        iou = inter / (predict_area.unsqueeze(-1) + target_area - inter)             # (:289-290)
        if self.is_tiou_recall:
            iou = iou * inter / target_area                                          # (:291-293)
        if T > 0:
            max_iou = iou.max(dim=-1)[0]                                             # (:294)
        else:
            max_iou = torch.full(predict_area.shape, float('-inf'), dtype=self.dtype)  # reduce_max over empty axis
        riou = rinter / (response_area + target_area.unsqueeze(-1) - rinter)         # (:296-297)
        if self.is_tiou_recall:
            riou = riou * rinter / target_area.unsqueeze(-1)                         # (:298-299)
        if T > 0:
            rmax, rarg = riou.max(dim=-1)                                            # (:300-301)
            # tf.arg_max returns the FIRST maximal index; torch.max may not -> recompute explicitly
            rarg = (riou == rmax.unsqueeze(-1)).float().argmax(dim=-1)
        else:
            rmax = torch.zeros(0, dtype=self.dtype)
            rarg = torch.zeros(0, dtype=torch.long)
        grid_xyz = torch.stack([row, col, rarg], dim=-1)                             # (:302)
        return max_iou.detach(), rmax.detach(), grid_xyz

    # ------------------------------------------------------------------ (:305-369)
    def _single_head_loss(self, hi, predict, target, max_iou, rmax, grid_xyz, max_pos, hw, box_num):
        H, W = hw
        grid_xyz = grid_xyz[max_pos]                                                 # (:324-325)
        object_mask = torch.zeros(H, W, box_num, dtype=self.dtype)                   # (:328-329)
        if grid_xyz.shape[0] > 0:
            object_mask[grid_xyz[:, 0], grid_xyz[:, 1], grid_xyz[:, 2]] = 1.0
        background_mask = (max_iou < self.iou_thresh).to(self.dtype) * (1 - object_mask)   # (:331-332)
        conf = predict[..., 4]
        noobj = -torch.log(1 - conf)                                                 # (:335)
        if self.is_focal_loss:
            noobj = noobj * torch.pow(conf, self.focal_gamma)                        # (:337)
        noobj = self.w_noobj[hi] * torch.sum(noobj * background_mask)                # (:338)
        rt = target[max_pos]                                                         # (:341)
        rp = predict[grid_xyz[:, 0], grid_xyz[:, 1], grid_xyz[:, 2]]                 # (:342)
        obj = -torch.log(rp[:, 4])                                                   # (:344)
        if self.is_focal_loss:
            obj = obj * (torch.pow(1 - rp[:, 4], self.focal_gamma) * self.focal_alpha)   # (:346)
        obj = self.w_obj[hi] * torch.sum(obj)                                        # (:347)
        scale = (2 - rt[:, 2] * rt[:, 3] / (H * W)).unsqueeze(-1)                    # (:350)
        cint = torch.floor(rt[:, 0:2])                                               # (:352)
        txy = rt[:, 0:2] - cint                                                      # (:353)
        pxy = rp[:, 0:2] - cint                                                      # (:354)
        lxy = -(txy * torch.log(pxy) + (1 - txy) * torch.log(1 - pxy))               # (:355)
        lxy = self.w_xy[hi] * torch.sum(scale * lxy)                                 # (:356)
        lwh = torch.square(torch.log(rt[:, 2:4]) - torch.log(rp[:, 2:4]))            # (:358)
        lwh = self.w_wh[hi] * torch.sum(scale * lwh)                                 # (:359)
        if self.class_num >= 1:
            onehot = torch.nn.functional.one_hot(rt[:, 4].long(), self.class_num).to(self.dtype)   # (:362)
            lcls = self.w_cls[hi] * torch.sum(-onehot * torch.log(rp[:, 5:]))        # (:363-364)
        else:
            lcls = torch.zeros((), dtype=self.dtype)                                 # (:366)
        return torch.stack([lxy, lwh, noobj, obj, lcls]), grid_xyz                   # (:368)

    # ------------------------------------------------------------------ (:166-222)
    def _single_image_loss(self, dec, tgt):
        valid = tgt[0][0][:, 0] >= 0                                                 # (:239)
        targets = [(t[valid], b[valid]) for (t, b) in tgt]                           # (:241-247)
        ious = [self._calc_iou(targets[h][0], targets[h][1], dec[h][1], dec[h][2], self.grid[h]) for h in range(3)]
        r8, r16, r32 = ious[0][1], ious[1][1], ious[2][1]
        pos = [(r8 >= r16) & (r8 >= r32), (r16 >= r8) & (r16 >= r32), (r32 >= r8) & (r32 >= r16)]   # (:203-208)
        losses, assigns = [], []
        for h in range(3):
            l, a = self._single_head_loss(h, dec[h][1], targets[h][0], ious[h][0], ious[h][1], ious[h][2], pos[h],
                                          self.grid[h], self.box_num[h])
            losses.append(l)
            assigns.append(a)
        return torch.stack(losses, dim=-1), assigns                                  # (:221) (5, 3)

    # ------------------------------------------------------------------ (:140-164)
    def _rectified(self, raws):
        n = raws[0].shape[0]
        self.current_num += n                                                        # (:152)
        rows = [self.rectified_weight[h] * torch.sum(torch.square(raws[h]), dim=[1, 2, 3, 4]).mean()
                for h in range(3)]                                                   # (:153-162)
        return torch.stack(rows).reshape(1, 3)

    # ------------------------------------------------------------------ (:81-138)
    def loss_heads(self, targets, raw_heads):
        """targets (N, T*5) padded with -1; raw_heads = 3 x (N,H,W,B,L) logits.  Returns the scalar total loss."""
        dec = self.decoder.decode_heads(raw_heads)                                   # (:99-101)
        tgt = self.label_decoder.decode(targets.to(self.dtype))                      # (:103-105)
        n = targets.shape[0]
        per_image, self.last_assign = [], []
        for i in range(n):                                                           # (:111) map_fn over images
            d = [(dec[h][0][i], dec[h][1][i], dec[h][2][i]) for h in range(3)]
            t = [(tgt[h][0][i], tgt[h][1][i]) for h in range(3)]
            l, a = self._single_image_loss(d, t)
            per_image.append(l)
            self.last_assign.append(a)
        self.per_image = torch.stack(per_image)                                      # (N, 5, 3)
        yl = self.per_image.mean(dim=0)                                              # (:112)
        names = ['coord_loss_xy', 'coord_loss_wh', 'noobj_iou_loss', 'obj_iou_loss', 'class_loss']
        for k, nm in enumerate(names):                                               # (:115-121)
            self.detail[nm] = yl[k].detach().clone()
        if self.current_num <= self.rectified_coord_num:                             # (:125-130)
            rect = self._rectified([d[0] for d in dec])
            total = torch.cat([yl, rect], dim=0)
            self.detail['rectified_coord_loss'] = rect[0].detach().clone()           # (:131-132)
        else:
            total = yl
            self.detail['rectified_coord_loss'] = torch.zeros(3, dtype=self.dtype)   # (:133-134)
        self.terms = total.detach().clone()
        return total.sum()                                                           # (:137)

    def loss(self, targets, predicts):
        """Reference signature: predicts is the merged (N, H/32, W/32, C) tensor."""
        return self.loss_heads(targets, self.decoder.unpack(predicts.to(self.dtype)))
