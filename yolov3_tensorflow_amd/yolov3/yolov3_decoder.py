"""YOLOv3Decoder with the reference's constructor / ``decode`` signature (yolov3/yolov3_decoder.py:12-87).  The reference builds TF ops;
here ``decode`` runs the HIP decode kernel (yolo_decode_head) and returns NumPy arrays, which is what run.py's test/predict modes
consume (run.py:60-68).  During training the same arithmetic runs fused inside the loss kernel."""
import numpy as np
import torch
from yolov3_tensorflow_amd import ops, backend


class YOLOv3Decoder(object):
    def __init__(self, head_grid_sizes, class_num, anchor_boxes):
        self.head_grid_sizes = [(int(h), int(w)) for (h, w) in head_grid_sizes]
        (self.head_8_height, self.head_8_width), (self.head_16_height, self.head_16_width), (self.head_32_height, self.head_32_width) = \
            self.head_grid_sizes
        self.box_num = [len(a) for a in anchor_boxes]
        self.head_8_box_num, self.head_16_box_num, self.head_32_box_num = self.box_num
        # anchors scaled to grid units, float32 products (reference :35-40)
        self.anchors = [np.asarray(a, dtype=np.float32) * np.asarray([w, h], dtype=np.float32)
                        for a, (h, w) in zip(anchor_boxes, self.head_grid_sizes)]
        self.coord_num, self.conf_num, self.class_num = 4, 1, int(class_num)
        self.box_len = self.coord_num + self.conf_num + self.class_num

    def _unpack(self, predicts):
        """reference :89-117 -- (N, H/32, W/32, C) merged -> 3 x (N, H, W, B, L)"""
        out, start = [], 0
        for (h, w), b, m in zip(self.head_grid_sizes, self.box_num, (16, 4, 1)):
            end = start + b * self.box_len * m
            out.append(np.reshape(predicts[..., start:end], [-1, h, w, b, self.box_len]))
            start = end
        return out

    def decode_device(self, head_logits, ldc=None):
        """the same decode without leaving the GPU: ``head_logits`` = the detector's three float32 device tensors (N, H, W, ldc) (row
        stride ldc >= B*L) -> ([decoded (N,H,W,B,L)] x 3, [boxes (N,H,W,B,4)] x 3) device tensors for
        YOLOv3PostProcessor.filter_boxes_device"""
        decoded, boxes = [], []
        for lg, (h, w), b, anc in zip(head_logits, self.head_grid_sizes, self.box_num, self.anchors):
            N, L = lg.shape[0], self.box_len
            row = int(ldc) if ldc is not None else lg.shape[-1]
            assert lg.dtype == torch.float32 and lg.is_contiguous() and lg.shape[1:3] == (h, w) and row >= b * L
            dec = torch.empty(N, h, w, b, L, device=lg.device)
            box = torch.empty(N, h, w, b, 4, device=lg.device)
            ops.decode_head(lg, N, h, w, b, L, row, torch.as_tensor(anc).to(lg.device), backend.epsilon(), decoded=dec, boxes=box)
            decoded.append(dec), boxes.append(box)
        return decoded, boxes

    def decode(self, predicts, with_scores=False):
        """reference :62-87 -> [(raw t_xywh (N,H,W,B,4), decoded (N,H,W,B,L), boxes (N,H,W,B,4))] x 3, order /8, /16, /32.
        with_scores=True appends (score (N,H,W,B), class index (N,H,W,B)) computed on the GPU."""
        predicts = np.asarray(predicts, dtype=np.float32)
        dev = torch.device('cuda:%d' % torch.cuda.current_device())
        res = []
        for raw, (h, w), b, anc in zip(self._unpack(predicts), self.head_grid_sizes, self.box_num, self.anchors):
            N, L = raw.shape[0], self.box_len
            lg = torch.as_tensor(np.ascontiguousarray(raw.reshape(N, h, w, b * L))).to(dev)
            dec = torch.empty(N, h, w, b, L, device=dev)
            box = torch.empty(N, h, w, b, 4, device=dev)
            sc = torch.empty(N, h, w, b, device=dev)
            ci = torch.empty(N, h, w, b, dtype=torch.int32, device=dev)
            ops.decode_head(lg, N, h, w, b, L, b * L, torch.as_tensor(anc).to(dev), backend.epsilon(), decoded=dec, boxes=box, score=sc,
                            cls_idx=ci)
            item = (raw[..., 0:4], dec.cpu().numpy(), box.cpu().numpy())
            if with_scores:
                item = item + (sc.cpu().numpy(), ci.cpu().numpy())
            res.append(item)
        return res
