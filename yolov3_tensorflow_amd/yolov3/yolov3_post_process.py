"""YOLOv3PostProcessor with the reference's static methods (yolov3/yolov3_post_process.py:10-205): score filter, cross-head per-class NMS,
box rescale, drawing.  Host-side NumPy like the reference (it runs after the network on a handful of boxes); the per-prediction score /
class arg-max can be taken from the GPU decode (YOLOv3Decoder.decode(with_scores=True)).  ``apply_nms`` reproduces the reference's
never-advanced ``start_index`` (:81-89) by default; ``fixed_indices=True`` gives globally unique ids."""
import numpy as np


class DeviceBoxes(object):
    """candidate rows of one batch on the GPU: per head rows (N, cap, 8) float32, counts (N,) int32, flat prediction index (N, cap) int32,
    keep flags (N, cap) uint8 (after apply_nms_device)"""

    def __init__(self, rows, counts, index, cap):
        self.rows, self.counts, self.index, self.cap, self.keep, self.fixed_indices = rows, counts, index, cap, None, False


class YOLOv3PostProcessor(object):
    HEAD_BOX_COLOR = [[255, 0, 0], [0, 255, 0], [0, 0, 255]]  # blue, green, red (BGR) for head /8, /16, /32

    @staticmethod
    def filter_boxes(head_8_prediction, head_8_boxes, head_16_prediction, head_16_boxes, head_32_prediction, head_32_boxes, score_thresh):
        """reference :20-43"""
        f = YOLOv3PostProcessor._filter_single_head_boxes
        return [f(head_8_prediction, head_8_boxes, score_thresh), f(head_16_prediction, head_16_boxes, score_thresh),
                f(head_32_prediction, head_32_boxes, score_thresh)]

    @staticmethod
    def filter_indices(prediction, score_thresh):
        """the 'decoded box indices' of the parity criterion: flat ((row*W)+col)*B+anchor of every score > thresh (reference :57-62)"""
        score = prediction[..., 4]
        if prediction.shape[-1] > 5:
            score = np.max(prediction[..., 5:], axis=-1) * score
        return np.where(np.reshape(score > score_thresh, [-1]))[0]

    @staticmethod
    def _filter_single_head_boxes(prediction, predict_boxes, score_thresh):
        """reference :45-77 -> (k, 8) [x0, y0, x1, y1, conf, class prob, class index, score] in normalised units"""
        height, width, box_num, box_len = prediction.shape
        all_score = prediction[:, :, :, 4]
        all_class_prob = np.ones_like(all_score)
        all_class_indices = np.zeros_like(all_score)
        if box_len > 5:
            all_class_prob = np.max(prediction[:, :, :, 5:], axis=-1)
            all_class_indices = np.argmax(prediction[:, :, :, 5:], axis=-1)
            all_score = all_class_prob * all_score
        pos = np.where(np.reshape(all_score > score_thresh, [-1]))
        if len(pos[0]) == 0:
            return np.empty(shape=(0, 8), dtype=np.float64)
        cols = [np.take(predict_boxes[:, :, :, 0], pos) / width, np.take(predict_boxes[:, :, :, 1], pos) / height,
                np.take(predict_boxes[:, :, :, 2], pos) / width, np.take(predict_boxes[:, :, :, 3], pos) / height,
                np.take(prediction[:, :, :, 4], pos), np.take(all_class_prob, pos), np.take(all_class_indices, pos), np.take(all_score, pos)]
        return np.transpose(np.concatenate(cols, axis=0))

    @staticmethod
    def apply_nms(boxes, nms_thresh, fixed_indices=False):
        """reference :79-106"""
        boxes = list(boxes)
        start_index = 0
        for i, head_boxes in enumerate(boxes):
            end_index = len(head_boxes)
            if end_index == 0:
                boxes[i] = np.reshape(head_boxes, (0, 9))
                continue
            indices = np.expand_dims(np.arange(start_index, start_index + end_index, dtype=np.float64), axis=-1)
            boxes[i] = np.concatenate([head_boxes, indices], axis=-1)
            if fixed_indices:
                start_index += end_index
        sorted_boxes = YOLOv3PostProcessor._apply_nms(np.concatenate(boxes, axis=0), nms_thresh)
        keep = set(box[-1] for box in sorted_boxes)
        for i, head_boxes in enumerate(boxes):
            if len(head_boxes) == 0:
                continue
            boxes[i] = [box for box in head_boxes if box[-1] in keep]
        return boxes

    @staticmethod
    def _apply_nms(boxes, nms_thresh):
        """reference :109-131 -- greedy NMS among boxes of the same class, by descending score"""
        sorted_boxes = sorted(boxes, key=lambda d: d[7], reverse=True)
        index, box_num = 0, len(sorted_boxes) - 1
        while index < box_num:
            same = [(index + 1 + i, box) for (i, box) in enumerate(sorted_boxes[(index + 1):]) if box[6] == sorted_boxes[index][6]]
            ious = [(i, YOLOv3PostProcessor._cal_iou(sorted_boxes[index], box)) for (i, box) in same]
            removed = 0
            for i, iou in ious:
                if iou > nms_thresh:
                    del sorted_boxes[i - removed]
                    removed += 1
                    box_num -= 1
            index += 1
        return sorted_boxes

    @staticmethod
    def _cal_iou(box, truth):
        """reference :134-147"""
        w = YOLOv3PostProcessor._overlap(box[0], box[2], truth[0], truth[2])
        h = YOLOv3PostProcessor._overlap(box[1], box[3], truth[1], truth[3])
        if w <= 0 or h <= 0:
            return 0
        inter_area = w * h
        union_area = (box[2] - box[0]) * (box[3] - box[1]) + (truth[2] - truth[0]) * (truth[3] - truth[1]) - inter_area
        return inter_area / union_area

    @staticmethod
    def _overlap(x1, x2, x3, x4):
        """reference :150-162"""
        return min(x2, x4) - max(x1, x3)

    # ------------------------------------------------------------------------------------------------ GPU path (whole batch)
    @staticmethod
    def filter_boxes_device(predictions, boxes, score_thresh, cap=1024):
        """batched _filter_single_head_boxes on the GPU (yolo_filter_boxes): ``predictions`` / ``boxes`` are the three decoded device
        tensors (N, H, W, B, L) / (N, H, W, B, 4) of YOLOv3Decoder.decode_device.  Raises if an image has more than ``cap`` hits."""
        import torch
        from yolov3_tensorflow_amd import ops
        rows, counts, index = [], [], []
        for pred, box in zip(predictions, boxes):
            N, H, W, B, L = pred.shape
            r = torch.empty(N, cap, 8, device=pred.device)
            c = torch.empty(N, dtype=torch.int32, device=pred.device)
            ix = torch.empty(N, cap, dtype=torch.int32, device=pred.device)
            ops.filter_boxes(pred, box, N, H, W, B, L, np.float32(score_thresh), cap, c, r, ix)
            rows.append(r), counts.append(c), index.append(ix)
        return DeviceBoxes(rows, counts, index, cap)

    @staticmethod
    def apply_nms_device(dev_boxes, nms_thresh, fixed_indices=False):
        """apply_nms for every image of the batch in one launch (yolo_nms_heads)"""
        import torch
        from yolov3_tensorflow_amd import ops
        N = dev_boxes.counts[0].shape[0]
        dev = dev_boxes.counts[0].device
        keep = [torch.zeros(N, dev_boxes.cap, dtype=torch.uint8, device=dev) for _ in range(3)]
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.nms_heads(dev_boxes.rows, dev_boxes.counts, N, dev_boxes.cap, nms_thresh, fixed_indices, keep, status)
        bad = int(status.item())
        if bad:
            raise RuntimeError('box selection overflow: an image has %d candidate boxes (cap %d per head, %d per image); raise the score '
                               'threshold or the cap' % (bad, dev_boxes.cap, ops.nms_max_candidates()))
        dev_boxes.keep, dev_boxes.fixed_indices = keep, bool(fixed_indices)
        return dev_boxes

    @staticmethod
    def boxes_to_host(dev_boxes, after_nms=True):
        """-> per image [head /8, /16, /32] float64 arrays in the reference's row format: (k, 8) filtered rows, or (k, 9) rows with the
        box id appended for the NMS survivors (what apply_nms returns)"""
        counts = [c.cpu().numpy() for c in dev_boxes.counts]
        rows = [r.cpu().numpy() for r in dev_boxes.rows]
        keep = [k.cpu().numpy().astype(bool) for k in dev_boxes.keep] if after_nms else None
        out = []
        for n in range(counts[0].shape[0]):
            per_head, start = [], 0
            for h in range(3):
                k = int(counts[h][n])
                if k > dev_boxes.cap:
                    raise RuntimeError('box selection overflow: %d hits in one head (cap %d)' % (k, dev_boxes.cap))
                r = rows[h][n, :k].astype(np.float64)
                if after_nms:
                    r = np.concatenate([r, np.arange(start, start + k, dtype=np.float64)[:, None]], axis=-1)[keep[h][n, :k]]
                    start += k if dev_boxes.fixed_indices else 0
                per_head.append(r)
            out.append(per_head)
        return out

    @staticmethod
    def resize_boxes(boxes, target_size):
        """reference :164-176"""
        return [head_boxes if len(head_boxes) == 0 else
                [np.concatenate([box[:4] * target_size, box[4:]], axis=-1) for box in head_boxes] for head_boxes in boxes]

    @staticmethod
    def visualize(image, boxes, src_box_size, image_path):
        """reference :178-205 -- draws with PIL (OpenCV is not a dependency here); boxes coloured per head"""
        from PIL import Image, ImageDraw
        img = (np.clip(np.asarray(image, dtype=np.float32), 0, 1) * 255).astype(np.uint8)[..., ::-1]      # BGR float -> RGB uint8
        pil = Image.fromarray(np.ascontiguousarray(img))
        draw = ImageDraw.Draw(pil)
        for head_boxes, color in zip(boxes, YOLOv3PostProcessor.HEAD_BOX_COLOR):
            for box in head_boxes:
                x0, y0, x1, y1 = [float(v) for v in box[:4]]
                draw.rectangle([x0, y0, x1, y1], outline=tuple(color[::-1]), width=2)
                draw.text((x0 + 2, y0 + 2), '%d:%.2f' % (int(box[6]), float(box[7])), fill=tuple(color[::-1]))
        pil.save(image_path)
