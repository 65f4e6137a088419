"""YOLOv3PostProcessor with the reference's static methods (yolov3/yolov3_post_process.py:10-205): score filter, cross-head per-class NMS,
box rescale, drawing.  run.py uses the GPU path (filter_boxes_device / apply_nms_device); the host methods are an independent vectorised
NumPy implementation of the same contract (array gathers, a stable arg-sort and alive-flag suppression instead of the reference's Python
list surgery), checked bit for bit against golden vectors produced by the reference's own module (tests/golden/postprocess_*.npz).
``apply_nms`` reproduces the reference's never-advanced ``start_index`` (:81-89) by default; ``fixed_indices=True`` gives globally unique ids."""
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
        """score filter of the three decoded heads (semantics of reference :20-43) -> [rows (k, 8)] x 3"""
        heads = ((head_8_prediction, head_8_boxes), (head_16_prediction, head_16_boxes), (head_32_prediction, head_32_boxes))
        return [YOLOv3PostProcessor._filter_single_head_boxes(p, b, score_thresh) for p, b in heads]

    @staticmethod
    def _scores(prediction):
        """per prediction (flattened row-major over (row, col, anchor)): score = conf * best class prob, best prob, best class (reference :49-59).
        Without class channels the probability is 1 and the class 0, in the prediction's dtype."""
        flat = np.asarray(prediction).reshape(-1, prediction.shape[-1])
        conf = flat[:, 4]
        if flat.shape[1] == 5:
            return conf, np.ones_like(conf), np.zeros_like(conf)
        classes = flat[:, 5:]
        best = classes.argmax(axis=1)
        prob = classes[np.arange(flat.shape[0]), best]
        return prob * conf, prob, best

    @staticmethod
    def filter_indices(prediction, score_thresh):
        """the 'decoded box indices' of the parity criterion: flat ((row*W)+col)*B+anchor of every score > thresh (reference :57-62)"""
        return np.flatnonzero(YOLOv3PostProcessor._scores(prediction)[0] > score_thresh)

    @staticmethod
    def _filter_single_head_boxes(prediction, predict_boxes, score_thresh):
        """-> (k, 8) rows [x0/W, y0/H, x1/W, y1/H, conf, class prob, class index, score] of the predictions whose score exceeds the
        threshold, in flat (row, col, anchor) order.  One fancy-index gather per table instead of the reference's eight np.take columns
        (reference :45-77); dtype as there: float64 once an integer class index is among the columns, else the prediction's own."""
        height, width = prediction.shape[0], prediction.shape[1]
        score, prob, best = YOLOv3PostProcessor._scores(prediction)
        hits = np.flatnonzero(score > score_thresh)
        if hits.size == 0:
            return np.empty((0, 8), dtype=np.float64)
        corners = np.asarray(predict_boxes).reshape(-1, 4)[hits]
        out = np.empty((hits.size, 8), dtype=np.result_type(corners.dtype, score.dtype, best.dtype))
        out[:, 0:4] = corners / np.array([width, height, width, height], dtype=corners.dtype)
        out[:, 4] = np.asarray(prediction).reshape(-1, prediction.shape[-1])[hits, 4]
        out[:, 5], out[:, 6], out[:, 7] = prob[hits], best[hits], score[hits]
        return out

    @staticmethod
    def apply_nms(boxes, nms_thresh, fixed_indices=False):
        """class-wise greedy NMS over the boxes of all three heads together (semantics of reference :79-131) -> per head the surviving rows
        with a 9th column, the box id.  The reference numbers the boxes of EVERY head from 0 (its running start index is never advanced,
        :81-89) and keeps a row when its id is among the survivors' ids -- of any head; that is reproduced unless ``fixed_indices``."""
        heads = [np.asarray(b, dtype=np.float64).reshape(-1, 8) for b in boxes]
        sizes = [len(b) for b in heads]
        first = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if fixed_indices else np.zeros(len(heads), dtype=np.int64)
        ids = [np.arange(f, f + k, dtype=np.float64) for f, k in zip(first, sizes)]
        if sum(sizes) == 0:
            return [np.empty((0, 9), dtype=np.float64) for _ in heads]
        rows = np.concatenate(heads, axis=0)
        survivors = YOLOv3PostProcessor._greedy_nms(rows, nms_thresh)
        kept_ids = np.unique(np.concatenate(ids)[survivors])
        out = []
        for b, i in zip(heads, ids):
            if len(b) == 0:
                out.append(np.empty((0, 9), dtype=np.float64))
                continue
            tagged = np.concatenate([b, i[:, None]], axis=1)
            out.append(list(tagged[np.isin(i, kept_ids)]))
        return out

    @staticmethod
    def _greedy_nms(rows, nms_thresh):
        """indices (into ``rows``) of the boxes that survive: visit in descending score (ties in input order, i.e. a stable sort), each
        surviving box suppresses the later boxes of its own class whose IoU with it exceeds the threshold.  float64 IoU, evaluated
        in the reference's operation order (:134-162) so that the comparison with the threshold is bit-identical."""
        order = np.argsort(-rows[:, 7], kind='stable')
        r = rows[order]
        x0, y0, x1, y1, cls = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 6]
        area = (x1 - x0) * (y1 - y0)
        alive = np.ones(len(r), dtype=bool)
        for i in range(len(r) - 1):
            if not alive[i]:
                continue
            cand = np.flatnonzero(alive[i + 1:] & (cls[i + 1:] == cls[i])) + i + 1
            if cand.size == 0:
                continue
            w = np.minimum(x1[i], x1[cand]) - np.maximum(x0[i], x0[cand])
            h = np.minimum(y1[i], y1[cand]) - np.maximum(y0[i], y0[cand])
            inter = w * h
            with np.errstate(divide='ignore', invalid='ignore'):
                iou = np.where((w <= 0) | (h <= 0), 0.0, inter / (area[i] + area[cand] - inter))
            alive[cand[iou > nms_thresh]] = False
        return order[alive]

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
        """corner coordinates (normalised) -> pixels of ``target_size`` = [W, H, W, H]; the other columns pass through (semantics of
        reference :164-176).  Heads keep their container type: an empty head stays as it is, the others become lists of rows"""
        scale = np.asarray(target_size)
        out = []
        for head_boxes in boxes:
            if len(head_boxes) == 0:
                out.append(head_boxes)
                continue
            rows = np.asarray(head_boxes, dtype=np.float64).reshape(len(head_boxes), -1).copy()
            rows[:, :4] = rows[:, :4] * scale
            out.append(list(rows))
        return out

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
