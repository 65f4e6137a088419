"""YOLOv3Loss with the reference's constructor and ``loss(targets, predicts)`` signature (yolov3/yolov3_loss.py:13-138).
The arithmetic (decode, IoU assignment, masks, xy/wh/conf/class/rectified terms AND the gradient w.r.t. the head logits)
runs in the fused HIP kernels behind ``yolo_loss_fwd_bwd``."""
import numpy as np
import torch
from yolov3_tensorflow_amd import ops, backend


class YOLOv3Loss(object):
    def __init__(self, head_grid_sizes, class_num, anchor_boxes, iou_thresh, loss_weights,
                 rectified_coord_num=0, rectified_loss_weight=None,
                 is_focal_loss=False, focal_alpha=0.25, focal_gamma=2.0, is_tiou_recall=False):
        if rectified_loss_weight is None:
            rectified_loss_weight = [0.01, 0.01, 0.01]                                 # reference :63-64
        elif len(rectified_loss_weight) != 3:
            raise ValueError('rectified_loss_weight must have length 3: weights of head 8, 16, 32')   # reference :65-66
        self.head_grid_sizes = [(int(h), int(w)) for (h, w) in head_grid_sizes]
        self.class_num = int(class_num)
        self.box_len = 4 + 1 + self.class_num
        self.anchor_boxes = [list(a) for a in anchor_boxes]
        self.box_num = [len(a) for a in self.anchor_boxes]
        self.iou_thresh = float(iou_thresh)
        self.loss_weights = [tuple(w) for w in loss_weights]
        self.rectified_coord_num = int(rectified_coord_num)
        self.rectified_loss_weight = [float(w) for w in rectified_loss_weight]
        self.is_focal_loss, self.focal_alpha, self.focal_gamma = bool(is_focal_loss), float(focal_alpha), float(focal_gamma)
        self.is_tiou_recall = bool(is_tiou_recall)
        self.model = None
        self.dev = None
        self.T = 0
        self._terms_host = np.zeros((6, 3), dtype=np.float32)

    # ------------------------------------------------------------------ device state
    def _config(self, ldc, T):
        return ops.make_loss_config(self.head_grid_sizes, self.class_num, self.anchor_boxes, self.iou_thresh, self.loss_weights,
                                    ldc, T, rectified_coord_num=self.rectified_coord_num,
                                    rectified_loss_weight=self.rectified_loss_weight, is_focal_loss=self.is_focal_loss,
                                    focal_alpha=self.focal_alpha, focal_gamma=self.focal_gamma,
                                    is_tiou_recall=self.is_tiou_recall, eps=backend.epsilon(), grad_scale16=backend.loss_scale())

    def _alloc(self, dev, N, ldc, T):
        self.dev, self.N, self.ldc, self.T = dev, N, list(ldc), T
        self.cfg = self._config(ldc, T)
        self.labels = -torch.ones(N, T, 5, device=dev)
        self.ws = torch.zeros(ops.loss_workspace_bytes(self.cfg, N), dtype=torch.uint8, device=dev)
        if not hasattr(self, 'current_num') or self.current_num.device != dev:
            self.current_num = torch.zeros(1, dtype=torch.int32, device=dev)            # reference :69
        self.terms = torch.zeros(6, 3, device=dev)
        self.total = torch.zeros(1, device=dev)
        self.assign = torch.zeros(N, T, 3, dtype=torch.int32, device=dev)

    def bind(self, model, T=8):
        grids = [(h.shape[1], h.shape[2]) for h in model.heads]
        if grids != self.head_grid_sizes:
            raise ValueError('head_grid_sizes %s do not match the model %s' % (self.head_grid_sizes, grids))
        for c, b in zip(model.head_channel_nums, self.box_num):
            if c != b * self.box_len:
                raise ValueError('head channels %d != box_num %d * box_len %d' % (c, b, self.box_len))
        self.model = model
        with torch.cuda.device(model.device):
            self._alloc(model.device, model.batch_size, model.ldc, max(T, self.T))

    def stage_labels(self, lab):
        """lab: (N, T*5) float32 padded with -1 (reference dataset/file_util.py:97)"""
        lab = lab.reshape(lab.shape[0], -1, 5)
        if lab.shape[0] != self.N:
            raise ValueError('labels batch %d != %d' % (lab.shape[0], self.N))
        if lab.shape[1] > self.T:          # more objects than slots: grow and force a re-capture
            with torch.cuda.device(self.dev):
                self._alloc(self.dev, self.N, self.ldc, int(lab.shape[1]))
            if self.model is not None:
                self.model._graphs = None
        if torch.is_tensor(lab) and lab.is_cuda:       # already on the device: a device-side copy
            self.labels.fill_(-1.0)
            self.labels[:, :lab.shape[1]].copy_(lab.to(torch.float32), non_blocking=True)
            return
        # through a ring of page-locked buffers: a copy from pageable host memory is synchronous on this runtime -- it waits for everything
        # queued on the stream, i.e. for the previous training step -- and made every file-fed step pay a device synchronisation
        ring = getattr(self, '_lab_ring', None)
        if ring is None or ring[0][0].shape != self.labels.shape:
            ring = self._lab_ring = [(torch.empty(self.labels.shape, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self._lab_next = 0
        pin, ev = ring[self._lab_next]
        self._lab_next = (self._lab_next + 1) % len(ring)
        ev.synchronize()                               # (the upload that last used this buffer, four batches ago)
        pin.fill_(-1.0)
        pin[:, :lab.shape[1]].copy_(torch.as_tensor(lab, dtype=torch.float32))
        self.labels.copy_(pin, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))

    def launch(self, model):
        """enqueue loss forward+backward for the model's current head logits; d(logits) lands in the heads' dy buffers"""
        scale = backend.loss_scale()
        if self.cfg.grad_scale16 != scale:           # backend.set_loss_scale() after compile: the optimizer divides by the CURRENT scale
            self.cfg.grad_scale16 = scale            # (radam.launch reads it per step), so the 16-bit d(logits) must carry the same one
            model._graphs = None                     # a captured hipGraph holds the old kernel argument
        ops.loss_fwd_bwd(self.cfg, self.N, self.N * model.world_size, [h.buf for h in model.heads], self.labels, self.current_num,
                         self.terms, self.total, self.ws, dlogits_bf16=[h.dy for h in model.heads], assign_out=self.assign)

    # ------------------------------------------------------------------ logging vectors (reference :72-79)
    def _term(self, k):
        return self.terms[k].detach().cpu().numpy() if self.dev is not None else self._terms_host[k]

    coord_loss_xy = property(lambda self: self._term(0))
    coord_loss_wh = property(lambda self: self._term(1))
    noobj_iou_loss = property(lambda self: self._term(2))
    obj_iou_loss = property(lambda self: self._term(3))
    class_loss = property(lambda self: self._term(4))
    rectified_coord_loss = property(lambda self: self._term(5))

    # ------------------------------------------------------------------ reference signature
    def loss(self, targets, predicts, return_grads=False):
        """targets (N, obj_num*5) padded with -1; predicts (N, H/32, W/32, C) merged heads (reference :81-138).
        Returns the scalar total loss (and optionally d loss / d predicts in the same merged layout)."""
        targets = torch.as_tensor(np.asarray(targets, dtype=np.float32))
        predicts = torch.as_tensor(np.asarray(predicts, dtype=np.float32))
        N = predicts.shape[0]
        dev = self.dev if self.dev is not None else torch.device('cuda:%d' % torch.cuda.current_device())
        L = self.box_len
        ldc = [ops.pad_channels(b * L) for b in self.box_num]
        T = max(targets.reshape(N, -1, 5).shape[1], 1)
        with torch.cuda.device(dev):
            saved = (self.model,)
            if self.dev is None or self.N != N or self.ldc != ldc or self.T < T:
                self._alloc(dev, N, ldc, max(T, self.T))
            logits, dl, start = [], [], 0
            for (h, w), b, m, ld in zip(self.head_grid_sizes, self.box_num, (16, 4, 1), ldc):     # un-merge (yolov3_decoder.py:99-117)
                end = start + b * L * m
                t = torch.zeros(N, h, w, ld)
                t[..., :b * L] = predicts[..., start:end].reshape(N, h, w, b * L)
                logits.append(t.to(dev))
                dl.append(torch.zeros(N, h, w, ld, device=dev))
                start = end
            self.stage_labels(targets)
            ops.loss_fwd_bwd(self.cfg, N, N, logits, self.labels, self.current_num, self.terms, self.total, self.ws, dlogits=dl,
                             assign_out=self.assign)
            total = float(self.total.item())
            self.model = saved[0]
        if not return_grads:
            return total
        H32, W32 = self.head_grid_sizes[2]
        grads = [d[..., :b * L].reshape(N, H32, W32, -1).cpu() for d, b in zip(dl, self.box_num)]
        return total, torch.cat(grads, dim=-1).numpy()
