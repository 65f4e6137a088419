"""FileUtil.get_dataset with the reference's arguments (dataset/file_util.py:62-114): label file ``name cx cy w h cls ...`` ->
an infinite (train) or single-pass (test) iterator of full batches
    images float32 (N, H, W, 3) in [0, 1], BGR, letterboxed with nearest-neighbour resize  -- a DEVICE tensor,
    labels float32 (N, T*5) = (cx, cy, w, h, cls) normalised to the letterboxed image, padded with -1
(+ image paths when is_test), i.e. the tensors the reference's tf.data pipeline feeds to keras fit (file_util.py:54-59,95-97).
The host only decodes the JPEGs (PIL; TensorFlow's decoder is not available, decoded pixels can differ in the last bit) and transforms
the few label numbers; resize, normalisation, channel order and augmentation run in one GPU kernel over the whole batch
(DeviceImagePipeline -> yolo_letterbox_augment).  Image and label stay paired by construction (the reference relies on identical
shuffle seeds of two dataset branches, file_util.py:79-88)."""
import ctypes
import math
import os
import numpy as np
from yolov3_tensorflow_amd.dataset.dataset_util import DatasetUtil


class _Slot(object):
    """one in-flight batch: pinned staging bytes + descriptor block, their device twins, and the event after which they may be reused"""

    def __init__(self):
        self.shm = None                               # shared-memory segment behind ``stage`` (process-pool decode), else None
        self.stage = self.dev = self.desc_host = self.desc_dev = self.event = None
        self.descs = self.total = self.views = None
        self.uploaded = self.alloc_stream = None      # upload-finished event; the stream the device buffers were allocated on (until first use)


class DeviceImagePipeline(object):
    """one batch of decoded uint8 RGB images (any sizes) -> float32 (N, H, W, 3) BGR [0, 1] on the GPU (and, optionally, the packed
    bf16 conv input).  Batches travel through a ring of slots (pinned staging buffer + descriptor block + device twins): decode threads
    write their pixels straight into the pinned buffer of a slot while the GPU still works on the previous ones."""

    def __init__(self, batch_size, image_size, device=None, max_pixels_per_image=4096 * 4096, slots=4, shared=False):
        import torch
        from yolov3_tensorflow_amd import ops, _lib
        if not torch.cuda.is_available():
            raise RuntimeError('the input pipeline runs on the GPU (yolo_letterbox_augment): no GPU is visible and there is no CPU fallback')
        self.torch, self.ops, self.lib = torch, ops, _lib
        self.N, (self.H, self.W) = int(batch_size), (int(image_size[0]), int(image_size[1]))
        self.device = torch.device('cuda:%d' % torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_pixels = int(max_pixels_per_image)
        self.workspace = torch.empty(ops.letterbox_workspace_bytes(self.N), dtype=torch.uint8, device=self.device)
        self._slots = [_Slot() for _ in range(max(2, int(slots)))]
        self._copy_stream = None
        self._next = 0
        # shared: the staging buffers are POSIX shared-memory segments that decode PROCESSES attach to by name (decode_worker.py), page-locked
        # in this process with hipHostRegister so that the upload stays an asynchronous DMA
        self.shared = bool(shared)
        if self.shared:
            import weakref
            weakref.finalize(self, DeviceImagePipeline._release, self._slots, torch)

    @staticmethod
    def _release(slots, torch):
        for slot in slots:
            if slot.shm is not None:
                try:
                    torch.cuda.cudart().cudaHostUnregister(slot.stage.data_ptr())
                except Exception:
                    pass
                slot.stage = None
                try:
                    slot.shm.close()
                    slot.shm.unlink()
                except Exception:
                    pass
                slot.shm = None

    def close(self):
        """unmap and unlink the shared staging segments (they live in /dev/shm until then)"""
        if self.shared:
            self.torch.cuda.synchronize(self.device)
            DeviceImagePipeline._release(self._slots, self.torch)

    def _alloc_stage(self, slot, nbytes):
        torch = self.torch
        if not self.shared:
            slot.stage = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            return
        from multiprocessing import shared_memory
        if slot.shm is not None:
            DeviceImagePipeline._release([slot], torch)
        slot.shm = shared_memory.SharedMemory(create=True, size=nbytes)
        slot.stage = torch.frombuffer(slot.shm.buf, dtype=torch.uint8, count=nbytes)
        err = torch.cuda.cudart().cudaHostRegister(slot.stage.data_ptr(), nbytes, 0)
        if int(err) != 0:
            raise RuntimeError('hipHostRegister of the shared staging buffer failed (%s)' % err)

    def reserve(self, nbytes):
        """allocate every slot's staging / device buffers for ``nbytes`` now (raises where the host cannot provide the memory)"""
        torch = self.torch
        for slot in self._slots:
            if slot.stage is None or slot.stage.numel() < nbytes:
                with torch.cuda.device(self.device):
                    self._alloc_stage(slot, int(nbytes))
                    slot.dev = torch.empty(slot.stage.numel(), dtype=torch.uint8, device=self.device)
                    slot.alloc_stream = torch.cuda.current_stream(self.device)

    @staticmethod
    def geometry(h, w, H, W):
        """tf.image.resize_image_with_pad: ratio = max(w/W, h/H) in float64, resized = floor(dim / ratio), offset = floor((target - dim/ratio) / 2)"""
        ratio = max(float(w) / float(W), float(h) / float(H))
        rh, rw = float(h) / ratio, float(w) / ratio
        return int(math.floor(rh)), int(math.floor(rw)), max(0, int(math.floor((H - rh) / 2))), max(0, int(math.floor((W - rw) / 2)))

    def acquire(self, sizes_hw):
        """reserve the next slot for N images of the given decoded sizes (any thread): -> slot with ``views[n]`` = uint8 (h, w, 3) windows
        of its pinned buffer to decode / copy into.  Blocks only if the GPU has not finished with that slot's previous batch."""
        torch = self.torch
        if len(sizes_hw) != self.N:
            raise ValueError('expected %d images, got %d' % (self.N, len(sizes_hw)))
        slot = self._slots[self._next]
        self._next = (self._next + 1) % len(self._slots)
        if slot.event is not None:
            slot.event.synchronize()
        total = 0
        descs = (self.lib.ImageDesc * self.N)()
        for n, (h, w) in enumerate(sizes_hw):
            h, w = int(h), int(w)
            if h * w > self.max_pixels or h < 1 or w < 1:
                raise ValueError('image %d has unsupported size %dx%d' % (n, h, w))
            nh, nw, top, left = self.geometry(h, w, self.H, self.W)
            if nh < 1 or nw < 1:
                raise ValueError('image %d (%dx%d) collapses to nothing at %dx%d' % (n, h, w, self.H, self.W))
            d = descs[n]
            d.offset, d.h, d.w, d.nh, d.nw, d.top, d.left = total, h, w, nh, nw, top, left
            total += (h * w * 3 + 15) // 16 * 16
        if slot.stage is None or slot.stage.numel() < total:
            with torch.cuda.device(self.device):
                self._alloc_stage(slot, int(total * 1.25))
                slot.dev = torch.empty(slot.stage.numel(), dtype=torch.uint8, device=self.device)
                slot.alloc_stream = torch.cuda.current_stream(self.device)
        if slot.desc_host is None:                  # (not under the branch above: reserve() may have sized the staging buffers already)
            with torch.cuda.device(self.device):
                slot.desc_host = torch.empty(self.N * ctypes.sizeof(self.lib.ImageDesc), dtype=torch.uint8).pin_memory()
                slot.desc_dev = torch.empty_like(slot.desc_host, device=self.device)
                slot.alloc_stream = torch.cuda.current_stream(self.device)
        stage = slot.stage.numpy()
        slot.descs, slot.total = descs, total
        slot.offsets = [int(descs[n].offset) for n in range(self.N)]
        slot.views = [stage[descs[n].offset:descs[n].offset + descs[n].h * descs[n].w * 3].reshape(descs[n].h, descs[n].w, 3)
                      for n in range(self.N)]
        return slot

    def run(self, slot, draws=None, out_f32=True, out_bf16x8=None):
        """upload the filled slot and launch the kernel on the current stream (consumer thread)"""
        torch = self.torch
        for n in range(self.N):
            for k, v in (draws[n] if draws is not None else DatasetUtil.NO_AUGMENT).items():
                setattr(slot.descs[n], k, v)
        ctypes.memmove(slot.desc_host.data_ptr(), ctypes.addressof(slot.descs), ctypes.sizeof(slot.descs))
        # the upload runs on its own stream: called while the consumer's previous step is still executing, it crosses PCIe beside that step
        # instead of queueing behind it (7036 -> 7503 images/s on the headline step); the kernel below waits for it.
        # (acquire() has already waited for this slot's previous kernel, so nothing still reads slot.dev)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        if slot.uploaded is None:
            slot.uploaded = torch.cuda.Event()
        if slot.alloc_stream is not None:       # fresh buffers: the caching allocator orders their reuse on the allocating stream only
            self._copy_stream.wait_stream(slot.alloc_stream)
            slot.alloc_stream = None
        with torch.cuda.stream(self._copy_stream):
            slot.dev[:slot.total].copy_(slot.stage[:slot.total], non_blocking=True)
            slot.desc_dev.copy_(slot.desc_host, non_blocking=True)
            slot.uploaded.record(self._copy_stream)
        torch.cuda.current_stream(self.device).wait_event(slot.uploaded)
        out = torch.empty(self.N, self.H, self.W, 3, device=self.device) if out_f32 else None
        self.ops.letterbox_augment(slot.dev, slot.desc_dev, self.N, self.H, self.W, draws is not None, self.workspace, out_f32=out,
                                   out_bf16x8=out_bf16x8)
        if slot.event is None:
            slot.event = torch.cuda.Event()
        slot.event.record(torch.cuda.current_stream(self.device))
        return out

    def __call__(self, images_rgb_u8, draws=None, out_f32=True, out_bf16x8=None):
        """images_rgb_u8: list of N uint8 (h, w, 3) arrays; draws: list of DatasetUtil.draw() dicts (None = no augmentation)"""
        for n, im in enumerate(images_rgb_u8):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('image %d must be uint8 (h, w, 3), got %s %s' % (n, im.dtype, im.shape))
        slot = self.acquire([im.shape[:2] for im in images_rgb_u8])
        for view, im in zip(slot.views, images_rgb_u8):
            np.copyto(view, im)
        return self.run(slot, draws, out_f32=out_f32, out_bf16x8=out_bf16x8)


class FileUtil(object):

    @staticmethod
    def _parse_label_file(label_path):
        """reference :21-38"""
        names, labels = [], []
        with open(label_path) as f:
            for line in f:
                parts = line.split()
                if not parts:
                    continue
                names.append(parts[0])
                labels.append(np.asarray(parts[1:], dtype=np.float32).reshape(-1, 5))
        return names, labels

    @staticmethod
    def transform_label(label, src_hw, image_size):
        """reference :48-53: xy' = xy * r + (1 - r) / 2, wh' = wh * r with r = (src_hw / dst_hw)[::-1] / max(src_hw / dst_hw), float32"""
        lab = np.asarray(label, dtype=np.float32).reshape(-1, 5).copy()
        if len(lab):
            src = np.asarray(src_hw, dtype=np.float32) / np.asarray(image_size, dtype=np.float32)
            ratio = src[::-1] / np.max(src)                      # (rw, rh): the side that fills the canvas has ratio 1
            lab[:, 0:2] = lab[:, 0:2] * ratio + (np.float32(1) - ratio) / np.float32(2.0)
            lab[:, 2:4] = lab[:, 2:4] * ratio
        return lab

    @staticmethod
    def read_image(path):
        """tf.read_file + tf.image.decode_jpeg (reference :44-45) -> uint8 (h, w, 3) RGB"""
        from PIL import Image
        return np.asarray(Image.open(path).convert('RGB'))

    @staticmethod
    def _copy_into(args):
        np.copyto(args[0], args[1])          # releases the GIL for the whole image

    @staticmethod
    def host_batches(file_path, image_dir, image_size, batch_size, is_augment=True, is_test=False, seed=800, pool=None, pipe=None,
                     rank=0, world=1, sizes=None):
        """the host half of get_dataset: (decoded images, padded transformed labels (N, T*5), augmentation draws or None, paths).
        ``pool``: a concurrent.futures executor that decodes the images of a batch in parallel (PIL releases the GIL while decoding);
        order, pairing and the random draws do not depend on it.  With ``pipe`` (a DeviceImagePipeline) the first element is a filled slot
        of its staging ring instead of a list of arrays: the decode threads also copy the pixels into the pinned memory.

        Data parallel (``world`` > 1): ``batch_size`` is the PER-RANK batch.  Every rank walks the same shuffled order with the same seed in
        GLOBAL batches of ``batch_size * world`` images and takes the contiguous slice ``[rank * batch_size, (rank + 1) * batch_size)`` of
        each -- the slice keras multi_gpu_model hands tower ``rank`` (reference trainer.py:40-43 splits one batch on axis 0).  The
        augmentation draws of the whole global batch come from the one shared stream, so the concatenation of the ranks' batches is exactly
        the batch (images, labels and draws) a single process with batch ``batch_size * world`` would train on."""
        names, labels = FileUtil._parse_label_file(file_path)
        if not names:
            raise ValueError('empty label file ' + file_path)
        rank, world = int(rank), int(world)
        if not 0 <= rank < world:
            raise ValueError('rank %d outside world %d' % (rank, world))
        t_max = max(len(l) for l in labels)
        rng = np.random.RandomState(seed)
        order = np.arange(len(names))
        gbatch = batch_size * world
        lo = rank * batch_size
        while True:
            if not is_test:
                rng.shuffle(order)                                   # shuffle-and-repeat (reference :79)
            for i in range(0, len(order) - gbatch + 1 if not is_test else len(order), gbatch):
                gidx = order[i:i + gbatch]
                if len(gidx) < gbatch:                               # keras fit gets full batches; pad the last test batch by wrap-around
                    gidx = np.concatenate([gidx, np.resize(order, gbatch - len(gidx))])
                gdraws = [DatasetUtil.draw(rng) for _ in gidx] if (is_augment and not is_test) else None
                idx = gidx[lo:lo + batch_size]
                draws = gdraws[lo:lo + batch_size] if gdraws is not None else None
                labs = -np.ones((batch_size, t_max, 5), dtype=np.float32)
                paths = [os.path.join(image_dir, names[j]) for j in idx]
                mapper = pool.map if pool is not None else map
                if sizes is not None:                                # planning only (process-pool decode): {path: (h, w)} from the file headers
                    imgs, shapes = None, [sizes[p] for p in paths]
                else:
                    imgs = list(mapper(FileUtil.read_image, paths))
                    shapes = [im.shape[:2] for im in imgs]
                    if pipe is not None:                             # into the pinned staging memory, also on the decode threads
                        arrays, imgs = imgs, pipe.acquire(shapes)
                        list(mapper(FileUtil._copy_into, zip(imgs.views, arrays)))
                for k, j in enumerate(idx):
                    lb = FileUtil.transform_label(labels[j], shapes[k], image_size)
                    labs[k, :len(lb)] = lb
                yield imgs, labs.reshape(batch_size, t_max * 5), draws, paths
            if is_test:
                return

    @staticmethod
    def get_dataset(file_path, image_dir, image_size, batch_size, is_augment=True, is_test=False, seed=800, device=None, num_workers=None,
                    prefetch=3, rank=None, world=None, decode_procs=None):
        """reference :62-114 (its tf.data pipeline decodes with AUTOTUNE parallelism and prefetches, :85-114).  Here ``num_workers`` threads
        (default: the CPUs of this process, at most 8 -- Python threads stop scaling there: ~3000 images/s of 500 x 375 JPEGs on the GPU box's host,
        tools/input_pipeline_bench.py) decode the JPEGs of a batch and a producer thread keeps ``prefetch`` decoded
        batches ahead of the GPU; resize / normalise / augment then run in the GPU kernel on the consumer side.
        ``batch_size`` is the batch THIS process trains on.  ``rank`` / ``world`` (default: RANK / WORLD_SIZE of a torchrun launch; 0 / 1 for
        a test set, which every rank reads whole) select this rank's slice of each global batch, see host_batches."""
        import concurrent.futures
        import queue
        import threading
        if decode_procs is None:
            decode_procs = int(os.environ.get('YOLO_DECODE_PROCS', '0'))
        if decode_procs and decode_procs > 0:
            return FileUtil._get_dataset_procs(file_path, image_dir, image_size, batch_size, is_augment, is_test, seed, device, int(decode_procs),
                                               max(2, prefetch), rank, world)
        pipe = DeviceImagePipeline(batch_size, image_size, device=device, slots=max(1, prefetch) + 3)
        if world is None:
            world = 1 if is_test else int(os.environ.get('WORLD_SIZE', '1'))
        if rank is None:
            rank = 0 if is_test else int(os.environ.get('RANK', '0'))
        if num_workers is None:
            num_workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))

        def batches():
            pool = concurrent.futures.ThreadPoolExecutor(max_workers=num_workers) if num_workers > 1 else None
            q = queue.Queue(maxsize=max(1, prefetch))
            stop = threading.Event()
            done = object()

            def produce():
                try:
                    for item in FileUtil.host_batches(file_path, image_dir, image_size, batch_size, is_augment, is_test, seed, pool=pool,
                                                      pipe=pipe, rank=rank, world=world):
                        while not stop.is_set():
                            try:
                                q.put(item, timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                    q.put(done)
                except BaseException as e:            # surface decode errors on the consumer side
                    q.put(e)

            t = threading.Thread(target=produce, daemon=True)
            t.start()
            try:
                while True:
                    item = q.get()
                    if item is done:
                        return
                    if isinstance(item, BaseException):
                        raise item
                    slot, labs, draws, paths = item
                    x = pipe.run(slot, draws)
                    yield (x, labs, paths) if is_test else (x, labs)
            finally:
                stop.set()
                if pool is not None:
                    pool.shutdown(wait=False)
        return batches()

    @staticmethod
    def _shm_free_bytes():
        try:
            st = os.statvfs('/dev/shm')
            return int(st.f_bavail) * int(st.f_frsize)
        except (OSError, AttributeError):
            return None

    @staticmethod
    def _get_dataset_procs(file_path, image_dir, image_size, batch_size, is_augment, is_test, seed, device, procs, window, rank, world):
        """get_dataset with the JPEG decode on ``procs`` worker PROCESSES (decode_worker.DecodePool: separate interpreters that import PIL and
        NumPy only and never touch the GPU).  The parent plans every batch -- order, labels, augmentation draws, and from the files' header
        sizes the byte offset of every image in a staging slot -- and hands out (segment, offset, size, path) tasks; the workers write the
        decoded pixels directly into the slot's shared, page-locked memory.  ``window`` batches are in flight at the pool while the consumer
        uploads and launches the previous ones: same batches, same order, same draws as the thread path (host_batches plans both)."""
        from yolov3_tensorflow_amd.dataset import decode_worker
        if world is None:
            world = 1 if is_test else int(os.environ.get('WORLD_SIZE', '1'))
        if rank is None:
            rank = 0 if is_test else int(os.environ.get('RANK', '0'))
        pool = decode_worker.DecodePool(procs)
        pipe = None
        try:
            names, _ = FileUtil._parse_label_file(file_path)
            paths_all = sorted(set(os.path.join(image_dir, n) for n in names))
            sizes = dict(zip(paths_all, pool.probe_sizes(paths_all)))
            # the staging slots are POSIX shared-memory segments (/dev/shm) page-locked with hipHostRegister: reserve ALL of them for the
            # largest possible batch now, where a host that cannot give them (a 64 MB container /dev/shm: SharedMemory(create=True) succeeds
            # sparsely and the first write dies with SIGBUS) can still be answered with the thread path
            slots = window + 3
            worst = sorted((h * w * 3 + 15) // 16 * 16 for h, w in sizes.values())[-batch_size:]
            slot_bytes = int((sum(worst) + (batch_size - len(worst)) * (worst[-1] if worst else 0)) * 1.25)
            free = FileUtil._shm_free_bytes()
            if free is not None and free < slots * slot_bytes + (8 << 20):
                raise OSError('/dev/shm has %.0f MB free, the %d staging slots need %.0f MB' % (free / 2**20, slots, slots * slot_bytes / 2**20))
            pipe = DeviceImagePipeline(batch_size, image_size, device=device, slots=slots, shared=True)
            pipe.reserve(slot_bytes)
        except (OSError, RuntimeError, MemoryError) as e:
            import warnings
            warnings.warn('decode_procs=%d: shared page-locked staging memory is not available (%s); falling back to the decode THREADS '
                          '(decode_procs=0)' % (procs, e))
            pool.close()
            if pipe is not None:
                pipe.close()
            return FileUtil.get_dataset(file_path, image_dir, image_size, batch_size, is_augment=is_augment, is_test=is_test, seed=seed,
                                        device=device, prefetch=window, rank=rank, world=world, decode_procs=0)

        def batches():
            inflight = []
            planner = FileUtil.host_batches(file_path, image_dir, image_size, batch_size, is_augment, is_test, seed, pool=None, pipe=None,
                                            rank=rank, world=world, sizes=sizes)           # plans only: it needs the sizes, not the pixels
            exhausted = False
            try:
                while True:
                    while not exhausted and len(inflight) < window:
                        try:
                            _, labs, draws, paths = next(planner)
                        except StopIteration:
                            exhausted = True
                            break
                        slot = pipe.acquire([sizes[p] for p in paths])
                        pool.submit([(slot.shm.name, off, sizes[p][0], sizes[p][1], p) for off, p in zip(slot.offsets, paths)])
                        inflight.append((slot, labs, draws, paths))
                    if not inflight:
                        return
                    slot, labs, draws, paths = inflight.pop(0)
                    pool.wait_oldest()                           # (re-raises a worker's exception here)
                    x = pipe.run(slot, draws)
                    yield (x, labs, paths) if is_test else (x, labs)
            finally:
                pool.close()
                pipe.close()
        return batches()
