"""Decode workers of the input pipeline (FileUtil.get_dataset(decode_procs=...)).  A worker is `python -m
yolov3_tensorflow_amd.dataset.decode_worker`: its own interpreter, started with subprocess (NOT multiprocessing: 'spawn' / 'forkserver'
children re-import the parent's __main__ script -- which in a training run has initialised the GPU), talking JSON lines over its
stdin / stdout.  Nothing heavy is imported here -- no torch, no GPU runtime: a worker is PIL + NumPy, it decodes a JPEG and writes the
pixels straight into the shared (and, in the parent, page-locked) staging memory of a DeviceImagePipeline slot.  The reference's
tf.data pipeline does the same job with AUTOTUNE map parallelism (dataset/file_util.py:80-88,113); Python threads stop scaling at ~3000
images/s here (the interpreter parts of PIL serialise), processes do not."""
import numpy as np

_shm = {}


def _attach(name):
    from multiprocessing import shared_memory
    shm = _shm.get(name)
    if shm is None:
        # (a worker only attaches: the parent created the segment and unlinks it; keep Python's resource tracker from unlinking it again)
        shm = shared_memory.SharedMemory(name=name)
        try:
            from multiprocessing import resource_tracker
            resource_tracker.unregister(shm._name, 'shared_memory')
        except Exception:
            pass
        if len(_shm) > 64:                       # slots that grew were replaced by new segments: drop the oldest handles
            old = next(iter(_shm))
            _shm.pop(old).close()
        _shm[name] = shm
    return shm


def probe_size(path):
    """(h, w) of an image file without decoding it (PIL parses the header only)"""
    from PIL import Image
    with Image.open(path) as im:
        w, h = im.size
    return int(h), int(w)


def decode_into(task):
    """task = (segment name, byte offset, h, w, path): decode the file as RGB uint8 into segment[offset : offset + h*w*3]"""
    name, offset, h, w, path = task
    from PIL import Image
    with Image.open(path) as im:
        if im.mode != 'RGB':
            im = im.convert('RGB')
        arr = np.asarray(im)
    if arr.shape != (h, w, 3):
        raise ValueError('%s decodes to %s, planned %s' % (path, arr.shape, (h, w, 3)))
    dst = np.ndarray((h, w, 3), np.uint8, buffer=_attach(name).buf, offset=offset)
    np.copyto(dst, arr)
    return 0


class DecodePool(object):
    """``procs`` worker interpreters.  submit(tasks) deals the tasks of one batch over the workers and returns a ticket; wait(ticket) blocks
    until every worker has answered for that batch (tickets complete in submission order) and raises what a worker raised.  Several batches
    may be in flight: the pipes buffer the (small) task lines."""

    def __init__(self, procs):
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ)
        env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
        self.workers = [subprocess.Popen([sys.executable, '-m', 'yolov3_tensorflow_amd.dataset.decode_worker'], stdin=subprocess.PIPE,
                                         stdout=subprocess.PIPE, env=env, cwd=root, bufsize=0) for _ in range(int(procs))]
        self.pending = []            # per ticket: the workers that got tasks

    def _call(self, w, msg):
        import json
        w.stdin.write((json.dumps(msg) + '\n').encode())

    def _reply(self, w):
        import json
        line = w.stdout.readline()
        if not line:
            raise RuntimeError('a decode worker died (exit code %s)' % w.poll())
        r = json.loads(line)
        if r.get('error'):
            raise RuntimeError('decode worker: ' + r['error'])
        return r.get('result')

    def probe_sizes(self, paths):
        n = len(self.workers)
        for k, w in enumerate(self.workers):
            self._call(w, {'op': 'probe', 'paths': paths[k::n]})
        out = [None] * len(paths)
        for k, w in enumerate(self.workers):
            out[k::n] = [tuple(x) for x in self._reply(w)]
        return out

    def submit(self, tasks):
        n = len(self.workers)
        used = []
        for k, w in enumerate(self.workers):
            part = tasks[k::n]
            if part:
                self._call(w, {'op': 'decode', 'tasks': part})
                used.append(w)
        self.pending.append(used)
        return len(self.pending) - 1

    def wait_oldest(self):
        used = self.pending.pop(0)
        err = None
        for w in used:                       # drain every worker's answer for this batch even if one failed: the streams stay in step
            try:
                self._reply(w)
            except RuntimeError as e:
                err = err or e
        if err is not None:
            raise err

    def close(self):
        for w in self.workers:
            try:
                w.stdin.close()
            except Exception:
                pass
        for w in self.workers:
            try:
                w.wait(timeout=2)
            except Exception:
                w.kill()
        self.workers = []


def _serve():
    import json
    import sys
    out = sys.stdout.buffer
    for line in sys.stdin.buffer:
        try:
            msg = json.loads(line)
            if msg['op'] == 'probe':
                res = [probe_size(p) for p in msg['paths']]
            else:
                res = [decode_into(tuple(t)) for t in msg['tasks']]
            out.write((json.dumps({'result': res}) + '\n').encode())
        except Exception as e:                                  # reported to the parent, which raises it; the worker keeps serving
            out.write((json.dumps({'error': '%s: %s' % (type(e).__name__, e)}) + '\n').encode())
        out.flush()


if __name__ == '__main__':
    _serve()
