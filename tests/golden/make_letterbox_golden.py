"""Build tests/golden/letterbox_frames.npz: the reference's OWN letterboxed frames of its 20 sample images -- dataset/test_result/*.jpg, written by
YOLOv3PostProcessor.visualize (yolov3_post_process.py:199-204, path configs.py:101) from the 480 x 384 network input that
dataset/file_util.py:47-59 produced (tf.image.resize_image_with_pad, NEAREST_NEIGHBOR), with the detected boxes drawn on top.  They are the
one output of the reference's input pipeline that the reference holds, so they pin the letterbox SCALE, OFFSET and nearest-neighbour
SAMPLING of oracle/dataset.py against the reference itself (tests/test_postprocess_dataset_cpu.py).

Only data is stored: the 20 files' bytes (JPEG, 1.7 MB; decoded with PIL by the test) and their names.  Run in the build container:
    python tests/golden/make_letterbox_golden.py"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/dataset/test_result'
names = sorted(n for n in os.listdir(REF) if n.endswith('.jpg'))
blobs = [np.frombuffer(open(os.path.join(REF, n), 'rb').read(), dtype=np.uint8) for n in names]
offsets = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
np.savez(os.path.join(ROOT, 'tests', 'golden', 'letterbox_frames.npz'), names=np.array(names), jpeg_bytes=np.concatenate(blobs), offsets=offsets)
print(len(names), 'frames,', int(offsets[-1]), 'bytes')
