"""Build tests/golden/sample20_320.npz: the reference's 20 sample images (dataset/test_sample/images + label.txt, the data its own
train/test flow points at, configs.py:31-34) decoded with PIL, letterboxed to 320x320 with nearest-neighbour resize and with the
labels transformed accordingly (dataset/file_util.py:47-55 semantics, oracle/dataset.py; the decoded sizes are stored too) -- BASELINE.json configs[0].  Only data is stored (uint8 RGB
pixels + float32 labels); run in the build container:  python tests/golden/make_sample20_fixture.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from yolov3_tensorflow_amd.dataset.file_util import FileUtil   # noqa: E402
from oracle import dataset as ods                               # noqa: E402

REF = '/root/reference/dataset/test_sample'
names, labels = FileUtil._parse_label_file(os.path.join(REF, 'label.txt'))
imgs, sizes, labs = [], [], -np.ones((len(names), 8, 5), np.float32)
for i, (n, l) in enumerate(zip(names, labels)):
    raw = FileUtil.read_image(os.path.join(REF, 'images', n))
    lb = ods.transform_label(l, raw.shape[0], raw.shape[1], (320, 320))
    imgs.append(ods.letterbox(raw, (320, 320)))
    sizes.append(raw.shape[:2])
    labs[i, :len(lb)] = lb
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'sample20_320.npz'), images_rgb_u8=np.stack(imgs),
                    labels=labs.reshape(len(names), 40), names=np.array(names), src_hw=np.asarray(sizes, np.int32))
print(np.stack(imgs).shape, labs[0, :2])
