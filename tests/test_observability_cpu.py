"""Observability row: TensorBoard event-file writer (CRC-32C known answers, TFRecord framing, protobuf fields
decoded back), MyTensorBoard's histogram fields (utils/board_callback.py:112-144) and DetailLossLogger's message format
(utils/logger_callback.py:87-139)."""
import glob
import os
import re
import struct
import numpy as np
from yolov3_tensorflow_amd.utils import event_file as ef


def _parse(msg):
    """generic protobuf decode -> list of (field, wire, value)"""
    out, pos = [], 0
    while pos < len(msg):
        key, shift = 0, 0
        while True:
            b = msg[pos]; pos += 1
            key |= (b & 0x7F) << shift; shift += 7
            if not b & 0x80:
                break
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, shift = 0, 0
            while True:
                b = msg[pos]; pos += 1
                v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
        elif wire == 1:
            v = struct.unpack('<d', msg[pos:pos + 8])[0]; pos += 8
        elif wire == 5:
            v = struct.unpack('<f', msg[pos:pos + 4])[0]; pos += 4
        elif wire == 2:
            n, shift = 0, 0
            while True:
                b = msg[pos]; pos += 1
                n |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            v = msg[pos:pos + n]; pos += n
        else:
            raise ValueError(wire)
        out.append((field, wire, v))
    return out


def test_crc32c_known_answers():
    assert ef.crc32c(b'123456789') == 0xE3069283                     # the CRC-32C check value (RFC 3720 B.4 family)
    assert ef.crc32c(b'\x00' * 32) == 0x8A9136AA                      # RFC 3720 B.4: 32 bytes of zeros
    assert ef.crc32c(b'\xff' * 32) == 0x62A8AB43                      # RFC 3720 B.4: 32 bytes of ones
    assert ef.crc32c(bytes(range(32))) == 0x46DD794E                  # RFC 3720 B.4: incrementing bytes
    assert ef.masked_crc32c(b'') == 0xA282EAD8                        # mask(0) = rotate(0) + delta


def test_event_file_round_trip(tmp_path):
    w = ef.EventFileWriter(str(tmp_path / 'logs'))
    w.add_scalar('loss', 12.5, 3)
    w.add_histogram('bn_gamma', -1.0, 2.0, 5, 4.0, 9.5, [0.0, 1.0, 2.0], [1, 3, 1], 7)
    w.close()
    assert os.path.basename(w.path).startswith('events.out.tfevents.')
    recs = ef.read_records(w.path)
    assert len(recs) == 3
    first = dict((f, v) for f, _, v in _parse(recs[0]))
    assert first[3] == b'brain.Event:2' and first[1] > 1e9
    ev = dict((f, v) for f, _, v in _parse(recs[1]))
    assert ev[2] == 3
    val = dict((f, v) for f, _, v in _parse(_parse(ev[5])[0][2]))
    assert val[1] == b'loss' and val[2] == 12.5
    ev = dict((f, v) for f, _, v in _parse(recs[2]))
    assert ev[2] == 7
    val = dict((f, v) for f, _, v in _parse(_parse(ev[5])[0][2]))
    histo = dict((f, v) for f, _, v in _parse(val[5]))
    assert val[1] == b'bn_gamma' and (histo[1], histo[2], histo[3], histo[4], histo[5]) == (-1.0, 2.0, 5.0, 4.0, 9.5)
    assert struct.unpack('<3d', histo[6]) == (0.0, 1.0, 2.0) and struct.unpack('<3d', histo[7]) == (1.0, 3.0, 1.0)
    # a flipped payload byte is detected
    raw = bytearray(open(w.path, 'rb').read())
    raw[20] ^= 1
    open(w.path, 'wb').write(bytes(raw))
    try:
        ef.read_records(w.path)
        assert False, 'corruption not detected'
    except ValueError:
        pass


class _FakeLoss(object):
    rectified_coord_loss = np.array([0.0, 0.0, 0.0], np.float32)
    coord_loss_xy = np.array([1.5, 2.5, 3.5], np.float32)
    coord_loss_wh = np.array([0.25, 0.5, 0.75], np.float32)
    noobj_iou_loss = np.array([10.0, 20.0, 30.0], np.float32)
    obj_iou_loss = np.array([4.0, 5.0, 6.0], np.float32)
    class_loss = np.array([7.0, 8.0, 9.0], np.float32)


class _FakeModel(object):
    class optimizer:
        lr = 0.002

    def regularization_losses(self):
        return 2.5e-4, 20, 4.65, 30

    def bn_gammas(self):
        return np.linspace(0.5, 1.5, 101).astype(np.float32)


def test_board_callback_writes_reference_layout(tmp_path):
    from yolov3_tensorflow_amd.utils.board_callback import MyTensorBoard
    tb = MyTensorBoard(log_dir=str(tmp_path / 'tb'))
    tb.set_model(_FakeModel(), _FakeLoss())
    tb.on_epoch_end(0, {'loss': 99.0})
    tb.on_epoch_end(1, {'loss': 98.0})
    tb.on_train_end()
    dirs = sorted(d for d in os.listdir(tmp_path / 'tb') if os.path.isdir(tmp_path / 'tb' / d))
    assert len(dirs) == 19 and 'bn_gamma' in dirs and 'head_16_noobj_iou_loss' in dirs and 'head_8_rectified_loss' in dirs

    def scalars(d):
        out = []
        for rec in ef.read_records(glob.glob(os.path.join(str(tmp_path / 'tb'), d, 'events.out.tfevents.*'))[0])[1:]:
            ev = dict((f, v) for f, _, v in _parse(rec))
            val = dict((f, v) for f, _, v in _parse(_parse(ev[5])[0][2]))
            out.append((ev.get(2, 0), val[1], val.get(2), val.get(5)))
        return out
    main = scalars('')
    assert [(s, t) for s, t, _, _ in main] == [(0, b'loss'), (0, b'learning_rate'), (1, b'loss'), (1, b'learning_rate')]
    assert main[0][2] == 99.0 and abs(main[1][2] - 0.002) < 1e-9
    sub = scalars('head_16_noobj_iou_loss')
    assert [(s, t, v) for s, t, v, _ in sub] == [(0, b'loss', 20.0), (1, b'loss', 20.0)]            # every sub-loss is tagged 'loss' (:99-101)
    hist = scalars('bn_gamma')
    h = dict((f, v) for f, _, v in _parse(hist[0][3]))
    g = _FakeModel().bn_gammas()
    counts, edges = np.histogram(g, bins=1000)
    assert h[3] == 101.0 and h[1] == float(g.min()) and h[2] == float(g.max()) and abs(h[4] - float(g.sum())) < 1e-9
    np.testing.assert_array_equal(np.frombuffer(h[6], '<f8'), edges[1:])                         # first edge dropped (:131-134)
    np.testing.assert_array_equal(np.frombuffer(h[7], '<f8'), counts.astype(np.float64))


def test_detail_loss_logger_format():
    from yolov3_tensorflow_amd.utils.logger_callback import DetailLossLogger
    lg = DetailLossLogger(verbose=2)
    lg.set_model(_FakeModel(), _FakeLoss())
    lg.on_train_begin(3, 7)
    lg.on_epoch_begin(0)
    msg = lg.format({'lr': 0.001, 'loss': 234.35})
    lines = msg.split('\n')
    assert re.match(r'^ - \d+s - lr: 1\.0000e-03 - loss: 234\.3500 - gamma_regular_loss\(20\): 2\.5000e-04 - kernel_regular_loss\(30\): 4\.6500 -  : $',
                    lines[1]), lines[1]
    assert lines[2] == ' - head: /8:'
    assert lines[3] == (' - rectified_loss: 0.0000e+00 - xy_loss: 1.5000 - wh_loss: 0.2500 - noobj_iou_loss: 10.0000 - obj_iou_loss: 4.0000'
                        ' - cls_loss: 7.0000 -  : ')
    assert lines[4] == ' - head: /16:' and lines[6] == ' - head: /32:' and 'cls_loss: 9.0000' in lines[7]
