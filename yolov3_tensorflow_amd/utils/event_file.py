"""Minimal TensorBoard event-file writer (the reference uses tf.summary.FileWriter, utils/board_callback.py:41-49,106-144; TensorFlow
is not a dependency here).  Implements exactly what that callback emits: Event records with a Summary holding simple_value scalars or a
HistogramProto, framed as TFRecords (little-endian length, masked CRC-32C of the length, payload, masked CRC-32C of the payload), in
files named ``events.out.tfevents.<unix time>.<host>`` starting with the ``brain.Event:2`` version record.

Wire format (protobuf field numbers from tensorflow/core/util/event.proto and framework/summary.proto):
  Event: 1 wall_time double, 2 step int64, 3 file_version string, 5 summary Summary
  Summary: 1 repeated Value;  Value: 1 tag string, 2 simple_value float, 5 histo HistogramProto
  HistogramProto: 1 min, 2 max, 3 num, 4 sum, 5 sum_squares (double), 6 bucket_limit, 7 bucket (packed repeated double)"""
import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _bytes_field(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _double_field(field, v):
    return _key(field, 1) + struct.pack('<d', float(v))


def encode_scalar_summary(tag, value):
    val = _bytes_field(1, tag.encode('utf-8')) + _key(2, 5) + struct.pack('<f', float(value))
    return _bytes_field(1, val)


def encode_histogram_summary(tag, hmin, hmax, num, total, sum_squares, bucket_limit, bucket):
    histo = (_double_field(1, hmin) + _double_field(2, hmax) + _double_field(3, num) + _double_field(4, total) +
             _double_field(5, sum_squares) +
             _bytes_field(6, b''.join(struct.pack('<d', float(x)) for x in bucket_limit)) +
             _bytes_field(7, b''.join(struct.pack('<d', float(x)) for x in bucket)))
    val = _bytes_field(1, tag.encode('utf-8')) + _bytes_field(5, histo)
    return _bytes_field(1, val)


def encode_event(wall_time, step=None, summary=None, file_version=None):
    ev = _double_field(1, wall_time)
    if step is not None:
        ev += _key(2, 0) + _varint(int(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode('utf-8'))
    if summary is not None:
        ev += _bytes_field(5, summary)
    return ev


class EventFileWriter(object):
    """tf.summary.FileWriter(logdir) for scalars and histograms"""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._write(encode_event(time.time(), file_version='brain.Event:2'))
        self.flush()

    def _write(self, payload):
        header = struct.pack('<Q', len(payload))
        self._f.write(header + struct.pack('<I', masked_crc32c(header)) + payload + struct.pack('<I', masked_crc32c(payload)))

    def add_scalar(self, tag, value, step):
        self._write(encode_event(time.time(), step, encode_scalar_summary(tag, value)))

    def add_histogram(self, tag, hmin, hmax, num, total, sum_squares, bucket_limit, bucket, step):
        self._write(encode_event(time.time(), step, encode_histogram_summary(tag, hmin, hmax, num, total, sum_squares, bucket_limit, bucket)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()


def read_records(path):
    """-> list of payload bytes; raises on a framing / CRC error (used by the tests and handy for inspection)"""
    out = []
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        header = data[pos:pos + 8]
        (n,) = struct.unpack('<Q', header)
        (hcrc,) = struct.unpack('<I', data[pos + 8:pos + 12])
        payload = data[pos + 12:pos + 12 + n]
        (pcrc,) = struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])
        if hcrc != masked_crc32c(header) or pcrc != masked_crc32c(payload) or len(payload) != n:
            raise ValueError('corrupt record at byte %d of %s' % (pos, path))
        out.append(payload)
        pos += 16 + n
    return out
