"""TensorFlow checkpoint wire format (reference trainer.py:47-67,90-91 go through TensorFlow): utils/tf_checkpoint.py against
known-answer vectors and hand-assembled bytes.  TensorFlow is not installed and the reference ships no checkpoint file, so nothing here is
pinned against TensorFlow's own output (parity unpinned); what is pinned: CRC-32C (RFC 3720 vectors + leveldb's mask test values), the
LevelDB table layout (a block assembled by hand from the table_format document must parse, and the builder must emit exactly those bytes),
varint / protobuf encodings (hand-encoded BundleEntryProto), and round trips incl. multi-block tables and corruption detection."""
import os
import struct
import numpy as np
import pytest


def test_crc32c_known_answers_and_mask():
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    # RFC 3720 B.4 / leveldb crc32c_test.cc StandardResults
    assert tc.crc32c(bytes(32)) == 0x8a9136aa
    assert tc.crc32c(bytes([0xff]) * 32) == 0x62a8ab43
    assert tc.crc32c(bytes(range(32))) == 0x46dd794e
    assert tc.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
    assert tc.crc32c(b'123456789') == 0xe3069283
    assert tc.crc32c(b'hello world') == tc.crc32c(b' world', tc.crc32c(b'hello'))            # Extend
    assert tc.crc32c(np.arange(1000, dtype=np.float32)) == tc.crc32c(np.arange(1000, dtype=np.float32).tobytes())
    c = tc.crc32c(b'foo')                                                                     # leveldb crc32c_test.cc Mask
    assert tc.mask(c) != c and tc.mask(tc.mask(c)) != c and tc.unmask(tc.mask(c)) == c and tc.unmask(tc.unmask(tc.mask(tc.mask(c)))) == c
    assert tc.mask(0) == 0xa282ead8


def test_varint_and_entry_proto_bytes():
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    assert tc.put_varint(0) == b'\x00' and tc.put_varint(127) == b'\x7f' and tc.put_varint(128) == b'\x80\x01' and tc.put_varint(300) == b'\xac\x02'
    assert tc.get_varint(b'\xac\x02\x07', 0) == (300, 2)
    # BundleEntryProto{dtype: DT_FLOAT(1), shape{dim{size:3} dim{size:64}}, offset: 300, size: 768, crc32c: 0x01020304}, by hand:
    want = bytes([0x08, 0x01,                                   # field 1 varint 1
                  0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x40,   # field 2 len 8: two dim messages (field 2), size (field 1)
                  0x20, 0xac, 0x02,                             # field 4 varint 300
                  0x28, 0x80, 0x06,                             # field 5 varint 768
                  0x35, 0x04, 0x03, 0x02, 0x01])                # field 6 fixed32
    assert tc.encode_entry(tc.DT_FLOAT, (3, 64), 0, 300, 768, 0x01020304) == want
    e = tc.decode_entry(want)
    assert (e['dtype'], e['shape'], e['shard_id'], e['offset'], e['size'], e['crc32c']) == (1, [3, 64], 0, 300, 768, 0x01020304)
    assert tc.encode_header(1) == bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])       # num_shards 1, (endianness LITTLE = default), version{producer 1}


def _hand_table(k1, v1, k2, v2):
    """two keys in one data block, assembled from leveldb's doc/table_format.md: entries (shared, non_shared, value_len varints), restart array,
    block trailer, empty metaindex block, index block with FindShortSuccessor(last key), footer"""
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    shared = 0
    while shared < min(len(k1), len(k2)) and k1[shared] == k2[shared]:
        shared += 1
    block = bytes([0, len(k1), len(v1)]) + k1 + v1 + bytes([shared, len(k2) - shared, len(v2)]) + k2[shared:] + v2
    block += struct.pack('<I', 0) + struct.pack('<I', 1)                      # one restart point at 0
    out = block + b'\x00' + struct.pack('<I', tc.mask(tc.crc32c(block + b'\x00')))
    meta_off = len(out)
    meta = struct.pack('<I', 0) + struct.pack('<I', 1)
    out += meta + b'\x00' + struct.pack('<I', tc.mask(tc.crc32c(meta + b'\x00')))
    idx_off = len(out)
    succ = k2[:1].replace(k2[:1], bytes([k2[0] + 1]))                         # shortest successor of the last key: first byte + 1
    handle = bytes([0, len(block)])                                           # varint offset 0, varint size (< 128)
    idx = bytes([0, len(succ), len(handle)]) + succ + handle + struct.pack('<I', 0) + struct.pack('<I', 1)
    out += idx + b'\x00' + struct.pack('<I', tc.mask(tc.crc32c(idx + b'\x00')))
    footer = bytes([meta_off, len(meta), idx_off, len(idx)])
    return out + footer + bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)


def test_table_matches_hand_assembled_bytes():
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    k1, v1, k2, v2 = b'conv2d/bias', b'AA', b'conv2d/kernel', b'BBB'
    hand = _hand_table(k1, v1, k2, v2)
    assert tc.read_table(hand) == [(k1, v1), (k2, v2)]                        # the reader parses the documented layout
    assert tc.build_table([(k1, v1), (k2, v2)]) == hand                       # and the builder emits exactly it
    bad = bytearray(hand)
    bad[3] ^= 1
    with pytest.raises(ValueError):
        tc.read_table(bytes(bad))                                             # block checksum
    with pytest.raises(ValueError):
        tc.read_table(hand[:-1] + b'\x00')                                    # magic
    with pytest.raises(ValueError):
        tc.build_table([(k2, v2), (k1, v1)])                                  # keys must be sorted


def test_table_restarts_and_multiple_blocks():
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    rng = np.random.default_rng(0)
    items = sorted({('layer_with_weights-%d/%s/.ATTRIBUTES/VARIABLE_VALUE' % (i, a)).encode(): rng.bytes(int(rng.integers(1, 60)))
                    for i in range(120) for a in ('kernel', 'gamma', 'beta')}.items())
    one = tc.build_table(items)                                               # 360 entries in one block: 23 restart points
    assert tc.read_table(one) == items
    many = tc.build_table(items, block_size=512)                              # ~60 data blocks, separators between them
    assert tc.read_table(many) == items and len(many) > len(one)
    assert tc._shortest_separator(b'abcd', b'abzz') == b'abd' and tc._shortest_separator(b'ab', b'abc') == b'ab'
    assert tc._short_successor(b'\xff\xffa') == b'\xff\xffb' and tc._short_successor(b'\xff') == b'\xff'


def test_checkpoint_round_trip_by_keras_names(tmp_path):
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    rng = np.random.default_rng(1)
    w = {'conv2d/kernel': rng.normal(size=(3, 3, 3, 64)).astype(np.float32),
         'batch_normalization_v1/gamma': np.ones(64, np.float32), 'batch_normalization_v1/beta': np.zeros(64, np.float32),
         'batch_normalization_v1/moving_mean': rng.normal(size=64).astype(np.float32),
         'batch_normalization_v1/moving_variance': rng.uniform(size=64).astype(np.float32),
         'conv2d_1/kernel': rng.normal(size=(1, 1, 64, 8)).astype(np.float32),
         'yolov3_head_32/kernel': rng.normal(size=(1, 1, 8, 15)).astype(np.float32), 'yolov3_head_32/bias': np.zeros(15, np.float32)}
    prefix = str(tmp_path / 'models' / 'lp-recognition-x-  3- 16.20000.ckpt')                 # the reference's stem keeps its spaces
    keys = tc.write_checkpoint(prefix, w)
    assert sorted(os.listdir(tmp_path / 'models')) == [os.path.basename(prefix) + '.data-00000-of-00001', os.path.basename(prefix) + '.index']
    assert keys['conv2d/kernel'] == 'layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'
    assert keys['batch_normalization_v1/moving_variance'] == 'layer_with_weights-1/moving_variance/.ATTRIBUTES/VARIABLE_VALUE'
    assert keys['yolov3_head_32/bias'] == 'layer_with_weights-3/bias/.ATTRIBUTES/VARIABLE_VALUE'
    back = tc.read_checkpoint(prefix)
    assert sorted(back) == sorted(w)
    for k in w:
        assert back[k].dtype == np.float32 and back[k].shape == w[k].shape
        np.testing.assert_array_equal(back[k], w[k])
    # the index: header first, then sorted keys incl. the object graph; every data range inside the shard, no overlap
    table = tc.read_table(open(prefix + '.index', 'rb').read())
    assert table[0] == (b'', tc.encode_header(1)) and [k for k, _ in table] == sorted(k for k, _ in table)
    assert tc.OBJECT_GRAPH_KEY in [k for k, _ in table]
    spans = sorted((tc.decode_entry(v)['offset'], tc.decode_entry(v)['size']) for k, v in table[1:])
    assert spans[0][0] == 0 and all(a + s == b for (a, s), (b, _) in zip(spans, spans[1:]))
    assert spans[-1][0] + spans[-1][1] == os.path.getsize(prefix + '.data-00000-of-00001')
    # a flipped bit in the data shard is caught by the tensor's CRC
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[len(raw) // 2] ^= 0x10
    open(prefix + '.data-00000-of-00001', 'wb').write(raw)
    with pytest.raises(ValueError):
        tc.read_checkpoint(prefix)


def test_name_based_checkpoint_and_state_file(tmp_path):
    """a tf.train.Saver-style bundle (keys = variable names, no object graph, two shards) reads by name; the ``checkpoint`` state file follows
    tf.train.latest_checkpoint / CheckpointState text format"""
    from yolov3_tensorflow_amd.utils import tf_checkpoint as tc
    a, b = np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(4, dtype=np.int64)
    prefix = str(tmp_path / 'model.ckpt-7')
    open(prefix + '.data-00000-of-00002', 'wb').write(a.tobytes())
    open(prefix + '.data-00001-of-00002', 'wb').write(b'\x00' * 8 + b.tobytes())
    items = [(b'', tc.encode_header(2)),
             (b'conv2d/kernel', tc.encode_entry(tc.DT_FLOAT, a.shape, 0, 0, a.nbytes, tc.mask(tc.crc32c(a)))),
             (b'global_step', tc.encode_entry(tc.DT_INT64, b.shape, 1, 8, b.nbytes, tc.mask(tc.crc32c(b))))]
    open(prefix + '.index', 'wb').write(tc.build_table(items))
    back = tc.read_checkpoint(prefix)
    np.testing.assert_array_equal(back['conv2d/kernel'], a)
    np.testing.assert_array_equal(back['global_step'], b)
    d = str(tmp_path)
    assert tc.latest_checkpoint(d) is None
    tc.update_checkpoint_state(d, 'model.ckpt-5')
    tc.update_checkpoint_state(d, 'model.ckpt-7')
    assert open(os.path.join(d, 'checkpoint')).read() == ('model_checkpoint_path: "model.ckpt-7"\n'
                                                          'all_model_checkpoint_paths: "model.ckpt-5"\nall_model_checkpoint_paths: "model.ckpt-7"\n')
    assert tc.latest_checkpoint(d) == prefix
