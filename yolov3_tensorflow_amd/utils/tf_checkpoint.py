"""TensorFlow checkpoint (tensor-bundle V2) reader / writer without TensorFlow.

The reference saves and restores weights through TensorFlow (yolov3/trainer.py:47-67 ``tf.train.latest_checkpoint`` + ``load_weights``, :90-91
``ModelCheckpoint(save_weights_only=True)`` -> ``model.save_weights('...ckpt')``), i.e. in this byte format:

  <prefix>.index                  an SSTable (the LevelDB table format of tensorflow/core/lib/io/table*): sorted key -> value
        ""                               BundleHeaderProto {num_shards, endianness, version}
        "<checkpoint key>"               BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
        "_CHECKPOINTABLE_OBJECT_GRAPH"   (object-based checkpoints, what Keras ``save_weights`` writes) a scalar string tensor holding the
                                         serialised TrackableObjectGraph: which variable (``full_name``) sits under which checkpoint key
  <prefix>.data-0000s-of-0000n    the tensors' raw little-endian bytes at [offset, offset + size)
  checkpoint                      text CheckpointState: model_checkpoint_path / all_model_checkpoint_paths

Table format: data blocks of prefix-compressed entries (varint32 shared, non_shared, value_len; key tail; value) with a restart point every
16 entries (every entry in the index block, as LevelDB / TensorFlow write it), then the uint32 restart offsets and their count; every block is followed by a 5-byte trailer (compression type 0, masked CRC-32C of
block + type); metaindex block, index block (separator key -> BlockHandle varint64 offset, size), 48-byte footer (two handles, padding, magic
0xdb4775248b80fb57).  Masked CRC = rotr15(crc32c) + 0xa282ead8.  Protobuf field numbers: tensor_bundle.proto, tensor_shape.proto,
trackable_object_graph.proto.

READING maps every variable to its Keras name through ``full_name`` of the object graph (or, for name-based ``tf.train.Saver`` files, the key
itself), so a checkpoint written by the reference loads whatever order Keras numbered its layers in.  WRITING numbers ``layer_with_weights-K``
in this package's layer-creation order; Keras numbers them in its functional-graph depth order, which is not reproduced -- TensorFlow users
read our files by variable name (``tf.train.load_checkpoint``), not through ``load_weights``.
PARITY UNPINNED against TensorFlow: it is not installed and the reference ships no checkpoint file; the format follows TensorFlow's sources /
the LevelDB table_format document and is checked by known-answer vectors (CRC-32C RFC 3720, hand-assembled blocks) and round trips."""
import ctypes
import os
import struct
import numpy as np

MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 262144              # tensorflow/core/lib/io/table_options.h
RESTART_INTERVAL = 16
HEADER_KEY = b''
OBJECT_GRAPH_KEY = b'_CHECKPOINTABLE_OBJECT_GRAPH'
DT_FLOAT, DT_STRING, DT_INT32, DT_INT64, DT_HALF, DT_BFLOAT16, DT_DOUBLE = 1, 7, 3, 9, 19, 14, 2
_NP = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8'), DT_HALF: np.dtype('<f2'), DT_DOUBLE: np.dtype('<f8')}
_DT = {v: k for k, v in _NP.items()}


# ------------------------------------------------------------------------------------------------------------------ checksums
def crc32c(data, seed=0):
    """CRC-32C through the native library (yolo_crc32c: slicing-by-8 on the host); ``data``: bytes / bytearray / contiguous ndarray"""
    from yolov3_tensorflow_amd import _lib
    lib = _lib.load()
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data)
        return lib.yolo_crc32c(ctypes.c_void_p(a.ctypes.data), a.nbytes, seed) if a.nbytes else seed & 0xFFFFFFFF
    b = bytes(data)
    return lib.yolo_crc32c(ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), len(b), seed) if b else seed & 0xFFFFFFFF


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def unmask(m):
    r = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------------------ varints / protobuf
def put_varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _field(num, wire):
    return put_varint((num << 3) | wire)


def pb_varint(num, value):
    return _field(num, 0) + put_varint(value)


def pb_bytes(num, payload):
    return _field(num, 2) + put_varint(len(payload)) + payload


def pb_fixed32(num, value):
    return _field(num, 5) + struct.pack('<I', value)


def pb_parse(buf):
    """-> list of (field number, wire type, value): value = int (varint / fixed) or bytes (length-delimited)"""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = get_varint(buf, pos)
        elif wire == 2:
            n, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise ValueError('truncated protobuf field')
            pos += n
        elif wire == 5:
            (v,) = struct.unpack_from('<I', buf, pos)
            pos += 4
        elif wire == 1:
            (v,) = struct.unpack_from('<Q', buf, pos)
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wire)
        out.append((num, wire, v))
    return out


def encode_header(num_shards=1):
    """BundleHeaderProto: 1 num_shards, 2 endianness (LITTLE = 0: default, not emitted), 3 version {1 producer = 1}"""
    return pb_varint(1, num_shards) + pb_bytes(3, pb_varint(1, 1))


def encode_shape(shape):
    """TensorShapeProto: 2 repeated dim {1 size}"""
    return b''.join(pb_bytes(2, pb_varint(1, int(d))) for d in shape)


def encode_entry(dtype, shape, shard_id, offset, size, crc_masked):
    """BundleEntryProto: 1 dtype, 2 shape, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32); proto3: zero-valued scalars are not emitted"""
    e = pb_varint(1, dtype) + pb_bytes(2, encode_shape(shape))
    if shard_id:
        e += pb_varint(3, shard_id)
    if offset:
        e += pb_varint(4, offset)
    if size:
        e += pb_varint(5, size)
    return e + pb_fixed32(6, crc_masked)


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for num, wire, v in pb_parse(buf):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            for n2, _, d in pb_parse(v):
                if n2 == 2:
                    size = 0
                    for n3, _, x in pb_parse(d):
                        if n3 == 1:
                            size = x
                    e['shape'].append(size)
                elif n2 == 3 and d:
                    raise ValueError('tensor of unknown rank')
        elif num == 3:
            e['shard_id'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc32c'] = v
        elif num == 7:
            e['slices'] += 1
    return e


def encode_object_graph(layers):
    """TrackableObjectGraph for a Keras model saved with ``save_weights``: node 0 = the model, one node per layer with weights (child
    ``layer_with_weights-K``), one node per variable (child named by the layer attribute) carrying the SerializedTensor
    {name 'VARIABLE_VALUE', full_name '<layer>/<attribute>', checkpoint_key 'layer_with_weights-K/<attribute>/.ATTRIBUTES/VARIABLE_VALUE'}.
    ``layers``: [(layer name, [attribute, ...])].  -> (serialised proto, {full_name: checkpoint_key})
    TrackableObject: 1 children {1 node_id, 2 local_name}, 2 attributes {1 name, 2 full_name, 3 checkpoint_key}"""
    nodes, keys = [b''], {}
    root_children = b''
    for k, (layer, attrs) in enumerate(layers):
        layer_id = len(nodes)
        nodes.append(None)
        children = b''
        for attr in attrs:
            var_id = len(nodes)
            full = '%s/%s' % (layer, attr)
            key = 'layer_with_weights-%d/%s/.ATTRIBUTES/VARIABLE_VALUE' % (k, attr)
            keys[full] = key
            nodes.append(pb_bytes(2, pb_bytes(1, b'VARIABLE_VALUE') + pb_bytes(2, full.encode()) + pb_bytes(3, key.encode())))
            children += pb_bytes(1, pb_varint(1, var_id) + pb_bytes(2, attr.encode()))
        nodes[layer_id] = children
        root_children += pb_bytes(1, pb_varint(1, layer_id) + pb_bytes(2, ('layer_with_weights-%d' % k).encode()))
    nodes[0] = root_children
    return b''.join(pb_bytes(1, n) for n in nodes), keys


def decode_object_graph(buf):
    """-> {checkpoint_key: full_name} of every SerializedTensor in the graph"""
    out = {}
    for num, _, node in pb_parse(buf):
        if num != 1:
            continue
        for n2, _, attr in pb_parse(node):
            if n2 != 2:
                continue
            fields = {n3: v for n3, _, v in pb_parse(attr)}
            if 3 in fields:
                out[fields[3].decode()] = fields.get(2, b'').decode()
    return out


# ------------------------------------------------------------------------------------------------------------------ SSTable
def _shortest_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator"""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xFF and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _short_successor(key):
    """leveldb BytewiseComparator::FindShortSuccessor"""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


class _BlockBuilder(object):
    def __init__(self, restart_interval=RESTART_INTERVAL):
        self.buf, self.restarts, self.counter, self.last = bytearray(), [0], 0, b''
        self.restart_interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.counter < self.restart_interval:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.counter += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def build_table(items, block_size=BLOCK_SIZE):
    """items: sorted [(key bytes, value bytes)] -> the table file's bytes"""
    out = bytearray()

    def write_block(contents):
        handle = (len(out), len(contents))
        trailer = b'\x00'
        out.extend(contents + trailer + struct.pack('<I', mask(crc32c(contents + trailer))))
        return handle

    # (LevelDB / TensorFlow table builders give the INDEX block a restart interval of 1: every handle entry carries its full key)
    data, index = _BlockBuilder(), _BlockBuilder(restart_interval=1)
    pending, last_key = None, None
    for key, value in items:
        if last_key is not None and key <= last_key:
            raise ValueError('table keys must be strictly increasing')
        if pending is not None:
            index.add(_shortest_separator(last_key, key), put_varint(pending[0]) + put_varint(pending[1]))
            pending = None
        data.add(key, value)
        last_key = key
        if data.size() >= block_size:
            pending = write_block(data.finish())
            data = _BlockBuilder()
    if not data.empty():
        pending = write_block(data.finish())
    if pending is not None:
        index.add(_short_successor(last_key), put_varint(pending[0]) + put_varint(pending[1]))
    meta = write_block(_BlockBuilder().finish())
    idx = write_block(index.finish())
    footer = put_varint(meta[0]) + put_varint(meta[1]) + put_varint(idx[0]) + put_varint(idx[1])
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<Q', MAGIC))
    return bytes(out)


def _read_block(buf, offset, size):
    contents = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(contents) != size or len(trailer) != 5:
        raise ValueError('truncated table block')
    if unmask(struct.unpack('<I', trailer[1:])[0]) != crc32c(contents + trailer[:1]):
        raise ValueError('table block checksum mismatch at byte %d' % offset)
    if trailer[0] != 0:
        raise ValueError('compressed table block (type %d): TensorFlow writes checkpoint indices uncompressed' % trailer[0])
    (nrestart,) = struct.unpack_from('<I', contents, size - 4)
    limit = size - 4 - 4 * nrestart
    pos, key, out = 0, b'', []
    while pos < limit:
        shared, pos = get_varint(contents, pos)
        non_shared, pos = get_varint(contents, pos)
        vlen, pos = get_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(buf):
    """-> [(key, value)] of a table file's bytes (every block checksum verified)"""
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != MAGIC:
        raise ValueError('not an SSTable (bad magic number)')
    footer = buf[-48:]
    _, pos = get_varint(footer, 0)
    _, pos = get_varint(footer, pos)
    ioff, pos = get_varint(footer, pos)
    isize, pos = get_varint(footer, pos)
    out = []
    for _, handle in _read_block(buf, ioff, isize):
        off, p = get_varint(handle, 0)
        size, _ = get_varint(handle, p)
        out.extend(_read_block(buf, off, size))
    return out


# ------------------------------------------------------------------------------------------------------------------ bundle
def _string_tensor_bytes(value):
    """scalar DT_STRING tensor on disk (tensor_bundle.cc WriteStringTensor): varint64 length, masked CRC-32C of the length (as uint32),
    then the bytes; the entry checksum covers the length word, that 4-byte checksum and the bytes"""
    n = len(value)
    crc = crc32c(struct.pack('<I', n) if n <= 0xFFFFFFFF else struct.pack('<Q', n))
    length_checksum = struct.pack('<I', mask(crc))
    crc = crc32c(length_checksum, crc)
    crc = crc32c(value, crc)
    return put_varint(n) + length_checksum + value, crc


def write_checkpoint(prefix, weights):
    """weights: ordered {'<layer>/<attribute>': ndarray} (Keras variable names) -> <prefix>.index + <prefix>.data-00000-of-00001 as Keras
    ``save_weights`` lays them out (object-based keys + the object graph).  Returns {full_name: checkpoint key}."""
    layers, seen = [], {}
    for full in weights:
        layer, attr = full.rsplit('/', 1)
        if layer not in seen:
            seen[layer] = len(layers)
            layers.append((layer, []))
        layers[seen[layer]][1].append(attr)
    graph, keys = encode_object_graph(layers)
    tensors = {keys[full].encode(): np.ascontiguousarray(np.asarray(a)) for full, a in weights.items()}
    items = [(HEADER_KEY, encode_header(1))]
    offset = 0
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    # both files are written under temporary names and renamed when complete: a crash mid-save leaves no torn checkpoint behind a name
    # that the 'checkpoint' state file (updated by the caller AFTER this returns) could point to
    tmp_data, tmp_index = prefix + '.data-00000-of-00001.tmp', prefix + '.index.tmp'
    with open(tmp_data, 'wb') as f:
        for key in sorted(list(tensors) + [OBJECT_GRAPH_KEY]):
            if key == OBJECT_GRAPH_KEY:
                payload, crc = _string_tensor_bytes(graph)
                items.append((key, encode_entry(DT_STRING, (), 0, offset, len(payload), mask(crc))))
            else:
                a = tensors[key]
                if a.dtype not in _DT:
                    a = a.astype('<f4')
                a = a.astype(a.dtype.newbyteorder('<'), copy=False)
                payload = a.tobytes()
                items.append((key, encode_entry(_DT[a.dtype], a.shape, 0, offset, len(payload), mask(crc32c(a)))))
            f.write(payload)
            offset += len(payload)
        f.flush()
        os.fsync(f.fileno())
    with open(tmp_index, 'wb') as f:
        f.write(build_table(sorted(items)))
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp_data, prefix + '.data-00000-of-00001')
    os.replace(tmp_index, prefix + '.index')
    return keys


def read_checkpoint(prefix):
    """-> {variable name: ndarray}.  Object-based checkpoints (Keras save_weights) are renamed through the object graph's ``full_name``;
    name-based ones (tf.train.Saver) already use the variable names as keys.  Every tensor's CRC-32C is verified."""
    with open(prefix + '.index', 'rb') as f:
        table = read_table(f.read())
    if not table or table[0][0] != HEADER_KEY:
        raise ValueError('%s.index has no bundle header' % prefix)
    header = {n: v for n, _, v in pb_parse(table[0][1])}
    num_shards = header.get(1, 0)
    if header.get(2, 0) != 0:
        raise ValueError('big-endian checkpoint')
    entries = {k: decode_entry(v) for k, v in table[1:]}
    shards = {}

    def shard(i):
        if i not in shards:
            shards[i] = np.memmap('%s.data-%05d-of-%05d' % (prefix, i, num_shards), dtype=np.uint8, mode='r')
        return shards[i]

    names = {}
    if OBJECT_GRAPH_KEY in entries:
        e = entries.pop(OBJECT_GRAPH_KEY)
        raw = bytes(shard(e['shard_id'])[e['offset']:e['offset'] + e['size']])
        n, pos = get_varint(raw, 0)
        graph = raw[pos + 4:pos + 4 + n]
        crc = crc32c(struct.pack('<I', n) if n <= 0xFFFFFFFF else struct.pack('<Q', n))
        if struct.unpack('<I', raw[pos:pos + 4])[0] != mask(crc) or unmask(e['crc32c']) != crc32c(graph, crc32c(raw[pos:pos + 4], crc)):
            raise ValueError('object graph checksum mismatch')
        names = decode_object_graph(graph)
    out = {}
    for key, e in entries.items():
        k = key.decode()
        if e['slices']:
            raise ValueError('%s: partitioned (sliced) variables are not supported' % k)
        if e['dtype'] not in _NP:
            continue                                   # e.g. the save counter (int64 is read; strings / resources are skipped)
        a = np.frombuffer(shard(e['shard_id'])[e['offset']:e['offset'] + e['size']], dtype=_NP[e['dtype']])
        if a.size != int(np.prod(e['shape'], dtype=np.int64)):
            raise ValueError('%s: %d bytes for shape %s' % (k, e['size'], e['shape']))
        if unmask(e['crc32c']) != crc32c(a):
            raise ValueError('%s: tensor checksum mismatch' % k)
        out[names.get(k, k) or k] = a.reshape(e['shape']).copy()
    return out


def exists(prefix):
    return os.path.exists(prefix + '.index')


# ------------------------------------------------------------------------------------------------------------------ CheckpointState
def update_checkpoint_state(directory, name, keep_all=True):
    """the ``checkpoint`` file tf.train.Saver / Keras maintain next to the shards (text CheckpointState)"""
    path = os.path.join(directory, 'checkpoint')
    previous = []
    if keep_all and os.path.exists(path):
        for line in open(path):
            if line.startswith('all_model_checkpoint_paths:'):
                previous.append(_unquote(line.split(':', 1)[1]))
    if name in previous:
        previous.remove(name)
    previous.append(name)
    with open(path, 'w') as f:
        f.write('model_checkpoint_path: %s\n' % _quote(name))
        for p in previous:
            f.write('all_model_checkpoint_paths: %s\n' % _quote(p))


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` (relative names resolve against the directory)"""
    path = os.path.join(directory, 'checkpoint')
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith('model_checkpoint_path:'):
            name = _unquote(line.split(':', 1)[1])
            return name if os.path.isabs(name) else os.path.join(directory, name)
    return None


def _quote(s):
    return '"%s"' % s.replace('\\', '\\\\').replace('"', '\\"')


def _unquote(s):
    s = s.strip()
    if len(s) >= 2 and s[0] == '"' and s[-1] == '"':
        s = s[1:-1]
    return s.replace('\\"', '"').replace('\\\\', '\\')
