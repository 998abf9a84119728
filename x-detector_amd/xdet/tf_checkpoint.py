"""From-scratch reader (and a minimal writer) for TensorFlow "V2" checkpoints -- tensor bundles -- so that the
weights the reference publishes (README.md:4; restored by tf.train.Saver in light_head_rfcn_eval.py:499 /
light_head_simple_demo.py:195, variable selection utility/train_helper.py:74-94) can drive this path without
a TensorFlow install.  No TF, no protobuf, no leveldb package: the three formats involved are small.

A checkpoint `<prefix>` is two kinds of file:
  <prefix>.index                  an SSTable (LevelDB table format) mapping
                                    ""              -> BundleHeaderProto {num_shards, endianness, version}
                                    "<variable name>" -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-0000i-of-0000n    raw little-endian tensor bytes, concatenated; entries point into them

SSTable layout (leveldb/table/format.h): data blocks, a metaindex block, an index block, a 48-byte footer
(two BlockHandles as varint64 pairs, zero padding, magic 0xdb4775248b80fb57).  A block is a run of
prefix-compressed entries [shared][non_shared][value_len] (varint32) + key suffix + value, then the restart
array (uint32 offsets + count), then a 1-byte compression type (0 none, 1 snappy) and a masked CRC32C.
Protobuf wire format: varint keys (field << 3 | wire type), wire types 0 varint, 1 fixed64, 2 length-delimited,
5 fixed32.  CRC32C mask (TF lib/hash/crc32c.h): rot15(crc) + 0xa282ead8.

Only what the reference's checkpoints need is interpreted: float32 / int64 / int32 dense tensors without slices;
everything else in the index (optimizer slots, global_step) is listed but only decoded on request.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}


class CheckpointError(ValueError):
    pass


# ---- CRC32C (Castagnoli), table driven; vectorised over byte columns for the data files ------------------------
def _crc_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
        t[i] = c
    return t


_CRC_T = _crc_table()
_CRC_L = [int(x) for x in _CRC_T]


def _crc32c_bytes(data, crc=0):
    c = crc ^ 0xffffffff
    t = _CRC_L
    for b in bytes(data):
        c = t[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


_CHUNK = 4096
_SHIFT = None      # 4 x 256 table: the register after _CHUNK zero bytes, per byte of the starting register


def _shift_tables():
    global _SHIFT
    if _SHIFT is None:
        c = (np.arange(256, dtype=np.uint32)[None, :] << (8 * np.arange(4, dtype=np.uint32))[:, None]).reshape(-1)
        for _ in range(_CHUNK):
            c = _CRC_T[c & 0xff] ^ (c >> 8)
        _SHIFT = c.reshape(4, 256)
    return _SHIFT


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like.  Large buffers (the 180 MB of a real checkpoint) are cut into 4 KB chunks whose
    registers are advanced in lock step by NumPy (one table lookup per byte COLUMN) and then folded together
    through the CRC's linearity: reg(A || B) = shift_|B|(reg(A)) xor reg_0(B)."""
    a = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, np.uint8)
    n = a.size
    if n < 16 * _CHUNK:
        return _crc32c_bytes(a.tobytes(), crc)
    m = n // _CHUNK
    body = a[:m * _CHUNK].reshape(m, _CHUNK)
    r = np.zeros(m, np.uint32)
    for j in range(_CHUNK):
        r = _CRC_T[(r ^ body[:, j]) & 0xff] ^ (r >> 8)
    sh = _shift_tables()
    sh = [[int(x) for x in row] for row in sh]
    state = (crc ^ 0xffffffff) & 0xffffffff
    for ri in r.tolist():
        state = sh[0][state & 0xff] ^ sh[1][(state >> 8) & 0xff] ^ sh[2][(state >> 16) & 0xff] ^ sh[3][state >> 24] ^ ri
    return _crc32c_bytes(a[m * _CHUNK:].tobytes(), state ^ 0xffffffff)


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


# ---- varints / protobuf ---------------------------------------------------------------------------------------
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise CheckpointError('varint too long')


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """yield (field number, wire type, value) of one serialized message"""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError('truncated protobuf field')
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError('unsupported protobuf wire type %d' % wt)
        yield f, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for f, _, v in _pb_fields(buf):
        if f == 2:                                   # repeated Dim dim = 2
            size = 0
            for g, _, x in _pb_fields(v):
                if g == 1:                           # int64 size = 1
                    size = _signed64(x)
            dims.append(size)
        elif f == 3 and v:                           # unknown_rank
            raise CheckpointError('tensor of unknown rank')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            e['shape'] = _parse_shape(v)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['sliced'] = True
    return e


def _parse_header(buf):
    h = {'num_shards': 1, 'endianness': 0, 'version': None}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            h['num_shards'] = v
        elif f == 2:
            h['endianness'] = v
        elif f == 3:
            h['version'] = v
    return h


# ---- snappy (raw block format), decompression only --------------------------------------------------------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError('corrupt snappy block')
        for _ in range(ln):                          # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError('snappy length mismatch')
    return bytes(out)


# ---- SSTable ----------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    if len(raw) != size or offset + size + 5 > len(buf):
        raise CheckpointError('block handle points outside the index file')
    ctype = buf[offset + size]
    if verify:
        want = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if mask_crc(crc32c(buf[offset:offset + size + 1])) != want:
            raise CheckpointError('index block checksum mismatch')
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise CheckpointError('unknown block compression %d' % ctype)


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError('block too small')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise CheckpointError('bad restart array')
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        val = bytes(block[pos:pos + vlen])
        pos += vlen
        yield key, val


def _table_items(buf, verify=True):
    if len(buf) < 48:
        raise CheckpointError('index file shorter than an SSTable footer')
    footer = buf[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError('not an SSTable (bad magic): is this a V1 checkpoint or a wrong file?')
    pos = 0
    _, pos = _get_varint(footer, pos)                # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for kv in _block_entries(_read_block(buf, boff, bsize, verify)):
            yield kv


class CheckpointReader(object):
    """tf.train.NewCheckpointReader's useful half: get_variable_to_shape_map(), has_tensor(), get_tensor()."""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        idx = prefix + '.index'
        if not os.path.exists(idx):
            raise CheckpointError('%s not found (pass the checkpoint PREFIX, e.g. model.ckpt-122320)' % idx)
        buf = open(idx, 'rb').read()
        self.entries = {}
        self.header = None
        for key, val in _table_items(buf, verify_index):
            if key == b'':
                self.header = _parse_header(val)
            else:
                self.entries[key.decode('utf-8')] = _parse_entry(val)
        if self.header is None:
            raise CheckpointError('bundle header entry missing')
        if self.header['endianness'] != 0:
            raise CheckpointError('big-endian bundles are not supported')
        self._shards = {}

    def get_variable_to_shape_map(self):
        return {k: list(e['shape']) for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def _shard(self, i):
        if i not in self._shards:
            path = '%s.data-%05d-of-%05d' % (self.prefix, i, self.header['num_shards'])
            if not os.path.exists(path):
                raise CheckpointError('data shard %s not found' % path)
            self._shards[i] = np.memmap(path, dtype=np.uint8, mode='r')
        return self._shards[i]

    def get_tensor(self, name, verify_crc=False):
        e = self.entries.get(name)
        if e is None:
            raise KeyError(name)
        if e['sliced']:
            raise CheckpointError('%s is stored as slices (partitioned variable): not supported' % name)
        dt = _DTYPES.get(e['dtype'])
        if dt is None:
            raise CheckpointError('%s has dtype enum %d: only float32 / int32 / int64 are supported' % (name, e['dtype']))
        n = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if n * dt.itemsize != e['size']:
            raise CheckpointError('%s: %d bytes stored, shape %s needs %d' % (name, e['size'], e['shape'], n * dt.itemsize))
        shard = self._shard(e['shard_id'])
        raw = shard[e['offset']:e['offset'] + e['size']]
        if raw.size != e['size']:
            raise CheckpointError('%s: data shard is truncated' % name)
        if verify_crc and e['crc32c'] is not None and mask_crc(crc32c(raw.tobytes())) != e['crc32c']:
            raise CheckpointError('%s: tensor checksum mismatch' % name)
        return np.frombuffer(raw.tobytes(), dtype=dt).reshape(e['shape']).copy()


# ---- writer (tests, and a way to hand a bundle to TF-side tools) -------------------------------------------------
def _pb_varint_field(f, v):
    return _put_varint(f << 3) + _put_varint(v)


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _build_block(items, restart_interval=16):
    out = bytearray()
    restarts = []
    last = b''
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_checkpoint(prefix, tensors, block_entries=24):
    """{name: ndarray} -> <prefix>.index + <prefix>.data-00000-of-00001 (one shard, no compression), byte layout
    as tensorflow/core/util/tensor_bundle writes it (entries sorted by name, CRC32C per tensor and per block)."""
    names = sorted(tensors)
    data = bytearray()
    items = [(b'', _pb_varint_field(1, 1) + _pb_varint_field(2, 0) +
              _pb_bytes_field(3, _pb_varint_field(1, 1)))]           # num_shards 1, little endian, version {producer 1}
    for n in names:
        a = np.asarray(tensors[n], order='C')
        if a.dtype not in _DT_OF:
            raise CheckpointError('%s: dtype %s not supported' % (n, a.dtype))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        shape = b''.join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in a.shape)
        entry = (_pb_varint_field(1, _DT_OF[a.dtype]) + _pb_bytes_field(2, shape) +
                 (_pb_varint_field(4, len(data)) if len(data) else b'') + _pb_varint_field(5, len(raw)) +
                 _put_varint((6 << 3) | 5) + struct.pack('<I', mask_crc(crc32c(raw))))
        items.append((n.encode('utf-8'), entry))
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    table = bytearray()
    index_items = []

    def emit(block):
        off = len(table)
        table.extend(block)
        table.append(0)                                               # kNoCompression
        table.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return _put_varint(off) + _put_varint(len(block))

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_items.append((chunk[-1][0], emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    table.extend(footer)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(table))
    return prefix
