"""F3: the from-scratch TensorFlow-V2 checkpoint (tensor bundle) reader.  No TensorFlow exists here to produce a
real checkpoint, so the pins are the public constants of the formats involved (CRC32C test vector and TF's mask,
the LevelDB table magic, protobuf wire bytes written out by hand) plus the committed bundle fixture
(tests/golden/tiny_bundle.*, written by tests/golden/make_tf_bundle_fixture.py)."""
import os
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))


def test_crc32c_known_answers():
    from xdet import tf_checkpoint as T
    assert T.crc32c(b'123456789') == 0xE3069283                   # the CRC-32C (Castagnoli) check value
    assert T.crc32c(b'\x00' * 32) == 0x8A9136AA                   # RFC 3720 B.4 test pattern
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert T.mask_crc(0) == 0xa282ead8                            # leveldb/TF mask: rot15 + delta
    assert T.crc32c(b'6789', T.crc32c(b'12345')) == 0xE3069283    # incremental


def test_protobuf_entry_bytes_by_hand():
    """BundleEntryProto{dtype: DT_FLOAT, shape: [3,3,4,1], offset: 300, size: 144, crc32c: 0x01020304}"""
    from xdet import tf_checkpoint as T
    shape = b'\x12\x02\x08\x03' * 2 + b'\x12\x02\x08\x04' + b'\x12\x02\x08\x01'
    msg = b'\x08\x01' + b'\x12' + bytes([len(shape)]) + shape + b'\x20\xac\x02' + b'\x28\x90\x01' + b'\x35\x04\x03\x02\x01'
    e = T._parse_entry(msg)
    assert e == {'dtype': 1, 'shape': (3, 3, 4, 1), 'shard_id': 0, 'offset': 300, 'size': 144, 'crc32c': 0x01020304,
                 'sliced': False}
    assert T._parse_header(b'\x08\x01\x10\x00\x1a\x02\x08\x01')['num_shards'] == 1


def test_committed_bundle_fixture_reads_back():
    from xdet import tf_checkpoint as T
    from make_tf_bundle_fixture import tensors
    rd = T.CheckpointReader(os.path.join(HERE, 'golden', 'tiny_bundle'))
    want = tensors()
    assert set(rd.get_variable_to_shape_map()) == set(want) and len(want) == 38
    for k, v in want.items():
        got = rd.get_tensor(k, verify_crc=True)
        assert got.dtype == v.dtype and np.array_equal(got, v), k
    assert rd.get_tensor('global_step') == 122320
    raw = open(os.path.join(HERE, 'golden', 'tiny_bundle.index'), 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57            # LevelDB table magic
    assert raw.count(b'xception_lighthead/block') <= 10                       # 37 keys, 10 full copies (5 block starts + 5 index keys): prefix-compressed


def test_snappy_blocks_are_understood():
    from xdet import tf_checkpoint as T
    # literal "abcd", copy(offset 4, len 8) with a 1-byte offset, literal "xyz", copy with 2-byte offset (offset 15, len 5)
    comp = bytes([20]) + bytes([3 << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([2 << 2]) + b'xyz' + \
        bytes([((5 - 1) << 2) | 2, 15, 0])
    assert T._snappy_decompress(comp) == b'abcdabcdabcdxyzabcda'


def test_corruption_is_detected(tmp_path):
    from xdet import tf_checkpoint as T
    p = str(tmp_path / 'ck')
    T.write_checkpoint(p, {'a/w': np.arange(12, dtype=np.float32).reshape(3, 4), 'b': np.int64(7)})
    rd = T.CheckpointReader(p)
    assert np.array_equal(rd.get_tensor('a/w', verify_crc=True), np.arange(12, dtype=np.float32).reshape(3, 4))
    d = bytearray(open(p + '.data-00000-of-00001', 'rb').read())
    d[5] ^= 0x40
    open(p + '.data-00000-of-00001', 'wb').write(bytes(d))
    with pytest.raises(T.CheckpointError):
        T.CheckpointReader(p).get_tensor('a/w', verify_crc=True)
    i = bytearray(open(p + '.index', 'rb').read())
    i[3] ^= 0x01
    open(p + '.index', 'wb').write(bytes(i))
    with pytest.raises(T.CheckpointError):
        T.CheckpointReader(p)
    with pytest.raises(T.CheckpointError):
        T.CheckpointReader(str(tmp_path / 'nothing_here'))


def test_lighthead_weights_round_trip_through_a_tf_checkpoint(tmp_path):
    """the whole eval graph's variable set (253 tensors at reduced widths would not exercise the names, so the real
    table at full size is used once): writer -> reader -> dict the detector takes, names/shapes checked, optimizer
    slots and global_step ignored, missing variables reported"""
    from xdet import weights as W
    w = W.make_lighthead_weights(1234)
    p = str(tmp_path / 'model.ckpt-1')
    W.save_weights_tf_checkpoint(p, w, global_step=1)
    got = W.load_weights_tf_checkpoint(p)
    assert set(got) == set(w)
    for k in ('block1_conv1/kernel', 'block9_sepconv2/pointwise_kernel', 'large_sep_feature/Branch_1/conv2d/kernel',
              'final_head/fc_loc/bias', 'batch_normalization_3/moving_variance'):
        assert np.array_equal(got[k], w[k]), k
    small = dict(w)
    del small['rpn_head/conv2d/bias']
    W.save_weights_tf_checkpoint(p, small)
    with pytest.raises(KeyError):
        W.load_weights_tf_checkpoint(p)


# ---- independent decoder / encoder: google.protobuf with descriptors written from the published .proto files --------
def _tf_bundle_messages():
    """BundleHeaderProto / BundleEntryProto / TensorShapeProto / TensorSliceProto / VersionDef as dynamic messages.
    Field numbers and types are the public schema (tensorflow/core/protobuf/tensor_bundle.proto,
    framework/tensor_shape.proto, framework/tensor_slice.proto, framework/versions.proto); nothing here shares code
    with xdet/tf_checkpoint.py's hand-rolled wire-format reader."""
    pb = pytest.importorskip('google.protobuf')
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = 'xdet_test_tensor_bundle.proto'
    fd.package = 'xdet_test'
    fd.syntax = 'proto3'

    def msg(name, fields, nested=()):
        m = fd.message_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = tname
        return m
    dim = descriptor_pb2.DescriptorProto()
    dim.name = 'Dim'
    for fname, num, ftype in (('size', 1, F.TYPE_INT64), ('name', 2, F.TYPE_STRING)):
        f = dim.field.add()
        f.name, f.number, f.type, f.label = fname, num, ftype, F.LABEL_OPTIONAL
    shape = msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.xdet_test.TensorShapeProto.Dim'),
                                     ('unknown_rank', 3, F.TYPE_BOOL, F.LABEL_OPTIONAL, '')])
    shape.nested_type.add().CopyFrom(dim)
    ext = descriptor_pb2.DescriptorProto()
    ext.name = 'Extent'
    for fname, num in (('start', 1), ('length', 2)):
        f = ext.field.add()
        f.name, f.number, f.type, f.label = fname, num, F.TYPE_INT64, F.LABEL_OPTIONAL
    sl = msg('TensorSliceProto', [('extent', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.xdet_test.TensorSliceProto.Extent')])
    sl.nested_type.add().CopyFrom(ext)
    msg('VersionDef', [('producer', 1, F.TYPE_INT32, F.LABEL_OPTIONAL, ''), ('min_consumer', 2, F.TYPE_INT32, F.LABEL_OPTIONAL, ''),
                       ('bad_consumers', 3, F.TYPE_INT32, F.LABEL_REPEATED, '')])
    msg('BundleHeaderProto', [('num_shards', 1, F.TYPE_INT32, F.LABEL_OPTIONAL, ''),
                              ('endianness', 2, F.TYPE_INT32, F.LABEL_OPTIONAL, ''),      # enum on the wire = varint
                              ('version', 3, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.xdet_test.VersionDef')])
    msg('BundleEntryProto', [('dtype', 1, F.TYPE_INT32, F.LABEL_OPTIONAL, ''),            # DataType enum = varint
                             ('shape', 2, F.TYPE_MESSAGE, F.LABEL_OPTIONAL, '.xdet_test.TensorShapeProto'),
                             ('shard_id', 3, F.TYPE_INT32, F.LABEL_OPTIONAL, ''),
                             ('offset', 4, F.TYPE_INT64, F.LABEL_OPTIONAL, ''),
                             ('size', 5, F.TYPE_INT64, F.LABEL_OPTIONAL, ''),
                             ('crc32c', 6, F.TYPE_FIXED32, F.LABEL_OPTIONAL, ''),
                             ('slices', 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.xdet_test.TensorSliceProto')])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:                                           # older protobuf
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName('xdet_test.' + n)) for n in ('BundleHeaderProto', 'BundleEntryProto')}


def test_fixture_index_decodes_identically_with_google_protobuf():
    """every value of the committed fixture's .index table, decoded by google.protobuf from the published schema, equals
    what the hand-rolled parser returns (names come from the SSTable layer, which protobuf does not touch)."""
    from xdet import tf_checkpoint as T
    M = _tf_bundle_messages()
    raw = open(os.path.join(HERE, 'golden', 'tiny_bundle.index'), 'rb').read()
    n = 0
    for key, val in T._table_items(raw):
        if key == b'':
            h = M['BundleHeaderProto']()
            h.ParseFromString(bytes(val))
            mine = T._parse_header(val)
            assert (h.num_shards, h.endianness) == (mine['num_shards'], mine['endianness'])
            assert h.version.producer == 1                   # kTensorBundleVersion
            continue
        e = M['BundleEntryProto']()
        e.ParseFromString(bytes(val))
        mine = T._parse_entry(val)
        assert mine == {'dtype': e.dtype, 'shape': tuple(d.size for d in e.shape.dim), 'shard_id': e.shard_id,
                        'offset': e.offset, 'size': e.size, 'crc32c': e.crc32c, 'sliced': len(e.slices) > 0}, key
        n += 1
    assert n == 38


def test_parser_reads_what_google_protobuf_writes():
    """the other direction: entries ENCODED by google.protobuf (64-bit offsets, a high shard id, a negative dim as a
    10-byte varint, named dims, a sliced variable, fields in a non-canonical order) through the hand-rolled reader."""
    from xdet import tf_checkpoint as T
    M = _tf_bundle_messages()
    rng = np.random.default_rng(5)
    for trial in range(50):
        e = M['BundleEntryProto']()
        e.dtype = int(rng.choice([1, 3, 9]))
        dims = [int(x) for x in rng.integers(0, 5000, rng.integers(0, 5))]
        if trial == 7:
            dims = [-1, 3]
        for i, d in enumerate(dims):
            dd = e.shape.dim.add()
            dd.size = d
            if trial % 5 == 0:
                dd.name = 'd%d' % i
        e.shard_id = int(rng.integers(0, 300))
        e.offset = int(rng.integers(0, 1 << 40))
        e.size = int(rng.integers(0, 1 << 33))
        e.crc32c = int(rng.integers(0, 1 << 32))
        if trial % 9 == 0:
            s = e.slices.add()
            x = s.extent.add()
            x.start, x.length = 0, 4
        got = T._parse_entry(e.SerializeToString())
        assert got == {'dtype': e.dtype, 'shape': tuple(dims), 'shard_id': e.shard_id, 'offset': e.offset,
                       'size': e.size, 'crc32c': e.crc32c, 'sliced': trial % 9 == 0}, trial
    h = M['BundleHeaderProto']()
    h.num_shards, h.endianness = 7, 1
    h.version.producer, h.version.min_consumer = 1, 0
    got = T._parse_header(h.SerializeToString())
    assert got['num_shards'] == 7 and got['endianness'] == 1
    # non-canonical field order is legal protobuf: size before dtype
    e = M['BundleEntryProto']()
    e.dtype, e.size = 1, 144
    a = e.SerializeToString()
    assert T._parse_entry(a[2:] + a[:2]) == T._parse_entry(a)
