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
