"""F3 on the GPU: the detector driven from a TensorFlow-V2 checkpoint bundle.  A bundle written to disk (the 45 M
parameter variable set under the reference's `xception_lighthead/` scope plus `global_step` and optimizer slots, as
utility/train_helper.py:74-94 expects to find and ignore) is read back by xdet.weights.load_weights_tf_checkpoint
(xdet/tf_checkpoint.py: SSTable index, protobuf entries, CRC32C -- checked against google.protobuf in
tests/test_tf_checkpoint.py) and must give the same detector, bit for bit.  No TensorFlow-written checkpoint exists in
this image (the published model.ckpt-122320 is a Google-Drive link, README.md:4), so loading THAT file stays untested."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_detector_built_from_a_loaded_bundle_is_bit_identical(tmp_path, lh_weights):
    from xdet import weights as W
    from xdet import tf_checkpoint as T
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    prefix = str(tmp_path / 'model.ckpt-122320')
    tensors = {'xception_lighthead/%s' % k: np.asarray(v, np.float32) for k, v in lh_weights.items()}
    # what a training run leaves next to the model variables: must be ignored by the loader
    tensors['global_step'] = np.asarray(122320, np.int64)
    tensors['xception_lighthead/block1_conv1/kernel/Momentum'] = np.zeros((3, 3, 3, 32), np.float32)
    T.write_checkpoint(prefix, tensors)
    loaded = W.load_weights_tf_checkpoint(prefix, verify_crc=True)
    assert set(loaded) == set(lh_weights)
    imgs = W.synthetic_images(2, 256, seed=3)
    out = {}
    for name, w in (('dict', lh_weights), ('bundle', loaded)):
        for mode in ('f16x3', 'f32'):
            set_precision(mode)
            try:
                det = LightHeadDetector(w, image_size=256, max_batch=2, rpn_post_nms_top_n=100)
            finally:
                set_precision('f32')
            det.forward(imgs)
            out[(name, mode)] = (det.detections(2), det.buffer('feat', 2).numpy())
            del det
    for mode in ('f16x3', 'f32'):
        (s0, b0), f0 = out[('dict', mode)]
        (s1, b1), f1 = out[('bundle', mode)]
        assert (s0 > 0).sum() > 50
        assert np.array_equal(s0, s1) and np.array_equal(b0, b1) and np.array_equal(f0, f1), mode
