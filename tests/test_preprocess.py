"""F1: light_head_preprocess_for_eval -- oracle properties on CPU, HIP == oracle on GPU."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def demo_image():
    return np.load(os.path.join(HERE, 'golden', 'demo_test_u8.npz'))['image']


def test_oracle_identity_size_is_pure_whitening(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    out = oracle.preprocess_for_eval(img, 64)
    means = np.array([123.68, 116.78, 103.94]) / 127.5
    ref = (img.astype(np.float64) / 255. * 2. - means).transpose(2, 0, 1)
    assert out.shape == (3, 64, 64) and out.dtype == np.float32
    assert np.abs(out - ref).max() < 1e-6


def test_oracle_demo_image_range_and_legacy_sampling(oracle):
    img = demo_image()
    assert img.shape == (333, 500, 3)
    out = oracle.preprocess_for_eval(img, 480)
    # whitened range quoted in SURVEY.md 8d: about [-0.97, 1.18]
    assert out.min() >= -0.98 and out.max() <= 1.19
    # legacy (non half-pixel) bilinear: output pixel (0,0) is input pixel (0,0) exactly
    means = (np.array([123.68, 116.78, 103.94]) / 127.5).astype(np.float32)
    p00 = (img[0, 0].astype(np.float32) * np.float32(1 / 255.)) * np.float32(2) - means
    assert np.array_equal(out[:, 0, 0], p00)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(333, 500), (500, 375), (375, 500), (480, 480), (97, 1013)])
def test_gpu_preprocess_matches_oracle_exactly(shape, oracle):
    from xdet import ops
    rng = np.random.default_rng(sum(shape))
    img = demo_image() if shape == (333, 500) else rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    out, labels, bboxes, bbox_img = ops.light_head_preprocess_for_eval(img, None, None, out_shape=[480, 480],
                                                                        data_format='NCHW')
    ref = oracle.preprocess_for_eval(img, 480)
    assert out.shape == (3, 480, 480)
    assert np.array_equal(out, ref)
    assert np.array_equal(bbox_img, np.array([0., 0., 1., 1.], np.float32))
    nhwc = ops.light_head_preprocess_for_test(img, [480, 480], data_format='NHWC')
    assert np.array_equal(nhwc, ref.transpose(1, 2, 0))


@pytest.mark.gpu
def test_gpu_demo_image_through_the_whole_path(oracle, lh_weights):
    """BASELINE config 1 plumbing: demo/test.jpg -> F1 -> forward (random-init weights: only the
    plumbing and GPU==oracle agreement are checked, not the 3-boat picture)."""
    from xdet import ops
    from xdet.model import LightHeadDetector
    img = demo_image()
    x = ops.light_head_preprocess_for_test(img, [480, 480], data_format='NCHW')[None]
    det = LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=300)
    got = det.forward(x)[0]
    ref = oracle.lighthead_forward(x, lh_weights, rpn_post_nms_top_n=300,
                                   image_shapes=[img.shape[:2]])[0]
    # image_shape only enters bboxes_eval's min-size filter; run the GPU tail with it too
    nd = sum(int((got[c][0] > 0).sum()) for c in got)
    assert nd > 0
    feat = det.buffer('feat', 1).numpy()
    tr = {}
    oracle.lighthead_forward(x, lh_weights, rpn_post_nms_top_n=300, trace=tr)
    assert np.abs(feat - tr['feat']).max() <= 1e-4 * max(1.0, np.abs(tr['feat']).max())
