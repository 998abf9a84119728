"""Activation range of the split-precision path (VERDICT r2 missing #5 / next #7).  The reference computes in f32
everywhere and its BN-less edges -- the (15,1) -> (1,15) chain of large_sep_kernel and the RPN hidden layer
(net/xception_body.py:381-400,450-475) -- have nothing that keeps activations small; an f16 hi plane overflows beyond
65504.  calibrate() gives every split operand a power-of-two exponent chosen from a calibration batch (planes hold
x * 2^-e, the consumer folds 2^e back: exact), so a checkpoint whose mid tensors leave the f16 range still runs on the
16-bit MFMA path.  The weights here are rescaled by powers of two in a way that leaves the network FUNCTION unchanged
(conv_a * 2^k, conv_b * 2^-k around a bias / ReLU) but pushes the tensors in between beyond 65504."""
import numpy as np
import pytest

from test_gpu_e2e import match_detections

pytestmark = pytest.mark.gpu


def _hot_weights(w, k_lsep=17, k_rpn=18):
    h = dict(w)
    for br in ('Branch_0', 'Branch_1'):
        h['large_sep_feature/%s/conv2d/kernel' % br] = w['large_sep_feature/%s/conv2d/kernel' % br] * np.float32(2.0 ** k_lsep)
        h['large_sep_feature/%s/conv2d/bias' % br] = w['large_sep_feature/%s/conv2d/bias' % br] * np.float32(2.0 ** k_lsep)
        h['large_sep_feature/%s/conv2d_1/kernel' % br] = w['large_sep_feature/%s/conv2d_1/kernel' % br] * np.float32(2.0 ** -k_lsep)
    h['rpn_head/conv2d/kernel'] = w['rpn_head/conv2d/kernel'] * np.float32(2.0 ** k_rpn)
    h['rpn_head/conv2d/bias'] = w['rpn_head/conv2d/bias'] * np.float32(2.0 ** k_rpn)
    for n in ('conv2d_1', 'conv2d_2'):
        h['rpn_head/%s/kernel' % n] = w['rpn_head/%s/kernel' % n] * np.float32(2.0 ** -k_rpn)
    return h


def test_a_net_with_small_activations_is_left_alone(lh_weights):
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(2, 256, seed=3)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=256, max_batch=2, rpn_post_nms_top_n=100)
    finally:
        set_precision('f32')
    det.forward(imgs, use_graph=True)
    s0, b0 = det.detections(2)
    assert det.calibrate(imgs) == {}
    scales = det.plane_scales()
    assert len(scales) >= 40 and not any(scales.values())          # every split operand is listed, none is scaled
    det.forward(imgs, use_graph=True)
    s1, b1 = det.detections(2)
    assert np.array_equal(s0, s1) and np.array_equal(b0, b1)


@pytest.mark.parametrize('lsep', ['spectral', 'direct'])
def test_mid_tensors_beyond_the_f16_range(oracle, lh_weights, lsep):
    from xdet._lib import XdetError
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    hot = _hot_weights(lh_weights)
    imgs = W.synthetic_images(2, 256, seed=3)
    tr = {}
    ref = oracle.lighthead_forward(imgs, hot, rpn_post_nms_top_n=100, trace=tr)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(hot, image_size=256, max_batch=2, rpn_post_nms_top_n=100, large_sep=lsep)
        cool = LightHeadDetector(lh_weights, image_size=256, max_batch=2, rpn_post_nms_top_n=100, large_sep=lsep)
    finally:
        set_precision('f32')
    with pytest.raises(XdetError, match='non-finite'):
        det.forward(imgs)                                         # uncalibrated: overflow -> NaN -> loud
    scaled = det.calibrate(imgs[:1])                              # one image calibrates, two are run
    print('calibrated [%s]:' % lsep, scaled)
    names = ' | '.join(scaled)
    assert 'rpn_head/conv2d' in names and 'large_sep_feature' in names and all(e > 0 for e in scaled.values())
    assert len(scaled) <= 4, scaled                               # only the tensors that needed it
    got = det.forward(imgs, use_graph=True)
    feat = det.buffer('feat', 2).numpy()
    assert np.isfinite(feat).all()
    assert float(np.abs(feat - tr['feat']).max()) < 1e-3
    total = matched = extra = 0
    for i in range(2):
        t, m, e = match_detections(got[i], ref[i])
        total, matched, extra = total + t, matched + m, extra + e
    assert total > 50 and matched == total and extra == 0, (total, matched, extra)
    # the rescaling is function-preserving: the calibrated hot net agrees with the untouched net far below 1e-3
    base = cool.forward(imgs, use_graph=True)
    worst = max(float(np.abs(got[i][c][0] - base[i][c][0]).max()) for i in range(2) for c in range(1, 21))
    assert worst < 2e-5, worst


@pytest.mark.parametrize('lsep', ['spectral', 'direct'])
def test_check_range_agrees_with_a_calibrated_net(lh_weights, lsep):
    """after calibrate() f32 tensors beyond 65504 are legitimate -- their planes hold x * 2^-e, and the
    spectral products y1 / y2 are never split at all.  check_range validates what is actually turned into f16 (the planes,
    the fused blocks' on-CU operands, the inputs of register-split convs) and only NaN / inf elsewhere: a checkpoint that
    NEEDED the calibration passes the validation afterwards, and still fails it before."""
    from xdet._lib import XdetError
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    hot = _hot_weights(lh_weights)
    imgs = W.synthetic_images(2, 256, seed=3)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(hot, image_size=256, max_batch=2, rpn_post_nms_top_n=100, large_sep=lsep, check_range=True)
        plain = LightHeadDetector(hot, image_size=256, max_batch=2, rpn_post_nms_top_n=100, large_sep=lsep)
    finally:
        set_precision('f32')
    with pytest.raises(XdetError):
        det.forward(imgs)                                         # uncalibrated: the validation (or the NaN guard) fires
    assert det.calibrate(imgs[:1])
    plain.calibrate(imgs[:1])
    got = det.forward(imgs)                                       # calibrated: mid tensors of ~3e5 in f32, and no complaint
    ref = plain.forward(imgs)
    for i in range(2):
        for c in range(1, 21):
            assert np.array_equal(got[i][c][0], ref[i][c][0]) and np.array_equal(got[i][c][1], ref[i][c][1])
    # and it still catches a real overflow: an image far outside the calibration batch
    imgs[1] *= 1e30
    with pytest.raises(XdetError, match=r'image\(s\) \[1\]'):
        det.forward(imgs)
