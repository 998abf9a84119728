"""BASELINE config 1 plumbing: the reference's single-image demo contract (light_head_simple_demo.py:57-69,
110-199: channels_last uint8 image, select_threshold 0.5, nms_topk 20, 1000 proposals, concatenated
(labels, scores, bboxes)) on the pixels of demo/test.jpg against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def demo_weights():
    """the seeded synthetic weights with the class head sharpened x4: at the demo's select_threshold of 0.5 the
    plain synthetic head (max class probability ~0.2) would select nothing and the test would be vacuous"""
    from xdet import weights as W
    w = dict(W.make_lighthead_weights(1234))
    w['final_head/fc_cls/kernel'] = w['final_head/fc_cls/kernel'] * np.float32(4.0)
    return w


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_demo_contract(prec, oracle):
    from xdet.demo import light_head_simple_demo, make_demo_detector, DEMO_FLAGS
    from xdet.runtime import set_precision
    img = np.load(os.path.join(HERE, 'golden', 'demo_test_u8.npz'))['image']
    assert img.dtype == np.uint8 and img.ndim == 3
    w = demo_weights()
    set_precision(prec)
    try:
        det = make_demo_detector(w)
    finally:
        set_precision('f32')
    labels, scores, bboxes = light_head_simple_demo(img, det)
    k = DEMO_FLAGS['nms_topk']
    assert labels.shape == (20 * k,) and scores.shape == (20 * k,) and bboxes.shape == (20 * k, 4)
    assert np.array_equal(labels, np.repeat(np.arange(1, 21), k))              # classes ascending, k slots each
    # the oracle on the oracle's own pre-processing of the same pixels, same flags, raw image shape for the size filter
    x = oracle.preprocess_for_eval(img, 480)                                    # [3,480,480]
    ref = oracle.lighthead_forward(x[None], w, rpn_post_nms_top_n=1000, select_threshold=0.5,
                                   nms_threshold=0.3, nms_topk=k, image_shapes=[img.shape[:2]])[0]
    total = matched = 0
    for c in range(1, 21):
        gs, gb = scores[(c - 1) * k:c * k], bboxes[(c - 1) * k:c * k]
        rs, rb = ref[c]
        kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
        assert np.all(gs[kg:] == 0) and np.all(gb[kg:] == 0)                    # zero padding behind the detections
        assert np.all(gs[:kg] > 0.5) and np.all(np.diff(gs[:kg]) <= 0)          # thresholded, sorted
        assert kg == kr, (c, kg, kr)
        used = np.zeros(kg, bool)
        for j in range(kr):
            d = np.where(used, np.inf, np.maximum(np.abs(gs[:kg] - rs[j]), np.abs(gb[:kg] - rb[j]).max(1)))
            assert d.min() < 1e-3, (c, j, float(d.min()))
            used[int(d.argmin())] = True
            matched += 1
        total += kr
    print('demo image: %d detections above 0.5, all matched within 1e-3' % total)
    assert total >= 5 and matched == total
    # the graph-replayed call returns the same arrays
    l2, s2, b2 = light_head_simple_demo(img, det, use_graph=True)
    assert np.array_equal(s2, scores) and np.array_equal(b2, bboxes)
