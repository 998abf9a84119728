"""Host-side logic of the restated graph: anchors, decode, top-k / NMS semantics, proposal and
detection post-processing branches, weight tables and FLOP accounting."""
import numpy as np
import pytest


def test_anchor_table(oracle):
    y, x, h, w = oracle.layer_anchors((480, 480), (30, 30))
    assert y.shape == (30, 30) and h.shape == (22,)
    assert y[0, 0] == np.float32(0.5 * 16 / 480) and x[3, 7] == np.float32(7.5 * 16 / 480)
    assert h[0] == np.float32(0.1) and w[0] == np.float32(0.1)            # extra scale first (:727-730)
    assert h[1] == np.float32(0.2) and w[1] == np.float32(0.2)            # s=.2 r=1
    assert np.isclose(h[2], 0.2 / np.sqrt(2)) and np.isclose(w[2], 0.2 * np.sqrt(2))   # r=2
    assert np.isclose(h[3], 0.2 / np.sqrt(.5))
    # zero deltas decode to the anchors themselves
    boxes = oracle.decode_all_anchors(np.zeros((1, 30 * 30 * 22, 4), np.float32), (y, x, h, w)).reshape(30, 30, 22, 4)
    assert np.allclose(boxes[4, 5, 1], [y[4, 5] - .1, x[4, 5] - .1, y[4, 5] + .1, x[4, 5] + .1], atol=1e-6)


def test_top_k_ties_lower_index_first(oracle):
    s = np.array([.5, .9, .5, .9, .1], np.float32)
    v, i = oracle.top_k(s, 4)
    assert list(i) == [1, 3, 0, 2]


def test_nms_semantics(oracle):
    b = np.array([[0, 0, 1, 1], [0, 0, 1, .9], [0, 0, .5, .5], [.6, .6, .9, .9], [0, 0, 0, 0], [1, 1, 0, 0]], np.float32)
    s = np.array([.9, .8, .7, .6, .5, .4], np.float32)
    # IoU(0,1)=.9 > .7 suppressed; IoU(0,2)=.25 kept; zero-area box never suppressed; flipped corners normalised
    assert list(oracle.non_max_suppression(b, s, 10, 0.7)) == [0, 2, 3, 4]
    assert list(oracle.non_max_suppression(b, s, 2, 0.7)) == [0, 2]
    # strict '>' : IoU exactly equal to the threshold survives
    b2 = np.array([[0, 0, 1, 1], [0, 0, 1, .5]], np.float32)
    assert list(oracle.non_max_suppression(b2, np.array([.9, .8], np.float32), 10, 0.5)) == [0, 1]
    assert oracle.iou_tf(b, 0, 5) == np.float32(1.0)


def test_get_proposals_branches(oracle):
    rng = np.random.default_rng(0)
    n = 200
    c = rng.uniform(.1, .9, (n, 2))
    hw = rng.uniform(.05, .3, (n, 2))
    boxes = np.concatenate([c - hw / 2, c + hw / 2], 1).astype(np.float32)
    scores = rng.uniform(.01, .99, n).astype(np.float32)
    tr = {}
    s, r = oracle.get_proposals_single(scores, boxes, 100, 50, 0.7, 16. / 480, tr)
    assert r.shape == (50, 4) and tr['n_cand'] == 100
    assert np.all(np.diff(tr['sorted_scores'][:100]) <= 0)
    if tr['n_keep'] < 50:      # upsample = tile of the kept set in order
        k = tr['n_keep']
        assert np.array_equal(r[:k], r[k:2 * k][:k]) or 2 * k > 50
    # nothing survives -> fallback box
    s0, r0 = oracle.get_proposals_single(scores, boxes * 0, 100, 8, 0.7, 16. / 480)
    assert np.all(r0 == np.array([.2, .2, .8, .8], np.float32))
    # a single survivor is tiled
    b1 = boxes * 0
    b1[7] = [.1, .1, .6, .6]
    s1, r1 = oracle.get_proposals_single(scores, b1, 100, 8, 0.7, 16. / 480)
    assert np.all(r1 == np.array([.1, .1, .6, .6], np.float32))


def test_bboxes_eval_shapes_and_padding(oracle):
    rng = np.random.default_rng(1)
    R = 40
    logits = (rng.standard_normal((R, 21)) * 3).astype(np.float32)
    c = rng.uniform(.2, .8, (R, 2))
    hw = rng.uniform(.1, .4, (R, 2))
    boxes = np.concatenate([c - hw / 2, c + hw / 2], 1).astype(np.float32)
    out = oracle.bboxes_eval(logits, boxes)
    assert sorted(out) == list(range(1, 21))
    for c_, (s, b) in out.items():
        assert s.shape == (200,) and b.shape == (200, 4)
        k = int((s > 0).sum())
        assert np.all(s[k:] == 0) and np.all(b[k:] == 0)
        assert np.all(np.diff(s[:k]) <= 0) and np.all(s[:k] > 0.01)
    assert oracle.filter_min_size((480, 480)) == np.float32(0.03)
    assert oracle.filter_min_size((10, 10)) == np.float32(0.03 * np.sqrt(np.float32(100) / np.float32(230400)))


def test_weight_tables_match_survey_counts():
    from xdet import weights as W
    def count(table, kinds):
        n = 0
        for name, kind, shape in table:
            if kind not in kinds:
                continue
            if kind == 'sep':
                n += 9 * shape[0] + shape[0] * shape[1]
            else:
                n += int(np.prod(shape)) + (shape[-1] if kind in ('convb', 'dense') else 0)
        return n
    xc = W.xception_conv_table()
    assert abs(count(xc, ('conv', 'sep')) - 20.75e6) < 0.02e6
    assert sum(1 for t in xc if t[1] == 'sep') == 34 and sum(1 for t in xc if t[1] == 'bn') == 40
    lh = W.lighthead_tables()
    rpn = [t for t in lh if t[0].startswith('rpn_head')]
    assert abs(count(rpn, ('convb',)) - 3.42e6) < 0.01e6
    ls = [t for t in lh if t[0].startswith('large_sep') and t[1] == 'convb']
    assert abs(count(ls, ('convb',)) - 19.49e6) < 0.01e6
    hd = [t for t in lh if t[0].startswith('final_head')]
    assert abs(count(hd, ('dense',)) - 1.06e6) < 0.01e6
    rn = W.resnet50_table()
    assert abs(count(rn, ('conv',)) - 23.45e6) < 0.06e6
    assert sum(1 for t in rn if t[1] == 'conv') == 53 and sum(1 for t in rn if t[1] == 'bn') == 49


def test_small_forward_runs_and_is_deterministic(oracle, lh_weights):
    from xdet import weights as W
    imgs = W.synthetic_images(1, 96, seed=5)
    tr = {}
    d1 = oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=20, trace=tr)
    d2 = oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=20)
    assert tr['mid'].shape == (1, 6, 6, 728) and tr['out'].shape == (1, 6, 6, 2048) and tr['feat'].shape == (1, 6, 6, 490)
    assert tr['proposals'].shape == (1, 20, 4) and tr['cls'].shape == (1, 20, 21)
    for c in d1[0]:
        assert np.array_equal(d1[0][c][0], d2[0][c][0]) and np.array_equal(d1[0][c][1], d2[0][c][1])


def test_resnet_trunk_shape(oracle):
    from xdet import weights as W
    w = W.make_resnet50_weights(4321)
    x = np.transpose(W.synthetic_images(1, 96, seed=6), (0, 2, 3, 1))
    y = oracle.resnet50_trunk(x, w)
    assert y.shape == (1, 3, 3, 2048) and np.all(y >= 0) and np.isfinite(y).all()
