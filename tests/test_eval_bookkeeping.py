"""F3 / F4 host logic: checkpoint-name import and the VOC TP/FP + AP bookkeeping."""
import numpy as np
import pytest


def test_voc_label_table():
    from xdet import evaluation as E
    assert E.VOC_LABELS['none'] == (0, 'Background')
    assert E.VOC_LABELS['boat'] == (4, 'Vehicle') and E.VOC_LABELS['tvmonitor'] == (20, 'Indoor')
    assert len(E.VOC_LABELS) == 21 and E.label2name_table()[15] == 'person'


def test_matching_rules():
    from xdet import evaluation as E
    g = np.array([[0, 0, .5, .5], [.5, .5, 1, 1], [0, .5, .5, 1]], np.float32)
    gl = np.array([1, 1, 2])
    diff = np.array([0, 1, 0])
    det = np.array([[0, 0, .5, .5],        # exact hit on gt0 -> TP
                    [0, 0, .5, .45],       # second hit on gt0 (IoU .9) -> FP (already matched)
                    [.5, .5, 1, 1],        # hits the difficult gt1 -> neither
                    [.6, 0, .9, .3],       # no overlap -> FP
                    [0, .5, .5, 1]], np.float32)   # gt2 is another class -> FP for class 1
    s = np.array([.9, .8, .7, .6, .5], np.float32)
    n, tp, fp = E.bboxes_matching(1, s, det, gl, g, diff)
    assert n == 1
    assert list(tp) == [True, False, False, False, False]
    assert list(fp) == [False, True, False, True, True]
    # IoU must be strictly greater than the threshold
    n, tp, fp = E.bboxes_matching(1, s[:1], np.array([[0, 0, .5, .25]], np.float32), gl, g, diff, 0.5)
    assert not tp[0] and fp[0]


def _brute_ap12(p, r):
    # area under the monotone envelope, evaluated on a dense recall grid
    env = np.array([p[r >= t].max() if np.any(r >= t) else 0. for t in np.linspace(0, 1, 200001)])
    return env.mean()


def test_average_precision_voc07_voc12():
    from xdet import evaluation as E
    rng = np.random.default_rng(0)
    tp = rng.random(400) < np.linspace(.9, .1, 400)
    fp = ~tp
    scores = np.linspace(1, .01, 400).astype(np.float32)
    p, r = E.precision_recall(220, tp, fp, scores)
    assert np.all(np.diff(r) >= 0) and r[-1] == tp.sum() / 220
    ap12 = E.average_precision_voc12(p, r)
    assert abs(ap12 - _brute_ap12(p, r)) < 2e-3
    ap07 = E.average_precision_voc07(p, r)
    ref07 = np.mean([p[r >= t].max() if np.any(r >= t) else 0. for t in np.arange(0, 1.1, .1)])
    assert abs(ap07 - ref07) < 1e-12
    # perfect detector
    assert E.average_precision_voc12(np.ones(5), np.linspace(.2, 1, 5)) == pytest.approx(1.0)
    assert E.average_precision_voc07(np.ones(5), np.linspace(.2, 1, 5)) == pytest.approx(1.0)


def test_streaming_accumulator_scores_padded_detector_output():
    from xdet import evaluation as E
    acc = E.StreamingTpFp()
    g = np.array([[.1, .1, .4, .4], [.5, .5, .9, .9]], np.float32)
    dets = {1: (np.zeros(200, np.float32), np.zeros((200, 4), np.float32)),
            2: (np.zeros(200, np.float32), np.zeros((200, 4), np.float32))}
    dets[1][0][:2] = [.9, .3]
    dets[1][1][:2] = [[.1, .1, .4, .4], [.5, .5, .9, .9]]
    for _ in range(3):
        acc.update_image(dets, np.array([1, 1]), g, np.array([0, 0]))
    ap07, ap12 = acc.average_precisions()
    assert acc.nobjects[1] == 6 and acc.scores[1].shape == (6,)        # zero padding never counted
    assert ap12[1] == pytest.approx(1.0) and ap07[1] == pytest.approx(1.0)
    assert acc.nobjects[2] == 0 and acc.scores[2].shape == (0,)


def test_checkpoint_name_import_roundtrip(tmp_path, lh_weights):
    from xdet import weights as W
    shapes = W.lighthead_variable_shapes()
    assert set(shapes) == set(lh_weights) and all(tuple(lh_weights[k].shape) == v for k, v in shapes.items())
    assert shapes['block5_sepconv2/depthwise_kernel'] == (3, 3, 728, 1)
    assert shapes['large_sep_feature/Branch_1/conv2d_1/kernel'] == (1, 15, 256, 490)
    small = {k: v for k, v in lh_weights.items()}
    path = str(tmp_path / 'ckpt.npz')
    W.save_weights_npz(path, small)
    extra = dict(np.load(path))
    extra['xception_lighthead/global_step:0'] = np.zeros((), np.int64)          # ignored
    np.savez(path, **extra)
    got = W.load_weights_npz(path)
    assert set(got) == set(lh_weights)
    assert all(np.array_equal(got[k], lh_weights[k]) for k in got)
    del extra['xception_lighthead/final_head/fc_loc/bias:0']
    np.savez(path, **extra)
    with pytest.raises(KeyError):
        W.load_weights_npz(path)


def test_bboxes_draw_on_img():
    """F4 drawing helper (utility/draw_toolbox.py:72-104): outline at int(coord*shape), class 0 and
    sub-pixel boxes skipped, image modified in place and returned."""
    from xdet.evaluation import bboxes_draw_on_img, COLORS_TABLEAU
    img = np.zeros((100, 200, 3), np.uint8)
    boxes = np.array([[0.2, 0.1, 0.8, 0.5], [0.1, 0.6, 0.9, 0.9], [0.5, 0.5, 0.504, 0.9]], np.float32)
    out = bboxes_draw_on_img(img, np.array([3, 0, 5]), np.array([0.9, 0.8, 0.7]), boxes, thickness=2)
    assert out is img
    col = np.array(COLORS_TABLEAU[3], np.uint8)
    assert np.array_equal(img[20, 60], col) and np.array_equal(img[80, 60], col)        # top / bottom edges
    assert np.array_equal(img[50, 20], col) and np.array_equal(img[50, 100], col)       # left / right edges
    assert not img[50, 60].any()                                                          # interior untouched
    assert not img[:, 110:].any()                                                         # class 0 and the thin box skipped


def test_against_the_references_own_voc_eval():
    """F4 pinned on reference-generated vectors: tests/golden/voc_eval_golden.npz holds a synthetic VOC-style set and
    what the reference's NumPy evaluation (voc_eval.py:98-265, run verbatim by tests/golden/make_voc_eval_golden.py)
    made of it.  The streaming matcher + PR/AP functions here must reproduce its recall / precision curves and both
    APs.  (Boxes are pixel (xmin, ymin, xmax, ymax) there, (ymin, xmin, ymax, xmax) here: IoU does not care.)"""
    import os
    from xdet import evaluation as E
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'voc_eval_golden.npz'))
    gt = g['gt']
    n_img = int(gt[:, 0].max()) + 1
    for ci, cls in enumerate(g['classes']):
        label = ci + 1
        det = g['%s_det' % cls]
        acc = E.StreamingTpFp()
        all_s, all_tp, all_fp, npos = [], [], [], 0
        for im in range(n_img):
            rows = gt[gt[:, 0] == im]
            glabels, gdiff = rows[:, 1].astype(int), rows[:, 2].astype(int)
            gboxes = rows[:, [4, 3, 6, 5]]
            d = det[det[:, 0] == im]
            d = d[np.argsort(-d[:, 1], kind='stable')]                    # per image in score order, as bboxes_eval emits
            acc.update_image({label: (d[:, 1], d[:, [3, 2, 5, 4]])}, glabels, gboxes, gdiff, 0.5)
            n, tp, fp = E.bboxes_matching(label, d[:, 1], d[:, [3, 2, 5, 4]], glabels, gboxes, gdiff, 0.5)
            npos += n
            all_s.append(d[:, 1]); all_tp.append(tp); all_fp.append(fp)
        # the reference's curves have one point per detection, also for the ignored ones (matched to a `difficult`
        # object: neither tp nor fp): rebuild exactly that from the matcher's per-image flags
        order = np.argsort(-np.concatenate(all_s))
        ctp = np.cumsum(np.concatenate(all_tp)[order].astype(np.float64))
        cfp = np.cumsum(np.concatenate(all_fp)[order].astype(np.float64))
        assert npos == acc.nobjects[label]
        assert np.allclose(ctp / npos, g['%s_rec' % cls], atol=1e-12), cls
        assert np.allclose(ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps), g['%s_prec' % cls], atol=1e-12), cls
        ap07, ap12 = acc.average_precisions()
        assert abs(ap07[label] - float(g['%s_ap07' % cls])) < 1e-12, (cls, ap07[label])
        assert abs(ap12[label] - float(g['%s_ap12' % cls])) < 1e-12, (cls, ap12[label])
