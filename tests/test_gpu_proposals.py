"""RPN tail on the GPU vs the oracle.  Given identical scores/boxes every discrete step
(filter, top-k order, NMS keep set, upsample order) must match exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth_rpn(rng, n, hw=30, A=22, spread=2.0):
    cls = (rng.standard_normal((n, hw, hw, 2 * A)) * spread).astype(np.float32)
    box = (rng.standard_normal((n, hw, hw, 4 * A)) * 0.4).astype(np.float32)
    return cls, box


def test_rpn_decode_matches_oracle(oracle):
    from xdet import ops
    rng = np.random.default_rng(0)
    cls, box = synth_rpn(rng, 2)
    anchors = oracle.layer_anchors((480, 480), (30, 30))
    obj, boxes = ops.rpn_decode(cls, box, anchors)
    ref_obj = oracle.softmax(cls.reshape(-1, 2))[:, -1].reshape(2, -1)
    ref_boxes = oracle.decode_all_anchors(box.reshape(2, -1, 4), anchors)
    assert np.abs(obj - ref_obj).max() <= 2e-7
    assert np.abs(boxes - ref_boxes).max() <= 1e-6 * max(1.0, np.abs(ref_boxes).max())
    # AnchorCreator mirror == oracle anchors
    ac = ops.AnchorCreator([480, 480], [(30, 30)], [[0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]], [[0.1]], [[1., 2., .5]], [16])
    (y, x, h, w), = ac.get_all_anchors()[0]
    for a, b in zip((y, x, h, w), anchors):
        assert np.array_equal(a, b)


def _boxes_scores(rng, n, cnt, scale=0.25):
    cy, cx = rng.uniform(-0.1, 1.1, (n, cnt)), rng.uniform(-0.1, 1.1, (n, cnt))
    h, w = rng.uniform(0.0, scale, (n, cnt)) + 0.01, rng.uniform(0.0, scale, (n, cnt)) + 0.01
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    scores = rng.uniform(0.001, 0.999, (n, cnt)).astype(np.float32)
    return scores, boxes


@pytest.mark.parametrize('n,cnt,pre,post,thr', [(2, 19800, 5000, 300, 0.7), (1, 19800, 5000, 1000, 0.7),
                                                (3, 4000, 600, 100, 0.5), (1, 55000, 5000, 300, 0.7)])
def test_get_proposals_exact(n, cnt, pre, post, thr, oracle):
    from xdet import ops
    rng = np.random.default_rng(cnt + post)
    scores, boxes = _boxes_scores(rng, n, cnt)
    scores[:, ::7] = scores[:, 3:4]            # many exact score ties -> index tie-break
    rois, counts = ops.get_proposals(scores, boxes, None, pre, post, thr, 16. / 480, False, 'channels_first',
                                     return_counts=True)
    traces = []
    ref = oracle.get_proposals(scores, boxes, pre, post, thr, 16. / 480, traces)
    for i in range(n):
        assert counts[i, 1] == traces[i]['n_cand']
        assert counts[i, 2] == min(traces[i]['n_keep'], post)
    assert np.array_equal(rois, ref)


@pytest.mark.parametrize('tied', [19800, 9000])
def test_get_proposals_long_candidate_list(tied, oracle):
    """more than 8192 keys share the threshold bin (here: identical scores, order = anchor index): the LDS
    sort steps aside and the counting rank kernel takes over; result still exact"""
    from xdet import ops
    rng = np.random.default_rng(tied)
    scores, boxes = _boxes_scores(rng, 2, 19800)
    scores[:, :tied] = 0.625
    rois, counts = ops.get_proposals(scores, boxes, None, 5000, 300, 0.7, 16. / 480, False, 'channels_first',
                                     return_counts=True)
    ref = oracle.get_proposals(scores, boxes, 5000, 300, 0.7, 16. / 480)
    assert counts[:, 3].min() > 8192 or tied < 19800
    assert np.array_equal(rois, ref)


def _clustered(rng, n, cnt, centres=40, jitter=0.02, scale=0.3):
    """boxes in tight groups: many IoU > thr pairs and long suppression chains (what a real RPN emits)"""
    c = rng.uniform(0.1, 0.9, (n, centres, 2))
    hw = rng.uniform(0.08, scale, (n, centres, 2))
    k = rng.integers(0, centres, (n, cnt))
    cy = np.take_along_axis(c[..., 0], k, 1) + rng.normal(0, jitter, (n, cnt))
    cx = np.take_along_axis(c[..., 1], k, 1) + rng.normal(0, jitter, (n, cnt))
    h = np.take_along_axis(hw[..., 0], k, 1) * np.exp(rng.normal(0, 0.1, (n, cnt)))
    w = np.take_along_axis(hw[..., 1], k, 1) * np.exp(rng.normal(0, 0.1, (n, cnt)))
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    return rng.uniform(0.001, 0.999, (n, cnt)).astype(np.float32), boxes


@pytest.mark.parametrize('n', [1, 3, 12, 20, 40, 70])
@pytest.mark.parametrize('post', [300, 1000])
def test_get_proposals_overlap_heavy_every_cluster_size(n, post, oracle):
    """The proposal NMS runs as clusters of 16 / 8 / 4 / 2 / 1 workgroups per image depending on the batch
    (proposals.hip nms_cluster_size); every form must make the reference's decisions (net/xception_body.py:57-67) on
    inputs where most candidates ARE suppressed (all pre_n candidates get visited, the kept list never fills at 1000),
    and an image's result must not depend on the batch it ran in."""
    from xdet import ops
    rng = np.random.default_rng(100 * n + post)
    scores, boxes = _clustered(rng, n, 19800)
    rois, counts = ops.get_proposals(scores, boxes, None, 5000, post, 0.7, 16. / 480, False, 'channels_first',
                                     return_counts=True)
    m = min(n, 2)
    traces = []
    ref = oracle.get_proposals(scores[:m], boxes[:m], 5000, post, 0.7, 16. / 480, traces)
    assert np.array_equal(rois[:m], ref)
    for i in range(m):
        assert counts[i, 2] == min(traces[i]['n_keep'], post)
    if post == 1000:
        assert counts[:, 2].max() < 1000 and counts[:, 1].min() == 5000      # suppression-bound, as intended
    if n > 2:
        alone = ops.get_proposals(scores[n - 1:], boxes[n - 1:], None, 5000, post, 0.7, 16. / 480, False, 'channels_first')
        assert np.array_equal(alone[0], rois[n - 1])


def test_get_proposals_suppression_chain(oracle):
    """A staircase of boxes, each overlapping only its neighbours above the threshold: candidate j's fate depends on
    j-1's, whose fate depends on j-2's ... -- the longest dependency chain the panel resolve can meet (one fixed-point
    round per link).  Greedy keeps every second box."""
    from xdet import ops
    n_box = 1500
    step = 0.0004
    y0 = 0.05 + step * np.arange(n_box)
    # IoU(j, j+1) = (0.003 - 0.0004) / (0.003 + 0.0004) = 0.765 > 0.7; IoU(j, j+2) = 0.58
    boxes = np.stack([y0, np.full(n_box, 0.1), y0 + 0.003, np.full(n_box, 0.9)], -1).astype(np.float32)
    scores = np.linspace(0.99, 0.5, n_box).astype(np.float32)[None]
    boxes = boxes[None]
    rois, counts = ops.get_proposals(scores, boxes, None, 1500, 1000, 0.7, 0.001, False, 'channels_first', return_counts=True)
    traces = []
    ref = oracle.get_proposals(scores, boxes, 1500, 1000, 0.7, 0.001, traces)
    assert traces[0]['n_keep'] == n_box // 2 and counts[0, 2] == n_box // 2
    assert np.array_equal(rois, ref)


@pytest.mark.parametrize('n', [1, 70])          # a 16-workgroup cluster per image / one workgroup per image
def test_get_proposals_panel_edges(n, oracle):
    """Shapes at the seams of the panel NMS: candidate counts that are exact multiples of the 256 / 512-candidate panels and
    one beside them, a single candidate, post_n = 1, thresholds 0 and >= 1, identical boxes (IoU = 1), and boxes that touch
    without overlapping.  Same rois and counts as the oracle in every case (net/xception_body.py:57-67,196-213)."""
    from xdet import ops
    rng = np.random.default_rng(11 + n)

    def check(scores, boxes, pre, post, thr, min_size=16. / 480):
        rois, counts = ops.get_proposals(scores, boxes, None, pre, post, thr, min_size, False, 'channels_first', return_counts=True)
        tr = []
        m = min(scores.shape[0], 2)
        ref = oracle.get_proposals(scores[:m], boxes[:m], pre, post, thr, min_size, tr)
        assert np.array_equal(rois[:m], ref), (pre, post, thr)
        for i in range(m):
            assert counts[i, 1] == tr[i]['n_cand'] and counts[i, 2] == min(tr[i]['n_keep'], post), (pre, post, thr, counts[i], tr[i]['n_keep'])
        if scores.shape[0] > 2:             # the last image of the batch equals its single-image run
            alone = ops.get_proposals(scores[-1:], boxes[-1:], None, pre, post, thr, min_size, False, 'channels_first')
            assert np.array_equal(alone[0], rois[-1])

    for cnt in (1, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024, 1025):     # every anchor valid: n_cand = cnt
        cy, cx = rng.uniform(0.3, 0.7, (n, cnt)), rng.uniform(0.3, 0.7, (n, cnt))
        h, w = rng.uniform(0.1, 0.3, (n, cnt)), rng.uniform(0.1, 0.3, (n, cnt))
        boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
        scores = rng.uniform(0.01, 0.99, (n, cnt)).astype(np.float32)
        check(scores, boxes, 2000, 300, 0.7)
    scores, boxes = _clustered(rng, n, 3000)
    check(scores, boxes, 1500, 1, 0.7)              # the first candidate, nothing else
    check(scores, boxes, 1500, 600, 0.0)            # threshold 0: any overlap suppresses
    check(scores, boxes, 1500, 600, 1.0)            # IoU > 1 never: the first 600 candidates
    check(scores, boxes, 1500, 1500, 0.5)           # post_n = pre_n
    # identical boxes (IoU = 1) in runs of 5, and a row of boxes that share an edge (IoU = 0)
    k = 400
    base = np.stack([np.full(k, 0.2), 0.002 * np.arange(k), np.full(k, 0.6), 0.002 * np.arange(k) + 0.3], -1)
    boxes = np.repeat(base, 5, 0)[None].repeat(n, 0).astype(np.float32)
    scores = rng.permutation(5 * k)[None].repeat(n, 0).astype(np.float32) / (5 * k + 1) + 0.001
    check(scores.astype(np.float32), boxes, 2000, 300, 0.7)
    x0 = 0.05 * np.arange(18)
    tiles = np.stack([np.full(18, 0.1), x0, np.full(18, 0.9), x0 + 0.05], -1)
    boxes = tiles[None].repeat(n, 0).astype(np.float32)
    scores = np.linspace(0.9, 0.1, 18, dtype=np.float32)[None].repeat(n, 0)
    check(scores, boxes, 100, 50, 0.0)              # touching boxes: intersection 0, none suppressed even at threshold 0


def test_get_proposals_few_and_none(oracle):
    """fewer survivors than post_n -> tiled upsample; none -> the [.2,.2,.8,.8] fallback (:196-213)."""
    from xdet import ops
    rng = np.random.default_rng(1)
    scores, boxes = _boxes_scores(rng, 2, 500)
    boxes[0, 40:] = [0.5, 0.5, 0.5, 0.5]        # zero-size -> filtered; image 0 keeps <= 40 candidates
    boxes[1, :] = [0.3, 0.3, 0.3001, 0.3001]    # all below rpn_min_size -> nothing survives
    rois, counts = ops.get_proposals(scores, boxes, None, 300, 64, 0.7, 16. / 480, False, 'channels_first',
                                     return_counts=True)
    ref = oracle.get_proposals(scores, boxes, 300, 64, 0.7, 16. / 480)
    assert 0 < counts[0, 2] < 64 and counts[1, 2] == 0
    assert np.array_equal(rois, ref)
    assert np.all(rois[1] == np.array([.2, .2, .8, .8], np.float32))


def test_ext_decode_and_bboxes_eval(oracle):
    from xdet import ops
    rng = np.random.default_rng(2)
    R = 300
    _, rois = _boxes_scores(rng, 1, R, 0.5)
    rois = oracle.bboxes_clip([0, 0, 1, 1], rois[0])
    reg = (rng.standard_normal((R, 4)) * 0.2).astype(np.float32)
    dec = ops.ext_decode_rois(rois, reg)
    ref_dec = oracle.ext_decode_rois(rois, reg)
    assert np.abs(dec - ref_dec).max() <= 1e-6
    logits = (rng.standard_normal((R, 21)) * 3).astype(np.float32)
    for shape in ((480, 480), (333, 500)):
        got = ops.bboxes_eval(logits, ref_dec, shape)
        ref = oracle.bboxes_eval(logits, ref_dec, shape)
        for c in range(1, 21):
            gs, gb = got[c]
            rs, rb = ref[c]
            assert gs.shape == (200,) and gb.shape == (200, 4)
            assert (gs > 0).sum() == (rs > 0).sum(), c
            assert np.abs(gs - rs).max() <= 1e-6
            assert np.abs(gb - rb).max() <= 1e-6


@pytest.mark.parametrize('R,spread', [(1000, 0.7), (1000, 3.0), (1024, 0.3), (7, 3.0)])
def test_bboxes_eval_at_the_reference_operating_point(R, spread, oracle):
    """A12 with rpn_post_nms_top_n = 1000 (light_head_rfcn_eval.py:111) and with the kernel's maximum of 1024 ROIs: flat
    logits put hundreds of ROIs above the class threshold (the rank sort runs over the packed valid keys, the NMS mask over
    400 sorted candidates), peaked ones a handful; overlap-heavy boxes so that the per-class NMS has work.  Batched call ==
    per-image calls; scores and boxes as the oracle's (utility/eval_helper.py:449-506)."""
    from xdet import ops
    rng = np.random.default_rng(R + int(10 * spread))
    n = 3
    _, boxes = _clustered(rng, n, R, centres=12, jitter=0.03)
    boxes = np.stack([oracle.bboxes_clip([0, 0, 1, 1], b) for b in boxes])
    logits = (rng.standard_normal((n, R, 21)) * spread).astype(np.float32)
    got = ops.bboxes_eval(logits, boxes, (480, 480))
    n_valid = 0
    for i in range(n):
        ref = oracle.bboxes_eval(logits[i], boxes[i], (480, 480))
        alone = ops.bboxes_eval(logits[i], boxes[i], (480, 480))
        for c in range(1, 21):
            gs, gb = got[i][c]
            rs, rb = ref[c]
            assert np.array_equal(gs, alone[c][0]) and np.array_equal(gb, alone[c][1])
            assert (gs > 0).sum() == (rs > 0).sum(), (i, c)
            assert np.abs(gs - rs).max() <= 1e-6 and np.abs(gb - rb).max() <= 1e-6, (i, c)
            n_valid += int((rs > 0).sum())
    assert n_valid > 0


def test_non_finite_head_outputs_are_loud(oracle, lh_weights):
    """A NaN / inf head logit compares false against every threshold: the image would silently lose its detections.
    bboxes_eval marks the (image, class) slot NaN instead and the host raises.  An activation beyond the f16 range of the
    split-precision convs (|x| > 65504) becomes inf -> NaN -> 0 at the next ReLU; the check_range option
    validates every activation tensor and reports through the same channel."""
    from xdet import ops
    from xdet._lib import XdetError
    from xdet.model import LightHeadDetector
    from xdet import weights as W
    rng = np.random.default_rng(4)
    R = 64
    boxes = oracle.bboxes_clip([0, 0, 1, 1], _boxes_scores(rng, 1, R, 0.5)[1][0])
    logits = (rng.standard_normal((R, 21)) * 3).astype(np.float32)
    clean = ops.bboxes_eval(logits, boxes, (480, 480))
    assert all(np.isfinite(clean[c][0]).all() for c in range(1, 21))
    for bad in (np.nan, np.inf, -np.inf):
        lg = logits.copy()
        lg[17, 5] = bad
        got = ops.bboxes_eval(lg, boxes, (480, 480))
        assert all(np.isnan(got[c][0][0]) for c in range(1, 21)), bad
    # end to end: an image far outside the whitened range drives the activations beyond the f16 range
    from xdet.runtime import set_precision
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=256, max_batch=2, rpn_post_nms_top_n=50, check_range=True)
        quiet = LightHeadDetector(lh_weights, image_size=256, max_batch=2, rpn_post_nms_top_n=50)
    finally:
        set_precision('f32')
    imgs = W.synthetic_images(2, 256, seed=3)
    det.forward(imgs)                                   # fine
    imgs[1] *= 1e30
    with pytest.raises(XdetError, match=r'image\(s\) \[1\]'):
        det.forward(imgs)
    # without the validation pass the overflow is laundered by the next ReLU (max(NaN, 0) = 0): image 1's detections are
    # then garbage without a NaN in them -- which is what the option is for; image 0 is untouched either way
    out = quiet.forward(imgs)
    ref = quiet.forward(W.synthetic_images(2, 256, seed=3))
    assert all(np.array_equal(out[0][c][0], ref[0][c][0]) for c in range(1, 21))


def test_overflow_in_the_dft_domain_is_loud(lh_weights):
    """the spectral large-separable path keeps DFT-domain planes and an un-normalised (15,1) conv
    output that check_range did not see, and its inverse transform's ReLU turned a NaN bin into 0 -> silent zeros in
    `feat`.  Now the DFT tensors are part of the validation pass, and the inverse DFT's ReLU propagates NaN so that even
    WITHOUT check_range the overflow reaches the head logits and the always-on guard.  The (15,1) kernels are scaled so
    that only the tensors between the two spectral GEMMs leave the f16 range (the backbone output stays O(1))."""
    from xdet._lib import XdetError
    from xdet.model import LightHeadDetector
    from xdet import weights as W
    from xdet.runtime import set_precision
    w = dict(lh_weights)
    for br in ('Branch_0', 'Branch_1'):
        w['large_sep_feature/%s/conv2d/kernel' % br] = lh_weights['large_sep_feature/%s/conv2d/kernel' % br] * np.float32(2.0 ** 17)
    imgs = W.synthetic_images(2, 256, seed=3)
    set_precision('f16x3')
    try:
        checked = LightHeadDetector(w, image_size=256, max_batch=2, rpn_post_nms_top_n=50, large_sep='spectral', check_range=True)
        quiet = LightHeadDetector(w, image_size=256, max_batch=2, rpn_post_nms_top_n=50, large_sep='spectral')
        fine = LightHeadDetector(lh_weights, image_size=256, max_batch=2, rpn_post_nms_top_n=50, large_sep='spectral',
                                 check_range=True)
    finally:
        set_precision('f32')
    fine.forward(imgs)                                   # the unscaled net passes the (now wider) validation
    with pytest.raises(XdetError, match=r'image\(s\) \[0, 1\]'):
        checked.forward(imgs)
    with pytest.raises(XdetError, match=r'non-finite'):
        quiet.forward(imgs)                              # NaN survives the inverse DFT's ReLU and reaches the guard
    assert not np.isfinite(quiet.buffer('feat', 2).numpy()).all()
