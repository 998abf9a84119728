"""Whole Light-Head R-CNN forward on the GPU vs the oracle on the same seeded 480x480 inputs
(BASELINE config 3: 300 proposals).  north_star tolerance: boxes/scores within 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope='module', params=['f32', 'f16x3', 'f16x3+spectral'])
def run(request, oracle, lh_weights):
    """f32 = exact f32 MFMA; f16x3 = the product arithmetic with the direct large-separable convs (what a
    small max_batch selects); f16x3+spectral = the same with the large-separable convs in the DFT domain
    (what bench-size batches select)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(2, 480, seed=0)
    prec, _, lsep = request.param.partition('+')
    set_precision(prec)
    try:
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=2, rpn_post_nms_top_n=300,
                                large_sep=lsep or 'direct')
    finally:
        set_precision('f32')
    got = det.forward(imgs)
    trace = {}
    ref = oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=300, trace=trace)
    return det, got, ref, trace, imgs


def rel_err(a, b):
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


def test_dense_stages(run):
    det, got, ref, tr, _ = run
    n = 2
    mid_x = det.buffer('mid_x', n).numpy()
    assert rel_err(np.maximum(mid_x, 0), tr['mid']) < 1e-4
    assert rel_err(det.buffer('out', n).numpy(), tr['out']) < 1e-4
    rpn = det.buffer('rpn_out', n).numpy()
    assert rel_err(rpn[..., :44], tr['rpn_cls']) < 1e-4
    assert rel_err(rpn[..., 44:132], tr['rpn_box']) < 1e-4
    assert rel_err(det.buffer('feat', n).numpy(), tr['feat']) < 1e-4
    na = 30 * 30 * 22
    assert np.abs(det.flat('objectness', (n, na)) - tr['objectness']).max() < 1e-4
    assert np.abs(det.flat('rpn_boxes', (n, na, 4)) - tr['rpn_boxes']).max() < 1e-4


def test_proposals_and_head(run):
    det, got, ref, tr, _ = run
    n = 2
    props = det.flat('proposals', (n, 300, 4))
    # same proposal set (order may differ only where two scores are within float noise)
    for i in range(n):
        a = props[i][np.lexsort(props[i].T)]
        b = tr['proposals'][i][np.lexsort(tr['proposals'][i].T)]
        assert np.abs(a - b).max() < TOL
    if np.abs(props - tr['proposals']).max() < 1e-5:      # identical order -> compare the head row by row
        cr = det.buffer('cls_reg', n).numpy().reshape(n, 300, -1)
        assert np.abs(cr[..., :21] - tr['cls']).max() < 1e-3
        assert np.abs(cr[..., 21:25] - tr['reg']).max() < 1e-3
        assert np.abs(det.flat('head_boxes', (n, 300, 4)) - tr['head_boxes']).max() < TOL


def test_predictions_dict(run, oracle):
    """`predictions` of the EstimatorSpec (light_head_rfcn_eval.py:409-433): classes = argmax and probabilities = max
    of the softmax'd head scores, bboxes_predict = the decoded head boxes, per ROI -- against the oracle's `cls` /
    `head_boxes`, ROI by ROI (paired through the proposal boxes)."""
    det, got, ref, tr, _ = run
    n = 2
    pred = det.predictions(n)
    assert pred['classes'].shape == (n, 300) and pred['probabilities'].shape == (n, 300)
    assert pred['bboxes_predict'].shape == (n, 300, 4)
    props = det.flat('proposals', (n, 300, 4))
    prob = oracle.softmax(tr['cls'].reshape(-1, 21)).reshape(n, 300, 21)
    for i in range(n):
        # two proposals whose scores tie within float noise may swap places: pair every GPU ROI with the oracle ROI that
        # has the same box (duplicates of the upsample tail carry identical results, any of them will do)
        d = np.abs(props[i][:, None, :] - tr['proposals'][i][None, :, :]).max(-1)
        j = d.argmin(1)
        assert d[np.arange(300), j].max() < 1e-5
        assert np.abs(pred['probabilities'][i] - prob[i, j].max(-1)).max() < TOL
        assert np.abs(pred['bboxes_predict'][i] - tr['head_boxes'][i, j]).max() < TOL
        # the arg-max may differ only where the two best classes of a ROI tie within the tolerance
        diff = pred['classes'][i] != prob[i, j].argmax(-1)
        if diff.any():
            top2 = np.sort(prob[i, j], -1)[..., -2:]
            assert np.all((top2[..., 1] - top2[..., 0])[diff] < TOL), 'arg-max class differs without a tie'
    assert (pred['classes'] > 0).sum() > 0     # some ROI is not background


def match_detections(got, ref, tol=TOL):
    """set matching per class: every oracle detection needs a distinct GPU detection whose score
    and box agree within tol (two detections whose scores differ by float noise may legitimately
    swap places in the score-ordered output)."""
    total = matched = extra = 0
    for c in ref:
        gs, gb = got[c]
        rs, rb = ref[c]
        kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
        assert np.all(gs[kg:] == 0) and np.all(gb[kg:] == 0)            # zero padding
        used = np.zeros(kg, bool)
        for j in range(kr):
            d = np.maximum(np.abs(gs[:kg] - rs[j]), np.abs(gb[:kg] - rb[j]).max(1)) if kg else np.array([])
            d = np.where(used, np.inf, d)
            if kg and d.min() < tol:
                used[int(d.argmin())] = True
                matched += 1
        total += kr
        extra += kg - int(used.sum())
    return total, matched, extra


def iou(a, b):
    ih = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0)
    iw = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - ih * iw
    return ih * iw / ua if ua > 0 else 0.0


# The matcher below is FROZEN (round 5): it knows two causes of a differing class list, both discrete decisions of the
# greedy per-class NMS that float noise can flip, and no third one is to be added -- a new kind of difference is a bug
# in the kernel that produced it.
THRESHOLD_TIE_BAND = 5e-4   # |IoU - nms_thr| of an admissible threshold tie (the seed-777 pair: 0.3000 +- 0.0003, tools/diag_nms_tie.py)


def assert_match_or_score_tie(got, ref, nms_thr=0.3, tol=TOL, tie=2e-5, allow_threshold_tie=False, stats=None):
    """Every oracle detection must be matched by a distinct GPU detection within `tol` and vice versa -- EXCEPT in a
    class list where the greedy per-class NMS met a score tie: two overlapping candidates (IoU > nms_thr) whose scores
    differ by less than the float noise of the scores (`tie`; the feature maps agree to ~1e-5) may be visited in
    either order, the first one suppresses the other, and the keep set behind them changes -- or where a candidate's IoU
    with a higher-scoring kept box sits within float noise of the NMS threshold (see below).  Such a list is accepted
    only if that pair is actually found (an oracle-only and a GPU-only detection with near-equal scores that suppress
    each other) and every other differing detection overlaps a differing detection of the other side (knock-on of
    the swap).  Returns (total, matched, lists explained by a tie)."""
    total = matched = ties = 0
    for c in ref:
        gs, gb = got[c]
        rs, rb = ref[c]
        kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
        assert np.all(gs[kg:] == 0) and np.all(gb[kg:] == 0)
        used = np.zeros(kg, bool)
        un = []
        for j in range(kr):
            d = np.where(used, np.inf, np.maximum(np.abs(gs[:kg] - rs[j]), np.abs(gb[:kg] - rb[j]).max(1))) if kg else np.array([np.inf])
            if d.min() < tol:
                used[int(d.argmin())] = True
                matched += 1
            else:
                un.append(j)
        ex = [k for k in range(kg) if not used[k]]
        total += kr
        if not un and not ex:
            continue
        seeds = [(j, k) for j in un for k in ex if abs(float(rs[j]) - float(gs[k])) < tie and iou(rb[j], gb[k]) > nms_thr]
        if not seeds:
            # the other discrete decision of the greedy NMS: a candidate whose IoU with a higher-scoring KEPT box is AT the
            # threshold is suppressed on one side and kept on the other.  Off unless the caller asks for it (only the run
            # whose head GEMM sums in split-K order does: test_second_weight_set); accepted only if, for every differing
            # detection, the other side keeps a higher-scoring box whose IoU with it lies within THRESHOLD_TIE_BAND of the
            # threshold, and for at most two detections of the class.
            assert allow_threshold_tie, ('class %d: %d oracle-only / %d gpu-only detections and no score tie explains them'
                                         % (c, len(un), len(ex)))

            def at_threshold(score, box, kept_s, kept_b):
                return any(float(ks) > score - tie and abs(iou(box, kb) - nms_thr) <= THRESHOLD_TIE_BAND for ks, kb in zip(kept_s, kept_b))
            assert len(un) + len(ex) <= 2, (c, len(un), len(ex))
            for j in un:
                assert at_threshold(float(rs[j]), rb[j], gs[:kg], gb[:kg]), ('class %d: oracle-only detection, no NMS threshold tie' % c, float(rs[j]), rb[j].tolist())
            for k in ex:
                assert at_threshold(float(gs[k]), gb[k], rs[:kr], rb[:kr]), ('class %d: gpu-only detection, no NMS threshold tie' % c, float(gs[k]), gb[k].tolist())
            ties += 1
            if stats is not None:
                stats.setdefault('threshold_ties', []).append(c)
            continue
        assert len(un) + len(ex) <= 8, (c, len(un), len(ex))
        for j in un:
            assert any(iou(rb[j], gb[k]) > nms_thr for k in ex), (c, 'oracle-only detection not explained', float(rs[j]))
        for k in ex:
            assert any(iou(gb[k], rb[j]) > nms_thr for j in un), (c, 'gpu-only detection not explained', float(gs[k]))
        ties += 1
        if stats is not None:
            stats.setdefault('score_ties', []).append(c)
    return total, matched, ties


def test_detections_within_1e3(run):
    det, got, ref, tr, _ = run
    total = matched = extra = 0
    for i in range(2):
        t, m, e = match_detections(got[i], ref[i])
        total, matched, extra = total + t, matched + m, extra + e
    print('detections: oracle %d, matched within 1e-3: %d, unmatched gpu: %d' % (total, matched, extra))
    assert total > 100
    assert matched == total and extra == 0, (matched, total, extra)


def test_graph_replay_equals_eager(run):
    det, got, _, _, imgs = run
    det.set_images(imgs)
    det.forward_device(2, use_graph=True)
    s1, b1 = det.detections(2)
    det.forward_device(2, use_graph=True)       # replay
    s2, b2 = det.detections(2)
    det.forward_device(2, use_graph=False)
    s3, b3 = det.detections(2)
    assert np.array_equal(s1, s2) and np.array_equal(b1, b2)
    assert np.array_equal(s1, s3) and np.array_equal(b1, b3)


def test_graph_builder_api_mirrors_reference(run, oracle, lh_weights):
    """XceptionBody / get_rpn / large_sep_kernel / get_proposals / get_head with the reference's
    signatures (net/xception_body.py:236,381,402,450,477), fed stage by stage."""
    from xdet import model as M
    det, _, _, tr, imgs = run
    with det.scope():
        mid, out = M.XceptionBody(imgs, 21, is_training=False, data_format='channels_first')
        assert rel_err(mid.numpy(), tr['mid']) < 1e-4
        cls, box = M.get_rpn(mid, 22, False, 'channels_first', 'rpn_head')
        assert cls.shape[1:] == (30, 30, 44) and box.shape[1:] == (30, 30, 88)
        assert rel_err(box.numpy(), tr['rpn_box']) < 1e-4
        feat = M.large_sep_kernel(out, 256, 490, False, 'channels_first', 'large_sep_feature')
        # feed the oracle's scores/boxes: the discrete stage must then agree exactly
        props = M.get_proposals(tr['objectness'], tr['rpn_boxes'], None, 5000, 300, 0.7, 16. / 480, False,
                                'channels_first')
        assert np.array_equal(props, tr['proposals'])
        c, r = M.get_head(tr['feat'], None, 7, 7, None, tr['proposals'], 21, False, False, 0, 'channels_first',
                          'final_head')
        assert np.abs(c - tr['cls']).max() < 2e-4 and np.abs(r - tr['reg']).max() < 2e-4


@pytest.mark.parametrize('size,R,lsep', [(800, 300, 'direct'), (480, 1000, 'direct'), (800, 300, 'spectral'),
                                         (256, 100, 'spectral')])
def test_other_baseline_configs(size, R, lsep, oracle, lh_weights):
    """BASELINE config 5 shape (800x800 -> 50x50 map, 55,000 anchors) and the reference's default
    rpn_post_nms_top_n=1000 (light_head_rfcn_eval.py:111), default product arithmetic (f16x3)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(1, size, seed=size)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=size, max_batch=1, rpn_post_nms_top_n=R, large_sep=lsep)
    finally:
        set_precision('f32')
    got = det.forward(imgs)
    tr = {}
    ref = oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=R, trace=tr)
    fm = size // 16
    assert tr['feat'].shape == (1, fm, fm, 490)
    assert rel_err(det.buffer('feat', 1).numpy(), tr['feat']) < 1e-4
    props = det.flat('proposals', (1, R, 4))
    # proposals as sets (two scores within float noise may swap places in the order): every proposal of the
    # oracle's set is in the GPU's set; a mismatch is only admissible where an NMS decision sat on the threshold
    d = np.abs(props[0][:, None, :] - tr['proposals'][0][None, :, :]).max(-1)
    n_same = int((d.min(1) < TOL).sum())
    print('size %d R %d [%s]: proposals with a match in the oracle set: %d / %d' % (size, R, lsep, n_same, R))
    if n_same != R:
        explain_proposal_mismatch(oracle, tr, props[0])
    total, matched, extra = match_detections(got[0], ref[0])
    print('size %d R %d [%s]: oracle detections %d matched %d extra %d' % (size, R, lsep, total, matched, extra))
    assert total > (20 if size >= 480 else 0)
    if n_same == R:
        assert matched == total and extra == 0, (matched, total, extra)
    else:       # one flipped keep/suppress decision changes a handful of head inputs
        assert matched >= total - max(2, total // 50) and extra <= max(2, total // 50)


def explain_proposal_mismatch(oracle, tr, gpu_props):
    """A proposal set may differ from the oracle's ONLY through an NMS decision that sits on the 0.7 threshold:
    a GPU-only box must have, among the oracle's proposals, a partner whose IoU with it is within 1e-5 of the
    threshold (the suppressor that won on one side and lost on the other).  Anything else is a real bug."""
    ref = tr['proposals'][0]
    d = np.abs(gpu_props[:, None, :] - ref[None, :, :]).max(-1)
    only_gpu = gpu_props[d.min(1) >= TOL]
    assert 0 < len(only_gpu) <= 3, len(only_gpu)
    for b in only_gpu:
        both = np.concatenate([b[None], ref])
        ious = np.array([oracle.iou_tf(both, 0, j) for j in range(1, len(both))])
        near = np.abs(ious - 0.7).min()
        print('  GPU-only proposal %s: closest oracle-side IoU to the 0.7 threshold: |IoU - 0.7| = %.2e' % (b, near))
        assert near < 1e-5, near


def test_net_misuse_is_reported_not_executed(lh_weights):
    """C-ABI error behaviour of the net handle: argument / state errors come back as codes
    (XdetError / InvalidArgumentError), nothing is launched, and the handle stays usable."""
    import ctypes
    import xdet
    from xdet._lib import lib, check, LightHeadConfig, c_void_p
    from xdet.model import LightHeadDetector
    # forward before build
    cfg = LightHeadConfig(image_size=480, max_batch=1, rpn_post_nms_top_n=300)
    h = c_void_p()
    check(lib().xdet_net_create(ctypes.byref(h), ctypes.byref(cfg)))
    with pytest.raises(xdet.XdetError):
        check(lib().xdet_net_forward(h, None, 1, None, None, None, None, 0, None))
    with pytest.raises(xdet.XdetError):
        check(lib().xdet_net_build(h))                      # no weights were set
    check(lib().xdet_net_destroy(h))
    # a weight with the wrong shape / an unknown variable name
    bad = dict(lh_weights)
    bad['block1_conv1/kernel'] = np.zeros((3, 3, 3, 31), np.float32)
    with pytest.raises(xdet.XdetError):
        LightHeadDetector(bad, image_size=480, max_batch=1, rpn_post_nms_top_n=300)
    det = LightHeadDetector(lh_weights, image_size=480, max_batch=2, rpn_post_nms_top_n=300)
    with pytest.raises(xdet.XdetError):
        det.buffer('no_such_buffer')
    with pytest.raises(xdet.XdetError):
        det.forward_device(3)                               # batch > max_batch
    with pytest.raises(xdet.XdetError):
        det.forward_device(0)
    from xdet import weights as W
    got = det.forward(W.synthetic_images(1, 480, seed=1))   # still works afterwards
    assert len(got) == 1 and len(got[0]) == 20


@pytest.mark.parametrize('lsep,ksplit', [('direct', 'off'), ('spectral', 'off'), ('direct', 'on'), ('spectral', 'on')])
def test_second_weight_set(lsep, ksplit, oracle):
    """The 1e-3 claim on an independent synthetic model: other seed (777, its own BN calibration), other gains on the
    score layers than the committed recipe (SYNTH_GAINS), other images -- so that the claim does not rest on one
    hand-picked score spread.  Feature maps within 1e-4 of their scale, proposals equal as sets, detections matched
    (a difference only where a per-class NMS score tie explains it)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    gains = {'rpn_head/conv2d_1/kernel': 1.0, 'rpn_head/conv2d_2/kernel': 0.5, 'final_head/fc_cls/kernel': 3.0,
             'final_head/fc_loc/kernel': 1.0}
    w = W.make_lighthead_weights(777, gains=gains)
    imgs = W.synthetic_images(2, 480, seed=555)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(w, image_size=480, max_batch=2, rpn_post_nms_top_n=300, large_sep=lsep, ksplit=ksplit)
    finally:
        set_precision('f32')
    got = det.forward(imgs)
    tr = {}
    ref = oracle.lighthead_forward(imgs, w, rpn_post_nms_top_n=300, trace=tr)
    assert rel_err(det.buffer('feat', 2).numpy(), tr['feat']) < 1e-4
    props = det.flat('proposals', (2, 300, 4))
    for i in range(2):
        a = props[i][np.lexsort(props[i].T)]
        b = tr['proposals'][i][np.lexsort(tr['proposals'][i].T)]
        assert np.abs(a - b).max() < TOL
    total = matched = ties = 0
    stats = {}
    for i in range(2):
        # ksplit='off' (the head GEMM sums K in one left-to-right pass, as the oracle's reference order does): the strict
        # round-3 gate -- a differing list needs a found score-tie pair, at most one list.  ksplit='on' (default; the
        # narrow head GEMM sums four K ranges and folds them): ONE pair of boxes of image 0 whose IoU is 0.3000 +- 0.0003
        # (tools/diag_nms_tie.py prints it) falls on the other side of the per-class NMS threshold in the two classes
        # where both boxes pass the score threshold: two lists, one cause, admitted by the exact-pair rule only.
        t, m, k = assert_match_or_score_tie(got[i], ref[i], allow_threshold_tie=(ksplit == 'on'), stats=stats)
        total, matched, ties = total + t, matched + m, ties + k
    print('seed 777 [%s, ksplit %s]: oracle detections %d matched %d, lists explained by a score tie %s, by a threshold tie %s'
          % (lsep, ksplit, total, matched, stats.get('score_ties', []), stats.get('threshold_ties', [])))
    assert total > 500
    assert len(stats.get('score_ties', [])) <= 1
    assert len(stats.get('threshold_ties', [])) <= (2 if ksplit == 'on' else 0)


def test_cross_fp8_is_opt_in_bounded_and_batch_invariant(oracle, lh_weights):
    """cross='fp8' (NOT the default: profiles/NOTES_r04.md): the 28 depthwise -> pointwise contractions take the cross
    terms of their split-precision products from fp8 copies of the operands, once calibrate() has measured the tensors.
    Measured: those GEMMs 21-25 % faster, the step +5 %, every stage ~10x further from the oracle than f16x3 (features
    1.4e-4 instead of 1.2e-5 of their range) -- which leaves no margin under the 1e-3 bar on the detections (largest
    difference 1.9e-3), so the mode stays off.  What the test pins: nothing changes before the calibration pass, the form
    is on for all 28 edges after it, the stage errors stay inside the measured envelope, and an image's results do not
    depend on the batch it arrives in (the x8 arithmetic is the same in every tile shape)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(2, 480, seed=0)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=2, rpn_post_nms_top_n=300, large_sep='spectral', cross='fp8')
        base = LightHeadDetector(lh_weights, image_size=480, max_batch=2, rpn_post_nms_top_n=300, large_sep='spectral')
    finally:
        set_precision('f32')
    assert det.x8_planes() == 0
    det.forward(imgs)
    base.forward(imgs)
    assert np.array_equal(det.buffer('feat', 2).numpy(), base.buffer('feat', 2).numpy())      # f16x3 until calibrated
    det.calibrate(W.synthetic_images(2, 480, seed=4242))
    assert det.x8_planes() >= 28, det.x8_planes()
    tr = {}
    oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=300, trace=tr)
    det.forward(imgs)
    both = {k: det.buffer(k, 2).numpy().copy() for k in ('mid_x', 'out', 'feat')}
    e_mid = rel_err(np.maximum(both['mid_x'], 0), tr['mid'])
    e_out, e_feat = rel_err(both['out'], tr['out']), rel_err(both['feat'], tr['feat'])
    print('cross=fp8: mid %.2e out %.2e feat %.2e' % (e_mid, e_out, e_feat))
    assert 1e-5 < e_mid < 3e-4 and e_out < 4e-4 and e_feat < 5e-4, (e_mid, e_out, e_feat)
    for i in range(2):                                   # each image alone: the same bits
        det.forward(imgs[i:i + 1])
        for k in both:
            assert np.array_equal(det.buffer(k, 1).numpy()[0], both[k][i]), (k, i)


def test_workspace_reuse_changes_memory_not_results(lh_weights):
    """option "workspace": 'reuse' (default) hands a dead tensor's block to a later tensor of the same size -- the middle
    flow's 24 x 2 tensors live in a handful of blocks -- 'ssa' keeps one block per tensor.  Same detections bit for bit,
    less memory; check_range (whose validation pass reads every tensor after the forward) selects 'ssa' by itself."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(3, 480, seed=77)
    set_precision('f16x3')
    try:
        reuse = LightHeadDetector(lh_weights, image_size=480, max_batch=3, rpn_post_nms_top_n=300)
        ssa = LightHeadDetector(lh_weights, image_size=480, max_batch=3, rpn_post_nms_top_n=300, workspace='ssa')
        checked = LightHeadDetector(lh_weights, image_size=480, max_batch=3, rpn_post_nms_top_n=300, check_range=True)
        # 'poison': reuse, with NaN bits behind every recycled f32 tensor (its loader slack and the rest of the larger block
        # it took over) in front of its producer, on every forward: no consumer may use a byte from behind its tensor
        poison = LightHeadDetector(lh_weights, image_size=480, max_batch=3, rpn_post_nms_top_n=300, workspace='poison')
    finally:
        set_precision('f32')
    a = reuse.forward(imgs, use_graph=True)
    b = ssa.forward(imgs)
    c = checked.forward(imgs)
    d = poison.forward(imgs)
    d2 = poison.forward(imgs, use_graph=True)
    for i in range(3):
        for k in range(1, 21):
            assert np.array_equal(a[i][k][0], b[i][k][0]) and np.array_equal(a[i][k][1], b[i][k][1]), (i, k)
            assert np.array_equal(a[i][k][0], c[i][k][0]) and np.array_equal(a[i][k][1], c[i][k][1]), (i, k)
            assert np.array_equal(d[i][k][0], b[i][k][0]) and np.array_equal(d[i][k][1], b[i][k][1]), ('poison', i, k)
            assert np.array_equal(d2[i][k][0], b[i][k][0]) and np.array_equal(d2[i][k][1], b[i][k][1]), ('poison graph', i, k)
    assert poison.memory()['recycled_bytes'] == reuse.memory()['recycled_bytes']
    for name in ('mid_x', 'out', 'feat'):                 # the named buffers are never recycled
        assert np.array_equal(reuse.buffer(name, 3).numpy(), ssa.buffer(name, 3).numpy()), name
    mr, ms, mc = reuse.memory(), ssa.memory(), checked.memory()
    print('workspace per image: reuse %.3f GB (recycled %.3f GB), ssa %.3f GB'
          % (mr['allocated_bytes'] / 3e9, mr['recycled_bytes'] / 3e9, ms['allocated_bytes'] / 3e9))
    assert ms['recycled_bytes'] == 0 and mc['recycled_bytes'] == 0
    assert mr['recycled_bytes'] > 0 and mr['allocated_bytes'] < 0.85 * ms['allocated_bytes']
    # a calibration measures every planes tensor right behind its producer (its block may be another tensor's by the end
    # of the forward): the exponents it finds are those of the one-block-per-tensor net
    nr = reuse.calibrate(imgs)
    ns = ssa.calibrate(imgs)
    assert nr == ns
    assert reuse.plane_scales() == ssa.plane_scales()


def test_repeated_forwards_below_max_batch(lh_weights):
    """A net built for 6 images runs 2, 2, 5, 1 and 6: the proposal stage's per-image control words (histogram, bad flag,
    the NMS cluster barrier's arrival counter and suppressed bits) are laid out for max_batch and must be cleared for the
    images of THIS call on every forward -- a counter left over from the previous forward lets a cluster's workgroups run
    through their barrier unsynchronised (round 6: seen as one missing detection at 800 x 800).  Every image's detections
    equal those of a max_batch = 1 net, bit for bit, at every cluster size the batches select (16, 16, 16, 16, 16)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(6, 256, seed=606)
    set_precision('f16x3')
    try:
        big = LightHeadDetector(lh_weights, image_size=256, max_batch=6, rpn_post_nms_top_n=300)
        one = LightHeadDetector(lh_weights, image_size=256, max_batch=1, rpn_post_nms_top_n=300)
    finally:
        set_precision('f32')
    ref = [one.forward(imgs[i:i + 1])[0] for i in range(6)]
    for lo, hi, graph in ((0, 2, False), (2, 4, True), (0, 5, False), (3, 4, True), (0, 6, True), (4, 6, False), (5, 6, False)):
        for rep in range(2):
            got = big.forward(imgs[lo:hi], use_graph=graph)
            for i in range(hi - lo):
                for c in range(1, 21):
                    assert np.array_equal(got[i][c][0], ref[lo + i][c][0]) and np.array_equal(got[i][c][1], ref[lo + i][c][1]), (lo, hi, rep, i, c)


@pytest.mark.parametrize('R', [300, 1000])
def test_nms_clusters_under_concurrent_load(lh_weights, R):
    """The proposal NMS of a small batch runs as clusters of 16 workgroups per image that wait for each other inside the
    kernel (proposals.hip nms_cluster_barrier).  Here two net instances run their forwards concurrently on two streams, three
    images each, 25 times by graph replay: each stream's clusters exchange their words while the OTHER stream's convs fill the
    caches and occupy CUs (uneven load, warm L1s: the conditions under which a missing release / acquire shows).  Every replay
    of every image must give the detections of a net that ran the image alone."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector, PipelinedDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(6, 480, seed=1606)
    set_precision('f16x3')
    try:
        pd = PipelinedDetector(lh_weights, ways=2, max_batch=6, image_size=480, rpn_post_nms_top_n=R)
        one = LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=R)
    finally:
        set_precision('f32')
    ref = [one.forward(imgs[i:i + 1])[0] for i in range(6)]
    for rep in range(25):
        got = pd.forward(imgs)
        for i in range(6):
            for c in range(1, 21):
                assert np.array_equal(got[i][c][0], ref[i][c][0]) and np.array_equal(got[i][c][1], ref[i][c][1]), (rep, i, c)


def test_pool_pass_that_writes_the_next_projections_input(lh_weights):
    """option "pool_sub": blocks 2-3's vertical pool pass also writes the raw subsampled planes the next block's projection
    reads and stores relu(sum) for its first separable conv (on, default), or both are made by their own passes / on load
    (off).  The same values either way: detections and feature maps bit for bit, at 480 and at a size with odd pooled maps."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    for size, n in ((480, 3), (256, 2)):
        imgs = W.synthetic_images(n, size, seed=91)
        set_precision('f16x3')
        try:
            on = LightHeadDetector(lh_weights, image_size=size, max_batch=n, rpn_post_nms_top_n=100)
            off = LightHeadDetector(lh_weights, image_size=size, max_batch=n, rpn_post_nms_top_n=100, pool_sub='off')
        finally:
            set_precision('f32')
        a = on.forward(imgs, use_graph=True)
        b = off.forward(imgs)
        for name in ('mid_x', 'out', 'feat'):
            assert np.array_equal(on.buffer(name, n).numpy(), off.buffer(name, n).numpy()), (size, name)
        for i in range(n):
            for k in range(1, 21):
                assert np.array_equal(a[i][k][0], b[i][k][0]) and np.array_equal(a[i][k][1], b[i][k][1]), (size, i, k)
        # the same exponents from a calibration (the pool pass's planes are measured like the subsample pass's were)
        assert on.calibrate(imgs) == off.calibrate(imgs) == {}
