"""BASELINE's full bench size (64 images per net instance, 300 proposals, f16x3, hipGraph replay) checked
through size-independent properties: the oracle needs ~1.5 s per image, so only sampled images are compared
with it directly; everything else is asserted through invariants of the path."""
import numpy as np
import pytest

from test_gpu_e2e import match_detections, assert_match_or_score_tie

pytestmark = pytest.mark.gpu

B = 64
PROBE = 37           # position of the probe image inside the big batch


@pytest.fixture(scope='module')
def big(lh_weights):
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(B, 480, seed=4242)
    set_precision('f16x3')
    try:
        # max_batch 64 selects the spectral large-separable convs ('auto'); the arithmetic form is a property
        # of the NET, so the single-image detector is built with the same form -- then results are bitwise
        # independent of the batch (a 'direct' net agrees with a 'spectral' one to ~1e-6, not bitwise)
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=B, rpn_post_nms_top_n=300)
        one = LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=300, large_sep='spectral')
    finally:
        set_precision('f32')
    det.set_images(imgs)
    det.forward_device(B, use_graph=True)
    s, b = det.detections(B)
    return det, one, imgs, s.copy(), b.copy()


def iou_matrix(bx):
    """bx [k,4] (ymin,xmin,ymax,xmax)"""
    y0 = np.maximum(bx[:, None, 0], bx[None, :, 0]); x0 = np.maximum(bx[:, None, 1], bx[None, :, 1])
    y1 = np.minimum(bx[:, None, 2], bx[None, :, 2]); x1 = np.minimum(bx[:, None, 3], bx[None, :, 3])
    inter = np.clip(y1 - y0, 0, None) * np.clip(x1 - x0, 0, None)
    area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
    return inter / np.maximum(area[:, None] + area[None, :] - inter, 1e-12)


def test_output_invariants(big):
    """bboxes_eval contract (light_head_rfcn_eval.py:263-287) on every one of the 64 x 20 class lists:
    scores above select_threshold, sorted, zero padded; boxes clipped to the image; survivors of the
    per-class NMS pairwise below the IoU threshold (NMS is idempotent on its own output)."""
    _, _, _, s, b = big
    assert s.shape == (B, 20, 200) and b.shape == (B, 20, 200, 4)
    assert np.isfinite(s).all() and np.isfinite(b).all()
    k = (s > 0).sum(-1)
    assert k.sum() > 5000
    worst = 0.0
    for i in range(B):
        for c in range(20):
            n = int(k[i, c])
            assert np.all(s[i, c, n:] == 0) and np.all(b[i, c, n:] == 0)
            if n == 0:
                continue
            sc, bx = s[i, c, :n], b[i, c, :n]
            assert np.all(sc > 0.01) and np.all(sc <= 1.0)
            assert np.all(np.diff(sc) <= 0)
            assert bx.min() >= 0.0 and bx.max() <= 1.0
            assert np.all(bx[:, 2] >= bx[:, 0]) and np.all(bx[:, 3] >= bx[:, 1])
            if n > 1:
                m = iou_matrix(bx.astype(np.float64))
                np.fill_diagonal(m, 0)
                worst = max(worst, float(m.max()))
    assert worst <= 0.3 + 1e-6, worst


def test_replay_and_eager_agree_at_full_size(big):
    det, _, _, s, b = big
    det.forward_device(B, use_graph=True)
    s2, b2 = det.detections(B)
    det.forward_device(B, use_graph=False)
    s3, b3 = det.detections(B)
    assert np.array_equal(s, s2) and np.array_equal(b, b2)
    assert np.array_equal(s, s3) and np.array_equal(b, b3)


def test_batch_invariance(big):
    """Images are independent units (what the multi-GPU sharding relies on): an image's detections do not
    depend on its batch, its position in the batch, or the tile shapes the batch size selects (256x256
    tiles at 64 images, 128x128 at 1) -- bit for bit."""
    det, one, imgs, s, b = big
    for pos in (0, PROBE, B - 1):
        got = one.forward(imgs[pos:pos + 1])
        for c in range(20):
            gs, gb = got[0][c + 1]
            assert np.array_equal(gs, s[pos, c]), (pos, c)
            assert np.array_equal(gb, b[pos, c]), (pos, c)
    # moving an image inside the batch moves its result with it
    perm = np.roll(np.arange(B), 5)
    det.set_images(imgs[perm])
    det.forward_device(B, use_graph=True)
    s2, b2 = det.detections(B)
    assert np.array_equal(s2, s[perm]) and np.array_equal(b2, b[perm])
    det.set_images(imgs)


def test_sampled_images_against_the_oracle(big, oracle, lh_weights):
    """two images of the full-size batch through the CPU oracle: every detection within 1e-3"""
    _, _, imgs, s, b = big
    total = matched = extra = 0
    for pos in (PROBE, 3):
        ref = oracle.lighthead_forward(imgs[pos:pos + 1], lh_weights, rpn_post_nms_top_n=300)
        got = {c + 1: (s[pos, c], b[pos, c]) for c in range(20)}
        t, m, e = match_detections(got, ref[0])
        total, matched, extra = total + t, matched + m, extra + e
    print('full-size batch, 2 sampled images: oracle %d matched %d extra %d' % (total, matched, extra))
    assert total > 100
    assert matched == total and extra == 0, (matched, total, extra)


def test_pipelined_detector_equals_single(big, lh_weights):
    """the 2-way concurrent front end (bench default) returns exactly the single detector's detections,
    also for a ragged batch (the last sub-batch short, one empty)"""
    from xdet.model import PipelinedDetector
    from xdet.runtime import set_precision
    _, _, imgs, s, b = big
    set_precision('f16x3')
    try:
        pd = PipelinedDetector(lh_weights, ways=2, max_batch=B, image_size=480, rpn_post_nms_top_n=300)
    finally:
        set_precision('f32')
    for n in (B, 41, 7):
        got = pd.forward(imgs[:n])
        assert len(got) == n
        for i in (0, n // 2, n - 1):
            for c in range(20):
                assert np.array_equal(got[i][c + 1][0], s[i, c]) and np.array_equal(got[i][c + 1][1], b[i, c])


@pytest.mark.parametrize('lsep', ['direct', 'spectral'])
def test_more_images_against_the_oracle(oracle, lh_weights, lsep):
    """a wider sample for the 1e-3 claim: 6 more images (another seed) through the default product
    arithmetic (f16x3, both forms of the large-separable convs), every oracle detection matched by a
    distinct GPU detection and vice versa"""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(6, 480, seed=20260928)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=6, rpn_post_nms_top_n=300, large_sep=lsep)
    finally:
        set_precision('f32')
    got = det.forward(imgs, use_graph=True)
    ref = oracle.lighthead_forward(imgs, lh_weights, rpn_post_nms_top_n=300)
    total = matched = ties = 0
    stats = {}
    for i in range(6):
        t, m, k = assert_match_or_score_tie(got[i], ref[i], stats=stats)       # score ties only (threshold ties: not admitted)
        total, matched, ties = total + t, matched + m, ties + k
    print('6 images, seed 20260928 [%s]: oracle %d matched %d, class lists explained by an NMS score tie: %s'
          % (lsep, total, matched, stats.get('score_ties', [])))
    assert total > 1000
    assert ties <= 1 and matched >= total - 4 * ties, (matched, total, ties)


@pytest.mark.parametrize('nb', [1, 3, 8, 17, 32])
def test_batch_invariance_across_batch_sizes(big, lh_weights, nb):
    """every batch size selects its own mix of tile shapes (128x64 ... 256x256) per layer: the first nb
    images computed as a batch of nb equal the same images inside the batch of 64, bit for bit"""
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    _, _, imgs, s, b = big
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=nb, rpn_post_nms_top_n=300, large_sep='spectral')
    finally:
        set_precision('f32')
    det.set_images(imgs[:nb])
    det.forward_device(nb, use_graph=True)
    s2, b2 = det.detections(nb)
    assert np.array_equal(s2, s[:nb]) and np.array_equal(b2, b[:nb])


def test_batch_invariance_at_1000_proposals(lh_weights):
    """The reference's operating point (rpn_post_nms_top_n = 1000, light_head_rfcn_eval.py:109-111): the proposal NMS of a
    batch of 1 / 3 / 12 / 20 / 40 / 70 images runs as clusters of 16 / 16 / 8 / 4 / 2 / 1 workgroups per image
    (proposals.hip nms_cluster_size), PsRoiAlign deals fewer than 8 images over several XCDs each, the frequency-bin GEMMs of
    1-2 images use their own tiles: an image's detections are the same bits in every one of these batches."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    NB = 70
    imgs = W.synthetic_images(8, 480, seed=1000)
    imgs = np.concatenate([imgs] * 9)[:NB]
    imgs[8:] += np.linspace(0.0, 0.05, NB - 8, dtype=np.float32)[:, None, None, None]     # no two images alike
    set_precision('f16x3')
    try:
        big = LightHeadDetector(lh_weights, image_size=480, max_batch=NB, rpn_post_nms_top_n=1000)
        one = LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=1000)
    finally:
        set_precision('f32')
    big.set_images(imgs)
    big.forward_device(NB, use_graph=True)
    s, b = big.detections(NB)
    s, b = s.copy(), b.copy()
    assert (s > 0).any(axis=(1, 2)).all()
    for nb in (1, 3, 12, 20, 40):
        big.set_images(imgs[:nb])
        big.forward_device(nb, use_graph=True)
        s2, b2 = big.detections(nb)
        assert np.array_equal(s2, s[:nb]) and np.array_equal(b2, b[:nb]), nb
    for pos in (0, 9, NB - 1):
        got = one.forward(imgs[pos:pos + 1])
        for c in range(20):
            assert np.array_equal(got[0][c + 1][0], s[pos, c]) and np.array_equal(got[0][c + 1][1], b[pos, c]), (pos, c)


def test_the_configuration_the_driver_benches(oracle, lh_weights):
    """BENCH_rNN's configuration, asserted instead of only timed (BASELINE config 4 at world size 1): 2 concurrent
    sub-batches x 128 images, f16x3 + spectral, hipGraph replay, input = raw uint8 VOC-shape images through the F1
    kernel into the nets' input buffers, detections written into one shared pair of buffers per parity, packed and
    all-gathered by the RCCL communicator with the double-buffered event protocol, no host sync between steps.
    Checked: the gathered records of >= 3 positions per sub-batch equal single-image nets bit for bit, and two sampled
    images agree with the oracle (own F1 restatement included) within 1e-3."""
    from xdet import dist as xd
    from xdet._lib import lib, check
    from xdet.model import LightHeadDetector
    from xdet.runtime import DeviceBuffer, set_precision, to_device
    SB, WAYS, S = 128, 2, 480
    Bt = SB * WAYS
    nc, k = 20, 200
    shapes = [(375, 500), (500, 375), (333, 500), (500, 333)]
    rng = np.random.default_rng(77)
    probes = {0: (0, 61, 127), 1: (0, 5, 127)}              # sub-batch -> positions compared with single-image nets
    # every image distinct at the probed positions, the rest cycle through a small pool (as bench.py --voc-stream does)
    pool = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes for _ in range(2)]
    raw = {}
    for i in range(WAYS):
        for j in range(SB):
            raw[(i, j)] = pool[(i * 3 + j) % len(pool)]
        for j in probes[i]:
            h, w = shapes[(i + j) % 4]
            raw[(i, j)] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    dev = {id(a): to_device(a) for a in {id(a): a for a in raw.values()}.values()}
    set_precision('f16x3')
    try:
        nets = [LightHeadDetector(lh_weights, image_size=S, max_batch=SB, rpn_post_nms_top_n=300) for _ in range(WAYS)]
        one = LightHeadDetector(lh_weights, image_size=S, max_batch=1, rpn_post_nms_top_n=300, large_sep='spectral')
    finally:
        set_precision('f32')
    comm = xd.Communicator(0, 1)
    det = [(DeviceBuffer(Bt * nc * k * 4, zero=True), DeviceBuffer(Bt * nc * k * 16, zero=True)) for _ in range(2)]
    for step in range(3):                                    # both buffer pairs get used; nothing syncs in between
        for i, nt in enumerate(nets):
            for j in range(SB):
                a = raw[(i, j)]
                check(lib().xdet_preprocess_eval(dev[id(a)].ptr, a.shape[0], a.shape[1], nt._images.ptr + j * 3 * S * S * 4, S,
                                                 nt.stream.handle))
        ds, db = det[step & 1]
        for i, nt in enumerate(nets):
            nt.forward_device(SB, use_graph=True, det_scores_ptr=ds.ptr + i * SB * nc * k * 4,
                              det_boxes_ptr=db.ptr + i * SB * nc * k * 16)
        comm.allgather_detections(ds.ptr, db.ptr, Bt, nc, k, streams=[nt.stream for nt in nets], double_buffered=True)
    g = comm.gathered()
    assert g.shape == (Bt, nc, k, 5)
    gs, gb = xd.unpack_detections(g)
    assert np.isfinite(gs).all() and ((gs > 0).reshape(Bt, -1).any(1)).all()     # every image produced detections
    from xdet.ops import light_head_preprocess_for_test
    total = matched = extra = 0
    for i in range(WAYS):
        for n_, j in enumerate(probes[i]):
            x = light_head_preprocess_for_test(raw[(i, j)], (S, S), 'NCHW')[None]
            got = one.forward(x, use_graph=True)
            pos = i * SB + j
            for c in range(nc):
                assert np.array_equal(got[0][c + 1][0], gs[pos, c]), (i, j, c)
                assert np.array_equal(got[0][c + 1][1], gb[pos, c]), (i, j, c)
            if n_ == 1:                                      # one image per sub-batch through the oracle (own F1 too)
                xo = oracle.preprocess_for_eval(raw[(i, j)], S)[None]
                ref = oracle.lighthead_forward(xo, lh_weights, rpn_post_nms_top_n=300)
                t, m, e = match_detections({c + 1: (gs[pos, c], gb[pos, c]) for c in range(nc)}, ref[0])
                total, matched, extra = total + t, matched + m, extra + e
    print('driver configuration (2 x 128, VOC stream, RCCL world 1): 6 positions bitwise, oracle %d matched %d extra %d'
          % (total, matched, extra))
    assert total > 50 and matched == total and extra == 0, (total, matched, extra)
    comm.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 5's shape (800 x 800 -> 50 x 50 map, 55,000 anchors) at a bench-like batch, through the same
# size-independent properties (the 1e-3 comparison with the oracle at this shape is tests/test_gpu_e2e.py::
# test_other_baseline_configs, one image; here the batch is what is under test)
# ---------------------------------------------------------------------------------------------------------------------
B8 = 32


@pytest.fixture(scope='module')
def big800(lh_weights):
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    imgs = W.synthetic_images(B8, 800, seed=800800)
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=800, max_batch=B8, rpn_post_nms_top_n=300)
    finally:
        set_precision('f32')
    det.set_images(imgs)
    det.forward_device(B8, use_graph=True)
    s, b = det.detections(B8)
    return det, imgs, s.copy(), b.copy()


def test_800_output_invariants(big800):
    _, _, s, b = big800
    assert s.shape == (B8, 20, 200) and b.shape == (B8, 20, 200, 4)
    assert np.isfinite(s).all() and np.isfinite(b).all()
    k = (s > 0).sum(-1)
    assert k.sum() > 1000, int(k.sum())
    worst = 0.0
    for i in range(B8):
        for c in range(20):
            n = int(k[i, c])
            assert np.all(s[i, c, n:] == 0) and np.all(b[i, c, n:] == 0)
            if n == 0:
                continue
            sc, bx = s[i, c, :n], b[i, c, :n]
            assert np.all(sc > 0.01) and np.all(sc <= 1.0)
            assert np.all(np.diff(sc) <= 0)
            assert bx.min() >= 0.0 and bx.max() <= 1.0
            assert np.all(bx[:, 2] >= bx[:, 0]) and np.all(bx[:, 3] >= bx[:, 1])
            if n > 1:
                m = iou_matrix(bx.astype(np.float64))
                np.fill_diagonal(m, 0)
                worst = max(worst, float(m.max()))
    assert worst <= 0.3 + 1e-6, worst


def test_800_replay_eager_and_batch_invariance(big800, lh_weights):
    """graph replay == eager; an image's detections do not depend on the batch it arrives in (32 -> 1 and 5), its
    position, or the tile shapes / launch splits the batch size selects (the 397^2 x 128 tensor of 32 images is cut into
    image ranges below 2 GiB by two kernels) -- bit for bit"""
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    det, imgs, s, b = big800
    det.forward_device(B8, use_graph=True)
    s2, b2 = det.detections(B8)
    det.forward_device(B8, use_graph=False)
    s3, b3 = det.detections(B8)
    assert np.array_equal(s, s2) and np.array_equal(b, b2)
    assert np.array_equal(s, s3) and np.array_equal(b, b3)
    set_precision('f16x3')
    try:
        # the arithmetic form of the large-separable convs is a property of the net: the same form as the big one
        small = LightHeadDetector(lh_weights, image_size=800, max_batch=5, rpn_post_nms_top_n=300, large_sep='spectral')
    finally:
        set_precision('f32')
    for pos in (0, 13, B8 - 1):
        got = small.forward(imgs[pos:pos + 1])
        for c in range(20):
            assert np.array_equal(got[0][c + 1][0], s[pos, c]), (pos, c)
            assert np.array_equal(got[0][c + 1][1], b[pos, c]), (pos, c)
    got = small.forward(imgs[20:25])
    for i in range(5):
        for c in range(20):
            assert np.array_equal(got[i][c + 1][0], s[20 + i, c]) and np.array_equal(got[i][c + 1][1], b[20 + i, c])
    perm = np.roll(np.arange(B8), 3)
    det.set_images(imgs[perm])
    det.forward_device(B8, use_graph=True)
    s4, b4 = det.detections(B8)
    assert np.array_equal(s4, s[perm]) and np.array_equal(b4, b[perm])
    det.set_images(imgs)


def test_800_sampled_image_against_the_oracle(big800, oracle, lh_weights):
    """one image of the 800 x 800 batch through the CPU oracle: every detection within 1e-3"""
    _, imgs, s, b = big800
    pos = 13
    ref = oracle.lighthead_forward(imgs[pos:pos + 1], lh_weights, rpn_post_nms_top_n=300)
    got = {c + 1: (s[pos, c], b[pos, c]) for c in range(20)}
    total, matched, ties = assert_match_or_score_tie(got, ref[0])
    print('800 x 800 batch of %d, image %d: oracle %d matched %d (lists explained by a score tie: %d)' % (B8, pos, total, matched, ties))
    assert total > 20 and ties <= 1
