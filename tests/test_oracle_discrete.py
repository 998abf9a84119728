"""A SECOND statement of the oracle's discrete stages (CPU suite).

The oracle's top-k / NMS / proposal / per-class detection code (oracle/lighthead_oracle.py) is "parity unpinned": the
reference holds no golden vectors for them and the TF 1.6 kernels they stand for cannot run here.  What can still be
done is to state the same published semantics twice, independently, and demand exact agreement on tie-heavy inputs:
the functions below were written from the reference's graph code (net/xception_body.py:41-213,402-448,
utility/eval_helper.py:278-361,365-470,556-587, light_head_rfcn_eval.py:263-287) and TensorFlow's documented kernel
behaviour (tf.nn.top_k: "if two elements are equal, the lower-index element appears first";
tf.image.non_max_suppression: greedy over descending score, a box is dropped when its IoU with an already selected box
is > iou_threshold, the IoU of a box with non-positive area is 0, corners may come in either order) as brute-force
O(n^2) matrix code -- not by reading the oracle's implementations.  A disagreement means one of the two statements is
wrong; the GPU kernels are tested against the oracle elsewhere."""
import numpy as np
import pytest

F = np.float32


# ---------------------------------------------------------------------------------------------------------------------
# brute-force restatements
# ---------------------------------------------------------------------------------------------------------------------
def bf_top_k(scores, k):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))[:k]
    return np.asarray(order, np.int64)


def bf_iou_matrix(b):
    """all pairs at once, float32 step by step as TF's NonMaxSuppression IOU() does"""
    b = np.asarray(b, F)
    y0, y1 = np.minimum(b[:, 0], b[:, 2]), np.maximum(b[:, 0], b[:, 2])
    x0, x1 = np.minimum(b[:, 1], b[:, 3]), np.maximum(b[:, 1], b[:, 3])
    area = ((y1 - y0).astype(F) * (x1 - x0).astype(F)).astype(F)
    iy0, iy1 = np.maximum(y0[:, None], y0[None]), np.minimum(y1[:, None], y1[None])
    ix0, ix1 = np.maximum(x0[:, None], x0[None]), np.minimum(x1[:, None], x1[None])
    inter = (np.maximum(iy1 - iy0, F(0)).astype(F) * np.maximum(ix1 - ix0, F(0)).astype(F)).astype(F)
    den = ((area[:, None] + area[None]).astype(F) - inter).astype(F)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = (inter / den).astype(F)
    dead = (area[:, None] <= 0) | (area[None] <= 0)
    return np.where(dead, F(0), iou)


def bf_nms(boxes, scores, max_out, thr):
    m = bf_iou_matrix(boxes)
    keep = []
    for i in bf_top_k(scores, len(scores)):
        if len(keep) >= max_out:
            break
        if all(not (m[i, k] > F(thr)) for k in keep):
            keep.append(int(i))
    return np.asarray(keep, np.int64)


def bf_clip(ref, b):
    b = np.asarray(b, F)
    y0, x0 = np.maximum(b[:, 0], F(ref[0])), np.maximum(b[:, 1], F(ref[1]))
    y1, x1 = np.minimum(b[:, 2], F(ref[2])), np.minimum(b[:, 3], F(ref[3]))
    return np.stack([np.minimum(y0, y1), np.minimum(x0, x1), y1, x1], 1).astype(F)


def bf_size_center_mask(b, min_size):
    ws, hs = (b[:, 3] - b[:, 1]).astype(F), (b[:, 2] - b[:, 0]).astype(F)
    xc, yc = (b[:, 1] + ws / F(2)).astype(F), (b[:, 0] + hs / F(2)).astype(F)
    return (ws > F(min_size)) & (hs > F(min_size)) & (xc > 0) & (yc > 0) & (xc < 1) & (yc < 1)


def bf_pad(x, n):
    x = np.asarray(x, F)
    if len(x) >= n:
        return x
    return np.concatenate([x, np.zeros((n - len(x),) + x.shape[1:], F)])


def bf_get_proposals_single(score, boxes, pre_n, post_n, nms_thr, min_size):
    """get_proposals for one image (net/xception_body.py:402-439): clip to the image, drop small / off-centre boxes,
    top pre_n by score, zero-pad to pre_n, NMS to post_n, zero-pad, then `_upsample_rois` with the identity standing in for
    tf.random_shuffle (the decision the oracle documents)."""
    b = bf_clip((0., 0., 1., 1.), boxes)
    keep = bf_size_center_mask(b, min_size)
    s, b = np.asarray(score, F)[keep], b[keep]
    idx = bf_top_k(s, min(len(s), pre_n))
    s, b = bf_pad(s[idx], pre_n), bf_pad(b[idx], pre_n)
    sel = bf_nms(b, s, post_n, nms_thr)
    s, b = bf_pad(s[sel], post_n), bf_pad(b[sel], post_n)
    live = s > 0
    s, b = s[live], b[live]
    if len(s) < 1:
        s, b = np.array([1.], F), np.array([[0.2, 0.2, 0.8, 0.8]], F)
    n = len(s)
    if n < post_n:
        left = post_n - n
        sel = np.concatenate([np.tile(np.arange(n), left // n + 1), np.arange(n)[:left % n]])
        s, b = s[sel], b[sel]
    return s, b


def bf_bboxes_eval(prob, boxes, image_shape, bbox_img, num_classes, select_thr, nms_thr, nms_topk, net_input):
    """the detection part of bboxes_eval (light_head_rfcn_eval.py:263-287) behind the softmax"""
    out = {}
    min_size = max(F(0.0001), F(F(0.03) * np.sqrt(F(image_shape[0] * image_shape[1]) / F(net_input[0] * net_input[1]))))
    for c in range(1, num_classes):
        s = np.asarray(prob[:, c], F)
        m = (s > F(select_thr)).astype(F)
        s, b = (s * m).astype(F), (np.asarray(boxes, F) * m[:, None]).astype(F)
        b = bf_clip(bbox_img, b)
        keep = bf_size_center_mask(b, min_size)
        s, b = bf_pad(s[keep], 100), bf_pad(b[keep], 100)          # the dict branch of filter_boxes keeps its default
        ref = np.asarray(bbox_img, F)
        b = ((b - np.array([ref[0], ref[1], ref[0], ref[1]], F)) /
             np.array([ref[2] - ref[0], ref[3] - ref[1], ref[2] - ref[0], ref[3] - ref[1]], F)).astype(F)
        idx = bf_top_k(s, min(len(s), 2 * nms_topk))
        s, b = s[idx], b[idx]
        sel = bf_nms(b, s, nms_topk, nms_thr)
        out[c] = (bf_pad(s[sel], nms_topk), bf_pad(b[sel], nms_topk))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# tie-heavy inputs
# ---------------------------------------------------------------------------------------------------------------------
def tie_heavy_boxes(rng, n, grid=24):
    """corners on a coarse grid (many exact duplicates, many IoUs that are exact ratios of small integers -- 0.5, 0.7,
    0.3 occur as exact values), some degenerate (zero area, swapped corners), scores from a handful of values"""
    y0 = rng.integers(0, grid, n)
    x0 = rng.integers(0, grid, n)
    h = rng.integers(0, grid // 2, n)
    w = rng.integers(0, grid // 2, n)
    b = np.stack([y0, x0, y0 + h, x0 + w], 1).astype(F) / F(grid)
    swap = rng.random(n) < 0.05
    b[swap] = b[swap][:, [2, 3, 0, 1]]
    s = rng.choice(np.linspace(0.05, 0.95, 7).astype(F), n).astype(F)
    return b, s


@pytest.mark.parametrize('seed', range(6))
def test_top_k_ties(oracle, seed):
    rng = np.random.default_rng(seed)
    s = rng.choice(np.array([0., 0.1, 0.5, 0.5000001, 0.9], F), 300).astype(F)
    for k in (1, 7, 150, 300):
        vals, idx = oracle.top_k(s, k)
        mine = bf_top_k(s, k)
        assert np.array_equal(np.asarray(idx), mine)
        assert np.array_equal(np.asarray(vals), s[mine])


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('thr', [0.3, 0.5, 0.7])
def test_nms_on_tie_heavy_boxes(oracle, seed, thr):
    rng = np.random.default_rng(100 + seed)
    b, s = tie_heavy_boxes(rng, 400)
    # pairs whose IoU is EXACTLY the threshold must exist, or the strictness of '>' is not exercised: integer corners,
    # intersection 10 x (10 thr) of a 10 x 10 box with a box inside it -> the float32 quotient IS float32(thr)
    exact = []
    for k, t in enumerate((3, 5, 7)):
        off = F(100 * (k + 1))
        exact += [[off, off, off + 10, off + 10], [off, off, off + 10, off + t]]
    b = np.concatenate([b, np.asarray(exact, F)])
    s = np.concatenate([s, rng.choice(np.linspace(0.05, 0.95, 7).astype(F), len(exact)).astype(F)])
    m = bf_iou_matrix(b)
    assert (m == F(thr)).any(), 'no exact-threshold pair'
    for max_out in (5, 50, 400):
        got = oracle.non_max_suppression(b, s, max_out, thr)
        assert np.array_equal(np.asarray(got), bf_nms(b, s, max_out, thr)), (seed, thr, max_out)


def test_iou_pairs(oracle):
    rng = np.random.default_rng(5)
    b, _ = tie_heavy_boxes(rng, 60)
    b = np.concatenate([b, rng.random((60, 4)).astype(F)])
    m = bf_iou_matrix(b)
    for i in range(len(b)):
        for j in range(len(b)):
            assert F(oracle.iou_tf(b, i, j)) == m[i, j], (i, j)


@pytest.mark.parametrize('seed', range(6))
def test_get_proposals_single(oracle, seed):
    """few / many / no survivors, the upsample tail, the fallback box"""
    rng = np.random.default_rng(200 + seed)
    n = [600, 600, 40, 600, 12, 600][seed]
    b, s = tie_heavy_boxes(rng, n)
    b = (b * F(1.3) - F(0.15)).astype(F)                    # some boxes leave the image: clipping + centre test matter
    if seed == 3:
        s[:] = 0.25                                         # every score ties
    if seed == 5:
        b[:, 2:] = b[:, :2]                                 # nothing survives the size filter -> fallback box
    for pre_n, post_n in ((100, 30), (5000, 300), (50, 1000)):
        gs, gb = oracle.get_proposals_single(s, b, pre_n, post_n, 0.7, 16. / 480)
        ms, mb = bf_get_proposals_single(s, b, pre_n, post_n, 0.7, 16. / 480)
        assert gs.shape == (post_n,) and gb.shape == (post_n, 4)
        assert np.array_equal(gs, ms) and np.array_equal(gb, mb), (seed, pre_n, post_n)


@pytest.mark.parametrize('seed', range(4))
def test_bboxes_eval_discrete_part(oracle, seed):
    """select -> clip -> filter -> resize -> sort -> per-class NMS behind the oracle's own softmax: tie-heavy boxes,
    logits from a small set of values (equal probabilities across ROIs), a non-square original image and a bbox_img
    that is not the unit box"""
    rng = np.random.default_rng(300 + seed)
    R, C = 300, 21
    boxes, _ = tie_heavy_boxes(rng, R, grid=20)
    logits = rng.choice(np.array([-2., 0., 1., 3., 5.], F), (R, C)).astype(F)
    image_shape = [(480, 480), (375, 500), (500, 333), (20, 30)][seed]
    bbox_img = [(0., 0., 1., 1.), (0.1, 0.05, 0.9, 1.0), (0., 0., 1., 1.), (0.2, 0.2, 0.7, 0.8)][seed]
    got = oracle.bboxes_eval(logits, boxes, image_shape=image_shape, bbox_img=bbox_img, num_classes=C,
                             select_threshold=0.01, nms_threshold=0.3, nms_topk=20, net_input=(480, 480))
    prob = oracle.softmax(logits)
    mine = bf_bboxes_eval(prob, boxes, image_shape, bbox_img, C, 0.01, 0.3, 20, (480, 480))
    n_det = 0
    for c in range(1, C):
        gs, gb = got[c]
        ms, mb = mine[c]
        assert np.array_equal(np.asarray(gs, F), ms), c
        assert np.array_equal(np.asarray(gb, F), mb), c
        n_det += int((ms > 0).sum())
    assert n_det > 20
