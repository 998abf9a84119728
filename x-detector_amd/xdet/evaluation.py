"""F4: eval bookkeeping behind the detector -- host-side NumPy, same names and
semantics as the reference so a detections dict from `LightHeadDetector.forward` can be scored
exactly the way `bboxes_eval` continues (light_head_rfcn_eval.py:288-338):

  bboxes_jaccard / bboxes_matching     utility/eval_helper.py:671-781
  StreamingTpFp (streaming_tp_fp_arrays), precision_recall,
  average_precision_voc07 / _voc12      utility/metrics.py:102-261
  VOC_LABELS                            dataset/dataset_common.py:27-55
"""
import numpy as np

VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
               'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')
_CATEGORY = dict(aeroplane='Vehicle', bicycle='Vehicle', bird='Animal', boat='Vehicle', bottle='Indoor', bus='Vehicle',
                 car='Vehicle', cat='Animal', chair='Indoor', cow='Animal', diningtable='Indoor', dog='Animal',
                 horse='Animal', motorbike='Vehicle', person='Person', pottedplant='Indoor', sheep='Animal',
                 sofa='Indoor', train='Vehicle', tvmonitor='Indoor')
VOC_LABELS = {'none': (0, 'Background')}
VOC_LABELS.update({name: (i + 1, _CATEGORY[name]) for i, name in enumerate(VOC_CLASSES)})


def label2name_table():
    """light_head_rfcn_eval.py:169-174."""
    return {pair[0]: name for name, pair in VOC_LABELS.items()}


def bboxes_jaccard(bbox_ref, bboxes):
    """IoU of one reference box against N boxes, 0 where the union is not positive (safe_divide)."""
    b = np.asarray(bboxes, np.float32).reshape(-1, 4)
    r = np.asarray(bbox_ref, np.float32).reshape(4)
    h = np.maximum(np.minimum(b[:, 2], r[2]) - np.maximum(b[:, 0], r[0]), 0.)
    w = np.maximum(np.minimum(b[:, 3], r[3]) - np.maximum(b[:, 1], r[1]), 0.)
    inter = h * w
    union = -inter + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) + (r[2] - r[0]) * (r[3] - r[1])
    out = np.zeros_like(inter)
    np.divide(inter, union, out=out, where=union > 0)
    return out


def bboxes_matching(label, scores, bboxes, glabels, gbboxes, gdifficults, matching_threshold=0.5):
    """Greedy Pascal-VOC matching of one class's detections (already sorted by score) against the
    ground truth: -> (n_gbboxes, tp[N], fp[N]).  A detection whose best-IoU ground truth is
    `difficult` is neither TP nor FP."""
    scores = np.asarray(scores)
    bboxes = np.asarray(bboxes, np.float32).reshape(-1, 4)
    glabels = np.asarray(glabels).reshape(-1)
    gbboxes = np.asarray(gbboxes, np.float32).reshape(-1, 4)
    gdiff = np.asarray(gdifficults).reshape(-1).astype(bool)
    same = glabels == label
    n_gbboxes = int(np.count_nonzero(same & ~gdiff))
    n = scores.shape[0]
    tp = np.zeros(n, bool)
    fp = np.zeros(n, bool)
    gmatch = np.zeros(glabels.shape[0], bool)
    if glabels.shape[0] == 0:
        return n_gbboxes, tp, np.ones(n, bool) & False
    for i in range(n):
        jac = bboxes_jaccard(bboxes[i], gbboxes) * same
        k = int(np.argmax(jac))
        match = jac[k] > matching_threshold
        if gdiff[k]:
            continue
        tp[i] = match and not gmatch[k]
        fp[i] = gmatch[k] or not match
        if match:
            gmatch[k] = True
    return n_gbboxes, tp, fp


class StreamingTpFp(object):
    """streaming_tp_fp_arrays: accumulates, per class, the ground-truth count and the
    (score, tp, fp) triplets of every detection that is TP or FP (score > 1e-4)."""
    def __init__(self, remove_zero_scores=True):
        self.remove_zero_scores = remove_zero_scores
        self.nobjects = {}
        self.scores, self.tp, self.fp = {}, {}, {}

    def update(self, c, num_gbboxes, tp, fp, scores):
        tp = np.asarray(tp, bool).reshape(-1)
        fp = np.asarray(fp, bool).reshape(-1)
        scores = np.asarray(scores, np.float32).reshape(-1)
        mask = tp | fp
        if self.remove_zero_scores:
            mask &= scores > 1e-4
            tp, fp, scores = tp[mask], fp[mask], scores[mask]
        self.nobjects[c] = self.nobjects.get(c, 0) + int(num_gbboxes)
        for store, v in ((self.scores, scores), (self.tp, tp), (self.fp, fp)):
            store[c] = np.concatenate([store[c], v]) if c in store else v

    def update_image(self, detections, glabels, gbboxes, gdifficults, matching_threshold=0.5):
        """detections: {class: (scores[K], boxes[K,4])} as returned by the detector for one image."""
        for c, (s, b) in detections.items():
            n, tp, fp = bboxes_matching(c, s, b, glabels, gbboxes, gdifficults, matching_threshold)
            self.update(c, n, tp, fp, s)

    def average_precisions(self):
        out07, out12 = {}, {}
        for c in self.nobjects:
            p, r = precision_recall(self.nobjects[c], self.tp[c], self.fp[c], self.scores[c])
            out07[c], out12[c] = average_precision_voc07(p, r), average_precision_voc12(p, r)
        return out07, out12


def precision_recall(num_gbboxes, tp, fp, scores):
    order = np.argsort(-np.asarray(scores, np.float32), kind='stable')
    ctp = np.cumsum(np.asarray(tp, bool)[order].astype(np.float64))
    cfp = np.cumsum(np.asarray(fp, bool)[order].astype(np.float64))
    recall = ctp / num_gbboxes if num_gbboxes > 0 else np.zeros_like(ctp)
    den = ctp + cfp
    precision = np.divide(ctp, den, out=np.zeros_like(ctp), where=den > 0)
    return precision, recall


def average_precision_voc12(precision, recall):
    p = np.concatenate([[0.], np.asarray(precision, np.float64), [0.]])
    r = np.concatenate([[0.], np.asarray(recall, np.float64), [1.]])
    p = np.maximum.accumulate(p[::-1])[::-1]
    return float(np.sum(p[1:] * (r[1:] - r[:-1])))


def average_precision_voc07(precision, recall):
    p = np.concatenate([np.asarray(precision, np.float64), [0.]])
    r = np.concatenate([np.asarray(recall, np.float64), [np.inf]])
    return float(sum(p[r >= t].max() / 11. for t in np.arange(0., 1.1, 0.1)))


# ---- drawing (utility/draw_toolbox.py:72-104) -------------------------------------------------
# Tableau-20 palette, white for background -- the colour table the reference indexes by class id
COLORS_TABLEAU = [(255, 255, 255), (31, 119, 180), (174, 199, 232), (255, 127, 14), (255, 187, 120), (44, 160, 44),
                  (152, 223, 138), (214, 39, 40), (255, 152, 150), (148, 103, 189), (197, 176, 213), (140, 86, 75),
                  (196, 156, 148), (227, 119, 194), (247, 182, 210), (127, 127, 127), (199, 199, 199),
                  (188, 189, 34), (219, 219, 141), (23, 190, 207), (158, 218, 229)]


def bboxes_draw_on_img(img, classes, scores, bboxes, thickness=2):
    """Same contract as the reference's bboxes_draw_on_img: img uint8 [H,W,3] (modified in place and
    returned), bboxes (ymin,xmin,ymax,xmax) in [0,1]; class 0 and boxes thinner than one pixel are
    skipped; corners are `int(coord * shape)`.  Rectangles are rasterised with NumPy (no OpenCV here;
    the outline is `thickness` pixels wide, centred on the box edge like cv2.rectangle); the
    'name/score%' caption is drawn with Pillow when it is importable and silently omitted otherwise."""
    h, w = img.shape[:2]
    names = label2name_table()
    try:
        from PIL import Image, ImageDraw
    except Exception:
        Image = ImageDraw = None
    captions = []
    for i in range(bboxes.shape[0]):
        c = int(classes[i])
        if c < 1:
            continue
        y0, x0 = int(bboxes[i][0] * h), int(bboxes[i][1] * w)
        y1, x1 = int(bboxes[i][2] * h), int(bboxes[i][3] * w)
        if y1 - y0 < 1 or x1 - x0 < 1:
            continue
        color = np.asarray(COLORS_TABLEAU[c % len(COLORS_TABLEAU)], img.dtype)
        lo, hi = thickness // 2, (thickness + 1) // 2

        def span(a, n):
            return slice(max(a - lo, 0), min(a + hi, n))
        ys, xs = slice(max(y0 - lo, 0), min(y1 + hi, h)), slice(max(x0 - lo, 0), min(x1 + hi, w))
        img[span(y0, h), xs] = color
        img[span(y1, h), xs] = color
        img[ys, span(x0, w)] = color
        img[ys, span(x1, w)] = color
        captions.append((x0, max(y0 - 12, 0), '%s/%.1f%%' % (names.get(c, str(c)), float(scores[i]) * 100), tuple(int(v) for v in color)))
    if ImageDraw is not None and captions:
        pil = Image.fromarray(img)
        d = ImageDraw.Draw(pil)
        for x, y, s, col in captions:
            box = d.textbbox((x, y), s)
            d.rectangle(box, fill=col)
            d.text((x, y), s, fill=(255, 255, 255))
        img[...] = np.asarray(pil)
    return img
