"""oracle/lighthead_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy fp32 CPU restatement of the reference's Light-Head R-CNN *eval* forward path
(`lighr_head_model_fn`, light_head_rfcn_eval.py:364-433 + `bboxes_eval` :263-287).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

PARITY PINNING
  * PsRoiAlign is pinned: oracle/psroialign_ref.c (and the NumPy
    twin `ps_roi_align_np` below) reproduce the known-answer vectors SURVEY.md 8c records
    for the reference's own test inputs (cpp/PSROIPooling/test_op.py:52-81).
  * Everything else is **parity unpinned**: the reference holds no golden tensors, no
    checkpoint, and its graph is TF 1.6 Python that cannot be imported here.  TF kernel
    semantics that live outside /root/reference (TensorFlow 1.6, README.md:16) are
    restated from TF's published behaviour and fixed by decision:
      - SAME padding: out=ceil(n/s), pad_total=max((out-1)s+k_eff-n,0), extra pixel at
        the bottom/right; max-pool ignores padded cells.
      - tf.nn.top_k: descending, ties -> lower index first.
      - tf.image.non_max_suppression (NonMaxSuppressionV2): greedy over descending score
        (stable for ties), suppress when IoU > thr (strict), IoU = 0 when either area <= 0,
        corner order normalised with min/max.
      - tf.random_shuffle in _upsample_rois (xception_body.py:208) is replaced by the
        identity permutation (the tail only ever duplicates kept ROIs).
    The dense arithmetic (conv / BN / pool / dense) is cross-checked against torch-CPU in
    tests/test_oracle_layers.py as an independent implementation.

Layout: activations NHWC inside the oracle; public entry points take the reference's
default NCHW image (`data_format='channels_first'`, light_head_rfcn_eval.py:85).
Boxes: (ymin, xmin, ymax, xmax) normalised; ROIs to the op: (cy, cx, h, w).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------
# dense layers (restating tf.layers.* semantics used by net/xception_body.py, net/resnet_v2.py)
# ----------------------------------------------------------------------------------------

def same_pad(n, k, s, d=1):
    k_eff = (k - 1) * d + 1
    out = -(-n // s)
    total = max((out - 1) * s + k_eff - n, 0)
    return total // 2, total - total // 2, out


def conv2d(x, w, stride=1, padding='SAME', dilation=1, bias=None):
    """x [N,H,W,Cin] f32, w [kh,kw,Cin,Cout] (HWIO). im2col + sgemm."""
    x = np.asarray(x, F32)
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    if padding == 'SAME':
        pt, pb, Ho = same_pad(H, kh, stride, dilation)
        pl, pr, Wo = same_pad(W, kw, stride, dilation)
    elif padding == 'VALID':
        pt = pb = pl = pr = 0
        Ho = (H - ((kh - 1) * dilation + 1)) // stride + 1
        Wo = (W - ((kw - 1) * dilation + 1)) // stride + 1
    else:  # explicit ((pt,pb),(pl,pr)) then VALID  (resnet_v2.fixed_padding :62-86)
        (pt, pb), (pl, pr) = padding
        Ho = (H + pt + pb - ((kh - 1) * dilation + 1)) // stride + 1
        Wo = (W + pl + pr - ((kw - 1) * dilation + 1)) // stride + 1
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))) if (pt or pb or pl or pr) else x
    out = np.empty((N, Ho, Wo, Cout), F32)
    wm = np.ascontiguousarray(w.reshape(kh * kw * Cin, Cout), F32)
    for n in range(N):
        if kh == 1 and kw == 1:
            cols = xp[n, 0:(Ho - 1) * stride + 1:stride, 0:(Wo - 1) * stride + 1:stride, :].reshape(Ho * Wo, Cin)
        else:
            cols = np.empty((Ho, Wo, kh * kw, Cin), F32)
            for i in range(kh):
                for j in range(kw):
                    y0, x0 = i * dilation, j * dilation
                    cols[:, :, i * kw + j, :] = xp[n, y0:y0 + (Ho - 1) * stride + 1:stride,
                                                   x0:x0 + (Wo - 1) * stride + 1:stride, :]
            cols = cols.reshape(Ho * Wo, kh * kw * Cin)
        out[n] = (cols @ wm).reshape(Ho, Wo, Cout)
    if bias is not None:
        out += bias.astype(F32)
    return out


def depthwise_conv2d(x, w, dilation=1):
    """3x3 depthwise, stride 1, SAME, multiplier 1.  w [3,3,C,1]."""
    N, H, W, C = x.shape
    kh, kw = w.shape[0], w.shape[1]
    pt, pb, _ = same_pad(H, kh, 1, dilation)
    pl, pr, _ = same_pad(W, kw, 1, dilation)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((N, H, W, C), F32)
    for i in range(kh):
        for j in range(kw):
            out += xp[:, i * dilation:i * dilation + H, j * dilation:j * dilation + W, :] * w[i, j, :, 0].astype(F32)
    return out


def separable_conv2d(x, dw, pw, dilation=1):
    """tf.layers.separable_conv2d: depthwise then pointwise, nothing in between
    (net/xception_body.py:224-231)."""
    return conv2d(depthwise_conv2d(x, dw, dilation), pw, 1, 'SAME')


def batch_norm(x, w, name, eps):
    """Inference BN: gamma*(x-mean)/sqrt(var+eps)+beta (net/xception_body.py:20-22,232)."""
    scale = (w[name + '/gamma'] / np.sqrt(w[name + '/moving_variance'] + F32(eps))).astype(F32)
    shift = (w[name + '/beta'] - w[name + '/moving_mean'] * scale).astype(F32)
    return (x * scale + shift).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def max_pool_3x3_s2_same(x):
    N, H, W, C = x.shape
    pt, pb, Ho = same_pad(H, 3, 2)
    pl, pr, Wo = same_pad(W, 3, 2)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    out = np.full((N, Ho, Wo, C), -np.inf, F32)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, xp[:, i:i + (Ho - 1) * 2 + 1:2, j:j + (Wo - 1) * 2 + 1:2, :])
    return out


def dense(x, w, name, act_relu=False):
    y = (x.astype(F32) @ w[name + '/kernel'].astype(F32) + w[name + '/bias'].astype(F32)).astype(F32)
    return relu(y) if act_relu else y


def softmax(x):
    x = x.astype(F32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)

# ----------------------------------------------------------------------------------------
# A2  XceptionBody  (net/xception_body.py:236-379)
# ----------------------------------------------------------------------------------------
XC_EPS = 1e-4


def _sep_bn(x, w, name, pre_relu=True, dilation=1, taps=None):
    if pre_relu:
        x = relu(x)
    y = separable_conv2d(x, w[name + '/depthwise_kernel'], w[name + '/pointwise_kernel'], dilation)
    if taps is not None:
        taps[name] = y
    return batch_norm(y, w, name + '_bn', XC_EPS)


def xception_body(x, w, taps=None):
    """x [N,H,W,3] -> (mid [N,h,w,728], out [N,h,w,2048]).  `taps`, if a dict, receives the
    pre-BN output of every conv (used by the BN calibration script)."""
    def conv_bn(x, name, bn, stride, padding):
        y = conv2d(x, w[name + '/kernel'], stride, padding)
        if taps is not None:
            taps[name] = y
        return batch_norm(y, w, bn, XC_EPS)

    x = relu(conv_bn(x, 'block1_conv1', 'block1_conv1_bn', 2, 'VALID'))
    x = relu(conv_bn(x, 'block1_conv2', 'block1_conv2_bn', 1, 'VALID'))
    res = conv_bn(x, 'conv2d_1', 'batch_normalization_1', 2, 'SAME')
    x = _sep_bn(x, w, 'block2_sepconv1', pre_relu=False, taps=taps)      # :268-277, no leading ReLU
    x = _sep_bn(x, w, 'block2_sepconv2', taps=taps)
    x = max_pool_3x3_s2_same(x) + res
    res = conv_bn(x, 'conv2d_2', 'batch_normalization_2', 2, 'SAME')
    x = _sep_bn(x, w, 'block3_sepconv1', taps=taps)
    x = _sep_bn(x, w, 'block3_sepconv2', taps=taps)
    x = max_pool_3x3_s2_same(x) + res
    res = conv_bn(x, 'conv2d_3', 'batch_normalization_3', 2, 'SAME')
    x = _sep_bn(x, w, 'block4_sepconv1', taps=taps)
    x = _sep_bn(x, w, 'block4_sepconv2', taps=taps)
    x = max_pool_3x3_s2_same(x) + res
    for b in range(5, 13):
        res = x
        for s in (1, 2, 3):
            x = _sep_bn(x, w, 'block%d_sepconv%d' % (b, s), taps=taps)
        x = x + res
    mid = relu(x)                                                          # :339
    res = conv_bn(x, 'conv2d_4', 'batch_normalization_4', 1, 'SAME')      # stride 1, :341
    x = _sep_bn(x, w, 'block13_sepconv1', taps=taps)
    x = _sep_bn(x, w, 'block13_sepconv2', taps=taps)
    x = x + res
    x = relu(_sep_bn(x, w, 'block14_sepconv1', pre_relu=False, dilation=2, taps=taps))   # :354-364
    x = relu(_sep_bn(x, w, 'block14_sepconv2', pre_relu=False, dilation=2, taps=taps))   # :366-376
    return mid, x

# ----------------------------------------------------------------------------------------
# A3  get_rpn (:381-400)   A8  large_sep_kernel (:450-475)
# ----------------------------------------------------------------------------------------

def get_rpn(mid, w, scope='rpn_head'):
    r = relu(conv2d(mid, w[scope + '/conv2d/kernel'], 1, 'SAME', bias=w[scope + '/conv2d/bias']))
    cls = conv2d(r, w[scope + '/conv2d_1/kernel'], 1, 'SAME', bias=w[scope + '/conv2d_1/bias'])
    box = conv2d(r, w[scope + '/conv2d_2/kernel'], 1, 'SAME', bias=w[scope + '/conv2d_2/bias'])
    return cls, box


def large_sep_kernel(x, w, scope='large_sep_feature', taps=None):
    outs = []
    for br in ('Branch_0', 'Branch_1'):
        a = conv2d(x, w['%s/%s/conv2d/kernel' % (scope, br)], 1, 'SAME', bias=w['%s/%s/conv2d/bias' % (scope, br)])
        b = conv2d(a, w['%s/%s/conv2d_1/kernel' % (scope, br)], 1, 'SAME', bias=w['%s/%s/conv2d_1/bias' % (scope, br)])
        outs.append(b)
    y = (outs[0] + outs[1]).astype(F32)
    if taps is not None:
        taps[scope] = y
    return relu(batch_norm(y, w, scope + '/batch_normalization', 1e-5))     # resnet_v2.py:37,41-50

# ----------------------------------------------------------------------------------------
# A5  AnchorCreator (preprocessing/anchor_manipulator.py:698-757)
# ----------------------------------------------------------------------------------------

def layer_anchors(img_shape=(480, 480), layer_shape=(30, 30), anchor_scale=(0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8),
                  extra_anchor_scale=(0.1,), anchor_ratio=(1., 2., .5), layer_step=16, offset=0.5):
    xs, ys = np.meshgrid(np.arange(layer_shape[1]), np.arange(layer_shape[0]))
    y_on = ((ys.astype(F32) + F32(offset)) * F32(layer_step) / F32(img_shape[0])).astype(F32)
    x_on = ((xs.astype(F32) + F32(offset)) * F32(layer_step) / F32(img_shape[1])).astype(F32)
    hs, ws = [], []
    for s in extra_anchor_scale:
        hs.append(s)
        ws.append(s)
    for s in anchor_scale:
        for r in anchor_ratio:
            hs.append(s / math.sqrt(r))
            ws.append(s * math.sqrt(r))
    return y_on, x_on, np.array(hs, F32), np.array(ws, F32)


# A6  decode_all_anchors(squeeze_inner=True)  (anchor_manipulator.py:641-669), prior_scaling 1
def decode_all_anchors(loc, anchors):
    """loc [N, H*W*A, 4] (cy,cx,h,w deltas) -> boxes [N, H*W*A, 4]."""
    yref, xref, href, wref = anchors
    Hh, Ww = yref.shape
    A = href.shape[0]
    l = loc.reshape(-1, Hh, Ww, A, 4).astype(F32)
    ph = np.exp(l[..., 2]) * href
    pw = np.exp(l[..., 3]) * wref
    pcy = l[..., 0] * href + yref[..., None]
    pcx = l[..., 1] * wref + xref[..., None]
    out = np.stack([pcy - ph / F32(2), pcx - pw / F32(2), pcy + ph / F32(2), pcx + pw / F32(2)], axis=-1)
    return out.reshape(-1, Hh * Ww * A, 4).astype(F32)


# A11  ext_decode_rois (anchor_manipulator.py:671-683)
def ext_decode_rois(rois, pred):
    rois = rois.astype(F32)
    pred = pred.astype(F32)
    href = rois[..., 2] - rois[..., 0]
    wref = rois[..., 3] - rois[..., 1]
    yref = rois[..., 0] + href / F32(2)
    xref = rois[..., 1] + wref / F32(2)
    ph = np.exp(pred[..., 2]) * href
    pw = np.exp(pred[..., 3]) * wref
    pcy = pred[..., 0] * href + yref
    pcx = pred[..., 1] * wref + xref
    return np.stack([pcy - ph / F32(2), pcx - pw / F32(2), pcy + ph / F32(2), pcx + pw / F32(2)], axis=-1).astype(F32)

# ----------------------------------------------------------------------------------------
# TF kernels restated: top_k, NonMaxSuppressionV2
# ----------------------------------------------------------------------------------------

def top_k(scores, k):
    """descending, ties -> lower index first (stable)."""
    order = np.argsort(-scores.astype(F32), kind='stable')[:k]
    return scores[order], order


def iou_tf(b, i, j):
    yi0, yi1 = min(b[i, 0], b[i, 2]), max(b[i, 0], b[i, 2])
    xi0, xi1 = min(b[i, 1], b[i, 3]), max(b[i, 1], b[i, 3])
    yj0, yj1 = min(b[j, 0], b[j, 2]), max(b[j, 0], b[j, 2])
    xj0, xj1 = min(b[j, 1], b[j, 3]), max(b[j, 1], b[j, 3])
    ai = F32(yi1 - yi0) * F32(xi1 - xi0)
    aj = F32(yj1 - yj0) * F32(xj1 - xj0)
    if ai <= 0 or aj <= 0:
        return F32(0)
    ih = max(F32(min(yi1, yj1) - max(yi0, yj0)), F32(0))
    iw = max(F32(min(xi1, xj1) - max(xi0, xj0)), F32(0))
    inter = F32(ih * iw)
    return F32(inter / F32(F32(ai + aj) - inter))


def _iou_row(b, i, sel):
    """IoU of box i against boxes `sel` (vectorised, fp32, same expression order as iou_tf)."""
    bb = b[sel]
    y0 = np.minimum(bb[:, 0], bb[:, 2]); y1 = np.maximum(bb[:, 0], bb[:, 2])
    x0 = np.minimum(bb[:, 1], bb[:, 3]); x1 = np.maximum(bb[:, 1], bb[:, 3])
    yi0, yi1 = min(b[i, 0], b[i, 2]), max(b[i, 0], b[i, 2])
    xi0, xi1 = min(b[i, 1], b[i, 3]), max(b[i, 1], b[i, 3])
    ai = F32(F32(yi1 - yi0) * F32(xi1 - xi0))
    aj = ((y1 - y0) * (x1 - x0)).astype(F32)
    ih = np.maximum(np.minimum(yi1, y1) - np.maximum(yi0, y0), F32(0)).astype(F32)
    iw = np.maximum(np.minimum(xi1, x1) - np.maximum(xi0, x0), F32(0)).astype(F32)
    inter = (ih * iw).astype(F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = (inter / ((ai + aj).astype(F32) - inter)).astype(F32)
    iou = np.where((aj <= 0) | (ai <= 0), F32(0), iou)
    return iou


def non_max_suppression(boxes, scores, max_output, thr):
    boxes = boxes.astype(F32)
    order = np.argsort(-scores.astype(F32), kind='stable')
    sel = []
    thr = F32(thr)
    for i in order:
        if len(sel) >= max_output:
            break
        if sel:
            if np.any(_iou_row(boxes, i, np.array(sel)) > thr):
                continue
        sel.append(int(i))
    return np.array(sel, np.int64)

# ----------------------------------------------------------------------------------------
# A7  get_proposals (net/xception_body.py:402-448, helpers :41-213)
# ----------------------------------------------------------------------------------------

def bboxes_clip(ref, b):
    """_bboxes_clip :173-194 / eval_helper.bboxes_clip :365-404."""
    ymin = np.maximum(b[:, 0], F32(ref[0])); xmin = np.maximum(b[:, 1], F32(ref[1]))
    ymax = np.minimum(b[:, 2], F32(ref[2])); xmax = np.minimum(b[:, 3], F32(ref[3]))
    ymin = np.minimum(ymin, ymax); xmin = np.minimum(xmin, xmax)
    return np.stack([ymin, xmin, ymax, xmax], axis=1).astype(F32)


def _center_filter_mask(b, min_size):
    ws = b[:, 3] - b[:, 1]
    hs = b[:, 2] - b[:, 0]
    xc = b[:, 1] + ws / F32(2)
    yc = b[:, 0] + hs / F32(2)
    return (ws > F32(min_size)) & (hs > F32(min_size)) & (xc > 0) & (yc > 0) & (xc < 1) & (yc < 1)


def _pad_rows(x, size):
    if x.shape[0] >= size:
        return x
    pad = [(0, size - x.shape[0])] + [(0, 0)] * (x.ndim - 1)
    return np.pad(x, pad)


def get_proposals_single(score, boxes, pre_n, post_n, nms_thr, min_size, trace=None):
    b = bboxes_clip([0., 0., 1., 1.], boxes.astype(F32))
    keep = _center_filter_mask(b, min_size)                              # _filter_and_sort_boxes :133-158
    s_k, b_k = score[keep].astype(F32), b[keep]
    s_s, idx = top_k(s_k, min(s_k.shape[0], pre_n))
    b_s = b_k[idx]
    n_cand = s_s.shape[0]
    s_p, b_p = _pad_rows(s_s, pre_n), _pad_rows(b_s, pre_n)
    sel = non_max_suppression(b_p, s_p, post_n, nms_thr)                 # _bboxes_nms :57-67
    s_n, b_n = _pad_rows(s_p[sel], post_n), _pad_rows(b_p[sel], post_n)
    pos = s_n > 0                                                        # _upsample_rois :196-213
    s_u, b_u = s_n[pos], b_n[pos]
    n_keep = s_u.shape[0]
    if n_keep < 1:
        s_u, b_u = np.array([1.], F32), np.array([[.2, .2, .8, .8]], F32)
    n = s_u.shape[0]
    if n < post_n:
        left = post_n - n
        idxs = np.concatenate([np.tile(np.arange(n), left // n + 1), np.arange(n)[:left % n]])
        s_u, b_u = s_u[idxs], b_u[idxs]
    if trace is not None:
        trace.update(sorted_scores=s_p, sorted_boxes=b_p, n_cand=n_cand, nms_sel=sel, n_keep=n_keep)
    return s_u.astype(F32), b_u.astype(F32)


def get_proposals(obj_score, boxes, pre_n=5000, post_n=1000, nms_thr=0.7, min_size=16. / 480, traces=None):
    outs = []
    for n in range(obj_score.shape[0]):
        tr = {} if traces is not None else None
        outs.append(get_proposals_single(obj_score[n], boxes[n], pre_n, post_n, nms_thr, min_size, tr)[1])
        if traces is not None:
            traces.append(tr)
    return np.stack(outs).astype(F32)

# ----------------------------------------------------------------------------------------
# A9  PsRoiAlign : C oracle (oracle/psroialign_ref.c) + NumPy twin for tiny cases
# ----------------------------------------------------------------------------------------
_clib = None


def build_c_oracle(force=False):
    so = os.path.join(_HERE, 'liboracle_psroialign.so')
    src = os.path.join(_HERE, 'psroialign_ref.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', so, src])
    return so


def build_cpp_baseline(force=False):
    """oracle/liblighthead_cpu.so: the C++17 + OpenMP restatement of the whole eval forward (lighthead_cpu.cpp; the
    CPU baseline SURVEY.md 8d(1) specifies) linked with psroialign_ref.c.  x86-64-v3 baseline ISA with an AVX-512
    clone of the GEMM micro-kernel picked at load time, so the same .so runs here and on the GPU box's host."""
    so = os.path.join(_HERE, 'liblighthead_cpu.so')
    srcs = [os.path.join(_HERE, 'lighthead_cpu.cpp'), os.path.join(_HERE, 'psroialign_ref.c')]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        obj = os.path.join(_HERE, 'psroialign_ref.o')
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-c', srcs[1], '-o', obj])
        subprocess.check_call(['g++', '-O3', '-std=c++17', '-march=x86-64-v3', '-fopenmp', '-fPIC', '-shared', '-o', so,
                               srcs[0], obj])
    return so


_cpp = None
FAST_PATH_DESCRIPTION = ('C++17 + OpenMP restatement of the reference graph (oracle/lighthead_cpu.cpp, fp32, AVX-512/AVX2; '
                         'image-parallel: one image per thread at a time)')


def effective_cpus():
    """CPUs this process can actually use at once: the scheduler affinity, cut by the container's cgroup CPU quota
    (cpu.max = "quota period": the GPU boxes show 256 hardware threads and grant 16 cores' worth of time -- threads beyond
    the quota only time-share it, which is why the baseline used to be "fastest at 16 threads")."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for f in ('/sys/fs/cgroup/cpu.max',):
        try:
            q, p = open(f).read().split()[:2]
            if q != 'max':
                n = min(n, max(1, int(math.ceil(float(q) / float(p)))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            n = min(n, max(1, int(math.ceil(q / p))))
    except (OSError, ValueError):
        pass
    return n


class CppForward(object):
    """the C++ baseline as an object that keeps its packed weights (building them is not part of a timed forward)"""
    def __init__(self, w, image_size=480, rpn_post_nms_top_n=300):
        global _cpp
        if _cpp is None:
            # a blocked OpenMP thread should sleep, not spin: the box may expose more hardware threads than the
            # container may use (256 on the GPU box), and spinning waiters then starve the workers
            os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
            _cpp = ctypes.CDLL(build_cpp_baseline())
            _cpp.lhcpu_create.restype = ctypes.c_void_p
            _cpp.lhcpu_error.restype = ctypes.c_char_p
        self.h = ctypes.c_void_p(_cpp.lhcpu_create(int(image_size), int(rpn_post_nms_top_n)))
        for name, arr in w.items():
            a = np.ascontiguousarray(arr, np.float32)
            dims = (ctypes.c_int64 * a.ndim)(*a.shape)
            _cpp.lhcpu_set_weight(self.h, name.encode(), ctypes.c_void_p(a.ctypes.data), a.ndim, dims)
        if _cpp.lhcpu_build(self.h) != 0:
            raise RuntimeError(_cpp.lhcpu_error(self.h).decode())
        self.S = image_size
        self.threads = int(_cpp.lhcpu_threads())

    def set_threads(self, n):
        _cpp.lhcpu_set_threads(int(n))
        self.threads = int(_cpp.lhcpu_threads())

    def tune_threads(self, images, candidates=None):
        """pick the OpenMP thread count that runs one call fastest on THIS box; returns {threads: seconds}.  Candidates:
        the CPUs the process can really use at once (effective_cpus(): affinity cut by the cgroup quota), twice and half
        that.  The baseline runs image-parallel when a call brings at least one image per two threads."""
        import time
        hw = os.cpu_count() or 1
        n = int(np.asarray(images).shape[0])
        if candidates is None:
            eff = effective_cpus()
            candidates = (eff, 2 * eff, max(eff // 2, 1))
        seen = {}
        for c in sorted(set(max(1, min(int(c), hw)) for c in candidates)):
            self.set_threads(c)
            self(images)
            t = time.time()
            self(images)
            seen[c] = time.time() - t
        self.set_threads(min(seen, key=seen.get))
        return seen

    def __call__(self, images_nchw):
        x = np.ascontiguousarray(images_nchw, np.float32)
        n = x.shape[0]
        s = np.zeros((n, 20, 200), np.float32)
        b = np.zeros((n, 20, 200, 4), np.float32)
        rc = _cpp.lhcpu_forward(self.h, ctypes.c_void_p(x.ctypes.data), n, ctypes.c_void_p(s.ctypes.data),
                                ctypes.c_void_p(b.ctypes.data))
        if rc != 0:
            raise RuntimeError(_cpp.lhcpu_error(self.h).decode())
        return [{c + 1: (s[i, c], b[i, c]) for c in range(20)} for i in range(n)]

    def __del__(self):
        try:
            _cpp.lhcpu_destroy(self.h)
        except Exception:
            pass


_fast_cache = {}


def lighthead_forward_fast(images_nchw, w, rpn_post_nms_top_n=300):
    """bench.py's cpu_baseline leg: the C++ restatement, weights packed once per (weights, R)."""
    key = (id(w), int(rpn_post_nms_top_n), int(images_nchw.shape[2]))
    if key not in _fast_cache:
        _fast_cache.clear()
        f = CppForward(w, images_nchw.shape[2], rpn_post_nms_top_n)
        f.tuning = f.tune_threads(images_nchw)
        _fast_cache[key] = f
    return _fast_cache[key](images_nchw)


def _c():
    global _clib
    if _clib is None:
        _clib = ctypes.CDLL(build_c_oracle())
        _clib.oracle_psroialign_fwd.restype = ctypes.c_int
        _clib.oracle_psroialign_grad.restype = ctypes.c_int
    return _clib


def ps_roi_align(inputs, rois, grid_w, grid_h, pool_method='max', layout='NCHW'):
    """Same positional signature as op_module.ps_roi_align (light_head_rfcn_eval.py:143-155).
    inputs [N,C,H,W] (or [N,H,W,C] with layout='NHWC'), rois [N,R,4] (cy,cx,h,w)."""
    inputs = np.ascontiguousarray(inputs, F32)
    rois = np.ascontiguousarray(rois, F32)
    if layout == 'NCHW':
        N, C, H, W = inputs.shape
        lay, ldc = 0, C
    else:
        N, H, W, C = inputs.shape
        lay, ldc = 1, C
    R = rois.shape[1]
    gs = grid_w * grid_h
    pooled = np.empty((N, R, gs, C // gs), F32)
    index = np.empty((N, R, gs, C // gs), np.int32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _c().oracle_psroialign_fwd(inputs.ctypes.data_as(fp), rois.ctypes.data_as(fp), pooled.ctypes.data_as(fp),
                                    index.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                    N, C, H, W, R, grid_w, grid_h, 1 if 'max' in pool_method else 0, lay, ldc)
    if rc != 0:
        raise ValueError('oracle_psroialign_fwd rc=%d' % rc)
    return pooled, index


def ps_roi_align_grad(inputs, rois, pooled_features_grad, pooled_index, grid_w, grid_h, pool_method='max'):
    """Same positional signature as op_module.ps_roi_align_grad (ps_roi_align_grad_op.cc:39-57;
    cpp/PSROIPooling/test_op.py:93-104).  inputs only gives the [N,C,H,W] shape."""
    N, C, H, W = inputs.shape
    rois = np.ascontiguousarray(rois, F32)
    g = np.ascontiguousarray(pooled_features_grad, F32)
    idx = np.ascontiguousarray(pooled_index, np.int32)
    R = rois.shape[1]
    out = np.empty((N, C, H, W), F32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _c().oracle_psroialign_grad(rois.ctypes.data_as(fp), g.ctypes.data_as(fp),
                                     idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), out.ctypes.data_as(fp),
                                     N, C, H, W, R, grid_w, grid_h, 1 if 'max' in pool_method else 0, 0, C)
    if rc != 0:
        raise ValueError('oracle_psroialign_grad rc=%d' % rc)
    return out


def ps_roi_align_np(inputs, rois, grid_w, grid_h, pool_method='max'):
    """Pure-NumPy-scalar twin of psroialign_ref.c (NCHW); small cases only."""
    inputs = np.asarray(inputs, F32)
    rois = np.asarray(rois, F32)
    N, C, H, W = inputs.shape
    R = rois.shape[1]
    gs = grid_w * grid_h
    bank = C // gs
    use_max = 'max' in pool_method
    pooled = np.zeros((N, R, gs, bank), F32)
    index = np.zeros((N, R, gs, bank), np.int32)
    FMIN = np.finfo(F32).tiny
    for n in range(N):
        for r in range(R):
            roi = rois[n, r]
            if roi[2] < FMIN or roi[3] < FMIN:
                continue
            yc = F32(roi[0] * F32(H)); xc = F32(roi[1] * F32(W))
            rh = max(F32(roi[2] * F32(H)), F32(1)); rw = max(F32(roi[3] * F32(W)), F32(1))
            ymin = max(F32(yc - F32(rh / F32(2))), F32(0)); xmin = max(F32(xc - F32(rw / F32(2))), F32(0))
            ymax = min(F32(yc + F32(rh / F32(2))), F32(F32(H) - FMIN)); xmax = min(F32(xc + F32(rw / F32(2))), F32(F32(W) - FMIN))
            bin_w = F32(F32(xmax - xmin) / F32(grid_w)); bin_h = F32(F32(ymax - ymin) / F32(grid_h))
            n_w = int(bin_w) + 1; n_h = int(bin_h) + 1
            step_w = F32(bin_w / F32(n_w)); step_h = F32(bin_h / F32(n_h))
            for pos in range(gs):
                row, col = pos // grid_w, pos % grid_w
                x0 = F32(xmin + F32(bin_w * F32(col))); y0 = F32(ymin + F32(bin_h * F32(row)))
                for ch in range(bank):
                    plane = inputs[n, pos * bank + ch]
                    acc = -np.finfo(F32).max if use_max else F32(0)
                    arg = 0
                    for i in range(n_h):
                        for j in range(n_w):
                            x = F32(np.float64(F32(x0 + F32(step_w * F32(j)))) + np.float64(step_w) / 2.)
                            y = F32(np.float64(F32(y0 + F32(step_h * F32(i)))) + np.float64(step_h) / 2.)
                            ix, iy = int(x), int(y)
                            fx32 = F32(x - F32(ix)); fy32 = F32(y - F32(iy))
                            fx = np.float64(fx32); fy = np.float64(fy32)
                            iy1, ix1 = min(iy + 1, H - 1), min(ix + 1, W - 1)
                            # C typing of ps_roi_align_op.cc:171-174: the first three products contain a
                            # `1.` literal and are evaluated in double; the fourth (fx*fy*f11) is
                            # float*float*float, i.e. two f32 roundings, then promoted for the sum.
                            v = (1. - fx) * (1. - fy) * np.float64(plane[iy, ix]) + (1. - fx) * fy * np.float64(plane[iy1, ix]) \
                                + fx * (1. - fy) * np.float64(plane[iy, ix1]) + np.float64(F32(F32(fx32 * fy32) * plane[iy1, ix1]))
                            t = F32(v)
                            if use_max:
                                if acc < t:
                                    acc, arg = t, n_w * i + j
                            else:
                                acc = F32(acc + t)
                    if not use_max:
                        acc = F32(acc / F32(n_h * n_w))
                    pooled[n, r, pos, ch] = acc
                    index[n, r, pos, ch] = arg if use_max else 0
    return pooled, index

# ----------------------------------------------------------------------------------------
# A10  get_head (net/xception_body.py:477-560), eval branch (using_ohem=False)
# ----------------------------------------------------------------------------------------

def point2center(b):
    h = b[..., 2] - b[..., 0]
    w = b[..., 3] - b[..., 1]
    return np.stack([b[..., 0] + h / F32(2), b[..., 1] + w / F32(2), h, w], axis=-1).astype(F32)


def get_head(feat_nhwc, proposals, w, grid_w=7, grid_h=7, scope='final_head', trace=None):
    yxhw = point2center(proposals.astype(F32))
    pooled, index = ps_roi_align(feat_nhwc, yxhw, grid_w, grid_h, 'max', layout='NHWC')
    N, R = proposals.shape[:2]
    x = pooled.reshape(N, R, -1)
    fc = dense(x, w, scope + '/subnet_fc', act_relu=True)
    cls = dense(fc, w, scope + '/fc_cls')
    reg = dense(fc, w, scope + '/fc_loc')
    if trace is not None:
        trace.update(pooled=pooled, pooled_index=index, subnet_fc=fc)
    return cls, reg

# ----------------------------------------------------------------------------------------
# A12  bboxes_eval detection part (light_head_rfcn_eval.py:263-287; utility/eval_helper.py)
# ----------------------------------------------------------------------------------------

def filter_min_size(image_shape, net_input=(480, 480), ratio=0.03):
    """eval_helper.filter_boxes :296."""
    v = F32(F32(int(image_shape[0]) * int(image_shape[1])) / F32(net_input[0] * net_input[1]))
    return max(F32(0.0001), F32(F32(ratio) * np.sqrt(v)))


def bboxes_eval(cls_logits, boxes, image_shape=(480, 480), bbox_img=(0., 0., 1., 1.), num_classes=21,
                select_threshold=0.01, nms_threshold=0.3, nms_topk=200, net_input=(480, 480)):
    """cls_logits [R,21], boxes [R,4] -> {c: (scores[nms_topk], boxes[nms_topk,4])}."""
    prob = softmax(cls_logits.reshape(-1, num_classes))
    boxes = boxes.reshape(-1, 4).astype(F32)
    min_size = filter_min_size(image_shape, net_input)
    out = {}
    for c in range(1, num_classes):
        s = prob[:, c]
        fmask = (s > F32(select_threshold)).astype(F32)                 # tf_bboxes_select_layer :581-585
        s = s * fmask
        b = boxes * fmask[:, None]
        b = bboxes_clip(bbox_img, b)                                    # bboxes_clip
        keep = _center_filter_mask(b, min_size)                         # filter_boxes (dict branch -> keep_top_k=100)
        s, b = _pad_rows(s[keep], 100), _pad_rows(b[keep], 100)
        ref = np.asarray(bbox_img, F32)     # a float32 tensor in the reference: the extent below is a float32 difference
        v = np.array([ref[0], ref[1], ref[0], ref[1]], F32)                        # bboxes_resize :423-447
        sc = np.array([ref[2] - ref[0], ref[3] - ref[1]] * 2, F32)                 # (found by tests/test_oracle_discrete.py)
        b = ((b - v) / sc).astype(F32)
        k = min(s.shape[0], nms_topk * 2)                               # bboxes_sort dict branch :348-355
        s, idx = top_k(s, k)
        b = b[idx]
        sel = non_max_suppression(b, s, nms_topk, nms_threshold)        # bboxes_nms :449-470
        out[c] = (_pad_rows(s[sel], nms_topk).astype(F32), _pad_rows(b[sel], nms_topk).astype(F32))
    return out

# ----------------------------------------------------------------------------------------
# A1  lighr_head_model_fn, eval mode (light_head_rfcn_eval.py:364-433)
# ----------------------------------------------------------------------------------------

def lighthead_forward(images_nchw, w, rpn_pre_nms_top_n=5000, rpn_post_nms_top_n=1000, rpn_nms_thres=0.7,
                      rpn_min_size=16. / 480, num_classes=21, select_threshold=0.01, nms_threshold=0.3,
                      nms_topk=200, image_shapes=None, trace=None):
    """images [N,3,S,S] whitened f32 -> list (per image) of {c: (scores[200], boxes[200,4])}.
    The reference evaluates with batch 1 (light_head_rfcn_eval.py:212,396); images here are
    independent, so a batch is just N such evaluations."""
    x = np.transpose(np.asarray(images_nchw, F32), (0, 2, 3, 1))
    N, S = x.shape[0], x.shape[1]
    mid, out = xception_body(x, w)
    rpn_cls, rpn_box = get_rpn(mid, w)
    feat = large_sep_kernel(out, w)
    Hh, Ww = rpn_cls.shape[1:3]
    A = rpn_cls.shape[3] // 2
    obj = softmax(rpn_cls.reshape(-1, 2))[:, -1].reshape(N, -1)         # :393-396
    loc = rpn_box.reshape(N, -1, 4)
    anchors = layer_anchors((S, S), (Hh, Ww), layer_step=16)
    rpn_boxes = decode_all_anchors(loc, anchors)
    ptr = [] if trace is not None else None
    proposals = get_proposals(obj, rpn_boxes, rpn_pre_nms_top_n, rpn_post_nms_top_n, rpn_nms_thres, rpn_min_size, ptr)
    htr = {} if trace is not None else None
    cls, reg = get_head(feat, proposals, w, trace=htr)
    head_boxes = ext_decode_rois(proposals, reg)
    dets = []
    for n in range(N):
        shp = (S, S) if image_shapes is None else image_shapes[n]
        dets.append(bboxes_eval(cls[n], head_boxes[n], shp, (0., 0., 1., 1.), num_classes,
                                select_threshold, nms_threshold, nms_topk, (S, S)))
    if trace is not None:
        trace.update(mid=mid, out=out, rpn_cls=rpn_cls, rpn_box=rpn_box, feat=feat, objectness=obj,
                     rpn_boxes=rpn_boxes, proposals=proposals, proposal_traces=ptr, cls=cls, reg=reg,
                     head_boxes=head_boxes, head=htr)
    return dets

# ----------------------------------------------------------------------------------------
# A13  ResNet-50 v2 trunk (net/resnet_v2.py:311-345, up to the final batch_norm_relu)
# ----------------------------------------------------------------------------------------
RN_EPS = 1e-5


def _fixed_pad_conv(x, k, stride, kernel):
    """conv2d_fixed_padding :89-100."""
    if stride > 1:
        pt = (k - 1) // 2
        pb = (k - 1) - pt
        return conv2d(x, kernel, stride, ((pt, pb), (pt, pb)))
    return conv2d(x, kernel, 1, 'SAME')


def resnet50_trunk(x, w, taps=None):
    ci = [0]
    bi = [0]

    def conv(x, k, stride):
        name = 'conv2d' if ci[0] == 0 else 'conv2d_%d' % ci[0]
        ci[0] += 1
        return _fixed_pad_conv(x, k, stride, w[name + '/kernel'])

    def bn_relu(x):
        name = 'batch_normalization' if bi[0] == 0 else 'batch_normalization_%d' % bi[0]
        bi[0] += 1
        if taps is not None:
            taps[name] = x
        return relu(batch_norm(x, w, name, RN_EPS))

    x = conv(x, 7, 2)
    x = max_pool_3x3_s2_same(x)
    for filters, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(blocks):
            shortcut = x
            y = bn_relu(x)
            s = stride if b == 0 else 1
            if b == 0:
                shortcut = conv(y, 1, s)
            y = conv(y, 1, 1)
            y = bn_relu(y)
            y = conv(y, 3, s)
            y = bn_relu(y)
            y = conv(y, 1, 1)
            x = y + shortcut
    return bn_relu(x)


# ----------------------------------------------------------------------------------------
# F1  light_head_preprocess_for_eval / _for_test (preprocessing/common_preprocessing.py:383-458,
#     tf_image.resize_image :307-319); TF1 legacy bilinear restated from TF's published kernel
#     (src = dst*in/out, no half-pixel centre) -- parity unpinned like the other TF kernels.
# ----------------------------------------------------------------------------------------

def preprocess_for_eval(image_u8, out_size=480):
    """uint8 [H,W,3] -> f32 [3,S,S] (data_format 'NCHW')."""
    img = np.asarray(image_u8, np.uint8)
    H, W = img.shape[:2]
    S = out_size
    means = np.array([123.68 / 127.5, 116.78 / 127.5, 103.94 / 127.5]).astype(F32)
    x = ((img.astype(F32) * F32(1.0 / 255.0)) * F32(2.0) - means).astype(F32)
    hs, ws = F32(F32(H) / F32(S)), F32(F32(W) / F32(S))
    fy = (np.arange(S, dtype=F32) * hs).astype(F32)
    fx = (np.arange(S, dtype=F32) * ws).astype(F32)
    y0, x0 = fy.astype(np.int64), fx.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    ly = (fy - y0.astype(F32)).astype(F32)[:, None, None]
    lx = (fx - x0.astype(F32)).astype(F32)[None, :, None]
    tl, tr = x[y0][:, x0], x[y0][:, x1]
    bl, br = x[y1][:, x0], x[y1][:, x1]
    top = (tl + ((tr - tl) * lx).astype(F32)).astype(F32)
    bot = (bl + ((br - bl) * lx).astype(F32)).astype(F32)
    out = (top + ((bot - top) * ly).astype(F32)).astype(F32)
    return np.ascontiguousarray(out.transpose(2, 0, 1))
