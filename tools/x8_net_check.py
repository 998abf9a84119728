#!/usr/bin/env python
"""Stage-by-stage error of the detector against the oracle: f16x3 vs cross='fp8' (GPU box)."""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet import weights as W                             # noqa: E402
from xdet.model import LightHeadDetector                  # noqa: E402
from xdet.runtime import set_precision                    # noqa: E402
from oracle import lighthead_oracle as oracle             # noqa: E402  (checker only)

lh = W.make_lighthead_weights(1234)
imgs = W.synthetic_images(2, 480, seed=0)
tr = {}
ref = oracle.lighthead_forward(imgs, lh, rpn_post_nms_top_n=300, trace=tr)


def rel(a, b):
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


for cross in ('f16', 'fp8'):
    set_precision('f16x3')
    det = LightHeadDetector(lh, image_size=480, max_batch=2, rpn_post_nms_top_n=300, large_sep='spectral', cross=cross)
    set_precision('f32')
    if cross == 'fp8':
        det.calibrate(W.synthetic_images(2, 480, seed=4242))
    got = det.forward(imgs)
    n = 2
    rpn = det.buffer('rpn_out', n).numpy()
    print('cross=%s  x8 planes %d' % (cross, det.x8_planes()))
    print('   mid %.2e  out %.2e  rpn_cls %.2e  rpn_box %.2e  feat %.2e' % (
        rel(np.maximum(det.buffer('mid_x', n).numpy(), 0), tr['mid']), rel(det.buffer('out', n).numpy(), tr['out']),
        rel(rpn[..., :44], tr['rpn_cls']), rel(rpn[..., 44:132], tr['rpn_box']), rel(det.buffer('feat', n).numpy(), tr['feat'])))
    na = 30 * 30 * 22
    print('   objectness %.2e  rpn_boxes %.2e' % (np.abs(det.flat('objectness', (n, na)) - tr['objectness']).max(),
                                                 np.abs(det.flat('rpn_boxes', (n, na, 4)) - tr['rpn_boxes']).max()))
    tot = mat = 0
    worst = 0.0
    for i in range(n):
        for c, (s, b) in ref[i].items():
            gs, gb = got[i][c]
            k = int((s > 0).sum())
            tot += k
            kk = min(k, int((gs > 0).sum()))
            d = max(float(np.abs(gs[:kk] - s[:kk]).max()) if kk else 0.0, float(np.abs(gb[:kk] - b[:kk]).max()) if kk else 0.0)
            worst = max(worst, d) if d < 0.05 else worst
            mat += int(((np.abs(gs[:kk] - s[:kk]) < 1e-3) & (np.abs(gb[:kk] - b[:kk]).max(1) < 1e-3)).sum())
    print('   detections %d, position-wise within 1e-3: %d, largest small difference %.2e' % (tot, mat, worst))
