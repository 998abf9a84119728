#!/usr/bin/env python
"""Why does a class list of the GPU detector differ from the oracle's?  (GPU box.)  Runs the second synthetic model of
tests/test_gpu_e2e.py::test_second_weight_set, finds every (image, class) list with an unmatched detection and prints, for
each such detection, the kept boxes of the OTHER side that overlap it near the NMS threshold, with scores -- the evidence
behind `assert_match_or_score_tie`'s two admissible causes (a score tie between two mutually suppressing candidates; a
candidate whose IoU with a kept box sits at the threshold).   python tools/diag_nms_tie.py [--no-ksplit | --ksplit-all]"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
sys.path.insert(0, os.path.join(R_, 'tests'))
import numpy as np                                                        # noqa: E402


def main():
    from oracle import lighthead_oracle as O
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision
    from test_gpu_e2e import iou
    gains = {'rpn_head/conv2d_1/kernel': 1.0, 'rpn_head/conv2d_2/kernel': 0.5, 'final_head/fc_cls/kernel': 3.0,
             'final_head/fc_loc/kernel': 1.0}
    w = W.make_lighthead_weights(777, gains=gains)
    imgs = W.synthetic_images(2, 480, seed=555)
    set_precision('f16x3')
    det = LightHeadDetector(w, image_size=480, max_batch=2, rpn_post_nms_top_n=300, ksplit='off' if '--no-ksplit' in sys.argv else 'all' if '--ksplit-all' in sys.argv else 'on')
    got = det.forward(imgs)
    tr = {}
    ref = O.lighthead_forward(imgs, w, rpn_post_nms_top_n=300, trace=tr)
    cr = det.buffer('cls_reg', 2).numpy().reshape(2, 300, -1)
    hb = det.flat('head_boxes', (2, 300, 4))
    print('max |cls logits - oracle| %.2e   max |head boxes - oracle| %.2e' %
          (float(np.abs(cr[..., :21] - tr['cls']).max()) if np.abs(det.flat('proposals', (2, 300, 4)) - tr['proposals']).max() < 1e-5 else float('nan'),
           float(np.abs(np.sort(hb.reshape(2, -1), 1) - np.sort(tr['head_boxes'].reshape(2, -1), 1)).max())))
    for i in range(2):
        for c in ref[i]:
            gs, gb = got[i][c]
            rs, rb = ref[i][c]
            kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
            used = np.zeros(kg, bool)
            un = []
            for j in range(kr):
                d = np.where(used, np.inf, np.maximum(np.abs(gs[:kg] - rs[j]), np.abs(gb[:kg] - rb[j]).max(1))) if kg else np.array([np.inf])
                if d.min() < 1e-3:
                    used[int(d.argmin())] = True
                else:
                    un.append(j)
            ex = [k for k in range(kg) if not used[k]]
            if not un and not ex:
                continue
            print('image %d class %d: oracle keeps %d, gpu keeps %d; oracle-only %s, gpu-only %s' % (i, c, kr, kg, un, ex))
            for side, idx, s_me, b_me, s_ot, b_ot, n_ot in (('oracle-only', un, rs, rb, gs, gb, kg), ('gpu-only', ex, gs, gb, rs, rb, kr)):
                for j in idx:
                    print('  %s: score %.7f box %s' % (side, s_me[j], np.round(b_me[j], 6).tolist()))
                    for m in range(n_ot):
                        v = iou(b_me[j], b_ot[m])
                        if v > 0.2:
                            print('      other side keeps: score %.7f IoU %.5f%s  box %s' %
                                  (s_ot[m], v, '  <-- at the threshold' if abs(v - 0.3) < 2e-3 else '', np.round(b_ot[m], 6).tolist()))


if __name__ == '__main__':
    main()
