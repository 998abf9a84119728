import sys
import os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np
from xdet import weights as W
from xdet.model import LightHeadDetector
from oracle import lighthead_oracle as O
w = W.make_lighthead_weights(1234)
for size, R, nimg in ((480, 300, 2), (256, 100, 1)):
    imgs = W.synthetic_images(nimg, size, seed=0 if size == 480 else 3)
    det = LightHeadDetector(w, image_size=size, max_batch=nimg, rpn_post_nms_top_n=R)
    got = det.forward(imgs)
    tr = {}
    ref = O.lighthead_forward(imgs, w, rpn_post_nms_top_n=R, trace=tr)
    n = nimg
    fm = tr['feat'].shape[1]
    na = fm * fm * 22
    obj = det.flat('objectness', (n, na)); print(size, 'obj err', np.abs(obj - tr['objectness']).max())
    rb = det.flat('rpn_boxes', (n, na, 4)); print('rpn_boxes err', np.abs(rb - tr['rpn_boxes']).max())
    props = det.flat('proposals', (n, R, 4))
    cnt = det.flat('prop_counts', (n, 4), np.int32); print('counts', cnt, [(t['n_cand'], t['n_keep']) for t in tr['proposal_traces']])
    d = np.abs(props - tr['proposals']).max(-1)
    print('proposal rows differing >1e-5:', (d > 1e-5).sum(axis=1), 'max', d.max())
    for i in range(n):
        bad = np.where(d[i] > 1e-5)[0]
        print(' img', i, 'bad idx', bad[:20])
    sb = det.flat('sorted_boxes', (n, 5000, 4)); ss = det.flat('sorted_scores', (n, 5000))
    for i in range(n):
        t = tr['proposal_traces'][i]
        ds = np.abs(ss[i] - t['sorted_scores']); db = np.abs(sb[i] - t['sorted_boxes']).max(-1)
        print(' sorted scores err', ds.max(), 'boxes rows >1e-5', (db > 1e-5).sum(), np.where(db > 1e-5)[0][:10])
    cr = det.buffer('cls_reg', n).numpy().reshape(n, R, -1)
    print('cls err', np.abs(cr[..., :21] - tr['cls']).max(), 'reg err', np.abs(cr[..., 21:25] - tr['reg']).max())
    hb = det.flat('head_boxes', (n, R, 4)); print('head_boxes err', np.abs(hb - tr['head_boxes']).max())
    for i in range(n):
        for c in range(1, 21):
            gs, gb = got[i][c]; rs, rbb = ref[i][c]
            kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
            if kg != kr or np.abs(gs - rs).max() > 1e-3 or np.abs(gb - rbb).max() > 1e-3:
                print('  det mismatch img', i, 'cls', c, kg, kr, np.abs(gs - rs).max(), np.abs(gb - rbb).max())
                # feed the oracle per-class with GPU cls/head boxes to isolate the post-processing
    # isolate: oracle post-processing on GPU head outputs
    for i in range(n):
        ref2 = O.bboxes_eval(cr[i, :, :21], hb[i], (size, size), (0., 0., 1., 1.), 21, 0.01, 0.3, 200, (size, size))
        mism = 0
        for c in range(1, 21):
            gs, gb = got[i][c]; rs, rbb = ref2[c]
            if int((gs > 0).sum()) != int((rs > 0).sum()) or np.abs(gs - rs).max() > 1e-5 or np.abs(gb - rbb).max() > 1e-5:
                mism += 1
                print('  POST mismatch img', i, 'cls', c, int((gs > 0).sum()), int((rs > 0).sum()))
        print(' post-processing-only mismatching classes:', mism)
