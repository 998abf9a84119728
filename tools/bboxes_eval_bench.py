"""A12 alone on real head outputs: the detector runs 8 synthetic images, its class logits / decoded boxes are tiled to a
batch of 128 and xdet_bboxes_eval is timed on them.  python tools/bboxes_eval_bench.py [R ...]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/x-detector_amd')
from xdet import weights as W
from xdet.model import LightHeadDetector
from xdet._lib import lib, check
from xdet.runtime import to_device, to_host, DeviceBuffer, Event, synchronize

w = W.make_lighthead_weights(1234)
imgs = W.synthetic_images(8, 480, seed=1)
for R in [int(a) for a in sys.argv[1:]] or [300, 1000]:
    det = LightHeadDetector(w, image_size=480, max_batch=8, rpn_post_nms_top_n=R)
    ref = det.forward(imgs)
    t = det.buffer('cls_reg', 8)
    ld = t.ld
    cls = to_host(t.ptr, (8, R, ld), np.float32)
    hb = det.flat('head_boxes', (8, R, 4))
    for N in (1, 128):
        reps = (N + 7) // 8
        c = np.ascontiguousarray(np.tile(cls, (reps, 1, 1))[:N]); b = np.ascontiguousarray(np.tile(hb, (reps, 1, 1))[:N])
        d_c, d_b = to_device(c), to_device(b)
        shapes = to_device(np.full((N, 2), 480, np.int32)); bimg = to_device(np.tile(np.array([0, 0, 1, 1], np.float32), (N, 1)))
        d_os, d_ob = DeviceBuffer(N * 20 * 200 * 4), DeviceBuffer(N * 20 * 200 * 16)
        call = lambda: check(lib().xdet_bboxes_eval(d_c.ptr, ld, d_b.ptr, N, R, 21, shapes.ptr, bimg.ptr, 480, 480, 0.01, 0.3, 200,
                                                    d_os.ptr, d_ob.ptr, None))
        for _ in range(3):
            call()
        a, e = Event(), Event()
        a.record()
        for _ in range(20):
            call()
        e.record(); synchronize()
        sc = to_host(d_os.ptr, (N, 20, 200), np.float32); bx = to_host(d_ob.ptr, (N, 20, 200, 4), np.float32)
        same = all(np.array_equal(sc[i, k], ref[i % 8][k + 1][0]) and np.array_equal(bx[i, k], ref[i % 8][k + 1][1])
                   for i in range(N) for k in range(20))
        print('R=%4d N=%3d  bboxes_eval %7.1f us per call   equal to the forward pass: %s' % (R, N, a.elapsed_ms(e) * 50, same), flush=True)
