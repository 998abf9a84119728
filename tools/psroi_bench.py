#!/usr/bin/env python
"""PsRoiAlign forward alone (the net's form: NHWC 30x30x490 map, 64 images x 300 ROIs) for ROI sets of controlled
size: fixed cost vs per-sample cost.   python tools/psroi_bench.py   (GPU box; XDET_PSROI=element for the generic kernel)"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet._lib import lib, check                          # noqa: E402
from xdet.runtime import DeviceBuffer, Event, Stream, to_device   # noqa: E402


def main():
    use_max = 0 if '--mean' in sys.argv else 1          # --mean: the 'mean' pooling form
    nchw = '--nchw' in sys.argv                         # --nchw: the op's public layout (one channel per lane)
    n, h, w, r, g, c, ldc = 64, 30, 30, 300, 7, 490, 512
    rng = np.random.default_rng(0)
    feat = to_device(rng.standard_normal((n, c, h, w) if nchw else (n, h, w, ldc)).astype(np.float32))
    pool = DeviceBuffer(n * r * c * 4)
    st = Stream()
    for name, size in [('1 px', 1. / 30), ('0.2', 0.2), ('0.45', 0.45), ('0.7', 0.7), ('full', 1.0), ('mixed', None)]:
        cy, cx = rng.uniform(0.3, 0.7, (n, r)), rng.uniform(0.3, 0.7, (n, r))
        if size is None:
            hh, ww = rng.uniform(0.05, 1.0, (n, r)), rng.uniform(0.05, 1.0, (n, r))
        else:
            hh = ww = np.full((n, r), size)
        rois = to_device(np.stack([cy, cx, hh, ww], -1).astype(np.float32))
        bins = np.minimum(hh, 1) * h / g
        samples = float(np.mean((np.floor(np.minimum(hh, 1) * h / g) + 1) * (np.floor(np.minimum(ww, 1) * w / g) + 1)))

        def run():
            check(lib().xdet_psroialign_fwd(feat.ptr, rois.ptr, pool.ptr, None, n, c, h, w, r, g, g, use_max, 0 if nchw else 1, c if nchw else ldc, c, 0, st.handle))
        run()
        st.synchronize()
        e0, e1 = Event(), Event()
        e0.record(st)
        for _ in range(20):
            run()
        e1.record(st)
        st.synchronize()
        us = e0.elapsed_ms(e1) / 20 * 1e3
        print('ROI size %-6s ~%5.1f samples/bin: %7.1f us  (%.2f ns per bin-sample-channel)' %
              (name, samples, us, us * 1e3 / (n * r * c * samples)))


if __name__ == '__main__':
    main()
