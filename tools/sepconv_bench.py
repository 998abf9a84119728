#!/usr/bin/env python
"""Micro-benchmark of the fused separable block (sepconv_fused.hip) on the entry-flow shapes (GPU box only).
    python tools/sepconv_bench.py --batch 16 --iters 10 [--only block2_sepconv2]"""
import argparse
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet.ops import SeparableConvBN                      # noqa: E402
from xdet.runtime import DeviceTensor, Event, Stream, set_precision   # noqa: E402

SHAPES = [('block2_sepconv1', 237, 64, 128), ('block2_sepconv2', 237, 128, 128),
          ('block3_sepconv1', 119, 128, 256), ('block3_sepconv2', 119, 256, 256),
          ('mid_sepconv (use --split)', 30, 728, 728)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--split', action='store_true', help='the two-kernel form instead')
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    st = Stream()
    for name, hw, cin, cout in SHAPES:
        if a.only and a.only not in name:
            continue
        x = DeviceTensor.from_numpy(rng.standard_normal((a.batch, hw, hw, cin)).astype(np.float32))
        dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
        pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        set_precision('f16x3')
        op = SeparableConvBN(dk, pk, None, None, relu=False)
        set_precision('f32')
        op(x, relu_in=True, fused=not a.split, stream=st)
        st.synchronize()
        e0, e1 = Event(), Event()
        e0.record(st)
        for _ in range(a.iters):
            op(x, relu_in=True, fused=not a.split, stream=st)
        e1.record(st)
        st.synchronize()
        ms = e0.elapsed_ms(e1) / a.iters
        gb = a.batch * hw * hw * (cin + cout) * 4 / 1e9
        print('%-18s %7.3f ms  %6.2f TB/s (in+out once)  %6.1f TFLOP/s' %
              (name, ms, gb / ms, 2.0 * a.batch * hw * hw * cin * cout / ms / 1e9))


if __name__ == '__main__':
    main()
