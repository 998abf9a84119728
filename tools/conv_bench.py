#!/usr/bin/env python
"""Micro-benchmark of the conv/dense MFMA kernel on the network's own GEMM shapes (GPU box only).
    python tools/conv_bench.py --precision f16x3 --batch 8"""
import argparse
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet.ops import Conv2D                               # noqa: E402
from xdet.runtime import DeviceTensor, Event, Stream, set_precision   # noqa: E402

SHAPES = [  # name, H, W, cin, cout, kh, kw, stride, pad
    ('block1_conv2 3x3 32->64 @239', 239, 239, 32, 64, 3, 3, 1, 'VALID'),
    ('block2 pw 128->128 @237', 237, 237, 128, 128, 1, 1, 1, 'SAME'),
    ('block3 pw 256->256 @119', 119, 119, 256, 256, 1, 1, 1, 'SAME'),
    ('block4 pw 728->728 @60', 60, 60, 728, 728, 1, 1, 1, 'SAME'),
    ('mid pw 728->728 @30', 30, 30, 728, 728, 1, 1, 1, 'SAME'),
    ('b14 pw 1536->2048 @30', 30, 30, 1536, 2048, 1, 1, 1, 'SAME'),
    ('rpn 3x3 728->512 @30', 30, 30, 728, 512, 3, 3, 1, 'SAME'),
    ('lsep 15x1 2048->512 @30', 30, 30, 2048, 512, 15, 1, 1, 'SAME'),
    ('lsep 1x15 512->490 @30', 30, 30, 512, 490, 1, 15, 1, 'SAME'),
    ('fc 490->2048 (R=300)', 300, 1, 490, 2048, 1, 1, 1, 'VALID'),
    ('head 2048->25 (R=300)', 300, 1, 2048, 25, 1, 1, 1, 'VALID'),
]


RESNET_SHAPES = [  # ResNet-50 v2 stages 3-4 (net/resnet_v2.py:142-184) + the detector's small-M layers
    ('r3 1x1 1024->256 @30', 30, 30, 1024, 256, 1, 1, 1, 'SAME'),
    ('r3 3x3 256->256 @30', 30, 30, 256, 256, 3, 3, 1, 'SAME'),
    ('r3 1x1 256->1024 @30', 30, 30, 256, 1024, 1, 1, 1, 'SAME'),
    ('r4 1x1 2048->512 @15', 15, 15, 2048, 512, 1, 1, 1, 'SAME'),
    ('r4 3x3 512->512 @15', 15, 15, 512, 512, 3, 3, 1, 'SAME'),
    ('r4 1x1 512->2048 @15', 15, 15, 512, 2048, 1, 1, 1, 'SAME'),
    ('r2 1x1 512->128 @60', 60, 60, 512, 128, 1, 1, 1, 'SAME'),
    ('r2 3x3 128->128 @60', 60, 60, 128, 128, 3, 3, 1, 'SAME'),
    ('r2 1x1 128->512 @60', 60, 60, 128, 512, 1, 1, 1, 'SAME'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--resnet', action='store_true', help='the ResNet-50 stage 2-4 shapes instead of the detector list')
    ap.add_argument('--ksplit', default='', help='comma list of split factors to time with --planes on the split-K kernel '
                                                 '(0 = the plain kernels); each as parallel-ranges / one-workgroup-per-tile')
    ap.add_argument('--precision', default='f32')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--planes', action='store_true', help='A operand pre-split into f16 planes (LDS-DMA kernel)')
    ap.add_argument('--x8', action='store_true', help='with --planes: the x8 form of the planes (fp8 cross terms; 1x1 stride-1 layers)')
    ap.add_argument('--custom', action='append', default=[],
                    help='extra 1x1 GEMM shape "H,W,cin,cout" (replaces the built-in list); repeatable')
    a = ap.parse_args()
    shapes = RESNET_SHAPES if a.resnet else SHAPES
    if a.custom:
        shapes = []
        for c in a.custom:
            H, W, cin, cout = [int(v) for v in c.split(',')]
            shapes.append(('gemm M=%d*B K=%d N=%d' % (H * W, cin, cout), H, W, cin, cout, 1, 1, 1, 'SAME'))
    set_precision(a.precision)
    rng = np.random.default_rng(0)
    st = Stream()
    tot_f = tot_t = 0
    for name, H, W, cin, cout, kh, kw, s, pad in shapes:
        if a.only and a.only not in name:
            continue
        x = DeviceTensor.from_numpy(rng.standard_normal((a.batch, H, W, cin)).astype(np.float32))
        k = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
        L = Conv2D(k, s, pad)
        y = L(x, stream=st)
        st.synchronize()
        e0, e1 = Event(), Event()
        from xdet._lib import lib, check
        if a.planes:
            from xdet.runtime import DeviceBuffer
            n = -(-a.batch * H * W // 16) * 16 * x.ld
            hi, lo = DeviceBuffer(n * 2 + 512, zero=True), DeviceBuffer(n * 2 + 512, zero=True)
            if a.x8:
                check(lib().xdet_split_f32_x8(x.ptr, hi.ptr, lo.ptr, a.batch * H * W, x.ld, 0, -5, st.handle))
                check(lib().xdet_conv_forward_planes_x8(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, -5, st.handle))
            else:
                check(lib().xdet_split_f32(x.ptr, hi.ptr, lo.ptr, a.batch * H * W, x.ld, 0, st.handle))
                check(lib().xdet_conv_forward_planes(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, st.handle))
            st.synchronize()
        if a.planes and a.ksplit:
            res = []
            for S in [int(v) for v in a.ksplit.split(',')]:
                for mode in ((0,) if S == 0 else (1, 2) if S > 1 else (2,)):
                    L.set_ksplit(S, mode, 512)
                    for _ in range(3):
                        check(lib().xdet_conv_forward_planes(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, st.handle))
                    st.synchronize()
                    e0.record(st)
                    for _ in range(a.iters):
                        check(lib().xdet_conv_forward_planes(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, st.handle))
                    e1.record(st)
                    st.synchronize()
                    res.append('S=%d%s %.1fus' % (S, {0: '', 1: 'p', 2: 's'}[mode] if S else '(plain)', e0.elapsed_ms(e1) / a.iters * 1e3))
            print('%-28s B=%-3d %s' % (name, a.batch, '  '.join(res)))
            continue
        e0.record(st)
        for _ in range(a.iters):
            if a.planes and a.x8:
                check(lib().xdet_conv_forward_planes_x8(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, -5, st.handle))
            elif a.planes:
                check(lib().xdet_conv_forward_planes(L.handle, hi.ptr, lo.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, st.handle))
            else:
                check(lib().xdet_conv_forward(L.handle, x.ptr, a.batch, H, W, x.ld, y.ptr, y.ld, None, 0, st.handle))
        e1.record(st)
        st.synchronize()
        ms = e0.elapsed_ms(e1) / a.iters
        fl = 2.0 * a.batch * y.shape[1] * y.shape[2] * cin * cout * kh * kw
        tot_f += fl
        tot_t += ms
        print('%-34s %8.3f ms  %7.1f TFLOP/s' % (name, ms, fl / ms / 1e9))
    print('%-34s %8.3f ms  %7.1f TFLOP/s' % ('TOTAL', tot_t, tot_f / tot_t / 1e9))


if __name__ == '__main__':
    main()
