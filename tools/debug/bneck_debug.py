"""Diagnosis of csrc/resnet_bneck.hip: with XDET_BNECK_DEBUG=1|2 the kernel writes mid1 / mid2 (hi + lo) into channels 0..63 of
`out`; compared here with the first / second layer of the three-launch form.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'x-detector_amd'))
from xdet import ops                                     # noqa: E402
from xdet._lib import lib, check                         # noqa: E402
from xdet.runtime import DeviceBuffer, DeviceTensor, set_precision, synchronize, to_device   # noqa: E402

N, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 8, 30))]
rng = np.random.RandomState(5)
cin, cmid = 256, 64
x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
ps = np.full(cin, 0.5, np.float32)
ph = np.full(cin, 0.1, np.float32)
pre = np.maximum(x * ps + ph, 0).astype(np.float32)
wa = (rng.standard_normal((1, 1, cin, cmid)) / np.sqrt(cin)).astype(np.float32)
wb = (rng.standard_normal((3, 3, cmid, cmid)) / np.sqrt(9 * cmid)).astype(np.float32)
wc = (rng.standard_normal((1, 1, cmid, cin)) / np.sqrt(cmid)).astype(np.float32)
one, zero = np.ones(cmid, np.float32), np.zeros(cmid, np.float32)
set_precision('f16x3')
A = ops.Conv2D(wa, scale=one, shift=zero + 0.05, relu=True)
B = ops.Conv2D(wb, scale=one, shift=zero - 0.02, relu=True)
C = ops.Conv2D(wc)
set_precision('f32')
dx, dpre = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(pre)
y1 = A(dpre, planes=True)
y2 = B(y1, planes=True)
ref = C(y2, planes=True, residual=dx).numpy()
dps, dph = to_device(ps), to_device(ph)
for mode, want in ((1, y1.numpy()), (2, y2.numpy()), (0, ref)):
    os.environ['XDET_BNECK_DEBUG'] = str(mode)
    out = DeviceTensor.empty((N, H, W, cin))
    check(lib().xdet_resnet_bneck_forward(A.handle, B.handle, C.handle, dps.ptr, dph.ptr, dx.ptr, N, H, W, out.ptr, None, None,
                                          None, None, None))
    synchronize()
    got = out.numpy()
    if mode:
        got = got[..., :cmid]
    d = np.abs(got - want)
    bad = ~(d <= 1e-5 * max(1.0, float(np.abs(want).max())))
    print('mode %d: max |d| %g, bad %d of %d, non-finite %d' % (mode, np.nanmax(d) if np.isfinite(d).any() else float('nan'), bad.sum(), bad.size, (~np.isfinite(got)).sum()))
    if bad.any():
        idx = np.argwhere(bad)
        for ax, name in enumerate('nyxc'):
            u, c = np.unique(idx[:, ax], return_counts=True)
            print('   axis %s: %s' % (name, dict(zip(u.tolist()[:64], c.tolist()[:64]))))
        i = tuple(idx[0])
        print('   first bad', i, 'got', got[i], 'want', want[i])
        print('   got[0,0,0,:8]', got[0, 0, 0, :8], 'want', want[0, 0, 0, :8])
        print('   got[0,1,3,:8]', got[0, min(1, H - 1), 3, :8], 'want', want[0, min(1, H - 1), 3, :8])
