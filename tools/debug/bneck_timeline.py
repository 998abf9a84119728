"""s_memtime stamps of workgroup 0 of csrc/resnet_bneck.hip (XDET_BNECK_DEBUG=9): per tile, the clocks between phase boundaries.
  python tools/debug/bneck_timeline.py [N H W]        (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'x-detector_amd'))
from xdet import ops                                     # noqa: E402
from xdet._lib import lib, check                         # noqa: E402
from xdet.runtime import DeviceTensor, set_precision, synchronize, to_device, to_host   # noqa: E402

N, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 120, 120))]
rng = np.random.RandomState(5)
cin, cmid = 256, 64
x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
wa = (rng.standard_normal((1, 1, cin, cmid)) / np.sqrt(cin)).astype(np.float32)
wb = (rng.standard_normal((3, 3, cmid, cmid)) / np.sqrt(9 * cmid)).astype(np.float32)
wc = (rng.standard_normal((1, 1, cmid, cin)) / np.sqrt(cmid)).astype(np.float32)
one, zero = np.ones(cmid, np.float32), np.zeros(cmid, np.float32)
set_precision('f16x3')
A = ops.Conv2D(wa, scale=one, shift=zero + 0.05, relu=True)
B = ops.Conv2D(wb, scale=one, shift=zero - 0.02, relu=True)
C = ops.Conv2D(wc)
set_precision('f32')
dx = DeviceTensor.from_numpy(x)
dps, dph = to_device(np.ones(cin, np.float32)), to_device(np.zeros(cin, np.float32))
out = DeviceTensor.empty((N, H, W, cin))
os.environ['XDET_BNECK_DEBUG'] = '0'
for _ in range(3):
    check(lib().xdet_resnet_bneck_forward(A.handle, B.handle, C.handle, dps.ptr, dph.ptr, dx.ptr, N, H, W, out.ptr, None, None, None, None, None))
synchronize()
os.environ['XDET_BNECK_DEBUG'] = '9'
check(lib().xdet_resnet_bneck_forward(A.handle, B.handle, C.handle, dps.ptr, dph.ptr, dx.ptr, N, H, W, out.ptr, None, None, None, None, None))
synchronize()
st = to_host(out.ptr, (64,), np.uint64).astype(np.int64)
names = ['wait all landed', 'transform0 + phase 1', 'epilogue 1', 'phase 2', 'epilogue 2 + barrier', 'phase 3', 'prologue + epilogue 3', '(loop)']
per = 8
for t in range(len(st) // per):
    row = st[t * per:(t + 1) * per]
    if row[0] == 0 or (t and row[0] < st[(t - 1) * per]) or np.any(np.diff(row) < 0) or row[-1] - row[0] > 10 ** 7:
        break                                        # (slots behind the workgroup's last tile hold whatever the output row held)
    nxt = st[(t + 1) * per] if (t + 1) * per < len(st) and 0 < st[(t + 1) * per] - row[-1] < 10 ** 6 else row[-1]
    d = list(np.diff(row)) + [nxt - row[-1]]
    print('tile %d: total %6d clk | ' % (t, nxt - row[0]) + ', '.join('%s %d' % (n, v) for n, v in zip(names, d)))
