#!/usr/bin/env python
"""Which conv path disagrees with a float64 GEMM, and where?  (GPU box)"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet.ops import Conv2D                               # noqa: E402
from xdet.runtime import DeviceTensor, set_precision      # noqa: E402

N, H, W, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (2, 30, 30, 728, 728))]
rng = np.random.default_rng(3)
x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
k = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
sh = rng.standard_normal(cout).astype(np.float32)
res = rng.standard_normal((N, H, W, cout)).astype(np.float32)
set_precision('f16x3')
g = lambda xx: xx.reshape(-1, cin).astype(np.float64) @ k.reshape(cin, cout).astype(np.float64) * sc + sh
ref = g(x)
ref2 = np.maximum(g(np.maximum(x, 0)) + res.reshape(-1, cout), 0)


def report(name, y, r):
    y = y.reshape(-1, cout)
    bad = np.abs(y - r) > 1e-3 * max(1, np.abs(r).max())
    print('%-28s max err %.3g  bad %d of %d' % (name, np.abs(y - r).max(), bad.sum(), bad.size))
    if bad.any():
        rows = np.nonzero(bad.any(1))[0]
        cols = np.nonzero(bad.any(0))[0]
        print('   rows', len(rows), rows[:12], '...', rows[-4:], ' rows%128:', sorted(set((rows % 128).tolist()))[:40])
        print('   cols', len(cols), cols[:24], '...', cols[-4:])
        r0 = rows[0]
        print('   row', r0, 'got', y[r0, cols[:6]], 'want', r[r0, cols[:6]])


report('split', Conv2D(k, 1, 'SAME', 1, sc, sh)(DeviceTensor.from_numpy(x)).numpy(), ref)
report('split res relu', Conv2D(k, 1, 'SAME', 1, sc, sh, relu=True)(DeviceTensor.from_numpy(x), residual=DeviceTensor.from_numpy(res), relu_in=True).numpy(), ref2)
report('planes', Conv2D(k, 1, 'SAME', 1, sc, sh)(DeviceTensor.from_numpy(x), planes=True).numpy(), ref)
report('planes res relu', Conv2D(k, 1, 'SAME', 1, sc, sh, relu=True)(DeviceTensor.from_numpy(x), residual=DeviceTensor.from_numpy(res), relu_in=True, planes=True).numpy(), ref2)
