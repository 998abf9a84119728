#!/usr/bin/env python
"""The x8 form of a pointwise layer (fp8 cross terms) against a float64 GEMM and against f16x3, op level (GPU box).
    python tools/x8_layer_check.py [N H W cin cout]"""
import os
import sys
import math

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
import numpy as np                                        # noqa: E402
from xdet.ops import Conv2D                               # noqa: E402
from xdet.runtime import DeviceTensor, set_precision      # noqa: E402

N, H, W, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 30, 30, 728, 728))]
rng = np.random.default_rng(5)
x = (rng.standard_normal((N, H, W, cin)) * np.exp(rng.normal(0, 1.0, (1, 1, 1, cin)))).astype(np.float32)
k = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
sh = rng.standard_normal(cout).astype(np.float32)
res = rng.standard_normal((N, H, W, cout)).astype(np.float32)
ref = x.reshape(-1, cin).astype(np.float64) @ k.reshape(cin, cout).astype(np.float64) * sc + sh + res.reshape(-1, cout)
e8 = int(math.ceil(math.log2(float(np.abs(x).max()) / 256.0)))
print('max|x| %.3g -> x8_exp %d' % (np.abs(x).max(), e8))
set_precision('f16x3')
L = Conv2D(k, 1, 'SAME', 1, sc, sh)
xd, rd = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(res)
for name, kw in (('f16x3', {}), ('x8', {'x8_exp': e8}), ('x8, exponent 2 too large', {'x8_exp': e8 + 2}), ('x8, exponent 2 too small (saturating)', {'x8_exp': e8 - 2})):
    y = L(xd, residual=rd, planes=True, **kw).numpy().reshape(-1, cout).astype(np.float64)
    d = y - ref
    print('%-40s max err / max|ref| %.2e   rms relative %.2e' % (name, np.abs(d).max() / np.abs(ref).max(), np.sqrt((d ** 2).mean() / (ref ** 2).mean())))
# the same rows in a small and a large batch: the arithmetic does not depend on the tile shape
a = L(xd, residual=rd, planes=True, x8_exp=e8).numpy()
b = L(DeviceTensor.from_numpy(x[3:5]), residual=DeviceTensor.from_numpy(res[3:5]), planes=True, x8_exp=e8).numpy()
print('batch-invariant bits:', bool(np.array_equal(a[3:5], b)))
