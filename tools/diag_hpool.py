import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/x-detector_amd')
import numpy as np
from xdet.ops import SeparableConvBN, max_pool_3x3_s2_same_add, separable_block_then_pool_add
from xdet.runtime import DeviceTensor, set_precision
for case in [(2, 237, 237, 128, 128, True), (1, 119, 119, 256, 256, True), (2, 29, 57, 96, 128, True)]:
    N, H, W, cin, cout, relu_in = case
    rng = np.random.default_rng(H * 13 + W + cin)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
    pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((N, -(-H // 2), -(-W // 2), cout)).astype(np.float32)
    set_precision('f16x3')
    op = SeparableConvBN(dk, pk, scale, shift, relu=False)
    set_precision('f32')
    xd, rd = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(res)
    for rep in range(3):
        whole = max_pool_3x3_s2_same_add(op(xd, relu_in=relu_in, fused=True), rd).numpy()
        split = separable_block_then_pool_add(op, xd, rd, relu_in=relu_in).numpy()
        bad = np.argwhere(whole != split)
        print(case, 'rep', rep, 'mismatches', len(bad), 'of', whole.size)
        if len(bad):
            print(' first', bad[:8].tolist())
            print(' n', np.unique(bad[:,0]), 'rows', np.unique(bad[:,1])[:20], 'cols', np.unique(bad[:,2])[:40], 'ch', np.unique(bad[:,3])[:40])
            b=bad[0]; print(whole[tuple(b)], split[tuple(b)])
