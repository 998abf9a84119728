#!/usr/bin/env python
"""Writes tests/golden/tiny_bundle.{index,data-00000-of-00001}: a small TensorFlow-V2-format checkpoint (tensor
bundle) produced by xdet/tf_checkpoint.py's pure-Python writer from seeded values -- the fixture of
tests/test_tf_checkpoint.py.  37 variables (so the index SSTable has several data blocks and shared key
prefixes), float32 model variables under the reference's scope, an optimizer slot and an int64 global_step."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'x-detector_amd'))
from xdet import tf_checkpoint as T   # noqa: E402


def tensors():
    rng = np.random.default_rng(20260928)
    t = {}
    for i in range(1, 13):
        base = 'xception_lighthead/block%d_sepconv1' % i
        t[base + '/depthwise_kernel'] = rng.standard_normal((3, 3, 4, 1)).astype(np.float32)
        t[base + '/pointwise_kernel'] = rng.standard_normal((1, 1, 4, 6)).astype(np.float32)
        t[base + '_bn/gamma'] = rng.uniform(0.5, 1.5, 6).astype(np.float32)
    t['xception_lighthead/block1_sepconv1/pointwise_kernel/Momentum'] = np.zeros((1, 1, 4, 6), np.float32)
    t['global_step'] = np.asarray(122320, np.int64)
    return t


if __name__ == '__main__':
    T.write_checkpoint(os.path.join(HERE, 'tiny_bundle'), tensors(), block_entries=8)
    print(sorted(os.listdir(HERE)))
