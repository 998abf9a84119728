"""Produce the BN moving-statistics fixtures used by the synthetic weights
(x-detector_amd/xdet/data/bn_calib_*.npz).  Run once in the build container:

    python tests/golden/make_calibration.py

For every BN layer, moving_mean / moving_variance are set to the per-channel statistics
of its input on one seeded synthetic batch, computed layer by layer with the CPU oracle
(so each later layer sees already-normalised activations).  SURVEY.md 8d.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'x-detector_amd'))

from oracle import lighthead_oracle as O      # noqa: E402
from xdet import weights as W                  # noqa: E402


def lighthead(seed=1234, n=1, size=480):
    w = W.make_lighthead_weights(seed, calibrated=False)
    x = np.transpose(W.synthetic_images(n, size, seed=99), (0, 2, 3, 1))

    def fwd(x, w, taps):
        mid, out = O.xception_body(x, w, taps)
        O.large_sep_kernel(out, w, taps=taps)

    def bn_of_tap(t):
        if t == 'large_sep_feature':
            return 'large_sep_feature/batch_normalization'
        m = {'conv2d_1': 'batch_normalization_1', 'conv2d_2': 'batch_normalization_2',
             'conv2d_3': 'batch_normalization_3', 'conv2d_4': 'batch_normalization_4'}
        return m.get(t, t + '_bn')

    return calibrate_fast(fwd, w, x, bn_of_tap)


def calibrate_fast(fwd, w, x, bn_of_tap):
    """Same result as `calibrate` but in ONE pass: the oracle calls taps[name]=y right
    before the matching batch_norm, so a dict subclass can install the statistics at
    that moment and the very same pass continues with the calibrated layer."""
    class Taps(dict):
        def __setitem__(self, t, y):
            bn = bn_of_tap(t)
            v = y.reshape(-1, y.shape[-1]).astype(np.float64)
            w[bn + '/moving_mean'] = v.mean(0).astype(np.float32)
            w[bn + '/moving_variance'] = np.maximum(v.var(0), 1e-12).astype(np.float32)
            dict.__setitem__(self, t, None)
    fwd(x, w, Taps())
    return {k: w[k] for k in w if k.endswith('moving_mean') or k.endswith('moving_variance')}


def resnet(seed=4321, n=1, size=480):
    w = W.make_resnet50_weights(seed, calibrated=False)
    x = np.transpose(W.synthetic_images(n, size, seed=98), (0, 2, 3, 1))
    return calibrate_fast(lambda x, w, taps: O.resnet50_trunk(x, w, taps), w, x, lambda t: t)


if __name__ == '__main__':
    out = os.path.join(ROOT, 'x-detector_amd', 'xdet', 'data')
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, 'bn_calib_lighthead_seed1234.npz'), **lighthead())
    # a second, independent weight set (other seed, other score gains in the tests) for the 1e-3 claim
    np.savez_compressed(os.path.join(out, 'bn_calib_lighthead_seed777.npz'), **lighthead(seed=777))
    np.savez_compressed(os.path.join(out, 'bn_calib_resnet50_seed4321.npz'), **resnet())
    print('wrote', os.listdir(out))
