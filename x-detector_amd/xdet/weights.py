"""Variable tables + seeded synthetic weights for the Light-Head R-CNN forward path.

The tables below are the TF variable names/shapes the reference graph creates
(scope `xception_lighthead/` stripped):
  * backbone            net/xception_body.py:243-376   (kernel HWIO, depthwise [3,3,C,1])
  * RPN head            net/xception_body.py:381-400   (scope `rpn_head`)
  * large separable     net/xception_body.py:450-475   (scope `large_sep_feature`)
  * light head          net/xception_body.py:540-558   (scope `final_head`, dense kernel [in,out])
  * ResNet-50 v2 trunk  net/resnet_v2.py:311-345       (BASELINE config 2)

There are no trained checkpoints in the reference, so tests and bench use seeded
random-init weights of that architecture: glorot-normal kernels as the reference
initialisers (xception_body.py:25-26), small biases, BN gamma~U(.8,1.2) beta~N(0,.1),
and BN moving statistics taken from `data/bn_calib_*.npz` (per-layer statistics of one
seeded batch, produced by tests/golden/make_calibration.py) so activations stay O(1)
through the 40 stacked layers.
"""
import os
import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')

XCEPTION_BN_EPS = 1e-4      # net/xception_body.py:20
RESNET_BN_EPS = 1e-5        # net/resnet_v2.py:37 (also large_sep BN, xception_body.py:475)


def xception_conv_table():
    """[(name, kind, shape)] in graph order; kind in conv|sep|bn."""
    t = []
    t.append(('block1_conv1', 'conv', (3, 3, 3, 32)))
    t.append(('block1_conv1_bn', 'bn', (32,)))
    t.append(('block1_conv2', 'conv', (3, 3, 32, 64)))
    t.append(('block1_conv2_bn', 'bn', (64,)))
    t.append(('conv2d_1', 'conv', (1, 1, 64, 128)))
    t.append(('batch_normalization_1', 'bn', (128,)))
    for name, cin, cout in (('block2_sepconv1', 64, 128), ('block2_sepconv2', 128, 128)):
        t.append((name, 'sep', (cin, cout)))
        t.append((name + '_bn', 'bn', (cout,)))
    t.append(('conv2d_2', 'conv', (1, 1, 128, 256)))
    t.append(('batch_normalization_2', 'bn', (256,)))
    for name, cin, cout in (('block3_sepconv1', 128, 256), ('block3_sepconv2', 256, 256)):
        t.append((name, 'sep', (cin, cout)))
        t.append((name + '_bn', 'bn', (cout,)))
    t.append(('conv2d_3', 'conv', (1, 1, 256, 728)))
    t.append(('batch_normalization_3', 'bn', (728,)))
    for name, cin, cout in (('block4_sepconv1', 256, 728), ('block4_sepconv2', 728, 728)):
        t.append((name, 'sep', (cin, cout)))
        t.append((name + '_bn', 'bn', (cout,)))
    for b in range(5, 13):
        for s in (1, 2, 3):
            name = 'block%d_sepconv%d' % (b, s)
            t.append((name, 'sep', (728, 728)))
            t.append((name + '_bn', 'bn', (728,)))
    t.append(('conv2d_4', 'conv', (1, 1, 728, 1024)))
    t.append(('batch_normalization_4', 'bn', (1024,)))
    for name, cin, cout in (('block13_sepconv1', 728, 728), ('block13_sepconv2', 728, 1024),
                            ('block14_sepconv1', 1024, 1536), ('block14_sepconv2', 1536, 2048)):
        t.append((name, 'sep', (cin, cout)))
        t.append((name + '_bn', 'bn', (cout,)))
    return t


def lighthead_tables(num_anchors=22, num_classes=21, grid=7, bank=10, depth_mid=256):
    depth_out = bank * grid * grid
    t = xception_conv_table()
    t.append(('rpn_head/conv2d', 'convb', (3, 3, 728, 512)))
    t.append(('rpn_head/conv2d_1', 'convb', (1, 1, 512, 2 * num_anchors)))
    t.append(('rpn_head/conv2d_2', 'convb', (1, 1, 512, 4 * num_anchors)))
    for br in ('Branch_0', 'Branch_1'):
        t.append(('large_sep_feature/%s/conv2d' % br, 'convb', (15, 1, 2048, depth_mid)))
        t.append(('large_sep_feature/%s/conv2d_1' % br, 'convb', (1, 15, depth_mid, depth_out)))
    t.append(('large_sep_feature/batch_normalization', 'bn', (depth_out,)))
    t.append(('final_head/subnet_fc', 'dense', (depth_out, 2048)))
    t.append(('final_head/fc_cls', 'dense', (2048, num_classes)))
    t.append(('final_head/fc_loc', 'dense', (2048, 4)))
    return t


def resnet50_table():
    """ResNet-50 v2 trunk (net/resnet_v2.py:311-345, bottleneck :142-184). Names follow
    tf.layers auto-numbering in graph-construction order: conv2d, conv2d_1, ...;
    batch_normalization, batch_normalization_1, ..."""
    t = []
    ci = [0]
    bi = [0]

    def conv(k, cin, cout):
        name = 'conv2d' if ci[0] == 0 else 'conv2d_%d' % ci[0]
        ci[0] += 1
        t.append((name, 'conv', (k, k, cin, cout)))
        return name

    def bn(c):
        name = 'batch_normalization' if bi[0] == 0 else 'batch_normalization_%d' % bi[0]
        bi[0] += 1
        t.append((name, 'bn', (c,)))
        return name

    conv(7, 3, 64)
    cin = 64
    for filters, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            bn(cin)
            if b == 0:
                conv(1, cin, 4 * filters)          # projection shortcut
            conv(1, cin, filters)
            bn(filters)
            conv(3, filters, filters)
            bn(filters)
            conv(1, filters, 4 * filters)
            cin = 4 * filters
    bn(cin)
    return t


def _glorot_std(shape, kind):
    if kind == 'dense':
        fan_in, fan_out = shape
    else:
        rf = shape[0] * shape[1]
        fan_in, fan_out = shape[2] * rf, shape[3] * rf
    return float(np.sqrt(2.0 / (fan_in + fan_out)))


def _fill(table, seed, calib_file, bn_eps_note=None):
    rng = np.random.default_rng(seed)
    w = {}
    for name, kind, shape in table:
        if kind in ('conv', 'convb'):
            w[name + '/kernel'] = (rng.standard_normal(shape, dtype=np.float32) * _glorot_std(shape, kind))
            if kind == 'convb':
                w[name + '/bias'] = rng.standard_normal(shape[3], dtype=np.float32) * np.float32(0.01)
        elif kind == 'sep':
            cin, cout = shape
            dshape = (3, 3, cin, 1)
            # a plain glorot depthwise kernel is ~1/sqrt(C); BN re-normalises anyway
            w[name + '/depthwise_kernel'] = rng.standard_normal(dshape, dtype=np.float32) * _glorot_std(dshape, 'conv')
            pshape = (1, 1, cin, cout)
            w[name + '/pointwise_kernel'] = rng.standard_normal(pshape, dtype=np.float32) * _glorot_std(pshape, 'conv')
        elif kind == 'dense':
            w[name + '/kernel'] = rng.standard_normal(shape, dtype=np.float32) * _glorot_std(shape, 'dense')
            w[name + '/bias'] = rng.standard_normal(shape[1], dtype=np.float32) * np.float32(0.01)
        elif kind == 'bn':
            c = shape[0]
            w[name + '/gamma'] = rng.uniform(0.8, 1.2, c).astype(np.float32)
            w[name + '/beta'] = (rng.standard_normal(c, dtype=np.float32) * np.float32(0.1))
            w[name + '/moving_mean'] = np.zeros(c, np.float32)
            w[name + '/moving_variance'] = np.ones(c, np.float32)
        else:
            raise ValueError(kind)
    if calib_file is not None and os.path.exists(calib_file):
        cal = np.load(calib_file)
        for k in cal.files:
            if k in w:
                w[k] = cal[k].astype(np.float32)
    return w


# gains applied to the last RPN / head layers so that the synthetic scores spread like a
# trained detector's: part of the committed weight recipe, not tuning knobs.
SYNTH_GAINS = {
    'rpn_head/conv2d_1/kernel': 0.5,
    'rpn_head/conv2d_2/kernel': 0.25,
    'final_head/fc_cls/kernel': 6.0,
    'final_head/fc_loc/kernel': 0.5,
}


def make_lighthead_weights(seed=1234, calibrated=True, gains=None, **kw):
    """seeded synthetic weights; `gains` overrides SYNTH_GAINS (tests use a second seed with other gains so that the
    parity claim does not rest on one hand-picked score spread)"""
    calib = os.path.join(_DATA_DIR, 'bn_calib_lighthead_seed%d.npz' % seed) if calibrated else None
    if calibrated and not os.path.exists(calib):
        raise FileNotFoundError('no BN calibration for seed %d (tests/golden/make_calibration.py)' % seed)
    w = _fill(lighthead_tables(**kw), seed, calib)
    for k, g in (SYNTH_GAINS if gains is None else gains).items():
        w[k] = (w[k] * np.float32(g)).astype(np.float32)
    return w


def make_resnet50_weights(seed=4321, calibrated=True):
    calib = os.path.join(_DATA_DIR, 'bn_calib_resnet50_seed%d.npz' % seed) if calibrated else None
    return _fill(resnet50_table(), seed, calib)


def synthetic_images(n, size=480, seed=0):
    """float32 [n,3,size,size] iid U(-1,1): the whitened range of
    preprocessing/common_preprocessing.py:391-392 is about [-0.97, 1.18]."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, (n, 3, size, size)).astype(np.float32)


# ---- F3: checkpoint / variable-name import ------------------------------------------------

def lighthead_variable_shapes(**kw):
    """{TF variable name: shape} of the eval graph (scope prefix stripped) -- what
    LightHeadDetector / xdet_net_set_weight expect."""
    out = {}
    for name, kind, shape in lighthead_tables(**kw):
        if kind in ('conv', 'convb'):
            out[name + '/kernel'] = tuple(shape)
            if kind == 'convb':
                out[name + '/bias'] = (shape[3],)
        elif kind == 'sep':
            out[name + '/depthwise_kernel'] = (3, 3, shape[0], 1)
            out[name + '/pointwise_kernel'] = (1, 1, shape[0], shape[1])
        elif kind == 'dense':
            out[name + '/kernel'] = tuple(shape)
            out[name + '/bias'] = (shape[1],)
        else:
            for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                out[name + '/' + v] = (shape[0],)
    return out


def load_weights_npz(path, model_scope='xception_lighthead', **kw):
    """Load a `{tf variable name: array}` archive dumped from the reference's checkpoint
    (`{v.name: sess.run(v) for v in tf.global_variables()}`; scope `--model_scope`,
    light_head_rfcn_eval.py:130) into the dict the detector takes: strips `<scope>/` and `:0`,
    ignores optimizer / global-step variables (the `ignore_missing_vars` spirit of
    utility/train_helper.py:5-72), and checks names and shapes against the graph's table."""
    want = lighthead_variable_shapes(**kw)
    arc = np.load(path)
    got = {}
    for k in arc.files:
        name = k[:-2] if k.endswith(':0') else k
        if model_scope and name.startswith(model_scope + '/'):
            name = name[len(model_scope) + 1:]
        if name in want:
            a = np.asarray(arc[k], np.float32)
            if tuple(a.shape) != want[name]:
                raise ValueError('variable %s has shape %s, expected %s' % (name, a.shape, want[name]))
            got[name] = a
    missing = sorted(set(want) - set(got))
    if missing:
        raise KeyError('checkpoint is missing %d variables, e.g. %s' % (len(missing), missing[:3]))
    return got


def load_weights_tf_checkpoint(prefix, model_scope='xception_lighthead', verify_crc=False, **kw):
    """Load the detector's weights straight from a TensorFlow V2 checkpoint (`<prefix>.index` +
    `<prefix>.data-0000x-of-0000y`, e.g. the published `model.ckpt-122320`; light_head_rfcn_eval.py:499,
    light_head_simple_demo.py:195) -- no TensorFlow needed (xdet/tf_checkpoint.py parses the bundle).  Same rules as
    load_weights_npz: `<model_scope>/` is stripped, optimizer slots / global_step are ignored
    (utility/train_helper.py:74-94 restores only the model variables), names and shapes are checked."""
    from .tf_checkpoint import CheckpointReader
    want = lighthead_variable_shapes(**kw)
    rd = CheckpointReader(prefix)
    got = {}
    for full in rd.entries:
        name = full
        if model_scope and name.startswith(model_scope + '/'):
            name = name[len(model_scope) + 1:]
        if name in want:
            a = rd.get_tensor(full, verify_crc=verify_crc)
            if a.dtype != np.float32:
                raise ValueError('variable %s is %s, expected float32' % (full, a.dtype))
            if tuple(a.shape) != want[name]:
                raise ValueError('variable %s has shape %s, expected %s' % (name, a.shape, want[name]))
            got[name] = a
    missing = sorted(set(want) - set(got))
    if missing:
        raise KeyError('checkpoint is missing %d variables, e.g. %s' % (len(missing), missing[:3]))
    return got


def save_weights_tf_checkpoint(prefix, weights, model_scope='xception_lighthead', global_step=None):
    from .tf_checkpoint import write_checkpoint
    t = {'%s/%s' % (model_scope, k): np.asarray(v, np.float32) for k, v in weights.items()}
    if global_step is not None:
        t['global_step'] = np.asarray(global_step, np.int64)
    return write_checkpoint(prefix, t)


def save_weights_npz(path, weights, model_scope='xception_lighthead'):
    np.savez(path, **{'%s/%s:0' % (model_scope, k): v for k, v in weights.items()})
