"""ResNet-50 v2 trunk (BASELINE config 2) on the GPU vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_resnet50_trunk_matches_oracle(oracle, precision):
    from xdet import weights as W
    from xdet.resnet import ResNet50Trunk
    from xdet.runtime import set_precision
    w = W.make_resnet50_weights(4321)
    imgs = W.synthetic_images(8, 480, seed=1)                 # BASELINE config 2: batch 8 of 480x480
    set_precision(precision)
    try:
        net = ResNet50Trunk(w, image_size=480, max_batch=8)
    finally:
        set_precision('f32')
    y = net.forward(imgs)
    ref = oracle.resnet50_trunk(np.transpose(imgs, (0, 2, 3, 1)), w)
    assert y.shape == ref.shape == (8, 15, 15, 2048)          # total stride 32 as written in the file
    err = float(np.abs(y - ref).max()) / max(1.0, float(np.abs(ref).max()))
    print('resnet50 trunk [%s], batch 8: max error relative to the output scale %.2e' % (precision, err))
    assert err <= 1e-4, err                                   # 50 stacked layers, same bar as the Xception features
    assert abs(net.flops_per_image() - 37.5e9) < 0.6e9


def test_resnet50_small_image_and_batch_tail(oracle):
    from xdet import weights as W
    from xdet.resnet import ResNet50Trunk
    w = W.make_resnet50_weights(4321)
    imgs = W.synthetic_images(3, 96, seed=2)
    net = ResNet50Trunk(w, image_size=96, max_batch=4)
    y = net.forward(imgs)                                      # N=3 < max_batch
    ref = oracle.resnet50_trunk(np.transpose(imgs, (0, 2, 3, 1)), w)
    assert np.abs(y - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_resnet50_graph_replay_is_the_eager_forward():
    """xdet_resnet_forward_graph: same bits as the eager launch sequence, also after new images were copied into the
    same device buffer (a replay reads the buffer, not a snapshot) and for a second batch size (its own graph)."""
    from xdet import weights as W
    from xdet.resnet import ResNet50Trunk
    from xdet.runtime import set_precision, to_host
    w = W.make_resnet50_weights(4321)
    set_precision('f16x3')
    try:
        net = ResNet50Trunk(w, image_size=160, max_batch=4)
    finally:
        set_precision('f32')
    for seed, n in [(3, 4), (4, 4), (5, 2), (6, 4)]:
        imgs = W.synthetic_images(n, 160, seed=seed)
        eager = net.forward(imgs)
        net.set_images(imgs)
        net.forward_device(n, use_graph=True)
        net.stream.synchronize()
        replay = to_host(net._out.ptr, (n,) + net.out_shape, np.float32)
        assert np.array_equal(eager, replay), (seed, n)


def test_trunk_pre_activations_beyond_the_f16_range(oracle):
    """VERDICT r3 missing #5: the trunk's split-precision planes carry an activation pre-scale too (xdet_resnet_calibrate).
    A pre-activation BN with gamma / beta x 2^17 in front of convs with kernels x 2^-17 is the same function (powers of
    two), but relu(bn(x)) is ~1e5-1e6: the f16 hi plane overflows.  Two kinds of producer are hit: the stand-alone
    bn_relu pass of the first block (its multiplier) and a closing conv's epilogue that emits the NEXT block's
    pre-activation (its folded BN carries 2^-e).  Uncalibrated the output is non-finite; calibrated it matches the oracle
    and the untouched net."""
    from xdet import weights as W
    from xdet.resnet import ResNet50Trunk
    from xdet.runtime import set_precision
    w = W.make_resnet50_weights(4321)
    hot = dict(w)
    k = np.float32(2.0 ** 17)
    # block 0 of stage 1: pre = relu(bn0(x)) feeds the projection (conv2d_1) and conv1 (conv2d_2)
    for n in ('gamma', 'beta'):
        hot['batch_normalization/' + n] = w['batch_normalization/' + n] * k
    for c in ('conv2d_1', 'conv2d_2'):
        hot[c + '/kernel'] = w[c + '/kernel'] / k
    # block 1: its pre-activation BN (batch_normalization_3) is folded into block 0's closing conv; conv1 = conv2d_5
    for n in ('gamma', 'beta'):
        hot['batch_normalization_3/' + n] = w['batch_normalization_3/' + n] * k
    hot['conv2d_5/kernel'] = w['conv2d_5/kernel'] / k
    imgs = W.synthetic_images(2, 160, seed=9)
    ref = oracle.resnet50_trunk(np.transpose(imgs, (0, 2, 3, 1)), hot)
    set_precision('f16x3')
    try:
        net = ResNet50Trunk(hot, image_size=160, max_batch=2)
        cool = ResNet50Trunk(w, image_size=160, max_batch=2)
    finally:
        set_precision('f32')
    assert not np.isfinite(net.forward(imgs)).all()                 # uncalibrated: overflow
    scaled = net.calibrate(imgs[:1])
    print('calibrated:', scaled)
    names = ' | '.join(scaled)
    assert 'batch_normalization (pre-activation planes)' in names and 'planes of the output' in names
    assert 1 <= len(scaled) <= 4 and all(e > 0 for e in scaled.values())
    y = net.forward(imgs)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(y).all() and float(np.abs(y - ref).max()) <= 1e-4 * scale
    assert float(np.abs(y - cool.forward(imgs)).max()) <= 2e-5 * scale      # function-preserving rescale
    assert cool.calibrate(imgs) == {}                                        # a tame net is left alone


def test_handles_are_checked():
    """a trunk handle given to a light-head entry point (and vice versa) is an InvalidArgumentError, not
    undefined behaviour."""
    import ctypes
    from xdet import weights as W
    from xdet._lib import lib, check, InvalidArgumentError
    from xdet.resnet import ResNet50Trunk
    net = ResNet50Trunk(W.make_resnet50_weights(4321), image_size=96, max_batch=1)
    k = ctypes.c_int()
    with pytest.raises(InvalidArgumentError):
        check(lib().xdet_net_calibrate(net.handle, net._images.ptr, 1, ctypes.byref(k), net.stream.handle))
    with pytest.raises(InvalidArgumentError):
        check(lib().xdet_net_forward(net.handle, net._images.ptr, 1, None, None, net._out.ptr, net._out.ptr, 0, net.stream.handle))
