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
