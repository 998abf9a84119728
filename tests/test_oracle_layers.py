"""The oracle's dense arithmetic against torch-CPU as an independent conv/pool/BN implementation
(torch is not the reference and never ships), and the TF padding rules fixed by decision."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
F = torch.nn.functional


def t_nchw(x):
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def from_t(y):
    return y.numpy().transpose(0, 2, 3, 1)


def test_same_pad_rule(oracle):
    # (n, k, s, d) -> (before, after, out): the extra pixel goes to the bottom/right
    assert oracle.same_pad(60, 3, 2) == (0, 1, 30)
    assert oracle.same_pad(237, 3, 2) == (1, 1, 119)
    assert oracle.same_pad(119, 3, 2) == (1, 1, 60)
    assert oracle.same_pad(237, 1, 2) == (0, 0, 119)
    assert oracle.same_pad(30, 3, 1, 2) == (2, 2, 30)
    assert oracle.same_pad(30, 15, 1) == (7, 7, 30)
    assert oracle.same_pad(100, 3, 2) == (0, 1, 50)


@pytest.mark.parametrize('H,cin,cout,k,stride,pad,dil', [
    (31, 8, 12, 3, 1, 'SAME', 1), (30, 8, 12, 3, 2, 'SAME', 1), (31, 8, 12, 3, 2, 'VALID', 1),
    (30, 6, 5, 1, 2, 'SAME', 1), (20, 4, 7, 3, 1, 'SAME', 2), (33, 3, 8, 7, 2, ((3, 3), (3, 3)), 1)])
def test_conv2d_vs_torch(H, cin, cout, k, stride, pad, dil, oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, H, H + 1, cin)).astype(np.float32)
    w = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    y = oracle.conv2d(x, w, stride, pad, dil)
    if pad == 'SAME':
        pt, pb, _ = oracle.same_pad(H, k, stride, dil)
        pl, pr, _ = oracle.same_pad(H + 1, k, stride, dil)
    elif pad == 'VALID':
        pt = pb = pl = pr = 0
    else:
        (pt, pb), (pl, pr) = pad
    xt = F.pad(t_nchw(x), (pl, pr, pt, pb))
    ref = from_t(F.conv2d(xt, torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), stride=stride, dilation=dil))
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 1e-4


def test_rect_kernels_vs_torch(oracle):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 12, 12, 5)).astype(np.float32)
    for kh, kw in ((15, 1), (1, 15)):
        w = rng.standard_normal((kh, kw, 5, 4)).astype(np.float32)
        y = oracle.conv2d(x, w, 1, 'SAME')
        ref = from_t(F.conv2d(t_nchw(x), torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))),
                              padding=(kh // 2, kw // 2)))
        assert np.abs(y - ref).max() < 1e-4


@pytest.mark.parametrize('dil', [1, 2])
def test_depthwise_vs_torch(dil, oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 14, 15, 6)).astype(np.float32)
    w = rng.standard_normal((3, 3, 6, 1)).astype(np.float32)
    y = oracle.depthwise_conv2d(x, w, dil)
    ref = from_t(F.conv2d(t_nchw(x), torch.from_numpy(np.ascontiguousarray(w.transpose(2, 3, 0, 1))), padding=dil,
                          dilation=dil, groups=6))
    assert np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize('H', [60, 61, 237])
def test_maxpool_vs_torch(H, oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, H, H, 3)).astype(np.float32)
    y = oracle.max_pool_3x3_s2_same(x)
    pt, pb, _ = oracle.same_pad(H, 3, 2)
    xt = F.pad(t_nchw(x), (pt, pb, pt, pb), value=float('-inf'))
    ref = from_t(F.max_pool2d(xt, 3, 2))
    assert np.array_equal(y, ref)


def test_batch_norm_and_dense_vs_torch(oracle):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 5, 5, 7)).astype(np.float32)
    w = {'bn/gamma': rng.uniform(.5, 1.5, 7).astype(np.float32), 'bn/beta': rng.standard_normal(7).astype(np.float32),
         'bn/moving_mean': rng.standard_normal(7).astype(np.float32),
         'bn/moving_variance': rng.uniform(.5, 2, 7).astype(np.float32)}
    y = oracle.batch_norm(x, w, 'bn', 1e-4)
    ref = from_t(F.batch_norm(t_nchw(x), torch.from_numpy(w['bn/moving_mean']), torch.from_numpy(w['bn/moving_variance']),
                              torch.from_numpy(w['bn/gamma']), torch.from_numpy(w['bn/beta']), False, 0., 1e-4))
    assert np.abs(y - ref).max() < 1e-5
    d = {'fc/kernel': rng.standard_normal((7, 4)).astype(np.float32), 'fc/bias': rng.standard_normal(4).astype(np.float32)}
    z = oracle.dense(x.reshape(-1, 7), d, 'fc', act_relu=True)
    refz = torch.relu(torch.from_numpy(x.reshape(-1, 7)) @ torch.from_numpy(d['fc/kernel']) + torch.from_numpy(d['fc/bias'])).numpy()
    assert np.abs(z - refz).max() < 1e-5
