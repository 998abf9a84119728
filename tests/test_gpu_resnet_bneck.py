"""The fused identity-bottleneck kernel (csrc/resnet_bneck.hip; net/resnet_v2.py:142-184) and the stem's own kernels
(resnet_stem.hip, the 7x7 conv from NCHW; maxpool3x3s2_bn_planes_kernel, pool + pre-activation; :311-330 + :142-156) against the three-launch / two-pass forms of the
same blocks: same products in the same order, so the trunk's output must be the same BITS -- on whole tiles (120 x 120
and 60 x 60 maps: 480 x 480 input), on ragged ones (40 x 40 / 20 x 20: 160 x 160 input; 24 x 24: 96 x 96) and for a batch
tail -- and, through tests/test_gpu_resnet.py, within 1e-4 of the oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trunk(w, size, max_batch, fused):
    from xdet.resnet import ResNet50Trunk
    from xdet.runtime import set_precision
    opts = {k: 'on' if fused else 'off' for k in ('bneck', 'stem_pool', 'stem7', 'preconv')}     # xdet_resnet_set_option
    set_precision('f16x3')
    try:
        return ResNet50Trunk(w, image_size=size, max_batch=max_batch, options=opts)
    finally:
        set_precision('f32')


@pytest.mark.parametrize('size,batch,n', [(480, 2, 2), (160, 3, 3), (96, 4, 3), (224, 1, 1)])
def test_fused_blocks_are_bit_identical_to_three_launches(size, batch, n):
    from xdet import weights as W
    w = W.make_resnet50_weights(4321)
    imgs = W.synthetic_images(n, size, seed=11 + size)
    fused = _trunk(w, size, batch, True)
    plain = _trunk(w, size, batch, False)
    a = fused.forward(imgs)
    b = plain.forward(imgs)
    assert np.isfinite(a).all()
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    # graph replay of the fused plan == its eager forward
    fused.set_images(imgs)
    fused.forward_device(n, use_graph=True)
    fused.stream.synchronize()
    from xdet.runtime import to_host
    assert np.array_equal(to_host(fused._out.ptr, (n,) + fused.out_shape, np.float32), a)


@pytest.mark.parametrize('size,batch,n', [(480, 2, 2), (160, 3, 3), (96, 4, 3), (224, 1, 1)])
def test_projection_folded_into_the_closing_conv(size, batch, n):
    """option "projcat": a projection block's output as ONE contraction [3x3 output | block input] x [w_c ; w_proj]
    (net/resnet_v2.py:160-184) against projection GEMM + closing GEMM + residual add.  One f32 accumulation instead of two
    roundings and an add: not the same bits, the same numbers to f32 rounding of a 4-block-deep difference (the oracle bar
    of tests/test_gpu_resnet.py is 1e-4 of the output's scale)."""
    from xdet import weights as W
    from xdet.resnet import ResNet50Trunk
    from xdet.runtime import set_precision, to_host
    w = W.make_resnet50_weights(4321)
    imgs = W.synthetic_images(n, size, seed=11 + size)
    outs = {}
    for flag in ('1', '0'):
        set_precision('f16x3')
        try:
            t = ResNet50Trunk(w, image_size=size, max_batch=batch, options={'projcat': 'on' if flag == '1' else 'off'})
        finally:
            set_precision('f32')
        outs[flag] = t.forward(imgs)
        if flag == '1':
            t.set_images(imgs)
            t.forward_device(n, use_graph=True)
            t.stream.synchronize()
            assert np.array_equal(to_host(t._out.ptr, (n,) + t.out_shape, np.float32), outs['1'])
            # batch invariance of the folded form: one image alone gives the bits it has inside the batch
            one = t.forward(imgs[:1])
            assert np.array_equal(one[0], outs['1'][0])
    a, b = outs['1'], outs['0']
    assert np.isfinite(a).all()
    scale = float(np.abs(b).max())
    assert float(np.abs(a - b).max()) <= 2e-5 * scale, (float(np.abs(a - b).max()), scale)
    assert not np.array_equal(a, b)      # (the two forms do differ in the last bits: the switch is live)


def _planes_to_f32(hi_buf, lo_buf, n_pix, ld):
    """[pix/16][ld/32][16][32] f16 hi / lo planes -> f32 [n_pix][ld] of hi + lo"""
    from xdet.runtime import to_host
    g = -(-n_pix // 16)
    hi = to_host(hi_buf.ptr, (g, ld // 32, 16, 32), np.float16).astype(np.float32)
    lo = to_host(lo_buf.ptr, (g, ld // 32, 16, 32), np.float16).astype(np.float32)
    return (hi + lo).transpose(0, 2, 1, 3).reshape(g * 16, ld)[:n_pix]


@pytest.mark.parametrize('N,H,W', [(2, 8, 30), (1, 12, 60), (3, 10, 37), (1, 5, 7), (2, 120, 120)])
def test_bneck_op_is_bit_identical_to_three_layers(N, H, W):
    """xdet_resnet_bneck_forward against relu(bn(x)) -> xdet_split_f32 -> xdet_conv_forward_planes x 3 on whole and ragged tiles"""
    import ctypes
    from xdet import ops
    from xdet._lib import lib, check
    from xdet.runtime import DeviceBuffer, DeviceTensor, set_precision, synchronize, to_device
    rng = np.random.RandomState(N * 1000 + H * 10 + W)
    cin, cmid = 256, 64
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    # the pre-activation BN: power-of-two scales, so that numpy's x * s + h (one rounding) is the kernel's fused multiply-add
    ps = rng.choice(np.array([0.5, 1.0, 2.0], np.float32), cin)
    ph = rng.uniform(-0.3, 0.3, cin).astype(np.float32)
    pre = np.maximum(x * ps + ph, 0).astype(np.float32)

    def bn(c):
        return rng.uniform(0.5, 1.5, c).astype(np.float32), rng.uniform(-0.2, 0.2, c).astype(np.float32)
    wa = (rng.standard_normal((1, 1, cin, cmid)) / np.sqrt(cin)).astype(np.float32)
    wb = (rng.standard_normal((3, 3, cmid, cmid)) / np.sqrt(9 * cmid)).astype(np.float32)
    wc = (rng.standard_normal((1, 1, cmid, cin)) / np.sqrt(cmid)).astype(np.float32)
    sa, ha = bn(cmid)
    sb, hb = bn(cmid)
    ns, nh = bn(cin)
    set_precision('f16x3')
    try:
        A = ops.Conv2D(wa, scale=sa, shift=ha, relu=True)
        B = ops.Conv2D(wb, scale=sb, shift=hb, relu=True)
        C = ops.Conv2D(wc)
    finally:
        set_precision('f32')
    dx, dpre = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(pre)
    y1 = A(dpre, planes=True)
    y2 = B(y1, planes=True)
    ref = C(y2, planes=True, residual=dx).numpy()

    n_pix = N * H * W
    nh16 = -(-n_pix // 16) * 16
    ohi, olo = DeviceBuffer(nh16 * cin * 2 + 512, zero=True), DeviceBuffer(nh16 * cin * 2 + 512, zero=True)
    out = DeviceTensor.empty((N, H, W, cin))
    dps, dph, dns, dnh = to_device(ps), to_device(ph), to_device(ns), to_device(nh)
    check(lib().xdet_resnet_bneck_forward(A.handle, B.handle, C.handle, dps.ptr, dph.ptr, dx.ptr, N, H, W, out.ptr, dns.ptr,
                                          dnh.ptr, ohi.ptr, olo.ptr, None))
    synchronize()
    got = out.numpy()
    if not np.array_equal(got, ref):
        bad = np.argwhere(got != ref)
        d = np.abs(got - ref)
        print('mismatches: %d of %d, max |d| %g (ref max %g); non-finite %d' % (len(bad), got.size, np.nanmax(d), np.abs(ref).max(), (~np.isfinite(got)).sum()))
        for ax, name in enumerate('nyxc'):
            u, c = np.unique(bad[:, ax], return_counts=True)
            print(' axis %s: %s' % (name, dict(zip(u.tolist()[:40], c.tolist()[:40]))))
    assert np.array_equal(got, ref)
    # the planes copy: relu(out * ns + nh) to the last bit of the f32 sum hi + lo (24 significant bits: hi and lo are 11 each, the
    # rest is below 2^-22 of the value)
    want = np.maximum(ref.reshape(n_pix, cin).astype(np.float64) * ns + nh, 0)
    pl = _planes_to_f32(ohi, olo, n_pix, cin)
    assert np.abs(pl - want).max() <= 2.0 ** -20 * max(1.0, float(np.abs(want).max()))
