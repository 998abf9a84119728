"""Dense / window kernels on the GPU vs the oracle's restatement of the tf.layers semantics.
Tolerance: exact-f32 MFMA differs from BLAS only by summation order -> 2e-5 of the output scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-5):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, (err, scale)


CONV_CASES = [
    # (N, H, W, cin, cout, kh, kw, stride, padding, dilation)
    (2, 30, 30, 728, 728, 1, 1, 1, 'SAME', 1),       # pointwise (block5-12)
    (1, 61, 61, 64, 128, 1, 1, 2, 'SAME', 1),        # residual projection, stride 2, odd size
    (1, 30, 30, 728, 512, 3, 3, 1, 'SAME', 1),       # RPN 3x3
    (1, 41, 41, 3, 32, 3, 3, 2, 'VALID', 1),         # stem conv1 (small-cin path)
    (1, 33, 33, 32, 64, 3, 3, 1, 'VALID', 1),        # stem conv2
    (1, 30, 30, 96, 64, 15, 1, 1, 'SAME', 1),        # large-sep (15,1)
    (1, 30, 30, 64, 490, 1, 15, 1, 'SAME', 1),       # large-sep (1,15), cout not a tile multiple
    (1, 300, 1, 490, 2048, 1, 1, 1, 'VALID', 1),     # dense subnet_fc (rows = ROIs)
    (1, 300, 1, 2048, 25, 1, 1, 1, 'VALID', 1),      # dense fc_cls+fc_loc (narrow N tile)
    (3, 17, 19, 40, 44, 3, 3, 1, 'SAME', 2),         # dilation 2, ragged channel counts
]


@pytest.fixture(params=['f32', 'f16x3'])
def precision(request):
    from xdet.runtime import set_precision
    set_precision(request.param)
    yield request.param
    set_precision('f32')


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_matches_oracle(case, oracle, precision):
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor
    N, H, W, cin, cout, kh, kw, stride, padding, dil = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    ref = oracle.conv2d(x, k, stride, padding, dil) * scale + shift
    y = Conv2D(k, stride, padding, dil, scale, shift)(DeviceTensor.from_numpy(x)).numpy()
    assert y.shape == ref.shape
    tol = 2e-5 if precision == 'f32' else 3e-5      # f16x3: ~2^-21 per product on top of summation order
    close(y, ref, tol)
    # fused residual + ReLU, and ReLU applied to the input on load
    res = rng.standard_normal(ref.shape).astype(np.float32)
    ref2 = np.maximum(oracle.conv2d(np.maximum(x, 0), k, stride, padding, dil) * scale + shift + res, 0)
    y2 = Conv2D(k, stride, padding, dil, scale, shift, relu=True)(DeviceTensor.from_numpy(x),
                                                                   residual=DeviceTensor.from_numpy(res),
                                                                   relu_in=True).numpy()
    close(y2, ref2, tol)
    if precision != 'f32' and cin >= 32 and stride == 1:
        # the same layer through pre-split f16 planes + LDS DMA (what a net uses for big contractions)
        y3 = Conv2D(k, stride, padding, dil, scale, shift, relu=True)(DeviceTensor.from_numpy(x),
                                                                       residual=DeviceTensor.from_numpy(res),
                                                                       relu_in=True, planes=True).numpy()
        close(y3, ref2, tol)
        y4 = Conv2D(k, stride, padding, dil, scale, shift)(DeviceTensor.from_numpy(x), planes=True).numpy()
        close(y4, ref, tol)


def test_plain_f16_mode_is_an_f16_gemm(oracle):
    """precision 'f16' == f32-accumulated product of f16-rounded operands (speed mode)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 30, 30, 256)).astype(np.float32)
    k = (rng.standard_normal((1, 1, 256, 128)) / 16).astype(np.float32)
    set_precision('f16')
    try:
        y = Conv2D(k, 1, 'SAME')(DeviceTensor.from_numpy(x)).numpy()
    finally:
        set_precision('f32')
    rb = lambda a: a.astype(np.float16).astype(np.float32)
    # weights are pre-scaled per output channel by a power of two before rounding: same rounding as
    # plain f16 for normal numbers
    ref = oracle.conv2d(rb(x), rb(k * 1024) / 1024, 1, 'SAME')
    close(y, ref, 2e-5)
    assert np.abs(y - oracle.conv2d(x, k, 1, 'SAME')).max() > 1e-4      # and it is NOT f32-accurate


def test_conv_explicit_padding_resnet_stem(oracle):
    """conv2d_fixed_padding(7, stride 2): pad 3/3 then VALID (net/resnet_v2.py:89-100)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 64, 64, 3)).astype(np.float32)
    k = (rng.standard_normal((7, 7, 3, 64)) / 12).astype(np.float32)
    ref = oracle.conv2d(x, k, 2, ((3, 3), (3, 3)))
    y = Conv2D(k, 2, 'EXPLICIT', explicit_pad=3)(DeviceTensor.from_numpy(x)).numpy()
    assert y.shape == ref.shape == (2, 32, 32, 64)
    close(y, ref)


@pytest.mark.parametrize('dil,relu_in,C,H', [(1, False, 64, 37), (1, True, 728, 30), (2, False, 1024, 30), (2, True, 40, 9)])
def test_depthwise_matches_oracle(dil, relu_in, C, H, oracle):
    from xdet.ops import DepthwiseConv2D
    from xdet.runtime import DeviceTensor
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, H, H + 3, C)).astype(np.float32)
    k = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    ref = oracle.depthwise_conv2d(np.maximum(x, 0) if relu_in else x, k, dil)
    y = DepthwiseConv2D(k, dil)(DeviceTensor.from_numpy(x), relu_in=relu_in).numpy()
    close(y, ref, 1e-6)


@pytest.mark.parametrize('N,H,W,C', [(1, 119, 119, 128), (2, 60, 60, 256), (1, 237, 237, 64), (3, 30, 30, 1024),
                                      (2, 33, 31, 96), (1, 5, 97, 32), (2, 9, 64, 160), (1, 4, 1, 64)])
def test_depthwise_tile_edges(N, H, W, C, oracle):
    """the tiled depthwise kernel at the network's own widths and at widths where the right halo column
    (x = W) falls on a DMA-segment / tile boundary (W % 32 in {0, 1, 23, 29, 31}), ragged last row tile,
    several tiles per workgroup"""
    from xdet.ops import DepthwiseConv2D
    from xdet.runtime import DeviceTensor
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    k = rng.standard_normal((3, 3, C, 1)).astype(np.float32)
    ref = oracle.depthwise_conv2d(np.maximum(x, 0), k, 1)
    y = DepthwiseConv2D(k, 1)(DeviceTensor.from_numpy(x), relu_in=True).numpy()
    close(y, ref, 1e-6)


SEP_CASES = [
    # (N, H, W, cin, cout, relu_in, relu_out): the entry-flow layers and the tile edges of the fused kernel
    (2, 237, 237, 64, 128, False, False),     # block2_sepconv1 (no leading ReLU)
    (1, 237, 237, 128, 128, True, False),     # block2_sepconv2
    (2, 119, 119, 128, 256, True, False),     # block3_sepconv1 (two 128-wide passes)
    (1, 119, 119, 256, 256, True, True),      # block3_sepconv2 (+ ReLU epilogue)
    (3, 30, 30, 32, 128, True, False),        # W == tile width
    (2, 31, 61, 96, 128, False, True),        # W = 2 tiles + 1, H % 4 = 3
    (1, 4, 1, 64, 128, True, False),          # one pixel column
    (5, 9, 29, 160, 256, True, False),        # several images per workgroup walk
    (2, 60, 60, 256, 728, True, False),       # block4_sepconv1: 728 -> 768 outputs as three 256-wide passes
    (1, 13, 31, 64, 96, False, True),         # fewer outputs than the 128-wide pass (masked channels)
    (1, 7, 33, 32, 1000, True, False),        # 1000 -> 1024: four passes
]


@pytest.mark.parametrize('case', SEP_CASES)
@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_fused_separable_block(case, prec, oracle):
    """the one-kernel separable block against the oracle, and bit for bit against the two-kernel form"""
    from xdet.ops import SeparableConvBN
    from xdet.runtime import DeviceTensor, set_precision
    N, H, W, cin, cout, relu_in, relu_out = case
    rng = np.random.default_rng(H * 7 + W * 3 + cin)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
    pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    set_precision(prec)
    try:
        op = SeparableConvBN(dk, pk, scale, shift, relu=relu_out)
    finally:
        set_precision('f32')
    xd = DeviceTensor.from_numpy(x)
    y_fused = op(xd, relu_in=relu_in, fused=True).numpy()
    y_split = op(xd, relu_in=relu_in, fused=False).numpy()
    assert np.array_equal(y_fused, y_split)
    if prec == 'f16x3':
        ref = oracle.separable_conv2d(np.maximum(x, 0) if relu_in else x, dk, pk) * scale + shift
        if relu_out:
            ref = np.maximum(ref, 0)
        close(y_fused, ref, 3e-5)


@pytest.mark.parametrize('N,H,W', [(2, 239, 239), (1, 3, 3), (3, 6, 33), (1, 35, 32), (2, 64, 61), (1, 7, 95)])
@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
@pytest.mark.parametrize('cout', [64, 40])
def test_conv3x3_staged_tile(N, H, W, prec, cout, oracle):
    """block1_conv2's kernel (3x3 VALID over 32 channels, input tile staged once in LDS, taps = shifted fragment
    reads) against the oracle and bit for bit against the implicit-GEMM kernel; ragged tiles in both directions"""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(N * 1000 + H * 7 + W)
    x = rng.standard_normal((N, H, W, 32)).astype(np.float32)
    k = (rng.standard_normal((3, 3, 32, cout)) / 17).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    set_precision(prec)
    try:
        op = Conv2D(k, 1, 'VALID', scale=scale, shift=shift, relu=True)
    finally:
        set_precision('f32')
    xd = DeviceTensor.from_numpy(x)
    y_gemm = op(xd, planes=True).numpy()
    y_tile = op(xd, planes=True, staged_tile=True).numpy()
    assert y_tile.shape == (N, H - 2, W - 2, cout)
    assert np.array_equal(y_tile, y_gemm)
    if prec == 'f16x3':
        close(y_tile, np.maximum(oracle.conv2d(x, k, 1, 'VALID') * scale + shift, 0), 3e-5)


@pytest.mark.parametrize('case', [(2, 237, 237, 128, 128, True), (1, 119, 119, 256, 256, True), (3, 60, 60, 64, 128, False),
                                  (2, 29, 57, 96, 128, True), (1, 4, 3, 32, 128, True), (2, 56, 28, 64, 256, False),
                                  (1, 9, 61, 160, 128, True)])
@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_separable_block_with_split_pool(case, prec, oracle):
    """block -> max_pooling2d(3, 2, 'same') -> + residual with the horizontal half of the pool in the block's epilogue
    (tiles 28 columns apart, windows from the accumulators by v_permlane32_swap) and a vertical pass: bit for bit the
    one-kernel block followed by the whole pool, odd and even sizes (pad 1/1 and 0/1), ragged last tiles."""
    from xdet.ops import SeparableConvBN, max_pool_3x3_s2_same_add, separable_block_then_pool_add
    from xdet.runtime import DeviceTensor, set_precision
    N, H, W, cin, cout, relu_in = case
    rng = np.random.default_rng(H * 13 + W + cin)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
    pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((N, -(-H // 2), -(-W // 2), cout)).astype(np.float32)
    set_precision(prec)
    try:
        op = SeparableConvBN(dk, pk, scale, shift, relu=False)
    finally:
        set_precision('f32')
    xd, rd = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(res)
    whole = max_pool_3x3_s2_same_add(op(xd, relu_in=relu_in, fused=True), rd).numpy()
    split = separable_block_then_pool_add(op, xd, rd, relu_in=relu_in).numpy()
    assert split.shape == whole.shape == res.shape
    assert np.array_equal(split, whole)
    no_res = separable_block_then_pool_add(op, xd, None, relu_in=relu_in).numpy()
    assert np.array_equal(no_res, max_pool_3x3_s2_same_add(op(xd, relu_in=relu_in, fused=True)).numpy())
    if prec == 'f16x3':
        y = oracle.separable_conv2d(np.maximum(x, 0) if relu_in else x, dk, pk) * scale + shift
        close(split, oracle.max_pool_3x3_s2_same(y) + res, 3e-5)


@pytest.mark.parametrize('H,W', [(237, 237), (119, 119), (60, 60), (7, 10)])
def test_maxpool_same_padding_asymmetry(H, W, oracle):
    """TF SAME puts the odd padding pixel at the bottom/right: 60->30 pads 0/1, 237->119 pads 1/1."""
    from xdet.ops import max_pool_3x3_s2_same_add
    from xdet.runtime import DeviceTensor
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, H, W, 40)).astype(np.float32)
    ref = oracle.max_pool_3x3_s2_same(x)
    res = rng.standard_normal(ref.shape).astype(np.float32)
    y = max_pool_3x3_s2_same_add(DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(res)).numpy()
    assert np.array_equal(y, ref + res)
    y0 = max_pool_3x3_s2_same_add(DeviceTensor.from_numpy(x)).numpy()
    assert np.array_equal(y0, ref)


@pytest.mark.parametrize('mag', [1e-4, 1e-2, 1.0, 1e2, 3e3, 2e4])
def test_split_precision_across_activation_magnitudes(mag, oracle):
    """x = hi + lo with f16 parts.  Inside this network activations are O(1)..O(100) (BN after every contraction, DFT
    bins <= 30x the signal); this pins what the f16x3 path does away from that range:
      * 1e-2 .. 3e3 (elements up to ~1e4): f32-class accuracy relative to the output scale;
      * a tensor that is small as a whole (1e-4): lo = x - hi falls below f16's normal range (2^-14) and degrades
        gracefully to an ABSOLUTE error of ~2^-25 per element -- ~2e-4 relative for such a tensor;
      * elements beyond f16's 65504: hi overflows and the result is not finite.  No layer of the net comes near
        either end with BN-normalised activations; a checkpoint that does needs a power-of-two activation
        pre-scale in the producing epilogue (not built)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(17)
    x = (rng.standard_normal((1, 30, 30, 256)) * mag).astype(np.float32)
    k = (rng.standard_normal((1, 1, 256, 128)) / 16).astype(np.float32)
    ref = oracle.conv2d(x, k, 1, 'SAME')
    set_precision('f16x3')
    try:
        y = Conv2D(k, 1, 'SAME')(DeviceTensor.from_numpy(x), planes=True).numpy()
    finally:
        set_precision('f32')
    if np.abs(x).max() > 65504:
        assert not np.isfinite(y).all()                    # documented: overflow is loud (inf / nan), never silent
        return
    err = float(np.abs(y - ref).max()) / float(np.abs(ref).max())
    print('activation magnitude %g: f16x3 error relative to the output scale %.2e' % (mag, err))
    assert np.isfinite(y).all()
    assert err < (1e-3 if mag < 1e-3 else 3e-5), err


def test_fused_separable_block_beyond_2gib():
    """75 images of 237 x 237 x 128 are 2.16 GB in and 2.16 GB out: the fused kernel's LDS DMA uses 32-bit buffer
    offsets, so the launcher cuts the batch into image ranges below 2 GiB (74 + 1 here).  Both forms must still agree
    bit for bit, including across the cut."""
    from xdet.ops import SeparableConvBN
    from xdet.runtime import DeviceTensor, set_precision
    N, H, W, cin, cout = 75, 237, 237, 128, 128
    rng = np.random.default_rng(5)
    one = rng.standard_normal((3, H, W, cin)).astype(np.float32)
    x = np.concatenate([one] * 25)                       # 75 images from 3 distinct ones: the answer must repeat too
    dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
    pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    set_precision('f16x3')
    try:
        op = SeparableConvBN(dk, pk, None, None, relu=False)
    finally:
        set_precision('f32')
    xd = DeviceTensor.from_numpy(x)
    del x
    y = op(xd, relu_in=True, fused=True).numpy()
    assert np.isfinite(y).all()
    for k in range(1, 25):
        assert np.array_equal(y[3 * k:3 * k + 3], y[:3]), k      # images 72..74 and 75th sit across / behind the cut
    y3 = op(DeviceTensor.from_numpy(one), relu_in=True, fused=False).numpy()
    assert np.array_equal(y[:3], y3)


def test_specialised_conv_entry_points_reject_what_they_cannot_run():
    """xdet_conv3x3_patch_forward / xdet_sepconv_fused_forward are shape-specialised kernels behind the C-ABI: a layer
    outside their shape class is an InvalidArgumentError (never a silent fall-back to another kernel)."""
    from xdet._lib import InvalidArgumentError
    from xdet.ops import Conv2D, SeparableConvBN
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(0)
    x32 = DeviceTensor.from_numpy(rng.standard_normal((1, 9, 9, 32)).astype(np.float32))
    x64 = DeviceTensor.from_numpy(rng.standard_normal((1, 9, 9, 64)).astype(np.float32))
    k = lambda *s: (rng.standard_normal(s) / 8).astype(np.float32)
    set_precision('f16x3')
    try:
        same = Conv2D(k(3, 3, 32, 64), 1, 'SAME')            # padded: the staged-tile kernel is VALID only
        wide_in = Conv2D(k(3, 3, 64, 64), 1, 'VALID')        # 64 input channels
        wide_out = Conv2D(k(3, 3, 32, 96), 1, 'VALID')       # > 64 outputs
        sep_wide = SeparableConvBN(k(3, 3, 64, 1), k(1, 1, 64, 384), None, None)   # 384 outputs: two-kernel form only
    finally:
        set_precision('f32')
    exact = Conv2D(k(3, 3, 32, 64), 1, 'VALID')              # created in f32 mode: no split-precision weights
    for op, x in [(same, x32), (wide_in, x64), (wide_out, x32), (exact, x32)]:
        with pytest.raises(InvalidArgumentError):
            op(x, planes=True, staged_tile=True)
    with pytest.raises(InvalidArgumentError):
        sep_wide(x64, fused=True)
    assert sep_wide(x64, fused=False).numpy().shape == (1, 9, 9, 384)


def test_fused_block_and_split_pool_random_shapes():
    """60 random (N, H, W, Cin, Cout, ReLU) draws -- single rows / columns, widths around the 28- and 30-column tile
    steps, every output-width class of the fused kernel (one 128-wide pass, 256-wide passes, masked channels): the
    one-kernel block equals depthwise -> split -> pointwise bit for bit, and block + split pool equals block + whole pool."""
    from xdet.ops import SeparableConvBN, max_pool_3x3_s2_same_add, separable_block_then_pool_add
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(2024)
    widths = [1, 2, 3, 13, 14, 15, 27, 28, 29, 30, 31, 32, 55, 56, 57, 59, 60, 61, 85, 91]
    for it in range(60):
        N = int(rng.integers(1, 4))
        H = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 17, 33]))
        W = int(rng.choice(widths))
        cin = int(rng.choice([32, 40, 64, 96, 128, 200, 256]))
        cout = int(rng.choice([70, 96, 128, 200, 256, 500, 512, 728, 1000]))     # padded to 128 or a multiple of 256
        relu_in, relu_out = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
        dk = rng.standard_normal((3, 3, cin, 1)).astype(np.float32) / 3
        pk = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = rng.standard_normal(cout).astype(np.float32)
        res = rng.standard_normal((N, -(-H // 2), -(-W // 2), cout)).astype(np.float32)
        set_precision('f16x3' if it % 3 else 'f16')
        try:
            op = SeparableConvBN(dk, pk, scale, shift, relu=relu_out)
        finally:
            set_precision('f32')
        tag = (it, N, H, W, cin, cout, relu_in, relu_out)
        xd, rd = DeviceTensor.from_numpy(x), DeviceTensor.from_numpy(res)
        fused = op(xd, relu_in=relu_in, fused=True)
        assert np.array_equal(fused.numpy(), op(xd, relu_in=relu_in, fused=False).numpy()), tag
        whole = max_pool_3x3_s2_same_add(fused, rd).numpy()
        assert np.array_equal(separable_block_then_pool_add(op, xd, rd, relu_in=relu_in).numpy(), whole), tag


KSPLIT_CASES = [
    # (N, H, W, cin, cout, kh, kw, stride, padding)    -- the layers the split-K kernel was built for, at small sizes
    (8, 15, 15, 512, 512, 3, 3, 1, 'SAME'),            # ResNet-50 stage 4 3x3 at BASELINE config 2's batch (144 K steps, 15 M tiles)
    (2, 30, 30, 256, 256, 3, 3, 1, 'SAME'),            # stage 3 3x3
    (2, 15, 15, 2048, 512, 1, 1, 1, 'SAME'),           # stage 4 reducing 1x1 (64 K steps)
    (1, 30, 30, 728, 512, 3, 3, 1, 'SAME'),            # RPN 3x3 of one image (207 K steps on 8 x 4 tiles)
    (1, 300, 1, 2048, 25, 1, 1, 1, 'VALID'),           # head fc_cls+fc_loc (N tile 64, 3 M tiles)
    (3, 17, 19, 96, 130, 3, 3, 1, 'SAME'),             # ragged: M tail, cout not a tile multiple (N tile 64), 27 K steps
    (1, 60, 60, 128, 128, 3, 3, 2, 'SAME'),            # strided 3x3 reading planes (ResNet stage openers)
]


@pytest.mark.parametrize('case', KSPLIT_CASES)
@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_ksplit_modes_are_bit_identical_and_match_the_plain_kernel(case, prec, oracle):
    """csrc/conv_mfma_ksplit.hip: (i) with ONE range the kernel reproduces the plain LDS-DMA kernels bit for bit (same
    per-element product order); (ii) for every ksplit the parallel-ranges mode (scratch slabs + last-arriver fold) and
    the one-workgroup-per-tile mode (fold at the range boundaries) give the SAME bits -- the split is a property of the
    layer, the mode a property of the launch; (iii) every split stays within the usual tolerance of the oracle; and the
    ticket counters clean up after themselves (a second launch gives the same bits)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    N, H, W, cin, cout, kh, kw, stride, padding = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    ref = np.maximum(oracle.conv2d(x, k, stride, padding, 1) * scale + shift, 0)
    res = None
    set_precision(prec)
    try:
        xd = DeviceTensor.from_numpy(x)
        conv = Conv2D(k, stride, padding, 1, scale, shift, relu=True)
        plain = conv(xd, planes=True).numpy()
        out = {}
        for S in (1, 2, 3, 4, 8):
            for mode in (1, 2):
                conv.set_ksplit(S, mode, 64)
                a = conv(xd, planes=True).numpy()
                b = conv(xd, planes=True).numpy()               # tickets were reset by the last arriver
                assert np.array_equal(a, b), (S, mode)
                out[(S, mode)] = a
            assert np.array_equal(out[(S, 1)], out[(S, 2)]), S    # parallel == sequential, bit for bit
            close(out[(S, 1)], ref, 3e-5 if prec == 'f16x3' else 2e-2)
        assert np.array_equal(out[(1, 1)], plain)                 # one range == the plain kernels
        conv.set_ksplit(4, 0, 448)                                # mode by grid size
        assert np.array_equal(conv(xd, planes=True).numpy(), out[(4, 1)])
        # residual + the f32 output's companions are the shared epilogue's business: one check through it
        res = rng.standard_normal(ref.shape).astype(np.float32)
        conv.set_ksplit(2, 1, 64)
        y = conv(xd, residual=DeviceTensor.from_numpy(res), planes=True).numpy()
        conv.set_ksplit(0)
        y0 = conv(xd, residual=DeviceTensor.from_numpy(res), planes=True).numpy()
    finally:
        set_precision('f32')
    close(y, np.maximum(oracle.conv2d(x, k, stride, padding, 1) * scale + shift + res, 0), 3e-5 if prec == 'f16x3' else 2e-2)
    close(y, y0, 3e-6 if prec == 'f16x3' else 2e-2)


def test_ksplit_fold_by_a_second_launch_beyond_the_ticket_array():
    """The parallel-ranges mode folds inside the GEMM launch (the last workgroup of a tile to arrive, conv_params.h ks_ticket:
    4096 tickets).  A launch of more tiles than that folds by a second launch (conv_ksplit_fold_kernel) -- the same left
    fold, the same epilogue: the same bits as the sequential mode."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(9)
    x = rng.standard_normal((40, 120, 120, 64)).astype(np.float32)           # 576,000 rows: 4,500 M tiles of 128
    k = (rng.standard_normal((1, 1, 64, 64)) / 8).astype(np.float32)
    set_precision('f16x3')
    try:
        xd = DeviceTensor.from_numpy(x)
        conv = Conv2D(k, 1, 'SAME', 1, None, None, relu=True)
        conv.set_ksplit(2, 1, 4608)
        par = conv(xd, planes=True).numpy()
        conv.set_ksplit(2, 2, 0)
        seq = conv(xd, planes=True).numpy()
    finally:
        set_precision('f32')
    assert np.array_equal(par, seq)
    assert np.isfinite(par).all() and np.abs(par).max() > 0


def test_ksplit_is_batch_invariant():
    """an image's rows do not depend on the batch it arrives in, although small batches run the ranges in parallel and
    large ones sequentially (mode 0: by grid size)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(5)
    x = rng.standard_normal((40, 15, 15, 512)).astype(np.float32)
    k = (rng.standard_normal((3, 3, 512, 512)) / np.sqrt(9 * 512)).astype(np.float32)
    set_precision('f16x3')
    try:
        conv = Conv2D(k, 1, 'SAME', 1, None, None, relu=False)
        conv.set_ksplit(4, 0, 448 // 4)
        big = conv(DeviceTensor.from_numpy(x), planes=True).numpy()          # 71 M tiles x 4 N tiles x 4 > 448: sequential
        for n0, n in ((0, 1), (7, 3), (20, 8)):
            small = conv(DeviceTensor.from_numpy(x[n0:n0 + n]), planes=True).numpy()   # parallel ranges
            assert np.array_equal(small, big[n0:n0 + n]), (n0, n)
    finally:
        set_precision('f32')


@pytest.mark.parametrize('dil', [1, 2])
def test_depthwise_tile_kernel_beyond_2gib(dil):
    """the 397 x 397 x 128 tensor of the 800 x 800 input at batch 96 is 7.7 GB; the tile kernel's DMA uses
    32-bit buffer offsets and round 3 sent the whole tensor to the slow per-row kernel.  Now the launcher cuts the batch
    into ranges of whole images below 2 GiB (26 + 2 here): same bits as image-by-image calls, also across the cut."""
    from xdet.ops import DepthwiseConv2D
    from xdet.runtime import DeviceTensor
    N, H, W, C = 28, 397, 397, 128                       # 80.7 MB per image: 26 images per range
    rng = np.random.default_rng(6)
    two = rng.standard_normal((2, H, W, C)).astype(np.float32)
    dk = rng.standard_normal((3, 3, C, 1)).astype(np.float32) / 3
    op = DepthwiseConv2D(dk, dil)
    x = np.concatenate([two] * 14)
    xd = DeviceTensor.from_numpy(x)
    del x
    y = op(xd, relu_in=True).numpy()
    ref = op(DeviceTensor.from_numpy(two), relu_in=True).numpy()
    for k in range(14):                                  # images 26, 27 sit behind the cut
        assert np.array_equal(y[2 * k:2 * k + 2], ref), k


def test_ksplit_fold_in_the_large_tile_kernel_is_the_same_function():
    """the RPN 3x3 conv carries ksplit = 8 (a single image is 32 tiles against 207 K steps).  At bench-size batches its grid
    is large and the layer runs on the 256 x 128 LDS-DMA kernel, which folds its accumulators at the range boundaries
    (FOLD) -- the same expression tree as the parallel-ranges launch a single image takes: same bits."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    rng = np.random.default_rng(8)
    x = rng.standard_normal((48, 30, 30, 728)).astype(np.float32)
    k = (rng.standard_normal((3, 3, 728, 512)) / np.sqrt(9 * 728)).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32)
    set_precision('f16x3')
    try:
        conv = Conv2D(k, 1, 'SAME', 1, None, b, relu=True)
        conv.set_ksplit(8, 0, 448 // 8)
        big = conv(DeviceTensor.from_numpy(x), planes=True).numpy()          # 169 x 4 tiles of 256 x 128: the FOLD kernel
        for n0, n in ((0, 1), (17, 1), (40, 2)):
            small = conv(DeviceTensor.from_numpy(x[n0:n0 + n]), planes=True).numpy()   # 8 .. 16 M tiles: parallel ranges
            assert np.array_equal(small, big[n0:n0 + n]), (n0, n)
        conv.set_ksplit(8, 2, 0)                                             # ... and the split-K kernel's own sequential mode
        assert np.array_equal(conv(DeviceTensor.from_numpy(x[:3]), planes=True).numpy(), big[:3])
    finally:
        set_precision('f32')



@pytest.mark.parametrize('shape', [(80, 30, 30, 728, 728, True, False), (70, 31, 29, 512, 1024, False, True),
                                   (300, 15, 15, 1024, 1536, True, True)])
def test_large_batch_pointwise_gemm_is_the_same_function(shape):
    """Pointwise layers at bench-size batches run on 256 x 256 tiles whose epilogue takes the form without per-row
    predicates (conv_epilogue_full: buffer stores, columns beyond ldo out of range) on every tile but the ragged last one.
    Same K order and the same epilogue arithmetic: an image's rows are the bits the small-batch kernels (128 x 64 deep ring,
    128 x 128) give it -- also across the ragged last M tile and the padded last N tile (728 -> 768)."""
    from xdet.ops import Conv2D
    from xdet.runtime import DeviceTensor, set_precision
    N, H, W, cin, cout, with_res, relu = shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((N, H, W, cout)).astype(np.float32) if with_res else None
    set_precision('f16x3')
    try:
        conv = Conv2D(k, 1, 'SAME', 1, sc, sh, relu=relu)
        big = conv(DeviceTensor.from_numpy(x), residual=DeviceTensor.from_numpy(res) if with_res else None, planes=True).numpy()
        for n0, n in ((0, 1), (N // 2, 3), (N - 2, 2)):
            small = conv(DeviceTensor.from_numpy(x[n0:n0 + n]),
                         residual=DeviceTensor.from_numpy(res[n0:n0 + n]) if with_res else None, planes=True).numpy()
            assert np.array_equal(small, big[n0:n0 + n]), (n0, n)
    finally:
        set_precision('f32')
    ref = x[:2].reshape(-1, cin).astype(np.float64) @ k.reshape(cin, cout).astype(np.float64) * sc + sh
    if with_res:
        ref = ref + res[:2].reshape(-1, cout)
    if relu:
        ref = np.maximum(ref, 0)
    assert np.abs(big[:2].reshape(-1, cout) - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
