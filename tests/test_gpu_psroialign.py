"""PsRoiAlign on the GPU through the C-ABI vs the oracle: bit-exact values AND indices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# known-answer vectors for the reference's own test inputs (cpp/PSROIPooling/test_op.py:52-81),
# as recorded in SURVEY.md 8c
KAT_ROIS = np.array([[[0.2, 0.2, 0.7, 0.7], [0.5, 0.5, 0.9, 0.9], [0.9, 0.9, 1., 1.]]], np.float32)
KAT = {
    'mean': ([[5.125, 6.5, 12., 13.375], [9.25, 11.375, 19.875, 22.], [17.5, 18.6875, 23.4375, 24.625]], [0, 0, 0]),
    'max': ([[7.1875, 8.5625, 14.0625, 15.4375], [13.75, 15.625, 23.125, 25.], [19.75, 20.625, 24.125, 25.]], [3, 8, 3]),
}


def kat_input():
    plane = np.arange(1, 26, dtype=np.float32).reshape(5, 5)
    return np.tile(plane, (1, 16, 1, 1)).astype(np.float32)


@pytest.mark.parametrize('method', ['mean', 'max'])
def test_reference_test_op_inputs(method, oracle):
    import xdet
    p, i = xdet.ps_roi_align(kat_input(), KAT_ROIS, 2, 2, method)
    assert p.shape == (1, 3, 4, 4) and i.shape == (1, 3, 4, 4) and i.dtype == np.int32
    vals, idx = KAT[method]
    for r in range(3):
        for b in range(4):
            assert np.all(p[0, r, b] == np.float32(vals[r][b]))
        assert np.all(i[0, r] == idx[r])
    po, io = oracle.ps_roi_align(kat_input(), KAT_ROIS, 2, 2, method)
    assert np.array_equal(p, po) and np.array_equal(i, io)


def random_rois(rng, n, r):
    cy, cx = rng.uniform(0.02, 0.98, (n, r)), rng.uniform(0.02, 0.98, (n, r))
    h, w = rng.uniform(0.01, 1.0, (n, r)), rng.uniform(0.01, 1.0, (n, r))
    rois = np.stack([cy, cx, h, w], -1).astype(np.float32)
    # edge cases: full image, 1-pixel, image border, the fallback box, degenerate (zero h / w)
    rois[:, 0] = [0.5, 0.5, 1.0, 1.0]
    rois[:, 1] = [0.5, 0.5, 1. / 30, 1. / 30]
    rois[:, 2] = [0.0, 0.0, 0.3, 0.3]
    rois[:, 3] = [1.0, 1.0, 0.2, 0.2]
    rois[:, 4] = [0.5, 0.5, 0.6, 0.6]
    rois[:, 5] = [0.5, 0.5, 0.0, 0.4]
    rois[:, 6] = [0.5, 0.5, 0.4, 0.0]
    rois[:, 7] = [0.999, 0.001, 0.001, 0.001]
    return rois


@pytest.mark.parametrize('method', ['max', 'mean'])
@pytest.mark.parametrize('shape', [(1, 490, 30, 30, 300, 7), (2, 490, 30, 30, 1000, 7), (1, 36, 50, 50, 64, 3),
                                   (3, 16, 5, 7, 9, 2)])
def test_random_bit_exact(method, shape, oracle):
    import xdet
    n, c, h, w, r, g = shape
    rng = np.random.default_rng(7)
    feat = rng.standard_normal((n, c, h, w)).astype(np.float32)
    rois = random_rois(rng, n, max(r, 8))[:, :r]
    p, i = xdet.ps_roi_align(feat, rois, g, g, method)
    po, io = oracle.ps_roi_align(feat, rois, g, g, method)
    assert np.array_equal(p, po)
    assert np.array_equal(i, io)


@pytest.mark.parametrize('method', ['max', 'mean'])
@pytest.mark.parametrize('shape', [(3, 30, 30, 300, 7, 7), (9, 50, 50, 77, 7, 7), (2, 13, 17, 40, 3, 5), (1, 120, 120, 33, 7, 7)])
@pytest.mark.parametrize('corners', [0, 1])
def test_nhwc_bank10_form_bit_exact(method, shape, corners, oracle):
    """The form the net runs: NHWC map with a padded channel stride, 10 channels per bin (psroialign_fwd_kernel: one
    wavefront per ROI, lanes over the output elements), ROIs as centres or as corners, images spread over the XCD-aware block order -- values and
    argmax indices bit for bit against the oracle (which takes NCHW + centres).  120 x 120 maps give bins with more than
    8 sample columns."""
    import ctypes
    from xdet._lib import lib, check
    from xdet.runtime import DeviceBuffer, to_device, to_host
    n, h, w, r, gh, gw = shape
    c = 10 * gh * gw
    ldc = -(-c // 32) * 32
    rng = np.random.default_rng(n * 100 + h)
    feat = rng.standard_normal((n, c, h, w)).astype(np.float32)
    rois = random_rois(rng, n, max(r, 8))[:, :r]
    po, io = oracle.ps_roi_align(feat, rois, gw, gh, method)
    nhwc = np.zeros((n, h, w, ldc), np.float32)
    nhwc[..., :c] = np.transpose(feat, (0, 2, 3, 1))
    rin = rois
    if corners:     # (ymin, xmin, ymax, xmax) whose _point2center is the centre form to the last bit: halves of small integers
        k = rng.integers(0, 64, (n, r, 4)).astype(np.float32) / 64
        ymin, xmin = np.minimum(k[..., 0], k[..., 2]), np.minimum(k[..., 1], k[..., 3])
        ymax, xmax = np.maximum(k[..., 0], k[..., 2]), np.maximum(k[..., 1], k[..., 3])
        rin = np.stack([ymin, xmin, ymax, xmax], -1).astype(np.float32)
        hh, ww = ymax - ymin, xmax - xmin
        cen = np.stack([ymin + hh / np.float32(2), xmin + ww / np.float32(2), hh, ww], -1).astype(np.float32)
        po, io = oracle.ps_roi_align(feat, cen, gw, gh, method)
    d_in, d_roi = to_device(nhwc), to_device(np.ascontiguousarray(rin))
    d_pool, d_idx = DeviceBuffer(n * r * c * 4), DeviceBuffer(n * r * c * 4)
    check(lib().xdet_psroialign_fwd(d_in.ptr, d_roi.ptr, d_pool.ptr, d_idx.ptr, n, c, h, w, r, gw, gh,
                                    1 if method == 'max' else 0, 1, ldc, c, corners, None))
    p = to_host(d_pool.ptr, (n, r, gh * gw, 10), np.float32)
    i = to_host(d_idx.ptr, (n, r, gh * gw, 10), np.int32)
    assert np.array_equal(p, po)
    assert np.array_equal(i, io)
    # without the index output (as inside the net)
    d_pool2 = DeviceBuffer(n * r * c * 4)
    check(lib().xdet_psroialign_fwd(d_in.ptr, d_roi.ptr, d_pool2.ptr, None, n, c, h, w, r, gw, gh,
                                    1 if method == 'max' else 0, 1, ldc, c, corners, None))
    assert np.array_equal(to_host(d_pool2.ptr, (n, r, gh * gw, 10), np.float32), po)


def test_empty_and_errors():
    import xdet
    feat = np.zeros((1, 8, 4, 4), np.float32)
    p, i = xdet.ps_roi_align(feat, np.zeros((1, 0, 4), np.float32), 2, 2, 'max')
    assert p.shape == (1, 0, 4, 2)
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat[0], np.zeros((1, 2, 4), np.float32), 2, 2, 'max')          # rank-4 NCHW
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat, np.zeros((1, 2, 5), np.float32), 2, 2, 'max')             # last dim 4
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat, np.zeros((2, 2, 4), np.float32), 2, 2, 'max')             # batch mismatch
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat, np.zeros((1, 2, 4), np.float32), 2, 2, 'median')          # pool_method
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat, np.zeros((1, 2, 4), np.float32), -1, 2, 'max')            # grid >= 0
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align(feat, np.zeros((1, 2, 4), np.float32), 3, 1, 'max')             # C % grid


@pytest.mark.parametrize('method', ['max', 'mean'])
@pytest.mark.parametrize('shape', [(1, 490, 30, 30, 300, 7), (2, 36, 50, 50, 64, 3), (3, 16, 5, 7, 9, 2)])
def test_grad_matches_oracle(method, shape, oracle):
    """F2: the backward scatters with float atomics as the reference does (ps_roi_align_grad_op.cu:118-134),
    so the summation order is unspecified: agreement with the sequential oracle is to rounding
    (|diff| <= 1e-5 * (sum of |terms|) bound, far below one term), not bit-exact."""
    import xdet
    n, c, h, w, r, g = shape
    rng = np.random.default_rng(9)
    feat = rng.standard_normal((n, c, h, w)).astype(np.float32)
    rois = random_rois(rng, n, max(r, 8))[:, :r]
    _, idx = oracle.ps_roi_align(feat, rois, g, g, method)
    G = rng.standard_normal(idx.shape).astype(np.float32)
    got = xdet.ps_roi_align_grad(feat, rois, G, idx, g, g, method)
    ref = oracle.ps_roi_align_grad(feat, rois, G, idx, g, g, method)
    mag = oracle.ps_roi_align_grad(feat, rois, np.abs(G), idx, g, g, method)
    assert got.shape == ref.shape
    assert np.all(np.abs(got - ref) <= 1e-5 * mag + 1e-30)
    assert np.array_equal(got == 0, ref == 0) or np.all(np.abs(got - ref)[(got == 0) != (ref == 0)] < 1e-6)


def test_grad_roundtrip_with_gpu_forward(oracle):
    """forward (GPU) -> its own pooled_index -> backward (GPU): adjoint identity on the device path."""
    import xdet
    rng = np.random.default_rng(3)
    n, c, h, w, r, g = 2, 490, 30, 30, 128, 7
    feat = rng.standard_normal((n, c, h, w)).astype(np.float32)
    rois = random_rois(rng, n, r)
    p0, idx = xdet.ps_roi_align(feat, rois, g, g, 'mean')
    G = rng.standard_normal(p0.shape).astype(np.float32)
    gx = xdet.ps_roi_align_grad(feat, rois, G, idx, g, g, 'mean')
    dx = rng.standard_normal(feat.shape).astype(np.float32)
    p1, _ = xdet.ps_roi_align(feat + dx, rois, g, g, 'mean')
    lhs = np.sum(gx.astype(np.float64) * dx)
    rhs = np.sum(G.astype(np.float64) * (p1.astype(np.float64) - p0))
    assert abs(lhs - rhs) <= 1e-3 * max(abs(lhs), abs(rhs))


def test_grad_errors():
    import xdet
    feat = np.zeros((1, 8, 4, 4), np.float32)
    z = np.zeros((1, 2, 4, 2), np.float32)
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align_grad(feat, np.zeros((2, 2, 4), np.float32), z, z.astype(np.int32), 2, 2, 'max')
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align_grad(feat, np.zeros((1, 2, 4), np.float32), z, z.astype(np.int32), 2, 2, 'avg')
    with pytest.raises(xdet.InvalidArgumentError):
        xdet.ps_roi_align_grad(feat, np.zeros((1, 2, 4), np.float32), z[:, :1], z.astype(np.int32), 2, 2, 'max')
    out = xdet.ps_roi_align_grad(feat, np.zeros((1, 2, 4), np.float32), z + 1, z.astype(np.int32), 2, 2, 'max')
    assert not out.any()          # all rois degenerate -> zero gradient


@pytest.mark.parametrize('method', ['max', 'mean'])
def test_randomized_sweep_against_the_c_oracle(method, oracle):
    """both forms of the kernel (one channel per lane: NCHW; two channels per lane with 8-byte
    corner loads: the net's NHWC form) against the C oracle on ~2.7e7 (element, sample) blends per method and form (1.1e8 in all): 24,000
    random ROIs of every size class on the net's 30x30x490 map -- 1-pixel, sub-bin, border-clamped, full-image,
    near-degenerate, centres on the map edge -- values AND argmax indices bit for bit."""
    import xdet
    from xdet._lib import lib, check
    from xdet.runtime import DeviceBuffer, to_device, to_host
    rng = np.random.default_rng(20260928)
    n, h, w, g = 8, 30, 30, 7
    c = 10 * g * g
    feat = rng.standard_normal((n, c, h, w)).astype(np.float32)
    r = 3000
    cy, cx = rng.uniform(0.0, 1.0, (n, r)), rng.uniform(0.0, 1.0, (n, r))
    size = np.exp(rng.uniform(np.log(1e-3), np.log(1.2), (n, r, 2)))          # 0.03 px ... larger than the image
    rois = np.stack([cy, cx, size[..., 0], size[..., 1]], -1).astype(np.float32)
    rois[:, :50, 2:] = 1.0 / 30                                                 # exactly one pixel
    rois[:, 50:60] = [0.5, 0.5, 1.0, 1.0]                                        # the whole image
    rois[:, 60:70, :2] = [0.0, 1.0]                                              # centred on a corner
    rois[:, 70:75, 2] = 0.0                                                      # degenerate: zero output
    p, i = xdet.ps_roi_align(feat, rois, g, g, method)
    po, io = oracle.ps_roi_align(feat, rois, g, g, method)
    samples = (np.floor(np.clip(rois[..., 2], 1 / 30, 1) * h / g) + 1) * (np.floor(np.clip(rois[..., 3], 1 / 30, 1) * w / g) + 1)
    print('%s: %d ROIs, ~%.2e (element, sample) blends per form' % (method, n * r, float(samples.sum()) * c))
    assert np.array_equal(p, po)
    assert np.array_equal(i, io)
    # the same ROIs through the NHWC / padded-stride form the net runs (two channels per lane)
    ldc = 512
    nhwc = np.zeros((n, h, w, ldc), np.float32)
    nhwc[..., :c] = feat.transpose(0, 2, 3, 1)
    d_f, d_r = to_device(nhwc), to_device(rois)
    d_p, d_i = DeviceBuffer(n * r * ldc * 4, zero=True), DeviceBuffer(n * r * ldc * 4, zero=True)
    check(lib().xdet_psroialign_fwd(d_f.ptr, d_r.ptr, d_p.ptr, d_i.ptr, n, c, h, w, r, g, g, 1 if method == 'max' else 0,
                                    1, ldc, ldc, 0, None))
    p2 = to_host(d_p.ptr, (n, r, ldc), np.float32)[..., :c].reshape(n, r, g * g, 10)
    i2 = to_host(d_i.ptr, (n, r, ldc), np.int32)[..., :c].reshape(n, r, g * g, 10)
    assert np.array_equal(p2, po)
    assert np.array_equal(i2, io)
