"""The PsRoiAlign oracle against the known-answer vectors for the reference's own test inputs
(cpp/PSROIPooling/test_op.py:52-81; values recorded in SURVEY.md 8c) and its NumPy twin."""
import numpy as np
import pytest

from test_gpu_psroialign import KAT, KAT_ROIS, kat_input, random_rois


@pytest.mark.parametrize('method', ['mean', 'max'])
def test_known_answer_vectors(method, oracle):
    vals, idx = KAT[method]
    for fn in (oracle.ps_roi_align, oracle.ps_roi_align_np):
        p, i = fn(kat_input(), KAT_ROIS, 2, 2, method)
        assert p.shape == (1, 3, 4, 4)
        for r in range(3):
            for b in range(4):
                assert np.all(p[0, r, b] == np.float32(vals[r][b])), (fn.__name__, r, b)
            assert np.all(i[0, r] == idx[r])


def test_hand_check_bin0_mean():
    """SURVEY.md 8c hand check: roi0/bin0 mean samples at 0.34375 and 1.03125 -> 5.125."""
    ys = xs = [0.34375, 1.03125]
    plane = np.arange(1, 26, dtype=np.float64).reshape(5, 5)
    acc = 0.
    for y in ys:
        for x in xs:
            iy, ix = int(y), int(x)
            fy, fx = y - iy, x - ix
            acc += (1 - fx) * (1 - fy) * plane[iy, ix] + (1 - fx) * fy * plane[iy + 1, ix] + \
                fx * (1 - fy) * plane[iy, ix + 1] + fx * fy * plane[iy + 1, ix + 1]
    assert acc / 4 == 5.125


@pytest.mark.parametrize('method', ['max', 'mean'])
def test_c_oracle_equals_numpy_twin(method, oracle):
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 18, 9, 11)).astype(np.float32)
    rois = random_rois(rng, 2, 12)
    p, i = oracle.ps_roi_align(feat, rois, 3, 3, method)
    p2, i2 = oracle.ps_roi_align_np(feat, rois, 3, 3, method)
    assert np.array_equal(p, p2) and np.array_equal(i, i2)
    # NHWC entry (what the fused pipeline feeds) == NCHW entry
    p3, i3 = oracle.ps_roi_align(np.ascontiguousarray(feat.transpose(0, 2, 3, 1)), rois, 3, 3, method, layout='NHWC')
    assert np.array_equal(p, p3) and np.array_equal(i, i3)
    # degenerate ROIs (h or w below FLT_MIN) pool to zero with index 0
    assert np.all(p[:, 5] == 0) and np.all(p[:, 6] == 0) and np.all(i[:, 5] == 0)


def test_light_head_shape_and_sample_counts(oracle):
    """490 = 7*7*10 channels on a 30x30 map: output [N,R,49,10]; n_h, n_w in 1..5 => index < 25."""
    rng = np.random.default_rng(4)
    feat = rng.standard_normal((1, 490, 30, 30)).astype(np.float32)
    rois = random_rois(rng, 1, 64)
    p, i = oracle.ps_roi_align(feat, rois, 7, 7, 'max')
    assert p.shape == (1, 64, 49, 10) and i.dtype == np.int32
    assert i.min() >= 0 and i.max() < 25
    full = i[0, 0]          # full-image ROI: bin 30/7 = 4.29 -> 5x5 samples
    assert full.max() > 15


@pytest.mark.parametrize('method', ['max', 'mean'])
def test_grad_is_the_adjoint_of_the_forward(method, oracle):
    """F2 pin (no reference vectors exist for the backward): PsRoiAlign is linear in `inputs` once the
    argmax samples are fixed, so <grad_output, dX> == <G, fwd(X + dX) - fwd(X)> for any dX that does not
    move an argmax ('mean': any dX).  Checked in float64-accumulated inner products."""
    rng = np.random.default_rng(11)
    n, c, h, w, r, g = 2, 36, 9, 11, 7, 3
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    rois = np.stack([rng.uniform(0.2, 0.8, (n, r)), rng.uniform(0.2, 0.8, (n, r)),
                     rng.uniform(0.1, 0.9, (n, r)), rng.uniform(0.1, 0.9, (n, r))], -1).astype(np.float32)
    rois[0, 0, 2:] = 0.          # degenerate roi: contributes nothing
    p0, idx = oracle.ps_roi_align(x, rois, g, g, method)
    G = rng.standard_normal(p0.shape).astype(np.float32)
    gx = oracle.ps_roi_align_grad(x, rois, G, idx, g, g, method)
    assert gx.shape == x.shape
    dx = (rng.standard_normal(x.shape) * (1e-4 if method == 'max' else 1.)).astype(np.float32)
    p1, idx1 = oracle.ps_roi_align(x + dx, rois, g, g, method)
    same = (idx1 == idx)
    assert same.mean() > 0.99
    lhs = np.sum(gx.astype(np.float64) * dx.astype(np.float64))
    rhs = np.sum((G.astype(np.float64) * (p1.astype(np.float64) - p0.astype(np.float64)))[same])
    # the moved-argmax elements are excluded on the rhs; remove their lhs part too
    if not same.all():
        Gm = np.where(same, G, 0).astype(np.float32)
        lhs = np.sum(oracle.ps_roi_align_grad(x, rois, Gm, idx, g, g, method).astype(np.float64) * dx)
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), 1e-3), (lhs, rhs)


def test_grad_hand_check_single_sample(oracle):
    """One roi covering one pixel-sized bin -> a single sample: its 4 bilinear weights sum to the grad."""
    x = np.zeros((1, 1, 6, 6), np.float32)
    rois = np.array([[[0.5, 0.5, 1. / 6, 1. / 6]]], np.float32)      # 1x1 px box around (3,3)
    _, idx = oracle.ps_roi_align(x, rois, 1, 1, 'max')
    gx = oracle.ps_roi_align_grad(x, rois, np.full((1, 1, 1, 1), 2., np.float32), idx, 1, 1, 'max')
    assert np.isclose(gx.sum(), 2.)
    assert np.count_nonzero(gx) <= 4
    ys, xs = np.nonzero(gx[0, 0])
    assert ys.min() >= 2 and ys.max() <= 3 and xs.min() >= 2 and xs.max() <= 3
