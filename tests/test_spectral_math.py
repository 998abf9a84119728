"""The algorithm behind csrc/spectral.hip, restated in NumPy and checked against the oracle's direct
convolution: a (15,1) SAME convolution == per-bin complex products in the L = F+14 point DFT domain of
the convolved axis, with the two real bins (DC, Nyquist) packed into bin 0 and every bin evaluated as ONE
real GEMM [Xr Xi] x [[Gr, Gi], [-Gi, Gr]].  Same tables, same packing, same block matrices as the C++ host
code (spectral_tables / spectral_weights); the HIP kernels themselves are checked on the GPU
(tests/test_gpu_e2e.py, '+spectral' cases)."""
import numpy as np
import pytest


def tables(F):
    L, NB = F + 14, (F + 14) // 2
    fwd = np.zeros((NB, 2, F)); inv = np.zeros((NB, 2, F))
    i = np.arange(F)
    fwd[0, 0] = 1.0; fwd[0, 1] = (-1.0) ** i
    inv[0, 0] = 1.0 / L; inv[0, 1] = (-1.0) ** i / L
    for b in range(1, NB):
        th = 2 * np.pi * ((b * i) % L) / L
        fwd[b, 0], fwd[b, 1] = np.cos(th), -np.sin(th)
        inv[b, 0], inv[b, 1] = 2 * np.cos(th) / L, -2 * np.sin(th) / L
    return fwd.astype(np.float32), inv.astype(np.float32)


def weights(w, F):
    """w [T, cin, cout] -> [NB, 2cin, 2cout] block matrices"""
    T, cin, cout = w.shape
    L, NB, pad = F + 14, (F + 14) // 2, T // 2
    j = (pad - np.arange(T)) % L
    out = np.zeros((NB, 2 * cin, 2 * cout), np.float64)
    out[0, :cin, :cout] = w.sum(0)
    out[0, cin:, cout:] = (w * ((-1.0) ** j)[:, None, None]).sum(0)
    for b in range(1, NB):
        th = 2 * np.pi * ((b * j) % L) / L
        gr = (w * np.cos(th)[:, None, None]).sum(0)
        gi = (w * -np.sin(th)[:, None, None]).sum(0)
        out[b, :cin, :cout], out[b, :cin, cout:] = gr, gi
        out[b, cin:, :cout], out[b, cin:, cout:] = -gi, gr
    return out.astype(np.float32)


@pytest.mark.parametrize('F', [16, 30, 50])
@pytest.mark.parametrize('axis', [1, 2])
def test_spectral_conv_equals_direct_conv(F, axis, oracle):
    rng = np.random.default_rng(F + axis)
    cin, cout, T = 24, 12, 15
    x = rng.standard_normal((2, F, F, cin)).astype(np.float32)
    w = (rng.standard_normal((T, cin, cout)) * 0.1).astype(np.float32)
    k = w[:, None] if axis == 1 else w[None]                       # (15,1) or (1,15) HWIO
    ref = oracle.conv2d(x, k, padding='SAME')
    fwd, inv = tables(F)
    Bm = weights(w, F)
    xm = np.moveaxis(x, axis, 2)                                   # [n, other, F, cin]
    re = np.einsum('bi,noic->nobc', fwd[:, 0], xm)
    im = np.einsum('bi,noic->nobc', fwd[:, 1], xm)
    A = np.concatenate([re, im], -1)                               # [n, other, NB, 2cin]
    Y = np.einsum('nobk,bkj->nobj', A, Bm)                         # one real GEMM per bin
    y = np.einsum('bi,nobc->noic', inv[:, 0], Y[..., :cout]) + np.einsum('bi,nobc->noic', inv[:, 1], Y[..., cout:])
    got = np.moveaxis(y, 2, axis)
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_mac_count_advantage():
    """per output pixel and (cin, cout) pair: 15 MACs direct vs 4 * (L/2) / F in the DFT domain"""
    for F, least in ((30, 5.0), (50, 5.8), (16, 3.9)):
        L = F + 14
        assert 15.0 / (4.0 * (L // 2) / F) > least
