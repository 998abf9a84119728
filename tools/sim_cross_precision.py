#!/usr/bin/env python
"""Accuracy of one split-precision GEMM when its two cross terms (a_lo*b_hi, a_hi*b_lo) are computed from narrower
operands (numpy model of the operand roundings only; accumulation exact).  Notes: profiles/NOTES_r04.md.
    python tools/sim_cross_precision.py"""
import numpy as np

rng = np.random.default_rng(0)


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def minifloat(x, mant, emin, emax_val):
    """round to a float with `mant` explicit mantissa bits, smallest normal 2^emin (subnormals below), saturating at emax_val"""
    x = np.asarray(x, np.float64)
    s = np.sign(x)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, emin)
    q = np.round(a / 2.0 ** (e - mant)) * 2.0 ** (e - mant)
    return s * np.minimum(q, emax_val)


def e4m3(x):
    return minifloat(x, 3, -6, 448.0)


def e2m3(x):
    return minifloat(x, 3, 0, 7.5)


def scaled(x, fmt, top, block):
    """x / s rounded to fmt, times s, with s a power of two per `block` elements along K (block=None: one per tensor)
    chosen so that the block's largest magnitude lands in (top/2, top]"""
    x = np.asarray(x, np.float64)
    out = np.empty_like(x)
    K = x.shape[-1]
    step = K if block is None else block
    for k0 in range(0, K, step):
        blk = x[..., k0:k0 + step]
        mx = np.abs(blk).max(axis=None if block is None else -1, keepdims=block is not None)
        mx = np.where(mx == 0, 1.0, mx)
        s = 2.0 ** np.ceil(np.log2(mx / top))
        out[..., k0:k0 + step] = fmt(blk / s) * s
    return out


M, K, N = 512, 736, 256
rows = []
for trial in range(3):
    # activations: a depthwise output -- per-channel scales over two decades, heavy-ish tails
    a = rng.standard_normal((M, K)) * np.exp(rng.normal(0, 1.2, (1, K))) * np.exp(rng.normal(0, 0.5, (M, 1)))
    b = rng.standard_normal((N, K)) / np.sqrt(K)
    a = a.astype(np.float32).astype(np.float64)
    b = b.astype(np.float32).astype(np.float64)
    ref = a @ b.T
    ah, bh = f16(a), f16(b)
    al, bl = f16(a - ah), f16(b - bh)
    hh = ah @ bh.T
    cases = {
        'f16x3 (today)': hh + ah @ bl.T + al @ bh.T,
        'fp8 e4m3, scale per tensor': hh + scaled(ah, e4m3, 256, None) @ scaled(bl, e4m3, 256, None).T
                                       + scaled(al, e4m3, 256, None) @ scaled(bh, e4m3, 256, None).T,
        'fp8 e4m3, scale per 32 (MX)': hh + scaled(ah, e4m3, 256, 32) @ scaled(bl, e4m3, 256, 32).T
                                        + scaled(al, e4m3, 256, 32) @ scaled(bh, e4m3, 256, 32).T,
        'fp6 e2m3, scale per 32 (MX)': hh + scaled(ah, e2m3, 7.5, 32) @ scaled(bl, e2m3, 7.5, 32).T
                                        + scaled(al, e2m3, 7.5, 32) @ scaled(bh, e2m3, 7.5, 32).T,
        'f16 product alone': hh,
    }
    sc = np.abs(ref).max()
    rows.append({k: (np.abs(v - ref).max() / sc, np.sqrt(((v - ref) ** 2).mean() / (ref ** 2).mean())) for k, v in cases.items()})
for k in rows[0]:
    print('%-30s max err / max|ref| %.2e   rms relative %.2e' % (k, np.mean([r[k][0] for r in rows]), np.mean([r[k][1] for r in rows])))
