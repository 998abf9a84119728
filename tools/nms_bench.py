"""Proposal stage (A7) alone: exactness against the oracle on overlap-heavy inputs at several batch sizes (the cluster
sizes of nms_panel_kernel) and its time per call.  python tools/nms_bench.py [--check] [--time]"""
import sys
import time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/x-detector_amd')
from xdet._lib import lib, check
from xdet.runtime import to_device, to_host, DeviceBuffer, Event, synchronize


def clustered(rng, n, cnt, centres=40, jitter=0.02, scale=0.3):
    """boxes in tight groups: many IoU > 0.7 pairs, long suppression chains"""
    c = rng.uniform(0.1, 0.9, (n, centres, 2))
    hw = rng.uniform(0.08, scale, (n, centres, 2))
    k = rng.integers(0, centres, (n, cnt))
    cy = np.take_along_axis(c[..., 0], k, 1) + rng.normal(0, jitter, (n, cnt))
    cx = np.take_along_axis(c[..., 1], k, 1) + rng.normal(0, jitter, (n, cnt))
    h = np.take_along_axis(hw[..., 0], k, 1) * np.exp(rng.normal(0, 0.1, (n, cnt)))
    w = np.take_along_axis(hw[..., 1], k, 1) * np.exp(rng.normal(0, 0.1, (n, cnt)))
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    scores = rng.uniform(0.001, 0.999, (n, cnt)).astype(np.float32)
    return scores, boxes


def spread(rng, n, cnt, scale=0.25):
    cy, cx = rng.uniform(-0.1, 1.1, (n, cnt)), rng.uniform(-0.1, 1.1, (n, cnt))
    h, w = rng.uniform(0.0, scale, (n, cnt)) + 0.01, rng.uniform(0.0, scale, (n, cnt)) + 0.01
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    return rng.uniform(0.001, 0.999, (n, cnt)).astype(np.float32), boxes


def run(scores, boxes, pre, post, thr, iters=0):
    N, n = scores.shape
    d_s, d_b = to_device(scores), to_device(boxes)
    ws = DeviceBuffer(lib().xdet_proposals_workspace_bytes(N, n, pre, post), zero=True)
    d_r, d_c = DeviceBuffer(N * post * 16), DeviceBuffer(N * 16)
    call = lambda: check(lib().xdet_get_proposals(d_s.ptr, d_b.ptr, N, n, pre, post, thr, 16. / 480, ws.ptr, d_r.ptr, d_c.ptr, None))
    call()
    rois = to_host(d_r.ptr, (N, post, 4), np.float32)
    counts = to_host(d_c.ptr, (N, 4), np.int32)
    us = None
    if iters:
        for _ in range(3):
            call()
        a, b = Event(), Event()
        a.record()
        for _ in range(iters):
            call()
        b.record()
        synchronize()
        us = a.elapsed_ms(b) * 1000 / iters
    return rois, counts, us


if __name__ == '__main__':
    rng = np.random.default_rng(7)
    if '--check' in sys.argv or len(sys.argv) == 1:
        from oracle import lighthead_oracle as O
        for N in (1, 2, 5, 12, 20, 40, 70):
            for gen, name in ((clustered, 'clustered'), (spread, 'spread')):
                for post in (300, 1000):
                    s, b = gen(rng, N, 19800)
                    rois, counts, _ = run(s, b, 5000, post, 0.7)
                    tr = []
                    ref = O.get_proposals(s[:min(N, 3)], b[:min(N, 3)], 5000, post, 0.7, 16. / 480, tr)
                    ok = np.array_equal(rois[:min(N, 3)], ref)
                    # the images the oracle did not do: every image of a batch must equal its single-image run
                    same = True
                    if N > 3:
                        r1, _, _ = run(s[N - 1:], b[N - 1:], 5000, post, 0.7)
                        same = np.array_equal(r1[0], rois[N - 1])
                    print('N=%3d %-9s post=%4d keep=%s exact=%s batch-invariant=%s' % (N, name, post, counts[:3, 2], ok, same), flush=True)
                    assert ok and same
    if '--time' in sys.argv or len(sys.argv) == 1:
        for N in (1, 8, 32, 128):
            for gen, name in ((clustered, 'clustered'), (spread, 'spread')):
                for post in (300, 1000):
                    s, b = gen(rng, N, 19800)
                    _, counts, us = run(s, b, 5000, post, 0.7, iters=20)
                    print('N=%3d %-9s post=%4d keep=%4d  get_proposals %8.1f us per call' % (N, name, post, counts[0, 2], us), flush=True)
