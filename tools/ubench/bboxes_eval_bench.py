"""Where does bboxes_eval_kernel's time go at batch 1?  (csrc/detect.hip; light_head_rfcn_eval.py:263-287)
Variants that switch phases off through the inputs: nothing valid (launch + softmax), all valid / nothing suppressed (rank sort + IoU
mask + the longest serial scan), all valid / everything suppressed by the first box (rank sort + mask + the shortest scan)."""
import ctypes
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'x-detector_amd'))
from xdet._lib import lib, check          # noqa: E402
from xdet.runtime import DeviceBuffer, Stream, to_device      # noqa: E402


def run(name, logits, boxes, thr, N=1, reps=200):
    R, nc = logits.shape[-2], logits.shape[-1]
    s = Stream()
    d_c, d_b = to_device(np.ascontiguousarray(logits, np.float32)), to_device(np.ascontiguousarray(boxes, np.float32))
    shapes = to_device(np.tile(np.array([480, 480], np.int32), (N, 1)))
    bimg = to_device(np.tile(np.array([0, 0, 1, 1], np.float32), (N, 1)))
    k = 200
    d_os, d_ob = DeviceBuffer(N * (nc - 1) * k * 4), DeviceBuffer(N * (nc - 1) * k * 16)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib().xdet_event_create(ctypes.byref(e0)))
    check(lib().xdet_event_create(ctypes.byref(e1)))

    def go():
        check(lib().xdet_bboxes_eval(d_c.ptr, nc, d_b.ptr, N, R, nc, shapes.ptr, bimg.ptr, 480, 480, ctypes.c_float(thr),
                                     ctypes.c_float(0.3), k, d_os.ptr, d_ob.ptr, s.handle))
    for _ in range(10):
        go()
    check(lib().xdet_event_record(e0, s.handle))
    for _ in range(reps):
        go()
    check(lib().xdet_event_record(e1, s.handle))
    ms = ctypes.c_float()
    check(lib().xdet_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    print(f'{name:45s} {ms.value / reps * 1000:8.1f} us per launch (back to back, N={N})')


def main():
    rng = np.random.RandomState(0)
    R, nc = 300, 21
    logits = rng.standard_normal((1, R, nc)).astype(np.float32) * 0.01
    # disjoint boxes: a 20 x 15 grid of small cells
    gy, gx = np.divmod(np.arange(R), 20)
    disjoint = np.stack([gy / 16 + 0.01, gx / 21 + 0.01, gy / 16 + 0.05, gx / 21 + 0.04], 1)[None].astype(np.float32)
    same = np.tile(np.array([0.2, 0.2, 0.8, 0.8], np.float32), (1, R, 1)) + rng.uniform(0, 0.01, (1, R, 4)).astype(np.float32)
    rand = np.sort(rng.uniform(0, 1, (1, R, 2, 2)).astype(np.float32), axis=2).transpose(0, 1, 2, 3).reshape(1, R, 4)
    rand = np.stack([rand[..., 0], rand[..., 1], rand[..., 2], rand[..., 3]], -1)
    run('nothing valid (thr 0.99)', logits, disjoint, 0.99)
    run('all valid, disjoint boxes (200 kept)', logits, disjoint, 0.01)
    run('all valid, one cluster (1 kept)', logits, same, 0.01)
    run('all valid, random boxes', logits, rand, 0.01)
    for N in (8, 128):
        run('all valid, random boxes', np.tile(logits, (N, 1, 1)), np.tile(rand, (N, 1, 1)), 0.01, N=N, reps=50)


if __name__ == '__main__':
    main()
