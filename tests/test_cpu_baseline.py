"""The C++/OpenMP CPU baseline (oracle/lighthead_cpu.cpp, what bench.py times as `cpu_baseline`) is itself checked
against the NumPy oracle: same detections within 1e-3 on a seeded 480x480 image -- a baseline that computes
something else would make the reported CPU number meaningless."""
import numpy as np


def test_cpp_baseline_matches_the_numpy_oracle(oracle, lh_weights):
    from xdet import weights as W
    img = W.synthetic_images(1, 480, seed=11)
    fwd = oracle.CppForward(lh_weights, 480, 300)
    assert fwd.threads >= 1
    got = fwd(img)
    ref = oracle.lighthead_forward(img, lh_weights, rpn_post_nms_top_n=300)
    total = matched = extra = 0
    for c in range(1, 21):
        gs, gb = got[0][c]
        rs, rb = ref[0][c]
        kg, kr = int((gs > 0).sum()), int((rs > 0).sum())
        used = np.zeros(kg, bool)
        for j in range(kr):
            d = np.where(used, np.inf, np.maximum(np.abs(gs[:kg] - rs[j]), np.abs(gb[:kg] - rb[j]).max(1))) if kg else np.array([np.inf])
            if d.min() < 1e-3:
                used[int(d.argmin())] = True
                matched += 1
        total += kr
        extra += kg - int(used.sum())
    assert total > 100 and matched == total and extra == 0, (total, matched, extra)
    # the bench entry point caches the packed weights
    again = oracle.lighthead_forward_fast(img, lh_weights, rpn_post_nms_top_n=300)
    assert np.array_equal(again[0][1][0], got[0][1][0])


def test_image_parallel_form_equals_layer_parallel_form(oracle, lh_weights):
    """a call with at least one image per two threads runs image-parallel (every thread takes whole images through the
    graph); smaller calls run layer-parallel.  Same arithmetic per image, so the same bits -- and the thread sweep picks
    one of its candidates."""
    from xdet import weights as W
    imgs = W.synthetic_images(2, 256, seed=12)
    fwd = oracle.CppForward(lh_weights, 256, 100)
    fwd.set_threads(2)
    a = fwd(imgs)                       # 2 images, 2 threads: image-parallel
    fwd.set_threads(8)
    b = fwd(imgs)                       # 2 images, 8 threads: layer-parallel
    n_det = 0
    for i in range(2):
        for c in range(1, 21):
            assert np.array_equal(a[i][c][0], b[i][c][0]) and np.array_equal(a[i][c][1], b[i][c][1]), (i, c)
            n_det += int((a[i][c][0] > 0).sum())
    assert n_det > 20
    seen = fwd.tune_threads(imgs, candidates=(1, 2))
    assert set(seen) == {1, 2} and fwd.threads in seen
