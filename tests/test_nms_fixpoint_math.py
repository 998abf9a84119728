"""The resolve step of the proposal NMS (csrc/proposals.hip nms_panel_kernel) is not a serial walk: it iterates
    keep_j = ok_j and not sup_j and not any(col_j & keep)
over all candidates of a panel at once until nothing changes.  These are the two facts the kernel relies on, checked on
random and adversarial suppression graphs against the greedy definition (net/xception_body.py:57-67 ->
tf.image.non_max_suppression: visit in score order, keep a candidate unless a box kept before it suppresses it):
  1. the equation has exactly one solution -- keep_j only depends on keep_i for i < j -- and it is the greedy keep set;
  2. iterating from `ok and not sup` reaches it in at most (longest dependency chain + 1) rounds, one more correct leading
     candidate per round at least.
Pure NumPy: the host-side statement of the algorithm (no GPU, no oracle import needed)."""
import numpy as np
import pytest


def greedy(upper, ok):
    n = len(ok)
    keep = np.zeros(n, bool)
    for j in range(n):
        keep[j] = ok[j] and not np.any(upper[:j, j] & keep[:j])
    return keep


def fixed_point(upper, ok, start=None):
    keep = ok.copy() if start is None else start.copy()
    for rounds in range(1, len(ok) + 2):
        new = ok & ~np.any(upper & keep[:, None], axis=0)       # column j: who suppresses me, masked by the current keep set
        if np.array_equal(new, keep):
            return keep, rounds
        keep = new
    raise AssertionError('no fixed point within n + 1 rounds')


def random_graph(rng, n, density):
    return np.triu(rng.random((n, n)) < density, 1)             # upper[i, j]: i (earlier) suppresses j


@pytest.mark.parametrize('n,density', [(64, 0.02), (256, 0.01), (512, 0.005), (512, 0.05), (200, 0.5)])
def test_fixed_point_is_the_greedy_keep_set(n, density):
    rng = np.random.default_rng(n + int(density * 1000))
    for _ in range(20):
        upper = random_graph(rng, n, density)
        ok = rng.random(n) > 0.1                                 # (suppressed by the kept list / padding lanes)
        want = greedy(upper, ok)
        got, rounds = fixed_point(upper, ok)
        assert np.array_equal(got, want)
        # any start converges to the same set: the solution is unique
        got2, _ = fixed_point(upper, ok, start=rng.random(n) > 0.5)
        assert np.array_equal(got2, want)
        assert rounds <= n + 1


def test_a_chain_needs_one_round_per_link():
    """every box suppresses only its successor: the keep set alternates and the iteration fixes one more candidate per
    round -- the worst case the kernel's round limit (panel size + 1) is sized for"""
    n = 128
    upper = np.zeros((n, n), bool)
    upper[np.arange(n - 1), np.arange(1, n)] = True
    ok = np.ones(n, bool)
    got, rounds = fixed_point(upper, ok)
    assert np.array_equal(got, np.arange(n) % 2 == 0)
    assert n // 2 <= rounds <= n + 1


def test_prefix_is_final_after_r_rounds():
    rng = np.random.default_rng(5)
    n = 300
    upper = random_graph(rng, n, 0.03)
    ok = np.ones(n, bool)
    want = greedy(upper, ok)
    keep = ok.copy()
    for r in range(1, 40):
        keep = ok & ~np.any(upper & keep[:, None], axis=0)
        assert np.array_equal(keep[:r], want[:r]), r             # the first r candidates are right after r rounds
