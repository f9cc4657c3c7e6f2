"""CPU known-answer tests of the oracle's pointnet2_ops restatement (FPS / ball query), no GPU.

pointnet2_ops v3.2.0 is not in /root/reference (parity unpinned); these cases pin the
restatement to hand-derived answers of the published kernels' semantics.
"""
import numpy as np


from fractions import Fraction


def _round_f32(fr: Fraction) -> np.float32:
    """Correctly rounded (nearest even) float32 of a positive rational in the normal range."""
    e = 0
    while Fraction(2) ** (e + 1) <= fr:
        e += 1
    while Fraction(2) ** e > fr:
        e -= 1
    scaled = fr / Fraction(2) ** (e - 23)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return np.float32(float(Fraction(n) * Fraction(2) ** (e - 23)))


def _mag(x, y) -> np.float32:
    """fma(0,0,fma(y,y,x*x)) in exact arithmetic = the kernel's |p|^2 for z = 0."""
    xx = _round_f32(Fraction(float(x)) ** 2)
    return _round_f32(Fraction(float(y)) ** 2 + Fraction(float(xx)))


def _point_with_mag(target: np.float32):
    x = np.float32(0.03)
    y = np.float32(np.sqrt(float(target) - float(np.float32(x * x))))
    lo = y
    for _ in range(64):
        lo = np.nextafter(lo, np.float32(0))
    for _ in range(128):
        if _mag(x, lo) == target:
            return x, lo
        lo = np.nextafter(lo, np.float32(1))
    raise AssertionError("no float pair with that squared norm")


def threshold_points():
    """-> (cloud [2,4,3], expected FPS indices [2,3]).

    Upstream skips a point with ``if (mag <= 1e-3) continue;`` -- ``mag`` is a float, the literal a double
    (0.001 < 1e-3f = 0.0010000000475), so a point with mag == 1e-3f exactly IS a candidate and the next float
    below is not."""
    t = np.float32(1e-3)
    on = _point_with_mag(t)
    below = _point_with_mag(np.nextafter(t, np.float32(0)))
    assert float(np.nextafter(t, np.float32(0))) <= 1e-3 < float(t)
    far = np.float32(1.0)
    # env 0: start point far away, then the on-threshold point (kept) and the below-threshold point (skipped)
    e0 = np.array([[far, 0, 0], [on[0], on[1], 0], [below[0], below[1], 0], [0, 0, 0]], np.float32)
    # env 1: only the below-threshold point and the origin beside the start point: nothing is ever a candidate but
    # the start point itself
    e1 = np.array([[far, 0, 0], [below[0], below[1], 0], [0, 0, 0], [below[1], below[0], 0]], np.float32)
    # picks: env 0 -> 0, then 1 (the only other candidate; d = (1-x)^2), then the larger running distance of
    # {0: 0, 1: 0} -> first strictly greater in scan order.  With bs = 4 every thread holds one point; ties in the
    # tree reduce keep the lower slot -> index 0.
    return np.stack([e0, e1]), np.array([[0, 1, 0], [0, 0, 0]], np.int32)


def test_fps_skip_threshold(oracle):
    x, expect = threshold_points()
    np.testing.assert_array_equal(oracle.fps(x, 3), expect)


def test_fps_start_and_empty(oracle):
    z = np.zeros((1, 70, 3), np.float32)
    assert (oracle.fps(z, 5) == 0).all()  # nothing is a candidate: besti stays 0


def test_ball_query_first_hits_and_padding(oracle):
    xyz = np.zeros((1, 6, 3), np.float32)
    xyz[0, :, 0] = [0.0, 0.05, 0.2, 0.09, 5.0, 0.01]
    q = np.zeros((1, 2, 3), np.float32)
    q[0, 1, 0] = 10.0
    idx, cnt = oracle.ball_query(q, xyz, 0.1, 3, return_counts=True)
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 3])  # first three in index order; index 5 never reached
    np.testing.assert_array_equal(idx[0, 1], [0, 0, 0])  # no hit: zero-initialised
    np.testing.assert_array_equal(cnt[0], [3, 0])
    idx, cnt = oracle.ball_query(q, xyz, 0.06, 4, return_counts=True)
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 5, 0])  # padded with the first hit
    assert cnt[0, 0] == 3
