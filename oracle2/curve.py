"""fyrox-math Curve (fyrox-math/src/curve.rs) and the scalar helpers of fyrox-math/src/lib.rs, on float32 scalars."""
import numpy as np

from .na import F, ONE, TWO, ZERO

CONSTANT, LINEAR, CUBIC = 0, 1, 2
THREE = F(3.0)


def lerpf(a, b, t):
    """fyrox-math/src/lib.rs:206-208"""
    return a + (b - a) * t


def cubicf(p0, p1, t, m0, m1):
    """fyrox-math/src/lib.rs:211-221 (cubic Hermite; tangents scaled by |p1 - p0|)"""
    t2 = t * t
    t3 = t2 * t
    scale = abs(p1 - p0)
    return ((TWO * t3 - THREE * t2 + ONE) * p0
            + (t3 - TWO * t2 + t) * m0 * scale
            + (-TWO * t3 + THREE * t2) * p1
            + (t3 - t2) * m1 * scale)


def stepf(p0, p1, t):
    """curve.rs:25-31"""
    return p1 if t == ONE else p0


def wrapf(n, min_limit, max_limit):
    """fyrox-math/src/lib.rs:179-203"""
    if n >= min_limit and n <= max_limit:
        return n
    if max_limit == ZERO and min_limit == ZERO:
        return ZERO
    max_limit = max_limit - min_limit
    offset = min_limit
    min_limit = ZERO
    n = n - offset
    num_of_max = F(np.floor(abs(n / max_limit)))
    if n >= max_limit:
        n = n - num_of_max * max_limit
    elif n < min_limit:
        n = n + (num_of_max + ONE) * max_limit
    return n + offset


def clampf(x, lo, hi):
    """f32::clamp"""
    if x < lo:
        return lo
    if x > hi:
        return hi
    return x


class Key:
    __slots__ = ("location", "value", "kind", "left_tangent", "right_tangent")

    def __init__(self, location, value, kind=LINEAR, left_tangent=0.0, right_tangent=0.0):
        self.location, self.value, self.kind = F(location), F(value), int(kind)
        self.left_tangent, self.right_tangent = F(left_tangent), F(right_tangent)


def interpolate(left: Key, right: Key, t):
    """curve.rs:87-132: the LEFT key's kind picks the interpolation; the right key contributes its left tangent only when
    both are cubic."""
    if left.kind == CONSTANT:
        return stepf(left.value, right.value, t)
    if left.kind == LINEAR:
        return lerpf(left.value, right.value, t)
    m1 = right.left_tangent if right.kind == CUBIC else ZERO
    return cubicf(left.value, right.value, t, left.right_tangent, m1)


class Curve:
    """Keys sorted by location (Curve::from sorts stably, curve.rs:170-192)."""

    def __init__(self, keys=()):
        self.keys = sorted(keys, key=lambda k: k.location)

    def value_at(self, location, hint: int):
        """curve.rs:254-314 (fetch_at + value_at).  Returns (value, new hint)."""
        location = F(location)
        ks = self.keys
        if not ks:
            return ZERO, hint
        first, last = ks[0], ks[-1]
        if location <= first.location:
            return first.value, 0
        if location >= last.location:
            return last.value, max(len(ks) - 1, 0)
        li = max(hint - 1, 0)
        if li < len(ks) and hint < len(ks):
            pl, pr = ks[li], ks[hint]
            if location >= pl.location and location < pr.location:
                t = (location - pl.location) / (pr.location - pl.location)
                return interpolate(pl, pr, t), hint
        # partition_point(|k| k.location < location)
        lo, hi = 0, len(ks)
        while lo < hi:
            mid = lo + (hi - lo) // 2
            if ks[mid].location < location:
                lo = mid + 1
            else:
                hi = mid
        hint = lo
        left, right = ks[max(hint - 1, 0)], ks[hint]
        t = (location - left.location) / (right.location - left.location)
        return interpolate(left, right, t), hint
