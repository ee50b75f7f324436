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


def find_important_points(points, epsilon, max_step=float("inf")):
    """gltf/simplify.rs:39-140 in numpy float32, written from the Rust source (recursive, as the reference): the indices of the points
    the glTF importer keeps."""
    f = np.float32
    xs = [f(p[0]) for p in points]
    ys = [f(p[1]) for p in points]
    n = len(points)
    if n == 0:
        return []
    keep = [False] * n
    keep[0] = keep[n - 1] = True
    eps, ms = f(epsilon), f(max_step)

    def span(start, end):
        if end <= start + 1:
            return
        x0, y0 = xs[start], ys[start]
        with np.errstate(divide="ignore", invalid="ignore"):
            slope = f(f(ys[end] - y0) / f(xs[end] - x0))
        far_i, far_d = 0, f(0.0)
        for i in range(start + 1, end):
            y_line = f(y0 + f(slope * f(xs[i] - x0)))
            dist = f(abs(f(ys[i] - y_line)))
            if far_d < dist:
                far_d, far_i = dist, i
        if far_i == 0 or far_d < eps:
            return
        keep[far_i] = True
        span(start, far_i)
        span(far_i, end)

    span(0, n - 1)
    if np.isfinite(ms):
        def find_step(start):
            sy = ys[start]
            for i in range(start + 1, n):
                if f(abs(f(ys[i] - sy))) > ms:
                    return max(i - 1, start + 1)
                if keep[i]:
                    return i
            return n - 1
        i = 1
        while i < n - 1:
            if keep[i]:
                i += 1
            else:
                nxt = find_step(i - 1)
                keep[nxt] = True
                i = max(nxt + 1, i + 1)
    res = [i for i, k in enumerate(keep) if k]
    if len(res) == 2 and f(abs(f(ys[res[0]] - ys[res[1]]))) < eps:
        res.pop()
    return res
