"""The importer / editor helpers (round 4): glTF curve simplification and BlendSpace triangulation.

* `curve_simplify` -- gltf/simplify.rs:39-140, pinned by the reference's 14 #[test]s (tests/golden/fyrox_unit_vectors.json
  "curve_simplify"): the oracle (C), the second restatement (oracle2, numpy) and the PRODUCT's host function all give the
  reference's indices, and the three agree on random curves with the importer's own parameters (gltf/animation.rs:50-65).
* `blend_space_triangulate` -- blendspace.rs:416-447 delegates to the `spade` crate (absent): oracle and product are pinned on the
  reference's one fixture (blendspace.rs:455-484) and checked for the properties of a Delaunay triangulation on random inputs."""
import json
import math
import os

import numpy as np
import pytest

import oracle
import oracle2.curve as o2c
from fyrox_amd import anim as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "fyrox_unit_vectors.json")))


def _ms(v):
    return float("inf") if v == "inf" else float(v)


@pytest.mark.parametrize("case", GOLD["curve_simplify"]["cases"], ids=lambda c: c["name"])
def test_simplify_reference_vectors(case):
    pts = case["points"]
    x, y = [p[0] for p in pts], [p[1] for p in pts]
    want = case["expect"]
    assert list(oracle.find_important_points(x, y, case["epsilon"], _ms(case["max_step"]))) == want
    assert o2c.find_important_points(pts, case["epsilon"], _ms(case["max_step"])) == want
    assert list(A.curve_simplify(x, y, case["epsilon"], _ms(case["max_step"]))) == want


def test_simplify_three_implementations_agree_on_importer_shaped_curves():
    """Sampled channels the way a glTF clip has them (30 - 240 keys, smooth motion + noise + flat stretches + steps) under the four
    bindings' parameters: Position 0.001 / inf, Rotation pi/180 / pi/4, Scale 0.1 / inf, morph weights 0.001 / inf."""
    rng = np.random.default_rng(20260923)
    params = [(0.001, float("inf")), (math.pi / 180.0, math.pi / 4.0), (0.1, float("inf")), (0.001, float("inf"))]
    kept_total = 0
    for trial in range(120):
        n = int(rng.integers(1, 240))
        t = np.cumsum(rng.uniform(0.01, 0.1, n)).astype(np.float32) if trial % 3 else (np.arange(n, dtype=np.float32) / np.float32(30.0))
        y = np.sin(t * rng.uniform(0.5, 6.0)) * rng.uniform(0.0, 3.0) + rng.normal(0, rng.choice([0.0, 1e-4, 1e-2]), n)
        if trial % 4 == 0 and n > 10:
            y[n // 3: n // 2] = y[n // 3]                       # a flat stretch
        if trial % 5 == 0 and n > 10:
            y[n // 2:] += 2.0                                   # a step
        y = y.astype(np.float32)
        eps, ms = params[trial % 4]
        a = list(oracle.find_important_points(t, y, eps, ms))
        b = o2c.find_important_points(list(zip(t, y)), eps, ms)
        c = list(A.curve_simplify(t, y, eps, ms))
        assert a == b == c, (trial, n, eps, ms)
        assert a[0] == 0 and (len(a) == 1 or a[-1] == n - 1) and a == sorted(set(a))
        kept_total += len(a)
    assert kept_total > 500


def test_simplify_ties_and_thresholds():
    """find_points_in_span's two comparisons on values a float holds exactly (gltf/simplify.rs:118-131): `far_point_dist < dist` keeps
    the FIRST of equally far points, `far_point_dist < epsilon` keeps a point that is EXACTLY epsilon off the line.  Expected indices by
    hand from the reference's text; the three implementations agree with them."""
    cases = [
        # (points, epsilon, expected)
        ([(0.0, 0.0), (1.0, 0.5), (2.0, 0.0)], 0.5, [0, 1, 2]),                  # 0.5 < 0.5 is false: kept
        ([(0.0, 0.0), (1.0, 0.5), (2.0, 0.0)], 0.5000001, [0]),                  # dropped; the two ends are then one value: one key (simplify.rs:62-64)
        ([(0.0, 0.0), (1.0, 2.0), (2.0, 2.0)], 1.0000001, [0, 2]),              # 1.0 off the chord, dropped; the ends differ by 2.0: two keys
        ([(0.0, 0.0), (1.0, 2.0), (2.0, 2.0)], 1.0, [0, 1, 2]),
        # keys 1 and 2 both lie 1.0 off the chord: key 1 wins, then key 2 is 0.5 off (1, 1) - (3, 0): dropped at 0.75.
        # (had key 2 won, key 1 would be 0.5 off (0, 0) - (2, 1) and the answer [0, 2, 3])
        ([(0.0, 0.0), (1.0, 1.0), (2.0, 1.0), (3.0, 0.0)], 0.75, [0, 1, 3]),
        ([(0.0, 0.0), (1.0, 1.0), (2.0, 1.0), (3.0, 0.0)], 0.5, [0, 1, 2, 3]),   # ... and kept when 0.5 is the threshold itself
    ]
    for pts, eps, want in cases:
        x, y = [p[0] for p in pts], [p[1] for p in pts]
        assert list(oracle.find_important_points(x, y, eps, float("inf"))) == want, (pts, eps)
        assert o2c.find_important_points(pts, eps, float("inf")) == want, (pts, eps)
        assert list(A.curve_simplify(x, y, eps, float("inf"))) == want, (pts, eps)


def test_simplify_argument_errors():
    from fyrox_amd import _native
    lib = _native.lib()
    import ctypes
    n = ctypes.c_uint32(7)
    assert lib.fyx_curve_simplify(None, None, 0, ctypes.c_float(0.1), ctypes.c_float(1.0), None, ctypes.byref(n)) == 0 and n.value == 0
    assert lib.fyx_curve_simplify(None, None, 3, ctypes.c_float(0.1), ctypes.c_float(1.0), None, ctypes.byref(n)) == _native.FYX_ERR_INVALID_ARG
    assert lib.fyx_curve_simplify(None, None, 0, ctypes.c_float(0.1), ctypes.c_float(1.0), None, None) == _native.FYX_ERR_INVALID_ARG


def _area2(p, t):
    a, b, c = p[t[0]], p[t[1]], p[t[2]]
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


def _hull_area2(p):
    pts = sorted(set(map(tuple, p.tolist())))
    if len(pts) < 3:
        return 0.0

    def half(seq):
        h = []
        for q in seq:
            while len(h) >= 2 and (h[-1][0] - h[-2][0]) * (q[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (q[0] - h[-2][0]) <= 0:
                h.pop()
            h.append(q)
        return h
    hull = half(pts)[:-1] + half(pts[::-1])[:-1]
    return sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1] for i in range(len(hull)))


def test_triangulation_reference_fixture():
    g = GOLD["blend_space_triangulation"]
    pts = np.asarray(g["points"], np.float32)
    assert oracle.blend_space_triangulate(pts).tolist() == g["triangles"]
    assert A.blend_space_triangulate(pts).tolist() == g["triangles"]


def test_triangulation_is_delaunay_and_both_implementations_agree():
    rng = np.random.default_rng(77)
    for trial in range(60):
        n = int(rng.integers(3, 24))
        if trial % 6 == 0:       # a lattice: co-circular quadruples, collinear triples
            g = np.stack(np.meshgrid(np.arange(4.0), np.arange(3.0)), -1).reshape(-1, 2)
            p = g[rng.permutation(len(g))[:max(n, 4)]].astype(np.float32)
        else:
            p = rng.uniform(-2, 2, (n, 2)).astype(np.float32)
        if trial % 7 == 0:
            p = np.concatenate([p, p[:2]])                      # repeated points add nothing
        a = oracle.blend_space_triangulate(p)
        b = A.blend_space_triangulate(p)
        assert a.tolist() == b.tolist(), trial
        pd = p.astype(np.float64)
        total = 0.0
        for t in a:
            assert t[0] > t[1] and t[0] > t[2]                  # newest point first
            ar = _area2(pd, t)
            assert ar > 0                                        # counter-clockwise, no slivers of zero area
            total += ar
            A_, B_, C_ = pd[t[0]], pd[t[1]], pd[t[2]]
            for q in range(len(pd)):
                if q in t or any((pd[q] == pd[k]).all() for k in t):
                    continue
                ax, ay, bx, by, cx, cy = *(A_ - pd[q]), *(B_ - pd[q]), *(C_ - pd[q])
                det = (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay)
                assert det <= 1e-9 * max(1.0, abs(ar)) , (trial, t, q)      # no point strictly inside a circumcircle
        assert abs(total - _hull_area2(pd)) <= 1e-6 * max(total, 1.0)      # the triangles tile the convex hull
        assert [tuple(x) for x in a.tolist()] == sorted(tuple(x) for x in a.tolist())


def test_triangulation_degenerate_inputs():
    assert A.blend_space_triangulate(np.zeros((0, 2), np.float32)).shape == (0, 3)
    assert A.blend_space_triangulate([[0, 0], [1, 1]]).shape == (0, 3)                 # fewer than three points: triangulate() is false
    assert A.blend_space_triangulate([[0, 0], [1, 0], [2, 0], [3, 0]]).shape == (0, 3)  # collinear: no inner face
    assert oracle.blend_space_triangulate([[0, 0], [1, 0], [2, 0], [3, 0]]).shape == (0, 3)
    with pytest.raises(ValueError):
        A.blend_space_triangulate([[0, 0], [1, float("nan")], [2, 0]])
    assert oracle.blend_space_triangulate([[0, 0], [1, float("inf")], [2, 0]]) is None


def test_triangulated_blend_space_drives_fetch_weights():
    """The product's triangles fed to the oracle's fetch_weights (blendspace.rs:338-414): inside the hull the three weights sum to 1
    and reproduce the sampling point."""
    rng = np.random.default_rng(5)
    p = rng.uniform(-1, 1, (9, 2)).astype(np.float32)
    tri = A.blend_space_triangulate(p)
    centre = p.mean(0)
    w = oracle.blend_space_fetch_weights(p, tri.astype(np.uint32), tuple(centre))
    assert w is not None and abs(sum(x for _, x in w) - 1.0) < 1e-5
    back = sum(np.float64(x) * p[i].astype(np.float64) for i, x in w)
    assert np.allclose(back, centre, atol=1e-5)
