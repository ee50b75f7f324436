"""Differential test of the two CPU restatements: oracle/ (C, the checker of the GPU kernels) against oracle2/ (numpy
float32, written from the Rust sources only, other data structures).  Any differing BIT fails: both follow the same
reference arithmetic and both reach libm's sinf / cosf for Euler tracks.  oracle2 is also pinned on its own against the
reference's golden vectors.  CPU only; no product code is involved."""
import json
import os

import numpy as np
import pytest

import oracle
import oracle2
from oracle2 import curve as c2
from oracle2 import na

import anim_cases as cases
from fyrox_amd import anim as A
from fyrox_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fyrox_unit_vectors.json")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, what
    # NaN payloads aside, every bit; -0.0 == 0.0 is accepted (a sum of zero terms has no defined sign in either)
    ok = (bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))
    assert ok.all(), f"{what}: {int((~ok).sum())} of {ok.size} values differ, first at {np.argwhere(~ok)[0]}: {a[~ok][0]!r} vs {b[~ok][0]!r}"


# ---- linear-blend skinning, palettes, transforms ------------------------------------------------------------------------

@pytest.mark.parametrize("seed", range(6))
def test_lbs_random_palettes_incl_projective(seed):
    rng = np.random.default_rng(1000 + seed)
    nb = int(rng.integers(1, 257))
    m = synth.make_mesh(int(rng.integers(1, 3000)), nb, 77 + seed, coherent=bool(seed % 2))
    pal = synth.make_palette(nb, 77 + seed).copy()
    if seed % 3 == 0:                                   # general 4x4 matrices: the homogeneous divide, n == 0 included
        pal = rng.normal(size=(nb, 16)).astype(np.float32)
        pal[0, 3] = pal[0, 7] = pal[0, 11] = pal[0, 15] = 0.0
    if seed % 3 == 1:
        pal[nb // 2, 12] = np.inf                       # inf * 0 = NaN must come out of both
    a = oracle.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=1)
    b = oracle2.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent)
    for k in a:
        same(a[k], b[k], f"lbs {k}")


@pytest.mark.parametrize("seed", range(8))
def test_local_transform_hierarchy_and_palette(seed):
    """Transforms with every field in use (pivots, offsets, pre-rotation, post-rotation matrix), random trees."""
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 40))
    ts = []
    for i in range(n):
        t = A.Transform.identity()
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t.local_rotation[:] = [float(x) for x in q.astype(np.float32)]
        t.local_position[:] = [float(x) for x in rng.normal(size=3).astype(np.float32)]
        t.local_scale[:] = [float(x) for x in rng.uniform(0.3, 2.0, size=3).astype(np.float32)]
        if seed % 2:
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            t.pre_rotation[:] = [float(x) for x in q.astype(np.float32)]
            t.post_rotation_matrix[:] = [float(x) for x in rng.normal(size=9).astype(np.float32)]
            for f in ("rotation_offset", "rotation_pivot", "scaling_offset", "scaling_pivot"):
                getattr(t, f)[:] = [float(x) for x in rng.normal(size=3).astype(np.float32)]
        ts.append(t)
    parent = np.asarray([-1] + [int(rng.integers(0, i)) for i in range(1, n)], np.int32)
    inv_bind = rng.normal(size=(n, 16)).astype(np.float32)
    rig = A.Rig(parent=parent, transforms=ts, inv_bind=inv_bind)
    s1, s2 = oracle.AnimScene(rig), oracle2.AnimScene(rig)
    same(s1.local_matrices(), s2.local_matrices(), "local matrices")
    same(s1.global_matrices(), s2.global_matrices(), "global matrices")
    bones = [int(x) for x in rng.integers(-1, n, size=20)]
    same(s1.palette(bones), s2.palette(bones), "palette")
    s1.close()


# ---- the animation path: every scenario of the suite, frame by frame ------------------------------------------------------

def _compare_frame(o1, o2, sc, f):
    gone = {a for fr, lst in sc.removals.items() if fr <= f for a in lst}
    for a in range(len(sc.animations)):
        if a in gone:
            continue
        for view in ("apply", "read"):      # (a node's pose is a list: what it applies, what a blend reads of it)
            same(o1.animation_pose(a, view), o2.animation_pose(a, view), f"{sc.name} frame {f}: animation {a} pose, {view} view")
        s1, s2 = o1.animation_state(a), o2.animation_state(a)
        assert s1["enabled"] == s2["enabled"] and s1["has_ended"] == s2["has_ended"], (sc.name, f, a)
        same([s1["time_position"]], [s2["time_position"]], f"{sc.name} frame {f}: animation {a} time")
        same(o1.animation_root_motion(a), o2.animation_root_motion(a), f"{sc.name} frame {f}: animation {a} root motion")
        while True:
            e1, e2 = o1.pop_event(a), o2.pop_event(a)
            assert e1 == e2, (sc.name, f, a, e1, e2)
            if e1 is None:
                break
    same(o1.node_trs(), o2.node_trs(), f"{sc.name} frame {f}: node TRS")
    same(o1.global_matrices(), o2.global_matrices(), f"{sc.name} frame {f}: global matrices")
    if sc.machine is not None:
        same(o1.machine_pose(), o2.machine_pose(), f"{sc.name} frame {f}: machine pose")
        for li in range(len(sc.machine.layers)):
            assert tuple(o1.layer_state(li)) == tuple(o2.layer_state(li)), (sc.name, f, li)
            same(o1.machine_root_motion(li), o2.machine_root_motion(li), f"{sc.name} frame {f}: layer {li} root motion")
            while True:
                e1, e2 = o1.pop_layer_event(li), o2.pop_layer_event(li)
                assert (None if e1 is None else tuple(e1)) == (None if e2 is None else tuple(e2)), (sc.name, f, li, e1, e2)
                if e1 is None:
                    break
        same(o1.machine_root_motion(-1), o2.machine_root_motion(-1), f"{sc.name} frame {f}: machine root motion")
    assert set(o1.props) == set(o2.props), (sc.name, f)
    for k, (kind1, v1) in o1.props.items():
        kind2, v2 = o2.props[k]
        lanes = {"real": 1, "v2": 2, "v3": 3, "v4": 4, "quat": 4}[kind2]
        assert {0: "real", 1: "v2", 2: "v3", 3: "v4", 4: "quat"}[int(kind1)] == kind2, (sc.name, f, k)
        same(np.asarray(v1[:lanes]), np.asarray(v2), f"{sc.name} frame {f}: property {k}")


def _run(sc, frames=None, probe=None):
    o1, o2 = cases.build_oracle(oracle, sc), cases.build_oracle(oracle2, sc)
    for f in range(sc.n_frames if frames is None else frames):
        for idx, par in sc.script.get(f, []):
            o1.set_parameter(idx, par)
            o2.set_parameter(idx, par)
        for a in sc.removals.get(f, []):
            o1.remove_animation(a)
            o2.remove_animation(a)
        for o in (o1, o2):
            if sc.machine is None:
                o.update_animations(sc.dt)
            else:
                o.update_machine(sc.dt)
        _compare_frame(o1, o2, sc, f)
        if probe is not None:
            probe(o1)
    o1.close()


SCENARIOS = [cases.random_attacks, cases.c5_blend_tree, cases.player_only, cases.transitions, cases.by_index, cases.blend_space, cases.layered,
             cases.fbx_like, cases.gltf_like, cases.morph_weights, cases.morph_weights_player, cases.property_kinds,
             cases.property_kinds_euler, cases.property_kinds_player, cases.removed_clips, cases.looping_root_motion,
             cases.with_root_motion_and_signals(cases.c5_blend_tree), cases.with_root_motion_and_signals(cases.transitions),
             cases.with_root_motion_and_signals(cases.layered), cases.duplicate_bindings, cases.duplicate_bindings_player, cases.duplicate_properties,
             cases.with_root_motion_and_signals(cases.duplicate_bindings)]


@pytest.mark.parametrize("make", SCENARIOS, ids=lambda m: getattr(m, "__name__", "rm"))
def test_scenarios_bit_for_bit(make):
    sc = make()
    _run(sc, frames=min(sc.n_frames, 40))


@pytest.mark.parametrize("make", cases.SUBNORMAL, ids=lambda m: m.__name__)
def test_scenarios_with_subnormal_values_bit_for_bit(make):
    """Positions and scales below 2^-126 (with_subnormal_values): numpy float32 and the C oracle both keep subnormal numbers, as Rust does;
    the poses must actually hold some."""
    sc = make()
    seen = [0, 0]

    def probe(o):
        for k, m in enumerate((o.local_matrices(), o.global_matrices())):
            seen[k] += int(((np.abs(m) < 1.17e-38) & (m != 0)).sum())

    _run(sc, frames=min(sc.n_frames, 40), probe=probe)
    assert seen[0] > 100 and seen[1] > 100, seen


@pytest.mark.parametrize("seed", range(40))
def test_random_machines_bit_for_bit(seed):
    _run(cases.random_machine(seed))


@pytest.mark.parametrize("seed", range(int(os.environ.get("FYX_FUZZ_SEEDS", 16))))
def test_random_machines_with_lists_of_values_bit_for_bit(seed):
    """random_machine(listy=True): further tracks on one (node, binding), kinds that fit nothing, property tracks of every kind."""
    _run(cases.random_machine(seed, listy=True))


@pytest.mark.parametrize("seed", range(12))
def test_random_machines_on_a_lattice_bit_for_bit(seed):
    """random_machine(lattice=True): blend-space points and sampling points on exactly representable coordinates -- sampling points ON the
    triangles' corners and edges, coinciding points, degenerate triangles (0 / 0 in get_barycentric_coords_2d: NaN weights on both sides)."""
    _run(cases.random_machine(seed, listy=bool(seed % 2), lattice=True))


@pytest.mark.parametrize("seed", range(int(os.environ.get("FYX_FUZZ_SEEDS", 16))))
def test_random_curves_bit_for_bit(seed):
    """random_curves: coinciding keys, mixed key kinds, empty and single-key curves, curves of a track on different time grids."""
    _run(cases.random_curves(seed))


# ---- oracle2 on its own against the reference's golden vectors ------------------------------------------------------------

def test_oracle2_against_the_reference_vectors():
    """The blocks of tests/golden/fyrox_unit_vectors.json (the reference's own #[test] vectors) that this restatement
    covers: curves, key interpolation, wrapf, quat_from_euler == from_euler_angles, hierarchy propagation, fetch_weights."""
    g = json.load(open(GOLDEN))
    F = na.F
    for case in g["curve_value_at"]["cases"]:
        cv = c2.Curve([c2.Key(*k) for k in case["keys"]])
        for loc, want in case["fetch"]:
            got, _ = cv.value_at(F(loc), 0)
            assert got == F(want), (case, loc, got, want)
    keys = {name: c2.Key(0.0, v[0], v[1], v[2], v[3]) for name, v in g["curve_key_interpolate"]["keys"].items()}
    for left, right, t, want in g["curve_key_interpolate"]["cases"]:
        assert c2.interpolate(keys[left], keys[right], F(t)) == F(want), (left, right, t)
    for n, lo, hi, want in g["wrapf"]["cases"]:
        assert c2.wrapf(F(n), F(lo), F(hi)) == F(want)
    # quat_from_euler((pi, pi, pi), XYZ) == from_euler_angles(pi, pi, pi) with exact f32 equality
    e = [F(x) for x in g["quat_from_euler"]["euler"]]
    qx = na.q_from_axis_angle((na.ONE, na.ZERO, na.ZERO), e[0])
    qy = na.q_from_axis_angle((na.ZERO, na.ONE, na.ZERO), e[1])
    qz = na.q_from_axis_angle((na.ZERO, na.ZERO, na.ONE), e[2])
    q = na.q_mul(na.q_mul(qz, qy), qx)
    (sr, cr), (sp, cp), (sy, cy) = (na.sin_cos(x * F(0.5)) for x in e)
    want = (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)
    assert all(a == b for a, b in zip(q, want)), (q, want)
    # hierarchy propagation: translation column of global = parent.global * local
    h = g["graph_hierarchy"]
    for key_l, key_g in (("local_position", "global_position"), ("local_position_after", "global_position_after")):
        ts = []
        for p in h[key_l]:
            t = A.Transform.identity()
            t.local_position[:] = [float(x) for x in p]
            ts.append(t)
        sc = oracle2.AnimScene(A.Rig(parent=np.asarray(h["parent"], np.int32), transforms=ts))
        glo = sc.global_matrices()
        assert np.array_equal(glo[:, 12:15], np.asarray(h[key_g], np.float32))
    for case in g["blend_space_fetch_weights"]["cases"]:
        node = oracle2.anim.BlendSpaceNode(A.BlendSpace(0, [A.BlendSpacePoint(tuple(p), 0) for p in case["points"]],
                                                        [tuple(t) for t in case["triangles"]]))
        got = node.fetch_weights((F(case["sampling_point"][0]), F(case["sampling_point"][1])))
        if case["expected"] is None:
            assert got is None
        else:
            assert [(i, float(w)) for i, w in got] == [(i, float(w)) for i, w in case["expected"]]
