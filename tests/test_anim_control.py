"""Host control plane of the pose path, on a control-only context (no GPU, no kernels).

The library's per-instance control code (time advance, transitions, parameters, pose-node evaluation
order) emits sample times + a fold program per frame.  Here a small Python interpreter executes that
program on the ORACLE's animation poses (with the oracle's blend primitives) and the result must equal
the oracle's own Machine::evaluate_pose / update_animations bit for bit -- which pins the control
plane's logic and the program semantics without touching the HIP kernels."""
import os

import numpy as np
import pytest

import fyrox_amd
from fyrox_amd import _native
from fyrox_amd import anim as A
from fyrox_amd import synth

import anim_cases as cases


@pytest.fixture()
def cctx():
    c = fyrox_amd.Context(control_only=True)
    yield c
    c.close()


def _bits(rec):
    return int(np.float32(rec[3]).view(np.uint32))


def _blend(orc, self_rec, other, w):
    """NodePose::blend_with on 12-float records (pose.rs:41-47)."""
    sm, om = _bits(self_rec), _bits(other)
    if sm == 0:
        return other.copy()
    out = self_rec.copy()
    both = sm & om
    if both & 1:
        out[0:3] = orc.vec_lerp(self_rec[0:3], other[0:3], w)
    if both & 2:
        out[8:11] = orc.vec_lerp(self_rec[8:11], other[8:11], w)
    if both & 4:
        out[4:8] = orc.quat_nlerp(self_rec[4:8], other[4:8], w)
    return out


def _empty():
    r = np.zeros(12, np.float32)
    r[7] = 1.0
    return r


def _blend2(orc, a, f, oa, of, w):
    """The same on the two views of a pose whose lists hold several values per binding (csrc/anim_kernels.hip, run_fold_dup): `a` what the
    pose applies, `f` what a blend reads of it; emptiness is the list's (f's bits), both views blend with the other's READ view."""
    if _bits(f) == 0:
        return oa.copy(), of.copy()

    def values(rec):
        if _bits(rec) == 0:
            return rec
        out = _blend(orc, rec, of, w)
        out[3] = rec[3]
        return out
    return values(a), values(f)


def run_program(orc, ops, anim_poses, layer_excluded, trs, read_poses=None):
    """Execute one instance's fold program for every node; returns the new node TRS (n,12).  anim_poses: what every animation's pose
    applies; read_poses (default: the same records): what a blend reads of it."""
    trs = trs.copy()
    dec = A.decode_ops(ops)
    read_poses = anim_poses if read_poses is None else read_poses
    for node in range(trs.shape[0]):
        stack = [(_empty(), _empty())]
        for name, arg, w in dec:
            if name == "BLEND_ANIM":
                stack[-1] = _blend2(orc, *stack[-1], anim_poses[arg][node], read_poses[arg][node], np.float32(w))
            elif name == "PUSH":
                stack.append((_empty(), _empty()))
            elif name == "POP_BLEND":
                child = stack.pop()
                stack[-1] = _blend2(orc, *stack[-1], *child, np.float32(w))
            elif name == "RESET":
                stack[-1] = (_empty(), _empty())
            elif name == "MASK":
                if node in layer_excluded[arg]:
                    stack[-1] = (_empty(), _empty())
            elif name in ("APPLY", "APPLY_ANIM"):
                rec = stack[-1][0] if name == "APPLY" else anim_poses[arg][node]
                m = _bits(rec)
                if m & 1:
                    trs[node, 0:3] = rec[0:3]
                if m & 2:
                    trs[node, 8:11] = rec[8:11]
                if m & 4:
                    trs[node, 4:8] = rec[4:8]
            elif name == "END":
                break
        assert len(stack) == 1
    return trs


def run_rm_program(orc, ops, slots, anim_rm):
    """Execute one instance's root-motion program over persistent slots of 8-float records
    {dp xyz, has, dr ijkw} (pose.rs:73,98-100, play.rs:97, lib.rs:340-343) with the oracle's primitives."""
    one = np.asarray([1], np.uint32).view(np.float32)[0]

    def default(has):
        r = np.zeros(8, np.float32)
        r[7] = 1.0
        if has:
            r[3] = one
        return r

    for code, dst, src, wbits in ops:
        name = A.RM_OP_NAMES[int(code)]
        if name == "END":
            break
        if name == "SET_ANIM":
            slots[dst] = anim_rm[src].copy()
        elif name == "COPY":
            slots[dst] = slots[src].copy()
        else:
            w = np.float32(np.uint32(wbits).view(np.float32))
            d = slots[dst].copy() if _bits(slots[dst]) else default(True)
            o = slots[src] if _bits(slots[src]) else default(False)
            d[0:3] = orc.vec_lerp(d[0:3], o[0:3], w)
            d[4:8] = orc.quat_nlerp(d[4:8], o[4:8], w)
            slots[dst] = d
    return slots


def _drain(pop):
    out = []
    while True:
        e = pop()
        if e is None:
            return out
        out.append(e)


@pytest.mark.parametrize("seed", range(40))
def test_control_plane_matches_oracle_on_random_machines(orc, cctx, seed):
    _check_control_plane(orc, cctx, cases.random_machine(seed))


@pytest.mark.parametrize("seed", range(int(os.environ.get("FYX_FUZZ_SEEDS", 16))))
def test_control_plane_matches_oracle_on_random_machines_with_lists_of_values(orc, cctx, seed):
    _check_control_plane(orc, cctx, cases.random_machine(seed, listy=True))


@pytest.mark.parametrize("seed", range(24))
def test_control_plane_matches_oracle_on_random_machines_on_a_lattice(orc, cctx, seed):
    """Sampling points ON the corners and edges of the blend spaces' triangles, coinciding points, degenerate triangles (NaN weights): the
    planner's fetch_weights makes the oracle's decisions (tools/mutants_host.py: the open third side of barycentric_is_inside)."""
    _check_control_plane(orc, cctx, cases.random_machine(seed, listy=bool(seed % 2), lattice=True))


@pytest.mark.parametrize("make", cases.ALL + cases.ALL_RM, ids=lambda f: f.__name__)
def test_control_plane_matches_oracle(orc, cctx, make):
    _check_control_plane(orc, cctx, make())


def _check_control_plane(orc, cctx, sc):
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(cctx, sc, n_instances=2)
    mode = 0 if sc.machine is None else 1
    excluded = [set(l.mask) for l in sc.machine.layers] if sc.machine else []
    trs = o.node_trs()
    n_frames = min(sc.n_frames, 48)
    rm_slots = None
    alive = [True] * len(sc.animations)
    poses = [o.animation_pose(a) for a in range(len(sc.animations))]
    reads = [o.animation_pose(a, "read") for a in range(len(sc.animations))]
    anim_rm = [None] * len(sc.animations)
    for f in range(n_frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        for a in sc.removals.get(f, []):     # AnimationContainer::remove: the handle stops resolving from here on
            o.remove_animation(a)
            p.remove_animation(a)
            alive[a] = False
            with pytest.raises(fyrox_amd.FyxError):
                p.set_enabled(a, True)
        # what the oracle's animations hold BEFORE this frame (stale poses of animations that do not tick)
        before = [o.animation_state(a)["time_position"] if alive[a] else 0.0 for a in range(len(sc.animations))]
        plan = p.plan(mode, sc.dt)
        if mode:
            o.update_machine(sc.dt)
        else:
            o.update_animations(sc.dt)
        # both instances run the same script: identical plans
        o0, o1, o2 = plan["offsets"]
        assert np.array_equal(plan["ops"][o0:o1], plan["ops"][o1:o2])
        assert np.array_equal(plan["times"][0], plan["times"][1])
        # sample times: a ticked animation is sampled at its time before the tick
        for a in range(len(sc.animations)):
            if not alive[a]:
                assert not plan["ticked"][0, a] & 1, (f, a)
                continue
            if plan["ticked"][0, a] & 1:
                assert plan["times"][0, a] == np.float32(before[a]), (f, a)
            assert p.animation_state(a, 1) == o.animation_state(a), (f, a)
            # signals -> events (the queues are drained only every third frame, so capacities matter)
            assert p.event_count(a, 1) == o.event_count(a), (f, a)
            if f % 3 == 2:
                ref = _drain(lambda: o.pop_event(a))
                assert _drain(lambda: p.pop_event(a, 0)) == ref, (f, a)
                assert _drain(lambda: p.pop_event(a, 1)) == ref, (f, a)
        if sc.machine:
            for li in range(len(sc.machine.layers)):
                assert p.layer_state(li, 0) == o.layer_state(li), (f, li)
                for strategy in (A.EVENTS_ALL, A.EVENTS_MAX_WEIGHT, A.EVENTS_MIN_WEIGHT):   # a query: nothing is consumed
                    assert p.collect_active_animations_events(li, strategy, 1) == o.collect_active_animations_events(li, strategy), \
                        (f, li, strategy)
                ref = _drain(lambda: o.pop_layer_event(li))
                assert _drain(lambda: p.pop_layer_event(li, 0)) == ref, (f, li)
                assert _drain(lambda: p.pop_layer_event(li, 1)) == ref, (f, li)
        # a removed animation's pose is the one its PlayAnimation node copied last (play.rs:93-99)
        poses = [o.animation_pose(a) if alive[a] else poses[a] for a in range(len(sc.animations))]
        reads = [o.animation_pose(a, "read") if alive[a] else reads[a] for a in range(len(sc.animations))]
        trs = run_program(orc, plan["ops"][o0:o1], poses, excluded, trs, reads)
        assert np.array_equal(trs.view(np.uint32), o.node_trs().view(np.uint32)), f"{sc.name}: frame {f}"
        if sc.track_root_motion and sc.machine:
            rp = p.plan_root_motion()
            r0, r1, r2 = rp["offsets"]
            assert np.array_equal(rp["ops"][r0:r1], rp["ops"][r1:r2])
            if rm_slots is None:
                rm_slots = [np.zeros(8, np.float32) for _ in range(rp["n_slots"])]
            anim_rm = [o.animation_root_motion(a) if alive[a] else anim_rm[a] for a in range(len(sc.animations))]
            rm_slots = run_rm_program(orc, rp["ops"][r0:r1], rm_slots, anim_rm)
            def norm(rec):   # None reads back as RootMotion::default()
                if _bits(rec):
                    return rec
                r = np.zeros(8, np.float32)
                r[7] = 1.0
                return r
            base = 0
            for li, layer in enumerate(sc.machine.layers):
                base += len(layer.nodes)
                assert np.array_equal(norm(rm_slots[base]).view(np.uint32), o.machine_root_motion(li).view(np.uint32)), (f, li)
                base += 1
            assert np.array_equal(norm(rm_slots[-1]).view(np.uint32), o.machine_root_motion(-1).view(np.uint32)), f
            for a, spec in enumerate(sc.animations):   # the slices and flags the root-motion kernel receives
                assert tuple(rp["slices"][0, a]) == tuple(np.float32(x) for x in spec.time_slice)
    o.close()


def test_a_looping_animation_on_its_last_key_has_not_ended(orc, cctx):
    """Animation::has_ended is `!looped && |time - end| <= eps` (lib.rs:736-738): a LOOPING clip whose time position sits exactly on the end of its
    slice (wrapf keeps a value inside [start, end] as it is) has not ended, and an IsAnimationEnded transition on it does not fire; switched to
    non-looping at the same position it has, and the transition fires on the next update.  (tools/mutants_host.py: a has_ended without the
    loop test survived the scenario suite -- no scenario ever parks a looping clip on its last key.)"""
    n_bones, seed = 6, synth.SEED_BASE + 41
    rig = synth.make_rig(n_bones, seed)
    td, tgt = synth.make_clip(n_bones, seed, 0, n_keys=5, fps=4.0, euler_every=10 ** 9)
    layer = A.MachineLayer(nodes=[A.PlayAnimation(0), A.PlayAnimation(0)], states=[A.State(0), A.State(1)],
                           transitions=[A.Transition(0, 1, 0.1, ("ended", 0))])
    sc = cases.Scenario("ended", rig, [td], [cases.AnimSpec(0, tgt, time_slice=(0.0, 1.0), speed=0.0, looped=True)],
                        A.Machine(parameters=[], layers=[layer]), n_frames=4, has_euler=False)
    o, p = cases.build_oracle(orc, sc), cases.build_product(cctx, sc, 1)
    try:
        orc._alib().fo_animation_set_time_position(o.anims[0], 1.0)
        p.set_time_position(0, 1.0)
        for f in range(3):
            assert p.animation_state(0) == o.animation_state(0) and not p.animation_state(0)["has_ended"]
            o.update_machine(0.05)
            p.plan(1, 0.05)
            assert p.layer_state(0) == o.layer_state(0) == (0, -1), "the transition must not fire on a looping clip"
        orc._alib().fo_animation_set_loop(o.anims[0], 0)
        p.set_loop(0, False)
        assert p.animation_state(0) == o.animation_state(0) and p.animation_state(0)["has_ended"]
        o.update_machine(0.05)
        p.plan(1, 0.05)
        assert p.layer_state(0) == o.layer_state(0) and p.layer_state(0)[1] == 0, "now it fires"
    finally:
        o.close()
        p.free()


def test_transitions_scenario_visits_every_state(orc):
    sc = cases.transitions()
    o = cases.build_oracle(orc, sc)
    seen_states, seen_transitions = set(), set()
    for f in range(sc.n_frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
        o.update_machine(sc.dt)
        s, t = o.layer_state(0)
        seen_states.add(s)
        seen_transitions.add(t)
    assert {0, 1, 2} <= seen_states and {0, 2, 3} <= seen_transitions
    o.close()


def _logic_by_index(cond, names):
    if cond[0] == "parameter":
        return ("parameter", names.index(cond[1]))
    return (cond[0],) + tuple(_logic_by_index(c, names) for c in cond[1:])


def test_logic_node_doc_example_fires_the_transition(orc, cctx):
    """The one output the reference asserts of LogicNode::calculate_value: its doc-test (transition.rs:93-115), `!Run && Jump` over Run =
    Rule(false), Jump = Rule(true) -> true.  Here the expression is the condition of a transition between two states: it has fired after
    one update exactly when the expression is true -- on the oracle, the second oracle and the product's planner; the other three rows of
    the truth table beside it (they follow from the same text, the reference asserts only the first)."""
    import json
    import oracle2
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fyrox_unit_vectors.json")))["logic_node_doc_example"]
    names = list(g["parameters"])
    cond = _logic_by_index(g["logic"], names)
    n_bones, seed = 5, synth.SEED_BASE + 43
    rig = synth.make_rig(n_bones, seed)
    td, tgt = synth.make_clip(n_bones, seed, 0, n_keys=4, fps=4.0, euler_every=10 ** 9)
    for run, jump in [(g["parameters"]["Run"], g["parameters"]["Jump"]), (True, True), (False, False), (True, False)]:
        want = g["expect"] if (run, jump) == (g["parameters"]["Run"], g["parameters"]["Jump"]) else ((not run) and jump)
        layer = A.MachineLayer(nodes=[A.PlayAnimation(0), A.PlayAnimation(0)], states=[A.State(0), A.State(1)],
                               transitions=[A.Transition(0, 1, 0.5, cond)])
        m = A.Machine(parameters=[A.Parameter(A.PARAM_RULE, run), A.Parameter(A.PARAM_RULE, jump)], layers=[layer])
        sc = cases.Scenario("logic_doc", rig, [td], [cases.AnimSpec(0, tgt)], m, n_frames=1, has_euler=False)
        o, p = cases.build_oracle(orc, sc), cases.build_product(cctx, sc, 1)
        o2 = cases.build_oracle(oracle2, sc)
        try:
            o.update_machine(0.1)
            o2.update_machine(0.1)
            p.plan(1, 0.1)
            fired = (-1, 0) if want else (0, -1)          # (active state, active transition): a fired transition clears the state (layer.rs:640-648)
            assert o.layer_state(0) == fired, (run, jump)
            assert o2.layer_state(0) == fired, (run, jump)
            assert p.layer_state(0) == fired, (run, jump)
        finally:
            o.close()
            p.free()


def _planned_blend_space_weights(cctx, pts, tris, sp):
    """BlendSpace::fetch_weights as the PRODUCT's planner decides it: a one-state machine whose root is a blend space over one PlayAnimation
    per point; the frame's fold program is BLEND_ANIM (animation, weight) x 3, APPLY, END -- the (index, weight) triple itself.  [] when
    nothing is blended (fetch_weights returned None)."""
    n_bones, seed = 4, synth.SEED_BASE + 44
    rig = synth.make_rig(n_bones, seed)
    td, tgt = synth.make_clip(n_bones, seed, 0, n_keys=3, fps=4.0, euler_every=10 ** 9)
    n = len(pts)
    layer = A.MachineLayer(nodes=[A.PlayAnimation(i) for i in range(n)] + [A.BlendSpace(0, [A.BlendSpacePoint(tuple(q), i) for i, q in enumerate(pts)],
                                                                                       [tuple(t) for t in tris])], states=[A.State(n)])
    m = A.Machine(parameters=[A.Parameter(A.PARAM_SAMPLING_POINT, tuple(sp))], layers=[layer])
    sc = cases.Scenario("bs", rig, [td], [cases.AnimSpec(0, tgt) for _ in range(max(n, 1))], m, n_frames=1, has_euler=False)
    p = cases.build_product(cctx, sc, 1)
    try:
        ops = p.plan(1, 0.1)["ops"]
    finally:
        p.free()
    return [(int(o[0]) >> 8, float(np.array([o[1]], np.uint32).view(np.float32)[0])) for o in ops if (int(o[0]) & 0xff) == 1]     # OP_BLEND_ANIM


def test_planner_fetch_weights_on_the_reference_vectors(orc, cctx):
    """The product's own fetch_weights against what the reference's tests hold: the five cases of blendspace.rs:486-537 (exact (index, weight)
    triples), test_get_barycentric_coords_2d (fyrox-math/src/lib.rs:1198-1209) and test_barycentric_is_inside (:1224-1233).  The last two
    assert the two helpers fetch_weights is made of; here they are reached through it: a triangle and a sampling point that PRODUCE the
    asserted coordinates -- in the unit right triangle a = (0, 0), b = (1, 0), c = (0, 1) a point p has v = p.x, w = p.y, u = 1 - v - w -- are
    inside (three distinct indices, the coordinates as weights) or not (the nearest-edge branch: the third index repeats the second)."""
    import json
    import oracle2
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fyrox_unit_vectors.json")))
    for case in g["blend_space_fetch_weights"]["cases"]:
        want = [] if case["expected"] is None else [(int(i), float(w)) for i, w in case["expected"]]
        assert _planned_blend_space_weights(cctx, case["points"], case["triangles"], case["sampling_point"]) == want, case
    for case in g["barycentric_coords_2d"]["cases"]:
        pts = [case["a"], case["b"], case["c"]]
        want = [(0, case["expect"][0]), (1, case["expect"][1]), (2, case["expect"][2])]
        assert _planned_blend_space_weights(cctx, pts, [(0, 1, 2)], case["p"]) == want, case
        assert orc.blend_space_fetch_weights(np.asarray(pts, np.float32), np.asarray([[0, 1, 2]], np.uint32), tuple(case["p"])) == want
    tri_pts = [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)]
    f32 = np.float32
    for case in g["barycentric_is_inside"]["cases"]:
        u, v, _ = case["bary"]
        sp = (float(f32(v)), float(f32(1.0) - f32(u) - f32(v)))                 # v = p.x, w = p.y
        uu = f32(1.0) - f32(sp[0]) - f32(sp[1])                                 # what get_barycentric_coords_2d makes of it (inv_denom == 1)
        assert (uu >= 0) == (u >= 0) and (f32(sp[0]) >= 0) == (v >= 0) and ((uu + f32(sp[0]) < 1) == (u + v < 1)), "the point realises the asserted case"
        for got in (_planned_blend_space_weights(cctx, tri_pts, [(0, 1, 2)], sp),
                    orc.blend_space_fetch_weights(np.asarray(tri_pts, np.float32), np.asarray([[0, 1, 2]], np.uint32), sp)):
            inside = got is not None and len(got) == 3 and [i for i, _ in got] == [0, 1, 2]
            assert inside == case["inside"], (case, got)
        node = oracle2.anim.BlendSpaceNode(A.BlendSpace(0, [A.BlendSpacePoint(q, 0) for q in tri_pts], [(0, 1, 2)]))
        got2 = node.fetch_weights((f32(sp[0]), f32(sp[1])))
        assert (got2 is not None and [i for i, _ in got2] == [0, 1, 2]) == case["inside"], (case, got2)


def test_mesh_upload_checks_its_layout_before_it_needs_a_device(cctx):
    """fyx_mesh_upload validates the vertex layout first (an attribute must END inside the vertex: AnimatedVertex is 68 bytes,
    vertex.rs:139-155), so the refusals are the same on a context without a device; a layout that passes then needs one."""
    aos = np.zeros(680, np.uint8)
    for kw, status in [(dict(off_pos=0, off_weights=60, off_indices=64), "FYX_ERR_INVALID_ARG"),       # weights 60 + 16 > 68
                       (dict(off_pos=57, off_weights=32, off_indices=48), "FYX_ERR_INVALID_ARG"),      # position 57 + 12 > 68
                       (dict(off_pos=0, off_weights=48, off_indices=65), "FYX_ERR_INVALID_ARG"),       # indices 65 + 4 > 68
                       (dict(off_pos=0, off_normal=60, off_weights=32, off_indices=48), "FYX_ERR_INVALID_ARG"),
                       (dict(off_pos=0, off_tangent=56, off_weights=16, off_indices=32), "FYX_ERR_INVALID_ARG"),
                       (dict(off_pos=0, off_weights=-1, off_indices=64), "FYX_ERR_MISSING_ATTRIBUTE")]:
        with pytest.raises(fyrox_amd.FyxError) as e:
            cctx.mesh_upload(14, aos, 10, 68, **kw)
        assert e.value.status == status, kw
    with pytest.raises(fyrox_amd.FyxError) as e:      # the reference's own layout fits exactly (indices 64 + 4 == 68): only the device is missing
        cctx.mesh_upload(14, aos, 10, 68, off_pos=0, off_normal=20, off_tangent=32, off_weights=48, off_indices=64)
    assert e.value.code == _native.FYX_ERR_NO_DEVICE


def test_control_only_context_refuses_data_path(cctx):
    l = _native.lib()
    sc = cases.by_index()
    p = cases.build_product(cctx, sc)
    with pytest.raises(fyrox_amd.FyxError) as e:
        p.update_machine(1 / 60)
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    with pytest.raises(fyrox_amd.FyxError) as e:
        p.read(A.READ_LOCAL_TRS)
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    with pytest.raises(fyrox_amd.FyxError) as e:
        cctx.malloc(64)
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    with pytest.raises(fyrox_amd.FyxError) as e:
        cctx.sync()
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    assert l.fyx_lbs_skin(cctx._h, 1, None, 1, 1, None, None, None, None) != 0
    # the exchange step needs a device too, and an all-gather needs a communicator first
    with pytest.raises(fyrox_amd.FyxError) as e:
        cctx.comm_unique_id()
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    with pytest.raises(fyrox_amd.FyxError) as e:
        cctx.comm_init(bytes(128), 0, 1)
    assert e.value.code == _native.FYX_ERR_NO_DEVICE
    with pytest.raises(fyrox_amd.FyxError) as e:
        cctx.allgather_f32(0, 16, 0)
    assert e.value.code == _native.FYX_ERR_INVALID_ARG
    cctx.comm_shutdown()   # nothing to shut down: a no-op, not an error


def test_builder_validation(cctx):
    sc = cases.by_index()
    p = cases.build_product(cctx, sc)
    l, h = cctx._l, cctx._h
    # Property bindings take every value kind (an unknown kind is an argument error).  What the reference accepts is accepted
    # (round 6): a kind that fits no transform binding, several tracks on one binding or one property of a node -- the node's pose is
    # a list (pose.rs:107-121; tests/anim_cases.py duplicate_bindings)
    td = A.AnimationTracksData([A.Track(A.BIND_PROPERTY0 + 2, A.KIND_VEC3, [A.Curve([A.CurveKey(0, 1)])] * 3)])
    A.upload_tracks_data(cctx, 1, td)
    td = A.AnimationTracksData([A.Track(A.BIND_PROPERTY0 + 2, 9, [A.Curve([A.CurveKey(0, 1)])])])
    with pytest.raises(fyrox_amd.FyxError) as e:
        A.upload_tracks_data(cctx, 1, td)
    assert e.value.code == _native.FYX_ERR_INVALID_ARG
    real = [A.Curve([A.CurveKey(0, 1)])]
    td = A.AnimationTracksData([A.Track(A.BIND_PROPERTY0 + 2, A.KIND_REAL, real), A.Track(A.BIND_PROPERTY0 + 2, A.KIND_REAL, real)])
    A.upload_tracks_data(cctx, 2, td)
    p.add_animation(2, [1, 1])                        # the same property of the same node twice: one slot, two values in the node's list
    assert p.property_count() == 1 and p.property_slot(1, 2) == 0
    p.add_animation(2, [1, 3])
    assert p.property_count() == 2 and p.property_slot(1, 2) == 0 and p.property_slot(3, 2) == 1 and p.property_slot(2, 2) == -1
    td = A.AnimationTracksData([A.Track(A.BIND_POSITION, A.KIND_QUAT, [A.Curve()] * 4)])
    A.upload_tracks_data(cctx, 1, td)                 # a quaternion bound to Position: never applied, but a value of the list
    td = A.AnimationTracksData([A.Track(A.BIND_POSITION, 9, [A.Curve()] * 4)])
    with pytest.raises(fyrox_amd.FyxError) as e:
        A.upload_tracks_data(cctx, 1, td)
    assert e.value.code == _native.FYX_ERR_INVALID_ARG
    c3 = [A.Curve([A.CurveKey(0, 1)])] * 3
    td = A.AnimationTracksData([A.Track(A.BIND_POSITION, A.KIND_VEC3, c3), A.Track(A.BIND_POSITION, A.KIND_VEC3, c3)])
    A.upload_tracks_data(cctx, 7, td)
    p.add_animation(7, [1, 1])                        # two Position tracks on one node
    p.add_animation(7, [1, 2])
    # key locations as Curve keeps them: sorted (duplicates allowed) and finite; anything else has bypassed Curve
    def raw_upload(locs):
        curve = A.Curve()
        curve.keys = [A.CurveKey(float(x), 1.0) for x in locs]       # NOT sorted by the constructor
        td = A.AnimationTracksData([A.Track(A.BIND_PROPERTY0, A.KIND_REAL, [curve])])
        A.upload_tracks_data(cctx, 8, td)
    raw_upload([0.0, 0.5, 0.5, 1.0])
    for bad_locs in ([0.0, 1.0, 0.5], [0.0, float("nan"), 1.0], [float("inf")], [1.0, -1.0]):
        with pytest.raises(fyrox_amd.FyxError) as e:
            raw_upload(bad_locs)
        assert e.value.code == _native.FYX_ERR_INVALID_ARG
    # a pose-node cycle is rejected (the reference would recurse forever)
    li = A.c_uint32()
    assert l.fyx_machine_add_layer(h, p.id, 1.0, A.byref(li)) == 0
    src = np.asarray([0], np.int32)  # node 0 of the new layer = itself
    rc = l.fyx_layer_add_blend_animations(h, p.id, li.value, 1, src.ctypes.data_as(A.c_void_p), None, None, None)
    assert rc == _native.FYX_ERR_INVALID_ARG
    # unknown ids
    assert l.fyx_animator_free(h, 424242) == _native.FYX_ERR_UNKNOWN_ID
    assert l.fyx_rig_free(h, p.base_id) == _native.FYX_ERR_INVALID_ARG  # in use
    # non topological parent order
    bad = A.Rig(parent=np.asarray([1, -1], np.int32), transforms=[A.Transform.identity()] * 2)
    with pytest.raises(fyrox_amd.FyxError):
        A.create_rig(cctx, 999, bad)


def test_threaded_crowd_planner_equals_the_serial_one():
    """A crowd's frame is planned by several host threads over instance ranges (option anim.threads); the merged
    programs, sample times, root-motion programs, layer states and event queues must equal the single-threaded
    plan exactly, with every instance in its own state."""
    n = 777
    ctxs = [fyrox_amd.Context(control_only=True) for _ in range(2)]
    ctxs[0].set_option("anim.threads", 1)
    ctxs[1].set_option("anim.threads", 6)
    ctxs[1].set_option("anim.split", 100)
    sc = cases.ALL_RM[2]()            # transitions + root motion + signals
    ps = [cases.build_product(c, sc, n_instances=n) for c in ctxs]
    for p in ps:
        for i in range(n):
            for a in range(len(sc.animations)):
                p.set_time_position(a, (i * 0.0137 + a * 0.31) % 0.5, instance=i)
    for f in range(40):
        for idx, par in sc.script.get(f, []):
            for p in ps:
                for i in range(0, n, 3):                       # a third of the crowd follows the script ...
                    p.set_parameter(idx, par, instance=i)
        if f == 12:
            for p in ps:
                for i in range(1, n, 3):                       # ... another third switches later
                    p.set_parameter(0, A.Parameter(A.PARAM_RULE, True), instance=i)
        plans = [p.plan(1, sc.dt) for p in ps]
        for k in ("times", "ticked", "offsets", "ops"):
            assert np.array_equal(plans[0][k], plans[1][k]), (f, k)
        rms = [p.plan_root_motion() for p in ps]
        for k in ("offsets", "ops", "slices"):
            assert np.array_equal(rms[0][k], rms[1][k]), (f, k)
        for i in (0, 1, 2, 128, 129, 500, n - 1):
            assert ps[0].layer_state(0, i) == ps[1].layer_state(0, i)
            for a in range(len(sc.animations)):
                assert ps[0].event_count(a, i) == ps[1].event_count(a, i)
    states = {ps[0].layer_state(0, i) for i in range(n)}
    assert len(states) > 1, "the crowd should have diverged"
    for c in ctxs:
        c.close()


def test_scene_planner_on_many_threads_equals_one_by_one_plans():
    """fyx_scene_plan (the host half of fyx_scene_update): 70 animators of every scenario kind and different crowd sizes,
    dealt out to the planner threads in runs -- one of them a crowd large enough to be split over the pool itself --
    must produce for every animator exactly the frame fyx_animator_plan produces alone: programs, sample times,
    root-motion programs, machine states, event queues."""
    ctxs = [fyrox_amd.Context(control_only=True) for _ in range(2)]
    ctxs[0].set_option("anim.threads", 1)
    ctxs[1].set_option("anim.threads", 5)
    ctxs[1].set_option("anim.split", 64)
    makes = list(cases.ALL) + list(cases.ALL_RM) + [lambda s=s: cases.random_machine(s) for s in range(12)]
    members = []
    for k in range(70):
        sc = makes[k % len(makes)]()
        members.append((sc, 300 if k == 7 else 1 + (k * 7) % 11))
    sets = [[cases.build_product(c, sc, n) for sc, n in members] for c in ctxs]
    for ps in sets:
        for (sc, n), p in zip(members, ps):
            for i in range(n):
                for a in range(len(sc.animations)):
                    p.set_time_position(a, (i * 0.0137 + a * 0.31) % 0.5, instance=i)
    dt = 1.0 / 50.0
    for f in range(24):
        for ps in sets:
            for (sc, n), p in zip(members, ps):
                for idx, par in sc.script.get(f, []):
                    p.set_parameter(idx, par, instance=n // 2)
        one_by_one = [p.plan(0 if sc.machine is None else 1, dt) for (sc, _), p in zip(members, sets[0])]
        A.scene_plan(ctxs[1], sets[1], dt)
        for k, ((sc, n), p0, p1) in enumerate(zip(members, sets[0], sets[1])):
            got = p1.plan(-1, 0.0)
            for key in ("times", "ticked", "offsets", "ops"):
                assert np.array_equal(one_by_one[k][key], got[key]), (f, sc.name, key)
            if sc.track_root_motion and sc.machine is not None:
                r0, r1 = p0.plan_root_motion(), p1.plan_root_motion()
                for key in ("offsets", "ops", "slices"):
                    assert np.array_equal(r0[key], r1[key]), (f, sc.name, key)
            for i in {0, n // 2, n - 1}:
                if sc.machine is not None:
                    for li in range(len(sc.machine.layers)):
                        assert p0.layer_state(li, i) == p1.layer_state(li, i)
                for a in range(len(sc.animations)):
                    assert p0.event_count(a, i) == p1.event_count(a, i)
    with pytest.raises(fyrox_amd.FyxError):
        A.scene_plan(ctxs[1], [sets[1][0], sets[1][0]], dt)       # listed twice
    for c in ctxs:
        c.close()


def test_scene_block_tables_cover_every_animators_own_grid_exactly_once():
    """The tables behind fyx_scene_update's one-launch-per-stage: for every stage, the entries of job k are exactly the
    workgroups the per-animator launch of that stage would start for animator k -- each once, nothing else -- for
    animators on both sampler forms, with and without properties and root motion, rigs from 6 to 300 nodes."""
    ctx = fyrox_amd.Context(control_only=True)
    members = [(cases.c5_blend_tree(), 1), (cases.c5_blend_tree(), 40), (cases.c5_blend_tree(), 65), (cases.c5_blend_tree(n_bones=130), 70), (cases.morph_weights(), 3), (cases.property_kinds(), 70),
               (cases.ALL_RM[2](), 2), (cases.ALL_RM[0](), 33), (cases.player_only(), 5), (cases.removed_clips(), 1)]
    big = cases.c5_blend_tree(n_bones=300)
    members.append((big, 2))
    ps = [cases.build_product(ctx, sc, n) for sc, n in members]
    S_SAMPLE, S_CROWD, S_PSAMPLE, S_RM, S_RMFOLD, S_U64, S_U128, S_U192, S_U256, S_PUPD = range(10)
    tables = {st: A.scene_tables(ctx, ps, st) for st in range(10)}

    def of(st, k):
        t = tables[st]
        return [tuple(int(v) for v in r[1:]) for r in t[t[:, 0] == k]]

    for k, ((sc, n), p) in enumerate(zip(members, ps)):
        na, nn, nps = len(sc.animations), sc.rig.n_nodes, p.property_count()
        crowd = n >= 32
        want = {(x, y, a) for a in range(na) for y in range(nn * 3) for x in range((n + 63) // 64)} if crowd else set()
        got = of(S_CROWD, k)
        assert len(got) == len(set(got)) and set(got) == want, (sc.name, "crowd sampler")
        want = set() if crowd else {(x, i, a) for a in range(na) for i in range(n) for x in range((nn * 16 + 255) // 256)}
        got = of(S_SAMPLE, k)
        assert len(got) == len(set(got)) and set(got) == want, (sc.name, "sampler")
        want = {(x, i, a) for a in range(na) for i in range(n) for x in range((nps + 255) // 256)} if nps else set()
        got = of(S_PSAMPLE, k)
        assert len(got) == len(set(got)) and set(got) == want, (sc.name, "property sampler")
        got = of(S_RM, k)
        if sc.track_root_motion:
            g = min((na * n * 16 + 255) // 256, 256 * 16)
            assert sorted(got) == [(x, g, 0) for x in range(g)], (sc.name, "root motion")
            assert sorted(of(S_RMFOLD, k)) == [(x, 0, 0) for x in range((n + 63) // 64)]
        else:
            assert got == [] and of(S_RMFOLD, k) == []
        stage = S_U64 + (4 if n <= 64 else min((nn + 63) // 64, 4)) - 1      # update_block_waves: four waves for few instances
        for st in (S_U64, S_U128, S_U192, S_U256):
            assert sorted(of(st, k)) == ([(i, 0, 0) for i in range(n)] if st == stage else []), (sc.name, "update", st)
        want = {(x, i, 0) for i in range(n) for x in range((nps + 63) // 64)} if nps else set()
        got = of(S_PUPD, k)
        assert len(got) == len(set(got)) and set(got) == want, (sc.name, "property update")
    # job indices are positions in the list
    assert set(int(j) for st in range(10) for j in tables[st][:, 0]) <= set(range(len(members)))
    ctx.close()


def test_entry_points_reject_bad_arguments_without_crashing(cctx):
    """Nothing throws or aborts across the boundary: null pointers, unknown ids, out-of-range indices and a null context
    come back as error codes from the batch / scene / removal / random-action entry points too."""
    l, h = cctx._l, cctx._h
    E = _native
    sc = cases.by_index()
    p = cases.build_product(cctx, sc)
    ids = np.asarray([p.id], np.uint64)
    bad = np.asarray([0xdeadbeef], np.uint64)
    pi, pb = ids.ctypes.data_as(A.c_void_p), bad.ctypes.data_as(A.c_void_p)
    assert l.fyx_scene_plan(None, pi, 1, 0.1) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_scene_plan(h, None, 1, 0.1) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_scene_plan(h, pb, 1, 0.1) == E.FYX_ERR_UNKNOWN_ID
    assert l.fyx_scene_plan(h, None, 0, 0.1) == 0
    assert l.fyx_scene_update(h, pi, 1, 0.1) == E.FYX_ERR_NO_DEVICE
    assert l.fyx_scene_update(None, pi, 1, 0.1) == E.FYX_ERR_INVALID_ARG
    n = A.c_uint32()
    assert l.fyx_debug_scene_tables(h, pi, 1, 99, None, 0, A.byref(n)) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_debug_scene_tables(h, pb, 1, 0, None, 0, A.byref(n)) == E.FYX_ERR_UNKNOWN_ID
    assert l.fyx_debug_scene_tables(h, pi, 1, 0, None, 0, None) == 0
    assert l.fyx_animator_remove_animation(h, p.id, 99) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_animator_remove_animation(h, 0xdeadbeef, 0) == E.FYX_ERR_UNKNOWN_ID
    assert l.fyx_animator_remove_animation(None, p.id, 0) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_state_add_random_action(h, p.id, 0, 0, 1, None, 3) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_state_add_random_action(h, p.id, 0, 99, 1, None, 0) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_state_add_random_action(h, p.id, 9, 0, 1, None, 0) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_state_add_random_action(h, p.id, 0, 0, 1, None, 0) == 0          # an empty list is legal (and draws nothing)
    assert l.fyx_animator_set_random_seed(h, p.id, 99, 1) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_animator_set_random_seed(h, p.id, A.ALL_INSTANCES, 1) == 0
    assert l.fyx_state_add_action(h, p.id, 0, 0, 1, 4, 0) == E.FYX_ERR_INVALID_ARG   # EnableRandomAnimation has its own call
    assert l.fyx_lbs_skin_batch(h, None, 2) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_lbs_skin_batch(h, None, 0) == 0
    assert l.fyx_lbs_skin_batch(None, None, 0) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_lbs_skin_ex_batch(h, None, None, 1) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_lbs_skin_ex_batch(h, None, None, 0) == 0
    job = (_native.SkinJob * 1)(_native.SkinJob(12345, None, 4, 1, None, None, None))
    assert l.fyx_lbs_skin_batch(h, job, 1) == E.FYX_ERR_UNKNOWN_ID                # validated before any device is needed
    assert l.fyx_comm_init(h, None, 0, 1) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_comm_unique_id(h, None) == E.FYX_ERR_INVALID_ARG
    assert l.fyx_allgather_f32(None, None, 0, None) == E.FYX_ERR_INVALID_ARG
    # plan mode -1 on an animator that never planned: empty frame, no crash
    fresh = cases.build_product(cctx, cases.player_only())
    got = fresh.plan(-1, 0.0)
    assert got["ops"].shape[0] == 0


@pytest.mark.parametrize("make", [cases.c5_blend_tree, cases.transitions, cases.layered, cases.blend_space, cases.removed_clips,
                                  cases.random_attacks, cases.signals_and_clocks] + [lambda s=s: cases.random_machine(s) for s in range(6)],
                         ids=lambda f: getattr(f, "__name__", "random_machine"))
def test_memoised_fold_programs_equal_programs_planned_from_scratch(make):
    """The planner reuses an instance's fold program of the last frame when nothing it depends on changed (MachineState's
    memo).  Two control-only contexts run the same scenario -- scripts, removals, per-instance parameter changes and all --
    one of them with the memo defeated by a no-op setter before every frame (any non-read API call on the animator
    invalidates): sample times, tick flags, offsets and programs must be equal on every frame, and so must the machine
    states and event counts.  (Both are separately held to the oracle by the tests above.)"""
    sc = make()
    if sc.machine is None:
        pytest.skip("player-only scenario: no fold program to memoise")
    ctxs = [fyrox_amd.Context(control_only=True) for _ in range(2)]
    n = 5
    ps = [cases.build_product(c, sc, n) for c in ctxs]
    for p in ps:
        for i in range(n):
            for a in range(len(sc.animations)):
                p.set_time_position(a, (i * 0.173 + a * 0.29) % 0.9, instance=i)
    for f in range(max(sc.n_frames, 30)):
        for p in ps:
            for idx, par in sc.script.get(f, []):
                p.set_parameter(idx, par, instance=(f % n) if f % 3 else A.ALL_INSTANCES)
            for a in sc.removals.get(f, []):
                p.remove_animation(a)
        keep = next(a for a in range(len(sc.animations)) if all(a not in lst for lst in sc.removals.values()))
        ps[1].set_loop(keep, True if sc.animations[keep].looped is None else sc.animations[keep].looped)   # same value: only invalidates
        got = [p.plan(1, sc.dt) for p in ps]
        for key in ("times", "ticked", "offsets", "ops"):
            assert np.array_equal(got[0][key], got[1][key]), (sc.name, f, key)
        for i in range(n):
            for li in range(len(sc.machine.layers)):
                assert ps[0].layer_state(li, i) == ps[1].layer_state(li, i), (sc.name, f, i)
            gone = {a for fr, lst in sc.removals.items() if fr <= f for a in lst}
            for a in range(len(sc.animations)):
                if a not in gone:
                    assert ps[0].event_count(a, i) == ps[1].event_count(a, i)
    for c in ctxs:
        c.close()


def test_palette_output_pairs_are_checked_and_looked_up_on_the_host(cctx):
    """fyx_animator_set_palette_output_pair / fyx_animator_current_palette are host-side registrations: argument errors are refused
    before anything changes, the lookup answers without a GPU (no anim.overlap: the first buffer)."""
    sc = cases.player_only(n_bones=8)
    p = cases.build_product(cctx, sc, 1)
    base = p.base_id
    A.create_bone_list(cctx, base + 50, base, list(range(8)))
    with pytest.raises(fyrox_amd.FyxError, match="not a palette output"):
        p.current_palette(base + 50)
    with pytest.raises(fyrox_amd.FyxError, match="same buffer"):
        p.set_palette_output_pair(base + 50, 0x10000, 0x10000)
    with pytest.raises(fyrox_amd.FyxError, match="16-byte"):
        p.set_palette_output_pair(base + 50, 0x10000, 0x20008)
    with pytest.raises(fyrox_amd.FyxError, match="without a first"):
        p.set_palette_output_pair(base + 50, 0, 0x20000)
    with pytest.raises(fyrox_amd.FyxError, match="not a palette output"):
        p.current_palette(base + 50)
    p.set_palette_output_pair(base + 50, 0x10000, 0x20000)
    assert p.current_palette(base + 50) == 0x10000
    p.set_palette_output(base + 50, 0x30000)              # the plain call replaces the pair
    assert p.current_palette(base + 50) == 0x30000
    p.set_palette_output(base + 50, 0)
    with pytest.raises(fyrox_amd.FyxError, match="not a palette output"):
        p.current_palette(base + 50)
    p.free()
