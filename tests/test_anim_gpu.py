"""GPU parity of the pose path (through the C ABI) against the CPU oracle: keyframe sampling, pose
folding (Machine / layers / transitions / pose nodes), apply, local matrices, hierarchy, palettes, and
the end-to-end chain into the skinning kernel (BASELINE configs C2, C3-scaled, C5).

Bar.  Everything is BIT-EXACT except values that depend on sin/cos of UnitQuaternionEuler tracks: libm's
sinf/cosf are not reproducible bit-for-bit on a GPU (the kernel uses the device's <= 1 ulp sincosf), so
Euler-driven rotations -- and what is computed from them -- are held to 1e-5 relative, the tolerance
BASELINE.json's north_star states; scenarios without Euler tracks must match bit for bit.
"""
import numpy as np
import pytest

import fyrox_amd
from fyrox_amd import anim as A
from fyrox_amd import synth

import anim_cases as cases

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def rel_err(got, ref):
    scale = max(float(np.abs(ref).max()), 1e-3)
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) / scale


def same_bits(got, ref):
    """Every bit -- except that a NaN equals a NaN: which NaN an invalid operation produces (sign, payload) is specified neither by
    IEEE 754 nor by Rust; x86 makes 0xffc00000 of 0 / 0, the GPU 0x7fc00000 (found by the fuzzer: a quaternion of four zeros, normalised,
    in a root-motion delta -- NaN on both sides, one of the four with the other sign)."""
    return (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))


def check(got, ref, exact, what):
    assert got.shape == ref.shape, what
    if exact:
        assert same_bits(got, ref).all(), f"{what}: max rel err {rel_err(got, ref):.3e}"
    else:
        assert rel_err(got, ref) <= REL_TOL, f"{what}: max rel err {rel_err(got, ref):.3e}"


def check_mixed(got, ref, loose, what):
    """Bit-exact wherever `loose` (a boolean array of the same shape) is False, within REL_TOL where it is True: only values
    that can carry the device's sincosf instead of libm's (rotations sampled from UnitQuaternionEuler tracks and what is
    computed from them) get the tolerance -- a regression anywhere else in such a scenario cannot hide under it."""
    assert got.shape == ref.shape == loose.shape, what
    tight = ~loose
    bad = ~same_bits(got, ref) & tight
    assert not bad.any(), f"{what}: {int(bad.sum())} value(s) outside the Euler-tainted set differ, first at {tuple(np.argwhere(bad)[0])}"
    if loose.any():
        assert rel_err(got[loose], ref[loose]) <= REL_TOL, f"{what}: max rel err {rel_err(got[loose], ref[loose]):.3e} (Euler-tainted values)"


def check_pose(got, ref, exact, what, loose=None):
    """12-float pose records: present bits must match exactly; values per the bar."""
    bits = ref[:, 3].view(np.uint32)
    assert np.array_equal(got[:, 3].view(np.uint32), bits), f"{what}: present bits"
    g, r = got.copy(), ref.copy()
    for arr in (g, r):   # values of absent bindings are unspecified
        arr[:, 3] = 0
        arr[(bits & 1) == 0, 0:3] = 0
        arr[(bits & 4) == 0, 4:8] = 0
        arr[(bits & 2) == 0, 8:12] = 0
    if exact or loose is None:
        check(np.ascontiguousarray(g), np.ascontiguousarray(r), exact, what)
    else:
        check_mixed(np.ascontiguousarray(g), np.ascontiguousarray(r), loose, what)


class EulerTaint:
    """Which read-back values of a scenario may differ from the oracle in their last bits.  A UnitQuaternionEuler rotation
    track (container.rs:268-276) is the only place the GPU's arithmetic is not the CPU's (sincosf): the sampled
    rotation of that (animation, node) is tainted; blending carries it into the node's local rotation, the local matrix and
    -- down the hierarchy -- into the global matrices of the node and all its descendants.  Root motion reads the root
    node's sampled rotation and rewrites its position and rotation, so a root-motion node with a Euler rotation track is
    tainted as a whole."""

    def __init__(self, sc):
        n = sc.rig.n_nodes
        self.pose = []                      # per animation: (n_nodes, 12) bool
        trs_nodes = set()
        for spec in sc.animations:
            td = sc.tracks_data[spec.tracks]
            m = np.zeros((n, 12), bool)
            eul = {int(b) for t, b in zip(td.tracks, spec.target) if t.binding == A.BIND_ROTATION and t.kind == A.KIND_QUAT_EULER}
            for b in eul:
                if 0 <= b < n:
                    m[b, 4:8] = True
            # update_root_motion fetches the FIRST Rotation track of the tracks data (lib.rs:507-534), whatever node it drives
            first_rot = next((t for t in td.tracks if t.binding == A.BIND_ROTATION), None)
            rm_tainted = spec.root_motion is not None and (spec.root_motion[0] in eul or
                                                           (first_rot is not None and first_rot.kind == A.KIND_QUAT_EULER))
            if rm_tainted and 0 <= spec.root_motion[0] < n:
                m[spec.root_motion[0], :] = True
            self.pose.append(m)
            trs_nodes |= {b for b in eul if 0 <= b < n}
            if rm_tainted and 0 <= spec.root_motion[0] < n:
                trs_nodes.add(-1 - spec.root_motion[0])      # marker: position too
        self.trs = np.zeros((n, 12), bool)
        self.local = np.zeros((n, 16), bool)
        self.glob = np.zeros((n, 16), bool)
        dirty = np.zeros(n, bool)
        for b in trs_nodes:
            if b >= 0:
                self.trs[b, 4:8] = True
                dirty[b] = True
            else:
                self.trs[-1 - b, :] = True
                dirty[-1 - b] = True
        self.local[dirty, :] = True
        parent = np.asarray(sc.rig.parent)
        for node in range(n):
            q = node
            while q >= 0:
                if dirty[q]:
                    self.glob[node, :] = True
                    break
                q = int(parent[q])



def _all_instances_equal_the_first(arr, what):
    """The instances of a scenario run get the same inputs (diverging instances are test_instances_diverge's business), so EVERY
    instance -- not only the first and the last, which are compared with the oracle -- must hold the first one's bits: with 70
    instances on the lanes of the crowd sampler that is 68 more lanes checked per frame."""
    a = np.ascontiguousarray(arr)
    if a.shape[0] < 3:
        return
    bits = a.view(np.uint32).reshape(a.shape[0], -1)
    bad = np.nonzero((bits != bits[0]).any(axis=1))[0]
    assert bad.size == 0, f"{what}: instances {bad[:8].tolist()} differ from instance 0"


def check_frame(p, o, sc, n_instances, f):
    """Everything the product can read back after frame f against the oracle's single instance."""
    exact = not sc.has_euler
    taint = None if exact else getattr(sc, "_euler_taint", None)
    if not exact and taint is None:
        taint = sc._euler_taint = EulerTaint(sc)
    gone = {a for fr, lst in sc.removals.items() if fr <= f for a in lst}   # AnimationContainer::remove'd by now
    for a in range(len(sc.animations)):
        if a in gone:      # its record stays what it was (the pose PlayAnimation nodes keep using); nothing to compare with
            continue
        # the pose's two views (a node's pose is a LIST: include/fyrox_hip.h, FYX_READ_ANIMATION_BLEND_VIEW): what it applies, what a blend reads of it
        # (an animator without a machine blends nothing and keeps the apply view only)
        for what, view in ((A.READ_ANIMATION_POSE, "apply"), (A.READ_ANIMATION_BLEND_VIEW, "read" if sc.machine is not None else "apply")):
            got = p.read(what + a)
            ref = o.animation_pose(a, view)
            for i in (0, n_instances - 1):
                check_pose(got[i], ref, exact, f"{sc.name} frame {f} animation {a} pose, {view} view (instance {i})", None if exact else taint.pose[a])
            _all_instances_equal_the_first(got, f"{sc.name} frame {f} animation {a} pose, {view} view")
    trs, loc, glo = p.read(A.READ_LOCAL_TRS), p.read(A.READ_LOCAL_MATRIX), p.read(A.READ_GLOBAL_MATRIX)
    for name, arr in (("node TRS", trs), ("local matrices", loc), ("global matrices", glo)):
        _all_instances_equal_the_first(arr, f"{sc.name} frame {f} {name}")
    for i in (0, n_instances - 1):
        if exact:
            check(trs[i], o.node_trs(), True, f"{sc.name} frame {f} node TRS")
            check(loc[i], o.local_matrices(), True, f"{sc.name} frame {f} local matrices")
            check(glo[i], o.global_matrices(), True, f"{sc.name} frame {f} global matrices")
        else:
            check_mixed(trs[i], o.node_trs(), taint.trs, f"{sc.name} frame {f} node TRS")
            check_mixed(loc[i], o.local_matrices(), taint.local, f"{sc.name} frame {f} local matrices")
            check_mixed(glo[i], o.global_matrices(), taint.glob, f"{sc.name} frame {f} global matrices")
    if sc.machine is not None:
        for li in range(len(sc.machine.layers)):
            assert p.layer_state(li, n_instances - 1) == o.layer_state(li)
    if p.property_count():
        check_properties(p, o, sc, exact, f"{sc.name} frame {f}")
    if sc.track_root_motion:
        for a in range(len(sc.animations)):
            if a in gone:
                continue
            check_root_motion(p.animation_root_motion(a), o.animation_root_motion(a), exact,
                              f"{sc.name} frame {f} animation {a} root motion")
        if sc.machine is not None:
            for li in range(-1, len(sc.machine.layers)):
                check_root_motion(p.machine_root_motion(li), o.machine_root_motion(li), exact,
                                  f"{sc.name} frame {f} root motion of " + ("the machine" if li < 0 else f"layer {li}"))


def run_scenario(ctx, orc, sc, n_instances=1, frames=None, check_every=1):
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, n_instances)
    frames = sc.n_frames if frames is None else frames
    for f in range(frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        for a in sc.removals.get(f, []):
            o.remove_animation(a)
            p.remove_animation(a)
        if sc.machine is None:
            o.update_animations(sc.dt)
            p.update_animations(sc.dt)
        else:
            o.update_machine(sc.dt)
            p.update_machine(sc.dt)
        if f % check_every and f != frames - 1:
            continue
        check_frame(p, o, sc, n_instances, f)
    return o, p


def check_properties(p, o, sc, exact, what):
    """Property{..} slots: every animation's pose values (variant, lanes, presence) and the values applied so far."""
    slots = {}
    for node in range(sc.rig.n_nodes):
        for prop in range(8):
            s = p.property_slot(node, prop)
            if s >= 0:
                slots[(node, prop)] = s
    assert len(slots) == p.property_count() and sorted(slots.values()) == list(range(len(slots)))

    def compare(rec, ref, ctx_):
        kind, lanes = ref
        assert int(rec["kind"]) == kind, f"{ctx_}: TrackValue variant"
        assert np.all(rec["reserved"] == 0)
        check(rec["value"], lanes, exact, ctx_)

    for a in range(len(sc.animations)):
        got, ref = p.read_properties(a), o.animation_properties(a)
        for i in (0, got.shape[0] - 1):
            for key, s in slots.items():
                present = bool(got[i, s]["present"])
                assert present == (key in ref), f"{what}: animation {a} property {key} presence"
                if present:
                    compare(got[i, s], ref[key], f"{what}: animation {a} property {key}")
    got = p.read_properties(-1)
    for i in (0, got.shape[0] - 1):
        for key, s in slots.items():
            applied = bool(got[i, s]["present"])
            assert applied == (key in o.props), f"{what}: property {key} applied"
            if applied:
                compare(got[i, s], o.props[key], f"{what}: applied property {key}")


def check_root_motion(got, ref, exact, what):
    """(n_instances, 8) fyx_root_motion records against the oracle's single instance."""
    for i in (0, got.shape[0] - 1):
        assert got[i, 3].view(np.uint32) == ref[3].view(np.uint32), f"{what}: Option tag (instance {i})"
        g, r = got[i].copy(), ref.copy()
        g[3] = r[3] = 0
        check(g, r, exact, f"{what} (instance {i})")


def _drain(pop):
    out = []
    while (e := pop()) is not None:
        out.append(e)
    return out


@pytest.fixture(autouse=True)
def _anim_defaults(ctx):
    ctx.set_option("anim.sample_form", 0)
    yield
    ctx.set_option("anim.sample_form", 0)


@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
def test_scenario_matches_oracle(ctx, orc, make):
    sc = make()
    o, p = run_scenario(ctx, orc, sc, n_instances=3)
    o.close()
    p.free()


@pytest.mark.parametrize("make", cases.SUBNORMAL, ids=lambda f: f.__name__)
@pytest.mark.parametrize("form", [1, 2], ids=["curves_on_lanes", "instances_on_lanes"])
def test_scenario_with_subnormal_values_matches_oracle(ctx, orc, make, form):
    """with_subnormal_values: Position keys and rest positions scaled by 1e-39, a third of the Scale tracks by 1e-13 -- samples, lerps, blends,
    local matrices, the hierarchy's products and the palettes in and below the subnormal range.  The reference's f32 arithmetic (Rust, IEEE)
    keeps subnormal numbers; every kernel of the path must: poses, TRS, matrices and palettes bit for bit against the oracle, both sampler forms."""
    sc = make()
    ctx.set_option("anim.sample_form", form)
    o, p = run_scenario(ctx, orc, sc, n_instances=3, frames=min(sc.n_frames, 40))
    g = p.read(A.READ_GLOBAL_MATRIX)
    assert int(((np.abs(g) < 1.17e-38) & (g != 0)).sum()) > 50, "the frame holds subnormal numbers"
    o.close()
    p.free()


@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
@pytest.mark.parametrize("n_instances", [2, 70])
def test_scenario_with_instances_on_the_lanes(ctx, orc, make, n_instances):
    """The crowd form of the sampler (anim.sample_form = 2: 64 instances of one (animation, node) per wave, instance-minor
    span hints) must give the same bits as the per-instance form; 70 instances = one full and one ragged wave."""
    sc = make()
    ctx.set_option("anim.sample_form", 2)
    o, p = run_scenario(ctx, orc, sc, n_instances=n_instances, frames=min(sc.n_frames, 30), check_every=3)
    o.close()
    p.free()


@pytest.mark.parametrize("seed", range(24))
def test_random_machines_match_oracle(ctx, orc, seed):
    """Randomly generated clips / machines / scripts (tests/anim_cases.py::random_machine): poses, transforms, matrices,
    property values, root motion and event queues, bit for bit; odd seeds use the crowd form of the sampler."""
    sc = cases.random_machine(seed)
    ctx.set_option("anim.sample_form", 2 if seed % 2 else 1)
    o, p = run_scenario(ctx, orc, sc, n_instances=2 + seed % 3)
    for a in range(len(sc.animations)):
        assert _drain(lambda: p.pop_event(a, 0)) == _drain(lambda: o.pop_event(a))
    for li in range(len(sc.machine.layers)):
        assert _drain(lambda: p.pop_layer_event(li, 1)) == _drain(lambda: o.pop_layer_event(li))
    o.close()
    p.free()


# (103 ... 344: found by tools/fuzz_gpu.py -- the root node's list holds two Positions or Rotations and a loop has just wrapped: the remainders
#  are taken by the first value, the one that stays finds None, lib.rs:575-578)
@pytest.mark.parametrize("seed", list(range(16)) + [103, 118, 205, 241, 344])
def test_random_machines_with_lists_of_values_match_oracle(ctx, orc, seed):
    """random_machine(listy=True): further tracks on a (node, binding) or property that already has one, anywhere in the track order,
    kinds that fit no binding, property tracks of every vector kind, tracks with too few curves -- under random machines, in all three
    sampler forms (seed % 3)."""
    sc = cases.random_machine(seed, listy=True)
    ctx.set_option("anim.sample_form", seed % 3)
    o, p = run_scenario(ctx, orc, sc, n_instances=2 + seed % 3)
    for a in range(len(sc.animations)):
        assert _drain(lambda: p.pop_event(a, 0)) == _drain(lambda: o.pop_event(a))
    o.close()
    p.free()


@pytest.mark.parametrize("seed", range(8))
def test_random_machines_on_a_lattice_match_oracle(ctx, orc, seed):
    """random_machine(lattice=True): sampling points ON the corners and edges of the blend spaces' triangles, coinciding points, degenerate
    triangles whose weights are NaN -- the fold on the device carries whatever weights the planner hands it, bit for bit."""
    sc = cases.random_machine(seed, listy=bool(seed % 2), lattice=True)
    ctx.set_option("anim.sample_form", seed % 3)
    o, p = run_scenario(ctx, orc, sc, n_instances=1 + seed % 3)
    o.close()
    p.free()


@pytest.mark.parametrize("seed", range(12))
def test_random_curves_match_oracle(ctx, orc, seed):
    """random_curves: coinciding keys, mixed key kinds, empty and single-key curves, tracks whose curves sit on different time grids, slices
    that reach past the keys -- all three sampler forms."""
    sc = cases.random_curves(seed)
    ctx.set_option("anim.sample_form", seed % 3)
    o, p = run_scenario(ctx, orc, sc, n_instances=1 + seed % 3 if seed % 4 else 70)
    o.close()
    p.free()


@pytest.mark.parametrize("make", cases.ALL_RM, ids=lambda f: f.__name__)
def test_root_motion_and_signals_match_oracle(ctx, orc, make):
    """Animation::update_root_motion on the GPU (root pose rewritten before blending), AnimationPose::root_motion
    through pose nodes / layers / machine, signal events and layer events."""
    sc = make()
    o, p = run_scenario(ctx, orc, sc, n_instances=3)
    gone = {a for lst in sc.removals.values() for a in lst}
    for a in range(len(sc.animations)):
        if a in gone:
            continue
        ref = _drain(lambda: o.pop_event(a))
        assert _drain(lambda: p.pop_event(a, 0)) == ref and _drain(lambda: p.pop_event(a, 2)) == ref
    if sc.machine is not None:
        for li in range(len(sc.machine.layers)):
            ref = _drain(lambda: o.pop_layer_event(li))
            assert _drain(lambda: p.pop_layer_event(li, 1)) == ref
    o.close()
    p.free()


# ---- run-time edits of a machine (fyx_machine_clear + builder calls + state restore; tests/test_machine_edits.py has the
# control plane's side of it on the CPU) ----

@pytest.mark.parametrize("make", [cases.transitions, cases.by_index, cases.layered, cases.looping_root_motion,
                                  cases.with_root_motion_and_signals(cases.transitions)],
                         ids=["transitions", "by_index", "layered", "looping_root_motion", "transitions_rm"])
def test_machine_resent_between_frames_is_invisible_on_the_device(ctx, orc, make):
    """The same definition sent again every third frame (inside transitions and cross-fades too): poses, matrices,
    layer states and root motion keep matching the oracle, whose machine is never touched."""
    sc = make()
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, 3)
    for f in range(min(sc.n_frames, 48)):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        if f % 3 == 2:
            for li in range(len(sc.machine.layers)):
                ref = _drain(lambda: o.pop_layer_event(li))
                assert _drain(lambda: p.pop_layer_event(li, 2)) == ref and _drain(lambda: p.pop_layer_event(li, 0)) == ref
                _drain(lambda: p.pop_layer_event(li, 1))
            p.rebuild_machine(sc.machine, sc.machine)
        o.update_machine(sc.dt)
        p.update_machine(sc.dt)
        check_frame(p, o, sc, 3, f)
    o.close()
    p.free()


def test_machine_edited_between_frames_matches_the_edit_in_place(ctx):
    """A real edit (transition times, then a state added with its transitions, then a blend node's weights) made in place
    on oracle2's machine objects and through fyx_machine_clear + builder calls + the state restore on the device."""
    import oracle2
    import test_machine_edits as E
    for make, edits in ((cases.transitions, {8: E.edit_retime, 12: E.edit_grow, 34: E.edit_recondition}),
                        (cases.layered, {9: E.edit_reweight, 21: E.edit_layer_weight_and_mask, 26: E.edit_swap_clip}),
                        (cases.by_index, {6: E.edit_by_index_times, 31: E.edit_grow})):
        sc = make()
        o = cases.build_oracle(oracle2, sc)
        p = cases.build_product(ctx, sc, 2)
        desc = sc.machine
        for f in range(min(sc.n_frames, 48)):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            if f in edits:
                new, in_place, maps = edits[f](desc)
                for li in range(len(desc.layers)):
                    _drain(lambda: o.pop_layer_event(li))
                in_place(o.machine)
                p.rebuild_machine(desc, new, **maps)
                desc = new
            o.update_machine(sc.dt)
            p.update_machine(sc.dt)
            sc.machine = desc                     # check_frame walks the current definition's layers
            check_frame(p, o, sc, 2, f)
        o.close()
        p.free()


def test_quaternion_only_blend_tree_is_bit_exact(ctx, orc):
    sc = cases.c5_blend_tree(euler_every=10 ** 9)
    assert not sc.has_euler
    o, p = run_scenario(ctx, orc, sc, n_instances=2, frames=30)
    o.close()
    p.free()


@pytest.mark.parametrize("n_clips", [1, 2, 3, 4, 5, 6])
def test_blend_node_operand_counts_around_the_straight_program_limit(ctx, orc, n_clips):
    """One layer, one state, one BlendAnimations node over n clips: up to four operands the update kernel runs the program in
    its straight form (all operand records requested together, anim_kernels.hip pose_update_body), above that in the fold
    interpreter -- the same blend() calls in the same order either way, so both must give the oracle's bits.  The clips
    are partial in different ways (a node missing from the first operand takes the copy rule, a value missing from a
    later one is dropped), and the quaternion tracks keep the comparison bit-exact."""
    nb, seed = 20, synth.SEED_BASE + 21
    rig = synth.make_rig(nb, seed)
    tds, anims = [], []
    for c in range(n_clips):
        td, tgt = synth.make_clip(nb, seed, clip=c, euler_every=10 ** 9)
        if c % 2 == 0:
            td, tgt = cases._partial(td, tgt, lambda b, t, c=c: (b + c) % 5 != 0 and not (b % 3 == 1 and t.binding == A.BIND_SCALE))
        tds.append(td)
        anims.append(cases.AnimSpec(c, tgt, speed=[1.0, 0.8, 1.3, -0.7, 2.1, 0.4][c]))
    nodes = [A.PlayAnimation(a) for a in range(n_clips)]
    nodes.append(A.BlendAnimations([A.BlendPose(a, [1.0, 0.5, 0.25, 0.75, 0.6, 0.1][a]) for a in range(n_clips)]))
    machine = A.Machine(parameters=[], layers=[A.MachineLayer(nodes=nodes, states=[A.State(root=n_clips)])])
    sc = cases.Scenario("blend_%d" % n_clips, rig, tds, anims, machine, has_euler=False, n_frames=24)
    o, p = run_scenario(ctx, orc, sc, n_instances=3)
    o.close()
    p.free()


@pytest.mark.parametrize("kind", [A.KEY_CONSTANT, A.KEY_LINEAR, A.KEY_CUBIC])
def test_key_kinds_sample_bit_exact(ctx, orc, kind):
    sc = cases.player_only(euler_every=10 ** 9, key_kind=kind)
    o, p = run_scenario(ctx, orc, sc, frames=40)
    o.close()
    p.free()


@pytest.mark.parametrize("form", [1, 2], ids=["curves_on_lanes", "instances_on_lanes"])
def test_sampling_hints_with_duplicate_key_locations(ctx, orc, form):
    """Duplicate key locations make Curve::value_at depend on the span hint carried between frames
    (curve.rs:254-314): the device keeps one hint per (instance, animation, track, curve).  Both forms of the sampler."""
    ctx.set_option("anim.sample_form", form)
    rig = synth.make_rig(4, 77)
    keys = [A.CurveKey(0.0, 0.0), A.CurveKey(0.25, 1.0), A.CurveKey(0.25, 5.0), A.CurveKey(0.5, 2.0),
            A.CurveKey(0.5, -3.0, A.KEY_CONSTANT), A.CurveKey(0.75, 4.0, A.KEY_CUBIC, 0.5, -0.25), A.CurveKey(1.0, 0.5)]
    cv = lambda s: A.Curve([A.CurveKey(k.location, k.value * s, k.kind, k.left_tangent, k.right_tangent) for k in keys])
    td = A.AnimationTracksData([A.Track(A.BIND_POSITION, A.KIND_VEC3, [cv(1.0), cv(-2.0), cv(0.5)])])
    sc = cases.Scenario("dup_keys", rig, [td], [cases.AnimSpec(0, np.asarray([2], np.int32), speed=1.0)], None,
                        n_frames=130, dt=1.0 / 64.0, has_euler=False)   # dt hits the duplicate locations exactly
    o, p = run_scenario(ctx, orc, sc)
    # and backwards, crossing the same keys with the hints left by the forward pass
    p.set_speed(0, -1.0)
    orc._alib().fo_animation_set_speed(o.anims[0], -1.0)
    for f in range(70):
        o.update_animations(sc.dt)
        p.update_animations(sc.dt)
        check_pose(p.read(A.READ_ANIMATION_POSE)[0], o.animation_pose(0), True, f"reverse frame {f}")
    o.close()
    p.free()


@pytest.mark.parametrize("form", [1, 2], ids=["curves_on_lanes", "instances_on_lanes"])
def test_span_records_take_every_exit(ctx, orc, form):
    """The sampler's span records (TrackHot, round 3): tracks whose curves share their key times are sampled from one record per
    span -- inside the hinted span, the next or the previous one -- and fall back to the per-curve records for everything else
    (the form with the curves on the lanes), or decide all of Curve::value_at on the span locations (the crowd form).
    One clip holds a Vector3 track and a quaternion track with common key times (128- and 256-byte records), a Vector3 track whose
    curves have DIFFERENT key times (no records) and one with a single key; key locations are multiples of 1/32 and dt = 1/64, so
    every second frame lands exactly on a key, the ends clamp, and the speed changes below make the hint miss by one span, by
    several, and across the loop point.  Every frame's pose is the oracle's, bit for bit."""
    rng = np.random.default_rng(4242)
    rig = synth.make_rig(6, 91)
    locs = [k / 32.0 for k in range(0, 33, 2)]                       # 17 keys over one second
    kinds = [A.KEY_LINEAR, A.KEY_CUBIC, A.KEY_CONSTANT]

    def curve(times, scale=1.0):
        return A.Curve([A.CurveKey(t, float(rng.normal()) * scale, kinds[i % 3], float(rng.normal()), float(rng.normal()))
                        for i, t in enumerate(times)])

    tracks, target = [], []
    tracks.append(A.Track(A.BIND_POSITION, A.KIND_VEC3, [curve(locs), curve(locs), curve(locs)])); target.append(1)
    tracks.append(A.Track(A.BIND_ROTATION, A.KIND_QUAT, [curve(locs), curve(locs), curve(locs), curve(locs)])); target.append(1)
    tracks.append(A.Track(A.BIND_SCALE, A.KIND_VEC3, [curve(locs), curve(locs[::2]), curve(locs)])); target.append(1)      # ragged: general path
    tracks.append(A.Track(A.BIND_POSITION, A.KIND_VEC3, [curve(locs[:1]), curve(locs[:1]), curve(locs[:1])])); target.append(2)   # one key
    tracks.append(A.Track(A.BIND_ROTATION, A.KIND_QUAT, [curve(locs[3:9])] * 4)); target.append(3)               # keys only in the middle: clamps
    td = A.AnimationTracksData(tracks)
    ctx.set_option("anim.sample_form", form)
    sc = cases.Scenario("span_records", rig, [td], [cases.AnimSpec(0, np.asarray(target, np.int32), speed=1.0)], None,
                        n_frames=70, dt=1.0 / 64.0, has_euler=False)
    o, p = run_scenario(ctx, orc, sc)
    f = 0
    for speed, frames in ((-1.0, 40), (3.0, 50), (-2.5, 60), (0.5, 30), (-7.0, 20), (1.0, 10)):
        p.set_speed(0, speed)
        orc._alib().fo_animation_set_speed(o.anims[0], speed)
        for _ in range(frames):
            o.update_animations(sc.dt)
            p.update_animations(sc.dt)
            check_pose(p.read(A.READ_ANIMATION_POSE)[0], o.animation_pose(0), True, f"speed {speed} frame {f}")
            f += 1
    for t in (0.0, 1.0, 0.4375, 0.03125, 0.96875, 0.5):      # jumps: the hints are wherever the last run left them
        p.set_time_position(0, t)
        orc._alib().fo_animation_set_time_position(o.anims[0], t)
        for _ in range(3):
            o.update_animations(sc.dt)
            p.update_animations(sc.dt)
            check_pose(p.read(A.READ_ANIMATION_POSE)[0], o.animation_pose(0), True, f"jump to {t}")
    o.close()
    p.free()


@pytest.mark.parametrize("n_instances", [3, 70], ids=["curves_on_lanes", "instances_on_lanes"])
def test_track_bindings_switched_off_and_on_between_frames(ctx, orc, n_instances):
    """TrackBinding::enabled toggled while the animation plays (track.rs: a disabled binding's track is skipped by
    Animation::update_pose, lib.rs:903-906): the node loses that value from the animation's pose, a second track bound to the
    same node and binding takes over if there is one.  The samplers find their tracks through per-animator descriptors built on
    the device (CrowdDesc) -- they have to follow every such change.  Both sampler forms (3 / 70 instances)."""
    sc = cases.c5_blend_tree(n_bones=12, seed=synth.SEED_BASE + 23, euler_every=10 ** 9)
    o, p = run_scenario(ctx, orc, sc, n_instances=n_instances, frames=6)
    tracks = sc.tracks_data[1].tracks
    target = sc.animations[1].target

    def toggle(track, on):
        p.set_track_enabled(1, track, on)
        orc._alib().fo_animation_bind(o.anims[1], track, int(target[track]), int(on))

    steps = {0: [(0, False), (4, False)], 5: [(1, False), (0, True)], 9: [(4, True), (2, False), (7, False)], 14: [(1, True), (2, True), (7, True)]}
    for f in range(20):
        for track, on in steps.get(f, []):
            if track < len(tracks):
                toggle(track, on)
        o.update_machine(sc.dt)
        p.update_machine(sc.dt)
        check_frame(p, o, sc, n_instances, 6 + f)
    o.close()
    p.free()


def test_instances_diverge(ctx, orc):
    """Per-instance state: different speeds / parameters per instance vs one oracle scene each."""
    sc = cases.transitions()
    n = 4
    os_ = [cases.build_oracle(orc, sc) for _ in range(n)]
    p = cases.build_product(ctx, sc, n)
    for i in range(n):
        p.set_speed(0, 0.5 + 0.5 * i, instance=i)
        orc._alib().fo_animation_set_speed(os_[i].anims[0], 0.5 + 0.5 * i)
    for f in range(60):
        for i in range(n):
            if f == 3 + 5 * i:     # each instance starts walking at a different frame
                par = A.Parameter(A.PARAM_RULE, True)
                p.set_parameter(0, par, instance=i)
                os_[i].set_parameter(0, par)
            os_[i].update_machine(sc.dt)
        p.update_machine(sc.dt)
        trs = p.read(A.READ_LOCAL_TRS)
        for i in range(n):
            check(trs[i], os_[i].node_trs(), True, f"frame {f} instance {i}")
            assert p.layer_state(0, i) == os_[i].layer_state(0)
    for o in os_:
        o.close()
    p.free()


@pytest.mark.parametrize("pack", [0, 2, 4])
def test_crowd_update_packs_instances_per_workgroup(ctx, orc, pack):
    """anim.update_pack: a crowd's lean update launch with 1 / 2 / 4 instances of a small rig per workgroup.  67 instances (not a
    multiple of the pack: the last workgroup repeats the last instance) at different speeds; eight of them against an oracle
    scene each, all of them against the unpacked launch bit for bit."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)      # quaternion tracks only: bit-exact against the oracle
    n, probe = 67, (0, 1, 2, 3, 31, 64, 65, 66)
    os_ = {i: cases.build_oracle(orc, sc) for i in probe}
    ps = []
    for pk in (0, pack):
        ctx.set_option("anim.update_pack", pk)
        p = cases.build_product(ctx, sc, n)
        for i in range(n):
            p.set_speed(0, 0.25 + 0.03 * i, instance=i)
        got = []
        for f in range(12):
            p.update_machine(sc.dt)
            got.append((p.read(A.READ_LOCAL_TRS).copy(), p.read(A.READ_GLOBAL_MATRIX).copy()))
        ps.append(got)
        p.free()
    ctx.set_option("anim.update_pack", 4)
    for i in probe:
        orc._alib().fo_animation_set_speed(os_[i].anims[0], 0.25 + 0.03 * i)
    for f in range(12):
        for i in probe:
            os_[i].update_machine(sc.dt)
            check(ps[1][f][0][i], os_[i].node_trs(), True, f"pack {pack} frame {f} instance {i} node TRS")
            check(ps[1][f][1][i], os_[i].global_matrices(), True, f"pack {pack} frame {f} instance {i} global matrices")
        for k in range(2):
            assert np.array_equal(ps[0][f][k].view(np.uint32), ps[1][f][k].view(np.uint32)), f"pack {pack} frame {f}: differs from one instance per workgroup"
    assert not np.array_equal(ps[1][11][0][0], ps[1][11][0][66])        # the instances do differ
    for o in os_.values():
        o.close()


def test_set_local_trs_places_instances(ctx, orc):
    sc = cases.by_index()
    n = 5
    p = cases.build_product(ctx, sc, n)
    os_ = [cases.build_oracle(orc, sc) for _ in range(n)]
    trs = np.zeros((n, 10), np.float32)
    trs[:, 0] = np.arange(n) * 3.0
    trs[:, 2] = -np.arange(n)
    trs[:, 3:7] = (0.0, 0.38268343, 0.0, 0.92387953)
    trs[:, 7:10] = 1.0
    node = 1   # by_index animates every node; pick one and re-place it after the update
    p.update_machine(sc.dt)
    p.set_local_trs(node, trs[1:], first_instance=1)
    p.update_transforms()
    glo = p.read(A.READ_GLOBAL_MATRIX)
    for i in range(n):
        os_[i].update_machine(sc.dt)
        if i >= 1:
            os_[i].set_local_trs(node, trs[i])
        check(glo[i], os_[i].global_matrices(), True, f"instance {i}")
    for o in os_:
        o.close()
    p.free()


# ---- end to end: pose -> palette -> skinning ---------------------------------------------------

def _skin_chain(ctx, orc, sc, mesh, n_instances, frames, bone_nodes, exact):
    o, p = run_scenario(ctx, orc, sc, n_instances=n_instances, frames=frames, check_every=10 ** 9)
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, bone_nodes)
    nb = len(bone_nodes)
    d_pal = ctx.malloc(n_instances * nb * 64)
    p.palette(base + 50, d_pal.ptr)
    pal = d_pal.download(np.float32, n_instances * nb * 16).reshape(n_instances, nb, 16)
    ref_pal = o.palette(bone_nodes)
    for i in (0, n_instances - 1):
        check(pal[i], ref_pal, exact, "palette")
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    nv = mesh.n_verts * n_instances
    d_pos, d_nrm, d_tan = ctx.malloc(nv * 12), ctx.malloc(nv * 12), ctx.malloc(nv * 16)
    ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_instances, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    ctx.join()
    got = {"pos": d_pos.download(np.float32, nv * 3).reshape(n_instances, -1, 3),
           "normal": d_nrm.download(np.float32, nv * 3).reshape(n_instances, -1, 3),
           "tangent": d_tan.download(np.float32, nv * 4).reshape(n_instances, -1, 4)}
    ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent, threads=0)
    for k in ("pos", "normal", "tangent"):
        for i in (0, n_instances - 1):
            check(got[k][i], ref[k], exact, f"skinned {k} (instance {i})")
    if exact:
        # the skinning kernel itself is bit-exact given the GPU-built palette
        ref2 = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[0], mesh.normal, mesh.tangent, threads=0)
        assert np.array_equal(got["pos"][0], ref2["pos"])
    for b in (d_pal, d_pos, d_nrm, d_tan):
        b.free()
    ctx.mesh_free(base + 60)
    o.close()
    p.free()


def test_c2_character_50k_verts_64_bones_one_clip(ctx, orc):
    """BASELINE C2: one character, 50k verts / 64 bones / 1 clip: player -> palette -> LBS vs CPU (1e-5)."""
    n_bones = 64
    rig = synth.make_rig(n_bones, synth.SEED_BASE + 2)
    td, tgt = synth.make_clip(n_bones, synth.SEED_BASE + 2, 0)
    sc = cases.Scenario("c2", rig, [td], [cases.AnimSpec(0, tgt)], None, n_frames=20)
    mesh = synth.make_mesh(50_000, n_bones, synth.SEED_BASE + 2)
    _skin_chain(ctx, orc, sc, mesh, 1, 20, list(range(n_bones)), exact=False)


def test_c5_machine_four_clip_blend_tree_100k_verts(ctx, orc):
    """BASELINE C5: Machine 4-clip blend tree -> palette -> 100k-vert LBS, end-to-end f32 tolerance vs CPU."""
    sc = cases.c5_blend_tree(n_bones=64)
    mesh = synth.make_mesh(100_000, 64, synth.SEED_BASE + 5)
    _skin_chain(ctx, orc, sc, mesh, 1, 60, list(range(64)), exact=False)


def test_c5_quaternion_tracks_end_to_end_bit_exact(ctx, orc):
    sc = cases.c5_blend_tree(n_bones=64, euler_every=10 ** 9)
    mesh = synth.make_mesh(20_000, 64, synth.SEED_BASE + 5)
    _skin_chain(ctx, orc, sc, mesh, 1, 25, list(range(64)), exact=True)


def test_c3_crowd_instances_from_pose_to_vertices(ctx, orc):
    """Scaled C3: 48 instances x 10k verts / 64 bones, per-instance palettes built on the GPU and consumed
    by the instanced skinning launch without leaving HBM.  Bones are a subset with an invalid handle."""
    sc = cases.layered(n_bones=64)
    mesh = synth.make_mesh(10_000, 32, synth.SEED_BASE + 3)
    bone_nodes = [2 * b for b in range(32)]
    bone_nodes[5] = -1   # Handle::NONE -> identity matrix (scene/mesh/mod.rs:789-791)
    _skin_chain(ctx, orc, sc, mesh, 48, 12, bone_nodes, exact=True)


def test_pipelined_frames_are_bit_identical_to_one_stream_frames(ctx, orc):
    """Option anim.overlap: frame n+1's pose kernels run under frame n's skinning (worker streams, two palette buffers).
    Same kernels, same inputs: the skinned vertices of EVERY frame equal those of the one-stream frame loop bit for bit
    (each frame's output lands in its own buffer, compared after the loop), and the last frame equals the oracle."""
    n_inst, n_frames, nb = 40, 14, 64
    mesh = synth.make_mesh(10_000, nb, synth.SEED_BASE + 3)
    results = []
    for overlap in (0, 1, 2):
        sc = cases.c5_blend_tree(n_bones=nb, seed=synth.SEED_BASE + 3, euler_every=10 ** 9)
        p = cases.build_product(ctx, sc, n_inst)
        base = p.base_id
        for i in range(n_inst):
            for a in range(len(sc.animations)):
                p.set_time_position(a, (i * 0.37 + a * 0.11) % 1.0, instance=i)
        A.create_bone_list(ctx, base + 50, base, list(range(nb)))
        pals = [ctx.malloc(n_inst * nb * 64) for _ in range(2)]
        ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        nv = mesh.n_verts * n_inst
        outs = [(ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)) for _ in range(n_frames)]
        ctx.set_option("debug.overlap", overlap)
        ctx.set_option("lbs.streams", 2 if overlap else 1)
        try:
            for f in range(n_frames):
                dp = pals[f & 1]
                p.set_palette_output(base + 50, dp.ptr)
                p.update_machine(sc.dt)
                ctx.lbs_skin_device(base + 60, dp.ptr, nb, n_inst, outs[f][0].ptr, outs[f][1].ptr, outs[f][2].ptr)
            ctx.join()
            ctx.sync()
        finally:
            ctx.set_option("anim.overlap", 0)
            ctx.set_option("lbs.streams", 2)
        results.append([[b.download(np.uint32, nv * w) for b, w in zip(o, (3, 3, 4))] for o in outs])
        last_pal = pals[(n_frames - 1) & 1].download(np.float32, n_inst * nb * 16).reshape(n_inst, nb, 16)
        if overlap:   # the last frame against the oracle (instance 0 and the last one, each at its own phase)
            for i in (0, n_inst - 1):
                o = cases.build_oracle(orc, sc)
                for a in range(len(sc.animations)):
                    orc._alib().fo_animation_set_time_position(o.anims[a], (i * 0.37 + a * 0.11) % 1.0)
                for f in range(n_frames):
                    o.update_machine(sc.dt)
                ref_pal = o.palette(list(range(nb)))
                assert np.array_equal(last_pal[i].view(np.uint32), ref_pal.view(np.uint32)), f"palette of instance {i}"
                ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent, threads=0)
                got = results[-1][-1][0].view(np.float32).reshape(n_inst, -1, 3)[i]
                assert np.array_equal(got, ref["pos"]), f"skinned positions of instance {i}"
                o.close()
        for b in pals + [x for o in outs for x in o]:
            b.free()
        ctx.mesh_free(base + 60)
        p.free()
    for f in range(n_frames):
        for k in range(3):
            for mode in (1, 2):
                assert np.array_equal(results[0][f][k], results[mode][f][k]), f"frame {f}, stream {k}: pipelined (anim.overlap = {mode}) != one-stream"


def test_animated_morph_weights_drive_blend_shapes_into_a_vertex_buffer(ctx, orc):
    """glTF-style chain, all on the device: Real Property tracks -> blended weights -> fyx_animator_blend_shape_weights
    (x / 100, defaults for shapes nothing animated yet) -> blend shapes + skinning with GPU-built palettes -> a complete
    AnimatedVertex vertex buffer.  The oracle does the same steps with the reference's formulas."""
    sc = cases.morph_weights(n_bones=16)
    n_inst, n_shapes, mesh_node = 3, 7, 4
    o, p = run_scenario(ctx, orc, sc, n_instances=n_inst, frames=30, check_every=10 ** 9)
    base = p.base_id
    bone_nodes = list(range(16))
    A.create_bone_list(ctx, base + 50, base, bone_nodes)
    d_pal = ctx.malloc(n_inst * 16 * 64)
    p.palette(base + 50, d_pal.ptr)
    # shape k of the mesh is property k of the mesh node; shape 6 is animated by nothing: its default weight applies
    slots = [p.property_slot(mesh_node, k) for k in range(n_shapes)]
    assert slots[6] == -1 and all(s >= 0 for s in slots[:6])
    defaults = np.asarray([10.0, 20.0, 30.0, 40.0, 50.0, 60.0, 35.0], np.float32)
    d_w = ctx.malloc(n_inst * n_shapes * 4)
    p.blend_shape_weights(slots, defaults, d_w.ptr)
    got_w = d_w.download(np.float32, n_inst * n_shapes).reshape(n_inst, n_shapes)
    ref_w = np.asarray([o.props[(mesh_node, k)][1][0] if (mesh_node, k) in o.props else defaults[k] for k in range(n_shapes)], np.float32) / np.float32(100.0)
    assert np.array_equal(got_w[0], ref_w) and np.array_equal(got_w[-1], ref_w)
    mesh = synth.make_mesh(5_003, 16, synth.SEED_BASE + 14, coherent=False)
    L = synth.ANIMATED_VERTEX
    ctx.mesh_upload(base + 60, mesh.to_animated_vertex_aos(), mesh.n_verts, L["stride"], off_pos=L["off_pos"],
                    off_normal=L["off_normal"], off_tangent=L["off_tangent"], off_weights=L["off_weights"],
                    off_indices=L["off_indices"])
    storage, plane, _ = synth.make_blend_shapes(mesh.n_verts, n_shapes, synth.SEED_BASE + 14)
    ctx.mesh_set_blend_shapes(base + 60, storage, n_shapes, plane)
    vb = ctx.malloc(n_inst * mesh.n_verts * L["stride"])
    ctx.lbs_skin_ex(base + 60, d_pal.ptr, 16, n_inst, d_blend_shape_weights=d_w.ptr, n_blend_shapes=n_shapes,
                    d_out_vertices=vb.ptr, out_stride=0)
    ctx.join()
    raw = vb.download(np.uint8, n_inst * mesh.n_verts * L["stride"]).reshape(n_inst, mesh.n_verts, L["stride"])
    bp, bn, bt = orc.apply_blend_shapes(mesh.pos, mesh.normal, mesh.tangent, storage, plane, ref_w)
    ref = orc.lbs_skin(bp, mesh.weights, mesh.indices, o.palette(bone_nodes), bn, bt, threads=0)
    for i in (0, n_inst - 1):
        for off, key in ((L["off_pos"], "pos"), (L["off_normal"], "normal"), (L["off_tangent"], "tangent")):
            got = np.ascontiguousarray(raw[i][:, off:off + 12]).view(np.float32)
            check(got, np.ascontiguousarray(ref[key][:, :3]), True, f"vertex buffer {key} (instance {i})")
    for b in (d_pal, d_w, vb):
        b.free()
    ctx.mesh_free(base + 60)
    o.close()
    p.free()


def test_property_values_survive_a_new_slot_and_a_new_animation(ctx, orc):
    """Adding an animation (and with it a property slot) after frames have run re-creates the property storage on the
    device; what every animation sampled so far, and what was applied last, must still be there -- in the reference an
    animation that does not tick keeps its pose and PlayAnimation nodes keep blending it (pose.rs:107-121)."""
    sc = cases.morph_weights(n_bones=10)
    n_inst = 3
    o, p = run_scenario(ctx, orc, sc, n_instances=n_inst, frames=12, check_every=10 ** 9)
    n_before = p.property_count()
    assert n_before > 0
    poses = [p.read_properties(a) for a in range(len(sc.animations))]
    applied = p.read_properties(-1)
    assert any(int(r["present"].sum()) for r in poses)
    # a clip that animates a property nobody had so far (a new slot) and is a new animation (a new row block)
    tr = A.Track(A.BIND_PROPERTY0 + 9, A.KIND_REAL, [A.Curve([A.CurveKey(0.0, 5.0), A.CurveKey(1.0, 50.0)])])
    A.upload_tracks_data(ctx, p.base_id + 40, A.AnimationTracksData([tr]))
    new_anim = p.add_animation(p.base_id + 40, [2], enabled=False)
    assert p.property_count() == n_before + 1
    for a, before in enumerate(poses):
        after = p.read_properties(a)
        for name in before.dtype.names:
            assert np.array_equal(after[name][:, :n_before], before[name]), (a, name)
        assert not after["present"][:, n_before:].any()
    after = p.read_properties(-1)
    for name in applied.dtype.names:
        assert np.array_equal(after[name][:, :n_before], applied[name]), name
    assert not p.read_properties(new_anim)["present"].any()
    o.close()
    p.free()


def test_palette_outputs_written_by_the_update_itself(ctx, orc):
    """fyx_animator_set_palette_output: the update kernel multiplies global * inv_bind while the matrices are still in
    LDS; the result must equal the separate gather (and the oracle) bit for bit, for two bone lists at once, one with an
    invalid handle, over several frames and after unregistering."""
    sc = cases.layered(n_bones=40)
    o, p = run_scenario(ctx, orc, sc, n_instances=5, frames=6, check_every=10 ** 9)
    base = p.base_id
    lists = {base + 51: list(range(40)), base + 52: [39, -1, 7, 3, 3, 0]}
    bufs = {}
    for bid, nodes in lists.items():
        A.create_bone_list(ctx, bid, base, nodes)
        bufs[bid] = ctx.malloc(5 * len(nodes) * 64)
        p.set_palette_output(bid, bufs[bid].ptr)
    for f in range(6, 10):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par); p.set_parameter(idx, par)
        o.update_machine(sc.dt); p.update_machine(sc.dt)
        for bid, nodes in lists.items():
            got = bufs[bid].download(np.float32, 5 * len(nodes) * 16).reshape(5, len(nodes), 16)
            sep = ctx.malloc(5 * len(nodes) * 64)
            p.palette(bid, sep.ptr)
            ref_gpu = sep.download(np.float32, 5 * len(nodes) * 16).reshape(5, len(nodes), 16)
            sep.free()
            assert np.array_equal(got.view(np.uint32), ref_gpu.view(np.uint32))
            check(got[0], o.palette(nodes), True, f"frame {f} palette output")
            check(got[4], o.palette(nodes), True, f"frame {f} palette output (last instance)")
    with pytest.raises(fyrox_amd.FyxError):
        ctx._check(ctx._l.fyx_bone_list_free(ctx._h, base + 52))       # still registered as an output
    p.set_palette_output(base + 52, 0)
    ctx._check(ctx._l.fyx_bone_list_free(ctx._h, base + 52))
    p.update_machine(sc.dt)                                            # the remaining output still works
    for b in bufs.values():
        b.free()
    o.close()
    p.free()


def test_large_rig_1024_nodes(ctx, orc):
    """The LDS-resident hierarchy walk at its upper limit (1024 nodes = 128 KiB of LDS), deep chains."""
    n = 1024
    rig = synth.make_rig(n, 99, chain_depth=200, exotic=True)
    td, tgt = synth.make_clip(n, 99, 0, n_keys=5, euler_every=10 ** 9)
    sc = cases.Scenario("big", rig, [td], [cases.AnimSpec(0, tgt, time_slice=(0.0, 4 / 30))], None, n_frames=3,
                        has_euler=False)
    o, p = run_scenario(ctx, orc, sc, n_instances=2)
    o.close()
    p.free()
    with pytest.raises(fyrox_amd.FyxError) as e:
        A.create_rig(ctx, 31337, synth.make_rig(1025, 1))
    assert e.value.code == fyrox_amd._native.FYX_ERR_UNSUPPORTED


@pytest.mark.parametrize("n,chain_depth", [(200, 1), (70, 3), (600, 599), (1024, 1023), (17, 16), (1, 1)],
                         ids=["flat_199_children", "levels_of_23", "chain_of_600", "chain_of_1024_takes_the_narrow_kernels", "one_chunk_levels", "one_node"])
def test_one_character_hierarchy_shapes(ctx, orc, n, chain_depth):
    """The one-character update kernels walk the hierarchy by chunks of sixteen nodes (a lane per matrix element): levels wider
    than a chunk, one-node levels, the deepest chain whose table still fits the LDS, and one whose table does not (it takes the
    uploaded control block and the narrow kernels).  One instance, quaternion tracks: bit-exact against the oracle."""
    rig = synth.make_rig(n, 77, chain_depth=chain_depth, exotic=True)
    td, tgt = synth.make_clip(n, 77, 0, n_keys=5, euler_every=10 ** 9)
    sc = cases.Scenario("shape", rig, [td], [cases.AnimSpec(0, tgt, time_slice=(0.0, 4 / 30))], None, n_frames=3, has_euler=False)
    o, p = run_scenario(ctx, orc, sc, n_instances=1)
    o.close()
    p.free()


# ---- fyx_scene_update: many animators, one launch per stage ----------------------------------------------------

def _scene_members():
    """Every scenario of the suite (blend trees, transitions, layers, Euler and quaternion tracks, properties, root motion
    with signals, random machines), with instance counts that put some on the curves-on-the-lanes sampler and some on
    the instances-on-the-lanes one."""
    makes = list(cases.ALL) + list(cases.ALL_RM) + [lambda s=s: cases.random_machine(s) for s in (3, 11, 19)]
    counts = [1, 3, 40, 1, 2, 70, 1, 5]
    return [(mk(), counts[k % len(counts)]) for k, mk in enumerate(makes)]


def test_scene_update_matches_the_oracle_for_every_member(ctx, orc):
    members = _scene_members()
    dt = 1.0 / 48.0     # one step for the whole scene, as Graph::update passes one dt to every node
    os_ = [cases.build_oracle(orc, sc) for sc, _ in members]
    ps = [cases.build_product(ctx, sc, n) for sc, n in members]
    frames = 48
    for f in range(frames):
        for (sc, _), o, p in zip(members, os_, ps):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            for a in sc.removals.get(f, []):
                o.remove_animation(a)
                p.remove_animation(a)
            if sc.machine is None:
                o.update_animations(dt)
            else:
                o.update_machine(dt)
        A.scene_update(ctx, ps, dt)
        if f % 6 == 0 or f == frames - 1:
            for (sc, n), o, p in zip(members, os_, ps):
                check_frame(p, o, sc, n, f)
    # events raised by the host control plane arrive as they do one by one
    for (sc, n), o, p in zip(members, os_, ps):
        gone = {a for lst in sc.removals.values() for a in lst}
        for a in range(len(sc.animations)):
            if a not in gone:
                assert _drain(lambda: p.pop_event(a, n - 1)) == _drain(lambda: o.pop_event(a)), f"{sc.name}: events of animation {a}"


def test_scene_update_is_bit_identical_to_one_by_one_updates(ctx, orc):
    """Two copies of the same scene, one stepped by fyx_scene_update and one animator by animator: every read-back is
    identical bit for bit (Euler scenarios included -- same kernels' bodies, same order), across a change of the
    scene's membership and order."""
    members = _scene_members()
    dt = 1.0 / 48.0
    pa = [cases.build_product(ctx, sc, n) for sc, n in members]
    pb = [cases.build_product(ctx, sc, n) for sc, n in members]
    order = list(range(len(members)))
    for f in range(30):
        if f == 10:
            order = order[::-1]             # same members, other order: other tables, same results
        if f == 20:
            order = order[::2]              # a smaller scene; the rest keep their state
        for k in order:
            sc = members[k][0]
            for idx, par in sc.script.get(f, []):
                pa[k].set_parameter(idx, par)
                pb[k].set_parameter(idx, par)
            for a in sc.removals.get(f, []):
                pa[k].remove_animation(a)
                pb[k].remove_animation(a)
            if sc.machine is None:
                pb[k].update_animations(dt)
            else:
                pb[k].update_machine(dt)
        A.scene_update(ctx, [pa[k] for k in order], dt)
        if f % 5 == 4:
            for k in range(len(members)):
                sc = members[k][0]
                gone = {a for fr, lst in sc.removals.items() if fr <= f for a in lst}
                for what in [A.READ_LOCAL_TRS, A.READ_LOCAL_MATRIX, A.READ_GLOBAL_MATRIX] + \
                        [A.READ_ANIMATION_POSE + a for a in range(len(sc.animations)) if a not in gone]:
                    ga, gb = pa[k].read(what), pb[k].read(what)
                    assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), f"{sc.name} frame {f} read {what}"
                if pa[k].property_count():
                    assert np.array_equal(pa[k].read_properties(-1).view(np.uint32), pb[k].read_properties(-1).view(np.uint32))
                if sc.track_root_motion and sc.machine is not None:
                    assert np.array_equal(pa[k].machine_root_motion(-1).view(np.uint32), pb[k].machine_root_motion(-1).view(np.uint32))


def test_scene_update_writes_registered_palettes_and_feeds_skinning(ctx, orc):
    """Palette outputs of several rigs written by the one update launch; then skinning from them."""
    specs = [(cases.c5_blend_tree(), 3), (cases.transitions(), 1), (cases.layered(), 2)]
    os_ = [cases.build_oracle(orc, sc) for sc, _ in specs]
    ps = [cases.build_product(ctx, sc, n) for sc, n in specs]
    pals = []
    for k, ((sc, n), p) in enumerate(zip(specs, ps)):
        bones = list(range(sc.rig.n_nodes))[::-1]
        A.create_bone_list(ctx, 9100 + k, p.base_id, bones)
        d = ctx.malloc(n * len(bones) * 64)
        p.set_palette_output(9100 + k, d.ptr)
        pals.append((bones, d))
    for f in range(12):
        for (sc, _), o in zip(specs, os_):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
            o.update_machine(1 / 30) if sc.machine is not None else o.update_animations(1 / 30)
        for (sc, _), p in zip(specs, ps):
            for idx, par in sc.script.get(f, []):
                p.set_parameter(idx, par)
        A.scene_update(ctx, ps, 1 / 30)
    ctx.sync()
    for (sc, n), o, (bones, d) in zip(specs, os_, pals):
        got = d.download(np.float32, n * len(bones) * 16).reshape(n, len(bones), 16)
        ref = o.palette(bones)
        for i in range(n):
            check(got[i], ref, not sc.has_euler, f"{sc.name} palette (instance {i})")


@pytest.mark.parametrize("make", [cases.c5_blend_tree, cases.player_only, cases.transitions, cases.layered], ids=lambda f: f.__name__)
def test_frame_forms_switched_between_frames(ctx, orc, make):
    """One character's frame has three forms -- sampler and update in one launch (anim.one_launch), two launches with the control block
    in the kernel arguments, two launches with the block uploaded (anim.inline_ctrl = 0) -- and the update two kernel forms
    (anim.update_lean).  Switching between them from frame to frame must be invisible: the one-launch form's device counter only
    counts its own launches."""
    sc = make()
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, 2)
    forms = [(1, 1, 1), (0, 1, 1), (1, 1, 0), (0, 0, 1), (1, 1, 1), (1, 0, 0), (1, 1, 1), (0, 1, 0)]
    try:
        for f in range(24):
            one, inline, lean = forms[f % len(forms)]
            ctx.set_option("anim.one_launch", one)
            ctx.set_option("anim.inline_ctrl", inline)
            ctx.set_option("anim.update_lean", lean)
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            if sc.machine is None:
                o.update_animations(sc.dt)
                p.update_animations(sc.dt)
            else:
                o.update_machine(sc.dt)
                p.update_machine(sc.dt)
            check_frame(p, o, sc, 2, f)
    finally:
        ctx.set_option("anim.one_launch", 1)
        ctx.set_option("anim.inline_ctrl", 1)
        ctx.set_option("anim.update_lean", 1)
    o.close()
    p.free()


def test_one_launch_frames_equal_two_launch_frames_over_many_frames(ctx):
    """The one-launch frame's update half reads the records its sampler half wrote in the SAME kernel, from other compute units
    (write-through stores, a device counter, an acquire).  3000 frames of two animators in the same state -- one in each form, a
    skinning launch behind every update to keep the memory system busy -- palettes bit for bit after every frame: a record read
    before it had landed would be last frame's."""
    sc = cases.c5_blend_tree()
    nb = sc.rig.n_nodes
    ps, pals = [], []
    for k in range(2):
        p = cases.build_product(ctx, sc, 1)
        A.create_bone_list(ctx, 9500 + k, p.base_id, list(range(nb)))
        d = ctx.malloc(nb * 64)
        p.set_palette_output(9500 + k, d.ptr)
        ps.append(p)
        pals.append(d)
    mesh = synth.make_mesh(20_000, nb, synth.SEED_BASE + 9)
    ctx.mesh_upload_soa(9510, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs = (ctx.malloc(20_000 * 12 + 64), ctx.malloc(20_000 * 12 + 64), ctx.malloc(20_000 * 16 + 64))
    try:
        for f in range(3000):
            for k, one in ((0, 1), (1, 0)):
                ctx.set_option("anim.one_launch", one)
                ps[k].update_machine(sc.dt)
                ctx.lbs_skin_device(9510, pals[k].ptr, nb, 1, outs[0].ptr, outs[1].ptr, outs[2].ptr)
            a, b = pals[0].download(np.uint32, nb * 16), pals[1].download(np.uint32, nb * 16)
            assert np.array_equal(a, b), f"frame {f}: the one-launch frame differs from the two-launch frame"
    finally:
        ctx.set_option("anim.one_launch", 1)
    for p in ps:
        p.free()


def test_scene_job_array_follows_what_changes_between_frames(ctx, orc):
    """fyx_scene_update keeps its per-animator job records on the device and sends them again only when their bytes change.  What
    changes here between frames: a palette output moved to another buffer, the list of animators reordered and shortened, an
    animator given another machine state (its fold program changes length).  Palettes against the oracle after every frame."""
    specs = [(cases.c5_blend_tree(euler_every=10 ** 6), 1), (cases.transitions(), 2), (cases.by_index(), 1), (cases.player_only(euler_every=10 ** 6), 3)]
    os_ = [cases.build_oracle(orc, sc) for sc, _ in specs]
    ps = [cases.build_product(ctx, sc, n) for sc, n in specs]
    bufs = []
    for k, ((sc, n), p) in enumerate(zip(specs, ps)):
        bones = list(range(sc.rig.n_nodes))
        A.create_bone_list(ctx, 9300 + k, p.base_id, bones)
        pair = [ctx.malloc(n * len(bones) * 64), ctx.malloc(n * len(bones) * 64)]
        p.set_palette_output(9300 + k, pair[0].ptr)
        bufs.append((bones, pair))
    orders = [[0, 1, 2, 3], [0, 1, 2, 3], [3, 2, 1, 0], [3, 2, 1, 0], [1, 3], [1, 3], [0, 1, 2, 3], [2, 0, 3, 1]]
    stepped = [0] * len(specs)
    for f in range(16):
        order = orders[f % len(orders)]
        which = (f // 3) % 2                       # the palette buffers move every third frame
        for k in order:
            sc, n = specs[k]
            ps[k].set_palette_output(9300 + k, bufs[k][1][which].ptr)
            for idx, par in sc.script.get(stepped[k], []):
                os_[k].set_parameter(idx, par)
                ps[k].set_parameter(idx, par)
            os_[k].update_machine(1 / 30) if sc.machine is not None else os_[k].update_animations(1 / 30)
            stepped[k] += 1
        A.scene_update(ctx, [ps[k] for k in order], 1 / 30)
        ctx.sync()
        for k in order:
            sc, n = specs[k]
            bones, pair = bufs[k]
            got = pair[which].download(np.float32, n * len(bones) * 16).reshape(n, len(bones), 16)
            ref = os_[k].palette(bones)
            for i in range(n):
                check(got[i], ref, True, f"frame {f}: {sc.name} palette (instance {i})")
    for o in os_:
        o.close()
    for p in ps:
        p.free()


def test_scene_update_argument_errors(ctx):
    sc = cases.by_index()
    p = cases.build_product(ctx, sc)
    A.scene_update(ctx, [], 1 / 60)     # an empty scene is fine
    with pytest.raises(fyrox_amd.FyxError) as e:
        A.scene_update(ctx, [p, p], 1 / 60)
    assert e.value.code == fyrox_amd._native.FYX_ERR_INVALID_ARG
    ids = np.asarray([p.id, 0xdead], np.uint64)
    assert ctx._l.fyx_scene_update(ctx._h, ids.ctypes.data_as(__import__("ctypes").c_void_p), 2, 1 / 60) == fyrox_amd._native.FYX_ERR_UNKNOWN_ID


def test_overlapped_update_then_sync_then_readback_without_a_skinning_call(orc):
    """anim.overlap: a pose update runs on the context's second frame stream.  An animator's FIRST frame joins the streams on its way
    (device state is created, the control block grows) -- which used to clear the "second stream has work" flag before the frame's
    kernels were enqueued there, so a fyx_sync that followed the update directly (no skinning call in between to set the flag again)
    returned before they had run.  Fresh context, every frame: update -> sync -> readback, palettes against the oracle."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb = sc.rig.n_nodes
    with fyrox_amd.Context(0) as c2:
        c2.set_option("anim.overlap", 1)
        for one_launch in (1, 0):
            c2.set_option("anim.one_launch", one_launch)
            for n_inst in (1, 40):
                o = cases.build_oracle(orc, sc)
                p = cases.build_product(c2, sc, n_inst)
                A.create_bone_list(c2, p.base_id + 50, p.base_id, list(range(nb)))
                pals = [c2.malloc(n_inst * nb * 64) for _ in range(2)]
                for d in pals:
                    d.upload(np.zeros(n_inst * nb * 16, np.float32))
                for f in range(6):
                    p.set_palette_output(p.base_id + 50, pals[f & 1].ptr)
                    o.update_machine(sc.dt)
                    p.update_machine(sc.dt)
                    c2.sync()
                    got = pals[f & 1].download(np.float32, n_inst * nb * 16).reshape(n_inst, nb, 16)
                    ref = o.palette(list(range(nb)))
                    for i in (0, n_inst - 1):
                        assert np.array_equal(got[i].view(np.uint32), ref.view(np.uint32)), f"one_launch={one_launch} instances={n_inst} frame {f}"
                o.close()
                p.free()
                for d in pals:
                    d.free()


@pytest.mark.parametrize("n_instances", [1, 3])
def test_list_poses_appear_while_the_machine_runs(ctx, orc, n_instances):
    """The animator goes from one device animation per animation to two (Animator::shadows) in the middle of playback: a machine blends
    two plain clips and a third whose tracks are all switched off; after six frames they are switched on -- and that clip's node poses
    are lists (two Positions on a node, a Real bound to Position, ...).  Pose records, sampled values and transforms of the running
    clips move to their new places (ensure_device_state's spread); every frame before and after is the oracle's."""
    n_bones, seed = 16, synth.SEED_BASE + 31
    rig = synth.make_rig(n_bones, seed)
    td0, t0 = synth.make_clip(n_bones, seed, 0, euler_every=10 ** 9)
    td1, t1 = synth.make_clip(n_bones, seed, 1, euler_every=10 ** 9)
    td2, t2 = cases._listy_tracks(n_bones, seed, 0)
    off = np.zeros(len(td2.tracks), np.uint8)
    anims = [cases.AnimSpec(0, t0, speed=1.1), cases.AnimSpec(1, t1, speed=-0.9), cases.AnimSpec(2, t2, enabled_tracks=off, speed=1.3)]
    layer = A.MachineLayer(
        nodes=[A.PlayAnimation(0), A.PlayAnimation(1), A.PlayAnimation(2),
               A.BlendAnimations([A.BlendPose(0, 0.6), A.BlendPose(2, 0.5)]),
               A.BlendAnimations([A.BlendPose(2, 1.0), A.BlendPose(3, 0.7), A.BlendPose(1, 0.4)])],
        states=[A.State(4)])
    sc = cases.Scenario("late_list_clip", rig, [td0, td1, td2], anims, A.Machine(parameters=[], layers=[layer]), n_frames=30, has_euler=False)
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, n_instances)
    try:
        for f in range(30):
            if f == 6:
                for t, node in enumerate(t2):
                    orc._alib().fo_animation_bind(o.anims[2], t, int(node), 1)
                    p.set_track_enabled(2, t, True)
            o.update_machine(sc.dt)
            p.update_machine(sc.dt)
            check_frame(p, o, sc, n_instances, f)
    finally:
        o.close()
        p.free()


def test_a_scene_with_a_member_whose_poses_are_lists(ctx, orc):
    """fyx_scene_update over characters of which one blends list poses (the two-record fold has no scene form: such a scene runs its
    members one by one): every member's transforms are its oracle's, frame by frame."""
    scs = [cases.c5_blend_tree(n_bones=24, euler_every=10 ** 6), cases.duplicate_bindings(), cases.transitions(), cases.duplicate_properties()]
    os_, ps = [cases.build_oracle(orc, sc) for sc in scs], [cases.build_product(ctx, sc, 2) for sc in scs]
    try:
        for f in range(24):
            for sc, o, p in zip(scs, os_, ps):
                for idx, par in sc.script.get(f, []):
                    o.set_parameter(idx, par)
                    p.set_parameter(idx, par)
                o.update_machine(sc.dt)
            A.scene_update(ctx, ps, scs[0].dt)
            for sc, o, p in zip(scs, os_, ps):
                check_frame(p, o, sc, 2, f)
    finally:
        for o in os_:
            o.close()
        for p in ps:
            p.free()
