"""Self-consistency of the animation/machine oracle (parity unpinned by the reference for these
functions -- see oracle/fyrox_oracle.h): the fold semantics of pose.rs / value.rs and the order of
operations of Animation::tick and MachineLayer::evaluate_pose, checked against hand-derived cases."""
import numpy as np
import pytest

from fyrox_amd import anim as A
from fyrox_amd import synth

import anim_cases as cases


def _const_clip(n_bones, pos, rot=None, scale=None, bones=None):
    """A clip whose tracks hold a single key: the pose is constant."""
    tracks, target = [], []
    for b in (range(n_bones) if bones is None else bones):
        if pos is not None:
            tracks.append(A.Track(A.BIND_POSITION, A.KIND_VEC3, [A.Curve([A.CurveKey(0.0, float(v))]) for v in pos]))
            target.append(b)
        if rot is not None:
            tracks.append(A.Track(A.BIND_ROTATION, A.KIND_QUAT, [A.Curve([A.CurveKey(0.0, float(v))]) for v in rot]))
            target.append(b)
        if scale is not None:
            tracks.append(A.Track(A.BIND_SCALE, A.KIND_VEC3, [A.Curve([A.CurveKey(0.0, float(v))]) for v in scale]))
            target.append(b)
    return A.AnimationTracksData(tracks), np.asarray(target, np.int32)


def _scene(orc, n_bones, clips, machine):
    rig = synth.make_rig(n_bones, 5)
    s = orc.AnimScene(rig)
    for td, tgt in clips:
        s.add_animation(s.add_tracks_data(td), tgt, time_slice=(0.0, 1.0))
    s.set_machine(machine)
    return s


def test_blend_is_a_sequential_fold_not_a_weighted_sum(orc):
    # BlendAnimations (blend.rs:136-164): out = p0; out = mix(out, p1, w1); out = mix(out, p2, w2); w0 unused
    p = [(1.0, 0.0, 0.0), (0.0, 2.0, 0.0), (0.0, 0.0, 4.0)]
    clips = [_const_clip(2, pos=v) for v in p]
    nodes = [A.PlayAnimation(0), A.PlayAnimation(1), A.PlayAnimation(2),
             A.BlendAnimations([A.BlendPose(0, 123.0), A.BlendPose(1, 0.5), A.BlendPose(2, 0.25)])]
    s = _scene(orc, 2, clips, A.Machine([], [A.MachineLayer(nodes=nodes, states=[A.State(3)])]))
    s.update_machine(1 / 60)
    f32 = np.float32
    a = np.asarray(p[0], f32) * (f32(1) - f32(0.5)) + np.asarray(p[1], f32) * f32(0.5)
    a = a * (f32(1) - f32(0.25)) + np.asarray(p[2], f32) * f32(0.25)
    got = s.machine_pose()
    assert np.array_equal(got[0, 0:3], a) and np.array_equal(got[1, 0:3], a)
    assert got[0, 3].view(np.uint32) == 1          # only Position is present
    assert np.array_equal(s.node_trs()[0, 0:3], a)
    s.close()


def test_empty_node_pose_copies_the_other_and_ignores_the_weight(orc):
    # NodePose::blend_with (pose.rs:41-47): clip 1 animates only bone 1 -> for bone 0 nothing to blend,
    # for bone 1 the first pose is empty so the second is COPIED even with weight 0.1
    clips = [_const_clip(2, pos=(1.0, 1.0, 1.0), bones=[0]), _const_clip(2, pos=(5.0, 6.0, 7.0), bones=[1])]
    nodes = [A.PlayAnimation(0), A.PlayAnimation(1), A.BlendAnimations([A.BlendPose(0, 1.0), A.BlendPose(1, 0.1)])]
    s = _scene(orc, 2, clips, A.Machine([], [A.MachineLayer(nodes=nodes, states=[A.State(2)])]))
    s.update_machine(1 / 60)
    got = s.machine_pose()
    assert got[0, 0:3].tolist() == [1.0, 1.0, 1.0]
    assert got[1, 0:3].tolist() == [5.0, 6.0, 7.0]
    s.close()


def test_values_only_in_the_other_pose_are_dropped(orc):
    # BoundValueCollection::blend_with (value.rs:438-444) iterates SELF's values only
    clips = [_const_clip(1, pos=(1.0, 0.0, 0.0)), _const_clip(1, pos=(3.0, 0.0, 0.0), scale=(2.0, 2.0, 2.0))]
    nodes = [A.PlayAnimation(0), A.PlayAnimation(1), A.BlendAnimations([A.BlendPose(0, 1.0), A.BlendPose(1, 0.5)])]
    s = _scene(orc, 1, clips, A.Machine([], [A.MachineLayer(nodes=nodes, states=[A.State(2)])]))
    s.update_machine(1 / 60)
    got = s.machine_pose()
    assert got[0, 3].view(np.uint32) == 1 and got[0, 0] == 2.0   # scale dropped, position blended
    assert s.node_trs()[0, 8:11].tolist() == [1.0, 1.0, 1.0]    # node scale untouched
    s.close()


def test_quaternion_blend_takes_the_short_way_and_normalises(orc):
    q0 = (0.0, 0.0, 0.0, 1.0)
    q1 = (0.0, -0.6, 0.0, -0.8)     # same rotation hemisphere flipped: dot < 0
    clips = [_const_clip(1, None, rot=q0), _const_clip(1, None, rot=q1)]
    nodes = [A.PlayAnimation(0), A.PlayAnimation(1), A.BlendAnimations([A.BlendPose(0, 1.0), A.BlendPose(1, 0.5)])]
    s = _scene(orc, 1, clips, A.Machine([], [A.MachineLayer(nodes=nodes, states=[A.State(2)])]))
    s.update_machine(1 / 60)
    r = s.machine_pose()[0, 4:8].astype(np.float64)
    assert abs(np.linalg.norm(r) - 1.0) < 1e-6
    # value.rs:449-454 negates SELF (q0), so the result lies between -q0 and q1
    expect = np.array([0.0, -0.3, 0.0, -0.9]) / np.linalg.norm([0.3, 0.9])
    assert np.allclose(r, expect, atol=1e-6)
    s.close()


def test_tick_samples_before_advancing_and_wraps(orc):
    # Animation::tick (lib.rs:471-496): pose at the OLD time; looped time wraps with wrapf
    keys = [A.CurveKey(0.0, 0.0), A.CurveKey(1.0, 10.0)]
    td = A.AnimationTracksData([A.Track(A.BIND_POSITION, A.KIND_VEC3, [A.Curve(keys), A.Curve(keys), A.Curve(keys)])])
    rig = synth.make_rig(1, 3)
    s = orc.AnimScene(rig)
    s.add_animation(s.add_tracks_data(td), [0], time_slice=(0.0, 1.0), speed=1.0)
    s.update_animations(0.75)
    assert s.animation_pose(0)[0, 0] == 0.0 and s.animation_state(0)["time_position"] == 0.75
    s.update_animations(0.5)
    assert s.animation_pose(0)[0, 0] == 7.5
    assert s.animation_state(0)["time_position"] == orc.wrapf(np.float32(1.25), 0.0, 1.0) == 0.25
    s.close()


def test_non_looped_animation_ends_and_clamps(orc):
    td, tgt = _const_clip(1, pos=(1.0, 2.0, 3.0))
    rig = synth.make_rig(1, 3)
    s = orc.AnimScene(rig)
    s.add_animation(s.add_tracks_data(td), tgt, time_slice=(0.0, 0.5), looped=False)
    for _ in range(40):
        s.update_animations(1 / 60)
    st = s.animation_state(0)
    assert st["time_position"] == 0.5 and st["has_ended"]
    s.close()


def test_transition_blend_factor_is_read_before_the_update(orc):
    # layer.rs:658-667: frame of the switch blends with factor 0 (= copy of source), then 1/3, 2/3, done
    clips = [_const_clip(1, pos=(0.0, 0.0, 0.0)), _const_clip(1, pos=(3.0, 0.0, 0.0))]
    layer = A.MachineLayer(nodes=[A.PlayAnimation(0), A.PlayAnimation(1)], states=[A.State(0), A.State(1)],
                           transitions=[A.Transition(0, 1, 0.75, ("parameter", 0))])
    s = _scene(orc, 1, clips, A.Machine([A.Parameter(A.PARAM_RULE, True)], [layer]))
    xs, states = [], []
    for _ in range(5):
        s.update_machine(0.25)
        xs.append(float(s.machine_pose()[0, 0]))
        states.append(s.layer_state(0))
    f32 = np.float32
    assert xs[0] == 0.0 and states[0] == (-1, 0)
    assert xs[1] == float(f32(3.0) * (f32(0.25) / f32(0.75)))
    assert xs[2] == float(f32(0.0) * (f32(1) - f32(0.5) / f32(0.75)) + f32(3.0) * (f32(0.5) / f32(0.75)))
    assert states[2] == (1, -1)                      # done after the third update
    assert xs[3] == 3.0 and xs[4] == 3.0
    s.close()


def test_layer_mask_removes_nodes_after_blending(orc):
    clips = [_const_clip(3, pos=(1.0, 1.0, 1.0))]
    layer = A.MachineLayer(nodes=[A.PlayAnimation(0)], states=[A.State(0)], mask=[1])
    s = _scene(orc, 3, clips, A.Machine([], [layer]))
    s.update_machine(1 / 60)
    bits = s.machine_pose()[:, 3].view(np.uint32).tolist()
    assert bits == [1, 0, 1]
    s.close()


def test_every_scenario_runs_and_moves_the_skeleton(orc):
    for make in cases.ALL:
        sc = make()
        o = cases.build_oracle(orc, sc)
        before = o.global_matrices().copy()
        for f in range(10):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
            o.update_machine(sc.dt) if sc.machine else o.update_animations(sc.dt)
        after = o.global_matrices()
        assert np.isfinite(after).all() and not np.array_equal(before, after), sc.name
        o.close()
