"""Run-time edits of a Machine (CPU, control-only context).

The engine edits its Machine in place between two frames (machine/mod.rs:280-312, layer.rs:202-283 and 412-525,
transition.rs:290, node/blendspace.rs:246-310) and the next evaluate_pose sees the edit with every other piece of
run-time state untouched.  Behind the C ABI the definition is re-sent and the run-time state carried over
(fyx_machine_clear + builder calls + fyx_layer_set_state / set_transition_state / set_node_state: fyrox_hip.h,
`Animator.rebuild_machine` is what the engine-side shim does).  Here the same edit is made IN PLACE on oracle2's machine
objects and through a rebuild on the product, in the middle of scripted runs (inside transitions and cross-fades too), and
every later frame must agree bit for bit: the fold programs executed with the oracle's primitives give the oracle's node
transforms, the layers are in the same states and have queued the same events, the animations hold the same clocks."""
import copy

import numpy as np
import pytest

import anim_cases as cases
import fyrox_amd
import oracle
import oracle2
from fyrox_amd import anim as A
from test_anim_control import _drain, run_program


@pytest.fixture()
def cctx():
    c = fyrox_amd.Context(control_only=True)
    yield c
    c.close()


# ---- edits: (machine description) -> None if not applicable, else (new description, in-place edit of oracle2, maps) ----
# maps = dict(state_maps=..., transition_maps=..., node_maps=...): OLD index -> NEW index per old layer, None = gone

def _layer_with(m, kind):
    for li, layer in enumerate(m.layers):
        for ni, n in enumerate(layer.nodes):
            if isinstance(n, kind):
                return li, ni
    return None


def edit_retime(m):
    """Transition::transition_time through transition_mut (layer.rs:451)"""
    if not any(l.transitions for l in m.layers):
        return None
    new = copy.deepcopy(m)
    for l in new.layers:
        for t in l.transitions:
            t.transition_time = float(np.float32(t.transition_time * 1.5 + 0.05))

    def in_place(om):
        for lo, ln in zip(om.layers, new.layers):
            for h, t in enumerate(ln.transitions):
                lo.edit_transition(h, transition_time=t.transition_time)
    return new, in_place, {}


def edit_recondition(m):
    """Transition::set_condition (transition.rs:290)"""
    for li, l in enumerate(m.layers):
        if l.transitions:
            new = copy.deepcopy(m)
            for t in new.layers[li].transitions[::2]:
                t.condition = ("not", t.condition)

            def in_place(om, li=li, new=new):
                for h, t in enumerate(new.layers[li].transitions):
                    om.layers[li].edit_transition(h, condition=t.condition)
            return new, in_place, {}
    return None


def edit_reweight(m):
    """BlendAnimations::pose_sources_mut: constants moved, a parameter weight becomes a constant and back"""
    hit = _layer_with(m, A.BlendAnimations)
    if hit is None:
        return None
    li, ni = hit
    new = copy.deepcopy(m)
    node = new.layers[li].nodes[ni]
    for k, b in enumerate(node.pose_sources):
        if b.parameter is None:
            b.weight = float(np.float32(min(1.0, b.weight * 0.5 + 0.3)))
            if k % 2 == 1 and new.parameters and new.parameters[0].kind == A.PARAM_WEIGHT:
                b.parameter = 0
        else:
            b.parameter, b.weight = None, 0.625
    node.pose_sources.reverse()

    def in_place(om):
        om.layers[li].edit_node(ni, node)
    return new, in_place, {}


def edit_swap_clip(m):
    """PlayAnimation::set_animation"""
    plays = [(li, ni) for li, l in enumerate(m.layers) for ni, n in enumerate(l.nodes) if isinstance(n, A.PlayAnimation)]
    if len(plays) < 2:
        return None
    new = copy.deepcopy(m)
    (l0, n0), (l1, n1) = plays[0], plays[-1]
    a0, a1 = m.layers[l0].nodes[n0].animation, m.layers[l1].nodes[n1].animation
    new.layers[l0].nodes[n0].animation, new.layers[l1].nodes[n1].animation = a1, a0

    def in_place(om):
        om.layers[l0].edit_node(n0, new.layers[l0].nodes[n0])
        om.layers[l1].edit_node(n1, new.layers[l1].nodes[n1])
    return new, in_place, {}


def edit_by_index_times(m):
    """IndexedBlendInput::blend_time in the middle of a cross-fade: prev_index and the accumulated time stay"""
    hit = _layer_with(m, A.BlendAnimationsByIndex)
    if hit is None:
        return None
    li, ni = hit
    new = copy.deepcopy(m)
    for i in new.layers[li].nodes[ni].inputs:
        i.blend_time = float(np.float32(i.blend_time * 2.0 + 0.02))

    def in_place(om):
        om.layers[li].edit_node(ni, new.layers[li].nodes[ni])
    return new, in_place, {}


def edit_blend_space_points(m):
    """BlendSpace::points_mut / set_points (node/blendspace.rs:252-270)"""
    hit = _layer_with(m, A.BlendSpace)
    if hit is None:
        return None
    li, ni = hit
    new = copy.deepcopy(m)
    for p in new.layers[li].nodes[ni].points:
        p.position = (float(np.float32(p.position[0] * 0.9 + 0.03)), float(np.float32(p.position[1] * 1.1 - 0.02)))

    def in_place(om):
        om.layers[li].edit_node(ni, new.layers[li].nodes[ni])
    return new, in_place, {}


def edit_grow(m):
    """add_node / add_state / add_transition between frames (layer.rs:202-245): a new state reachable from every other one
    on a Rule parameter, and a way back"""
    rule = next((i for i, p in enumerate(m.parameters) if p.kind == A.PARAM_RULE), None)
    li = next((i for i, l in enumerate(m.layers) if l.states), None)
    if li is None:
        return None
    new = copy.deepcopy(m)
    l = new.layers[li]
    anim = next((n.animation for n in l.nodes if isinstance(n, A.PlayAnimation)), 0)
    node = A.PlayAnimation(anim)
    l.nodes.append(node)
    state = A.State(len(l.nodes) - 1, on_enter_actions=[(A.ACTION_REWIND, anim)])
    l.states.append(state)
    s_new = len(l.states) - 1
    cond = ("parameter", rule) if rule is not None else ("not", ("parameter", 99))    # parameter 99 does not exist: false
    added = [A.Transition(s, s_new, 0.125, cond) for s in range(s_new)] + [A.Transition(s_new, 0, 0.2, ("not", cond))]
    l.transitions.extend(added)

    def in_place(om):
        lo = om.layers[li]
        lo.add_node(node)
        lo.add_state(state)
        for t in added:
            lo.add_transition(t)
    return new, in_place, {}


def edit_layer_weight_and_mask(m):
    """MachineLayer::set_weight / set_mask (layer.rs:530-545) sent with the rest of the definition"""
    if len(m.layers) < 2:
        return None
    new = copy.deepcopy(m)
    new.layers[1].weight = float(np.float32(new.layers[1].weight * 0.5 + 0.1))
    new.layers[1].mask = sorted(set(new.layers[1].mask) ^ {1, 2})

    def in_place(om):
        om.layers[1].weight = oracle2.anim.F(new.layers[1].weight)
        om.layers[1].mask = set(new.layers[1].mask)
    return new, in_place, {}


def edit_permute(m):
    """No edit at all for the game: the shim happens to flatten the state pool in another order (reversed), and the
    transitions rotated within groups that cannot compete (the FIRST transition out of the active state whose condition
    holds fires, layer.rs:605-651, so the relative order of transitions with one source is kept).  Exercises the index
    maps of the state restore.  The tests apply it as the last edit of a run: the in-place functions of the other edits
    take the oracle's and the product's indices to be the same."""
    new = copy.deepcopy(m)
    state_maps, transition_maps = {}, {}
    for li, l in enumerate(new.layers):
        ns, nt = len(l.states), len(l.transitions)
        smap = {s: ns - 1 - s for s in range(ns)}
        # transitions sorted by (source descending, old index): every source's group keeps its order
        order = sorted(range(nt), key=lambda t: (-l.transitions[t].source, t))
        tmap = {old: new_i for new_i, old in enumerate(order)}
        entry = l.entry_state if l.entry_state is not None else (0 if ns else None)
        l.states.reverse()
        l.transitions = [l.transitions[t] for t in order]
        for t in l.transitions:
            t.source = smap.get(t.source, t.source)
            t.dest = smap.get(t.dest, t.dest)
        l.entry_state = None if entry is None else smap[entry]
        state_maps[li], transition_maps[li] = smap, tmap
    return new, (lambda om: None), {"state_maps": state_maps, "transition_maps": transition_maps}


EDITS = [edit_retime, edit_recondition, edit_reweight, edit_swap_clip, edit_by_index_times, edit_blend_space_points, edit_grow,
         edit_layer_weight_and_mask, edit_permute]


# ---- one scripted run with edits at given frames ----------------------------------------------------------------------

def _inverse(maps, li):
    mp = (maps or {}).get(li)
    return None if mp is None else {v: k for k, v in mp.items()}


def _run(cctx, sc, edits_at, n_frames=None, n_instances=2):
    """`edits_at`: {frame: edit function}.  Returns the number of edits that applied."""
    o = cases.build_oracle(oracle2, sc)
    p = cases.build_product(cctx, sc, n_instances=n_instances)
    desc = sc.machine                      # the product's current definition
    inv_state = {li: None for li in range(len(desc.layers))}       # NEW (product) index -> the oracle's index
    inv_trans = {li: None for li in range(len(desc.layers))}
    trs = o.node_trs()
    applied = 0
    n_frames = min(sc.n_frames, 64) if n_frames is None else n_frames
    for f in range(n_frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        if f in edits_at:
            res = edits_at[f](desc)
            if res is not None:
                new, in_place, maps = res
                if f % 2:                                    # half of the edits with the layers' events still queued:
                    for li in range(len(desc.layers)):      # rebuild_machine keeps them (the reference's queues survive edits)
                        ref = _drain(lambda: o.pop_layer_event(li))
                        for inst in range(n_instances):
                            got = _drain(lambda: p.pop_layer_event(li, inst))
                            assert _map_events(got, inv_state[li], inv_trans[li]) == ref
                in_place(o.machine)
                p.rebuild_machine(desc, new, **maps)
                for li in range(len(new.layers)):           # compose with earlier permutations
                    for inv, key in ((inv_state, "state_maps"), (inv_trans, "transition_maps")):
                        step = _inverse(maps.get(key), li)
                        if step is not None:
                            prev = inv.get(li)
                            inv[li] = {k: (v if prev is None else prev.get(v, v)) for k, v in step.items()}
                        elif li not in inv:
                            inv[li] = None
                desc = new
                applied += 1
        plan = p.plan(1, sc.dt)
        o.update_machine(sc.dt)
        offs = plan["offsets"]
        for inst in range(1, n_instances):      # every instance runs the same script
            assert np.array_equal(plan["ops"][offs[0]:offs[1]], plan["ops"][offs[inst]:offs[inst + 1]]), f
        for a in range(len(sc.animations)):
            assert p.animation_state(a, n_instances - 1) == o.animation_state(a), (sc.name, f, a)
        for li in range(len(desc.layers)):
            s, t = p.layer_state(li, 0)
            back_s = s if inv_state.get(li) is None or s < 0 else inv_state[li][s]
            back_t = t if inv_trans.get(li) is None or t < 0 else inv_trans[li][t]
            assert (back_s, back_t) == o.layer_state(li), (sc.name, f, li)
            if f % 3 == 2 or f == n_frames - 1:      # queues are read every third frame: edits find events still queued
                ref = _drain(lambda: o.pop_layer_event(li))
                for inst in range(n_instances):
                    got = _drain(lambda: p.pop_layer_event(li, inst))
                    assert _map_events(got, inv_state.get(li), inv_trans.get(li)) == ref, (sc.name, f, li)
        poses = [o.animation_pose(a) for a in range(len(sc.animations))]
        excluded = [set(l.mask) for l in desc.layers]
        trs = run_program(oracle, plan["ops"][offs[0]:offs[1]], poses, excluded, trs)
        assert np.array_equal(trs.view(np.uint32), o.node_trs().view(np.uint32)), f"{sc.name}: frame {f}"
    o.close()
    p.free()
    return applied


def _map_events(events, inv_state, inv_trans):
    out = []
    for kind, a, b in events:
        if kind in (A.EVENT_STATE_ENTER, A.EVENT_STATE_LEAVE):
            a = a if inv_state is None or a < 0 else inv_state[a]
        elif kind == A.EVENT_ACTIVE_STATE_CHANGED:
            a = a if inv_state is None or a < 0 else inv_state[a]
            b = b if inv_state is None or b < 0 else inv_state[b]
        elif kind == A.EVENT_ACTIVE_TRANSITION_CHANGED:
            a = a if inv_trans is None or a < 0 else inv_trans[a]
        out.append((kind, a, b))
    return out


SCENARIOS = [cases.transitions, cases.by_index, cases.blend_space, cases.layered, cases.c5_blend_tree, cases.gltf_like]


def _combos():
    """every (scenario, edit) pair the edit applies to"""
    out = []
    for make in SCENARIOS:
        m = make().machine
        out += [pytest.param(make, edit, id=f"{edit.__name__}-{make.__name__}") for edit in EDITS if edit(m) is not None]
    return out


@pytest.mark.parametrize("make,edit", _combos())
def test_edit_in_place_equals_rebuild_with_state_carried_over(cctx, make, edit):
    sc = make()
    # twice: early, and at a frame that lies inside a transition / cross-fade for the scripted scenarios
    assert _run(cctx, sc, {7: edit, 33: edit} if edit is not edit_permute else {33: edit}) == (2 if edit is not edit_permute else 1)


def test_edits_at_every_frame_of_a_transition(cctx):
    """the state restore at every phase of a transition: elapsed_time / blend_factor in flight, the frame it fires, the
    frame it completes"""
    for f in range(4, 20):
        _run(cctx, cases.transitions(), {f: edit_retime, f + 1: edit_permute}, n_frames=40, n_instances=1)


def test_edits_at_every_frame_of_a_cross_fade(cctx):
    for f in range(3, 14):
        _run(cctx, cases.by_index(), {f: edit_by_index_times, f + 2: edit_grow, f + 3: edit_permute}, n_frames=40, n_instances=1)


@pytest.mark.parametrize("seed", range(40))
def test_random_machines_survive_a_sequence_of_edits(cctx, seed):
    sc = cases.random_machine(seed)
    rng = np.random.default_rng(seed)
    frames = sorted(int(x) for x in rng.choice(np.arange(2, 40), size=5, replace=False))
    edits = {f: EDITS[int(rng.integers(0, len(EDITS) - 1))] for f in frames[:-1]}     # EDITS[-1] is edit_permute:
    edits[frames[-1]] = edit_permute                                                   # last (see its docstring)
    _run(cctx, sc, edits, n_frames=48)


def test_removed_state_and_transitions_are_compacted_away(cctx):
    """Pool::free leaves a hole and every other handle keeps its meaning; the shim re-sends the surviving items densely
    and maps the state through its handle tables.  The added state is removed again before it was ever entered."""
    sc = cases.transitions()
    grown = {}

    def grow(m):
        res = edit_grow(m)
        grown["n_states"], grown["n_transitions"] = len(m.layers[0].states), len(m.layers[0].transitions)
        return res

    def shrink(m):
        ns, nt = grown["n_states"], grown["n_transitions"]
        new = copy.deepcopy(m)
        l = new.layers[0]
        dead_t = [t for t in range(len(l.transitions)) if l.transitions[t].source == ns or l.transitions[t].dest == ns]
        # the game frees the first state it had before as well as the one it added?  no: only the added one
        keep_t = [t for t in range(len(l.transitions)) if t not in dead_t]
        l.transitions = [l.transitions[t] for t in keep_t]
        l.states = l.states[:ns]
        assert len(l.transitions) == nt

        def in_place(om):
            for t in dead_t:
                om.layers[0].remove_transition(t)
            om.layers[0].remove_state(ns)
        tmap = {t: (keep_t.index(t) if t in keep_t else None) for t in range(len(m.layers[0].transitions))}
        smap = {s: (s if s < ns else None) for s in range(len(m.layers[0].states))}
        return new, in_place, {"state_maps": {0: smap}, "transition_maps": {0: tmap}}

    # grown before the first transition fires (a state added DURING a transition would become the active one, see
    # test_state_added_while_a_transition_is_in_flight_becomes_active), shrunk inside it
    assert _run(cctx, sc, {3: grow, 9: shrink, 31: edit_retime}) == 3


@pytest.mark.parametrize("orc_mod", [oracle, oracle2], ids=["oracle", "oracle2"])
@pytest.mark.parametrize("explicit_entry", [None, 0, 1], ids=["entry_never_set", "entry_0", "entry_1"])
def test_layer_reset_is_the_references(cctx, orc_mod, explicit_entry):
    """MachineLayer::reset (layer.rs:288-296): transitions reset, active_state = entry_state, active_transition untouched.
    add_state makes the first state ACTIVE but does not make it the ENTRY state (layer.rs:229-235 vs :209-212), so on a
    layer whose entry state was never set, reset() leaves no active state: the layer goes quiet unless a transition is
    still active -- both restatements and the product agree, frame by frame."""
    for reset_at in (3, 6, 7, 8, 12, 35):
        sc = cases.transitions()
        sc.machine.layers[0].entry_state = explicit_entry
        o = cases.build_oracle(orc_mod, sc)
        p = cases.build_product(cctx, sc, n_instances=2)
        trs = o.node_trs()
        for f in range(50):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            if f == reset_at:
                o.reset_layer(0)
                p.reset_layer(0)
                if explicit_entry is None:
                    assert p.layer_state(0, 0)[0] == -1
            plan = p.plan(1, sc.dt)
            o.update_machine(sc.dt)
            assert p.layer_state(0, 1) == o.layer_state(0), (reset_at, f)
            poses = [o.animation_pose(a) for a in range(len(sc.animations))]
            o0, o1 = plan["offsets"][:2]
            trs = run_program(oracle, plan["ops"][o0:o1], poses, [set()], trs)
            assert np.array_equal(trs.view(np.uint32), o.node_trs().view(np.uint32)), (reset_at, f)
        o.close()
        p.free()


def test_state_added_while_a_transition_is_in_flight_becomes_active(cctx):
    """layer.rs:229-235 tests active_state alone, and active_state is NONE while a transition runs: a state added then
    becomes the active one under the running transition (which overwrites it when it completes).  The append-only
    builder call does the same per instance; so does a rebuild, for the first state it finds added."""
    sc = cases.transitions()
    for via_rebuild in (False, True):
        o = cases.build_oracle(oracle2, sc)
        p = cases.build_product(cctx, sc, n_instances=2)
        desc, trs, seen = sc.machine, o.node_trs(), False
        for f in range(40):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            if f == 8:
                assert o.layer_state(0) == (-1, 0)                   # inside idle -> walk
                new, in_place, _ = edit_grow(desc)
                in_place(o.machine)
                assert o.layer_state(0) == (3, 0)                    # the added state is active, the transition goes on
                if via_rebuild:
                    p.rebuild_machine(desc, new)
                else:                                                # the same calls the in-place edit made, appended
                    l_new = new.layers[0]
                    p._check(p._l.fyx_layer_add_play_animation(p._h, p.id, 0, l_new.nodes[-1].animation, None))
                    p._check(p._l.fyx_layer_add_state(p._h, p.id, 0, l_new.states[-1].root, None))
                    for kind, anim in l_new.states[-1].on_enter_actions:
                        p._check(p._l.fyx_state_add_action(p._h, p.id, 0, 3, 1, kind, anim))
                    for t in l_new.transitions[len(desc.layers[0].transitions):]:
                        code = A._i32(A.encode_logic(t.condition))
                        p._check(p._l.fyx_layer_add_transition(p._h, p.id, 0, t.source, t.dest, t.transition_time, A._ptr(code), len(code), None))
                desc = new
                assert p.layer_state(0, 0) == (3, 0) and p.layer_state(0, 1) == (3, 0)
                seen = True
            plan = p.plan(1, sc.dt)
            o.update_machine(sc.dt)
            assert p.layer_state(0, 1) == o.layer_state(0), (via_rebuild, f)
            for a in range(len(sc.animations)):
                assert p.animation_state(a, 0) == o.animation_state(a), (via_rebuild, f, a)
            poses = [o.animation_pose(a) for a in range(len(sc.animations))]
            o0, o1 = plan["offsets"][:2]
            trs = run_program(oracle, plan["ops"][o0:o1], poses, [set()], trs)
            assert np.array_equal(trs.view(np.uint32), o.node_trs().view(np.uint32)), (via_rebuild, f)
        assert seen
        o.close()
        p.free()


def test_state_accessors_round_trip_and_validate(cctx):
    sc = cases.by_index()
    p = cases.build_product(cctx, sc, n_instances=3)
    p.set_layer_state(0, -1, -1, instance=1)
    assert p.layer_state(0, 1) == (-1, -1) and p.layer_state(0, 0) == (0, -1)
    p.set_node_state(0, 3, 2, 0.0625, instance=2)
    assert p.node_state(0, 3, 2) == (2, 0.0625) and p.node_state(0, 3, 0) == (None, 0.0)
    p.set_node_state(0, 3, None, 0.0)
    assert p.node_state(0, 3, 2) == (None, 0.0)
    p.set_parameter(0, A.Parameter(A.PARAM_INDEX, 2), instance=1)
    assert p.get_parameter(0, 1) == A.Parameter(A.PARAM_INDEX, 2) and p.get_parameter(0, 0) == A.Parameter(A.PARAM_INDEX, 0)
    for bad in (lambda: p.set_layer_state(0, 1, -1), lambda: p.set_layer_state(0, 0, 0), lambda: p.set_layer_state(1, 0, -1),
                lambda: p.set_node_state(0, 0, 1, 0.0), lambda: p.node_state(0, 9), lambda: p.transition_state(0, 0),
                lambda: p.set_transition_state(0, 0, 0.0, 0.0), lambda: p.get_parameter(1), lambda: p.get_parameter(0, 3),
                lambda: p.set_layer_state(0, 0, -1, instance=3), lambda: p.reset_layer(2)):
        with pytest.raises(fyrox_amd.FyxError):
            bad()
    sc2 = cases.transitions()
    q = cases.build_product(cctx, sc2)
    q.set_transition_state(0, 2, 0.05, 0.5)
    assert q.transition_state(0, 2) == (np.float32(0.05), 0.5) and q.transition_state(0, 0) == (0.0, 0.0)
    q.machine_clear()
    with pytest.raises(fyrox_amd.FyxError):
        q.layer_state(0)
    with pytest.raises(fyrox_amd.FyxError):
        q.set_parameter(0, A.Parameter(A.PARAM_RULE, True))
    q.set_machine(sc2.machine)          # and the animator takes a definition again
    assert q.layer_state(0) == (0, -1)
    q.plan(1, sc2.dt)
    p.free()
    q.free()


def test_sync_machine_re_sends_only_when_the_definition_changed(cctx):
    """The shim's per-frame call: nothing when the game left its Machine alone (parameter VALUES are not the
    definition: Machine::set_parameter travels through fyx_machine_set_parameter), the definition again after an edit."""
    sc = cases.transitions()
    o = cases.build_oracle(oracle2, sc)
    sc_nomachine = copy.copy(sc)
    sc_nomachine.machine = None
    p = cases.build_product(cctx, sc_nomachine, n_instances=1)
    desc = copy.deepcopy(sc.machine)
    assert p.sync_machine(desc) is True                 # first time: sent
    trs, sent = o.node_trs(), 1
    for f in range(40):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        if f in (8, 20):                                # the game edits its machine in place ...
            new, in_place, _ = (edit_retime if f == 8 else edit_recondition)(desc)
            in_place(o.machine)
            desc = new
        sent += p.sync_machine(desc)                    # ... and the shim notices on its own
        plan = p.plan(1, sc.dt)
        o.update_machine(sc.dt)
        assert p.layer_state(0, 0) == o.layer_state(0), f
        poses = [o.animation_pose(a) for a in range(len(sc.animations))]
        o0, o1 = plan["offsets"][:2]
        trs = run_program(oracle, plan["ops"][o0:o1], poses, [set()], trs)
        assert np.array_equal(trs.view(np.uint32), o.node_trs().view(np.uint32)), f
    assert sent == 3
    o.close()
    p.free()


def test_rebuild_carries_every_instances_own_state(cctx):
    """Two instances of one animator in different states (one has walked through a transition, the other is inside
    another one): the definition re-sent between frames, each instance keeps ITS run-time state and ITS parameter values
    -- every instance against its own oracle."""
    sc = cases.transitions()
    scripts = [sc.script, {12: [(0, A.Parameter(A.PARAM_RULE, True))], 26: [(1, A.Parameter(A.PARAM_RULE, True))]}]
    os_ = [cases.build_oracle(oracle2, sc) for _ in scripts]
    p = cases.build_product(cctx, sc, n_instances=2)
    desc = sc.machine
    trs = [o.node_trs() for o in os_]
    states_seen = set()
    for f in range(48):
        for i, script in enumerate(scripts):
            for idx, par in script.get(f, []):
                os_[i].set_parameter(idx, par)
                p.set_parameter(idx, par, instance=i)
        if f in (9, 14, 15, 30):
            new, in_place, _ = (edit_retime if f != 30 else edit_grow)(desc)
            for o in os_:
                in_place(o.machine)
            for li in range(len(desc.layers)):
                for i, o in enumerate(os_):
                    assert _drain(lambda: p.pop_layer_event(li, i)) == _drain(lambda: o.pop_layer_event(li))
            p.rebuild_machine(desc, new)
            desc = new
        plan = p.plan(1, sc.dt)
        offs = plan["offsets"]
        for i, o in enumerate(os_):
            o.update_machine(sc.dt)
            assert p.layer_state(0, i) == o.layer_state(0), (f, i)
            for a in range(len(sc.animations)):
                assert p.animation_state(a, i) == o.animation_state(a), (f, i, a)
            poses = [o.animation_pose(a) for a in range(len(sc.animations))]
            trs[i] = run_program(oracle, plan["ops"][offs[i]:offs[i + 1]], poses, [set()], trs[i])
            assert np.array_equal(trs[i].view(np.uint32), o.node_trs().view(np.uint32)), (f, i)
        states_seen.add((p.layer_state(0, 0), p.layer_state(0, 1)))
    assert any(a != b for a, b in states_seen)          # the instances really were in different states
    for o in os_:
        o.close()
    p.free()


def test_layer_events_queued_at_the_time_of_an_edit_are_not_lost(cctx):
    """The reference's event queue is a field of the layer and survives any edit of the layer; behind the C ABI the
    queue would go with fyx_machine_clear, so rebuild_machine takes the events out first and serves them again --
    with the indices the re-sent definition gives their states and transitions (the last rebuild reorders the pools)."""
    sc = cases.transitions()
    o = cases.build_oracle(oracle2, sc)
    p = cases.build_product(cctx, sc, n_instances=2)
    desc, kept_seen = sc.machine, 0
    inv_s = inv_t = None
    for f in range(45):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        if f in (6, 16, 36):
            new, in_place, maps = (edit_retime if f == 6 else edit_recondition if f == 16 else edit_permute)(desc)
            in_place(o.machine)
            p.rebuild_machine(desc, new, **maps)
            kept_seen += sum(len(v) for v in p._kept_layer_events.values())
            if maps:
                inv_s, inv_t = _inverse(maps["state_maps"], 0), _inverse(maps["transition_maps"], 0)
            desc = new
        p.plan(1, sc.dt)
        o.update_machine(sc.dt)
    ref = _drain(lambda: o.pop_layer_event(0))
    assert len(ref) >= 8 and kept_seen >= 6          # the script fired transitions before every one of the edits
    for inst in range(2):
        assert _map_events(_drain(lambda: p.pop_layer_event(0, inst)), inv_s, inv_t) == ref
    o.close()
    p.free()
