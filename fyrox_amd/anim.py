"""Host-side mirror of the reference's animation interface for the pose path.

Plain descriptions named after the reference's types (fyrox-math Curve/CurveKey, fyrox-animation
Track / AnimationTracksData / Animation / Machine / MachineLayer / State / Transition / PoseNode,
fyrox-impl Transform) plus `Animator`, a thin wrapper over the C ABI (include/fyrox_hip.h, second
half).  Nothing here computes a pose: descriptions are flattened into the arrays the C ABI takes and
every per-bone operation runs in the HIP kernels of libfyrox_hip.so.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, byref, c_float, c_int, c_int32, c_uint8, c_uint32, c_void_p
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

# ValueBinding / TrackValueKind / CurveKeyKind (values of include/fyrox_hip.h)
BIND_POSITION, BIND_SCALE, BIND_ROTATION = 0, 1, 2
BIND_PROPERTY0 = 3   # ValueBinding::Property{name, ..}: BIND_PROPERTY0 + id (tracks of every TrackValueKind)
KIND_REAL, KIND_VEC2, KIND_VEC3, KIND_VEC4, KIND_QUAT_EULER, KIND_QUAT = range(6)
KEY_CONSTANT, KEY_LINEAR, KEY_CUBIC = 0, 1, 2
PARAM_WEIGHT, PARAM_RULE, PARAM_INDEX, PARAM_SAMPLING_POINT = range(4)
ACTION_NONE, ACTION_REWIND, ACTION_ENABLE, ACTION_DISABLE, ACTION_ENABLE_RANDOM = range(5)   # ENABLE_RANDOM: (kind, [handles])
LOGIC_PARAMETER, LOGIC_AND, LOGIC_OR, LOGIC_XOR, LOGIC_NOT, LOGIC_IS_ANIMATION_ENDED = range(6)
ALL_INSTANCES = 0xFFFFFFFF
READ_LOCAL_TRS, READ_LOCAL_MATRIX, READ_GLOBAL_MATRIX, READ_ANIMATION_POSE = 0, 1, 2, 16
READ_ANIMATION_BLEND_VIEW = 65536      # + animation: what a blend reads of the animation's pose (include/fyrox_hip.h)
OP_NAMES = ("END", "BLEND_ANIM", "PUSH", "POP_BLEND", "RESET", "MASK", "APPLY", "APPLY_ANIM")
RM_OP_NAMES = ("END", "SET_ANIM", "BLEND", "COPY")
EVENTS_ALL, EVENTS_MAX_WEIGHT, EVENTS_MIN_WEIGHT = range(3)
EVENT_STATE_ENTER, EVENT_STATE_LEAVE, EVENT_ACTIVE_STATE_CHANGED, EVENT_ACTIVE_TRANSITION_CHANGED = range(4)


# ---- curves / tracks (fyrox-math/src/curve.rs, fyrox-animation/src/track.rs, container.rs) ----------

@dataclass
class CurveKey:
    location: float
    value: float
    kind: int = KEY_LINEAR
    left_tangent: float = 0.0   # CurveKeyKind::Cubic only
    right_tangent: float = 0.0


class Curve:
    """Keys kept sorted by location (stable), as Curve::from / add_key do (curve.rs:164-174)."""

    def __init__(self, keys: Sequence[CurveKey] = ()):
        self.keys: List[CurveKey] = sorted(keys, key=lambda k: k.location)

    def __len__(self):
        return len(self.keys)


@dataclass
class Track:
    binding: int
    kind: int
    curves: List[Curve]


@dataclass
class AnimationTracksData:
    tracks: List[Track] = field(default_factory=list)

    def flatten(self):
        """-> (track descriptors, location, value, kind, left_tangent, right_tangent) in the order
        fyx_tracks_data_upload expects (track-major, curve-major)."""
        descs = (TrackDesc * max(len(self.tracks), 1))()
        loc, val, kind, lt, rt = [], [], [], [], []
        for i, t in enumerate(self.tracks):
            assert len(t.curves) <= 4
            descs[i].binding = t.binding
            descs[i].kind = t.kind
            descs[i].n_curves = len(t.curves)
            for c, cv in enumerate(t.curves):
                descs[i].curve_n_keys[c] = len(cv)
                for k in cv.keys:
                    loc.append(k.location); val.append(k.value); kind.append(k.kind)
                    lt.append(k.left_tangent); rt.append(k.right_tangent)
        f = lambda a: np.asarray(a, np.float32)
        return descs, f(loc), f(val), np.asarray(kind, np.uint8), f(lt), f(rt)

    def time_length(self) -> float:
        """TrackDataContainer::time_length over all tracks (what fit_length_to_content uses)."""
        m = 0.0
        for t in self.tracks:
            for c in t.curves:
                if c.keys:
                    m = max(m, c.keys[-1].location)
        return m


class TrackDesc(Structure):
    _fields_ = [("binding", c_int32), ("kind", c_int32), ("n_curves", c_uint32), ("curve_n_keys", c_uint32 * 4)]


# ---- scene::transform::Transform ----------------------------------------------------------------

class Transform(Structure):
    _fields_ = [("local_position", c_float * 3), ("local_rotation", c_float * 4), ("local_scale", c_float * 3),
                ("pre_rotation", c_float * 4), ("post_rotation_matrix", c_float * 9),
                ("rotation_offset", c_float * 3), ("rotation_pivot", c_float * 3),
                ("scaling_offset", c_float * 3), ("scaling_pivot", c_float * 3)]

    @staticmethod
    def identity() -> "Transform":
        t = Transform()
        t.local_rotation[:] = (0, 0, 0, 1)
        t.local_scale[:] = (1, 1, 1)
        t.pre_rotation[:] = (0, 0, 0, 1)
        t.post_rotation_matrix[:] = (1, 0, 0, 0, 1, 0, 0, 0, 1)
        return t


@dataclass
class Rig:
    parent: np.ndarray                    # (n,) int32, parent[i] < i or -1
    transforms: List[Transform]
    inv_bind: Optional[np.ndarray] = None  # (n,16) column-major, None = identity

    @property
    def n_nodes(self) -> int:
        return len(self.transforms)


# ---- Machine (fyrox-animation/src/machine) --------------------------------------------------------

@dataclass
class Parameter:
    kind: int
    value: Union[float, bool, int, Tuple[float, float]] = 0.0

    def packed(self):
        if self.kind == PARAM_WEIGHT:
            return float(self.value), 0.0, 0
        if self.kind == PARAM_RULE:
            return 0.0, 0.0, 1 if self.value else 0
        if self.kind == PARAM_INDEX:
            return 0.0, 0.0, int(self.value)
        return float(self.value[0]), float(self.value[1]), 0


@dataclass
class PlayAnimation:
    animation: int


@dataclass
class BlendPose:
    pose_source: int
    weight: float = 0.0               # PoseWeight::Constant
    parameter: Optional[int] = None   # PoseWeight::Parameter


@dataclass
class BlendAnimations:
    pose_sources: List[BlendPose]


@dataclass
class IndexedBlendInput:
    blend_time: float
    pose_source: int


@dataclass
class BlendAnimationsByIndex:
    index_parameter: int
    inputs: List[IndexedBlendInput]


@dataclass
class BlendSpacePoint:
    position: Tuple[float, float]
    pose_source: int


@dataclass
class BlendSpace:
    sampling_parameter: int
    points: List[BlendSpacePoint]
    triangles: List[Tuple[int, int, int]] = field(default_factory=list)


PoseNode = Union[PlayAnimation, BlendAnimations, BlendAnimationsByIndex, BlendSpace]


@dataclass
class State:
    root: int
    on_enter_actions: List[Tuple[int, int]] = field(default_factory=list)  # (ACTION_*, animation)
    on_leave_actions: List[Tuple[int, int]] = field(default_factory=list)


@dataclass
class Transition:
    source: int
    dest: int
    transition_time: float
    condition: tuple = ("parameter", -1)   # ("parameter", p) | ("and"|"or"|"xor", a, b) | ("not", a) | ("ended", anim)


@dataclass
class MachineLayer:
    nodes: List[PoseNode] = field(default_factory=list)
    states: List[State] = field(default_factory=list)
    transitions: List[Transition] = field(default_factory=list)
    weight: float = 1.0
    entry_state: Optional[int] = None
    mask: List[int] = field(default_factory=list)


@dataclass
class Machine:
    parameters: List[Parameter] = field(default_factory=list)
    layers: List[MachineLayer] = field(default_factory=list)


def encode_logic(cond) -> List[int]:
    """LogicNode tree -> prefix int code (FYX_LOGIC_*)."""
    op = cond[0]
    if op == "parameter":
        return [LOGIC_PARAMETER, int(cond[1])]
    if op == "ended":
        return [LOGIC_IS_ANIMATION_ENDED, int(cond[1])]
    if op == "not":
        return [LOGIC_NOT] + encode_logic(cond[1])
    code = {"and": LOGIC_AND, "or": LOGIC_OR, "xor": LOGIC_XOR}[op]
    return [code] + encode_logic(cond[1]) + encode_logic(cond[2])


# ---- C ABI wrapper --------------------------------------------------------------------------------

def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


PROPERTY_VALUE = np.dtype([("value", np.float32, (4,)), ("present", np.uint32), ("kind", np.uint32), ("reserved", np.uint32, (2,))])
VALUE_REAL, VALUE_VEC2, VALUE_VEC3, VALUE_VEC4, VALUE_QUAT = 0, 1, 2, 3, 4   # FYX_VALUE_*: TrackValue variants


class Animator:
    """n_instances copies of one rig + its AnimationPlayer animations (+ optionally a Machine)."""

    def __init__(self, ctx, animator_id: int, rig_id: int, rig: Rig, n_instances: int = 1):
        self.ctx, self.id, self.rig_id, self.rig, self.n_instances = ctx, animator_id, rig_id, rig, n_instances
        self._l, self._h = ctx._l, ctx._h
        self.n_animations = 0
        ctx._check(self._l.fyx_animator_create(self._h, animator_id, rig_id, n_instances))

    def _check(self, rc):
        self.ctx._check(rc)

    # -- animations ------------------------------------------------------------------------------
    def add_animation(self, tracks_id: int, track_target, track_enabled=None, *, time_slice=None, speed=None,
                      looped=None, enabled=None) -> int:
        tgt = _i32(track_target)
        en = None if track_enabled is None else np.ascontiguousarray(track_enabled, dtype=np.uint8)
        out = c_uint32()
        self._check(self._l.fyx_animator_add_animation(self._h, self.id, tracks_id, _ptr(tgt), _ptr(en), byref(out)))
        a = out.value
        self.n_animations = a + 1
        if looped is not None:
            self.set_loop(a, looped)
        if time_slice is not None:
            self.set_time_slice(a, *time_slice)
        if speed is not None:
            self.set_speed(a, speed)
        if enabled is not None:
            self.set_enabled(a, enabled)
        return a

    def remove_animation(self, a: int) -> None:
        """AnimationContainer::remove: the index no longer resolves (see fyx_animator_remove_animation)."""
        self._check(self._l.fyx_animator_remove_animation(self._h, self.id, a))

    def set_time_slice(self, a, start, end, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_set_time_slice(self._h, self.id, a, instance, start, end))

    def set_time_position(self, a, t, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_set_time_position(self._h, self.id, a, instance, t))

    def set_speed(self, a, s, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_set_speed(self._h, self.id, a, instance, s))

    def set_loop(self, a, looped, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_set_loop(self._h, self.id, a, instance, int(bool(looped))))

    def set_enabled(self, a, enabled, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_set_enabled(self._h, self.id, a, instance, int(bool(enabled))))

    def rewind(self, a, instance=ALL_INSTANCES):
        self._check(self._l.fyx_animation_rewind(self._h, self.id, a, instance))

    def set_track_enabled(self, a, track, enabled):
        self._check(self._l.fyx_animation_set_track_enabled(self._h, self.id, a, track, int(bool(enabled))))

    def animation_state(self, a, instance=0) -> dict:
        t, e, d = c_float(), c_int(), c_int()
        self._check(self._l.fyx_animation_get_state(self._h, self.id, a, instance, byref(t), byref(e), byref(d)))
        return {"time_position": t.value, "enabled": bool(e.value), "has_ended": bool(d.value)}

    # -- signals / events ------------------------------------------------------------------------
    def add_signal(self, a: int, time: float, enabled: bool = True) -> int:
        """Animation::add_signal; the returned index stands for the signal's {Uuid, name}."""
        out = c_uint32()
        self._check(self._l.fyx_animation_add_signal(self._h, self.id, a, time, int(bool(enabled)), byref(out)))
        return out.value

    def set_signal_enabled(self, a: int, signal: int, enabled: bool) -> None:
        self._check(self._l.fyx_animation_set_signal_enabled(self._h, self.id, a, signal, int(bool(enabled))))

    def set_max_event_capacity(self, a: int, capacity: int, instance=ALL_INSTANCES) -> None:
        self._check(self._l.fyx_animation_set_max_event_capacity(self._h, self.id, a, instance, capacity))

    def pop_event(self, a: int, instance: int = 0) -> Optional[int]:
        """Animation::pop_event -> signal index or None."""
        out = c_int32()
        self._check(self._l.fyx_animation_pop_event(self._h, self.id, a, instance, byref(out)))
        return None if out.value < 0 else out.value

    def event_count(self, a: int, instance: int = 0) -> int:
        out = c_uint32()
        self._check(self._l.fyx_animation_event_count(self._h, self.id, a, instance, byref(out)))
        return out.value

    def clear_events(self, a: int, instance=ALL_INSTANCES) -> None:
        self._check(self._l.fyx_animation_clear_events(self._h, self.id, a, instance))

    def pop_layer_event(self, layer: int, instance: int = 0) -> Optional[Tuple[int, int, int]]:
        """MachineLayer::pop_event -> (kind, a, b) or None.  Events that were still queued when the definition was re-sent
        (rebuild_machine keeps them, indices translated) come first: an edit loses none."""
        kept = getattr(self, "_kept_layer_events", {}).get((layer, instance))
        if kept:
            return kept.pop(0)
        ev = (c_int32 * 3)()
        has = c_int()
        self._check(self._l.fyx_layer_pop_event(self._h, self.id, layer, instance, ev, byref(has)))
        return (ev[0], ev[1], ev[2]) if has.value else None

    def collect_active_animations_events(self, layer: int, strategy: int = 0, instance: int = 0):
        """MachineLayer::collect_active_animations_events -> (source tuple, [(animation, signal), ...])."""
        cap = 256
        ev = np.zeros((cap, 2), np.int32)
        n = c_uint32()
        src = (c_int32 * 4)()
        self._check(self._l.fyx_layer_collect_active_animations_events(self._h, self.id, layer, instance, strategy, _ptr(ev),
                                                                       cap, byref(n), src))
        assert n.value <= cap
        return tuple(src), [(int(a), int(s_)) for a, s_ in ev[:n.value]]

    # -- root motion -----------------------------------------------------------------------------
    def set_root_motion_settings(self, a: int, node: Optional[int], ignore_x=False, ignore_y=False, ignore_z=False,
                                 ignore_rotations=False) -> None:
        """Animation::set_root_motion_settings (None clears them)."""
        self._check(self._l.fyx_animation_set_root_motion_settings(
            self._h, self.id, a, -1 if node is None else int(node), int(bool(ignore_x)), int(bool(ignore_y)),
            int(bool(ignore_z)), int(bool(ignore_rotations))))

    def track_root_motion(self, enabled: bool = True) -> None:
        self._check(self._l.fyx_animator_track_root_motion(self._h, self.id, int(bool(enabled))))

    def animation_root_motion(self, a: int) -> np.ndarray:
        """(n_instances, 8) float32 view of fyx_root_motion: dp xyz, has (u32 bits), dr ijkw."""
        out = np.zeros((self.n_instances, 8), np.float32)
        self._check(self._l.fyx_animation_read_root_motion(self._h, self.id, a, _ptr(out)))
        return out

    def machine_root_motion(self, layer: int = -1) -> np.ndarray:
        out = np.zeros((self.n_instances, 8), np.float32)
        self._check(self._l.fyx_absm_read_root_motion(self._h, self.id, layer, _ptr(out)))
        return out

    def plan_root_motion(self) -> dict:
        """Test hook: the root-motion program of the frame plan() planned last."""
        off = np.zeros(self.n_instances + 1, np.uint32)
        cap = max(4096, 128 * self.n_instances)
        ops = np.zeros((cap, 4), np.uint32)
        n, ns = c_uint32(), c_uint32()
        slices = np.zeros((self.n_instances, max(self.n_animations, 1), 2), np.float32)
        self._check(self._l.fyx_animator_plan_root_motion(self._h, self.id, _ptr(off), _ptr(ops), cap, byref(n),
                                                          byref(ns), _ptr(slices)))
        if n.value > cap:
            raise RuntimeError("program larger than the wrapper's buffer; plan_root_motion() is a test hook")
        return {"offsets": off, "ops": ops[:n.value], "n_slots": ns.value, "slices": slices[:, :self.n_animations]}

    # -- Property{..} slots ------------------------------------------------------------------------
    def property_count(self) -> int:
        out = c_uint32()
        self._check(self._l.fyx_animator_property_count(self._h, self.id, byref(out)))
        return out.value

    def property_slot(self, node: int, property_id: int) -> int:
        out = c_int32()
        self._check(self._l.fyx_animator_property_slot(self._h, self.id, node, property_id, byref(out)))
        return out.value

    def set_random_seed(self, seed: int, instance: int = ALL_INSTANCES) -> None:
        """State of the EnableRandomAnimation generator: `seed` for one instance; for all, seed + (i + 1) * golden."""
        self._check(self._l.fyx_animator_set_random_seed(self._h, self.id, instance, ctypes.c_uint64(seed & (2 ** 64 - 1))))

    def read_properties(self, animation: int = -1) -> np.ndarray:
        """(n_instances, n_slots) records of PROPERTY_VALUE (fyx_property_value): value[4], present, kind.
        animation < 0: applied values."""
        out = np.zeros((self.n_instances, self.property_count()), PROPERTY_VALUE)
        self._check(self._l.fyx_animator_read_properties(self._h, self.id, animation, _ptr(out)))
        return out

    def blend_shape_weights(self, slots, default_weights, d_out: int) -> None:
        sl = _i32(slots)
        dw = np.ascontiguousarray(default_weights, dtype=np.float32)
        assert len(sl) == len(dw)
        self._check(self._l.fyx_animator_blend_shape_weights(self._h, self.id, len(sl), _ptr(sl), _ptr(dw), d_out))

    # -- machine ---------------------------------------------------------------------------------
    def set_machine(self, m: Machine) -> None:
        L = self._l
        for p in m.parameters:
            f0, f1, u = p.packed()
            self._check(L.fyx_machine_add_parameter(self._h, self.id, p.kind, f0, f1, u, None))
        for layer in m.layers:
            li = c_uint32()
            self._check(L.fyx_machine_add_layer(self._h, self.id, layer.weight, byref(li)))
            li = li.value
            if layer.mask:
                mk = _i32(layer.mask)
                self._check(L.fyx_layer_set_mask(self._h, self.id, li, _ptr(mk), len(mk)))
            for n in layer.nodes:
                if isinstance(n, PlayAnimation):
                    self._check(L.fyx_layer_add_play_animation(self._h, self.id, li, n.animation, None))
                elif isinstance(n, BlendAnimations):
                    src = _i32([b.pose_source for b in n.pose_sources])
                    par = _i32([-1 if b.parameter is None else b.parameter for b in n.pose_sources])
                    wc = np.asarray([b.weight for b in n.pose_sources], np.float32)
                    self._check(L.fyx_layer_add_blend_animations(self._h, self.id, li, len(src), _ptr(src), _ptr(par),
                                                                 _ptr(wc), None))
                elif isinstance(n, BlendAnimationsByIndex):
                    src = _i32([i.pose_source for i in n.inputs])
                    bt = np.asarray([i.blend_time for i in n.inputs], np.float32)
                    self._check(L.fyx_layer_add_blend_animations_by_index(self._h, self.id, li, n.index_parameter,
                                                                          len(src), _ptr(src), _ptr(bt), None))
                elif isinstance(n, BlendSpace):
                    pts = np.asarray([p.position for p in n.points], np.float32).reshape(-1, 2)
                    src = _i32([p.pose_source for p in n.points])
                    tri = np.asarray(n.triangles, np.uint32).reshape(-1, 3)
                    self._check(L.fyx_layer_add_blend_space(self._h, self.id, li, n.sampling_parameter, len(src),
                                                            _ptr(pts), _ptr(src), len(tri), _ptr(tri), None))
                else:
                    raise TypeError(n)
            for si, s in enumerate(layer.states):
                self._check(L.fyx_layer_add_state(self._h, self.id, li, s.root, None))
                for on_enter, actions in ((1, s.on_enter_actions), (0, s.on_leave_actions)):
                    for kind, anim in actions:
                        if kind == ACTION_ENABLE_RANDOM:
                            ch = np.asarray(anim, np.int64).astype(np.uint32)   # -1 = Handle::NONE -> an invalid index
                            self._check(L.fyx_state_add_random_action(self._h, self.id, li, si, on_enter, _ptr(ch), len(ch)))
                        else:
                            self._check(L.fyx_state_add_action(self._h, self.id, li, si, on_enter, kind, anim))
            for t in layer.transitions:
                code = _i32(encode_logic(t.condition))
                self._check(L.fyx_layer_add_transition(self._h, self.id, li, t.source, t.dest, t.transition_time,
                                                       _ptr(code), len(code), None))
            if layer.entry_state is not None:
                self._check(L.fyx_layer_set_entry_state(self._h, self.id, li, layer.entry_state))

    def set_parameter(self, index: int, p: Parameter, instance=ALL_INSTANCES) -> None:
        f0, f1, u = p.packed()
        self._check(self._l.fyx_machine_set_parameter(self._h, self.id, index, instance, p.kind, f0, f1, u))

    def layer_state(self, layer: int, instance: int = 0) -> Tuple[int, int]:
        s, t = c_int32(), c_int32()
        self._check(self._l.fyx_layer_get_state(self._h, self.id, layer, instance, byref(s), byref(t)))
        return s.value, t.value

    # -- run-time edits: the definition is re-sent, the run-time state carried over (fyrox_hip.h) -------------
    def machine_clear(self) -> None:
        self._check(self._l.fyx_machine_clear(self._h, self.id))

    def get_parameter(self, index: int, instance: int = 0) -> Parameter:
        k, f0, f1, u = c_int(), c_float(), c_float(), c_uint32()
        self._check(self._l.fyx_machine_get_parameter(self._h, self.id, index, instance, byref(k), byref(f0), byref(f1), byref(u)))
        if k.value == PARAM_WEIGHT:
            return Parameter(PARAM_WEIGHT, f0.value)
        if k.value == PARAM_RULE:
            return Parameter(PARAM_RULE, bool(u.value))
        if k.value == PARAM_INDEX:
            return Parameter(PARAM_INDEX, int(u.value))
        return Parameter(PARAM_SAMPLING_POINT, (f0.value, f1.value))

    def set_layer_state(self, layer: int, active_state: int, active_transition: int, instance=ALL_INSTANCES) -> None:
        self._check(self._l.fyx_layer_set_state(self._h, self.id, layer, instance, active_state, active_transition))

    def transition_state(self, layer: int, transition: int, instance: int = 0) -> Tuple[float, float]:
        e, b = c_float(), c_float()
        self._check(self._l.fyx_layer_get_transition_state(self._h, self.id, layer, instance, transition, byref(e), byref(b)))
        return e.value, b.value

    def set_transition_state(self, layer: int, transition: int, elapsed: float, blend_factor: float, instance=ALL_INSTANCES) -> None:
        self._check(self._l.fyx_layer_set_transition_state(self._h, self.id, layer, instance, transition, elapsed, blend_factor))

    def node_state(self, layer: int, node: int, instance: int = 0) -> Tuple[Optional[int], float]:
        """BlendAnimationsByIndex: (prev_index or None, blend_time)"""
        h, p, t = c_int(), c_uint32(), c_float()
        self._check(self._l.fyx_layer_get_node_state(self._h, self.id, layer, instance, node, byref(h), byref(p), byref(t)))
        return (p.value if h.value else None), t.value

    def set_node_state(self, layer: int, node: int, prev_index: Optional[int], blend_time: float, instance=ALL_INSTANCES) -> None:
        self._check(self._l.fyx_layer_set_node_state(self._h, self.id, layer, instance, node, 0 if prev_index is None else 1,
                                                     0 if prev_index is None else prev_index, blend_time))

    def reset_layer(self, layer: int, instance=ALL_INSTANCES) -> None:
        """MachineLayer::reset (layer.rs:288-296)"""
        self._check(self._l.fyx_layer_reset(self._h, self.id, layer, instance))

    def sync_machine(self, m: Machine) -> bool:
        """What the shim calls every frame before the update: the definition again if it differs from the one sent last
        (the game edited its Machine in place), run-time state carried over.  True when it re-sent.  Descriptions here are
        dense lists, so indices keep their meaning (identity maps); the Rust shim compares `machine_signature`s and maps
        through its handle tables."""
        import copy
        sent = getattr(self, "_sent_machine", None)
        if sent is None:
            self.set_machine(m)
        elif sent != m:
            self.rebuild_machine(sent, m)
        else:
            return False
        self._sent_machine = copy.deepcopy(m)
        return True

    def rebuild_machine(self, old: Machine, new: Machine, layer_map=None, state_maps=None, transition_maps=None,
                        node_maps=None, parameter_map=None) -> None:
        """What the engine-side shim does after the game has edited its Machine in place: read the run-time state of
        every instance, clear, re-send the definition, put the state back.  The maps translate OLD indices to NEW ones
        (the shim derives them from its handle -> index tables): `layer_map[old_layer]`, `state_maps[old_layer][old_state]`
        ...; a missing map is the identity, an entry of None (or an index past the new definition) means the item is gone
        and its state is dropped -- an active state that is gone becomes Handle::NONE."""
        def m(maps, key, i, limit):
            if i is None or i < 0:
                return -1
            mp = None if maps is None else (maps.get(key) if isinstance(maps, dict) else maps[key])
            j = i if mp is None else (mp.get(i) if isinstance(mp, dict) else (mp[i] if i < len(mp) else None))
            return -1 if j is None or j < 0 or j >= limit else j

        saved = []
        for inst in range(self.n_instances):
            rec = {"params": [self.get_parameter(p, inst) for p in range(len(old.parameters))], "layers": []}
            for li, layer in enumerate(old.layers):
                rec["layers"].append({
                    "state": self.layer_state(li, inst),
                    "transitions": [self.transition_state(li, t, inst) for t in range(len(layer.transitions))],
                    "nodes": {n: self.node_state(li, n, inst) for n, nd in enumerate(layer.nodes)
                              if isinstance(nd, BlendAnimationsByIndex)}})
            saved.append(rec)
        # the layers' event queues live in the layer objects in the reference and survive any edit; here they would go with
        # fyx_machine_clear, so they are taken out first and kept, with the indices the NEW definition gives their subjects
        kept = getattr(self, "_kept_layer_events", None)
        if kept is None:
            kept = self._kept_layer_events = {}
        moved = {}
        for li_old, layer in enumerate(old.layers):
            li = m({0: layer_map} if layer_map is not None else None, 0, li_old, len(new.layers))
            for inst in range(self.n_instances):
                evs = []
                while True:          # pop_layer_event: the events kept by an earlier rebuild first, then the library's
                    e = self.pop_layer_event(li_old, inst)
                    if e is None:
                        break
                    kind, a, b = e
                    ns, nt = (len(new.layers[li].states), len(new.layers[li].transitions)) if li >= 0 else (0, 0)
                    if kind == EVENT_ACTIVE_TRANSITION_CHANGED:
                        a = m(transition_maps, li_old, a, nt)
                    else:
                        a = m(state_maps, li_old, a, ns)
                        if kind == EVENT_ACTIVE_STATE_CHANGED:
                            b = m(state_maps, li_old, b, ns)
                    evs.append((kind, a, b))
                if li >= 0 and evs:
                    moved[(li, inst)] = evs
        kept.clear()
        kept.update(moved)
        self.machine_clear()
        self.set_machine(new)
        for inst, rec in enumerate(saved):
            for p_old, val in enumerate(rec["params"]):
                p_new = m({0: parameter_map} if parameter_map is not None else None, 0, p_old, len(new.parameters))
                if p_new >= 0:      # Machine::set_parameter replaces the value, kind included (machine/mod.rs:233-245)
                    self.set_parameter(p_new, val, inst)
            for li_old, lrec in enumerate(rec["layers"]):
                li = m({0: layer_map} if layer_map is not None else None, 0, li_old, len(new.layers))
                if li < 0:
                    continue
                nl = new.layers[li]
                s, t = lrec["state"]
                s_new = m(state_maps, li_old, s, len(nl.states))
                if s < 0:
                    # MachineLayer::add_state makes a state active whenever none is (layer.rs:229-235) -- also while a
                    # transition is in flight: the first state this edit ADDED takes the place
                    kept_states = {m(state_maps, li_old, k, len(nl.states)) for k in range(len(old.layers[li_old].states))}
                    added = [k for k in range(len(nl.states)) if k not in kept_states]
                    if added:
                        s_new = added[0]
                self.set_layer_state(li, s_new, m(transition_maps, li_old, t, len(nl.transitions)), inst)
                for t_old, (el, bf) in enumerate(lrec["transitions"]):
                    t_new = m(transition_maps, li_old, t_old, len(nl.transitions))
                    if t_new >= 0:
                        self.set_transition_state(li, t_new, el, bf, inst)
                for n_old, (prev, bt) in lrec["nodes"].items():
                    n_new = m(node_maps, li_old, n_old, len(nl.nodes))
                    if n_new >= 0 and isinstance(nl.nodes[n_new], BlendAnimationsByIndex):
                        self.set_node_state(li, n_new, prev, bt, inst)

    # -- per frame -------------------------------------------------------------------------------
    def update_animations(self, dt: float) -> None:
        """AnimationPlayer::update"""
        self._check(self._l.fyx_animation_player_update(self._h, self.id, dt))

    def update_machine(self, dt: float) -> None:
        """AnimationBlendingStateMachine::update"""
        self._check(self._l.fyx_absm_update(self._h, self.id, dt))

    def update_transforms(self) -> None:
        self._check(self._l.fyx_animator_update_transforms(self._h, self.id))

    def palette(self, bones_id: int, d_out: int) -> None:
        self._check(self._l.fyx_animator_palette(self._h, self.id, bones_id, d_out))

    def set_palette_output(self, bones_id: int, d_out: int) -> None:
        """The update calls write this palette themselves from now on (d_out = 0 unregisters)."""
        self._check(self._l.fyx_animator_set_palette_output(self._h, self.id, bones_id, d_out or None))

    def set_palette_output_pair(self, bones_id: int, d_out: int, d_out_alt: int) -> None:
        """Two buffers for pipelined frames (option anim.overlap): the frames of the two frame streams write one each."""
        self._check(self._l.fyx_animator_set_palette_output_pair(self._h, self.id, bones_id, d_out or None, d_out_alt or None))

    def current_palette(self, bones_id: int) -> int:
        """The buffer of the palette output that the most recent update call wrote."""
        p = c_void_p()
        self._check(self._l.fyx_animator_current_palette(self._h, self.id, bones_id, byref(p)))
        return p.value or 0

    def set_skin_output(self, bones_id: int, mesh_id: int, d_pos: int = 0, d_normal: int = 0, d_tangent: int = 0) -> None:
        """Every update of the animator also skins `mesh_id` with the palette output of `bones_id` (all outputs 0: remove)."""
        self._check(self._l.fyx_animator_set_skin_output(self._h, self.id, bones_id, mesh_id, d_pos or None, d_normal or None, d_tangent or None))

    def set_local_trs(self, node: int, trs, first_instance: int = 0) -> None:
        trs = np.ascontiguousarray(trs, dtype=np.float32).reshape(-1, 10)
        self._check(self._l.fyx_animator_set_local_trs(self._h, self.id, node, first_instance, trs.shape[0], _ptr(trs)))

    def read(self, what: int) -> np.ndarray:
        width = 16 if what in (READ_LOCAL_MATRIX, READ_GLOBAL_MATRIX) else 12
        out = np.empty((self.n_instances, self.rig.n_nodes, width), np.float32)
        self._check(self._l.fyx_animator_read(self._h, self.id, what, _ptr(out)))
        return out

    def device_ptr(self, what: int) -> int:
        p = c_void_p()
        self._check(self._l.fyx_animator_device_ptr(self._h, self.id, what, byref(p)))
        return p.value or 0

    def plan(self, mode: int, dt: float) -> dict:
        """Advance the control plane one frame WITHOUT running kernels; returns what would be sent."""
        na = max(self.n_animations, 1)
        times = np.zeros((self.n_instances, na), np.float32)
        ticked = np.zeros((self.n_instances, na), np.uint8)
        off = np.zeros(self.n_instances + 1, np.uint32)
        cap = max(4096, 128 * self.n_instances)
        ops = np.zeros((cap, 2), np.uint32)
        n = c_uint32()
        self._check(self._l.fyx_animator_plan(self._h, self.id, mode, dt, _ptr(times), _ptr(ticked), _ptr(off),
                                              _ptr(ops), cap, byref(n)))
        if n.value > cap:
            raise RuntimeError("program larger than the wrapper's buffer; plan() is a test hook")
        return {"times": times[:, :self.n_animations], "ticked": ticked[:, :self.n_animations], "offsets": off,
                "ops": ops[:n.value]}

    def free(self) -> None:
        self._check(self._l.fyx_animator_free(self._h, self.id))


def scene_update(ctx, animators: Sequence["Animator"], dt: float) -> None:
    """fyx_scene_update: one frame of every listed animator (machine update where it has a machine, player update
    otherwise) -- same results as updating them one by one, one kernel launch per stage for all of them."""
    ids = np.asarray([a.id for a in animators], np.uint64)
    ctx._check(ctx._l.fyx_scene_update(ctx._h, _ptr(ids) if len(ids) else None, len(ids), dt))


def scene_plan(ctx, animators: Sequence["Animator"], dt: float) -> None:
    """fyx_scene_plan: the host half of scene_update (works on a control-only context); read each animator's frame with
    Animator.plan(-1, 0.0)."""
    ids = np.asarray([a.id for a in animators], np.uint64)
    ctx._check(ctx._l.fyx_scene_plan(ctx._h, _ptr(ids) if len(ids) else None, len(ids), dt))


def scene_tables(ctx, animators: Sequence["Animator"], stage: int) -> np.ndarray:
    """fyx_debug_scene_tables: (n_blocks, 4) uint32 {job, x, y, z} of one stage of the scene launch."""
    ids = np.asarray([a.id for a in animators], np.uint64)
    n = c_uint32()
    ctx._check(ctx._l.fyx_debug_scene_tables(ctx._h, _ptr(ids), len(ids), stage, None, 0, byref(n)))
    out = np.zeros((max(n.value, 1), 4), np.uint32)
    ctx._check(ctx._l.fyx_debug_scene_tables(ctx._h, _ptr(ids), len(ids), stage, _ptr(out), n.value, byref(n)))
    return out[:n.value]


def upload_tracks_data(ctx, tracks_id: int, td: AnimationTracksData) -> None:
    descs, loc, val, kind, lt, rt = td.flatten()
    ctx._check(ctx._l.fyx_tracks_data_upload(ctx._h, tracks_id, len(td.tracks), descs, len(loc), _ptr(loc), _ptr(val),
                                             _ptr(kind), _ptr(lt), _ptr(rt)))


def create_rig(ctx, rig_id: int, rig: Rig) -> None:
    n = rig.n_nodes
    arr = (Transform * n)(*rig.transforms)
    parent = _i32(rig.parent)
    ib = None if rig.inv_bind is None else np.ascontiguousarray(rig.inv_bind, dtype=np.float32).reshape(n, 16)
    ctx._check(ctx._l.fyx_rig_create(ctx._h, rig_id, n, _ptr(parent), arr, _ptr(ib)))


def create_bone_list(ctx, bones_id: int, rig_id: int, bone_nodes) -> None:
    b = _i32(bone_nodes)
    ctx._check(ctx._l.fyx_bone_list_create(ctx._h, bones_id, rig_id, len(b), _ptr(b)))


def curve_simplify(x, y, epsilon: float, max_step: float = float("inf")) -> np.ndarray:
    """Indices of the points the glTF importer keeps (fyx_curve_simplify; gltf/simplify.rs:39-66)."""
    import ctypes
    from . import _native
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    out = np.zeros(max(x.size, 1), np.uint32)
    n = ctypes.c_uint32()
    rc = _native.lib().fyx_curve_simplify(_ptr(x), _ptr(y), x.size, ctypes.c_float(epsilon), ctypes.c_float(max_step), _ptr(out), ctypes.byref(n))
    if rc:
        raise ValueError(f"fyx_curve_simplify -> {rc}")
    return out[:n.value].copy()


def blend_space_triangulate(points_xy) -> np.ndarray:
    """(n_triangles, 3) point indices (fyx_blend_space_triangulate; blendspace.rs:416-447)."""
    import ctypes
    from . import _native
    pts = np.ascontiguousarray(points_xy, np.float32).reshape(-1, 2)
    cap = 4 * pts.shape[0] + 4
    out = np.zeros(3 * cap, np.uint32)
    n = ctypes.c_uint32()
    rc = _native.lib().fyx_blend_space_triangulate(_ptr(pts), pts.shape[0], _ptr(out), cap, ctypes.byref(n))
    if rc:
        raise ValueError(f"fyx_blend_space_triangulate -> {rc}")
    return out[:3 * n.value].reshape(-1, 3).copy()


def decode_ops(ops: np.ndarray) -> List[Tuple[str, int, float]]:
    out = []
    for x, y in ops:
        out.append((OP_NAMES[int(x) & 0xFF], int(x) >> 8, float(np.uint32(y).view(np.float32))))
    return out
